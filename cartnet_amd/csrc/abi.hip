// Error reporting and version for the C ABI (include/cartnet_hip.h).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void cartnet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* cartnet_last_error(void) { return g_err; }
extern "C" int cartnet_abi_version(void) { return 5; }
