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
extern "C" int cartnet_abi_version(void) { return 12; }

// sizeof of the structs that cross the ABI, in the order of the header: a binding in another language (ctypes here)
// compares them with its own mirrors at load time instead of finding a layout mismatch as a GPU fault.
extern "C" int cartnet_abi_struct_sizes(size_t* out, int32_t capacity) {
  const size_t sizes[] = {sizeof(CartnetGemmArgs), sizeof(CartnetShard),       sizeof(CartnetCollated),
                          sizeof(CartnetGemmProfile), sizeof(CartnetGroups),   sizeof(CartnetLayerParams),
                          sizeof(CartnetLayerBuffers), sizeof(CartnetParams),  sizeof(CartnetModel),
                          sizeof(CartnetBatch),        sizeof(CartnetGateGemmArgs),
                          sizeof(CartnetIcfConv),      sizeof(CartnetIcfParams),    sizeof(CartnetIcfModel)};
  const int n = (int)(sizeof(sizes) / sizeof(sizes[0]));
  for (int i = 0; i < n && i < capacity; ++i) out[i] = sizes[i];
  return n;
}
