// Second half of the persistent kernel's compiled forms (gemm_f32p.hip: CN_P_FORMS_B): a translation unit of its own so that
// the two halves compile side by side.
#define CN_P_UNIT_B
#include "gemm_f32p.hip"
