// Third part of the persistent kernel's compiled forms (gemm_f32p.hip: CN_P_FORMS_C), compiled side by side with the others.
#define CN_P_UNIT_C
#include "gemm_f32p.hip"
