// gemv_q40.hip — Q4_0 instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q40, q40_traits, LFAMD_TYPE_Q8_0)
