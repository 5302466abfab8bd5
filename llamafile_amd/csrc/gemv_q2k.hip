// gemv_q2k.hip — Q2_K instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q2k, unused, LFAMD_TYPE_Q8_K)
