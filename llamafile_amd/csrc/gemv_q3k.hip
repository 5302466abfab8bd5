// gemv_q3k.hip — Q3_K instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q3k, unused, LFAMD_TYPE_Q8_K)
