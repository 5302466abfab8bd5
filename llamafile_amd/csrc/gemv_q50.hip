// gemv_q50.hip — Q5_0 instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q50, unused, LFAMD_TYPE_Q8_0)
