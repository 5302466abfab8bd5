// gemv_q41.hip — Q4_1 instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q41, unused, LFAMD_TYPE_Q8_1)
