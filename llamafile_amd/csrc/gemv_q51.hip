// gemv_q51.hip — Q5_1 instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q51, unused, LFAMD_TYPE_Q8_1)
