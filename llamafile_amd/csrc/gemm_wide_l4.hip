// gemm_wide_l4.hip — instantiations of the 128x128 MFMA body (gemm_wide_impl.h) for one group of weight types
#include "gemm_wide_impl.h"

WIDE_INSTANTIATE(q40, LFAMD_TYPE_Q4_0)
WIDE_INSTANTIATE(q41, LFAMD_TYPE_Q4_1)
