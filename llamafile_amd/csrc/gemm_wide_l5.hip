// gemm_wide_l5.hip — instantiations of the 128x128 MFMA body (gemm_wide_impl.h) for one group of weight types
#include "gemm_wide_impl.h"

WIDE_INSTANTIATE(q50, LFAMD_TYPE_Q5_0)
WIDE_INSTANTIATE(q51, LFAMD_TYPE_Q5_1)
