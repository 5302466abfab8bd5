// gemm_wide_k5.hip — instantiations of the 128x128 MFMA body (gemm_wide_impl.h) for one group of weight types
#include "gemm_wide_impl.h"

WIDE_INSTANTIATE_MOE(q5k, LFAMD_TYPE_Q5_K)
