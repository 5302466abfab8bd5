// gemm_wide_k4.hip — instantiations of the 128x128 MFMA body (gemm_wide_impl.h) for one group of weight types
#include "gemm_wide_impl.h"

WIDE_INSTANTIATE_MOE(q4k, LFAMD_TYPE_Q4_K)
