// gemm_wide_c16.hip — instantiations of the 128x128 MFMA body (gemm_wide_impl.h) for one group of weight types
#include "gemm_wide_impl.h"

WIDE_INSTANTIATE(q2k, LFAMD_TYPE_Q2_K)
WIDE_INSTANTIATE(q3k, LFAMD_TYPE_Q3_K)
