// gemm_wide_misc.hip — instantiations of the 128x128 MFMA body (gemm_wide_impl.h) for one group of weight types
#include "gemm_wide_impl.h"

WIDE_INSTANTIATE(iq4xs, LFAMD_TYPE_IQ4_XS)
WIDE_INSTANTIATE(f16, LFAMD_TYPE_F16)
WIDE_INSTANTIATE(bf16, LFAMD_TYPE_BF16)
