// gemv_q6k.hip — Q6_K instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q6k, q6k_traits, LFAMD_TYPE_Q8_K)
GEMV_INSTANTIATE_IDS(q6k, q6k_traits)
GEMV_INSTANTIATE_IDS_PAIR(q6k, q6k_traits)
