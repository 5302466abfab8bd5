// gemv_q4k.hip — Q4_K instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q4k, q4k_traits, LFAMD_TYPE_Q8_K)
GEMV_INSTANTIATE_IDS(q4k, q4k_traits)
GEMV_INSTANTIATE_IDS_PAIR(q4k, q4k_traits)
