// gemv_q5k.hip — Q5_K instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(q5k, q5k_traits, LFAMD_TYPE_Q8_K)
GEMV_INSTANTIATE_IDS(q5k, q5k_traits)
GEMV_INSTANTIATE_IDS_PAIR(q5k, q5k_traits)
