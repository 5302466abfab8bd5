// gemv_iq4xs.hip — IQ4_XS instantiations of the decode GEMV (gemv_impl.h)
#include "gemv_impl.h"

GEMV_INSTANTIATE(iq4xs, unused, LFAMD_TYPE_Q8_K)
