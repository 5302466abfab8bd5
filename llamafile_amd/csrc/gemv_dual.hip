// gemv_dual.hip — two-type instantiations of the decode GEMV (gemv_impl.h, gemv_kq_dual_kernel): the K-quant pairs a
// Q4_K_M / Q5_K_M file puts on one activation vector (attn_q/k in Q4_K or Q5_K, attn_v in Q6_K)
#include "gemv_impl.h"

GEMV_INSTANTIATE_DUAL(q4k_q6k, q4k_traits, q6k_traits)
GEMV_INSTANTIATE_DUAL(q5k_q6k, q5k_traits, q6k_traits)
