// gemv_q80b.hip — Q8_0 (bit-exact) decode GEMV, activations already in Q8_0 blocks; kernels in gemv_impl.h
#include "gemv_impl.h"

hipError_t lfamd_gemv_go_q80_q80(int nc, const q80_mats &qm, long n, long k, const void *B, size_t brb, long col0, int vregs32,
                                 int precise, hipStream_t s) {
    hipError_t e = hipSuccess;
    DISPATCH_NC(launch_q80, LFAMD_TYPE_Q8_0, nc, qm, n, k, B, brb, col0, vregs32, precise, s)
    return e;
}
