// gemv_q80.hip — Q8_0 (bit-exact) decode GEMV, f32 activations (quantised in-kernel); kernels in gemv_impl.h
#include "gemv_impl.h"

hipError_t lfamd_gemv_go_q80_f32(int nc, const q80_mats &qm, long n, long k, const void *B, size_t brb, long col0, int vregs32,
                                 int precise, hipStream_t s) {
    hipError_t e = hipSuccess;
    DISPATCH_NC(launch_q80, LFAMD_TYPE_F32, nc, qm, n, k, B, brb, col0, vregs32, precise, s)
    return e;
}
