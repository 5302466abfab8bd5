// batched.hip — F16 strided-batched / pointer-array GEMM for the attention products KQ and KQV.
//
// Interface of the reference's tinyblasGemmStridedBatchedEx / tinyblasGemmBatchedEx (llamafile/tinyblas.h:59-71; kernels
// llamafile/tinyblas.cu:141-226, 652-857) for the ONE operand arrangement ggml uses them with (ggml_cuda_mul_mat_batched_cublas,
// ggml-cuda.cu.patch:18231-18376): transa = T, transb = N, f16 operands, f16 or f32 result —
//     C_b[j * ldc + i] = alpha * sum_l A_b[i * lda + l] * B_b[j * ldb + l] + beta * C_b[j * ldc + i]
// i.e. per batch exactly llamafile_sgemm's C = A^T B with both operands' k contiguous.  (beta is only read when nonzero,
// like the reference: tinyblas.cu:212-222.)
//
// Not a translation of the reference's 16 x 16 thread-tile kernel: one wave = one 32 x 32 output tile on
// v_mfma_f32_32x32x16_f16 (f32 accumulate — the reference's COMPUTE_32F; a COMPUTE_16F request is served the same way,
// which is at least as accurate), both fragments loaded straight from global memory — k is contiguous in both operands,
// so lane (row = lane & 31, half = lane >> 5) reads its eight halves of a K-step as one 16-byte load.  Four waves per
// work-group = a 64 x 64 tile; batches are grid.z.  Attention shapes are small and latency-bound (k = 128 for KQ); the
// LDS-tiled bodies of the weight GEMMs would not pay here.
#include "lfamd_device.h"
#include "../../include/lfamd_hip.h"

extern "C" void lfamd_set_error(const char *msg);

namespace {

struct bat_args {
    long m, n, k;
    float alpha, beta;
    const uint8_t *A, *B;
    uint8_t *C;
    const void *const *Aarr, *const *Barr; // pointer-array form when non-null (device arrays of `batch` pointers)
    void *const *Carr;
    long lda, ldb, ldc;
    long long sa, sb, sc; // strides in ELEMENTS
};

// eight halves of row `row` starting at k0, zero past k; ALIGNED: one 16-byte load
template <bool ALIGNED>
__device__ static inline half8_t load8(const _Float16 *base, long ld, long row, long k0, long k) {
    const _Float16 *p = base + row * ld + k0;
    if constexpr (ALIGNED) {
        if (k0 + 8 <= k)
            return *(const half8_t *)p;
    }
    half8_t v;
#pragma unroll
    for (int e = 0; e < 8; e++)
        v[e] = k0 + e < k ? p[e] : (_Float16)0;
    return v;
}

template <bool ALIGNED, bool C_F32>
__global__ __launch_bounds__(256) void gemm_batched_f16_kernel(const bat_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const long b = blockIdx.z;
    const _Float16 *A = a.Aarr ? (const _Float16 *)a.Aarr[b] : (const _Float16 *)a.A + b * a.sa;
    const _Float16 *B = a.Barr ? (const _Float16 *)a.Barr[b] : (const _Float16 *)a.B + b * a.sb;
    const long m0 = (long)blockIdx.x * 64 + (wave & 1) * 32, n0 = (long)blockIdx.y * 64 + (wave >> 1) * 32;
    if (m0 >= a.m || n0 >= a.n)
        return;
    // rows past the edge are clamped for the loads and masked at the store
    const long ar = m0 + i < a.m ? m0 + i : a.m - 1, br = n0 + i < a.n ? n0 + i : a.n - 1;
    float16_t_ acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long k0 = 0; k0 < a.k; k0 += 16) {
        const half8_t fa = load8<ALIGNED>(A, a.lda, ar, k0 + 8 * h, a.k);
        const half8_t fb = load8<ALIGNED>(B, a.ldb, br, k0 + 8 * h, a.k);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0); // A operand = the m index, B operand = the n index
    }
    // lane (i, h) holds column n0 + i; register r holds row m0 + (r & 3) + 8 (r >> 2) + 4 h
    const long col = n0 + i;
    if (col >= a.n)
        return;
    if constexpr (C_F32) {
        float *C = (a.Carr ? (float *)a.Carr[b] : (float *)a.C + b * a.sc) + col * a.ldc;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < a.m) {
                float v = a.alpha * acc[r];
                if (a.beta != 0.0f)
                    v += a.beta * C[row];
                C[row] = v;
            }
        }
    } else {
        _Float16 *C = (a.Carr ? (_Float16 *)a.Carr[b] : (_Float16 *)a.C + b * a.sc) + col * a.ldc;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < a.m) {
                float v = a.alpha * acc[r];
                if (a.beta != 0.0f)
                    v += a.beta * (float)C[row];
                C[row] = (_Float16)v;
            }
        }
    }
}

int launch(const bat_args &a, int Ctype, int batch, bool aligned, hipStream_t s) {
    if (a.m < 0 || a.n < 0 || a.k < 0 || batch < 0 || a.lda < a.k || a.ldb < a.k || a.ldc < a.m ||
        (Ctype != LFAMD_TYPE_F16 && Ctype != LFAMD_TYPE_F32)) {
        lfamd_set_error("lfamd_gemm_*batched_f16: bad dimensions / leading dimensions / result type");
        return LFAMD_ERR_INVALID;
    }
    if (a.m == 0 || a.n == 0 || batch == 0)
        return LFAMD_OK;
    if (batch > 65535) {
        lfamd_set_error("lfamd_gemm_*batched_f16: more than 65535 batches");
        return LFAMD_ERR_UNSUPPORTED;
    }
    const dim3 grid((unsigned)((a.m + 63) / 64), (unsigned)((a.n + 63) / 64), (unsigned)batch);
    const bool f32 = Ctype == LFAMD_TYPE_F32;
    if (aligned && f32)
        gemm_batched_f16_kernel<true, true><<<grid, 256, 0, s>>>(a);
    else if (aligned)
        gemm_batched_f16_kernel<true, false><<<grid, 256, 0, s>>>(a);
    else if (f32)
        gemm_batched_f16_kernel<false, true><<<grid, 256, 0, s>>>(a);
    else
        gemm_batched_f16_kernel<false, false><<<grid, 256, 0, s>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lfamd_set_error(hipGetErrorString(e));
        return LFAMD_ERR_HIP;
    }
    return LFAMD_OK;
}

} // namespace

extern "C" {

int lfamd_gemm_strided_batched_f16(long m, long n, long k, float alpha, const void *d_A, long lda, long long strideA,
                                   const void *d_B, long ldb, long long strideB, float beta, void *d_C, int Ctype, long ldc,
                                   long long strideC, int batch, void *stream) {
    bat_args a{};
    a.m = m, a.n = n, a.k = k, a.alpha = alpha, a.beta = beta;
    a.A = (const uint8_t *)d_A, a.B = (const uint8_t *)d_B, a.C = (uint8_t *)d_C;
    a.lda = lda, a.ldb = ldb, a.ldc = ldc, a.sa = strideA, a.sb = strideB, a.sc = strideC;
    // 16-byte loads need every row of every batch on a 16-byte boundary
    const bool aligned = (((uintptr_t)d_A | (uintptr_t)d_B) & 15) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (strideA % 8) == 0 &&
                         (strideB % 8) == 0;
    return launch(a, Ctype, batch, aligned, (hipStream_t)stream);
}

// pointer-array form (ggml uses it when dims 2 / 3 broadcast: ggml-cuda.cu.patch:18330-18376); the arrays live on the device
int lfamd_gemm_batched_f16(long m, long n, long k, float alpha, const void *const *d_Aarray, long lda, const void *const *d_Barray,
                           long ldb, float beta, void *const *d_Carray, int Ctype, long ldc, int batch, void *stream) {
    bat_args a{};
    a.m = m, a.n = n, a.k = k, a.alpha = alpha, a.beta = beta;
    a.Aarr = d_Aarray, a.Barr = d_Barray, a.Carr = d_Carray;
    a.lda = lda, a.ldb = ldb, a.ldc = ldc;
    return launch(a, Ctype, batch, false, (hipStream_t)stream); // (alignment of the individual pointers is not known on the host)
}
}
