// development probe: cycles per v_mfma_f32_32x32x16_f16 on one SIMD as a function of (waves per SIMD, accumulators per wave,
// filler instructions per MFMA), everything in registers — what the matrix pipe gives before LDS / memory enter.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate_probe.hip -o tools/mfma_rate_probe.bin && tools/mfma_rate_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float16_t_ __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)

// NACC accumulators per wave, VALU filler instructions (v_pk_fma_f16 on private registers) per MFMA, LDSR ds_read_b128 per MFMA
template <int NACC, int VALU, int LDSR>
__global__ __launch_bounds__(512) void probe(float *out, unsigned long long *cyc, int iters, int waves) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 16384 / 4; e += blockDim.x)
        ((uint32_t *)lds)[e] = 0x3c003c00u + (e & 0xff);
    __syncthreads();
    if (wave >= waves)
        return;
    float16_t_ acc[NACC];
    for (int a = 0; a < NACC; a++)
        for (int r = 0; r < 16; r++)
            acc[a][r] = 0.f;
    half8_t A, B;
    for (int e = 0; e < 8; e++)
        A[e] = (_Float16)(0.01f * (lane % 13 + e)), B[e] = (_Float16)(0.02f * (lane % 7 + e));
    half2_t f[4] = {{(_Float16)1.0f, (_Float16)2.0f}, {(_Float16)0.5f, (_Float16)0.25f}, {(_Float16)3.0f, (_Float16)1.5f}, {(_Float16)0.1f, (_Float16)0.2f}};
    const half2_t s = {(_Float16)1.0001f, (_Float16)0.9999f}, o = {(_Float16)0.001f, (_Float16)-0.001f};
    const uint32_t la = (uint32_t)(uintptr_t)lds + lane * 16;
    half8_t L[2];
    L[0] = A, L[1] = B;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[a], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU; v++)
                f[v & 3] = __builtin_elementwise_fma(f[v & 3], s, o);
#pragma unroll
            for (int r = 0; r < LDSR; r++)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(L[r & 1]) : "v"(la), "n"(1024 * 0));
            if (LDSR)
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LDSR) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int a = 0; a < NACC; a++)
        sum += acc[a][lane & 15];
    sum += (float)f[0][0] + (float)f[1][1] + (float)f[2][0] + (float)f[3][1] + (float)L[0][0] + (float)L[1][1];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (lane == 0)
        cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NACC, int VALU, int LDSR>
static void run(int waves, float *out, unsigned long long *cyc) {
    const int iters = 2000 / NACC;
    CHECK(hipMemset(cyc, 0, 256 * 8 * 8));
    probe<NACC, VALU, LDSR><<<256, 512, 16384>>>(out, cyc, iters, waves);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[256 * 8];
    CHECK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    // waves w and w + 4 share a SIMD (8 waves: two per SIMD); per-SIMD MFMA count = (waves per SIMD) * iters * NACC
    double worst = 0, sum = 0;
    int cnt = 0;
    for (int b = 0; b < 256; b++)
        for (int w = 0; w < waves; w++) {
            const double c = (double)h[b * 8 + w];
            if (c > worst)
                worst = c;
            sum += c, cnt++;
        }
    const double per_simd = (waves > 4 ? 2.0 : 1.0) * iters * NACC;
    printf("waves/CU %d  acc/wave %d  VALU/MFMA %d  ds_read/MFMA %d :  %6.1f cycles per MFMA per SIMD (mean wave), %6.1f (slowest wave)\n", waves, NACC, VALU,
           LDSR, sum / cnt / per_simd, worst / per_simd);
}

int main() {
    float *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    for (int waves : {4, 8}) {
        run<1, 0, 0>(waves, out, cyc);
        run<2, 0, 0>(waves, out, cyc);
        run<4, 0, 0>(waves, out, cyc);
        run<2, 2, 0>(waves, out, cyc);
        run<2, 5, 0>(waves, out, cyc);
        run<4, 2, 0>(waves, out, cyc);
        run<4, 3, 0>(waves, out, cyc);
        run<4, 5, 0>(waves, out, cyc);
        run<2, 5, 1>(waves, out, cyc);
        run<4, 2, 1>(waves, out, cyc);
        run<4, 3, 1>(waves, out, cyc);
    }
    return 0;
}
