// development probe: cycles per v_mfma_f32_32x32x16_f16 for the K-step structure of gemm_lw.hip, feature by feature.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float16_t_ __attribute__((ext_vector_type(16)));
union frag_u {
    half8_t v;
    half2_t p[4];
};
__device__ static inline half2_t as_half2(uint32_t u) {
    union {
        uint32_t u;
        half2_t h;
    } c;
    c.u = u;
    return c.h;
}
__device__ static inline half2_t pk_fma(half2_t a, half2_t b, half2_t c) {
    return __builtin_elementwise_fma(a, b, c);
}

// FEAT bits: 1 = dequant VALU (12 per K-step), 2 = fragment reads from LDS (4 ds_read_b128 per K-step, two K-steps ahead),
//            4 = barrier every 8 K-steps with 4 parked waves, 8 = VALU without the sched_group_barrier interleave
template <int FEAT>
__global__ __launch_bounds__(512) void probe(float *out, unsigned long long *cyc, const uint32_t *qsrc, int iters) {
    extern __shared__ unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((FEAT & 4) && wave >= 4) {
        for (int it = 0; it < iters; it++)
            __builtin_amdgcn_s_barrier();
        return;
    }
    if (!(FEAT & 4) && wave >= 4)
        return;
    for (int e = threadIdx.x; e < 32768 / 4; e += 256)
        ((uint32_t *)lds)[e] = 0x3c003c00u + (e & 0xff);
    __syncthreads();
    float16_t_ acc[4];
    for (int a = 0; a < 4; a++)
        for (int r = 0; r < 16; r++)
            acc[a][r] = 0.f;
    uint32_t qw[8];
    for (int e = 0; e < 8; e++)
        qw[e] = qsrc[(threadIdx.x * 8 + e) & 1023];
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic));
    const half2_t S = {(_Float16)0.01f, (_Float16)0.01f}, O = {(_Float16)-10.24f, (_Float16)-10.24f};
    const half2_t S16 = {(_Float16)0.000625f, (_Float16)0.000625f}, O16 = {(_Float16)-0.64f, (_Float16)-0.64f};
    auto dequant = [&](uint32_t x) -> half8_t {
        frag_u f;
        const uint32_t y = x >> 8;
        f.p[0] = pk_fma(as_half2((x & 0x000F000Fu) | magic), S, O);
        f.p[1] = pk_fma(as_half2((x & 0x00F000F0u) | magic), S16, O16);
        f.p[2] = pk_fma(as_half2((y & 0x000F000Fu) | magic), S, O);
        f.p[3] = pk_fma(as_half2((y & 0x00F000F0u) | magic), S16, O16);
        return f.v;
    };
    const uint32_t lbase = (uint32_t)(uintptr_t)lds + (lane & 31) * 256 + (lane >> 5) * 16;
    half8_t F[4][4];
    auto read_frags = [&](half8_t(&f)[4], int t8) {
        const uint32_t a = lbase + ((t8 * 32) ^ ((lane & 15) * 16));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(a));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(f[1]) : "v"(a));
        asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(f[2]) : "v"(a));
        asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(f[3]) : "v"(a));
    };
    half8_t A;
    for (int e = 0; e < 8; e++)
        A[e] = (_Float16)(0.01f * (threadIdx.x % 13 + e));
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++)
            F[a][b] = A;
    half8_t wf = dequant(qw[0]);
    if (FEAT & 2) {
        read_frags(F[0], 0);
        read_frags(F[1], 1);
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t8 = 0; t8 < 8; t8++) {
            if (FEAT & 2) {
                read_frags(F[(t8 + 2) & 3], (t8 + 2) & 7);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(F[t8 & 3][0]), "+v"(F[t8 & 3][1]), "+v"(F[t8 & 3][2]), "+v"(F[t8 & 3][3]));
            }
            half8_t wn = wf;
            if (FEAT & 1) {
                wn = dequant(qw[(t8 + 1) & 7]);
                qw[t8] += 0x01010101u; // keeps the work loop-variant
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t8 & 3][a], wf, acc[a], 0, 0, 0);
            if ((FEAT & 1) && !(FEAT & 8)) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            }
            wf = wn;
        }
        if (FEAT & 4)
            __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0;
    for (int a = 0; a < 4; a++)
        for (int r = 0; r < 16; r++)
            s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        cyc[2 * blockIdx.x] = t1 - t0;
        cyc[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int FEAT>
static void run(const char *name) {
    const int blocks = 256, iters = 2000, lds = 150000;
    float *out;
    uint32_t *q;
    unsigned long long *cyc, h[2 * 256];
    (void)hipMalloc(&out, sizeof(float) * blocks * 512);
    (void)hipMalloc(&cyc, sizeof(h));
    (void)hipMalloc(&q, 4096);
    (void)hipMemset(q, 0x5a, 4096);
    (void)hipFuncSetAttribute((const void *)probe<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 3; rep++)
        probe<FEAT><<<blocks, 512, lds>>>(out, cyc, q, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int b = 0; b < blocks; b++)
        c += h[2 * b], r += h[2 * b + 1];
    printf("%-46s %.1f cycles/MFMA (%.0f per K-step), clock %.0f MHz\n", name, c / blocks / (iters * 32.0),
           c / blocks / (iters * 8.0), c / r * 100.0);
    (void)hipFree(out), (void)hipFree(cyc), (void)hipFree(q);
}

int main() {
    run<0>("bare MFMA");
    run<1>("+ dequant VALU, interleaved");
    run<9>("+ dequant VALU, compiler order");
    run<2>("+ LDS fragments");
    run<3>("+ dequant + LDS fragments");
    run<7>("+ dequant + LDS + barrier/8 K-steps");
    return 0;
}
