// development probe: do the matrix pipe and the vector ALU of a SIMD run side by side?  (MI355X)
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap_probe.hip -o /tmp/mvo.bin && /tmp/mvo.bin
// One work-group per CU, 4 / 8 waves (one / two per SIMD).  An iteration is 4 x v_mfma_i32_32x32x32_i8 (independent accumulators)
// and / or 16 x v_pk_mul_lo_u16 (independent chains); printed: s_memtime cycles per iteration as a wave sees them.
//   mfma        : the MFMAs alone                           valu        : the VALU instructions alone
//   both        : 16 VALU then 4 MFMA, unrelated registers   dependent   : the MFMAs' B operand is what the VALU just wrote
//   interleaved : 4 x (4 VALU, 1 MFMA), unrelated registers
//   split roles : 8 waves, waves 0..3 run `mfma`, waves 4..7 run `valu` (both times printed)
//   with loaders: 8 waves, waves 0..3 run `interleaved`, waves 4..7 issue LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB each, an
//                 L2-resident source) back to back in groups of PIECES with a vmcnt(0) wait between groups; printed: the computing
//                 waves' cycles per iteration and the pieces a loader issued meanwhile
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REPS 256

#define MFMA(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VALU4(r0, r1, r2, r3, s)                                                                                     \
    asm volatile("v_pk_mul_lo_u16 %0, %4, %0\n\tv_pk_mul_lo_u16 %1, %4, %1\n\tv_pk_mul_lo_u16 %2, %4, %2\n\tv_pk_mul_lo_u16 %3, %4, %3" \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)                                                            \
                 : "v"(s))

__device__ int g_stop[256];
__device__ unsigned long long g_pieces[256 * 4];
template <int PIECES>
__global__ __launch_bounds__(512) void probe_dma(unsigned long long *out, uint32_t *sink, const uint8_t *src) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[8 * 16384];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave >= 4) { // loaders: groups of PIECES pieces into this wave's 16 KiB of LDS until the computing waves are done
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds + (uint32_t)(wave * 16384);
        const uint32_t dst_u = __builtin_amdgcn_readfirstlane(dst);
        const uint8_t *base = src + (size_t)(blockIdx.x % 64) * 65536 + (size_t)(wave - 4) * 16384;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
        const uint8_t *b = (const uint8_t *)(((uintptr_t)hi << 32) | lo);
        asm volatile("s_nop 4" ::"s"(b));
        unsigned long long cnt = 0;
        volatile int *stop = &g_stop[blockIdx.x];
        while (true) {
#pragma unroll
            for (int p = 0; p < PIECES; p++)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((uint32_t)(lane * 16 + p * 1024)), "s"(b),
                             "s"(dst_u + (uint32_t)(p * 1024))
                             : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            cnt += PIECES;
            if (*stop >= 4)
                break;
        }
        if (lane == 0)
            g_pieces[blockIdx.x * 4 + wave - 4] = cnt;
        return;
    }
    v16i acc[4];
    for (int k = 0; k < 4; k++)
        for (int e = 0; e < 16; e++)
            acc[k][e] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3};
    v4i b[4];
    uint32_t r[16];
    for (int k = 0; k < 16; k++)
        r[k] = threadIdx.x * 3 + k;
    for (int k = 0; k < 4; k++)
        b[k] = v4i{(int)r[4 * k], (int)r[4 * k + 1], (int)r[4 * k + 2], (int)r[4 * k + 3]};
    uint32_t s = 0x00010001u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REPS; i++) {
        VALU4(r[0], r[1], r[2], r[3], s);
        MFMA(acc[0], a, b[0]);
        VALU4(r[4], r[5], r[6], r[7], s);
        MFMA(acc[1], a, b[1]);
        VALU4(r[8], r[9], r[10], r[11], s);
        MFMA(acc[2], a, b[2]);
        VALU4(r[12], r[13], r[14], r[15], s);
        MFMA(acc[3], a, b[3]);
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        out[blockIdx.x * 16 + wave] = t1 - t0;
        atomicAdd(&g_stop[blockIdx.x], 1);
    }
    uint32_t x = 0;
    for (int k = 0; k < 16; k++)
        x ^= r[k];
    for (int k = 0; k < 4; k++)
        for (int e = 0; e < 16; e++)
            x ^= (uint32_t)acc[k][e];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long *out, uint32_t *sink) {
    const int wave = threadIdx.x >> 6;
    v16i acc[4];
    for (int k = 0; k < 4; k++)
        for (int e = 0; e < 16; e++)
            acc[k][e] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3};
    v4i b[4];
    uint32_t r[16];
    for (int k = 0; k < 16; k++)
        r[k] = threadIdx.x * 3 + k;
    for (int k = 0; k < 4; k++)
        b[k] = v4i{(int)r[4 * k], (int)r[4 * k + 1], (int)r[4 * k + 2], (int)r[4 * k + 3]};
    uint32_t s = 0x00010001u;
    int mode = MODE;
    if (MODE == 5)
        mode = wave < 4 ? 0 : 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
        for (int i = 0; i < REPS; i++) {
            MFMA(acc[0], a, b[0]);
            MFMA(acc[1], a, b[1]);
            MFMA(acc[2], a, b[2]);
            MFMA(acc[3], a, b[3]);
        }
    } else if (mode == 1) {
        for (int i = 0; i < REPS; i++) {
            VALU4(r[0], r[1], r[2], r[3], s);
            VALU4(r[4], r[5], r[6], r[7], s);
            VALU4(r[8], r[9], r[10], r[11], s);
            VALU4(r[12], r[13], r[14], r[15], s);
        }
    } else if (mode == 2) {
        for (int i = 0; i < REPS; i++) {
            VALU4(r[0], r[1], r[2], r[3], s);
            VALU4(r[4], r[5], r[6], r[7], s);
            VALU4(r[8], r[9], r[10], r[11], s);
            VALU4(r[12], r[13], r[14], r[15], s);
            MFMA(acc[0], a, b[0]);
            MFMA(acc[1], a, b[1]);
            MFMA(acc[2], a, b[2]);
            MFMA(acc[3], a, b[3]);
        }
    } else if (mode == 3) {
        for (int i = 0; i < REPS; i++) {
            VALU4(b[0].x, b[0].y, b[0].z, b[0].w, s);
            VALU4(b[1].x, b[1].y, b[1].z, b[1].w, s);
            VALU4(b[2].x, b[2].y, b[2].z, b[2].w, s);
            VALU4(b[3].x, b[3].y, b[3].z, b[3].w, s);
            MFMA(acc[0], a, b[0]);
            MFMA(acc[1], a, b[1]);
            MFMA(acc[2], a, b[2]);
            MFMA(acc[3], a, b[3]);
        }
    } else if (mode == 4) {
        for (int i = 0; i < REPS; i++) {
            VALU4(r[0], r[1], r[2], r[3], s);
            MFMA(acc[0], a, b[0]);
            VALU4(r[4], r[5], r[6], r[7], s);
            MFMA(acc[1], a, b[1]);
            VALU4(r[8], r[9], r[10], r[11], s);
            MFMA(acc[2], a, b[2]);
            VALU4(r[12], r[13], r[14], r[15], s);
            MFMA(acc[3], a, b[3]);
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 16 + wave] = t1 - t0;
    uint32_t x = 0;
    for (int k = 0; k < 16; k++)
        x ^= r[k];
    for (int k = 0; k < 4; k++)
        for (int e = 0; e < 16; e++)
            x ^= (uint32_t)acc[k][e];
    for (int k = 0; k < 4; k++)
        x ^= (uint32_t)(b[k].x ^ b[k].y ^ b[k].z ^ b[k].w);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    unsigned long long *out, h[256 * 16];
    uint32_t *sink;
    hipMalloc(&out, sizeof(h));
    hipMalloc(&sink, 256 * 512 * 4);
    struct {
        const char *name;
        void (*fn)(unsigned long long *, uint32_t *);
        int split;
    } ks[] = {{"mfma (4 MFMA)", probe<0>, 0},        {"valu (16 VALU)", probe<1>, 0},           {"both (16 VALU, 4 MFMA)", probe<2>, 0},
              {"dependent (16 VALU -> 4 MFMA)", probe<3>, 0}, {"interleaved 4 x (4 VALU, 1 MFMA)", probe<4>, 0}, {"split roles", probe<5>, 1}};
    for (auto &k : ks)
        for (int waves : {4, 8}) {
            if (k.split && waves != 8)
                continue;
            hipMemset(out, 0, sizeof(h));
            for (int rep = 0; rep < 3; rep++)
                hipLaunchKernelGGL(k.fn, dim3(256), dim3(waves * 64), 0, 0, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double sa = 0, sb = 0;
            for (int bI = 0; bI < 256; bI++)
                for (int w = 0; w < waves; w++)
                    (w < 4 ? sa : sb) += (double)h[bI * 16 + w];
            if (k.split)
                printf("%-36s %d waves/WG: MFMA waves %7.1f, VALU waves %7.1f cycles per iteration\n", k.name, waves, sa / (256 * 4) / REPS,
                       sb / (256 * 4) / REPS);
            else
                printf("%-36s %d waves/WG: %7.1f cycles per iteration and wave\n", k.name, waves, (sa + sb) / (256 * waves) / REPS);
        }
    {
        uint8_t *src;
        hipMalloc(&src, 64 * 65536);
        hipMemset(src, 1, 64 * 65536);
        auto run = [&](const char *name, void (*fn)(unsigned long long *, uint32_t *, const uint8_t *), int pieces) {
            int zero[256] = {0};
            double per = 0, pcs = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipMemcpyToSymbol(HIP_SYMBOL(g_stop), zero, sizeof(zero));
                hipMemset(out, 0, sizeof(h));
                hipLaunchKernelGGL(fn, dim3(256), dim3(512), 0, 0, out, sink, src);
                hipDeviceSynchronize();
                hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
                unsigned long long pc[256 * 4];
                hipMemcpyFromSymbol(pc, HIP_SYMBOL(g_pieces), sizeof(pc));
                double sa = 0, sp = 0;
                for (int bI = 0; bI < 256; bI++)
                    for (int w = 0; w < 4; w++)
                        sa += (double)h[bI * 16 + w], sp += (double)pc[bI * 4 + w];
                per = sa / (256 * 4) / REPS, pcs = sp / (256 * 4) / REPS;
            }
            printf("%-36s computing waves %7.1f cycles per iteration; %.2f pieces per loader and iteration (groups of %d)\n", name, per, pcs, pieces);
        };
        run("with loaders (groups of 10)", probe_dma<10>, 10);
        run("with loaders (groups of 4)", probe_dma<4>, 4);
        run("with loaders (groups of 1)", probe_dma<1>, 1);
    }
    return 0;
}
