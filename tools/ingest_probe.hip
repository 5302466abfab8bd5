// development probe: how fast can ONE work-group per CU take in an L2-resident working set shaped like the prefill
// GEMM's (4096 x 4096 x 512 as 128-row x 64-token tiles: 295 KB of packed weights + 512 KB of f16 activations per tile),
// by path: LDS-DMA (global_load_lds_dwordx4) vs loads to registers, by who issues them, and how deep the ring is.
//   hipcc -O3 --offload-arch=gfx950 tools/ingest_probe.hip -o tools/ingest_probe.bin && tools/ingest_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t_ __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)

__device__ static inline uint32_t lds_addr(const void *p) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)(const uint8_t *)p);
}
__device__ static inline const uint8_t *uniform_ptr(const void *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ static inline void glds1x16(const void *base, uint32_t lds_dst, uint32_t o0) {
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(o0), "s"(base), "s"(lds_dst)
                 : "memory");
}
__device__ static inline void gload16(u32x4 &dst, const void *base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}

// tile of work-group b: XCD-aware, super-tiles of 8 row blocks x 4 token tiles per XCD (gemm_common.h tile_of, simplified
// for 32 x 8 tiles: XCD x = b & 7 takes row blocks 8 (x & 3) .. +8 and token tiles 4 (x >> 2) .. +4)
__device__ static inline void tile_of(int b, int &rb, int &ct) {
    const int x = b & 7, l = b >> 3; // l = 0..31 inside the XCD
    rb = 8 * (x & 3) + (l & 7);
    ct = 4 * (x >> 2) + (l >> 3);
}

// MODE 0: everything by LDS-DMA; LW loader waves (the first LW waves of the work-group) issue, the rest only meet the barrier
// MODE 1: activations by LDS-DMA (LW waves), weights by global_load_dwordx4 to registers in ALL waves (xor-folded)
// MODE 2: everything to registers (all waves)
// ABYTES = activation bytes per token and super-block (512: f16 codes, 256: int8 codes)
// stage = one super-block of K (256): ACT = 64 tokens x ABYTES, W = 128 rows x 144 B = 18 KB; STAGES - 1 stages in flight
template <int MODE, int LW, int STAGES, int ABYTES, int COMPUTE = 0>
__global__ __launch_bounds__(512) void ingest(const uint8_t *__restrict__ X, const uint8_t *__restrict__ W, int nb, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int ACT = 64 * ABYTES, WB = 128 * 144, SLOT = ACT + (MODE == 0 ? WB : 0);
    constexpr int APIECES = ACT / 1024, WPIECES = WB / 1024;
    constexpr int AW = MODE == 2 ? (APIECES + 7) / 8 : APIECES / LW;          // activation pieces per issuing wave and stage
    constexpr int WW = MODE == 0 ? (WPIECES + LW - 1) / LW : (WPIECES + 7) / 8; // weight pieces per issuing wave and stage
    constexpr int NPW = AW + WW;
    static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 2) * NPW < 64, "vmcnt is 6 bits; four register sets");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rb, ct;
    tile_of(blockIdx.x, rb, ct);
    const uint32_t lds0 = lds_addr(lds);
    const uint8_t *xb = X + (size_t)ct * 64 * ABYTES; // [nb][512 tokens][ABYTES]: a tile's tokens of one super-block are contiguous
    const size_t xstride = (size_t)512 * ABYTES;
    const uint8_t *wb = W + (size_t)rb * nb * WB;     // [rb][nb][18432]
    u32x4 fold = {0, 0, 0, 0};
    const uint32_t lo = (uint32_t)lane * 16;
    u32x4 r[4][AW + WW];
    const bool loader = MODE == 2 || wave < LW;
    float16_t_ macc0, macc1;
    for (int e = 0; e < 16; e++)
        macc0[e] = 0.f, macc1[e] = 0.f;

    auto issue = [&](int b, auto setc) {
        constexpr int set = decltype(setc)::value;
        const int bb = b < nb ? b : nb - 1;
        const uint32_t slot = lds0 + (uint32_t)(b % STAGES) * SLOT;
        const uint8_t *xs = uniform_ptr(xb + (size_t)bb * xstride), *ws = uniform_ptr(wb + (size_t)bb * WB);
        if (loader || MODE == 1) {
#pragma unroll
            for (int q = 0; q < AW; q++) {
                if constexpr (MODE == 2) {
                    int p = q * 8 + wave;
                    p = p < APIECES ? p : APIECES - 1;
                    gload16(r[set][q], xs, (uint32_t)(p * 1024) + lo);
                } else if (loader) {
                    const int p = wave * AW + q;
                    glds1x16(xs, slot + (uint32_t)(p * 1024), (uint32_t)(p * 1024) + lo);
                }
            }
#pragma unroll
            for (int q = 0; q < WW; q++) {
                if constexpr (MODE == 0) {
                    int p = q * LW + wave;
                    p = p < WPIECES ? p : WPIECES - 1;
                    glds1x16(ws, slot + ACT + (uint32_t)(p * 1024), (uint32_t)(p * 1024) + lo);
                } else {
                    int p = q * 8 + wave;
                    p = p < WPIECES ? p : WPIECES - 1;
                    gload16(r[set][AW + q], ws, (uint32_t)(p * 1024) + lo);
                }
            }
        }
    };
    auto step = [&](int b, auto setc, auto nextc) {
        constexpr int set = decltype(setc)::value;
        // MODE 1: non-loader waves only have their WW register loads per stage in flight
        if (MODE == 1 && !loader)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * WW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * NPW) : "memory");
        if constexpr (MODE != 2)
            __builtin_amdgcn_s_barrier();
        issue(b + STAGES - 1, nextc);
#pragma unroll
        for (int q = 0; q < AW + WW; q++)
            if (MODE == 2 || (MODE == 1 && q >= AW)) {
                asm volatile("" : "+v"(r[set][q]));
                fold ^= r[set][q];
            }
        if constexpr (COMPUTE > 0) { // the waves that do not load run COMPUTE x (2 ds_read_b128 + 2 MFMA) per stage on the landed bytes
            if (!loader || COMPUTE >= 100) {
                const uint32_t a0 = lds0 + (uint32_t)(b % STAGES) * SLOT + (uint32_t)((lane & 31) * 256 + (lane >> 5) * 16);
#pragma unroll 4
                for (int t = 0; t < COMPUTE % 100; t++) {
                    half8_t f0, f1;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(f0), "=&v"(f1)
                                 : "v"(a0 + (uint32_t)(((2 * (t & 7)) ^ (lane & 15)) * 16)));
                    macc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, f1, macc0, 0, 0, 0);
                    macc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1, f0, macc1, 0, 0, 0);
                }
            }
        }
        if constexpr (MODE != 2) { // a token read of the landed stage
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds0 + (uint32_t)(b % STAGES) * SLOT + threadIdx.x * 4));
            fold.x ^= v;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // prologue: stages 0 .. STAGES-2 into register sets 0 .. STAGES-2
    issue(0, I0{});
    if constexpr (STAGES >= 3)
        issue(1, I1{});
    if constexpr (STAGES >= 4)
        issue(2, I2{});
    // stage b uses set b & 3 and issues stage b + STAGES - 1 into set (b + STAGES - 1) & 3
    for (int b = 0; b < nb; b += 4) {
        if constexpr (STAGES == 2) {
            step(b, I0{}, I1{}), step(b + 1, I1{}, I2{}), step(b + 2, I2{}, I3{}), step(b + 3, I3{}, I0{});
        } else if constexpr (STAGES == 3) {
            step(b, I0{}, I2{}), step(b + 1, I1{}, I3{}), step(b + 2, I2{}, I0{}), step(b + 3, I3{}, I1{});
        } else {
            step(b, I0{}, I3{}), step(b + 1, I1{}, I0{}), step(b + 2, I2{}, I1{}), step(b + 3, I3{}, I2{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (COMPUTE > 0)
        fold.y ^= (uint32_t)(macc0[3] + macc1[5]);
    if ((fold.x ^ fold.y ^ fold.z ^ fold.w) == 0x12345678u)
        out[blockIdx.x] = fold.x;
}

// launch floor: the same grid with nothing to do
__global__ __launch_bounds__(512) void empty_kernel(uint32_t *out) {
    if (threadIdx.x == 1000000)
        out[0] = 1;
}

template <int MODE, int LW, int STAGES, int ABYTES, int COMPUTE = 0>
static void run(const char *name, const uint8_t *X, const uint8_t *W, uint32_t *out, int nb) {
    constexpr int ACT = 64 * ABYTES, WB = 128 * 144, SLOT = ACT + (MODE == 0 ? WB : 0);
    const size_t lds = MODE == 2 ? 1024 : (size_t)STAGES * SLOT;
    auto k = ingest<MODE, LW, STAGES, ABYTES, COMPUTE>;
    CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++)
        k<<<256, 512, lds>>>(X, W, nb, out);
    CHECK(hipDeviceSynchronize());
    const int iters = 50;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++)
        k<<<256, 512, lds>>>(X, W, nb, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    const double bytes = (double)nb * (ACT + WB);
    printf("%-44s lds %6zu  %7.2f us/launch  %6.1f GB/s per CU  %6.2f TB/s chip\n", name, lds, us, bytes / us / 1e3, bytes * 256 / us / 1e6);
}

int main() {
    const int nb = 16; // K = 4096
    uint8_t *X, *W;
    uint32_t *out;
    const size_t xbytes = (size_t)nb * 512 * 512, wbytes = (size_t)32 * nb * 128 * 144;
    CHECK(hipMalloc(&X, xbytes));
    CHECK(hipMalloc(&W, wbytes));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(X, 1, xbytes));
    CHECK(hipMemset(W, 2, wbytes));
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; i++)
            empty_kernel<<<256, 512, 131072>>>(out);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 50; i++)
            empty_kernel<<<256, 512, 131072>>>(out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty 256 x 512 threads x 128 KiB LDS: %.2f us/launch\n", ms * 1e3 / 50);
    }
    run<0, 8, 3, 512>("LDS-DMA all, 8 loader waves, 3 stages", X, W, out, nb);
    run<0, 8, 2, 512>("LDS-DMA all, 8 loader waves, 2 stages", X, W, out, nb);
    run<0, 4, 3, 512>("LDS-DMA all, 4 loader waves, 3 stages", X, W, out, nb);
    run<0, 2, 3, 512>("LDS-DMA all, 2 loader waves, 3 stages", X, W, out, nb);
    run<1, 8, 3, 512>("LDS-DMA acts (8 waves) + weights to VGPR", X, W, out, nb);
    run<1, 4, 3, 512>("LDS-DMA acts (4 waves) + weights to VGPR", X, W, out, nb);
    run<1, 8, 4, 512>("LDS-DMA acts (8 waves, 4 stages) + W to VGPR", X, W, out, nb);
    run<2, 8, 3, 512>("all to VGPR (8 waves), 2 stages in flight", X, W, out, nb);
    run<2, 8, 4, 512>("all to VGPR (8 waves), 3 stages in flight", X, W, out, nb);
    run<0, 4, 3, 512, 16>("DMA all by 4 waves + 4 waves x 32 MFMA/stage", X, W, out, nb);
    run<0, 4, 3, 512, 8>("DMA all by 4 waves + 4 waves x 16 MFMA/stage", X, W, out, nb);
    run<0, 8, 3, 512, 108>("DMA all by 8 waves, each also 16 MFMA/stage", X, W, out, nb);
    run<0, 8, 3, 512, 116>("DMA all by 8 waves, each also 32 MFMA/stage", X, W, out, nb);
    run<1, 8, 3, 512, 108>("DMA acts + W to VGPR, 8 waves x 16 MFMA/st", X, W, out, nb);
    run<0, 8, 3, 256>("int8 acts: LDS-DMA all, 8 waves, 3 stages", X, W, out, nb);
    run<1, 8, 4, 256>("int8 acts: LDS-DMA acts + W to VGPR, 4 st", X, W, out, nb);
    run<2, 8, 4, 256>("int8 acts: all to VGPR, 3 in flight", X, W, out, nb);
    return 0;
}
