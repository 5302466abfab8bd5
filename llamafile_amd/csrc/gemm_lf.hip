// gemm_lf.hip — Q8_0 batches (n > 8) on the f16 matrix cores, straight from the RESIDENT P80 image (the one the bit-exact
// vecdot GEMV reads: no second copy of the weights), scaled operands: f16(d * q) x f16(d8 * q8), f32 accumulate.  That is what
// the reference's GPU path computes for a Q8_0 batch (dequantise to f16, f16 GEMM: llamafile/tinyblas.cu:142-226, 475-560;
// ggml-cuda.cu.patch ggml_cuda_op_mul_mat_cublas) and the north star's tolerance for f16 MFMA paths, <= 1e-3 against the oracle
// (tinyblas_cpu.h:934-971 restated); LFAMD_FLAG_PRECISE / LFAMD_FLAG_Q80_EXACT ask for the bit-exact kernel (gemm_q80.hip).
//
// Shape of the work (the skeleton of gemm_i8.hip): work-group = 128 weight rows x 32 NT tokens, 8 waves.  Waves 4..7 only LOAD —
// everything arrives by LDS-DMA, R - 1 stages ahead in a ring of R, one barrier per stage; waves 0..3 compute, one per SIMD,
// 32 rows x 32 NT tokens each over the whole K.  A stage is ONE quad of Q8_0 blocks (128 weights per row):
//   weights     : 4 row tiles x 4 P80 tiles (8 rows x 4 blocks: 1024 B of codes + 64 B of f16 scales) — a P80 tile's codes are one
//                 DMA piece; lane (r, j) of a tile holds bytes 4 j .. 4 j + 3 of the quad's FOUR blocks, so a 16-byte LDS read
//                 gives lane (row, K half h) the operands of two K-steps: K-step s = 2 jj + e of the stage multiplies
//                 {block 2 e, block 2 e + 1} x {bytes 4 j .. 4 j + 3}, j = 4 h + jj — a permutation of the stage's 128 weights
//                 that the activation image is written in as well (prep_lf_kernel).  The piece's chunks are placed in LDS at
//                 position r * 8 + (j ^ (4 (T & 1) + (r >> 1))) of tile T (chosen on the SOURCE address): every 16-lane group
//                 of a ds_read_b128 then covers all 64 banks.
//   dequantise  : q ^ 0x80 = q + 128 as a byte; v_perm builds the f16 pair 1024 + u, a packed add of -1152 gives q exactly, a
//                 packed multiply d * q rounded once: 14 VALU per K-step and wave, feeding NT MFMAs — issued BETWEEN the MFMAs of
//                 the K-step before (a wave issues in order: NT MFMAs back to back hold its instruction stream for 32 (NT - 1)
//                 cycles, and what followed them started only then).
//   what bounds : the computing wave's own instruction stream (ablations on a stamped build, profiles/r04_lf_stamps.txt): with the
//                 MFMAs taken OUT a stage still takes 884 / 1255 cycles at 128 x 64 / 128 x 128 (512 / 1024 are the MFMAs'), without
//                 the dequantisation 676 / 1284, without the code-fragment reads 939 / 1350, without any DMA in the steady state
//                 the same as with it; LDS bandwidth is not it (147 KB of reads per 128 x 128 stage = 574 cycles at 256 B per clock).
//                 The large grids are paced by the loaders (L2 -> LDS, ~10 TB/s over the chip).
//   activations : f16(d8 * q8) (quantize_row_q8_0 arithmetic, or the caller's Q8_0 blocks), [quad][token][256 B] in the K order
//                 above, 16-byte chunks XOR-swizzled by token on the source address.
#include "gemm_wide_impl.h"

#include <stdlib.h>

typedef _Float16 lf_half4 __attribute__((ext_vector_type(4)));

#define LF_W 0        // 16 P80 tiles x 1024 B of codes
#define LF_WD 16384   // 16 x 64 B of block scales
#define LF_X 17408    // 32 NT tokens x 256 B

// LDS-DMA pieces: 64 lanes x 16 B (x 4 B) from base + voff to LDS at lds_dst + 16 (4) * lane.  SGPR base kept by scalar adds; M0
// written and not restored (nothing else in this kernel reads it).
__device__ static inline void lf_dma16(const void *base, uint32_t lds_dst, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__device__ static inline void lf_dma4(const void *base, uint32_t lds_dst, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int IMM>
__device__ static inline void lf_dsr16(u32x4 &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}
__device__ static inline void lf_dsr8(uint2 &dst, uint32_t addr) {
    asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr));
}

// The dequantisation of one fragment quarter as asm blocks that sit between two MFMAs (K-step pipeline of gemm_lf_q80_kernel):
//   u = w ^ 0x80808080 (q + 128 as bytes); t = perm -> the f16 pair (1024 + u); t += -1152 (q exactly); f = t * (d, d) (ONE rounding).
// The multiply of quarter j - 1 leads the block of quarter j: a packed multiply right behind the packed add it reads needs a wait
// state.  `after` is the accumulator of the MFMA in front of the block: not read, it only orders the block behind that MFMA.
__device__ static inline void lf_q_first(uint32_t &u, half2_t &t, uint32_t w, uint32_t k64, uint32_t sel, uint32_t m1152, const float16_t_ &after) {
    asm volatile("v_xor_b32 %0, 0x80808080, %2\n\tv_perm_b32 %1, %3, %0, %4\n\tv_pk_add_f16 %1, %5, %1"
                 : "=&v"(u), "=&v"(t)
                 : "v"(w), "s"(k64), "v"(sel), "v"(m1152), "v"(after));
}
__device__ static inline void lf_q_mul_second(half2_t &f, half2_t &t, half2_t tp, half2_t d, uint32_t u, uint32_t k64, uint32_t sel, uint32_t m1152,
                                              const float16_t_ &after) {
    asm volatile("v_pk_mul_f16 %0, %2, %3\n\tv_perm_b32 %1, %5, %4, %6\n\tv_pk_add_f16 %1, %7, %1"
                 : "=&v"(f), "=&v"(t)
                 : "v"(tp), "v"(d), "v"(u), "s"(k64), "v"(sel), "v"(m1152), "v"(after));
}
__device__ static inline void lf_q_mul_first(half2_t &f, uint32_t &u, half2_t &t, half2_t tp, half2_t d, uint32_t w, uint32_t k64, uint32_t sel,
                                             uint32_t m1152, const float16_t_ &after) {
    asm volatile("v_pk_mul_f16 %0, %3, %4\n\tv_xor_b32 %1, 0x80808080, %5\n\tv_perm_b32 %2, %6, %1, %7\n\tv_pk_add_f16 %2, %8, %2"
                 : "=&v"(f), "=&v"(u), "=&v"(t)
                 : "v"(tp), "v"(d), "v"(w), "s"(k64), "v"(sel), "v"(m1152), "v"(after));
}
// 64-token tile: two quarters per block
__device__ static inline void lf_h_first(half2_t &t0, half2_t &t1, uint32_t w, uint32_t k64, uint32_t sel0, uint32_t sel1, uint32_t m1152,
                                         const float16_t_ &after) {
    uint32_t u;
    asm volatile("v_xor_b32 %0, 0x80808080, %3\n\tv_perm_b32 %1, %4, %0, %5\n\tv_perm_b32 %2, %4, %0, %6\n\tv_pk_add_f16 %1, %7, %1\n\tv_pk_add_f16 %2, %7, %2"
                 : "=&v"(u), "=&v"(t0), "=&v"(t1)
                 : "v"(w), "s"(k64), "v"(sel0), "v"(sel1), "v"(m1152), "v"(after));
}
__device__ static inline void lf_h_mul_second(half2_t &f0, half2_t &f1, half2_t &t2, half2_t &t3, half2_t tp0, half2_t tp1, half2_t d, uint32_t w,
                                              uint32_t k64, uint32_t sel0, uint32_t sel1, uint32_t m1152, const float16_t_ &after) {
    uint32_t u;
    asm volatile("v_pk_mul_f16 %0, %5, %7\n\tv_xor_b32 %4, 0x80808080, %8\n\tv_pk_mul_f16 %1, %6, %7\n\tv_perm_b32 %2, %9, %4, %10\n\t"
                 "v_perm_b32 %3, %9, %4, %11\n\tv_pk_add_f16 %2, %12, %2\n\tv_pk_add_f16 %3, %12, %3"
                 : "=&v"(f0), "=&v"(f1), "=&v"(t2), "=&v"(t3), "=&v"(u)
                 : "v"(tp0), "v"(tp1), "v"(d), "v"(w), "s"(k64), "v"(sel0), "v"(sel1), "v"(m1152), "v"(after));
}

#ifdef LF_STAMPS // development (tools/lf_stamps.py): s_memtime sums of one work-group's compute wave 0 and loader wave 4
__device__ unsigned long long g_lf_stamps[16];
extern "C" int lfamd_debug_lf_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lf_stamps), sizeof(g_lf_stamps));
}
#define LF_T() __builtin_amdgcn_s_memtime()
#endif

template <int NT>
__global__ __launch_bounds__(512) void gemm_lf_q80_kernel(const gemm_mats mats, int nq, const _Float16 *__restrict__ Xh, long n, long n_pad,
                                                           int n_rb, int n_ct) {
#ifdef LF_CHECK_NQ // tools/isa_hazards.py: a fixed trip count
    nq = LF_CHECK_NQ;
#endif
    constexpr int SLOT = LF_X + NT * 8192;
    constexpr int RING = NT == 2 ? 4 : 3; // (a ring of three at 128 x 64 measured 3-11 % slower: profiles/r04_q80_batch_body.txt)
    constexpr int COLS = 32 * NT;
    constexpr int PIECES = 5 + 2 * NT; // of a loader and stage
    __shared__ __attribute__((aligned(16))) uint8_t lds[RING * SLOT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
#ifdef LF_STAMPS
    const unsigned long long st_start = LF_T();
    const bool st_on = blockIdx.x == 40;
    unsigned long long st_a = 0, st_b = 0, st_c = 0;
#endif

    // ---- tile of this work-group (the order of gemm_ks: XCD-aware super-tiles)
    int rb, ct;
    {
        const int n_wg = n_rb * n_ct;
        const int id = (int)blockIdx.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = id & 7;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
        tile_of(L, n_rb, n_ct, rb, ct);
    }
    int mj = 0;
#pragma unroll
    for (int jj = 1; jj < GEMM_MAX_MATS; jj++)
        if (jj < mats.count && rb >= mats.rb_end[jj - 1])
            mj = jj;
    if (mj > 0)
        rb -= mats.rb_end[mj - 1];
    const uint8_t *__restrict__ A = mats.A[mj];
    float *__restrict__ C = mats.C[mj];
    const long m = mats.m[mj], ldc = mats.ldc[mj];
    const long n0 = (long)ct * COLS;
    const int n_tiles8 = (int)((m + 7) / 8);
    const uint32_t lds0 = lds_addr(lds);

    if (wave >= 4) {
        // ================= loader waves: PIECES pieces per stage each, RING - 1 stages ahead, counted waits =================
        // (eight loader waves with half the pieces each measured 1-5 % slower: the loaders do not set the pace)
        const int lw = wave - 4;
        // 0..3: the codes of P80 tiles 4 (4 rb + lw) + T (tiles past the matrix: its last one); lane = position p of the tile in LDS
        uint32_t voffW[4];
#pragma unroll
        for (int T = 0; T < 4; T++) {
            const int t8 = (rb * 4 + lw) * 4 + T;
            const int r = lane >> 3, j = (lane & 7) ^ ((T & 1) * 4 + (r >> 1));
            voffW[T] = (uint32_t)(t8 < n_tiles8 ? t8 : n_tiles8 - 1) * (uint32_t)nq * (uint32_t)P80_TILE + (uint32_t)((r * 8 + j) * 16);
        }
        const uint32_t dstW = (uint32_t)(LF_W + lw * 4096);
        // 4: the block scales of those four tiles: 4 x 64 B, one dword per lane
        uint32_t voffD;
        {
            const int t8 = (rb * 4 + lw) * 4 + (lane >> 4);
            voffD = (uint32_t)(t8 < n_tiles8 ? t8 : n_tiles8 - 1) * (uint32_t)nq * (uint32_t)P80_TILE + (uint32_t)(P80_D + (lane & 15) * 4);
        }
        const uint32_t dstD = (uint32_t)(LF_WD + lw * 256);
        // 5..: activation pieces 2 NT lw + e: tokens 4 p .. 4 p + 3, lane = (token 4 p + (lane >> 4), slot lane & 15) <- chunk slot ^ (token & 15)
        uint32_t voffX[2 * NT];
#pragma unroll
        for (int e = 0; e < 2 * NT; e++) {
            const int tk = 4 * (2 * NT * lw + e) + (lane >> 4);
            voffX[e] = (uint32_t)(tk * 256 + (((lane & 15) ^ (tk & 15)) << 4));
        }
        const uint32_t dstX = (uint32_t)(LF_X + 2 * NT * lw * 1024);
        const size_t xstride = (size_t)n_pad * 256;
        const uint8_t *wt_n = uniform_ptr(A);
        const uint8_t *xh_n = uniform_ptr((const uint8_t *)Xh + (size_t)n0 * 256);
        // five wait states between the VALU writes of those SGPRs (v_readfirstlane) and the first vector-memory instruction that
        // reads them: hipcc pads such hazards itself, but not in front of an asm statement (tools/isa_hazards.py checks the ISA)
        asm volatile("s_nop 4" ::"s"(wt_n), "s"(xh_n));
        auto dma_stage = [&](uint32_t slot_base) {
#pragma unroll
            for (int T = 0; T < 4; T++)
                lf_dma16(wt_n, slot_base + dstW + (uint32_t)(T * 1024), voffW[T]);
            lf_dma4(wt_n, slot_base + dstD, voffD);
#pragma unroll
            for (int e = 0; e < 2 * NT; e++)
                lf_dma16(xh_n, slot_base + dstX + (uint32_t)(e * 1024), voffX[e]);
            wt_n += P80_TILE, xh_n += xstride; // (past the last quad nothing more is issued)
        };
#pragma unroll
        for (int st = 0; st < RING - 1; st++)
            if (st < nq) // (uniform)
                dma_stage(lds0 + (uint32_t)(st * SLOT));
        // stage 0 has landed when only the stages issued behind it are in flight
        {
            const int behind = (nq < RING - 1 ? nq : RING - 1) - 1;
            if (behind >= 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
            else if (behind == 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory"); // stage 0 has landed for everybody
        int slot_i = RING - 1; // slot of stage b + RING - 1
        if constexpr (RING == 3) {
            // A ring of three leaves the loader no slack under ONE barrier per stage: it would issue stage b + 1 and wait for it at
            // once (stamps: 805 cycles of wait + 1079 of issue per stage against 1530 of K-steps).  Two barriers per stage instead:
            // B1 — everybody is done with stage b - 1, its slot takes stage b + 2; B2 — in front of the computing waves' first read
            // of stage b + 1 (K-step 6) — stage b + 1, issued a whole stage ago, has landed.
            for (int b = 0; b < nq; b++) {
#ifdef LF_STAMPS
                const unsigned long long t0 = LF_T();
#endif
                asm volatile("s_barrier" ::: "memory"); // B1
#ifdef LF_STAMPS
                const unsigned long long t1 = LF_T();
#endif
                if (b + 2 < nq) {
                    dma_stage(lds0 + (uint32_t)(slot_i * SLOT));
#ifdef LF_STAMPS
                    st_c += LF_T() - t1;
#endif
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                slot_i = slot_i + 1 == RING ? 0 : slot_i + 1;
#ifdef LF_STAMPS
                const unsigned long long t2 = LF_T();
#endif
                asm volatile("s_barrier" ::: "memory"); // B2
#ifdef LF_STAMPS
                st_a += t2 - t1, st_b += (t1 - t0) + (LF_T() - t2); // (st_a: issue + vmcnt wait; st_c: the issue alone)
#endif
            }
        } else {
        for (int b = 0; b < nq; b++) {
            // this wave's pieces of stage b + 1 (all but those of the stages issued behind it); past the barrier everybody's have
            // landed and everybody is done with stage b - 1, whose slot takes stage b + RING - 1
            int behind = nq - 2 - b; // stages issued behind stage b + 1
            behind = behind < 0 ? 0 : behind > RING - 3 ? RING - 3 : behind;
#ifdef LF_STAMPS
            const unsigned long long t0 = LF_T();
#endif
            if (behind >= 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LF_STAMPS
            const unsigned long long t1 = LF_T();
#endif
            asm volatile("s_barrier" ::: "memory");
#ifdef LF_STAMPS
            const unsigned long long t2 = LF_T();
#endif
            if (b + RING - 1 < nq)
                dma_stage(lds0 + (uint32_t)(slot_i * SLOT));
            slot_i = slot_i + 1 == RING ? 0 : slot_i + 1;
#ifdef LF_STAMPS
            st_a += t1 - t0, st_b += t2 - t1, st_c += LF_T() - t2;
#endif
        }
        }
#ifdef LF_STAMPS
        if (st_on && wave == 4 && lane == 0)
            g_lf_stamps[8] = st_a, g_lf_stamps[9] = st_b, g_lf_stamps[10] = st_c, g_lf_stamps[11] = LF_T() - st_start;
#endif
        return;
    }

    // ================= compute waves: row tile rw, NT token tiles =================
    const int rw = wave;
    const int rt = rb * 4 + rw;
    const int T = i >> 3, r = i & 7;
    // ---- this lane's LDS read addresses in slot 0: the weight chunk of K-step pair jj, its scales, the code chunk of K-step s
    uint32_t adW[4];
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
        adW[jj] = lds0 + (uint32_t)(LF_W + (rw * 4 + T) * 1024 + (r * 8 + ((4 * h + jj) ^ ((T & 1) * 4 + (r >> 1)))) * 16);
    const uint32_t adD = lds0 + (uint32_t)(LF_WD + (rw * 4 + T) * 64 + r * 8);
    uint32_t adX[8];
#pragma unroll
    for (int s = 0; s < 8; s++)
        adX[s] = lds0 + (uint32_t)(LF_X + i * 256 + (((2 * s + h) ^ (i & 15)) << 4));

    float16_t_ acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 16; e++)
            acc[t][e] = 0.0f;

    asm volatile("s_barrier" ::: "memory"); // stage 0 has landed for everybody

    // Registers of the LDS pipeline (all reads are asm, every wait is counted: LDS reads of a wave return in issue order): XF[2][NT]
    // the code fragments ONE K-step ahead (two ahead measured the same), WQ[2] the weight chunks (one serves two K-steps) a pair
    // ahead of their dequantisation, SC the stage's scales.  In K-step s the wave issues X(s + 1) [NT reads], at even s the weight
    // chunk of K-steps s + 2, s + 3, at s = 6 the next stage's scales — from the NEXT slot once past this stage's end: that stage has
    // landed, the barrier at the top of this one said so — then waits for what the PREVIOUS step issued: younger are exactly this
    // step's own reads.
    u32x4 XF[2][NT], WQ[2];
    uint2 SC;
#define LF_WAIT(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N))
#define LF_XREAD(S, BASE)                                                                                                     \
    do {                                                                                                                      \
        lf_dsr16<0>(XF[(S)&1][0], adX[(S)&7] + (BASE));                                                                        \
        lf_dsr16<8192>(XF[(S)&1][1], adX[(S)&7] + (BASE));                                                                     \
        if (NT == 4) {                                                                                                        \
            lf_dsr16<16384>(XF[(S)&1][NT - 2], adX[(S)&7] + (BASE));                                                           \
            lf_dsr16<24576>(XF[(S)&1][NT - 1], adX[(S)&7] + (BASE));                                                           \
        }                                                                                                                     \
    } while (0)
#define LF_XTIE(S)                                                                                                            \
    do {                                                                                                                      \
        asm volatile("" : "+v"(XF[(S)&1][0]), "+v"(XF[(S)&1][1]));                                                              \
        if (NT == 4)                                                                                                          \
            asm volatile("" : "+v"(XF[(S)&1][NT - 2]), "+v"(XF[(S)&1][NT - 1]));                                                \
    } while (0)
    // The K-steps are a software pipeline: the fragment of K-step s + 1 is dequantised BETWEEN the MFMAs of K-step s (stamps of the
    // form that dequantised and multiplied in the same step: 1311 / 1986 cycles per stage at NT = 2 / 4 for 512 / 1024 of MFMA and no
    // barrier wait at all; now 1060 / 1530-1580, tools/lf_stamps.py).
    frag_u FW[2];  // K-step s multiplies FW[s & 1]
    half2_t Dk[4]; // per block of the quad: (d, d)
    uint32_t m1152u; // (-1152, -1152) in a VGPR: as an SGPR operand of the packed add, hipcc pads a wait state in front of the multiply
    asm volatile("v_mov_b32 %0, 0xe480e480" : "=v"(m1152u));
    const half2_t m1152 = as_half2(m1152u);
#define LF_DK()                                                                                                               \
    do {                                                                                                                      \
        asm volatile("" : "+v"(SC));                                                                                          \
        Dk[0] = as_half2(__builtin_amdgcn_perm(SC.x, SC.x, 0x01000100u)), Dk[1] = as_half2(__builtin_amdgcn_perm(SC.x, SC.x, 0x03020302u)); \
        Dk[2] = as_half2(__builtin_amdgcn_perm(SC.y, SC.y, 0x01000100u)), Dk[3] = as_half2(__builtin_amdgcn_perm(SC.y, SC.y, 0x03020302u)); \
    } while (0)
    // fragment quarter J of K-step S1: dword J >> 1 of the step's chunk half, byte pair J & 1; scale of block 2 (S1 & 1) + (J >> 1)
#define LF_QW(S1, J) (((S1)&1) ? ((J) < 2 ? WQ[((S1) >> 1) & 1].z : WQ[((S1) >> 1) & 1].w) : ((J) < 2 ? WQ[((S1) >> 1) & 1].x : WQ[((S1) >> 1) & 1].y))
#define LF_QD(S1, J) Dk[2 * ((S1)&1) + ((J) >> 1)]
    half2_t TQ[4]; // q as f16 pairs, one slot ahead of their multiply
    uint32_t UQ;   // a dword's q + 128 bytes between its two quarters
    uint32_t sel_lo, sel_hi, k64;
    asm volatile("v_mov_b32 %0, 0x04010400\n\tv_mov_b32 %1, 0x04030402\n\ts_mov_b32 %2, 0x64646464" : "=v"(sel_lo), "=v"(sel_hi), "=s"(k64));
    LF_XREAD(0, 0u);
    lf_dsr16<0>(WQ[0], adW[0]);
    lf_dsr8(SC, adD);
    LF_WAIT(0);
    asm volatile("" : "+v"(WQ[0]));
    LF_DK();
#pragma unroll
    for (int J = 0; J < 4; J++) { // the whole fragment of K-step 0 but its last multiplies (they open K-step 0)
        const uint32_t w = J < 2 ? WQ[0].x : WQ[0].y;
        TQ[J] = as_half2(__builtin_amdgcn_perm(0x64646464u, w ^ 0x80808080u, (J & 1) ? 0x04030402u : 0x04010400u)) + m1152;
    }
    FW[0].p[0] = TQ[0] * Dk[0], FW[0].p[1] = TQ[1] * Dk[0];
    if (NT == 4)
        FW[0].p[2] = TQ[2] * Dk[1];

    int slot_c = 0;
    for (int b = 0; b < nq; b++) {
        const uint32_t so = (uint32_t)(slot_c * SLOT);
        slot_c = slot_c + 1 == RING ? 0 : slot_c + 1;
        const uint32_t so_n = (uint32_t)(slot_c * SLOT);
#ifdef LF_STAMPS
        const unsigned long long t0 = LF_T();
#endif
        asm volatile("s_barrier" ::: "memory"); // ring of 4: stage b + 1 has landed for everybody; ring of 3: B1, everybody is done with stage b - 1
#ifdef LF_STAMPS
        const unsigned long long t1 = LF_T();
#endif
// K-step S: issue X(S + 1), at even S the weight chunk of K-steps S + 2, S + 3, at S = 6 the next stage's scales (from the NEXT slot
// once past this stage's end: it has landed, the barrier above said so); the last multiplies of THIS step's fragment; wait for
// everything the previous step issued; then the MFMAs of S, each followed by an asm block with a part of fragment S + 1 (the blocks
// are volatile and read the accumulator of the MFMA in front of them, so the order MFMA - block - MFMA holds; sched_barrier per step)
#define LF_STEP(S)                                                                                                            \
    do {                                                                                                                      \
        if (RING == 3 && (S) == 6)                                                                                            \
            asm volatile("s_barrier" ::: "memory"); /* B2: stage b + 1 has landed for everybody */                            \
        LF_XREAD((S) + 1, (S) == 7 ? so_n : so);                                                                              \
        if (((S)&1) == 0)                                                                                                     \
            lf_dsr16<0>(WQ[(((S) + 2) >> 1) & 1], adW[(((S) + 2) >> 1) & 3] + ((S) == 6 ? so_n : so));                           \
        if ((S) == 6)                                                                                                         \
            lf_dsr8(SC, adD + so_n);                                                                                          \
        if (NT == 2)                                                                                                          \
            FW[(S)&1].p[2] = TQ[2] * LF_QD(S, 2);                                                                               \
        FW[(S)&1].p[3] = TQ[3] * LF_QD(S, 3);                                                                                   \
        LF_WAIT(NT + (((S)&1) == 0) + ((S) == 6));                                                                            \
        LF_XTIE(S);                                                                                                           \
        asm volatile("" : "+v"(WQ[(((S) + 1) >> 1) & 1]));                                                                     \
        if ((S) == 7)                                                                                                         \
            LF_DK();                                                                                                          \
        _Pragma("unroll") for (int t = 0; t < NT; t++) {                                                                       \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, XF[(S)&1][t]), FW[(S)&1].v, acc[t], 0, 0, 0); \
            if (NT == 4) {                                                                                                    \
                if (t == 0)                                                                                                   \
                    lf_q_first(UQ, TQ[0], LF_QW((S) + 1, 0), k64, sel_lo, m1152u, acc[t]);                                       \
                else if (t == 1)                                                                                              \
                    lf_q_mul_second(FW[((S) + 1) & 1].p[0], TQ[1], TQ[0], LF_QD((S) + 1, 0), UQ, k64, sel_hi, m1152u, acc[t]);      \
                else if (t == 2)                                                                                              \
                    lf_q_mul_first(FW[((S) + 1) & 1].p[1], UQ, TQ[2], TQ[1], LF_QD((S) + 1, 1), LF_QW((S) + 1, 2), k64, sel_lo, m1152u, acc[t]); \
                else                                                                                                          \
                    lf_q_mul_second(FW[((S) + 1) & 1].p[2], TQ[3], TQ[2], LF_QD((S) + 1, 2), UQ, k64, sel_hi, m1152u, acc[t]);      \
            } else if (t == 0) {                                                                                              \
                lf_h_first(TQ[0], TQ[1], LF_QW((S) + 1, 0), k64, sel_lo, sel_hi, m1152u, acc[t]);                                \
            } else {                                                                                                          \
                lf_h_mul_second(FW[((S) + 1) & 1].p[0], FW[((S) + 1) & 1].p[1], TQ[2], TQ[3], TQ[0], TQ[1], LF_QD((S) + 1, 0),       \
                                LF_QW((S) + 1, 2), k64, sel_lo, sel_hi, m1152u, acc[t]);                                         \
            }                                                                                                                 \
        }                                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
    } while (0)
        LF_STEP(0);
        LF_STEP(1);
        LF_STEP(2);
        LF_STEP(3);
        LF_STEP(4);
        LF_STEP(5);
        LF_STEP(6);
        LF_STEP(7);
#undef LF_STEP
#ifdef LF_STAMPS
        st_a += t1 - t0, st_b += LF_T() - t1;
#endif
    }
#ifdef LF_STAMPS
    if (st_on && wave == 0 && lane == 0)
        g_lf_stamps[0] = st_a, g_lf_stamps[1] = st_b, g_lf_stamps[2] = (unsigned long long)nq, g_lf_stamps[3] = LF_T() - st_start;
#endif
    LF_WAIT(0); // (the reads of a stage that does not exist: slot contents never used)
    LF_XTIE(0);
    asm volatile("" : "+v"(WQ[0]), "+v"(SC));
#undef LF_WAIT
#undef LF_XREAD
#undef LF_XTIE
#undef LF_DK
#undef LF_QW
#undef LF_QD

    // ---- store: lane (i, h) holds weight row 32 rt + i, register e = token n0 + 32 t + 8 (e >> 2) + 4 h + (e & 3)
    if ((long)rt * 32 + i < m) {
        const long row = (long)rt * 32 + i;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const long tk = n0 + 32 * t + 8 * (e >> 2) + 4 * h + (e & 3);
                if (tk < n)
                    C[tk * ldc + row] = acc[t][e];
            }
    }
#ifdef LF_STAMPS
    if (st_on && wave == 0 && lane == 0)
        g_lf_stamps[4] = LF_T() - st_start;
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Activation staging: f16(d8 * q8) of Q8_0-quantised rows (quantize_row_q8_0: d = amax / 127 kept as f16, code = roundf(x / d);
// or the caller's Q8_0 blocks), written as Xh [quad][n_pad][256 B]: 16-byte chunk 2 s + h of a token and quad = K-step s = 2 jj + e,
// K half h = {block 2 e, block 2 e + 1} x {elements 4 j .. 4 j + 3}, j = 4 h + jj.  One thread per eight consecutive elements
// (two groups j = 2 o, 2 o + 1 of block bq of the quad); the block maximum over the four threads of a block by DPP.
template <bool F32IN>
__global__ __launch_bounds__(256) void prep_lf_kernel(const uint8_t *__restrict__ X, size_t x_row_bytes, long n, long n_pad, int nq,
                                                       _Float16 *__restrict__ Xh) {
    const long tok = (long)blockIdx.x;
    const int c8 = (int)blockIdx.y * 256 + (int)threadIdx.x; // eight elements 8 c8 .. 8 c8 + 7 of the row
    const bool live = c8 < nq * 16;
    const int cc = live ? c8 : 0;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float d = 0.0f;
    if (tok < n) {
        if constexpr (F32IN) {
            const float4 *p = (const float4 *)((const float *)(X + (size_t)tok * x_row_bytes) + (size_t)cc * 8);
            const float4 f0 = p[0], f1 = p[1];
            v[0] = f0.x, v[1] = f0.y, v[2] = f0.z, v[3] = f0.w, v[4] = f1.x, v[5] = f1.y, v[6] = f1.z, v[7] = f1.w;
            float am = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; e++)
                am = fmaxf(am, fabsf(v[e]));
            am = fmaxf(am, dpp_f32<DPP_XOR1>(am)); // the block's four threads are one quad of lanes
            am = fmaxf(am, dpp_f32<DPP_XOR2>(am));
            const float dd = am / 127.0f;
            const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
            d = (float)(_Float16)dd; // the scale as the block stores it
#pragma unroll
            for (int e = 0; e < 8; e++)
                v[e] = roundf(v[e] * id);
        } else {
            const lfamd_block_q8_0 *bq = (const lfamd_block_q8_0 *)(X + (size_t)tok * x_row_bytes) + (cc >> 2);
            d = h2f(bq->d);
#pragma unroll
            for (int e = 0; e < 8; e++)
                v[e] = (float)bq->qs[(cc & 3) * 8 + e];
        }
    }
    if (!live)
        return;
    const int quad = cc >> 4, blk = (cc >> 2) & 3, o = cc & 3;
    uint8_t *dst = (uint8_t *)Xh + ((size_t)quad * n_pad + tok) * 256;
#pragma unroll
    for (int g = 0; g < 2; g++) { // group j = 2 o + g: K half j >> 2, pair jj = j & 3; K-step 2 jj + (blk >> 1), second half of the chunk for odd blocks
        const int j = 2 * o + g, hh = j >> 2, jj = j & 3, s = 2 * jj + (blk >> 1);
        lf_half4 w;
#pragma unroll
        for (int e = 0; e < 4; e++)
            w[e] = (_Float16)(d * v[4 * g + e]);
        *(lf_half4 *)(dst + (2 * s + hh) * 16 + (blk & 1) * 8) = w;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// F16 / BF16 weights (the float tinyBLAS types, tinyblas_cpu.h:419-613): the same skeleton without a dequantisation — the RAW rows
// are the MFMA operand as they lie.  Work-group = 128 rows x 64 tokens; a stage = 128 weights per row: 32 KB of weight rows
// ([row][256 B], 16-byte chunks XOR-swizzled by row on the source address) + 16 KB of activations (the image of
// lfamd_launch_prep_float: [super-block][token][512 B] in the weight's type), ring of 3; four loader waves (eight with half the
// pieces each measured 3-9 % slower), four computing waves, fragments one K-step ahead.  f32 accumulate on the matrix cores like the reference's fmaf chains.
#define LFF_X 32768
#define LFF_SLOT (32768 + 16384)
template <bool BF, bool M16>
__global__ __launch_bounds__(512) void gemm_lf_float_kernel(const gemm_mats mats, int nq, size_t a_row_bytes, const uint16_t *__restrict__ Xh,
                                                                      long n, long n_pad, int n_rb, int n_ct) {
#ifdef LF_CHECK_NQ
    nq = LF_CHECK_NQ;
#endif
    constexpr int NT = 2, RING = 3;
    constexpr int WP = 8, XP = 4, PIECES = WP + XP; // of a loader and stage
    __shared__ __attribute__((aligned(16))) uint8_t lds[RING * LFF_SLOT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    int rb, ct;
    {
        const int n_wg = n_rb * n_ct;
        const int id = (int)blockIdx.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = id & 7;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
        tile_of(L, n_rb, n_ct, rb, ct);
    }
    int mj = 0;
#pragma unroll
    for (int jj = 1; jj < GEMM_MAX_MATS; jj++)
        if (jj < mats.count && rb >= mats.rb_end[jj - 1])
            mj = jj;
    if (mj > 0)
        rb -= mats.rb_end[mj - 1];
    const uint8_t *__restrict__ A = mats.A[mj];
    float *__restrict__ C = mats.C[mj];
    const long m = mats.m[mj], ldc = mats.ldc[mj];
    const long n0 = (long)ct * 64;
    const uint32_t lds0 = lds_addr(lds);

    if (wave >= 4) {
        const int lw = wave - 4;
        // weight pieces WP lw + e: rows 4 p .. 4 p + 3 of the block (rows past the matrix: its last one), lane = (row, slot) <- chunk slot ^ (row & 15)
        uint32_t voffW[WP], voffX[XP];
#pragma unroll
        for (int e = 0; e < WP; e++) {
            const int rl = 4 * (WP * lw + e) + (lane >> 4);
            const long row = (long)rb * 128 + rl;
            voffW[e] = (uint32_t)((size_t)(row < m ? row : m - 1) * a_row_bytes) + (uint32_t)(((lane & 15) ^ (rl & 15)) << 4);
        }
#pragma unroll
        for (int e = 0; e < XP; e++) {
            const int tk = 4 * (XP * lw + e) + (lane >> 4);
            voffX[e] = (uint32_t)(tk * 512 + (((lane & 15) ^ (tk & 15)) << 4));
        }
        const uint32_t dstW = (uint32_t)(WP * lw * 1024), dstX = (uint32_t)(LFF_X + XP * lw * 1024);
        const size_t xstride = (size_t)n_pad * 512;
        const uint8_t *wt_n = uniform_ptr(A);
        const uint8_t *xh_n = uniform_ptr((const uint8_t *)Xh + (size_t)n0 * 512);
        asm volatile("s_nop 4" ::"s"(wt_n), "s"(xh_n)); // (VALU-written SGPRs in front of asm loads: tools/isa_hazards.py)
        int q_n = 0;
        auto dma_stage = [&](uint32_t slot_base) {
#pragma unroll
            for (int e = 0; e < WP; e++)
                lf_dma16(wt_n, slot_base + dstW + (uint32_t)(e * 1024), voffW[e]);
#pragma unroll
            for (int e = 0; e < XP; e++)
                lf_dma16(xh_n, slot_base + dstX + (uint32_t)(e * 1024), voffX[e]);
            wt_n += 256;
            xh_n += (q_n & 1) ? xstride - 256 : 256; // the other half of the token rows / the next super-block
            q_n++;
        };
#pragma unroll
        for (int st = 0; st < RING - 1; st++)
            if (st < nq) // (uniform)
                dma_stage(lds0 + (uint32_t)(st * LFF_SLOT));
        if (nq >= 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory"); // stage 0 has landed for everybody
        int slot_i = RING - 1;
        static_assert(RING == 3, "two barriers per stage (cf. gemm_lf_q80_kernel's ring of three)");
        for (int b = 0; b < nq; b++) {
            asm volatile("s_barrier" ::: "memory"); // B1: everybody is done with stage b - 1, its slot takes stage b + 2
            if (b + 2 < nq) {
                dma_stage(lds0 + (uint32_t)(slot_i * LFF_SLOT));
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); // stage b + 1, issued a whole stage ago
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            slot_i = slot_i + 1 == RING ? 0 : slot_i + 1;
            asm volatile("s_barrier" ::: "memory"); // B2: stage b + 1 has landed for everybody
        }
        return;
    }

    const int rw = wave;
    const int rt = rb * 4 + rw;
    if constexpr (M16) {
        // The same tile on v_mfma_f32_16x16x32: a K-step is 32 weights, the wave's 32 rows x 64 tokens are 2 x 4 tiles of 16 x 16 —
        // the same operand bytes, accumulators and MFMA cycles as the 32 x 32 x 16 form; what differs is the clock the chip holds
        // under it (MI355X_MICROARCH.md, DVFS give-back item 7).  Lane (r16, kq): 16-byte chunk 4 s + kq of row / token r16.
        const int r16 = lane & 15, kq = lane >> 4;
        uint32_t aW[4], aX[4]; // K-step s: row tile 0 / token tile 0 (the others: + 4096 per tile of 16)
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
            aW[sx] = lds0 + (uint32_t)((rw * 32 + r16) * 256 + (((4 * sx + kq) ^ r16) << 4));
            aX[sx] = lds0 + (uint32_t)(LFF_X + r16 * 256 + (((4 * sx + kq) ^ r16) << 4));
        }
        float4_t_ ac[4][2];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 2; u++)
                ac[t][u] = float4_t_{0.0f, 0.0f, 0.0f, 0.0f};
        asm volatile("s_barrier" ::: "memory"); // stage 0 has landed for everybody
        u32x4 W2[2][2], X2[2][4];
#define LFF_READ(BUF, SX, BASE)                                                                                               \
    do {                                                                                                                      \
        lf_dsr16<0>(W2[BUF][0], aW[SX] + (BASE));                                                                              \
        lf_dsr16<4096>(W2[BUF][1], aW[SX] + (BASE));                                                                           \
        lf_dsr16<0>(X2[BUF][0], aX[SX] + (BASE));                                                                              \
        lf_dsr16<4096>(X2[BUF][1], aX[SX] + (BASE));                                                                           \
        lf_dsr16<8192>(X2[BUF][2], aX[SX] + (BASE));                                                                           \
        lf_dsr16<12288>(X2[BUF][3], aX[SX] + (BASE));                                                                          \
    } while (0)
        LFF_READ(0, 0, 0u);
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        int slot_c = 0;
        for (int b = 0; b < nq; b++) {
            const uint32_t so = (uint32_t)(slot_c * LFF_SLOT);
            slot_c = slot_c + 1 == RING ? 0 : slot_c + 1;
            const uint32_t so_n = (uint32_t)(slot_c * LFF_SLOT);
            asm volatile("s_barrier" ::: "memory"); // B1: everybody is done with stage b - 1
#define LFF_STEP16(S)                                                                                                         \
    do {                                                                                                                      \
        if ((S) == 3)                                                                                                         \
            asm volatile("s_barrier" ::: "memory"); /* B2: stage b + 1 has landed for everybody */                            \
        LFF_READ(((S) + 1) & 1, ((S) + 1) & 3, (S) == 3 ? so_n : so);                                                          \
        asm volatile("s_waitcnt lgkmcnt(6)"                                                                                   \
                     : "+v"(W2[(S)&1][0]), "+v"(W2[(S)&1][1]), "+v"(X2[(S)&1][0]), "+v"(X2[(S)&1][1]), "+v"(X2[(S)&1][2]),      \
                       "+v"(X2[(S)&1][3]));                                                                                    \
        _Pragma("unroll") for (int t = 0; t < 4; t++) _Pragma("unroll") for (int u = 0; u < 2; u++) {                            \
            if constexpr (BF)                                                                                                 \
                ac[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, X2[(S)&1][t]),                  \
                                                                   __builtin_bit_cast(bf16x8_t, W2[(S)&1][u]), ac[t][u], 0, 0, 0); \
            else                                                                                                              \
                ac[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, X2[(S)&1][t]),                    \
                                                                  __builtin_bit_cast(half8_t, W2[(S)&1][u]), ac[t][u], 0, 0, 0);  \
        }                                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
    } while (0)
            LFF_STEP16(0);
            LFF_STEP16(1);
            LFF_STEP16(2);
            LFF_STEP16(3);
#undef LFF_STEP16
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(W2[0][0]), "+v"(W2[0][1]), "+v"(X2[0][0]), "+v"(X2[0][1]), "+v"(X2[0][2]), "+v"(X2[0][3])); // (a stage that does not exist)
#undef LFF_READ
        // lane (r16, kq) holds weight row 32 rt + 16 u + r16, register j = token n0 + 16 t + 4 kq + j
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const long row = (long)rt * 32 + 16 * u + r16;
            if (row < m)
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const long tk = n0 + 16 * t + 4 * kq + j;
                        if (tk < n)
                            C[tk * ldc + row] = ac[t][u][j];
                    }
        }
        return;
    }
    uint32_t adW[8], adX[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        adW[s] = lds0 + (uint32_t)((rw * 32 + i) * 256 + (((2 * s + h) ^ (i & 15)) << 4));
        adX[s] = lds0 + (uint32_t)(LFF_X + i * 256 + (((2 * s + h) ^ (i & 15)) << 4));
    }
    float16_t_ acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 16; e++)
            acc[t][e] = 0.0f;
    asm volatile("s_barrier" ::: "memory"); // stage 0 has landed for everybody
    // fragments ONE K-step ahead: in step s the wave issues W(s + 1), X(s + 1) [from the NEXT slot once s = 7: landed, the barrier at
    // the top of this stage said so] and waits for step s's: younger are exactly this step's three reads
    u32x4 WF[2], XF[2][NT];
    lf_dsr16<0>(WF[0], adW[0]);
    lf_dsr16<0>(XF[0][0], adX[0]);
    lf_dsr16<8192>(XF[0][1], adX[0]);
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    int slot_c = 0;
    for (int b = 0; b < nq; b++) {
        const uint32_t so = (uint32_t)(slot_c * LFF_SLOT);
        slot_c = slot_c + 1 == RING ? 0 : slot_c + 1;
        const uint32_t so_n = (uint32_t)(slot_c * LFF_SLOT);
        asm volatile("s_barrier" ::: "memory"); // B1: everybody is done with stage b - 1
#define LFF_STEP(S)                                                                                                           \
    do {                                                                                                                      \
        if ((S) == 7)                                                                                                         \
            asm volatile("s_barrier" ::: "memory"); /* B2: stage b + 1 has landed for everybody */                            \
        lf_dsr16<0>(WF[((S) + 1) & 1], adW[((S) + 1) & 7] + ((S) == 7 ? so_n : so));                                            \
        lf_dsr16<0>(XF[((S) + 1) & 1][0], adX[((S) + 1) & 7] + ((S) == 7 ? so_n : so));                                         \
        lf_dsr16<8192>(XF[((S) + 1) & 1][1], adX[((S) + 1) & 7] + ((S) == 7 ? so_n : so));                                      \
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(WF[(S)&1]), "+v"(XF[(S)&1][0]), "+v"(XF[(S)&1][1]));                          \
        _Pragma("unroll") for (int t = 0; t < NT; t++) {                                                                       \
            if constexpr (BF)                                                                                                 \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, XF[(S)&1][t]),                    \
                                                                 __builtin_bit_cast(bf16x8_t, WF[(S)&1]), acc[t], 0, 0, 0);     \
            else                                                                                                              \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, XF[(S)&1][t]),                      \
                                                                __builtin_bit_cast(half8_t, WF[(S)&1]), acc[t], 0, 0, 0);       \
        }                                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
    } while (0)
        LFF_STEP(0);
        LFF_STEP(1);
        LFF_STEP(2);
        LFF_STEP(3);
        LFF_STEP(4);
        LFF_STEP(5);
        LFF_STEP(6);
        LFF_STEP(7);
#undef LFF_STEP
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(WF[0]), "+v"(XF[0][0]), "+v"(XF[0][1])); // (reads of a stage that does not exist)
    if ((long)rt * 32 + i < m) {
        const long row = (long)rt * 32 + i;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const long tk = n0 + 32 * t + 8 * (e >> 2) + 4 * h + (e & 3);
                if (tk < n)
                    C[tk * ldc + row] = acc[t][e];
            }
    }
}

// A: RAW rows of m x k 16-bit values (row stride a_row_bytes, 16-byte aligned); Xh: the image lfamd_launch_prep_float wrote.
extern "C" hipError_t lfamd_launch_gemm_lf_float(int Atype, const void *A, size_t a_row_bytes, long m, long k, const void *Xh, long n, long n_pad,
                                                 float *C, long ldc, hipStream_t s) {
    if (n <= 0 || m <= 0)
        return hipSuccess;
    if (k % 256 || (a_row_bytes & 15) || ((uintptr_t)A & 15) || (Atype != LFAMD_TYPE_F16 && Atype != LFAMD_TYPE_BF16))
        return hipErrorInvalidValue;
    gemm_mats mats;
    const int n_rb = (int)((m + 127) / 128), n_ct = (int)((n + 63) / 64);
    mats.count = 1;
    mats.moe_cnt = mats.moe_poff = mats.moe_slot_row = nullptr, mats.expert_bytes = 0, mats.moe_ct_max = 0;
    for (int q = 0; q < GEMM_MAX_MATS; q++)
        mats.A[q] = (const uint8_t *)A, mats.C[q] = C, mats.m[q] = q ? 0 : m, mats.ldc[q] = q ? 0 : ldc, mats.rb_end[q] = n_rb;
    const int nq = (int)(k / 128);
    const unsigned grid = (unsigned)(n_rb * n_ct);
    static const bool m16 = !(getenv("LFAMD_LFF_M16") && atoi(getenv("LFAMD_LFF_M16")) == 0); // (A/B runs: 0 = the 32 x 32 x 16 form)
    if (Atype == LFAMD_TYPE_BF16) {
        if (m16)
            gemm_lf_float_kernel<true, true><<<grid, 512, 0, s>>>(mats, nq, a_row_bytes, (const uint16_t *)Xh, n, n_pad, n_rb, n_ct);
        else
            gemm_lf_float_kernel<true, false><<<grid, 512, 0, s>>>(mats, nq, a_row_bytes, (const uint16_t *)Xh, n, n_pad, n_rb, n_ct);
    } else {
        if (m16)
            gemm_lf_float_kernel<false, true><<<grid, 512, 0, s>>>(mats, nq, a_row_bytes, (const uint16_t *)Xh, n, n_pad, n_rb, n_ct);
        else
            gemm_lf_float_kernel<false, false><<<grid, 512, 0, s>>>(mats, nq, a_row_bytes, (const uint16_t *)Xh, n, n_pad, n_rb, n_ct);
    }
    return hipGetLastError();
}

// Which launches take this body: Q8_0, rows of whole quads.  Tile width: see lf_nt.
static int lf_nt(long row_blocks128, long n) {
    static const int force = getenv("LFAMD_LF_NT") ? atoi(getenv("LFAMD_LF_NT")) : 0; // A/B runs
    if (force == 2 || force == 4)
        return force;
    // 128 x 64 tiles while they fit ONE round of the 256 CUs, else 128 x 128 (4096 x 4096 x 512: 37.2 against 43.3 us; 4608 x 4096: 55.3
    // against 44.8; 6144 x 4096: 53.7 against 44.1; 14336 x 4096: 116 against 97) — the two give the same bits
    return row_blocks128 * ((n + 63) / 64) <= 256 ? 2 : 4;
}

extern "C" size_t lfamd_gemm_lf_workspace(long k, long n) { // Xh
    const size_t n_pad = ((size_t)n + 127) / 128 * 128;
    return n_pad * (size_t)k * 2;
}

// B: f32 rows or Q8_0 blocks; ws: lfamd_gemm_lf_workspace(k, n) bytes; A[j]: P80 images of m[j] x k.
extern "C" hipError_t lfamd_launch_gemm_lf_q80(int count, const void *const *A, const long *m, long k, int Btype, const void *B,
                                               size_t b_row_bytes, long n, float *const *C, const long *ldc, void *ws, hipStream_t s) {
    if (n <= 0 || count <= 0)
        return hipSuccess;
    if (count > GEMM_MAX_MATS || k % 128 || (Btype != LFAMD_TYPE_F32 && Btype != LFAMD_TYPE_Q8_0))
        return hipErrorInvalidValue;
    const int nq = (int)(k / 128);
    const long n_pad = (n + 127) / 128 * 128;
    _Float16 *Xh = (_Float16 *)ws;
    const dim3 pg((unsigned)n_pad, (unsigned)((nq * 16 + 255) / 256));
    if (Btype == LFAMD_TYPE_F32)
        prep_lf_kernel<true><<<pg, 256, 0, s>>>((const uint8_t *)B, b_row_bytes, n, n_pad, nq, Xh);
    else
        prep_lf_kernel<false><<<pg, 256, 0, s>>>((const uint8_t *)B, b_row_bytes, n, n_pad, nq, Xh);
    gemm_mats mats;
    int n_rb = 0;
    mats.count = 0;
    mats.moe_cnt = mats.moe_poff = mats.moe_slot_row = nullptr, mats.expert_bytes = 0, mats.moe_ct_max = 0;
    for (int j = 0; j < count; j++) {
        if (m[j] <= 0)
            continue;
        if ((size_t)((m[j] + 7) / 8) * (size_t)nq * P80_TILE >= ((size_t)1 << 32)) // (the loaders address a tile by a 32-bit byte offset)
            return hipErrorInvalidValue;
        const int q = mats.count++;
        mats.A[q] = (const uint8_t *)A[j], mats.C[q] = C[j], mats.m[q] = m[j], mats.ldc[q] = ldc[j];
        n_rb += (int)((m[j] + 127) / 128);
        mats.rb_end[q] = n_rb;
    }
    if (mats.count == 0)
        return hipGetLastError();
    for (int q = mats.count; q < GEMM_MAX_MATS; q++)
        mats.A[q] = mats.A[0], mats.C[q] = mats.C[0], mats.m[q] = 0, mats.ldc[q] = 0, mats.rb_end[q] = n_rb;
    if (lf_nt(n_rb, n) == 4) {
        const int n_ct = (int)((n + 127) / 128);
        gemm_lf_q80_kernel<4><<<(unsigned)(n_rb * n_ct), 512, 0, s>>>(mats, nq, Xh, n, n_pad, n_rb, n_ct);
    } else {
        const int n_ct = (int)((n + 63) / 64);
        gemm_lf_q80_kernel<2><<<(unsigned)(n_rb * n_ct), 512, 0, s>>>(mats, nq, Xh, n, n_pad, n_rb, n_ct);
    }
    return hipGetLastError();
}
