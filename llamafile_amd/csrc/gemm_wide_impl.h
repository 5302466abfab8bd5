// gemm_wide.hip — prefill GEMM, "wide" body: 128 x 128 output tile per work-group, K streamed once.
//
// Same arithmetic as gemm_mfma.hip (exact integer codes on v_mfma_f32_32x32x16_f16, the two f32 scales applied
// once per super-block; reference: mul_mat_qX_K_q8_K_T, iqk_mul_mat.inc:601-643), different shape of the work:
//
//   * work-group = 8 waves = 4 row tiles x 2 column halves on a 128-row x 128-token output tile, full K per
//     wave (no intra-work-group split-K, no LDS combine, ONE barrier per super-block).  Two waves per SIMD: the
//     LDS-DMA / ds_read issue of one wave overlaps the MFMA + VALU stream of the other (a 4-wave, one-wave-per-
//     SIMD variant with 32 x 128 per wave was issue-bound: every LDS-DMA piece blocks the only wave 60-180 cycles).
//   * K is split across work-groups (KS = 1, 2, 4 ...) when the tile grid alone cannot fill 256 CUs; partial
//     tiles meet in C with f32 atomic adds onto zeros (the activation-prep kernel zeroes C).  With KS = 2 the
//     result is deterministic (0 + a + b, and a + b commutes), so KS is capped at 2.
//   * all loads are issued in inline asm (LDS-DMA for the activation tile / d8 / mins operand, global_load
//     with an SGPR base for the weights), one super-block ahead, and retired by ONE s_waitcnt vmcnt(0) +
//     s_barrier per super-block: hipcc's waitcnt pass cannot count LDS-DMA and otherwise drains the prefetch
//     in the middle of the MFMA phase (cdna_hip_programming.md §5.7).  Per-lane source offsets are loop
//     invariant (16 VGPRs); the super-block advance is an SGPR add — no VALU address arithmetic in the loop.
#pragma once
#include "gemm_common.h"
#include <type_traits>
#ifndef GEMM_DIAG
#define GEMM_DIAG 0
#endif

#define WD_COLS 128
#define WD_XSTAGE (WD_COLS * XT_ROW_BYTES) // 64 KiB of f16 codes per super-block

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int TYPE>
struct wide_w {
    u32x4 qs[4];
    u32x4 hd;    // Q4_K: {d, dmin, scales[12]};  Q6_K: 16 int8 scales
    u32x4 qh[2]; // Q6_K only
    uint32_t dw; // Q6_K only (f16 bits)
    u32x4 q8[4]; // IQ4_XS byte image: K-steps 8..15 (qs[] holds 0..7)
    u32x4 f[16]; // F16 / BF16: the 16 K-step fragments of the lane's row, straight from the RAW row
};

template <int IMM>
__device__ static inline void gload16(u32x4 &dst, const void *base, uint32_t voff) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int IMM>
__device__ static inline void gload4(uint32_t &dst, const void *base, uint32_t voff) {
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int IMM>
__device__ static inline void gload2(uint32_t &dst, const void *base, uint32_t voff) {
    asm volatile("s_nop 4\n\tglobal_load_ushort %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}

// four LDS-DMA pieces (1 KiB each) to consecutive LDS slots from one SGPR base + per-lane offsets
__device__ static inline void glds4(const void *base, uint32_t lds_dst, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3) {
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(lds_dst)
                 : "memory", "scc");
}
__device__ static inline void glds1x16(const void *base, uint32_t lds_dst, uint32_t o0) {
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(o0), "s"(base), "s"(lds_dst)
                 : "memory");
}
__device__ static inline void glds1x4(const void *base, uint32_t lds_dst, uint32_t o0) {
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(o0), "s"(base), "s"(lds_dst)
                 : "memory");
}

// make a wave-uniform pointer provably so (an "s" operand must be), via two v_readfirstlane
__device__ static inline const uint8_t *uniform_ptr(const void *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// (gemm_ks.hip / gemm_kr.hip use LEAN load forms whose asm carries no wait states of its own: a VALU write of an SGPR —
// v_readfirstlane — needs five wait states before a vector-memory instruction reads that SGPR, and hipcc's hazard recogniser
// does not see inside an asm statement.  They follow uniform_ptr() with an `s_nop 4` fence that consumes the pointers;
// tools/isa_hazards.py checks the rule on the final ISA.)

// LDS fragment read hipcc neither counts nor moves: the K loop below keeps the next K-step's four fragments in
// flight under the current step's MFMAs and waits with a counted lgkmcnt (left to itself hipcc reuses ONE
// fragment register and waits lgkmcnt(0) in front of every MFMA — the whole LDS latency, 64 times per super-block)
template <int IMM>
__device__ static inline void dsr16(half8_t &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}
template <int N>
__device__ static inline void ds_wait(half8_t &a, half8_t &b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

__device__ static inline uint32_t lds_addr(const void *p) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)(const uint8_t *)p);
}

#if GEMM_DIAG == 3
__device__ unsigned long long g_wide_stamps[8 * 64];
extern "C" int lfamd_debug_wide_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wide_stamps), sizeof(g_wide_stamps));
}
#endif

// Up to GEMM_MAX_MATS weight matrices of one type and row length that consume the SAME activations (attn_q/k/v,
// ffn_gate/up) share one prep and one launch: their 128-row blocks are concatenated (rb_end = exclusive prefix).
#define GEMM_MAX_MATS 4
struct gemm_mats {
    const uint8_t *A[GEMM_MAX_MATS];
    float *C[GEMM_MAX_MATS];
    long m[GEMM_MAX_MATS];
    long ldc[GEMM_MAX_MATS];
    int rb_end[GEMM_MAX_MATS];
    int count;
    // GGML_OP_MUL_MAT_ID batches (MOE kernels only): A[0] = the expert stack, C[0] = result rows; token slots are grouped
    // by expert on the device (moe.hip, moe_route_kernel): expert e owns slots [poff[e], poff[e] + cnt[e]), slot -> result
    // row through slot_row.  The grid covers the worst case; work-groups beyond an expert's count exit at once.
    const int *moe_cnt, *moe_poff, *moe_slot_row;
    long expert_bytes;
    int moe_ct_max;
};

template <int TYPE, bool MOE = false>
__global__ __launch_bounds__(512) void gemm_wide_kernel(const gemm_mats mats, int nb, const _Float16 *__restrict__ Xh,
                                                        const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, long n,
                                                        long n_pad, int n_rb, int n_ct, int ks_n, int nbs) {
    __shared__ __attribute__((aligned(16))) uint8_t xs[2][WD_XSTAGE];     // activation codes, XOR-swizzled rows
    // 32-blocks: f16 scale per block, Q8_0 / Q8_1 activations (8 d8 per 256).  Q4_0 is resident in P40; Q4_1 / Q5_0 / Q5_1
    // come as a per-call PCL image (generic.hip, wprep32): L5 = fifth bits, L1 = w = d*q + m with the s = d8*sum(q8) term
    constexpr bool LEGACY = TYPE == LFAMD_TYPE_Q4_0 || TYPE == LFAMD_TYPE_Q4_1 || TYPE == LFAMD_TYPE_Q5_0 || TYPE == LFAMD_TYPE_Q5_1;
    constexpr bool L5 = TYPE == LFAMD_TYPE_Q5_0 || TYPE == LFAMD_TYPE_Q5_1;
    constexpr bool L1 = TYPE == LFAMD_TYPE_Q4_1 || TYPE == LFAMD_TYPE_Q5_1;
    constexpr float LOFF = TYPE == LFAMD_TYPE_Q4_0 ? 8.0f : TYPE == LFAMD_TYPE_Q5_0 ? 16.0f : 0.0f;
    // PCK image built per call (generic.hip, wprep16): 16-wide sub-blocks, int8 scale each; Q2_K also 16 mins
    constexpr bool CANON16 = TYPE == LFAMD_TYPE_Q2_K || TYPE == LFAMD_TYPE_Q3_K;
    constexpr bool MINS16 = TYPE == LFAMD_TYPE_Q2_K;
    constexpr bool BYTES8 = TYPE == LFAMD_TYPE_IQ4_XS; // PC8 byte image built per call (generic.hip, wprep8)
    constexpr bool BLK8 = LEGACY; // eight 32-blocks per super-block: d8 per block
    // float tinyBLAS types (tinyblas_cpu.h:419-613): no dequantisation, no scales — the lane's 8 consecutive halves of a
    // K-step are 16 contiguous bytes of the RAW row; f32 accumulate on the matrix cores like the reference's fmaf chains
    constexpr bool FLT = TYPE == LFAMD_TYPE_F16 || TYPE == LFAMD_TYPE_BF16;
    __shared__ __attribute__((aligned(16))) float d8s[2][(BLK8 ? 8 : 1) * WD_COLS]; // d8 of the 128 tokens
    __shared__ __attribute__((aligned(16))) uint8_t xms[2][WD_COLS * 32]; // Q4_K mins operand rows; L1: the 8 x 128 f32 s values
    constexpr int TILE = (TYPE == LFAMD_TYPE_Q4_K || TYPE == LFAMD_TYPE_Q4_0) ? P4K_TILE : LEGACY ? PCL_TILE : TYPE == LFAMD_TYPE_Q5_K ? P5K_TILE : CANON16 ? PCK_TILE : BYTES8 ? PC8_TILE : P6K_TILE;
    constexpr bool MINS = TYPE == LFAMD_TYPE_Q4_K || TYPE == LFAMD_TYPE_Q5_K; // Q4_K family: {d, dmin, 6-bit scales/mins}
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;

    int ks, rb, ct, moe_left = 0;
    const uint8_t *__restrict__ A;
    float *__restrict__ C;
    long m, ldc, n0;
    if constexpr (MOE) {
        // (expert, row block, token tile of that expert); the routing kernel ran earlier on this stream
        // block id = (token tile, expert, row block), row blocks fastest: consecutive ids go round-robin over the XCDs,
        // so the live tiles (low token-tile index) spread over all eight instead of piling onto XCD 0 and 1, and they
        // are dispatched before the tiles that only exit
        const int per_ct = (int)(gridDim.x / mats.moe_ct_max); // experts * n_rb
        ct = blockIdx.x / per_ct;
        const int rem = blockIdx.x - ct * per_ct;
        const int e = rem / n_rb;
        rb = rem - e * n_rb;
        const int cnt_e = mats.moe_cnt[e];
        moe_left = cnt_e - ct * WD_COLS; // token slots of this tile that carry a row
        if (moe_left <= 0)
            return; // uniform over the work-group
        ks = 0;
        A = mats.A[0] + (size_t)e * mats.expert_bytes;
        C = mats.C[0];
        m = mats.m[0], ldc = mats.ldc[0];
        n0 = (long)mats.moe_poff[e] + (long)ct * WD_COLS;
    } else {
        // XCD-aware order: block ids go round-robin over the 8 XCDs; give each XCD a contiguous run of the order
        // (K-split index slowest, then super-tiles of 8 x 4 tiles, see tile_of)
        const int n_tiles = n_rb * n_ct, n_wg = n_tiles * ks_n;
        const int id = blockIdx.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = id & 7;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
        ks = L / n_tiles;
        tile_of(L - ks * n_tiles, n_rb, n_ct, rb, ct);
        int mj = 0;
#pragma unroll
        for (int jj = 1; jj < GEMM_MAX_MATS; jj++)
            if (jj < mats.count && rb >= mats.rb_end[jj - 1])
                mj = jj;
        if (mj > 0)
            rb -= mats.rb_end[mj - 1];
        A = mats.A[mj];
        C = mats.C[mj];
        m = mats.m[mj], ldc = mats.ldc[mj];
        n0 = (long)ct * WD_COLS;
    }
    const long n_row_tiles = (m + 31) / 32;
    const int rw = wave & 3, ch = wave >> 2; // row tile and 64-token column half of this wave
    const long rt = (long)rb * 4 + rw;
    const bool active = rt < n_row_tiles;
    const int b0 = ks * nbs, b1 = min(nb, b0 + nbs), nit = b1 - b0; // this work-group's super-blocks
    if (nit <= 0)
        return; // uniform over the work-group

    // ---- loop-invariant per-lane offsets
    // activation pieces: wave-instruction e of this wave copies token rows 32*wave + 2e + h; lane slot p = i of a
    // row receives logical 16-B chunk p ^ (row & 15).  Pieces 4q..4q+3 share one M0 setting.
    uint32_t xo[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int nn = 16 * wave + 2 * e + h;
        xo[e] = (uint32_t)(nn * 512 + ((i ^ (nn & 15)) * 16));
    }
    const uint32_t wo = lane * 16, ho = i * 16 + ((MINS || LEGACY || CANON16) ? P4K_HDR : P6K_SC) - 4096; // P5K_HDR == PCK_SC == P4K_HDR
    const uint32_t xmo = (uint32_t)((32 * (wave & 3) + (lane >> 1)) * 32 + (lane & 1) * 16);
    const uint32_t d8o = (uint32_t)((lane >> 5) * n_pad * 4 + (lane & 31) * 16); // LEGACY d8 rows
    const uint8_t *xbase = (const uint8_t *)Xh + (size_t)n0 * 512;           // + b * n_pad * 512
    const uint8_t *wbase = A + (size_t)(active ? rt : 0) * nb * (FLT ? 32 * 512 : TILE);
    constexpr int WSTEP = FLT ? 512 : TILE; // bytes from one super-block of this wave's rows to the next
    // FLT: this lane's row of the tile (clamped to the matrix: the RAW tensor has exactly m rows), 8 h halves in
    const long frow = min((active ? rt : 0) * 32 + (long)i, m - 1) - (active ? rt : 0) * 32;
    const uint32_t fo = (uint32_t)(frow * (long)nb * 512 + h * 16);
    (void)fo;
    const uint8_t *xmbase = (const uint8_t *)Xm + (size_t)n0 * 32;           // + b * n_pad * 32
    const uint32_t xs_a[2] = {lds_addr(xs[0]), lds_addr(xs[1])};
    const uint32_t d8_a[2] = {lds_addr(d8s[0]), lds_addr(d8s[1])};
    const uint32_t xm_a[2] = {lds_addr(xms[0]), lds_addr(xms[1])};

    auto prefetch = [&](int b, int st, wide_w<TYPE> &w) {
#if GEMM_DIAG == 2 // development: compute chain only (operands loaded twice)
        if (b > b0 + 1)
            return;
#endif
        const uint8_t *xb = uniform_ptr(xbase + (size_t)b * n_pad * 512);
        const uint32_t dst = xs_a[st] + wave * 8192;
        glds4(xb, dst, xo[0], xo[1], xo[2], xo[3]);
        glds4(xb, dst + 4096, xo[4], xo[5], xo[6], xo[7]);
        if constexpr (FLT) {
        } else if constexpr (BLK8) { // 8 rows of 128 f32: waves 4..7 copy two rows each (one 16-byte piece)
            if (wave >= 4)
                glds1x16(uniform_ptr(d8T + ((size_t)b * 8 + 2 * (wave - 4)) * n_pad + n0), d8_a[st] + (wave - 4) * 1024, d8o);
        } else if (wave >= 6) // d8 of tokens 64*(wave-6) + lane
            glds1x4(uniform_ptr(d8T + (size_t)b * n_pad + n0), d8_a[st] + (wave - 6) * 256, (uint32_t)((wave - 6) * 256 + lane * 4));
        if constexpr (MINS || MINS16) {
            if (wave < 4)
                glds1x16(uniform_ptr(xmbase + (size_t)b * n_pad * 32), xm_a[st] + wave * 1024, xmo);
        }
        if constexpr (L1) { // Xm carries sT [nb*8][n_pad] f32 here
            if (wave < 4)
                glds1x16(uniform_ptr((const float *)Xm + ((size_t)b * 8 + 2 * wave) * n_pad + n0), xm_a[st] + wave * 1024, d8o);
        }
        const uint8_t *tile = uniform_ptr(wbase + (size_t)b * WSTEP);
        const uint8_t *tile_h = uniform_ptr(tile + 4096), *tile_d = uniform_ptr(tile + P6K_D);
        if constexpr (!FLT) {
            gload16<0>(w.qs[0], tile, wo);
            gload16<1024>(w.qs[1], tile, wo);
            gload16<2048>(w.qs[2], tile, wo);
            gload16<3072>(w.qs[3], tile, wo);
        }
        if constexpr (FLT) {
            gload16<0>(w.f[0], tile, fo), gload16<32>(w.f[1], tile, fo), gload16<64>(w.f[2], tile, fo), gload16<96>(w.f[3], tile, fo);
            gload16<128>(w.f[4], tile, fo), gload16<160>(w.f[5], tile, fo), gload16<192>(w.f[6], tile, fo), gload16<224>(w.f[7], tile, fo);
            gload16<256>(w.f[8], tile, fo), gload16<288>(w.f[9], tile, fo), gload16<320>(w.f[10], tile, fo), gload16<352>(w.f[11], tile, fo);
            gload16<384>(w.f[12], tile, fo), gload16<416>(w.f[13], tile, fo), gload16<448>(w.f[14], tile, fo), gload16<480>(w.f[15], tile, fo);
        } else if constexpr (BYTES8) {
            gload16<0>(w.q8[0], tile_h, wo);
            gload16<1024>(w.q8[1], tile_h, wo);
            gload16<2048>(w.q8[2], tile_h, wo);
            gload16<3072>(w.q8[3], tile_h, wo);
            gload16<0>(w.hd, uniform_ptr(tile + PC8_HDR), (uint32_t)(i * 16));
        } else if constexpr (CANON16) {
            gload16<0>(w.hd, tile_h, ho);            // 16 int8 scales
            if constexpr (MINS16)
                gload16<512>(w.qh[0], tile_h, ho);   // 16 uint8 mins (PCK_MN)
            gload4<1024>(w.dw, tile_h, (uint32_t)(i * 4)); // {d, dmin} (PCK_D)
        } else if constexpr (MINS || LEGACY) {
            gload16<0>(w.hd, tile_h, ho);
            if constexpr (TYPE == LFAMD_TYPE_Q5_K)
                gload16<512>(w.qh[0], tile_h, wo);
            if constexpr (L1)
                gload16<512>(w.qh[1], tile_h, ho);  // eight f16 m (PCL_M)
            if constexpr (L5)
                gload16<1024>(w.qh[0], tile_h, wo); // fifth bits (PCL_QH) // P5K_QH = 4608
        } else {
            gload16<0>(w.qh[0], tile_h, wo);    // P6K_QH = 4096
            gload16<1024>(w.qh[1], tile_h, wo);
            gload16<0>(w.hd, tile_h, ho);
            gload2<0>(w.dw, tile_d, (uint32_t)(i * 2));
        }
    };
    // The same loads, spread over the K-steps of the super-block being computed: an LDS-DMA piece blocks the
    // issuing wave for 100-200 cycles (measured with s_memtime: a burst of 10 right after the barrier cost every
    // wave 1000-1800 cycles with both waves of a SIMD stalled together); one piece per K-step stalls one wave
    // while its SIMD partner keeps the MFMA pipe busy.  Weights (HBM latency) first, activation pieces next, the
    // small operands last; nothing is issued during the last five K-steps so the data lands before the barrier.
    auto prefetch_step = [&](int t, int b, int st, wide_w<TYPE> &w) {
        if (t == 0) {
            const uint8_t *tile = uniform_ptr(wbase + (size_t)b * WSTEP);
            const uint8_t *tile_h = uniform_ptr(tile + 4096), *tile_d = uniform_ptr(tile + P6K_D);
            if constexpr (!FLT) {
                gload16<0>(w.qs[0], tile, wo);
                gload16<1024>(w.qs[1], tile, wo);
                gload16<2048>(w.qs[2], tile, wo);
                gload16<3072>(w.qs[3], tile, wo);
            }
            if constexpr (FLT) {
                gload16<0>(w.f[0], tile, fo), gload16<32>(w.f[1], tile, fo), gload16<64>(w.f[2], tile, fo), gload16<96>(w.f[3], tile, fo);
                gload16<128>(w.f[4], tile, fo), gload16<160>(w.f[5], tile, fo), gload16<192>(w.f[6], tile, fo), gload16<224>(w.f[7], tile, fo);
                gload16<256>(w.f[8], tile, fo), gload16<288>(w.f[9], tile, fo), gload16<320>(w.f[10], tile, fo), gload16<352>(w.f[11], tile, fo);
                gload16<384>(w.f[12], tile, fo), gload16<416>(w.f[13], tile, fo), gload16<448>(w.f[14], tile, fo), gload16<480>(w.f[15], tile, fo);
            } else if constexpr (BYTES8) {
                gload16<0>(w.q8[0], tile_h, wo);
                gload16<1024>(w.q8[1], tile_h, wo);
                gload16<2048>(w.q8[2], tile_h, wo);
                gload16<3072>(w.q8[3], tile_h, wo);
                gload16<0>(w.hd, uniform_ptr(tile + PC8_HDR), (uint32_t)(i * 16));
            } else if constexpr (CANON16) {
                gload16<0>(w.hd, tile_h, ho);            // 16 int8 scales
                if constexpr (MINS16)
                    gload16<512>(w.qh[0], tile_h, ho);   // 16 uint8 mins (PCK_MN)
                gload4<1024>(w.dw, tile_h, (uint32_t)(i * 4)); // {d, dmin} (PCK_D)
            } else if constexpr (MINS || LEGACY) {
                gload16<0>(w.hd, tile_h, ho);
                if constexpr (TYPE == LFAMD_TYPE_Q5_K)
                    gload16<512>(w.qh[0], tile_h, wo);
                if constexpr (L1)
                    gload16<512>(w.qh[1], tile_h, ho);  // eight f16 m (PCL_M)
                if constexpr (L5)
                    gload16<1024>(w.qh[0], tile_h, wo); // fifth bits (PCL_QH)
            } else {
                gload16<0>(w.qh[0], tile_h, wo);
                gload16<1024>(w.qh[1], tile_h, wo);
                gload16<0>(w.hd, tile_h, ho);
                gload2<0>(w.dw, tile_d, (uint32_t)(i * 2));
            }
        } else if (t <= 8) {
            const int e = t - 1;
            glds1x16(uniform_ptr(xbase + (size_t)b * n_pad * 512), xs_a[st] + wave * 8192 + e * 1024, xo[e]);
        } else if (t == 9) {
            if constexpr (FLT) {
            } else if constexpr (BLK8) {
                if (wave >= 4)
                    glds1x16(uniform_ptr(d8T + ((size_t)b * 8 + 2 * (wave - 4)) * n_pad + n0), d8_a[st] + (wave - 4) * 1024, d8o);
            } else if (wave >= 6)
                glds1x4(uniform_ptr(d8T + (size_t)b * n_pad + n0), d8_a[st] + (wave - 6) * 256, (uint32_t)((wave - 6) * 256 + lane * 4));
        } else if (t == 10) {
            if constexpr (MINS || MINS16) {
                if (wave < 4)
                    glds1x16(uniform_ptr(xmbase + (size_t)b * n_pad * 32), xm_a[st] + wave * 1024, xmo);
            }
            if constexpr (L1) {
                if (wave < 4)
                    glds1x16(uniform_ptr((const float *)Xm + ((size_t)b * 8 + 2 * wave) * n_pad + n0), xm_a[st] + wave * 1024, d8o);
            }
        }
    };
    // retire every load of the stage (this wave's), then meet the other waves: their LDS-DMA has landed too, and
    // everybody has finished reading the stage that the next prefetch overwrites
    auto arrive = [&](wide_w<TYPE> &w) {
        if constexpr (FLT)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier"
                         : "+v"(w.f[0]), "+v"(w.f[1]), "+v"(w.f[2]), "+v"(w.f[3]), "+v"(w.f[4]), "+v"(w.f[5]), "+v"(w.f[6]), "+v"(w.f[7]),
                           "+v"(w.f[8]), "+v"(w.f[9]), "+v"(w.f[10]), "+v"(w.f[11]), "+v"(w.f[12]), "+v"(w.f[13]), "+v"(w.f[14]),
                           "+v"(w.f[15])
                         :
                         : "memory");
        else if constexpr (L1 || L5)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier"
                         : "+v"(w.qs[0]), "+v"(w.qs[1]), "+v"(w.qs[2]), "+v"(w.qs[3]), "+v"(w.hd), "+v"(w.qh[0]), "+v"(w.qh[1])
                         :
                         : "memory");
        else if constexpr (TYPE == LFAMD_TYPE_Q4_K || LEGACY)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier"
                         : "+v"(w.qs[0]), "+v"(w.qs[1]), "+v"(w.qs[2]), "+v"(w.qs[3]), "+v"(w.hd)
                         :
                         : "memory");
        else if constexpr (BYTES8)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier"
                         : "+v"(w.qs[0]), "+v"(w.qs[1]), "+v"(w.qs[2]), "+v"(w.qs[3]), "+v"(w.hd), "+v"(w.q8[0]), "+v"(w.q8[1]),
                           "+v"(w.q8[2]), "+v"(w.q8[3])
                         :
                         : "memory");
        else if constexpr (CANON16)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier"
                         : "+v"(w.qs[0]), "+v"(w.qs[1]), "+v"(w.qs[2]), "+v"(w.qs[3]), "+v"(w.hd), "+v"(w.qh[0]), "+v"(w.dw)
                         :
                         : "memory");
        else if constexpr (TYPE == LFAMD_TYPE_Q5_K)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier"
                         : "+v"(w.qs[0]), "+v"(w.qs[1]), "+v"(w.qs[2]), "+v"(w.qs[3]), "+v"(w.hd), "+v"(w.qh[0])
                         :
                         : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier"
                         : "+v"(w.qs[0]), "+v"(w.qs[1]), "+v"(w.qs[2]), "+v"(w.qs[3]), "+v"(w.hd), "+v"(w.qh[0]),
                           "+v"(w.qh[1]), "+v"(w.dw)
                         :
                         : "memory");
    };

    float16_t_ acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[nt][r] = 0.0f;

    // byte offset of this lane's fragment chunk inside a token row: chunk (c ^ (i & 15)) with c = 2t + h
    uint32_t xoff[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
        xoff[u] = xs_a[0] + ch * 32768 + (uint32_t)(i * XT_ROW_BYTES + ((((2 * u + h) & 15) ^ (i & 15)) * 16));
    const float16_t_ zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t magic = opaque_magic();
    const uint32_t d8_lane = d8_a[0] + (uint32_t)((ch * 64 + 4 * h) * 4);
    (void)d8_lane;

    auto compute = [&](auto stc, const wide_w<TYPE> &w, int bn, wide_w<TYPE> &wn) {
        constexpr int st = decltype(stc)::value;
#if GEMM_DIAG == 1 // development: memory + barrier chain only
        acc[0][0] += (float)(w.qs[0].x ^ w.qs[1].y ^ w.qs[2].z ^ w.qs[3].w ^ w.hd.x) + d8s[st][lane];
        return;
#endif
        // the four token tiles' fragments of K-step t (all offsets immediates; stage 1 is 64 KiB up: in the address)
        auto read_frags = [&](half8_t(&f)[2], int t) {
            const uint32_t a = xoff[t & 7] + (st ? WD_XSTAGE : 0);
            if ((2 * t) & 16) {
                dsr16<256>(f[0], a);
                dsr16<16384 + 256>(f[1], a);
            } else {
                dsr16<0>(f[0], a);
                dsr16<16384>(f[1], a);
            }
        };
        float16_t_ tmp[2];
        const uint32_t qw[16] = {w.qs[0].x, w.qs[0].y, w.qs[0].z, w.qs[0].w, w.qs[1].x, w.qs[1].y, w.qs[1].z, w.qs[1].w,
                                 w.qs[2].x, w.qs[2].y, w.qs[2].z, w.qs[2].w, w.qs[3].x, w.qs[3].y, w.qs[3].z, w.qs[3].w};
        if constexpr (FLT) {
            typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
            half8_t F[2][2];
            read_frags(F[0], 0);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                if (t + 1 < 16) {
                    read_frags(F[(t + 1) & 1], t + 1);
                    ds_wait<2>(F[t & 1][0], F[t & 1][1]);
                } else {
                    ds_wait<0>(F[t & 1][0], F[t & 1][1]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    if constexpr (TYPE == LFAMD_TYPE_F16)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t & 1][nt], __builtin_bit_cast(half8_t, w.f[t]), acc[nt], 0, 0, 0);
                    else
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, F[t & 1][nt]),
                                                                          __builtin_bit_cast(bf16x8_t, w.f[t]), acc[nt], 0, 0, 0);
                }
                prefetch_step(t, bn, st ^ 1, wn);
            }
        } else if constexpr (MINS) {
            const uint32_t hq5[4] = {w.qh[0].x, w.qh[0].y, w.qh[0].z, w.qh[0].w}; // Q5_K only
            (void)hq5;
            const float d = h2f((uint16_t)(w.hd.x & 0xffff)), dmin = h2f((uint16_t)(w.hd.x >> 16));
            uint32_t sc03, sc47, mn03, mn47;
            q4k_scales_bytes(w.hd.y, w.hd.z, w.hd.w, sc03, sc47, mn03, mn47);
            half8_t F[2][2];
            read_frags(F[0], 0);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                q4_consts2 cp; // constants of sub-blocks j & ~1, j | 1 (recomputed identically for the odd j: CSE'd)
                cp = q4_consts_pair(j < 4 ? sc03 : sc47, (j & 2) ? 2 : 0);
                const int hsel = j & 1;
                const half2_t S = {cp.S[hsel], cp.S[hsel]}, O = {cp.O[hsel], cp.O[hsel]};
                const half2_t S16 = {cp.S16[hsel], cp.S16[hsel]}, O16 = {cp.O16[hsel], cp.O16[hsel]};
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int t = 2 * j + e;
                    half8_t wf;
                    if constexpr (TYPE == LFAMD_TYPE_Q5_K)
                        wf = dequant_q5(qw[t], hq5[t >> 2] >> (t & 3), S, O, S16, O16, magic);
                    else
                        wf = dequant_q4(qw[t], S, O, S16, O16, magic);
                    if (t + 1 < 16) {
                        read_frags(F[(t + 1) & 1], t + 1);
                        ds_wait<2>(F[t & 1][0], F[t & 1][1]);
                    } else {
                        ds_wait<0>(F[t & 1][0], F[t & 1][1]);
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t & 1][nt], wf, t == 0 ? zero16 : tmp[nt], 0, 0, 0);
                    prefetch_step(t, bn, st ^ 1, wn);
                }
            }
            // mins: one MFMA per token tile, K = 16 = {lo parts | hi parts} of the 8 pair sums
            frag_u wm;
            const float mscale = h ? 64.0f : 1.0f;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t mw = p < 2 ? mn03 : mn47;
                const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
                half2_t v = {(_Float16)(m0 * mscale), (_Float16)(m1 * mscale)};
                wm.p[p] = v;
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const half8_t xm = *(const half8_t *)(xms[st] + (ch * 64 + nt * 32 + i) * 32 + h * 16);
                const float16_t_ tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wm.v, zero16, 0, 0, 0);
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const float4_t_ d8 = *(const float4_t_ *)(&d8s[st][ch * 64 + nt * 32 + 8 * r4 + 4 * h]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        const float u = fmaf(-dmin, tm[r], d * tmp[nt][r]);
                        acc[nt][r] = fmaf(u, d8[e], acc[nt][r]);
                    }
                }
            }
        } else if constexpr (BYTES8) {
            // IQ4_XS: w = d * sc_j * kvalues[code], eight 32-wide sub-blocks; the byte image holds kvalues[code] + 128
            const uint32_t bw[32] = {w.qs[0].x, w.qs[0].y, w.qs[0].z, w.qs[0].w, w.qs[1].x, w.qs[1].y, w.qs[1].z, w.qs[1].w,
                                     w.qs[2].x, w.qs[2].y, w.qs[2].z, w.qs[2].w, w.qs[3].x, w.qs[3].y, w.qs[3].z, w.qs[3].w,
                                     w.q8[0].x, w.q8[0].y, w.q8[0].z, w.q8[0].w, w.q8[1].x, w.q8[1].y, w.q8[1].z, w.q8[1].w,
                                     w.q8[2].x, w.q8[2].y, w.q8[2].z, w.q8[2].w, w.q8[3].x, w.q8[3].y, w.q8[3].z, w.q8[3].w};
            const float d = h2f((uint16_t)(w.hd.z & 0xffff));
            half8_t F[2][2];
            read_frags(F[0], 0);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float scf = (float)(int)(int8_t)(((j < 4 ? w.hd.x : w.hd.y) >> (8 * (j & 3))) & 0xff);
                const half2_t S = bcast_h2(scf), O = bcast_h2(-1152.0f * scf);
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int t = 2 * j + e;
                    const half8_t wf = dequant_bytes(bw[2 * t], bw[2 * t + 1], S, O);
                    if (t + 1 < 16) {
                        read_frags(F[(t + 1) & 1], t + 1);
                        ds_wait<2>(F[t & 1][0], F[t & 1][1]);
                    } else {
                        ds_wait<0>(F[t & 1][0], F[t & 1][1]);
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t & 1][nt], wf, t == 0 ? zero16 : tmp[nt], 0, 0, 0);
                    prefetch_step(t, bn, st ^ 1, wn);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const float4_t_ d8 = *(const float4_t_ *)(&d8s[st][ch * 64 + nt * 32 + 8 * r4 + 4 * h]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        acc[nt][r] = fmaf(d * tmp[nt][r], d8[e], acc[nt][r]);
                    }
                }
        } else if constexpr (CANON16) {
            // w = d * sc_t * q - dmin * mn_t per 16-wide sub-block t (one K-step); operand sc_t * (code - OFF), exact in f16
            constexpr float OFF = TYPE == LFAMD_TYPE_Q3_K ? 4.0f : 0.0f;
            const uint32_t scw[4] = {w.hd.x, w.hd.y, w.hd.z, w.hd.w};
            const float d = h2f((uint16_t)(w.dw & 0xffff)), dmin = h2f((uint16_t)(w.dw >> 16));
            half8_t F[2][2];
            read_frags(F[0], 0);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const float scf = (float)(int)(int8_t)((scw[t >> 2] >> (8 * (t & 3))) & 0xff);
                const half2_t S = bcast_h2(scf);
                half8_t wf;
                if constexpr (OFF != 0.0f) {
                    wf = dequant_q4_off(qw[t], S, OFF, magic);
                } else {
                    const half2_t O = bcast_h2(-1024.0f * scf), S16 = bcast_h2(scf * 0.0625f), O16 = bcast_h2(-64.0f * scf);
                    wf = dequant_q4(qw[t], S, O, S16, O16, magic);
                }
                if (t + 1 < 16) {
                    read_frags(F[(t + 1) & 1], t + 1);
                    ds_wait<2>(F[t & 1][0], F[t & 1][1]);
                } else {
                    ds_wait<0>(F[t & 1][0], F[t & 1][1]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
                    tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t & 1][nt], wf, t == 0 ? zero16 : tmp[nt], 0, 0, 0);
                prefetch_step(t, bn, st ^ 1, wn);
            }
            frag_u wm; // Q2_K: the 16 mins of this lane's row, half h = K entries 8h .. 8h+7 (matching the bsums operand)
            if constexpr (MINS16) {
                const uint32_t mw0 = h ? w.qh[0].z : w.qh[0].x, mw1 = h ? w.qh[0].w : w.qh[0].y;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t mw = p < 2 ? mw0 : mw1;
                    half2_t v = {(_Float16)(float)((mw >> (16 * (p & 1))) & 0xff), (_Float16)(float)((mw >> (16 * (p & 1) + 8)) & 0xff)};
                    wm.p[p] = v;
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                float16_t_ tm = zero16;
                if constexpr (MINS16) {
                    const half8_t xm = *(const half8_t *)(xms[st] + (ch * 64 + nt * 32 + i) * 32 + h * 16);
                    tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wm.v, zero16, 0, 0, 0);
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const float4_t_ d8 = *(const float4_t_ *)(&d8s[st][ch * 64 + nt * 32 + 8 * r4 + 4 * h]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        const float u = MINS16 ? fmaf(-dmin, tm[r], d * tmp[nt][r]) : d * tmp[nt][r];
                        acc[nt][r] = fmaf(u, d8[e], acc[nt][r]);
                    }
                }
            }
        } else if constexpr (LEGACY) {
            const uint32_t hdw[4] = {w.hd.x, w.hd.y, w.hd.z, w.hd.w}; // eight f16 block scales of this lane's row
            const uint32_t mdw[4] = {w.qh[1].x, w.qh[1].y, w.qh[1].z, w.qh[1].w}; // L1: eight f16 m
            const uint32_t hq5[4] = {w.qh[0].x, w.qh[0].y, w.qh[0].z, w.qh[0].w}; // L5: fifth bits
            (void)mdw;
            (void)hq5;
            const uint32_t s8base = xm_a[0] + (uint32_t)((ch * 64 + 4 * h) * 4);
            (void)s8base;
            const uint32_t d8base = d8_lane; // LDS address of d8s[0][ch * 64 + 4 * h]
            half8_t F[2][2];
            read_frags(F[0], 0);
#pragma unroll
            for (int bl = 0; bl < 8; bl++) {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int t = 2 * bl + e;
                    half8_t wf;
                    if constexpr (TYPE == LFAMD_TYPE_Q4_0)
                        wf = dequant_q40(qw[t], magic);
                    else
                        wf = dequant_legacy<L5>(qw[t], L5 ? (hq5[t >> 2] >> (t & 3)) : 0u, LOFF, magic);
                    if (t + 1 < 16) {
                        read_frags(F[(t + 1) & 1], t + 1);
                        ds_wait<2>(F[t & 1][0], F[t & 1][1]);
                    } else {
                        ds_wait<0>(F[t & 1][0], F[t & 1][1]);
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t & 1][nt], wf, e == 0 ? zero16 : tmp[nt], 0, 0, 0);
                    prefetch_step(t, bn, st ^ 1, wn);
                }
                // per 32-block: acc += (<q - 8, q8> * d8[token]) * d[row].  The d8 reads are asm: hipcc hoists plain LDS
                // reads of all eight blocks above the MFMAs and then spills ~500 registers — among them asm-loaded
                // weight registers whose data has not landed yet (garbage).
                const float dbl = h2f((uint16_t)((bl & 1) ? (hdw[bl >> 1] >> 16) : (hdw[bl >> 1] & 0xffff)));
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    // one loop-invariant base register, everything else an immediate (computed addresses get hoisted
                    // out of the K loop by the dozen and spilled)
                    float4_t_ d8[4];
                    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%5+32\n\t"
                                 "ds_read_b128 %2, %4 offset:%5+64\n\tds_read_b128 %3, %4 offset:%5+96\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(d8[0]), "=&v"(d8[1]), "=&v"(d8[2]), "=&v"(d8[3])
                                 : "v"(d8base), "n"(st * 8 * WD_COLS * 4 + bl * WD_COLS * 4 + nt * 128));
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int r = 4 * r4 + e;
                            acc[nt][r] = fmaf(tmp[nt][r] * d8[r4][e], dbl, acc[nt][r]);
                        }
                    if constexpr (L1) { // + m[row] * s[token]   (iqk_mul_mat.inc:1110-1127, MinusType1)
                        float4_t_ s8[4];
                        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%5+32\n\t"
                                     "ds_read_b128 %2, %4 offset:%5+64\n\tds_read_b128 %3, %4 offset:%5+96\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(s8[0]), "=&v"(s8[1]), "=&v"(s8[2]), "=&v"(s8[3])
                                     : "v"(s8base), "n"(st * 8 * WD_COLS * 4 + bl * WD_COLS * 4 + nt * 128));
                        const float mbl = h2f((uint16_t)((bl & 1) ? (mdw[bl >> 1] >> 16) : (mdw[bl >> 1] & 0xffff)));
#pragma unroll
                        for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int r = 4 * r4 + e;
                                acc[nt][r] = fmaf(mbl, s8[r4][e], acc[nt][r]);
                            }
                    }
                }
                // pin the scaling HERE: an empty volatile asm on the accumulators is ordered before the next K-step's
                // asm reads.  Left free, hipcc sinks all eight blocks' scaling below the MFMAs, keeps 8 x (tmp + d8)
                // live and spills ~500 registers — among them asm-loaded weight registers whose data has not landed.
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
            }
        } else {
            const uint32_t hw[8] = {w.qh[0].x, w.qh[0].y, w.qh[0].z, w.qh[0].w, w.qh[1].x, w.qh[1].y, w.qh[1].z, w.qh[1].w};
            const uint32_t scw[4] = {w.hd.x, w.hd.y, w.hd.z, w.hd.w};
            const float dw = h2f((uint16_t)w.dw);
            half8_t F[2][2];
            read_frags(F[0], 0);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const float scf = (float)(int)(int8_t)((scw[t >> 2] >> (8 * (t & 3))) & 0xff);
                const half2_t S = bcast_h2(scf);
                uint32_t H = hw[t >> 1];
                if (t & 1)
                    H >>= 2;
                const half8_t wf = dequant_q6(qw[t], H, S);
                if (t + 1 < 16) {
                    read_frags(F[(t + 1) & 1], t + 1);
                    ds_wait<2>(F[t & 1][0], F[t & 1][1]);
                } else {
                    ds_wait<0>(F[t & 1][0], F[t & 1][1]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
                    tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t & 1][nt], wf, t == 0 ? zero16 : tmp[nt], 0, 0, 0);
                prefetch_step(t, bn, st ^ 1, wn);
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const float4_t_ d8 = *(const float4_t_ *)(&d8s[st][ch * 64 + nt * 32 + 8 * r4 + 4 * h]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        acc[nt][r] = fmaf(dw * tmp[nt][r], d8[e], acc[nt][r]);
                    }
                }
        }
    };

    // ---- pipeline: super-block it+1 is in flight (registers w[(it+1)&1], LDS stage (it+1)&1) while it computes
#if GEMM_DIAG == 3 // development: in-kernel time stamps of work-group 0 (s_memtime; never in the product build)
#define STAMP(slot)                                                                                              \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (blockIdx.x == 0 && lane == 0 && stamp_n < 60)                                                        \
            g_wide_stamps[wave * 64 + stamp_n++] = __builtin_amdgcn_s_memtime();                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
    int stamp_n = 0;
#else
#define STAMP(slot)
#endif
    wide_w<TYPE> wa, wb;
    STAMP(0);
    prefetch(b0, 0, wa);
    for (int it = 0; it < nit; it += 2) {
        STAMP(1);
        arrive(wa);
        STAMP(2);
        STAMP(3);
        compute(std::integral_constant<int, 0>{}, wa, b0 + (it + 1 < nit ? it + 1 : it), wb); // clamped: a redundant reload at the end
        if (it + 1 < nit) { // uniform
            STAMP(1);
            arrive(wb);
            STAMP(2);
            STAMP(3);
            compute(std::integral_constant<int, 1>{}, wb, b0 + (it + 2 < nit ? it + 2 : it + 1), wa);
        }
    }
    STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the clamped tail prefetch must not outlive the work-group's LDS

    // ---- store: reg r of token tile nt is token n0 + 32nt + (r&3) + 8(r>>2) + 4h, weight row 32rt + i
    if (active) {
        const long row = rt * 32 + i;
        if (row < m) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int tl = ch * 64 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const long tok = n0 + tl;
                    if constexpr (MOE) {
                        if (tl < moe_left)
                            C[(long)mats.moe_slot_row[tok] * ldc + row] = acc[nt][r];
                    } else if (tok < n) {
                        if (ks_n == 1)
                            C[tok * ldc + row] = acc[nt][r];
                        else
                            __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(C + tok * ldc + row),
                                                                    acc[nt][r]);
                    }
                }
        }
    }
}


// ---- one translation unit per group of weight types (gemm_wide_*.hip) instantiates the kernel through this stamp, so the
// ~20 instantiations compile in parallel instead of serially in one file
#define WIDE_ARGS                                                                                                      \
    const gemm_mats &mats, int nb, const void *Xh, const void *d8T, const void *Xm, long n, long n_pad, int n_rb, int n_ct,    \
        int ks, int nbs, unsigned n_wg, int moe, hipStream_t s
#define WIDE_INSTANTIATE(NAME, TYPE)                                                                                   \
    hipError_t lfamd_wide_go_##NAME(WIDE_ARGS) {                                                                       \
        (void)moe;                                                                                                     \
        gemm_wide_kernel<TYPE, false><<<n_wg, 512, 0, s>>>(mats, nb, (const _Float16 *)Xh, (const float *)d8T,         \
                                                           (const _Float16 *)Xm, n, n_pad, n_rb, n_ct, ks, nbs);       \
        return hipGetLastError();                                                                                      \
    }
#define WIDE_INSTANTIATE_MOE(NAME, TYPE)                                                                               \
    hipError_t lfamd_wide_go_##NAME(WIDE_ARGS) {                                                                       \
        if (moe)                                                                                                       \
            gemm_wide_kernel<TYPE, true><<<n_wg, 512, 0, s>>>(mats, nb, (const _Float16 *)Xh, (const float *)d8T,      \
                                                              (const _Float16 *)Xm, n, n_pad, n_rb, n_ct, ks, nbs);    \
        else                                                                                                           \
            gemm_wide_kernel<TYPE, false><<<n_wg, 512, 0, s>>>(mats, nb, (const _Float16 *)Xh, (const float *)d8T,     \
                                                               (const _Float16 *)Xm, n, n_pad, n_rb, n_ct, ks, nbs);   \
        return hipGetLastError();                                                                                      \
    }
