// gemm_kr.hip — prefill GEMM, large grids: 256 weight rows x 128 tokens per work-group on the resident Q4_K layout, scaled
// operands (the arithmetic of gemm_lw.hip / gemm_ks.hip: f16(d * sc * q) x f16(d8 * code * 2^-e(token)), mins as one more
// MFMA per token tile and super-block, output column times 2^e in the store; reference: mul_mat_qX_K_q8_K_T,
// iqk_mul_mat.inc:601-643; <= 1e-3 against the oracle).
//
// The sibling of gemm_ks.hip for shapes whose 128 x 128 tiles more than fill the chip (ffn_gate / ffn_up, output.weight at
// 512 tokens).  What the stamps and ablations of gemm_ks showed bounds these kernels is the number of instructions a wave
// issues per MFMA, not the matrix pipe, the ingest path or LDS: one dequantised weight fragment (9 VALU) must feed as many
// MFMAs as possible, and every fragment read, wait and load instruction must be shared by as many MFMAs as possible.  So:
//   * a wave owns 32 rows x 128 tokens: one dequantised fragment -> FOUR MFMAs (2.25 VALU per MFMA; the 128 x 64 tile: 4.5);
//   * eight such waves = 256 rows per work-group, no split of K between waves (no exchange at the end), all eight compute
//     AND load: weights HBM -> VGPR (private to the wave), activations by LDS-DMA (shared), as in gemm_ks.hip;
//   * the 128-token activation stage is 64 KiB per super-block — a ring of four does not fit LDS — so a period is HALF a
//     super-block: 32 KiB of f16 operands ([128 tokens][256 B], 16-byte chunks XOR-swizzled by token on the source address)
//     + the mins operand of 64 of the tokens, four stages in LDS, one s_barrier per period, eight K-steps = 32 MFMAs per
//     wave and period;
//   * waves 4..7 run HALF A PERIOD behind waves 0..3 (K-steps 4..7 of a period's stage are carried over the next barrier,
//     their fragments fetched before it): the two waves of a SIMD never stand at a barrier, a cold start or a constants
//     block together;
//   * a 256 x 128 tile takes in 50 KB per period for 256 MFMAs (the 128 x 128 tile of gemm_lw: 41 KB for 128).
#include "gemm_wide_impl.h"

#define KR_XM 32768
#define KR_SLOT (32768 + 2048)
#define KR_STAGES 4
#define KR_COLS 128

template <int IMM>
__device__ static inline void kr_dma16(const void *base, uint32_t lds_dst, uint32_t voff) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst), "n"(IMM) : "memory", "scc");
}
__device__ static inline void kr_dma4(const void *base, uint32_t lds_dst, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int IMM>
__device__ static inline void kr_ld16(u32x4 &dst, const void *base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int IMM>
__device__ static inline void kr_dsr16(half8_t &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

#ifdef KR_BOUNDS // development: every global access checked against its buffer; the first violation is recorded, not executed
__device__ unsigned long long g_kr_oob[8];
extern "C" int lfamd_debug_kr_oob(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_kr_oob), sizeof(g_kr_oob));
}
#define KR_CHECK(what, base, voff, imm, size, lo, hi)                                                                  \
    do {                                                                                                               \
        const unsigned long long a_ = (unsigned long long)(uintptr_t)(base) + (unsigned long long)(voff) + (imm);      \
        if (a_ < (unsigned long long)(uintptr_t)(lo) || a_ + (size) > (unsigned long long)(uintptr_t)(hi)) {           \
            if (atomicCAS(&g_kr_oob[0], 0ull, 1ull + (what)) == 0ull) {                                                \
                g_kr_oob[1] = a_, g_kr_oob[2] = (unsigned long long)(uintptr_t)(lo), g_kr_oob[3] = (unsigned long long)(uintptr_t)(hi); \
                g_kr_oob[4] = ((unsigned long long)blockIdx.x << 32) | threadIdx.x, g_kr_oob[5] = (unsigned long long)(dbg_period);  \
            }                                                                                                          \
            bad_ = true;                                                                                               \
        }                                                                                                              \
    } while (0)
#endif

// mats.rb_end counts 256-ROW blocks here (lfamd_kr_go)
// MOE: GGML_OP_MUL_MAT_ID batches (gemm_mats: token slots grouped by expert; the grid covers the worst case, work-groups beyond an
// expert's rows exit at once; slot -> result row through moe_slot_row)
template <int TYPE, bool MOE = false>
__global__ __launch_bounds__(512) void gemm_kr_kernel(const gemm_mats mats, int nb, const _Float16 *__restrict__ Xh,
                                                      const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, long n,
                                                      long n_pad, int n_rb, int n_ct) {
    static_assert(TYPE == LFAMD_TYPE_Q4_K, "resident P4K layout");
#ifdef KS_CHECK_NB // tools/isa_hazards.py: a fixed trip count unrolls the period loop into straight-line code
    nb = KS_CHECK_NB;
#endif
    __shared__ __attribute__((aligned(16))) uint8_t lds[KR_STAGES * KR_SLOT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int np = 2 * nb; // periods

    // ---- tile of this work-group: XCD-aware super-tiles (gemm_common.h tile_of) over (256-row block, 128-token tile)
    int rb, ct, mj = 0, moe_left = 0;
    long n0;
    const uint8_t *__restrict__ A;
    if constexpr (MOE) { // (expert, row block) fastest, token tile slowest: the order of gemm_lw's grouped launch
        const int per_ct = (int)gridDim.x / mats.moe_ct_max;
        ct = (int)blockIdx.x / per_ct;
        const int rem = (int)blockIdx.x - ct * per_ct;
        const int e = rem / n_rb;
        rb = rem - e * n_rb;
        moe_left = mats.moe_cnt[e] - ct * KR_COLS;
        if (moe_left <= 0)
            return;
        A = mats.A[0] + (size_t)e * mats.expert_bytes;
        n0 = (long)mats.moe_poff[e] + (long)ct * KR_COLS;
    } else {
        const int n_wg = n_rb * n_ct;
        const int id = (int)blockIdx.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = id & 7;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
        tile_of(L, n_rb, n_ct, rb, ct);
#pragma unroll
        for (int jj = 1; jj < GEMM_MAX_MATS; jj++)
            if (jj < mats.count && rb >= mats.rb_end[jj - 1])
                mj = jj;
        if (mj > 0)
            rb -= mats.rb_end[mj - 1];
        A = mats.A[mj];
        n0 = (long)ct * KR_COLS;
    }
    float *__restrict__ C = mats.C[mj];
    const long m = mats.m[mj], ldc = mats.ldc[mj];
    const long n_row_tiles = (m + 31) / 32;
    const long rt = (long)rb * 8 + wave;
    const bool active = rt < n_row_tiles;
    const int ntl = MOE ? __builtin_amdgcn_readfirstlane(moe_left >= KR_COLS ? 4 : (moe_left + 31) >> 5) : 4; // live 32-token quarters
    const uint32_t lds0 = lds_addr(lds);

    float16_t_ acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[nt][r] = 0.0f;
    const uint32_t magic = opaque_magic();

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // Everything from the first load to the last wait exists TWICE, once per wave group (the two run different schedules):
    // no register with a load in flight ever crosses a branch or a join (tools/isa_hazards.py; cf. gemm_ks.hip).
    auto run = [&](auto grpc) {
        constexpr int GRP = decltype(grpc)::value;
        // ---- this wave's loads.  Period p = (super-block p >> 1, half p & 1) lives in LDS slot / register set p & 3.
        const uint32_t wlo = (uint32_t)lane * 16, hlo = (uint32_t)i * 16 + P4K_HDR; // (the header lies beyond the 12-bit offset field)
        // activations: pieces 4 wave .. 4 wave + 3 of the period's 32; piece q = token rows 4 q .. 4 q + 3 of this half,
        // lane = (row + (lane >> 4), 16-byte slot lane & 15) <- logical chunk slot ^ (row & 15)
        uint32_t xo[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int row = 4 * (4 * wave + e) + (lane >> 4);
            xo[e] = (uint32_t)(row * 512 + (((lane & 15) ^ (row & 15)) * 16));
        }
        const uint32_t xd0 = (uint32_t)(4 * wave * 1024);
        const uint32_t xmd = (uint32_t)(KR_XM + 256 * wave), xmo = (uint32_t)(256 * wave + lane * 4);
        // source pointers of the NEXT period to copy / to load, in SGPRs, advanced by scalar adds; clamped at the last period
        const size_t xstride = (size_t)n_pad * 512, xmstride = (size_t)n_pad * 32;
        const uint8_t *xs_n = uniform_ptr((const uint8_t *)Xh + (size_t)n0 * 512);
        const uint8_t *xm_n = uniform_ptr((const uint8_t *)Xm + (size_t)n0 * 32);
        const uint8_t *wt_n = uniform_ptr(A + (size_t)(active ? rt : 0) * nb * P4K_TILE); // tile of the super-block to load next
    // five wait states between the VALU writes of those SGPRs (v_readfirstlane) and the first vector-memory instruction that
    // reads them: hipcc pads such hazards itself, but not in front of an asm statement (tools/isa_hazards.py checks the ISA)
    asm volatile("s_nop 4" ::"s"(xs_n), "s"(xm_n), "s"(wt_n));
        int px_n = 0, bw_n = 0; // the period the activation pointers are at, the super-block wt_n is at
        auto advance_x = [&]() {
            if (px_n + 1 < np) { // (uniform) half 0 -> half 1: the other 256 bytes of the token rows / the other 64 tokens' mins
                if (px_n & 1)
                    xs_n += xstride - 256, xm_n += xmstride - 2048;
                else
                    xs_n += 256, xm_n += 2048;
            }
            px_n++;
        };
        auto advance_w = [&]() { // behind the loads of a super-block's SECOND half (clamped: every address stays inside the tensor)
            if (bw_n + 1 < nb)
                wt_n += P4K_TILE, bw_n++;
        };
#ifdef KR_BOUNDS
        int dbg_period = 0; // (counts mem_op calls)
#endif
        u32x4 qa[4], qb[4], hd[2]; // weights of period p in set p & 3; the row header of super-block b in hd[b & 1]
        // the loads a wave issues in period p (set = p & 3): j = 0..3 its pieces of DMA(p + 2), 4 its mins-operand piece
        // (into slot (p + 2) & 3), 5, 6 W(p + 3) (into set (p + 3) & 3), 7 — when p + 3 starts a super-block, i.e. p odd — its
        // row header
        auto mem_op = [&](int j, auto setc) {
            constexpr int set = decltype(setc)::value;
            constexpr int wset = (set + 3) & 3, WH = (set + 3) & 1; // the register set and the K half of period p + 3
            const uint32_t slot = lds0 + (uint32_t)(((set + 2) & 3) * KR_SLOT);
#ifdef KR_BOUNDS
            {
                bool bad_ = false;
                const uint8_t *xh_lo = (const uint8_t *)Xh, *xh_hi = xh_lo + (size_t)nb * n_pad * 512;
                const uint8_t *xm_lo = (const uint8_t *)Xm, *xm_hi = xm_lo + (size_t)nb * n_pad * 32;
                const uint8_t *a_lo = A, *a_hi = A + (size_t)n_row_tiles * nb * P4K_TILE;
                if (j < 4)
                    KR_CHECK(10 + j, xs_n, xo[j], 0, 16, xh_lo, xh_hi);
                else if (j == 4)
                    KR_CHECK(14, xm_n, xmo, 0, 4, xm_lo, xm_hi);
                else if (j == 5)
                    KR_CHECK(15, wt_n, wlo, WH * 2048, 16, a_lo, a_hi);
                else if (j == 6)
                    KR_CHECK(16, wt_n, wlo, WH * 2048 + 1024, 16, a_lo, a_hi);
                else if (WH == 0)
                    KR_CHECK(17, wt_n, hlo, 0, 16, a_lo, a_hi);
                if (__builtin_amdgcn_ballot_w64(bad_)) { // (uniform) skip the instruction, keep the pointer bookkeeping
                    if (j == 4)
                        advance_x();
                    if (j == 6 && WH == 1)
                        advance_w();
                    dbg_period++;
                    return;
                }
                dbg_period++;
            }
#endif
            if (j == 0)
                kr_dma16<0>(xs_n, slot + xd0, xo[0]);
            else if (j == 1)
                kr_dma16<1024>(xs_n, slot + xd0, xo[1]);
            else if (j == 2)
                kr_dma16<2048>(xs_n, slot + xd0, xo[2]);
            else if (j == 3)
                kr_dma16<3072>(xs_n, slot + xd0, xo[3]);
            else if (j == 4) {
                kr_dma4(xm_n, slot + xmd, xmo);
                advance_x();
            } else if (j == 5)
                kr_ld16<WH * 2048>(qa[wset], wt_n, wlo);
            else if (j == 6) {
                kr_ld16<WH * 2048 + 1024>(qb[wset], wt_n, wlo);
                if constexpr (WH == 1)
                    advance_w();
            } else if constexpr (WH == 0) { // p + 3 starts a super-block: its row header -> hd[((p + 3) >> 1) & 1] (set 1 -> 0, set 3 -> 1)
                kr_ld16<0>(hd[set == 1 ? 0 : 1], wt_n, hlo);
            }
        };
        // ---- compute state
        // fragment chunk of K-step t8: token row i, logical chunk 2 t8 + h.  Slots 0 / 1 are reached through the instruction's
        // offset field, slots 2 / 3 from the address plus two slots (one v_add per K-step: the register file is full)
        uint32_t xoffA[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            xoffA[u] = lds0 + (uint32_t)(i * 256 + (((2 * u + h) ^ (i & 15)) * 16));
        const uint32_t xm_offA = lds0 + (uint32_t)(KR_XM + i * 32 + h * 16);
        half8_t F[2][4], fxm[2]; // fragments: period-local K-step j uses F[j & 1], fetched one K-step (>= 128 MFMA cycles) ahead
        half8_t wfc;             // (group 1) the weight fragment carried over a barrier
        struct p_consts {
            q4_consts2 cp0, cp1; // this period's four sub-blocks
        };
        p_consts K2[2]; // period p uses K2[p & 1], computed at the end of period p - 1
        frag_u wm[2];   // mins weights of super-block b in wm[b & 1]: f16(-dmin * m_j) in the lower K half of the operand
        // constants of period p (set = p & 3) from the row header of its super-block; on even p also the mins weights
        auto make_consts = [&](auto setc) {
            constexpr int set = decltype(setc)::value;
            constexpr int half = set & 1, hsel = (set >> 1) & 1; // super-block parity of periods 0,1 | 2,3
            const u32x4 &hdr = hd[hsel];
            uint32_t sc03, sc47, mn03, mn47;
            q4k_scales_bytes(hdr.y, hdr.z, hdr.w, sc03, sc47, mn03, mn47);
            const uint32_t scw = half ? sc47 : sc03;
            const half2_t dh2 = as_half2(__builtin_amdgcn_perm(hdr.x, hdr.x, 0x01000100u));
            K2[set & 1].cp0 = q4_consts_pair_scaled(scw, 0, dh2), K2[set & 1].cp1 = q4_consts_pair_scaled(scw, 2, dh2);
            if constexpr (half == 0) {
                const float ndmin = -h2f((uint16_t)(hdr.x >> 16));
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t mw = p < 2 ? mn03 : mn47;
                    const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
                    const half2_t v = {(_Float16)(h ? 0.0f : m0 * ndmin), (_Float16)(h ? 0.0f : m1 * ndmin)};
                    wm[hsel].p[p] = v;
                }
            }
        };
        auto read_frags = [&](half8_t(&f)[4], auto setc, int t8) {
            constexpr int set = decltype(setc)::value;
            constexpr int SOFF = (set & 1) * KR_SLOT;
            const uint32_t a = set < 2 ? xoffA[t8] : xoffA[t8] + 2 * KR_SLOT;
            kr_dsr16<SOFF>(f[0], a);
            kr_dsr16<SOFF + 8192>(f[1], a);
            kr_dsr16<SOFF + 16384>(f[2], a);
            kr_dsr16<SOFF + 24576>(f[3], a);
        };
        // the mins operand of 64 tokens: part 0 (token tiles 0, 1) came with the super-block's first stage (slot set - 1 in the odd
        // period that uses it), part 1 (token tiles 2, 3) with the second
        auto read_mins = [&](auto setc, auto partc) {
            constexpr int set = decltype(setc)::value, slot = decltype(partc)::value ? set : (set + 3) & 3;
            constexpr int SOFF = (slot & 1) * KR_SLOT;
            const uint32_t a = slot < 2 ? xm_offA : xm_offA + 2 * KR_SLOT;
            kr_dsr16<SOFF>(fxm[0], a);
            kr_dsr16<SOFF + 1024>(fxm[1], a);
        };
        auto dq = [&](auto setc, int t8) -> half8_t {
            constexpr int set = decltype(setc)::value;
            const p_consts &kc = K2[set & 1];
            const uint32_t qw[8] = {qa[set].x, qa[set].y, qa[set].z, qa[set].w, qb[set].x, qb[set].y, qb[set].z, qb[set].w};
            const q4_consts2 &cp = (t8 & 4) ? kc.cp1 : kc.cp0;
            const int hs = (t8 >> 1) & 1;
            const half2_t S = {cp.S[hs], cp.S[hs]}, O = {cp.O[hs], cp.O[hs]};
            const half2_t S16 = {cp.S16[hs], cp.S16[hs]}, O16 = {cp.O16[hs], cp.O16[hs]};
            return dequant_q4(qw[t8], S, O, S16, O16, magic);
        };
        // MOE: an expert's LAST token tile usually holds a handful of slots (an expert of a 512-token top-2 batch over 8 experts has
        // 128 +- 11): its dead 32-token quarters skip their MFMAs (wave-uniform branches; the fragments are still fetched, so every
        // counted wait stays as it is).  Such a work-group still streams and dequantises its 256 rows, but retires in about half the time.
        auto mfma4 = [&](const half8_t &wf, half8_t(&f)[4]) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, f[0], acc[0], 0, 0, 0);
            if (!MOE || ntl > 1)
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, f[1], acc[1], 0, 0, 0);
            if (!MOE || ntl > 2)
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, f[2], acc[2], 0, 0, 0);
            if (!MOE || ntl > 3)
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, f[3], acc[3], 0, 0, 0);
        };
        // All four mins MFMAs of a super-block sit between K-steps 3 and 4 of its SECOND stage — the same place in every token
        // tile's sum, so a token's result does not depend on where in the tile it stands (tests permute the tokens and compare
        // bits).  Part 0 is in flight (fetched at the start of this K-step group); part 1 is fetched here.
        auto mfma_mins = [&](auto setc) {
            constexpr int set = decltype(setc)::value;
            constexpr int hsel = (set >> 1) & 1;
            static_assert((set & 1) == 1, "second stage of the super-block");
            asm volatile("" : "+v"(fxm[0]), "+v"(fxm[1]));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm[hsel].v, fxm[0], acc[0], 0, 0, 0);
            if (!MOE || ntl > 1)
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm[hsel].v, fxm[1], acc[1], 0, 0, 0);
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1])); // (the operand registers are free again)
            read_mins(setc, std::integral_constant<int, 1>{});
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fxm[0]), "+v"(fxm[1]));
            if (!MOE || ntl > 2)
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm[hsel].v, fxm[0], acc[2], 0, 0, 0);
            if (!MOE || ntl > 3)
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm[hsel].v, fxm[1], acc[3], 0, 0, 0);
        };
        auto wait_frags = [&](half8_t(&f)[4], auto youngerc) {
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(decltype(youngerc)::value));
        };
        using Y0 = std::integral_constant<int, 0>;
        using Y4 = std::integral_constant<int, 4>;
        using Y6 = std::integral_constant<int, 6>;
        // waits: per period p a wave issues 7 (p even) or 8 (p odd) loads, in the order above
        //   stage p has landed when only the loads issued after DMA(p) are in flight: W(p + 1) (2, + header for odd p) and all of
        //   period p - 1 (7 + [p even]): 10 either way
        //   the header of super-block (p + 1) / 2 (p odd) has landed when only periods p - 1's and p's loads are in flight: 7 + 8

        // ---- prologue: the issue order of the steady state for periods -3, -2, -1 (p + 3 = 0, then DMA(0), W(1), DMA(1), W(2))
        // period -3 (odd: set 1 -> wset 0, header of super-block 0): W(0) + header
        mem_op(5, I1{});
        mem_op(6, I1{});
        mem_op(7, I1{});
        // period -2 (set 2): DMA(0) -> slot 0, W(1) -> set 1
#pragma unroll
        for (int j = 0; j < 7; j++)
            mem_op(j, I2{});
        // period -1 (set 3): DMA(1) -> slot 1, W(2) -> set 2 and the header of super-block 1
#pragma unroll
        for (int j = 0; j < 8; j++)
            mem_op(j, I3{});
        asm volatile("s_waitcnt vmcnt(15)" : "+v"(hd[0])::"memory"); // W(0) and its header have landed
        make_consts(I0{});

        if constexpr (GRP == 0) {
            auto period = [&](auto setc) {
                constexpr int set = decltype(setc)::value;
                using N1 = std::integral_constant<int, (set + 1) & 3>;
                asm volatile("s_waitcnt vmcnt(10)" : "+v"(qa[set]), "+v"(qb[set])::"memory");
                asm volatile("s_barrier" ::: "memory");
                constexpr bool MINS = (set & 1) == 1;
                read_frags(F[0], setc, 0);
                if constexpr (MINS)
                    read_mins(setc, I0{});
                mem_op(0, setc);
                mem_op(1, setc);
                half8_t wf = dq(setc, 0);
#pragma unroll
                for (int t8 = 0; t8 < 8; t8++) {
                    // younger than K-step t8's fragments: those of t8 + 1 and (K-step 0 of a second stage) the two mins fragments
                    if (t8 + 1 < 8)
                        read_frags(F[(t8 + 1) & 1], setc, t8 + 1);
                    if (t8 == 0 && MINS)
                        wait_frags(F[0], Y6{});
                    else if (t8 + 1 < 8)
                        wait_frags(F[t8 & 1], Y4{});
                    else
                        wait_frags(F[t8 & 1], Y0{});
                    half8_t wn = wf;
                    if (t8 + 1 < 8)
                        wn = dq(setc, t8 + 1);
                    mfma4(wf, F[t8 & 1]);
                    if constexpr (MINS)
                        if (t8 == 3)
                            mfma_mins(setc);
                    if (t8 < 6)
                        mem_op(t8 + 2, setc);
                    wf = wn;
                }
                if constexpr ((set & 1) == 1) // the next period starts a super-block: its header must have landed
                    asm volatile("s_waitcnt vmcnt(15)" : "+v"(hd[set == 1 ? 1 : 0])::"memory");
                make_consts(N1{});
            };
#ifdef KS_CHECK_NB
#pragma unroll
#endif
            for (int p = 0; p < np; p += 4) {
                period(I0{});
                period(I1{});
                if (p + 2 < np) {
                    period(I2{});
                    period(I3{});
                }
            }
        } else {
            // period-local K-steps j = 0..3: K-steps 4..7 of the stage in set - 1 (the fragments of 4 are in flight in F[0], the
            // weight fragment of 4 is wfc); j = 4..7: K-steps 0..3 of the stage in `set`
            auto first_half_of_new = [&](auto setc, half8_t wf) { // j = 4..7: K-steps 0..3; fetches K-steps 1..4
                constexpr bool MINS = (decltype(setc)::value & 1) == 1;
#pragma unroll
                for (int j = 4; j < 8; j++) {
                    read_frags(F[(j + 1) & 1], setc, j - 3); // (K-step 4 is consumed behind the next barrier)
                    if (MINS && j == 4)
                        read_mins(setc, I0{});
                    if (MINS && (j == 4 || j == 5))
                        wait_frags(F[j & 1], Y6{}); // (the mins fragments are younger than K-step 1's)
                    else
                        wait_frags(F[j & 1], Y4{});
                    half8_t wn = dq(setc, j - 3);
                    mfma4(wf, F[j & 1]);
                    if constexpr (MINS)
                        if (j == 7)
                            mfma_mins(setc);
                    mem_op(j, setc);
                    wf = wn;
                }
                wfc = wf;
            };
            auto period = [&](auto setc) {
                constexpr int set = decltype(setc)::value;
                using OS = std::integral_constant<int, (set + 3) & 3>;
                using N1 = std::integral_constant<int, (set + 1) & 3>;
                asm volatile("s_waitcnt vmcnt(10)" : "+v"(qa[set]), "+v"(qb[set])::"memory");
                asm volatile("s_barrier" ::: "memory");
                half8_t wf = wfc;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // one K-step ahead: the old stage's 5, 6, 7, then the new one's 0 (its data landed before the barrier)
                    if (j < 3)
                        read_frags(F[(j + 1) & 1], OS{}, j + 5);
                    else
                        read_frags(F[(j + 1) & 1], setc, 0);
                    wait_frags(F[j & 1], Y4{});
                    half8_t wn = j < 3 ? dq(OS{}, j + 5) : dq(setc, 0);
                    mfma4(wf, F[j & 1]);
                    mem_op(j, setc);
                    wf = wn;
                }
                first_half_of_new(setc, wf);
                if constexpr ((set & 1) == 1)
                    asm volatile("s_waitcnt vmcnt(15)" : "+v"(hd[set == 1 ? 1 : 0])::"memory");
                make_consts(N1{});
            };
            // period 0: nothing to finish; a cold start on stage 0's K-steps 0..3 (its four DMA pieces first)
            {
                asm volatile("s_waitcnt vmcnt(10)" : "+v"(qa[0]), "+v"(qb[0])::"memory");
                asm volatile("s_barrier" ::: "memory");
                read_frags(F[0], I0{}, 0);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    mem_op(j, I0{});
                first_half_of_new(I0{}, dq(I0{}, 0));
                make_consts(I1{});
            }
#ifdef KS_CHECK_NB
#pragma unroll
#endif
            for (int p = 1; p < np; p += 4) {
                period(I1{});
                if (p + 1 < np) {
                    period(I2{});
                    period(I3{});
                }
                if (p + 3 < np)
                    period(I0{});
            }
            // K-steps 4..7 of the last period's stage (np is even: set 1 or 3); nothing more to fetch
            auto last_half = [&](auto osc) {
                half8_t wf = wfc;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (j < 3) {
                        read_frags(F[(j + 1) & 1], osc, j + 5);
                        wait_frags(F[j & 1], Y4{});
                    } else {
                        wait_frags(F[j & 1], Y0{});
                    }
                    half8_t wn = wf;
                    if (j < 3)
                        wn = dq(osc, j + 5);
                    mfma4(wf, F[j & 1]);
                    wf = wn;
                }
            };
            if (((np - 1) & 3) == 1)
                last_half(I1{});
            else
                last_half(I3{});
        }
        // (the register sets stay allocated up to this wait — the clamped look-ahead loads of the last periods still write them —
        // and the wait sits INSIDE the branch: no register with a load in flight may reach the join)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]), "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3]),
                       "+v"(hd[0]), "+v"(hd[1])::"memory");
    }; // run
    if (wave < 4)
        run(std::integral_constant<int, 0>{});
    else
        run(std::integral_constant<int, 1>{});

    // ---- store: lane (i, h) holds token n0 + 32 nt + i, reg r = weight row 32 rt + (r&3) + 8 (r>>2) + 4 h
    if (active) {
        const bool vec = (ldc & 3) == 0 && (m & 3) == 0 && (((uintptr_t)C) & 15) == 0;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const long tok = n0 + nt * 32 + i;
            if (MOE ? nt * 32 + i >= moe_left : tok >= n)
                continue;
            const float ts = d8T[tok]; // 2^e of the token's normalised staging (pack.hip, prep_scaled_kernel): exact
            const long crow = MOE ? (long)mats.moe_slot_row[tok] : tok;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const long row0 = rt * 32 + 8 * g + 4 * h;
                float *dst = C + crow * ldc + row0;
#ifdef KR_BOUNDS
                {
                    bool bad_ = false;
                    const int dbg_period = -1;
                    if (row0 < m)
                        KR_CHECK(20, dst, 0, 0, 4, C, C + ((n - 1) * ldc + m));
                    if (bad_)
                        continue;
                }
#endif
                if (vec) {
                    if (row0 < m)
                        *(float4 *)dst = make_float4(acc[nt][4 * g] * ts, acc[nt][4 * g + 1] * ts, acc[nt][4 * g + 2] * ts, acc[nt][4 * g + 3] * ts);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (row0 + e < m)
                            dst[e] = acc[nt][4 * g + e] * ts;
                }
            }
        }
    }
}

// may the 256 x 128 body run this launch?  (LFAMD_GEMM_NO_KR: the loader-wave body's 128 x 128 tile instead — A/B runs)
bool lfamd_kr_ok(int Atype) {
    static const bool off = getenv("LFAMD_GEMM_NO_KR") != nullptr;
    return !off && Atype == LFAMD_TYPE_Q4_K;
}

// mats as the other launchers fill it (rb_end in 128-row blocks); n_ct = token tiles of 128
hipError_t lfamd_kr_go(int Atype, const gemm_mats &mats128, int nb, const void *Xh, const void *d8T, const void *Xm, long n, long n_pad,
                       int n_ct, hipStream_t s) {
    if (Atype != LFAMD_TYPE_Q4_K || nb < 1)
        return hipErrorInvalidValue;
    gemm_mats mats = mats128;
    int n_rb = 0;
    for (int j = 0; j < GEMM_MAX_MATS; j++) {
        if (j < mats.count)
            n_rb += (int)((mats.m[j] + 255) / 256);
        mats.rb_end[j] = n_rb;
    }
    gemm_kr_kernel<LFAMD_TYPE_Q4_K><<<(unsigned)(n_rb * n_ct), 512, 0, s>>>(mats, nb, (const _Float16 *)Xh, (const float *)d8T,
                                                                             (const _Float16 *)Xm, n, n_pad, n_rb, n_ct);
    return hipGetLastError();
}

// GGML_OP_MUL_MAT_ID batches on scaled operands: one launch over (token tile, expert, 256-row block); mats as
// lfamd_launch_gemm_wide_moe fills it.  (512 tokens x top-2 of 8 experts = 128 rows per expert: one 128-token tile each, and a
// dequantised fragment feeds four MFMAs instead of the loader-wave body's 128-row work-groups.)
hipError_t lfamd_kr_moe_go(int Atype, const gemm_mats &mats, int nb, const void *Xh, const void *d8T, const void *Xm, long n_pad, int experts,
                           int ct_max, hipStream_t s) {
    if (Atype != LFAMD_TYPE_Q4_K || nb < 1 || experts < 1 || ct_max < 1)
        return hipErrorInvalidValue;
    const int n_rb = (int)((mats.m[0] + 255) / 256);
    gemm_kr_kernel<LFAMD_TYPE_Q4_K, true><<<(unsigned)(experts * n_rb * ct_max), 512, 0, s>>>(mats, nb, (const _Float16 *)Xh, (const float *)d8T,
                                                                                              (const _Float16 *)Xm, n_pad, n_pad, n_rb, ct_max);
    return hipGetLastError();
}
