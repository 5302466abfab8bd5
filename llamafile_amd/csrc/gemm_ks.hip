// gemm_ks.hip — prefill GEMM, "K-split waves" body for the resident Q4_K layout on scaled operands: 128 weight rows x 64
// tokens per work-group, the tile that fills 256 CUs on the shapes the 128 x 128 tile cannot (4096 x 4096 x 512).
//
// Arithmetic: the scaled-operand form of gemm_lw.hip (f16(d * sc * q) x f16(d8 * code * 2^-e(token)), mins as one more MFMA
// per token tile and super-block, output column times 2^e in the store; reference: mul_mat_qX_K_q8_K_T,
// iqk_mul_mat.inc:601-643; <= 1e-3 against the oracle).
//
// What bounds this tile is not the matrix pipe but what a CU can TAKE IN: 295 KB of packed weights + 512 KB of f16
// activations per work-group, and a CU ingests ~75 GB/s from L2 whatever the path (tools/ingest_probe.hip: LDS-DMA, loads
// to registers and any mix of the two all land at 19-20 TB/s chip-wide, 8.7 us for this shape) against 6.9-8.6 us of MFMA
// time.  The loader-wave body (gemm_lw.hip) leaves that ingest to four waves that stop at every counted wait, and its
// one compute wave per SIMD has nobody to cover its own stalls (27 us).  Here:
//   * all eight waves compute AND load.  Wave w = (row tile rw = w & 3, K half kh = w >> 2) owns 32 rows x 64 tokens over
//     HALF of every super-block's K-steps; the two halves meet once, through LDS, after the K loop.  Two computing waves
//     per SIMD: one wave's LDS latency, DMA issue and dequantisation VALU run under the other's MFMAs (a wave alone
//     issues one VALU per 4 cycles, two waves one per 2).
//   * weights never touch LDS: a wave's nibbles are private to it, so they go HBM -> VGPR (two 16-byte loads + the row
//     header per super-block), three super-blocks ahead (HBM latency), in four static register sets.
//   * activations: one stage = one super-block = 32 KiB of f16 operands ([K half][64 tokens][256 B], 16-byte chunks
//     XOR-swizzled by token on the SOURCE address) + 2 KiB mins operand, by LDS-DMA, five pieces per wave and stage, two
//     stages ahead in a ring of three; ONE s_barrier per super-block.
//   * every wait is counted: a wave's loads retire in issue order (W(b+1), DMA(b+1), W(b+2) = 11 younger than what stage b
//     needs), so `s_waitcnt vmcnt(11)` + the barrier is "stage b has landed for everybody".
#include "gemm_wide_impl.h"

#define KS_XHALF 16384
#define KS_XM 32768
#define KS_SLOT (32768 + 2048)
#define KS_STAGES 4
#define KS_COLS 64
#ifndef KS_PRIO
#define KS_PRIO 1
#endif
#define KS_VM_STAGE "11" // loads issued after everything stage b needs: W(b + 1), DMA(b + 1), W(b + 2)
#define KS_VM_HDR "16"   // ... after W(b + 1): DMA(b + 1), W(b + 2), DMA(b + 2), W(b + 3)

__device__ static inline half8_t as_h8(u32x4 v) {
    return __builtin_bit_cast(half8_t, v);
}

#if GEMM_DIAG == 6 // development: s_memtime stamps of work-group KS_STAMP_WG, waves 0 and 4 (tools/ks_stamps.py)
#define KS_STAMP_WG 100
__device__ unsigned long long g_ks_stamps[2 * 256];
extern "C" int lfamd_debug_ks_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ks_stamps), sizeof(g_ks_stamps));
}
#define KSTAMP()                                                                                                     \
    do {                                                                                                             \
        if (blockIdx.x == KS_STAMP_WG && lane == 0 && rw == 0 && stamp_n < 256)                                      \
            g_ks_stamps[kh * 256 + stamp_n++] = __builtin_amdgcn_s_memtime();                                        \
    } while (0)
#else
#define KSTAMP()
#endif

// lean issue forms: the addresses are SGPR values kept by scalar adds (no VALU-written SGPR in front of a vector-memory
// instruction, so no wait states), M0 is written and never restored (nothing else in this kernel reads it)
template <int IMM>
__device__ static inline void ks_dma16(const void *base, uint32_t lds_dst, uint32_t voff) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst), "n"(IMM) : "memory", "scc");
}
__device__ static inline void ks_dma4(const void *base, uint32_t lds_dst, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int IMM>
__device__ static inline void ks_ld16(u32x4 &dst, const void *base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int IMM>
__device__ static inline void ks_dsr16(half8_t &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

template <int TYPE>
__global__ __launch_bounds__(512) void gemm_ks_kernel(const gemm_mats mats, int nb, const _Float16 *__restrict__ Xh,
                                                      const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, long n,
                                                      long n_pad, int n_rb, int n_ct) {
    static_assert(TYPE == LFAMD_TYPE_Q4_K, "resident P4K layout");
#ifdef KS_CHECK_NB // tools/isa_hazards.py: a fixed trip count unrolls the stage loop into straight-line code in execution order
    nb = KS_CHECK_NB;
#endif
    __shared__ __attribute__((aligned(16))) uint8_t lds[KS_STAGES * KS_SLOT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int rw = wave & 3, kh = wave >> 2;

    // ---- tile of this work-group (the order of gemm_lw: XCD-aware super-tiles)
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
    const long n0 = (long)ct * KS_COLS;
    const long n_row_tiles = (m + 31) / 32;
    const long rt = (long)rb * 4 + rw;
    const bool active = rt < n_row_tiles;
    const uint32_t lds0 = lds_addr(lds);
#if GEMM_DIAG == 6
    int stamp_n = 0;
#endif
    KSTAMP();

    // ---- this wave's loads
    // weights: groups 2 kh, 2 kh + 1 of the row tile's super-block (K-steps 8 kh .. 8 kh + 7) and the 32 row headers
    const uint32_t wlo = (uint32_t)lane * 16, hlo = (uint32_t)i * 16 + (uint32_t)(P4K_HDR) - (uint32_t)kh * 2048;
    // activations: pieces 4 wave .. 4 wave + 3 of the stage's 32; piece p = (K half p >> 4, token rows 4 (p & 15) .. + 3),
    // lane = (row + (lane >> 4), 16-byte slot lane & 15) <- logical chunk slot ^ (row & 15)
    uint32_t xo[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int p = 4 * wave + e, half = p >> 4, row = 4 * (p & 15) + (lane >> 4);
        xo[e] = (uint32_t)(row * 512 + half * 256 + (((lane & 15) ^ (row & 15)) * 16));
    }
    const uint32_t xd0 = (uint32_t)((wave >> 2) * KS_XHALF + ((4 * wave) & 15) * 1024); // LDS offset of piece 4 wave (the next three follow)
    const uint32_t xmd = (uint32_t)(KS_XM + 256 * wave), xmo = (uint32_t)(256 * wave + lane * 4);
    float16_t_ acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[nt][r] = 0.0f;
    const uint32_t magic = opaque_magic();

    // Everything from the first load to the last wait exists TWICE, once per K half (the two halves run different schedules,
    // below): no register with a load in flight ever crosses a branch or a join, where hipcc reconciles two register
    // allocations with copies — copies of registers whose data has not landed (found by tools/isa_hazards.py).
    auto run = [&](auto khc) {
    constexpr int KHV = decltype(khc)::value;
    // source pointers of the NEXT stage to copy (acts, mins operand) / to load (weights), kept in SGPRs and advanced by scalar
    // adds (clamped at the last super-block: the look-ahead past the end re-reads it into dead slots / register sets)
    const size_t xstride = (size_t)n_pad * 512, xmstride = (size_t)n_pad * 32;
    const uint8_t *xs_n = uniform_ptr((const uint8_t *)Xh + (size_t)n0 * 512);
    const uint8_t *xm_n = uniform_ptr((const uint8_t *)Xm + (size_t)n0 * 32);
    const uint8_t *wt_n = uniform_ptr(A + (size_t)(active ? rt : 0) * nb * P4K_TILE + (size_t)kh * 2048);
    // five wait states between the VALU writes of those SGPRs (v_readfirstlane) and the first vector-memory instruction that
    // reads them: hipcc pads such hazards itself, but not in front of an asm statement (tools/isa_hazards.py checks the ISA)
    asm volatile("s_nop 4" ::"s"(xs_n), "s"(xm_n), "s"(wt_n));
    int bx_n = 0, bw_n = 0; // the super-blocks those pointers are at
    auto advance_x = [&]() {
        if (bx_n + 1 < nb) // (uniform)
            xs_n += xstride, xm_n += xmstride;
        bx_n++;
    };
    auto advance_w = [&]() {
        if (bw_n + 1 < nb)
            wt_n += P4K_TILE;
        bw_n++;
    };

    u32x4 qa[4], qb[4], hd[4]; // four register sets: super-block b lives in set b & 3
    // the eight loads a wave issues in period b (set = b & 3): j = 0..3 its pieces of DMA(b + 2), 4 its mins-operand piece
    // (into LDS slot (b + 2) & 3), 5..7 W(b + 3) (into register set (b + 3) & 3); the source pointers advance behind 4 and 7
    auto mem_op = [&](int j, auto setc) {
        constexpr int set = decltype(setc)::value;
        constexpr int wset = (set + 3) & 3;
        const uint32_t slot = lds0 + (uint32_t)(((set + 2) & 3) * KS_SLOT);
        if (j == 0)
            ks_dma16<0>(xs_n, slot + xd0, xo[0]);
        else if (j == 1)
            ks_dma16<1024>(xs_n, slot + xd0, xo[1]);
        else if (j == 2)
            ks_dma16<2048>(xs_n, slot + xd0, xo[2]);
        else if (j == 3)
            ks_dma16<3072>(xs_n, slot + xd0, xo[3]);
        else if (j == 4) {
            ks_dma4(xm_n, slot + xmd, xmo);
            advance_x();
        } else if (j == 5)
            ks_ld16<0>(qa[wset], wt_n, wlo);
        else if (j == 6)
            ks_ld16<1024>(qb[wset], wt_n, wlo);
        else {
            ks_ld16<0>(hd[wset], wt_n, hlo);
            advance_w();
        }
    };

    // ---- compute state
    // fragment chunk of K-step t8 (0..7 inside this wave's K half): token row i, logical chunk 2 t8 + h; two address sets:
    // slots 0 / 1 and slots 2 / 3 are reached from them through the instruction's offset field
    uint32_t xoffA[8], xoffB[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        xoffA[u] = lds0 + (uint32_t)(kh * KS_XHALF + i * 256 + (((2 * u + h) ^ (i & 15)) * 16));
        xoffB[u] = xoffA[u] + 2 * KS_SLOT;
    }
    const uint32_t xm_offA = lds0 + (uint32_t)(KS_XM + i * 32 + h * 16), xm_offB = xm_offA + 2 * KS_SLOT;
    half8_t F[4][2], fxm[2]; // fragment ring: K-step j of a period (0..7) uses F[j & 3], fetched two K-steps ahead
    half8_t wfc;             // (kh = 1) the weight fragment carried over a barrier

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // dequantisation constants of this wave's four sub-blocks of a super-block (from its row header) and, on the super-blocks
    // whose mins this wave adds ((b & 1) == kh), the mins weights: f16(-dmin * m_j) in the lower K half of the operand
    struct sb_consts {
        q4_consts2 cp0, cp1;
        frag_u wm;
    };
    sb_consts K2[2]; // super-block b uses K2[b & 1], computed at the end of period b - 1
    auto make_consts = [&](auto setc, auto khc) {
        constexpr int set = decltype(setc)::value, KH = decltype(khc)::value;
        sb_consts &o = K2[set & 1];
        uint32_t sc03, sc47, mn03, mn47;
        q4k_scales_bytes(hd[set].y, hd[set].z, hd[set].w, sc03, sc47, mn03, mn47);
        const uint32_t scw = KH ? sc47 : sc03;
        const half2_t dh2 = as_half2(__builtin_amdgcn_perm(hd[set].x, hd[set].x, 0x01000100u));
        o.cp0 = q4_consts_pair_scaled(scw, 0, dh2), o.cp1 = q4_consts_pair_scaled(scw, 2, dh2);
        if constexpr ((set & 1) == KH) {
            const float ndmin = -h2f((uint16_t)(hd[set].x >> 16));
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t mw = p < 2 ? mn03 : mn47;
                const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
                const half2_t v = {(_Float16)(h ? 0.0f : m0 * ndmin), (_Float16)(h ? 0.0f : m1 * ndmin)};
                o.wm.p[p] = v;
            }
        }
    };
    // fragment reads of K-step t8 of the super-block in slot / register set `set`
    auto read_frags = [&](half8_t(&f)[2], auto setc, int t8) {
        constexpr int set = decltype(setc)::value;
        constexpr int SOFF = (set & 1) * KS_SLOT; // offset field: slot `set` from address set A (slots 0, 1) or B (slots 2, 3)
        const uint32_t a = set < 2 ? xoffA[t8] : xoffB[t8];
        ks_dsr16<SOFF>(f[0], a);
        ks_dsr16<SOFF + 8192>(f[1], a);
    };
    auto read_mins = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
        constexpr int SOFF = (set & 1) * KS_SLOT;
        const uint32_t a = set < 2 ? xm_offA : xm_offB;
        ks_dsr16<SOFF>(fxm[0], a);
        ks_dsr16<SOFF + 1024>(fxm[1], a);
    };
    // the weight fragment of K-step t8 of the super-block in register set `set`
    auto dq = [&](auto setc, int t8) -> half8_t {
        constexpr int set = decltype(setc)::value;
        const sb_consts &kc = K2[set & 1];
        const uint32_t qw[8] = {qa[set].x, qa[set].y, qa[set].z, qa[set].w, qb[set].x, qb[set].y, qb[set].z, qb[set].w};
        const q4_consts2 &cp = (t8 & 4) ? kc.cp1 : kc.cp0;
        const int hsel = (t8 >> 1) & 1;
        const half2_t S = {cp.S[hsel], cp.S[hsel]}, O = {cp.O[hsel], cp.O[hsel]};
        const half2_t S16 = {cp.S16[hsel], cp.S16[hsel]}, O16 = {cp.O16[hsel], cp.O16[hsel]};
        return dequant_q4(qw[t8], S, O, S16, O16, magic);
    };
    auto mfma2 = [&](const half8_t &wf, half8_t(&f)[2]) {
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, f[nt], acc[nt], 0, 0, 0);
    };
    // wait until only the `younger` newest LDS reads are outstanding, and pin the fragment group's registers behind the wait
    auto wait_frags = [&](half8_t(&f)[2], auto youngerc) {
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(decltype(youngerc)::value));
    };
    using Y0 = std::integral_constant<int, 0>;
    using Y2 = std::integral_constant<int, 2>;
    using Y4 = std::integral_constant<int, 4>;
    using Y6 = std::integral_constant<int, 6>;

    // ---- prologue: the issue order of the steady state (period s issues DMA(s + 2), then W(s + 3)) for periods -3, -2, -1
#pragma unroll
    for (int j = 5; j < 8; j++)
        mem_op(j, I1{}); // W(0) -> register set 0
#pragma unroll
    for (int j = 0; j < 8; j++)
        mem_op(j, I2{}); // DMA(0) -> slot 0, W(1)
#pragma unroll
    for (int j = 0; j < 8; j++)
        mem_op(j, I3{}); // DMA(1) -> slot 1, W(2)
    asm volatile("s_waitcnt vmcnt(" KS_VM_HDR ")" : "+v"(hd[0])::"memory"); // W(0) has landed

    // One period = one super-block = eight K-steps for every wave, between two work-group barriers.  The two K halves run
    // HALF A PERIOD APART: the kh = 0 waves start their half of super-block b right behind barrier b (a cold start: first
    // fragment reads, then K-steps 0..7); the kh = 1 waves first finish K-steps 4..7 of super-block b - 1 — fragments fetched
    // before the barrier, the data landed a period ago — and start super-block b's K-steps 0..3 in the middle of the period,
    // carrying 4..7 over the next barrier.  The two waves of a SIMD therefore never stand at a cold start, a constants block
    // or a barrier together: one of them always has MFMAs to issue (with both halves in step 600-800 of a period's 2300
    // cycles had no MFMA in flight on either wave, tools/ks_stamps.py).  In both: a period's eight loads go out one per K-step.
    if constexpr (KHV == 0) {
        using KH = I0;
        make_consts(I0{}, KH{});
        auto period = [&](auto setc) {
            constexpr int set = decltype(setc)::value;
            constexpr bool MINS = (set & 1) == 0;
            using N1 = std::integral_constant<int, (set + 1) & 3>;
            // everything this wave loaded for stage b has landed (11 younger loads may be in flight); behind the barrier
            // everybody's pieces have, and everybody is done reading the slot that stage b + 2 goes to
            KSTAMP();
            asm volatile("s_waitcnt vmcnt(" KS_VM_STAGE ")" : "+v"(qa[set]), "+v"(qb[set])::"memory");
            KSTAMP();
            asm volatile("s_barrier" ::: "memory");
            KSTAMP();
            read_frags(F[0], setc, 0);
            read_frags(F[1], setc, 1);
            if constexpr (MINS)
                read_mins(setc);
            mem_op(0, setc);
            mem_op(1, setc);
            KSTAMP();
            half8_t wf = dq(setc, 0);
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) {
                // counted waits: younger than K-step t8's fragments are those of t8 + 1, t8 + 2 and (K-steps 0, 1) the mins'
                if (t8 + 2 < 8) {
                    read_frags(F[(t8 + 2) & 3], setc, t8 + 2);
                    if (MINS && t8 < 2)
                        wait_frags(F[t8 & 3], Y6{});
                    else
                        wait_frags(F[t8 & 3], Y4{});
                } else if (t8 + 1 < 8) {
                    wait_frags(F[t8 & 3], Y2{});
                } else {
                    wait_frags(F[t8 & 3], Y0{});
                }
                half8_t wn = wf;
                if (t8 + 1 < 8)
                    wn = dq(setc, t8 + 1);
                mfma2(wf, F[t8 & 3]);
                if constexpr (MINS)
                    if (t8 == 3) {
                        asm volatile("" : "+v"(fxm[0]), "+v"(fxm[1])); // (landed: older than K-step 2's fragments)
                        mfma2(K2[0].wm.v, fxm);
                    }
                if (t8 < 6)
                    mem_op(t8 + 2, setc);
                wf = wn;
            }
            // the next super-block's constants, under this one's last MFMAs: its header (W(b + 1)) has landed once only the 16
            // loads issued after it are in flight
            asm volatile("s_waitcnt vmcnt(" KS_VM_HDR ")" : "+v"(hd[(set + 1) & 3])::"memory");
            make_consts(N1{}, KH{});
        };
#ifdef KS_CHECK_NB
#pragma unroll
#endif
        for (int b = 0; b < nb; b += 4) {
            period(I0{});
            if (b + 1 < nb)
                period(I1{});
            if (b + 2 < nb)
                period(I2{});
            if (b + 3 < nb)
                period(I3{});
        }
        // (the register sets stay allocated up to this wait — the clamped look-ahead loads of the last periods still write them —
        // and the wait sits INSIDE the branch: no register with a load in flight may reach the join, where the two branches'
        // allocations are reconciled by copies)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]), "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3]),
                       "+v"(hd[0]), "+v"(hd[1]), "+v"(hd[2]), "+v"(hd[3])::"memory");
    } else {
        using KH = I1;
#ifndef KS_NO_SETPRIO
        // the second-dispatched half of a work-group loses every issue arbitration against its older SIMD partner (priority, then
        // age): with equal priorities the kh = 1 waves took 2080 cycles per period, the kh = 0 waves 1400 and then waited 700
        // at the barrier.  Static priority for the younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4).
        __builtin_amdgcn_s_setprio(KS_PRIO);
#endif
        make_consts(I0{}, KH{});
        // period-local K-steps j = 0..3: K-steps 4..7 of the super-block in set `oset` = set - 1 (the fragments of 4 and 5 are in
        // flight in F[0], F[1], the weight fragment of 4 is wfc); j = 4..7: K-steps 0..3 of the one in `set`
        auto first_half_of_new = [&](auto setc, half8_t wf) { // j = 4..7
            constexpr int set = decltype(setc)::value;
            constexpr bool MINS = (set & 1) == 1;
#pragma unroll
            for (int j = 4; j < 8; j++) {
                read_frags(F[(j + 2) & 3], setc, j - 2); // K-steps 2, 3, then 4, 5 (consumed behind the next barrier)
                if (MINS && j == 4)
                    read_mins(setc);
                if (MINS && j == 4)
                    wait_frags(F[j & 3], Y6{});
                else if (MINS && j == 5)
                    wait_frags(F[j & 3], Y6{});
                else
                    wait_frags(F[j & 3], Y4{});
                half8_t wn = dq(setc, j - 3);
                mfma2(wf, F[j & 3]);
                if constexpr (MINS)
                    if (j == 7) {
                        asm volatile("" : "+v"(fxm[0]), "+v"(fxm[1])); // (landed: older than the fragments waited for at j = 6)
                        mfma2(K2[1].wm.v, fxm);
                    }
                mem_op(j, setc);
                wf = wn;
            }
            wfc = wf;
        };
        auto period = [&](auto setc) {
            constexpr int set = decltype(setc)::value;
            using OS = std::integral_constant<int, (set + 3) & 3>;
            using N1 = std::integral_constant<int, (set + 1) & 3>;
            KSTAMP();
            asm volatile("s_waitcnt vmcnt(" KS_VM_STAGE ")" : "+v"(qa[set]), "+v"(qb[set])::"memory");
            KSTAMP();
            asm volatile("s_barrier" ::: "memory");
            KSTAMP();
            half8_t wf = wfc;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // two K-steps ahead: the old super-block's 6, 7, then the new one's 0, 1 (its data landed before the barrier)
                if (j < 2)
                    read_frags(F[(j + 2) & 3], OS{}, j + 6);
                else
                    read_frags(F[(j + 2) & 3], setc, j - 2);
                wait_frags(F[j & 3], Y4{});
                half8_t wn = j < 3 ? dq(OS{}, j + 5) : dq(setc, 0);
                mfma2(wf, F[j & 3]);
                mem_op(j, setc);
                wf = wn;
            }
            KSTAMP();
            first_half_of_new(setc, wf);
            asm volatile("s_waitcnt vmcnt(" KS_VM_HDR ")" : "+v"(hd[(set + 1) & 3])::"memory");
            make_consts(N1{}, KH{});
        };
        // period 0: nothing to finish; a cold start on super-block 0's K-steps 0..3 (its four DMA pieces first)
        {
            KSTAMP();
            asm volatile("s_waitcnt vmcnt(" KS_VM_STAGE ")" : "+v"(qa[0]), "+v"(qb[0])::"memory");
            KSTAMP();
            asm volatile("s_barrier" ::: "memory");
            KSTAMP();
            read_frags(F[0], I0{}, 0);
            read_frags(F[1], I0{}, 1);
#pragma unroll
            for (int j = 0; j < 4; j++)
                mem_op(j, I0{});
            KSTAMP();
            first_half_of_new(I0{}, dq(I0{}, 0));
            asm volatile("s_waitcnt vmcnt(" KS_VM_HDR ")" : "+v"(hd[1])::"memory");
            make_consts(I1{}, KH{});
        }
#ifdef KS_CHECK_NB
#pragma unroll
#endif
        for (int b = 1; b < nb; b += 4) {
            period(I1{});
            if (b + 1 < nb)
                period(I2{});
            if (b + 2 < nb)
                period(I3{});
            if (b + 3 < nb)
                period(I0{});
        }
        // K-steps 4..7 of the last super-block (register set / slot (nb - 1) & 3); nothing more to fetch
        auto last_half = [&](auto osc) {
            half8_t wf = wfc;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j < 2) {
                    read_frags(F[(j + 2) & 3], osc, j + 6);
                    wait_frags(F[j & 3], Y4{});
                } else if (j == 2) {
                    wait_frags(F[j & 3], Y2{});
                } else {
                    wait_frags(F[j & 3], Y0{});
                }
                half8_t wn = wf;
                if (j < 3)
                    wn = dq(osc, j + 5);
                mfma2(wf, F[j & 3]);
                wf = wn;
            }
        };
        switch ((nb - 1) & 3) { // (uniform)
        case 0: last_half(I0{}); break;
        case 1: last_half(I1{}); break;
        case 2: last_half(I2{}); break;
        default: last_half(I3{}); break;
        }
        // (the register sets stay allocated up to this wait — the clamped look-ahead loads of the last periods still write them —
        // and the wait sits INSIDE the branch: no register with a load in flight may reach the join, where the two branches'
        // allocations are reconciled by copies)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]), "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3]),
                       "+v"(hd[0]), "+v"(hd[1]), "+v"(hd[2]), "+v"(hd[3])::"memory");
    }
    }; // run
    if (kh == 0)
        run(std::integral_constant<int, 0>{});
    else
        run(std::integral_constant<int, 1>{});
    // nothing lands in LDS any more (the clamped look-ahead stages have), and behind the barrier everybody is done reading
    KSTAMP();
    asm volatile("s_barrier" ::: "memory");
    KSTAMP();

    // ---- the two K halves meet: wave (rw, kh) hands the partial of token tile 1 - kh to its partner (rw, 1 - kh) through
    // LDS and finishes token tile kh.  a + b is the same float either way round: deterministic.
    {
        const uint32_t mine = lds0 + (uint32_t)((rw * 2 + kh) * 4096 + lane * 16), theirs = lds0 + (uint32_t)((rw * 2 + (1 - kh)) * 4096 + lane * 16);
        const float16_t_ give = kh ? acc[0] : acc[1];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4_t_ v = {give[4 * g], give[4 * g + 1], give[4 * g + 2], give[4 * g + 3]};
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(mine), "v"(v), "n"(g * 1024) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float16_t_ fin = kh ? acc[1] : acc[0];
        float4_t_ got[4];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                     "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(got[0]), "=&v"(got[1]), "=&v"(got[2]), "=&v"(got[3])
                     : "v"(theirs)
                     : "memory");
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                fin[4 * g + e] += got[g][e];

        // ---- store token tile kh: lane (i, h) holds token n0 + 32 kh + i, reg r = weight row 32 rt + (r&3) + 8 (r>>2) + 4 h
        if (active) {
            const long tok = n0 + kh * 32 + i;
            if (tok < n) {
                const bool vec = (ldc & 3) == 0 && (m & 3) == 0 && (((uintptr_t)C) & 15) == 0;
                const float ts = d8T[tok]; // 2^e of the token's normalised staging (pack.hip, prep_scaled_kernel): exact
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const long row0 = rt * 32 + 8 * g + 4 * h;
                    float *dst = C + tok * ldc + row0;
                    if (vec) {
                        if (row0 < m)
                            *(float4 *)dst = make_float4(fin[4 * g] * ts, fin[4 * g + 1] * ts, fin[4 * g + 2] * ts, fin[4 * g + 3] * ts);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (row0 + e < m)
                                dst[e] = fin[4 * g + e] * ts;
                    }
                }
            }
        }
    }
#if GEMM_DIAG == 6
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KSTAMP();
#endif
}

// may the K-split-waves body run this launch?  (LFAMD_GEMM_NO_KS: the loader-wave body's 128 x 64 tile instead — A/B runs)
bool lfamd_ks_ok(int Atype) {
    static const bool off = getenv("LFAMD_GEMM_NO_KS") != nullptr;
    return !off && Atype == LFAMD_TYPE_Q4_K;
}

hipError_t lfamd_ks_go(int Atype, const gemm_mats &mats, int nb, const void *Xh, const void *d8T, const void *Xm, long n, long n_pad,
                       int n_rb, int n_ct, hipStream_t s) {
    if (Atype != LFAMD_TYPE_Q4_K || nb < 1)
        return hipErrorInvalidValue;
    gemm_ks_kernel<LFAMD_TYPE_Q4_K><<<(unsigned)(n_rb * n_ct), 512, 0, s>>>(mats, nb, (const _Float16 *)Xh, (const float *)d8T,
                                                                             (const _Float16 *)Xm, n, n_pad, n_rb, n_ct);
    return hipGetLastError();
}
