// gemm_lw.hip — prefill GEMM, "loader-wave" body for the resident Q4_K / Q5_K / Q6_K layouts.
//
// Two arithmetic forms of the reference's mul_mat_qX_K_q8_K_T (iqk_mul_mat.inc:601-643):
//   exact  (FAST = false; LFAMD_FLAG_PRECISE, MUL_MAT_ID on request): integer codes on v_mfma_f32_32x32x16_f16, f32 scales
//          once per super-block, as gemm_wide / gemm_mfma;
//   scaled (FAST = true; the default for every batch): f16(d * sc * q) x f16(d8 * code * 2^-e(token)), nothing per
//          super-block, output column times 2^e in the store (<= 1e-3 against the oracle).
// What changes against the 8-wave bodies is who does what:
//
//   * waves 0-3 COMPUTE: 32 weight rows x 128 tokens each (work-group tile 128 x 128).  One dequantised fragment (9-11
//     VALU instructions) feeds FOUR MFMAs; with the per-super-block scaling that is ~6 VALU per MFMA — the 32 x 64 wave
//     tile of gemm_wide needs ~10 and is vector-issue bound (PMC: VALU and MFMA cycles add up instead of overlapping).
//     Weights are read from LDS (no register double buffer), so the wave stays inside the 256 registers that two
//     waves per SIMD allow.
//   * waves 4-7 LOAD: every LDS-DMA piece of the work-group (activation codes, packed weights, headers, d8, mins
//     operand) is issued by them, two stages ahead, with counted vmcnt.  An LDS-DMA piece blocks its issuer for
//     100-200 cycles; here that time belongs to a wave with nothing else to do, on a SIMD whose other wave computes.
//   * stage = HALF a super-block (128 k): 32 KiB of activation codes + 8 KiB of nibbles (+ headers / d8 / mins operand
//     with the first half), three stages in LDS (157.5 KiB); loaders and compute waves synchronise through two counters
//     in LDS ("landed" / "released" stages), no work-group barrier inside the K loop.
#include "gemm_wide_impl.h"

#define LW_X 0          // 128 tokens x 256 B, XOR-swizzled 16-byte chunks
#define LW_W 32768      // 4 row tiles x 2 KiB (the two nibble groups of this half)
#define LW_HDR 40960    // 4 x 1 KiB: 512 B {d, dmin, scales[12]} per row, written twice (a DMA piece is a whole wave) (first half only)
#define LW_QH 45056     // Q5_K: 4 x 1 KiB of fifth bits                  (first half only)
#define LW_D8 49152     // 128 f32                                        (first half only)
#define LW_XM 49664     // 128 x 32 B mins operand                        (first half only)
#define LW_SLOT 53760
#define LW_STAGES 3
#define LW_FLAG (LW_STAGES * LW_SLOT) // landed-stage counter (+1 per loader wave and stage), at +4: released-stage counter (+1 per compute wave and stage)

#if GEMM_DIAG == 4 // development: s_memtime stamps around every barrier of work-group 0 (wave 0 computes, wave 4 loads)
__device__ unsigned long long g_lw_stamps[2 * 128];
extern "C" int lfamd_debug_lw_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lw_stamps), sizeof(g_lw_stamps));
}
#define LSTAMP(role)                                                                                             \
    do {                                                                                                         \
        if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && stamp_n < 128)                                     \
            g_lw_stamps[(role)*128 + stamp_n++] = __builtin_amdgcn_s_memtime();                                  \
    } while (0)
#else
#define LSTAMP(role)
#endif

// NT = token tiles of 32 per work-group: 4 (128 x 128 tile) or, scaled-operand body only, 2 (128 x 64: grids of 96 .. 191
// tiles of 128 x 128 — attn_output, ffn_down of an 8B model at 512 tokens — fill the 256 CUs with these instead)
// The work of work-group `bid` of `gdim` (the plain kernel passes its block index and grid size; the two-type kernel gives
// each type its own sub-grid); `lds` = the work-group's LW_STAGES * LW_SLOT + 16 bytes.
template <int TYPE, bool MOE, bool FAST, int NT>
__device__ __forceinline__ void gemm_lw_body(const gemm_mats &mats, int nb, const _Float16 *__restrict__ Xh,
                                             const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, long n, long n_pad,
                                             int n_rb, int n_ct, int ks, float *__restrict__ P, const int bid, const int gdim,
                                             uint8_t *lds) {
    static_assert(TYPE == LFAMD_TYPE_Q4_K || TYPE == LFAMD_TYPE_Q5_K || (TYPE == LFAMD_TYPE_Q6_K && FAST),
                  "resident K-quant layouts; Q6_K on the scaled-operand body only");
    static_assert(NT == 4 || (NT == 2 && FAST && !MOE), "the 64-token tile exists for the scaled-operand body only");
    constexpr int COLS = 32 * NT;
    constexpr bool Q5 = TYPE == LFAMD_TYPE_Q5_K, Q6 = TYPE == LFAMD_TYPE_Q6_K;
    constexpr int TILE = Q5 ? P5K_TILE : Q6 ? P6K_TILE : P4K_TILE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;

    // ---- tile of this work-group (same orders as gemm_wide)
    int rb, ct, moe_left = 0, kpart = 0;
    const uint8_t *__restrict__ A;
    float *__restrict__ C;
    long m, ldc, n0;
    if constexpr (MOE) {
        const int per_ct = gdim / mats.moe_ct_max;
        ct = bid / per_ct;
        const int rem = bid - ct * per_ct;
        const int e = rem / n_rb;
        rb = rem - e * n_rb;
        moe_left = mats.moe_cnt[e] - ct * WD_COLS;
        if (moe_left <= 0)
            return;
        A = mats.A[0] + (size_t)e * mats.expert_bytes;
        C = mats.C[0];
        m = mats.m[0], ldc = mats.ldc[0];
        n0 = (long)mats.moe_poff[e] + (long)ct * WD_COLS;
    } else {
        // ks > 1 (scaled-operand body, few-token batches of one matrix): K is cut into ks parts, every (tile, part) is a
        // work-group that writes its partial tile to P[part][token][row]; lw_ksplit_reduce sums the parts in order
        const int n_tiles = n_rb * n_ct, n_wg = n_tiles * ks;
        const int id = bid, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = id & 7;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
        kpart = L / n_tiles;
        tile_of(L - kpart * n_tiles, n_rb, n_ct, rb, ct);
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
        n0 = (long)ct * COLS;
    }
    const long n_row_tiles = (m + 31) / 32;
    const int nbs = (nb + ks - 1) / ks, b_base = kpart * nbs; // this work-group's super-blocks [b_base, b_base + H / 2)
    const int H = 2 * (b_base + nbs <= nb ? nbs : nb - b_base); // half super-blocks
#if GEMM_DIAG == 4
    int stamp_n = 0;
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), real0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t lds0 = lds_addr(lds);
    const uint32_t flag_addr = lds0 + LW_FLAG, rel_addr = flag_addr + 4;
    if (threadIdx.x == 0)
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:4\n\ts_waitcnt lgkmcnt(0)" ::"v"(flag_addr), "v"(0u) : "memory");
    // spin until the counter at `addr` reaches `need` (uniform; the slow path sleeps between polls)
    auto wait_counter = [&](uint32_t addr, uint32_t need) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        while (__builtin_amdgcn_readfirstlane(v) < need) {
            __builtin_amdgcn_s_sleep(2);
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        }
    };
    // a compute wave is done with a stage: every LDS read of it has returned (reads and this add execute in order)
    auto release_stage = [&]() {
        if (lane == 0)
            asm volatile("ds_add_u32 %0, %1" ::"v"(rel_addr), "v"(1u) : "memory");
    };
    asm volatile("s_barrier" ::: "memory");

    if (wave >= 4) {
        // =================================== loader waves ===================================
        const int lw = wave - 4;
        const long rtl = (long)rb * 4 + lw; // the row tile whose weights this wave copies
        const uint8_t *wt0 = A + (size_t)(rtl < n_row_tiles ? rtl : 0) * nb * TILE;
        // activation pieces: piece p = 8 lw + e holds token rows 4p .. 4p+3; lane = (row 4p + (lane >> 4), slot lane & 15)
        uint32_t xo[2 * NT];
#pragma unroll
        for (int e = 0; e < 2 * NT; e++) {
            const int row = 4 * (2 * NT * lw + e) + (lane >> 4);
            xo[e] = (uint32_t)(row * 512 + (((lane & 15) ^ (row & 15)) * 16));
        }
        const uint32_t xmo = (uint32_t)((32 * lw + (lane >> 1)) * 32 + (lane & 1) * 16);
        const uint8_t *xbase = (const uint8_t *)Xh + (size_t)n0 * 512;
        const uint8_t *xmbase = (const uint8_t *)Xm + (size_t)n0 * 32;

        // all LDS-DMA of stage hb (clamped at the end: rewrites a dead slot), in two parts: A = the first four activation
        // pieces, B = the rest
        auto issue = [&](int hb, auto halfc, auto partc) {
            constexpr int half = decltype(halfc)::value, part = decltype(partc)::value;
            const int hbc = hb < H ? hb : H - 2 + half;
            const int b = b_base + (hbc >> 1);
            const uint32_t slot = lds0 + (uint32_t)(hb % LW_STAGES) * LW_SLOT;
            const uint8_t *xs = uniform_ptr(xbase + (size_t)b * n_pad * 512 + half * 256);
#pragma unroll
            for (int e = NT * part; e < NT * part + NT; e++)
                glds1x16(xs, slot + LW_X + (2 * NT * lw + e) * 1024, xo[e]);
            if constexpr (part == 1) {
                const uint8_t *tile = uniform_ptr(wt0 + (size_t)b * TILE);
                const uint8_t *wg0 = uniform_ptr(tile + (2 * half) * 1024), *wg1 = uniform_ptr(tile + (2 * half + 1) * 1024);
                glds1x16(wg0, slot + LW_W + lw * 2048, (uint32_t)(lane * 16));
                glds1x16(wg1, slot + LW_W + lw * 2048 + 1024, (uint32_t)(lane * 16));
                if constexpr (Q6) { // upper two bits of this half's codes; with the first half the 16 int8 scales and the f16 d
                    glds1x16(uniform_ptr(tile + P6K_QH + half * 1024), slot + LW_QH + lw * 1024, (uint32_t)(lane * 16));
                    if constexpr (half == 0) {
                        glds1x16(uniform_ptr(tile + P6K_SC), slot + LW_HDR + lw * 1024, (uint32_t)((lane & 31) * 16));
                        glds1x4(uniform_ptr(tile + P6K_D), slot + LW_XM + lw * 256, (uint32_t)((lane & 15) * 4)); // 64 B, four times
                    }
                } else if constexpr (half == 0) {
                    // 512-byte header: the upper half-wave copies the same rows again into the second half of the 1 KiB slot
                    glds1x16(uniform_ptr(tile + P4K_HDR), slot + LW_HDR + lw * 1024, (uint32_t)((lane & 31) * 16));
                    if constexpr (Q5)
                        glds1x16(uniform_ptr(tile + P5K_QH), slot + LW_QH + lw * 1024, (uint32_t)(lane * 16));
                    if constexpr (!FAST)
                        glds1x4(uniform_ptr(d8T + (size_t)b * n_pad + n0), slot + LW_D8 + (lw & 1) * 256, (uint32_t)((lw & 1) * 256 + lane * 4));
                    if (lw < NT) // 32 tokens x 32 B per wave
                        glds1x16(uniform_ptr(xmbase + (size_t)b * n_pad * 32), slot + LW_XM + lw * 1024, xmo);
                }
            }
        };
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        // Protocol.  Barrier B_s closes half-step s: every compute wave is done with stage s, so its slot may be refilled
        // (stage s+3).  That a stage has LANDED is published apart from the barrier, by a counter in LDS (+1 per loader
        // wave and stage): after B_(s-1) the loader issues part A of stage s+2, waits until only those four pieces are in
        // flight (stage s+1, issued a half-step ago, is then complete), bumps the counter and issues part B.  The FAST
        // compute waves poll the counter late in half-step s and fetch their first operands of stage s+1 BEFORE B_s
        // (no LDS round trip without MFMAs in flight at the start of a half-step).
        auto landed = [&]() {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT) : "memory"); // only part A of the newest stage in flight
            if (lane == 0)
                asm volatile("ds_add_u32 %0, %1" ::"v"(flag_addr), "v"(1u) : "memory");
        };
        issue(0, H0{}, H0{});
        issue(0, H0{}, H1{});
        issue(1, H1{}, H0{});
        issue(1, H1{}, H1{});
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NT + 2 + (Q6 ? 1 : 0)) : "memory"); // stage 0 landed (only stage 1's pieces in flight)
#ifdef LW_EXP_NODMA // development: timing without the steady-state DMA (results are garbage)
#define LW_DMA_IF if (hb == 0)
#else
#define LW_DMA_IF
#endif
        for (int hb = 0; hb < H; hb += 2) {
            LSTAMP(1);
            wait_counter(rel_addr, 4u * (uint32_t)hb); // the slot of stage hb+2 was stage hb-1's
            LW_DMA_IF issue(hb + 2, H0{}, H0{});
            landed(); // stage hb+1
            LSTAMP(1);
            LW_DMA_IF issue(hb + 2, H0{}, H1{});
            LSTAMP(1);
            wait_counter(rel_addr, 4u * (uint32_t)(hb + 1)); // the slot of stage hb+3 was stage hb's
            LSTAMP(1);
            LW_DMA_IF issue(hb + 3, H1{}, H0{});
            landed(); // stage hb+2
            LSTAMP(1);
            LW_DMA_IF issue(hb + 3, H1{}, H1{});
            LSTAMP(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing may land after the work-group has left
        return;
    }

    // =================================== compute waves ===================================
    const int rw = wave;
    const long rt = (long)rb * 4 + rw;
    const bool active = rt < n_row_tiles;
    float16_t_ acc[NT], tmp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[nt][r] = 0.0f, tmp[nt][r] = 0.0f;
    // fragment chunk of K-step t8 (0..7 inside the half): row i, logical chunk 2 t8 + h, slot = chunk ^ (i & 15)
    uint32_t xoff[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
        xoff[u] = (uint32_t)(i * 256 + (((2 * u + h) ^ (i & 15)) * 16)) + LW_X;
    const float16_t_ zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t magic = opaque_magic();
    u32x4 hd = {0, 0, 0, 0}, hq = {0, 0, 0, 0};

    auto half_step = [&](uint32_t slot, uint32_t slot_first, auto halfc) {
        constexpr int half = decltype(halfc)::value;
        // this half's 8 K-step dwords (two groups) and, with the first half, the row header
        u32x4 qa, qb;
        if constexpr (half == 0) {
            asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%4+1024\n\tds_read_b128 %2, %5 offset:%6\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(qa), "=&v"(qb), "=&v"(hd)
                         : "v"(slot + rw * 2048 + lane * 16), "n"(LW_W), "v"(slot + rw * 1024 + i * 16), "n"(LW_HDR));
            if constexpr (Q5)
                asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(hq) : "v"(slot + rw * 1024 + lane * 16), "n"(LW_QH));
        } else {
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%3+1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(qa), "=&v"(qb)
                         : "v"(slot + rw * 2048 + lane * 16), "n"(LW_W));
        }
        const uint32_t qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        const uint32_t hq5[4] = {hq.x, hq.y, hq.z, hq.w};
        (void)hq5;
        uint32_t sc03, sc47, mn03, mn47;
        q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
        const uint32_t scw = half ? sc47 : sc03;

        auto read_frags = [&](half8_t(&f)[4], int t8) {
            const uint32_t a = xoff[t8] + slot;
            dsr16<0>(f[0], a);
            dsr16<8192>(f[1], a);
            dsr16<16384>(f[2], a);
            dsr16<24576>(f[3], a);
        };
        // K loop, software-pipelined by one K-step: the fragment of K-step t8+1 is dequantised BETWEEN the four MFMAs of
        // K-step t8 (a wave blocks on MFMA issue while the pipe is busy; VALU work issued right after an MFMA runs in
        // its shadow, up to ~24 cycles per MFMA — four MFMAs back to back followed by eleven VALU hide only the last
        // gap).  sched_group_barrier pins the interleave: 1 MFMA, 3 VALU, four times.
        auto consts = [&](int jj, half2_t &S, half2_t &O, half2_t &S16, half2_t &O16) {
            const q4_consts2 cp = q4_consts_pair(scw, (jj & 2) ? 2 : 0);
            const int hsel = jj & 1;
            S = half2_t{cp.S[hsel], cp.S[hsel]}, O = half2_t{cp.O[hsel], cp.O[hsel]};
            S16 = half2_t{cp.S16[hsel], cp.S16[hsel]}, O16 = half2_t{cp.O16[hsel], cp.O16[hsel]};
        };
        auto dequant = [&](int t8) -> half8_t {
            half2_t S, O, S16, O16;
            consts(t8 >> 1, S, O, S16, O16);
            if constexpr (Q5) // K-step 8 half + t8: group (8 half + t8) >> 2, position & 3
                return dequant_q5(qw[t8], hq5[2 * half + (t8 >> 2)] >> (t8 & 3), S, O, S16, O16, magic);
            else
                return dequant_q4(qw[t8], S, O, S16, O16, magic);
        };
        half8_t F[3][4]; // fragments two K-steps ahead: an LDS read takes ~200 cycles here, a K-step 128
        // epilogue operands (mins fragment + 16 token scales) of token tile nt, two tiles in flight
        half8_t exm[2];
        float4_t_ ed8[2][4];
        auto read_epi = [&](int nt) {
            asm volatile("ds_read_b128 %0, %5 offset:%6\n\tds_read_b128 %1, %7 offset:%8\n\tds_read_b128 %2, %7 offset:%8+32\n\t"
                         "ds_read_b128 %3, %7 offset:%8+64\n\tds_read_b128 %4, %7 offset:%8+96"
                         : "=&v"(exm[nt & 1]), "=&v"(ed8[nt & 1][0]), "=&v"(ed8[nt & 1][1]), "=&v"(ed8[nt & 1][2]), "=&v"(ed8[nt & 1][3])
                         : "v"(slot_first + (nt * 32 + i) * 32 + h * 16), "n"(LW_XM), "v"(slot_first + (nt * 32 + 4 * h) * 4), "n"(LW_D8));
        };
        read_frags(F[0], 0);
        read_frags(F[1], 1);
        half8_t wf = dequant(0);
#pragma unroll
        for (int t8 = 0; t8 < 8; t8++) {
#define LW_WAITF(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F[t8 % 3][0]), "+v"(F[t8 % 3][1]), "+v"(F[t8 % 3][2]), "+v"(F[t8 % 3][3]))
            if (t8 + 2 < 8) {
                read_frags(F[(t8 + 2) % 3], t8 + 2);
                LW_WAITF(8);
            } else if (t8 + 1 < 8) {
                if constexpr (half == 1) {
                    read_epi(0);
                    LW_WAITF(9);
                } else {
                    LW_WAITF(4);
                }
            } else {
                if constexpr (half == 1) {
                    read_epi(1);
                    LW_WAITF(10);
                } else {
                    LW_WAITF(0);
                }
            }
#undef LW_WAITF
            half8_t wn = wf;
            if (t8 + 1 < 8)
                wn = dequant(t8 + 1);
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
                tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t8 % 3][nt], wf, (half == 0 && t8 == 0) ? zero16 : tmp[nt], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); // three VALU in its shadow
            }
            wf = wn;
        }
        if constexpr (half == 1) {
            // ---- super-block done: mins (one MFMA per token tile) and acc += d8 * (d * tmp - dmin * tm)
            const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
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
            for (int nt = 0; nt < 4; nt++) {
                const int b = nt & 1;
                if (nt < 3)
                    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(exm[b]), "+v"(ed8[b][0]), "+v"(ed8[b][1]), "+v"(ed8[b][2]), "+v"(ed8[b][3]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(exm[b]), "+v"(ed8[b][0]), "+v"(ed8[b][1]), "+v"(ed8[b][2]), "+v"(ed8[b][3]));
                const float16_t_ tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(exm[b], wm.v, zero16, 0, 0, 0);
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        const float u = fmaf(-dmin, tm[r], d * tmp[nt][r]);
                        acc[nt][r] = fmaf(u, ed8[b][r4][e], acc[nt][r]);
                    }
                if (nt + 2 < 4) {
                    asm volatile("" : "+v"(acc[nt])); // the tile's scaling is done before its operand registers are reloaded
                    read_epi(nt + 2);
                }
            }
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])); // keep the scaling here (cf. gemm_wide LEGACY)
        }
    };

    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    asm volatile("s_barrier" ::: "memory"); // stage 0 landed (loaders waited before their arrival)
    if constexpr (FAST) {
        // ---- scaled operands (pack.hip prep mode 2): Xh = f16(d8 * code), the weight fragment = f16(d * sc * q), the
        // mins one more MFMA per token tile with f16(d8 * S_j) x f16(-dmin * m_j): everything accumulates straight into
        // acc, nothing is scaled per super-block.  The K-step pipeline runs ACROSS the half-step barriers: the first
        // operands of the next stage are fetched during K-steps 6 and 7 of this one (see the loader's protocol).
        u32x4 qa, qb;
        frag_u wm; // mins weights of the current super-block: f16(-dmin * m_j), zero in the upper K half
        wm.v = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        half8_t F[4][NT], fxm[NT];
        uint32_t fl = 0;
        auto read_frags = [&](half8_t(&f)[NT], uint32_t slot, int t8) {
            const uint32_t a = xoff[t8] + slot;
            dsr16<0>(f[0], a);
            dsr16<8192>(f[1], a);
            if constexpr (NT == 4) {
                dsr16<16384>(f[2], a);
                dsr16<24576>(f[3], a);
            }
        };
        auto pin = [&](half8_t(&f)[NT]) { // the registers of a fragment group are defined from here on (after a counted wait)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
                asm volatile("" : "+v"(f[nt]));
        };
        // weight fragment of K-step t8 of a half from the nibble words (wa, wb), the row header hdr and the fifth bits hqv
        // (Q6_K: hdr = the row's 16 int8 scales, hqv = the upper code bits of THIS half, dbits = its f16 d)
        auto dq_of = [&](auto halfc, int t8, const u32x4 &wa, const u32x4 &wb, const u32x4 &hdr, const u32x4 &hqv,
                         uint32_t dbits) -> half8_t {
            constexpr int half = decltype(halfc)::value;
            if constexpr (Q6) { // f16(d * sc) * (code - 32): the K-step is one 16-wide sub-block (cf. gemm_wide_impl.h)
                const uint32_t qw[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                const uint32_t scw4[4] = {hdr.x, hdr.y, hdr.z, hdr.w}, hw[4] = {hqv.x, hqv.y, hqv.z, hqv.w};
                const int tt = 8 * half + t8;
                const float scf = (float)(int)(int8_t)((scw4[tt >> 2] >> (8 * (tt & 3))) & 0xff);
                const half2_t S = bcast_h2(scf * h2f((uint16_t)(dbits & 0xffff)));
                uint32_t H = hw[t8 >> 1];
                if (t8 & 1)
                    H >>= 2;
                return dequant_q6(qw[t8], H, S);
            }
            uint32_t sc03, sc47, mn03, mn47;
            q4k_scales_bytes(hdr.y, hdr.z, hdr.w, sc03, sc47, mn03, mn47);
            const uint32_t scw = half ? sc47 : sc03;
            const half2_t dh2 = as_half2(__builtin_amdgcn_perm(hdr.x, hdr.x, 0x01000100u));
            const q4_consts2 cp = q4_consts_pair_scaled(scw, (t8 & 4) ? 2 : 0, dh2);
            const int hsel = (t8 >> 1) & 1;
            const half2_t S = {cp.S[hsel], cp.S[hsel]}, O = {cp.O[hsel], cp.O[hsel]};
            const half2_t S16 = {cp.S16[hsel], cp.S16[hsel]}, O16 = {cp.O16[hsel], cp.O16[hsel]};
            const uint32_t qw[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            if constexpr (Q5) {
                const uint32_t hq5[4] = {hqv.x, hqv.y, hqv.z, hqv.w};
                return dequant_q5(qw[t8], hq5[2 * half + (t8 >> 2)] >> (t8 & 3), S, O, S16, O16, magic);
            } else {
                return dequant_q4(qw[t8], S, O, S16, O16, magic);
            }
        };
        uint32_t dw6 = 0; // Q6_K: the row's f16 d
        auto dq = [&](auto halfc, int t8) -> half8_t { return dq_of(halfc, t8, qa, qb, hd, hq, dw6); }; // from the CURRENT operands
        const uint32_t d6_off = (uint32_t)(rw * 256 + i * 2) + LW_XM;
        const uint32_t wq_off = (uint32_t)(rw * 2048 + lane * 16) + LW_W, hd_off = (uint32_t)(rw * 1024 + i * 16) + LW_HDR;
        const uint32_t hq_off = (uint32_t)(rw * 1024 + lane * 16) + LW_QH, xm_off = (uint32_t)(i * 32 + h * 16) + LW_XM;
        half8_t wf;

        auto fast_half = [&](uint32_t slot, uint32_t slot_next, uint32_t slot_first, uint32_t need, auto halfc) {
            constexpr int half = decltype(halfc)::value;
            using HN = std::integral_constant<int, 1 - half>;
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) {
                // counted waits: lgkmcnt(N) = every LDS read older than the N youngest has returned
                constexpr int M = (half == 0 && !Q6) ? NT : 0; // mins fragments of this super-block, read in K-step 2 of its first half
                if (t8 < 6) {
                    if (t8 == 4)
                        asm volatile("ds_read_b32 %0, %1" : "=v"(fl) : "v"(flag_addr));
                    read_frags(F[(t8 + 2) & 3], slot, t8 + 2);
                    if (t8 == 2 && M > 0) { // the mins operand came with this (first-half) stage: landed
                        if constexpr (NT == 4)
                            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                                         "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                                         : "=&v"(fxm[0]), "=&v"(fxm[1]), "=&v"(fxm[NT - 2]), "=&v"(fxm[NT - 1])
                                         : "v"(slot + xm_off));
                        else
                            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024"
                                         : "=&v"(fxm[0]), "=&v"(fxm[1])
                                         : "v"(slot + xm_off));
                    }
                    // younger than this K-step's fragments: the next two fragment groups, the mins fragments from K-step 2
                    // until they are older than the group waited for (K-step 5), the counter read of K-step 4
                    if (t8 < 2)
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * NT));
                    else if (t8 < 4)
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * NT + M));
                    else if (t8 == 4)
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * NT + M + 1));
                    else
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * NT + 1));
                    pin(F[t8 & 3]);
                    if (t8 == 5 && M > 0)
                        pin(fxm); // (older than K-step 5's fragments)
                } else if (t8 == 6) {
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fl) : "n"(NT));
                    pin(F[2]);
                    uint32_t fv = __builtin_amdgcn_readfirstlane(fl);
                    while (fv < need) { // the next stage has not landed yet (rare: the loaders run a stage ahead)
                        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fl) : "v"(flag_addr) : "memory");
                        fv = __builtin_amdgcn_readfirstlane(fl);
                    }
                    read_frags(F[0], slot_next, 0);
                } else {
                    // younger than K-step 7's fragments: [mins NT] + qa/qb 2 + [header 1 (+ fifth bits 1)] + fragments NT
                    // K-step 6's look-ahead reads are older than the next stage's two fragment groups: wait for them here
                    // too, so that the next half's first weight fragment is built under this K-step's MFMAs instead of
                    // between two half-steps
                    read_frags(F[1], slot_next, 1);
                    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(qa), "+v"(qb), "+v"(hd), "+v"(hq), "+v"(dw6) : "n"(NT));
                    pin(F[3]);
                }
                if (t8 == 3 && M > 0) { // mins weights of this super-block: f16(-dmin * m_j), zero in the upper K half
                    uint32_t sc03, sc47, mn03, mn47;
                    q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
                    const float ndmin = -h2f((uint16_t)(hd.x >> 16));
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const uint32_t mw = p < 2 ? mn03 : mn47;
                        const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
                        half2_t v = {(_Float16)(h ? 0.0f : m0 * ndmin), (_Float16)(h ? 0.0f : m1 * ndmin)};
                        wm.p[p] = v;
                    }
                }
                half8_t wn = wf;
                if (t8 + 1 < 8)
                    wn = dq(halfc, t8 + 1);
                else
                    wn = dq_of(HN{}, 0, qa, qb, hd, hq, dw6); // the next stage's operands (K-step 6's reads, waited for above)
                if (t8 == 6) {
                    // The last fragment of this half (K-step 7) is built: qa / qb — and, at the end of a super-block, its
                    // header words — are dead, so the next stage's are read STRAIGHT INTO THE SAME REGISTERS.  (A second set
                    // of variables copied over after the wait let hipcc place that register copy BEFORE the wait: an
                    // intermittent read of data still in flight.)  The mins weights are taken from the header first.
                    asm volatile("" : "+v"(wn)); // K-step 7's fragment is complete before its source registers are reloaded
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(qa), "=&v"(qb) : "v"(slot_next + wq_off));
                    if constexpr (Q6) // the upper code bits come with every half
                        asm volatile("ds_read_b128 %0, %1" : "=v"(hq) : "v"(slot_next + hq_off));
                    if constexpr (half == 1) {
                        asm volatile("ds_read_b128 %0, %1" : "=v"(hd) : "v"(slot_next + hd_off));
                        if constexpr (Q5)
                            asm volatile("ds_read_b128 %0, %1" : "=v"(hq) : "v"(slot_next + hq_off));
                        if constexpr (Q6)
                            asm volatile("ds_read_u16 %0, %1" : "=v"(dw6) : "v"(slot_next + d6_off));
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; nt++) // weights are the A operand here: a lane ends up with 4 consecutive ROWS per token
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, F[t8 & 3][nt], acc[nt], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NT; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 12 / NT, 0); // its share of the next fragment's VALU
                }
                if (t8 == 5 && M > 0) { // the mins: NT more MFMAs among the first half's, straight into the accumulators
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm.v, fxm[nt], acc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                }
                wf = wn;
            }
        };

        // prologue: stage 0
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(qa), "=&v"(qb), "=&v"(hd)
                     : "v"(lds0 + wq_off), "v"(lds0 + hd_off));
        if constexpr (Q5 || Q6)
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(hq) : "v"(lds0 + hq_off));
        if constexpr (Q6)
            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dw6) : "v"(lds0 + d6_off));
        read_frags(F[0], lds0, 0);
        read_frags(F[1], lds0, 1);
        wf = dq(H0{}, 0);
        for (int hb = 0; hb < H; hb += 2) {
            const uint32_t s0 = lds0 + (uint32_t)(hb % LW_STAGES) * LW_SLOT, s1 = lds0 + (uint32_t)((hb + 1) % LW_STAGES) * LW_SLOT;
            const uint32_t s2 = lds0 + (uint32_t)((hb + 2) % LW_STAGES) * LW_SLOT;
            LSTAMP(0);
            fast_half(s0, s1, s0, 4u * (uint32_t)(hb + 1), H0{});
            LSTAMP(0);
            release_stage();
            LSTAMP(0);
            fast_half(s1, s2, s0, 4u * (uint32_t)(hb + 2), H1{});
            LSTAMP(0);
            release_stage();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the look-ahead reads of the (clamped) stage past the end
    } else {
    for (int hb = 0; hb < H; hb += 2) {
        const uint32_t s0 = lds0 + (uint32_t)(hb % LW_STAGES) * LW_SLOT, s1 = lds0 + (uint32_t)((hb + 1) % LW_STAGES) * LW_SLOT;
        LSTAMP(0);
        wait_counter(flag_addr, 4u * (uint32_t)hb); // stage hb landed (stage 0: the barrier above)
        half_step(s0, s0, H0{});
        LSTAMP(0);
        release_stage();
        wait_counter(flag_addr, 4u * (uint32_t)(hb + 1));
        LSTAMP(0);
        half_step(s1, s0, H1{});
        LSTAMP(0);
        release_stage();
    }
    }

#if GEMM_DIAG == 4 // shader clock against the 100 MHz reference over the whole K loop
    if (blockIdx.x == 0 && lane == 0 && wave == 0) {
        g_lw_stamps[126] = __builtin_amdgcn_s_memtime() - clk0;
        g_lw_stamps[127] = __builtin_amdgcn_s_memrealtime() - real0;
    }
#endif
    // ---- store.  FAST: lane (i, h) holds token n0 + 32nt + i, reg r = weight row 32rt + (r&3) + 8(r>>2) + 4h: four
    // consecutive rows per register quad -> 16 dwordx4 stores per wave instead of 64 dword stores (the store tail is
    // issue-bound: 6.1k cycles of a 62k-cycle work-group with dword stores)
    if constexpr (FAST) {
        if (active) {
            const bool part = !MOE && ks > 1; // partial tile: rows padded to the row-block grid, no token scale yet
            if (part)
                C = P + (size_t)kpart * (size_t)n_pad * (size_t)(n_rb * 128), ldc = (long)n_rb * 128, m = ldc;
            const bool vec = (ldc & 3) == 0 && (m & 3) == 0 && (((uintptr_t)C) & 15) == 0;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const long tok = n0 + nt * 32 + i;
                long crow = tok; // the row of C this token's outputs go to
                if constexpr (MOE) {
                    if (nt * 32 + i >= moe_left)
                        continue;
                    crow = (long)mats.moe_slot_row[tok];
                } else if (tok >= n) {
                    continue;
                }
                const float ts = part ? 1.0f : d8T[tok]; // 2^e of the token's normalised staging (pack.hip, prep_scaled_kernel): exact
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const long row0 = rt * 32 + 8 * g + 4 * h;
                    float *dst = C + crow * ldc + row0;
                    if (vec) {
                        if (row0 < m)
                            *(float4 *)dst = make_float4(acc[nt][4 * g] * ts, acc[nt][4 * g + 1] * ts, acc[nt][4 * g + 2] * ts,
                                                         acc[nt][4 * g + 3] * ts);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (row0 + e < m)
                                dst[e] = acc[nt][4 * g + e] * ts;
                    }
                }
            }
        }
    } else if (active) {
        const long row = rt * 32 + i;
        if (row < m) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int tl = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const long tok = n0 + tl;
                    if constexpr (MOE) {
                        if (tl < moe_left)
                            C[(long)mats.moe_slot_row[tok] * ldc + row] = acc[nt][r];
                    } else if (tok < n) {
                        C[tok * ldc + row] = acc[nt][r];
                    }
                }
        }
    }
#if GEMM_DIAG == 4 // cycles of this wave's C store, to the last write acknowledged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && lane == 0 && wave == 0)
        g_lw_stamps[125] = __builtin_amdgcn_s_memtime() - clk0 - g_lw_stamps[126];
#endif
}

template <int TYPE, bool MOE, bool FAST, int NT>
__global__ __launch_bounds__(512) void gemm_lw_kernel(const gemm_mats mats, int nb, const _Float16 *__restrict__ Xh,
                                                      const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, long n,
                                                      long n_pad, int n_rb, int n_ct, int ks, float *__restrict__ P) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[LW_STAGES * LW_SLOT + 16];
    gemm_lw_body<TYPE, MOE, FAST, NT>(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, P, (int)blockIdx.x, (int)gridDim.x, lds);
}

// Two weight types in ONE launch of the 128 x 128 scaled-operand body (attn_q/k in Q4_K or Q5_K with attn_v in Q6_K at
// prefill, cf. gemv_kq_dual_kernel): work-groups [0, grid_a) run type A's body over mats_a, the rest type B's over mats_b.
template <int TA, int TB>
__global__ __launch_bounds__(512) void gemm_lw_dual_kernel(const gemm_mats mats_a, const gemm_mats mats_b, int nb,
                                                           const _Float16 *__restrict__ Xh, const float *__restrict__ d8T,
                                                           const _Float16 *__restrict__ Xm, long n, long n_pad, int n_rb_a,
                                                           int n_rb_b, int n_ct, int grid_a) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[LW_STAGES * LW_SLOT + 16];
    if ((int)blockIdx.x < grid_a)
        gemm_lw_body<TA, false, true, 4>(mats_a, nb, Xh, d8T, Xm, n, n_pad, n_rb_a, n_ct, 1, nullptr, (int)blockIdx.x, grid_a, lds);
    else
        gemm_lw_body<TB, false, true, 4>(mats_b, nb, Xh, d8T, Xm, n, n_pad, n_rb_b, n_ct, 1, nullptr, (int)blockIdx.x - grid_a,
                                         (int)gridDim.x - grid_a, lds);
}

hipError_t lfamd_lw_dual_go(int type_a, const gemm_mats &ma, int n_rb_a, int type_b, const gemm_mats &mb, int n_rb_b, int nb,
                            const void *Xh, const void *d8T, const void *Xm, long n, long n_pad, int n_ct, hipStream_t s) {
    if (type_b != LFAMD_TYPE_Q6_K || (type_a != LFAMD_TYPE_Q4_K && type_a != LFAMD_TYPE_Q5_K))
        return hipErrorInvalidValue;
    const unsigned grid_a = (unsigned)(n_rb_a * n_ct), grid_b = (unsigned)(n_rb_b * n_ct);
    if (type_a == LFAMD_TYPE_Q4_K)
        gemm_lw_dual_kernel<LFAMD_TYPE_Q4_K, LFAMD_TYPE_Q6_K><<<grid_a + grid_b, 512, 0, s>>>(
            ma, mb, nb, (const _Float16 *)Xh, (const float *)d8T, (const _Float16 *)Xm, n, n_pad, n_rb_a, n_rb_b, n_ct, (int)grid_a);
    else
        gemm_lw_dual_kernel<LFAMD_TYPE_Q5_K, LFAMD_TYPE_Q6_K><<<grid_a + grid_b, 512, 0, s>>>(
            ma, mb, nb, (const _Float16 *)Xh, (const float *)d8T, (const _Float16 *)Xm, n, n_pad, n_rb_a, n_rb_b, n_ct, (int)grid_a);
    return hipGetLastError();
}

hipError_t lfamd_lw_go(int Atype, const gemm_mats &mats, int nb, const void *Xh, const void *d8T, const void *Xm, long n, long n_pad,
                       int n_rb, int n_ct, unsigned n_wg, int moe, int fast, int nt, int ks, float *P, hipStream_t s) {
    if ((nt != 4 && nt != 2) || (nt == 2 && (!fast || moe)) || ks < 1 || (ks > 1 && (!fast || moe || !P || mats.count != 1)))
        return hipErrorInvalidValue;
#define LW_GO(T, M, F, N)                                                                                              \
    gemm_lw_kernel<T, M, F, N><<<n_wg, 512, 0, s>>>(mats, nb, (const _Float16 *)Xh, (const float *)d8T, (const _Float16 *)Xm, n,  \
                                                    n_pad, n_rb, n_ct, ks, P)
#define LW_GO2(T)                                                                                                      \
    do {                                                                                                               \
        if (moe && fast)                                                                                               \
            LW_GO(T, true, true, 4);                                                                                   \
        else if (moe)                                                                                                  \
            LW_GO(T, true, false, 4);                                                                                  \
        else if (fast && nt == 2)                                                                                      \
            LW_GO(T, false, true, 2);                                                                                  \
        else if (fast)                                                                                                 \
            LW_GO(T, false, true, 4);                                                                                  \
        else                                                                                                           \
            LW_GO(T, false, false, 4);                                                                                 \
    } while (0)
    if (Atype == LFAMD_TYPE_Q4_K) {
        LW_GO2(LFAMD_TYPE_Q4_K);
    } else if (Atype == LFAMD_TYPE_Q5_K) {
        LW_GO2(LFAMD_TYPE_Q5_K);
    } else if (Atype == LFAMD_TYPE_Q6_K && fast) { // scaled-operand body only
        if (moe)
            LW_GO(LFAMD_TYPE_Q6_K, true, true, 4);
        else if (nt == 2)
            LW_GO(LFAMD_TYPE_Q6_K, false, true, 2);
        else
            LW_GO(LFAMD_TYPE_Q6_K, false, true, 4);
    } else {
        return hipErrorInvalidValue;
    }
#undef LW_GO2
#undef LW_GO
    return hipGetLastError();
}

// Sum of the K parts of a split launch (fixed order: deterministic) times the token's output scale -> C.
__global__ void lw_ksplit_reduce(const float *__restrict__ P, int ks, long n, long n_pad, long ldp, long m, float *__restrict__ C,
                                 long ldc, const float *__restrict__ tok_scale) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x; // (token, row quad)
    const long quads = ldp / 4, tok = q / quads, r0 = (q - tok * quads) * 4;
    if (tok >= n || r0 >= m)
        return;
    float4 a = *(const float4 *)(P + tok * ldp + r0);
    for (int kp = 1; kp < ks; kp++) {
        const float4 b = *(const float4 *)(P + ((size_t)kp * n_pad + tok) * ldp + r0);
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    const float ts = tok_scale[tok];
    const float v[4] = {a.x * ts, a.y * ts, a.z * ts, a.w * ts};
    for (int e = 0; e < 4; e++)
        if (r0 + e < m)
            C[tok * ldc + r0 + e] = v[e];
}

hipError_t lfamd_lw_ksplit_reduce(const float *P, int ks, long n, long n_pad, long ldp, long m, float *C, long ldc,
                                  const float *tok_scale, hipStream_t s) {
    const long threads = n * (ldp / 4);
    lw_ksplit_reduce<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(P, ks, n, n_pad, ldp, m, C, ldc, tok_scale);
    return hipGetLastError();
}
