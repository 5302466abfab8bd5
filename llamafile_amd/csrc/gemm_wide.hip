// gemm_wide.hip — launchers of the 128 x 128 MFMA body (kernel: gemm_wide_impl.h; instantiations: gemm_wide_*.hip)
#include "gemm_wide_impl.h"
#include <stdlib.h>

hipError_t lfamd_wide_go_q4k(WIDE_ARGS);
hipError_t lfamd_wide_go_q5k(WIDE_ARGS);
hipError_t lfamd_wide_go_q6k(WIDE_ARGS);
hipError_t lfamd_wide_go_q40(WIDE_ARGS);
hipError_t lfamd_wide_go_q41(WIDE_ARGS);
hipError_t lfamd_wide_go_q50(WIDE_ARGS);
hipError_t lfamd_wide_go_q51(WIDE_ARGS);
hipError_t lfamd_wide_go_q2k(WIDE_ARGS);
hipError_t lfamd_wide_go_q3k(WIDE_ARGS);
hipError_t lfamd_wide_go_iq4xs(WIDE_ARGS);
hipError_t lfamd_wide_go_f16(WIDE_ARGS);
hipError_t lfamd_wide_go_bf16(WIDE_ARGS);

hipError_t lfamd_lw_go(int Atype, const gemm_mats &mats, int nb, const void *Xh, const void *d8T, const void *Xm, long n, long n_pad,
                       int n_rb, int n_ct, unsigned n_wg, int moe, int fast, int nt, int ks, float *P, hipStream_t s);
bool lfamd_ks_ok(int Atype);
bool lfamd_kr_ok(int Atype);
hipError_t lfamd_kr_go(int Atype, const gemm_mats &mats128, int nb, const void *Xh, const void *d8T, const void *Xm, long n, long n_pad,
                       int n_ct, hipStream_t s);
hipError_t lfamd_kr_moe_go(int Atype, const gemm_mats &mats, int nb, const void *Xh, const void *d8T, const void *Xm, long n_pad, int experts,
                           int ct_max, hipStream_t s);
hipError_t lfamd_ks_go(int Atype, const gemm_mats &mats, int nb, const void *Xh, const void *d8T, const void *Xm, long n, long n_pad,
                       int n_rb, int n_ct, hipStream_t s);
hipError_t lfamd_lw_ksplit_reduce(const float *P, int ks, long n, long n_pad, long ldp, long m, float *C, long ldc,
                                  const float *tok_scale, hipStream_t s);

// Few-token batches (n <= 128) of one matrix whose 128 x 64 tiles leave most CUs idle: K is cut into up to 8 parts, the
// partial tiles go to a workspace and are summed by a second kernel (4096 x 4096 x 32: 32 work-groups of 16 serial
// super-blocks become 256 of 2).
extern "C" size_t lfamd_gemm_lw_ksplit_bytes(long m, long n) {
    if (m <= 0 || n <= 8 || n > 128)
        return 0;
    const long n_rb = (m + 127) / 128, tiles2 = n_rb * ((n + 63) / 64);
    return tiles2 > 128 ? 0 : (size_t)8 * 128 * (size_t)n_rb * 128 * sizeof(float);
}
static int lw_ksplit(int tiles2, int nb) {
    int ks = 1;
    while (ks < 8 && tiles2 * ks * 2 <= 256 && nb / (ks * 2) >= 2)
        ks *= 2;
    return ks;
}

// Q4_K / Q5_K without K split run the loader-wave body (gemm_lw.hip) unless the caller asks for the plain one.
// `mode` (an argument of every launcher here) bit 0: plain body; bit 1: the activations were staged SCALED (pack.hip, prep mode 2) for the scaled-operand
// loader-wave body — only set when lfamd_gemm_wide_scaled_ok() said that body will run.
// 128 x 128 tiles from which the scaled-operand body uses them instead of 128 x 64: a 128 x 64 work-group takes ~0.73 of a
// 128 x 128 one (2720 vs 3680 cycles per super-block), so two rounds of the small tile (129 .. 256 big tiles = 258 .. 512
// small ones on 256 CUs) lose against one partial round of the big tile; up to 128 big tiles the small tile's single round wins.
#define LW_FULL_GRID 129
static bool lw_allowed(int mode) {
    static const bool env_plain = getenv("LFAMD_GEMM_NO_LW") != nullptr;
    return !(mode & 1) && !env_plain;
}

// The first rb_cut row blocks of `mats` / the rest (a matrix straddling the cut is split: weights by row tiles, C by rows).
static void split_mats(const gemm_mats &mats, int rb_cut, size_t row_tile_bytes, gemm_mats &lo, gemm_mats &hi) {
    lo = mats, hi = mats;
    lo.count = hi.count = 0;
    int lo_end = 0, hi_end = 0;
    for (int j = 0; j < mats.count; j++) {
        const int start = j ? mats.rb_end[j - 1] : 0, end = mats.rb_end[j];
        if (start < rb_cut) { // rows [0, (min(end, rb_cut) - start) * 128) of matrix j
            const int i = lo.count++;
            const int rbs = (end < rb_cut ? end : rb_cut) - start;
            lo.A[i] = mats.A[j], lo.C[i] = mats.C[j], lo.ldc[i] = mats.ldc[j];
            lo.m[i] = end <= rb_cut ? mats.m[j] : (long)rbs * 128;
            lo_end += rbs;
            lo.rb_end[i] = lo_end;
        }
        if (end > rb_cut) {
            const int i = hi.count++;
            const int skip = start < rb_cut ? rb_cut - start : 0; // row blocks of matrix j that went to `lo`
            hi.A[i] = mats.A[j] + (size_t)skip * 4 * row_tile_bytes;
            hi.C[i] = mats.C[j] + (size_t)skip * 128, hi.ldc[i] = mats.ldc[j];
            hi.m[i] = mats.m[j] - (long)skip * 128;
            hi_end += end - start - skip;
            hi.rb_end[i] = hi_end;
        }
    }
    for (int i = lo.count; i < GEMM_MAX_MATS; i++)
        lo.A[i] = lo.A[0], lo.C[i] = lo.C[0], lo.m[i] = 0, lo.ldc[i] = 0, lo.rb_end[i] = lo_end;
    for (int i = hi.count; i < GEMM_MAX_MATS; i++)
        hi.A[i] = hi.A[0], hi.C[i] = hi.C[0], hi.m[i] = 0, hi.ldc[i] = 0, hi.rb_end[i] = hi_end;
}

static hipError_t wide_go(int Atype, int mode, float *P, size_t P_bytes, WIDE_ARGS) {
    const int g_scaled = (mode >> 1) & 1;
    const bool q45 = Atype == LFAMD_TYPE_Q4_K || Atype == LFAMD_TYPE_Q5_K;
    if ((q45 || (Atype == LFAMD_TYPE_Q6_K && g_scaled)) && lw_allowed(mode)) {
        if (g_scaled && !moe) {
            // scaled operands: 128 x 128 tiles when they fill the chip, else 128 x 64 (twice the work-groups, no K split)
            static const int full_grid = getenv("LFAMD_LW_FULL_GRID") ? atoi(getenv("LFAMD_LW_FULL_GRID")) : LW_FULL_GRID; // (tuning)
            if (n_rb * n_ct >= full_grid && lfamd_kr_ok(Atype)) // 256 x 128 tiles on the row-split body (gemm_kr.hip)
                return lfamd_kr_go(Atype, mats, nb, Xh, d8T, Xm, n, n_pad, n_ct, s);
            if (n_rb * n_ct >= full_grid) {
                // full rounds of 128 x 128 tiles; a last round of at most 128 of them (half the CUs idle) runs as one round of
                // 128 x 64 tiles instead (0.73 of the time): ffn_gate + ffn_up at 512 tokens = 896 tiles = 3 rounds + 128
                static const bool no_tail = getenv("LFAMD_LW_NO_TAIL_SPLIT") != nullptr; // (tests compare the two)
                const int tiles = n_rb * n_ct, rem = tiles % 256;
                if (!no_tail && tiles > 256 && rem > 0 && rem <= 128 && rem % n_ct == 0) {
                    const int rb_cut = (tiles - rem) / n_ct;
                    const size_t tile_bytes = (size_t)nb * (Atype == LFAMD_TYPE_Q5_K ? P5K_TILE : Atype == LFAMD_TYPE_Q6_K ? P6K_TILE : P4K_TILE);
                    gemm_mats lo, hi;
                    split_mats(mats, rb_cut, tile_bytes, lo, hi);
                    hipError_t e = lfamd_lw_go(Atype, lo, nb, Xh, d8T, Xm, n, n_pad, rb_cut, n_ct, (unsigned)(rb_cut * n_ct), 0, 1, 4, 1,
                                               nullptr, s);
                    if (e != hipSuccess)
                        return e;
                    const int n_ct2 = (int)((n + 63) / 64), rb_hi = n_rb - rb_cut;
                    if (lfamd_ks_ok(Atype))
                        return lfamd_ks_go(Atype, hi, nb, Xh, d8T, Xm, n, n_pad, rb_hi, n_ct2, s);
                    return lfamd_lw_go(Atype, hi, nb, Xh, d8T, Xm, n, n_pad, rb_hi, n_ct2, (unsigned)(rb_hi * n_ct2), 0, 1, 2, 1, nullptr, s);
                }
                return lfamd_lw_go(Atype, mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, (unsigned)(n_rb * n_ct), 0, 1, 4, 1, nullptr, s);
            }
            const int n_ct2 = (int)((n + 63) / 64);
            const int ksp = (mats.count == 1 && P && n_pad == 128) ? lw_ksplit(n_rb * n_ct2, nb) : 1;
            if (ksp > 1 && P_bytes >= (size_t)ksp * 128 * (size_t)n_rb * 128 * sizeof(float)) {
                hipError_t e = lfamd_lw_go(Atype, mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct2, (unsigned)(n_rb * n_ct2 * ksp), 0, 1, 2,
                                           ksp, P, s);
                if (e != hipSuccess)
                    return e;
                return lfamd_lw_ksplit_reduce(P, ksp, n, n_pad, (long)n_rb * 128, mats.m[0], mats.C[0], mats.ldc[0],
                                              (const float *)d8T, s);
            }
            if (lfamd_ks_ok(Atype)) // 128 x 64 tiles on the K-split-waves body (gemm_ks.hip)
                return lfamd_ks_go(Atype, mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct2, s);
            return lfamd_lw_go(Atype, mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct2, (unsigned)(n_rb * n_ct2), 0, 1, 2, 1, nullptr, s);
        }
        if (ks == 1 && (q45 || (moe && g_scaled))) // exact codes; the grouped MUL_MAT_ID launch also on scaled operands (Q6_K: only)
            return lfamd_lw_go(Atype, mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, n_wg, moe, moe ? g_scaled : 0, 4, 1, nullptr, s);
    }
    if (g_scaled)
        return hipErrorInvalidValue; // scaled activations reached a body that expects integer codes
    switch (Atype) {
    case LFAMD_TYPE_Q4_K:
        return lfamd_wide_go_q4k(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_Q5_K:
        return lfamd_wide_go_q5k(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_Q6_K:
        return lfamd_wide_go_q6k(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_Q4_0:
        return lfamd_wide_go_q40(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_Q4_1:
        return lfamd_wide_go_q41(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_Q5_0:
        return lfamd_wide_go_q50(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_Q5_1:
        return lfamd_wide_go_q51(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_Q2_K:
        return lfamd_wide_go_q2k(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_Q3_K:
        return lfamd_wide_go_q3k(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_IQ4_XS:
        return lfamd_wide_go_iq4xs(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_F16:
        return lfamd_wide_go_f16(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    case LFAMD_TYPE_BF16:
        return lfamd_wide_go_bf16(mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, n_wg, moe, s);
    default:
        return hipErrorInvalidValue;
    }
}

// C = 0 for the K-split accumulation (rows [0,n) x [0,m) of a matrix with leading dimension ldc)
__global__ void zero_c_kernel(float *__restrict__ C, long ldc, long m, long n) {
    const long total = m * n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long j = idx / m;
        C[j * ldc + (idx - j * m)] = 0.0f;
    }
}

// How the wide body splits K: enough work-groups for 256 CUs, at least 2 super-blocks per split.
static int wide_ksplit(int n_tiles, int nb) {
    int ks = 1;
    while (n_tiles * ks < 192 && nb / (ks * 2) >= 2 && ks < 2) // ks <= 2 keeps the sum order-independent
        ks *= 2;
    return ks;
}

extern "C" int lfamd_gemm_wide_ksplit(long m, long k, long n_pad) {
    const int n_rb = (int)((m + 127) / 128), n_ct = (int)(n_pad / WD_COLS);
    return wide_ksplit(n_rb * n_ct, (int)(k / 256));
}

// Will a wide launch of these (fused) matrices run the loader-wave body, i.e. may the caller stage scaled activations?
extern "C" int lfamd_gemm_wide_scaled_ok(int Atype, int plain) {
    static const bool env_off = getenv("LFAMD_GEMM_NO_SCALED") != nullptr, env_plain = getenv("LFAMD_GEMM_NO_LW") != nullptr;
    return !(env_off || env_plain || plain ||
             (Atype != LFAMD_TYPE_Q4_K && Atype != LFAMD_TYPE_Q5_K && Atype != LFAMD_TYPE_Q6_K));
}

extern "C" hipError_t lfamd_launch_gemm_wide_multi(int Atype, int count, const void *const *A, const long *m, long k,
                                                   const void *Xh, const void *d8T, const void *Xm, long n, long n_pad,
                                                   float *const *C, const long *ldc, int mode, void *P, size_t P_bytes,
                                                   hipStream_t s) {
    if (n <= 0 || count <= 0)
        return hipSuccess;
    if (n_pad % WD_COLS || count > GEMM_MAX_MATS)
        return hipErrorInvalidValue;
    const int nb = (int)(k / 256);
    gemm_mats mats;
    int n_rb = 0;
    mats.count = 0;
    mats.moe_cnt = mats.moe_poff = mats.moe_slot_row = nullptr, mats.expert_bytes = 0, mats.moe_ct_max = 0;
    for (int j = 0; j < count; j++) {
        if (m[j] <= 0)
            continue;
        const int i = mats.count++;
        mats.A[i] = (const uint8_t *)A[j], mats.C[i] = C[j], mats.m[i] = m[j], mats.ldc[i] = ldc[j];
        n_rb += (int)((m[j] + 127) / 128);
        mats.rb_end[i] = n_rb;
    }
    if (mats.count == 0)
        return hipSuccess;
    for (int i = mats.count; i < GEMM_MAX_MATS; i++)
        mats.A[i] = mats.A[0], mats.C[i] = mats.C[0], mats.m[i] = 0, mats.ldc[i] = 0, mats.rb_end[i] = n_rb;
    const int n_ct = (int)(n_pad / WD_COLS);
    const bool scaled_lw = (mode & 2) && lw_allowed(mode) &&
                           (Atype == LFAMD_TYPE_Q4_K || Atype == LFAMD_TYPE_Q5_K || Atype == LFAMD_TYPE_Q6_K);
    const int ks = scaled_lw ? 1 : wide_ksplit(n_rb * n_ct, nb); // the scaled-operand body never splits K
    const int nbs = (nb + ks - 1) / ks;
    const int n_wg = n_rb * n_ct * ks;
    if (ks > 1) {
        for (int i = 0; i < mats.count; i++) {
            long total = mats.m[i] * n;
            int blocks = (int)((total + 256 * 8 - 1) / (256 * 8));
            zero_c_kernel<<<blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks), 256, 0, s>>>(mats.C[i], mats.ldc[i], mats.m[i], n);
        }
    }
    return wide_go(Atype, mode, (float *)P, P_bytes, mats, nb, Xh, d8T, Xm, n, n_pad, n_rb, n_ct, ks, nbs, (unsigned)n_wg, 0, s);
}

hipError_t lfamd_lw_dual_go(int type_a, const gemm_mats &ma, int n_rb_a, int type_b, const gemm_mats &mb, int n_rb_b, int nb,
                            const void *Xh, const void *d8T, const void *Xm, long n, long n_pad, int n_ct, hipStream_t s);

static int fill_gemm_mats(gemm_mats &mats, int count, const void *const *A, const long *m, float *const *C, const long *ldc) {
    int n_rb = 0;
    mats.count = 0;
    mats.moe_cnt = mats.moe_poff = mats.moe_slot_row = nullptr, mats.expert_bytes = 0, mats.moe_ct_max = 0;
    for (int j = 0; j < count; j++) {
        if (m[j] <= 0)
            continue;
        const int i = mats.count++;
        mats.A[i] = (const uint8_t *)A[j], mats.C[i] = C[j], mats.m[i] = m[j], mats.ldc[i] = ldc[j];
        n_rb += (int)((m[j] + 127) / 128);
        mats.rb_end[i] = n_rb;
    }
    for (int i = mats.count; i < GEMM_MAX_MATS; i++)
        mats.A[i] = mats.A[0], mats.C[i] = mats.C[0], mats.m[i] = 0, mats.ldc[i] = 0, mats.rb_end[i] = n_rb;
    return n_rb;
}

// Scaled staged activations, two groups of matrices of two K-quant types (type_b = Q6_K, type_a = Q4_K | Q5_K): ONE launch of
// the 128 x 128 loader-wave body when the combined grid fills more than half the chip.  Returns hipErrorNotSupported when
// that launch does not apply (the caller then launches the groups one after the other).
extern "C" hipError_t lfamd_launch_gemm_wide_dual(int type_a, int count_a, const void *const *A_a, const long *m_a,
                                                  float *const *C_a, const long *ldc_a, int type_b, int count_b,
                                                  const void *const *A_b, const long *m_b, float *const *C_b, const long *ldc_b,
                                                  long k, const void *Xh, const void *d8T, const void *Xm, long n, long n_pad,
                                                  int mode, hipStream_t s) {
    static const bool off = getenv("LFAMD_GEMM_NO_DUAL") != nullptr;
    if (off || !(mode & 2) || !lw_allowed(mode) || type_b != LFAMD_TYPE_Q6_K || (type_a != LFAMD_TYPE_Q4_K && type_a != LFAMD_TYPE_Q5_K) ||
        count_a <= 0 || count_b <= 0 || count_a > GEMM_MAX_MATS || count_b > GEMM_MAX_MATS || n_pad % WD_COLS)
        return hipErrorNotSupported;
    gemm_mats ma, mb;
    const int n_rb_a = fill_gemm_mats(ma, count_a, A_a, m_a, C_a, ldc_a), n_rb_b = fill_gemm_mats(mb, count_b, A_b, m_b, C_b, ldc_b);
    const int n_ct = (int)(n_pad / WD_COLS), tiles = (n_rb_a + n_rb_b) * n_ct;
    if (ma.count == 0 || mb.count == 0 || tiles < LW_FULL_GRID || tiles > 256)
        return hipErrorNotSupported; // (beyond one round the separate launches with their tail handling do as well)
    return lfamd_lw_dual_go(type_a, ma, n_rb_a, type_b, mb, n_rb_b, (int)(k / 256), Xh, d8T, Xm, n, n_pad, n_ct, s);
}

extern "C" hipError_t lfamd_launch_gemm_wide(int Atype, const void *A, long m, long k, const void *Xh, const void *d8T,
                                             const void *Xm, long n, long n_pad, float *C, long ldc, int mode, void *P, size_t P_bytes,
                                             hipStream_t s) {
    return lfamd_launch_gemm_wide_multi(Atype, 1, &A, &m, k, Xh, d8T, Xm, n, n_pad, &C, &ldc, mode, P, P_bytes, s);
}

// GGML_OP_MUL_MAT_ID batches: one launch over (expert, row block, token tile); see gemm_mats.  n_pad = slots staged by the
// activation prep (>= poff[experts-1] + cnt, multiple of 128); ct_max = token tiles of the worst case (all rows to one expert).
extern "C" hipError_t lfamd_launch_gemm_wide_moe(int Atype, const void *W, long expert_bytes, int experts, long m, long k,
                                                 const void *Xh, const void *d8T, const void *Xm, long n_pad, const int *cnt,
                                                 const int *poff, const int *slot_row, int ct_max, float *C, long ldc,
                                                 int mode, hipStream_t s) {
    if (m <= 0 || experts <= 0 || ct_max <= 0)
        return hipSuccess;
    const int nb = (int)(k / 256);
    gemm_mats mats;
    for (int i = 0; i < GEMM_MAX_MATS; i++)
        mats.A[i] = (const uint8_t *)W, mats.C[i] = C, mats.m[i] = m, mats.ldc[i] = ldc, mats.rb_end[i] = 0;
    mats.count = 1;
    mats.moe_cnt = cnt, mats.moe_poff = poff, mats.moe_slot_row = slot_row, mats.expert_bytes = expert_bytes, mats.moe_ct_max = ct_max;
    const int n_rb = (int)((m + 127) / 128);
    const unsigned n_wg = (unsigned)experts * n_rb * ct_max;
    if (Atype != LFAMD_TYPE_Q4_K && Atype != LFAMD_TYPE_Q5_K && Atype != LFAMD_TYPE_Q6_K)
        return hipErrorInvalidValue;
    // scaled operands, Q4_K experts: the 256 x 128 row-split body (gemm_kr.hip) — LFAMD_MOE_NO_KR: the loader-wave body (A/B runs)
    static const bool no_kr = getenv("LFAMD_MOE_NO_KR") != nullptr;
    if ((mode & 2) && lw_allowed(mode) && !no_kr && lfamd_kr_ok(Atype))
        return lfamd_kr_moe_go(Atype, mats, nb, Xh, d8T, Xm, n_pad, experts, ct_max, s);
    return wide_go(Atype, mode, nullptr, 0, mats, nb, Xh, d8T, Xm, n_pad, n_pad, n_rb, ct_max, 1, nb, n_wg, 1, s);
}
