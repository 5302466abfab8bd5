// gemv.hip — dispatch of the decode GEMVs (kernels: gemv_impl.h; instantiations: gemv_q4k / q5k / q6k / q40 / q41 / q50 / q51 / q2k / q3k / iq4xs / q80.hip)
#include "gemv_impl.h"

hipError_t lfamd_gemv_go_q4k(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_q5k(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_q6k(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_q40(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_q2k(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_q3k(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_iq4xs(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_q41(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_q50(GEMV_GO_ARGS);
hipError_t lfamd_gemv_go_q51(GEMV_GO_ARGS);
hipError_t lfamd_gemv_ids_go_q4k(int, const gemv_mats &, int, long, const void *, size_t, hipStream_t);
hipError_t lfamd_gemv_ids_go_q5k(int, const gemv_mats &, int, long, const void *, size_t, hipStream_t);
hipError_t lfamd_gemv_ids_go_q6k(int, const gemv_mats &, int, long, const void *, size_t, hipStream_t);
hipError_t lfamd_gemv_ids_pair_go_q4k(int, const gemv_mats &, const gemv_mats &, int, long, const void *, const void *, size_t, hipStream_t);
hipError_t lfamd_gemv_ids_pair_go_q5k(int, const gemv_mats &, const gemv_mats &, int, long, const void *, const void *, size_t, hipStream_t);
hipError_t lfamd_gemv_ids_pair_go_q6k(int, const gemv_mats &, const gemv_mats &, int, long, const void *, const void *, size_t, hipStream_t);
hipError_t lfamd_gemv_go_q80_f32(int, const q80_mats &, long, long, const void *, size_t, long, int, int, hipStream_t);
hipError_t lfamd_gemv_go_q80_q80(int, const q80_mats &, long, long, const void *, size_t, long, int, int, hipStream_t);

hipError_t lfamd_gemv_dual_go_q4k_q6k(int, const gemv_mats &, int, const gemv_mats &, int, long, const void *, size_t, hipStream_t);
hipError_t lfamd_gemv_dual_go_q5k_q6k(int, const gemv_mats &, int, const gemv_mats &, int, long, const void *, size_t, hipStream_t);

// LDS budget: keep one launch's activation image under 160 KiB; otherwise split the columns.
static int max_cols_for(size_t per_col_bytes) {
    size_t cap = 150 * 1024;
    int nc = (int)(cap / (per_col_bytes ? per_col_bytes : 1));
    return nc < 1 ? 0 : (nc > 8 ? 8 : nc);
}

// Btype: the weight type's vec_dot type (pre-quantised rows) or LFAMD_TYPE_F32 (quantise in-kernel).
// `count` matrices (<= GEMV_MAX_MATS for the K-quants, 1 for Q8_0) of the same type and k share B.
extern "C" hipError_t lfamd_launch_gemv_multi(int Atype, int count, const void *const *A, const long *m, long k, int Btype,
                                              const void *B, size_t b_row_bytes, long n, float *const *C, const long *ldc,
                                              int vregs32, int precise, hipStream_t s) {
    if (count <= 0 || n <= 0)
        return hipSuccess;
    size_t per_col;
    if (Atype == LFAMD_TYPE_Q8_0)
        per_col = (size_t)((k / 32 + 3) / 4) * X80_QUAD;
    else
        per_col = (size_t)(k / 256) * XBLK;
    int step = max_cols_for(per_col);
    if (step == 0)
        return hipErrorInvalidValue;
    // deep rows on the 8-wave x 4-block kernels: six and more columns per launch spill registers (256 VGPRs + 29..53
    // spilled).  Past 32 super-blocks two passes of at most five columns are faster (4096 x 14336, n = 8: Q4_K 47.9 -> 45.8
    // us, Q6_K 79.9 -> 67.0); at 32 the second pass costs more than the spills (4096 x 8192: 24.5 vs 29.7 us)
    if (Atype != LFAMD_TYPE_Q8_0 && k / 256 > 32 && step > 5)
        step = 5;
    const bool f32in = Btype == LFAMD_TYPE_F32;
    hipError_t e = hipSuccess;
    if (Atype == LFAMD_TYPE_Q8_0) {
        for (int j0 = 0; j0 < count && e == hipSuccess; j0 += GEMV_MAX_MATS) { // groups of up to four matrices per launch
            q80_mats qm;
            long rgs = 0;
            qm.count = 0;
            for (int j = j0; j < count && j < j0 + GEMV_MAX_MATS; j++) {
                if (m[j] <= 0)
                    continue;
                const int i = qm.count++;
                qm.A[i] = (const uint8_t *)A[j], qm.C[i] = C[j], qm.m[i] = m[j], qm.ldc[i] = ldc[j];
                rgs += (m[j] + 7) / 8;
                rgs = (rgs + Q80_WAVES - 1) / Q80_WAVES * Q80_WAVES; // a work-group never straddles two matrices
                qm.rg_end[i] = rgs;
            }
            if (qm.count == 0)
                continue;
            for (int i = qm.count; i < GEMV_MAX_MATS; i++)
                qm.A[i] = qm.A[0], qm.C[i] = qm.C[0], qm.m[i] = 0, qm.ldc[i] = 0, qm.rg_end[i] = rgs;
            for (long col0 = 0; col0 < n && e == hipSuccess; col0 += step) {
                int nc = (int)((n - col0) < step ? (n - col0) : step);
                e = f32in ? lfamd_gemv_go_q80_f32(nc, qm, n, k, B, b_row_bytes, col0, vregs32, precise, s)
                          : lfamd_gemv_go_q80_q80(nc, qm, n, k, B, b_row_bytes, col0, vregs32, precise, s);
            }
        }
        return e;
    }
    if (count > GEMV_MAX_MATS ||
        (Atype != LFAMD_TYPE_Q4_K && Atype != LFAMD_TYPE_Q5_K && Atype != LFAMD_TYPE_Q6_K && Atype != LFAMD_TYPE_Q4_0 &&
         Atype != LFAMD_TYPE_Q2_K && Atype != LFAMD_TYPE_Q3_K && Atype != LFAMD_TYPE_IQ4_XS && Atype != LFAMD_TYPE_Q4_1 &&
         Atype != LFAMD_TYPE_Q5_0 && Atype != LFAMD_TYPE_Q5_1))
        return hipErrorInvalidValue;
    gemv_mats mats;
    int n_ht = 0;
    mats.count = 0;
    mats.ids = nullptr, mats.expert_bytes = 0, mats.experts = 0;
    for (int i = 0; i < GEMV_MAX_MATS; i++)
        mats.id_idx[i] = 0;
    for (int j = 0; j < count; j++) {
        if (m[j] <= 0)
            continue;
        int i = mats.count++;
        mats.A[i] = (const uint8_t *)A[j];
        mats.C[i] = C[j];
        mats.m[i] = m[j];
        mats.ldc[i] = ldc[j];
        n_ht += (int)(((m[j] + 31) / 32) * 2);
        mats.ht_end[i] = n_ht;
    }
    if (mats.count == 0)
        return hipSuccess;
    for (int i = mats.count; i < GEMV_MAX_MATS; i++) {
        mats.A[i] = mats.A[0], mats.C[i] = mats.C[0], mats.m[i] = 0, mats.ldc[i] = 0, mats.ht_end[i] = n_ht;
    }
    for (long col0 = 0; col0 < n && e == hipSuccess; col0 += step) {
        int nc = (int)((n - col0) < step ? (n - col0) : step);
        const int f = f32in ? 1 : 0;
        if (Atype == LFAMD_TYPE_Q4_K)
            e = lfamd_gemv_go_q4k(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else if (Atype == LFAMD_TYPE_Q4_0)
            e = lfamd_gemv_go_q40(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else if (Atype == LFAMD_TYPE_Q5_K)
            e = lfamd_gemv_go_q5k(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else if (Atype == LFAMD_TYPE_Q2_K)
            e = lfamd_gemv_go_q2k(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else if (Atype == LFAMD_TYPE_Q3_K)
            e = lfamd_gemv_go_q3k(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else if (Atype == LFAMD_TYPE_IQ4_XS)
            e = lfamd_gemv_go_iq4xs(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else if (Atype == LFAMD_TYPE_Q4_1)
            e = lfamd_gemv_go_q41(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else if (Atype == LFAMD_TYPE_Q5_0)
            e = lfamd_gemv_go_q50(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else if (Atype == LFAMD_TYPE_Q5_1)
            e = lfamd_gemv_go_q51(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
        else
            e = lfamd_gemv_go_q6k(nc, f, mats, n_ht, k, B, b_row_bytes, col0, s);
    }
    return e;
}

static int fill_mats(gemv_mats &mats, int count, const void *const *A, const long *m, float *const *C, const long *ldc) {
    int n_ht = 0;
    mats.count = 0;
    mats.ids = nullptr, mats.expert_bytes = 0, mats.experts = 0;
    for (int i = 0; i < GEMV_MAX_MATS; i++)
        mats.id_idx[i] = 0;
    for (int j = 0; j < count; j++) {
        if (m[j] <= 0)
            continue;
        const int i = mats.count++;
        mats.A[i] = (const uint8_t *)A[j], mats.C[i] = C[j], mats.m[i] = m[j], mats.ldc[i] = ldc[j];
        n_ht += (int)(((m[j] + 31) / 32) * 2);
        mats.ht_end[i] = n_ht;
    }
    for (int i = mats.count; i < GEMV_MAX_MATS; i++)
        mats.A[i] = mats.A[0], mats.C[i] = mats.C[0], mats.m[i] = 0, mats.ldc[i] = 0, mats.ht_end[i] = n_ht;
    return n_ht;
}

// ONE activation row, two groups of matrices of two K-quant types (type_b = Q6_K, type_a = Q4_K or Q5_K): one launch.
extern "C" hipError_t lfamd_launch_gemv_dual(int type_a, int count_a, const void *const *A_a, const long *m_a, float *const *C_a,
                                             const long *ldc_a, int type_b, int count_b, const void *const *A_b, const long *m_b,
                                             float *const *C_b, const long *ldc_b, long k, int Btype, const void *B,
                                             size_t b_row_bytes, hipStream_t s) {
    if (type_b != LFAMD_TYPE_Q6_K || (type_a != LFAMD_TYPE_Q4_K && type_a != LFAMD_TYPE_Q5_K) || count_a <= 0 || count_b <= 0 ||
        count_a > GEMV_MAX_MATS || count_b > GEMV_MAX_MATS || (size_t)(k / 256) * XBLK > 150 * 1024)
        return hipErrorInvalidValue;
    gemv_mats ma, mb;
    const int n_ht_a = fill_mats(ma, count_a, A_a, m_a, C_a, ldc_a), n_ht_b = fill_mats(mb, count_b, A_b, m_b, C_b, ldc_b);
    if (ma.count == 0 || mb.count == 0)
        return hipErrorInvalidValue; // (the caller sends an empty group through the one-type path)
    const int f = Btype == LFAMD_TYPE_F32 ? 1 : 0;
    return type_a == LFAMD_TYPE_Q4_K ? lfamd_gemv_dual_go_q4k_q6k(f, ma, n_ht_a, mb, n_ht_b, k, B, b_row_bytes, s)
                                     : lfamd_gemv_dual_go_q5k_q6k(f, ma, n_ht_a, mb, n_ht_b, k, B, b_row_bytes, s);
}

// GGML_OP_MUL_MAT_ID for ONE activation row: `count` (<= GEMV_MAX_MATS) outputs C[j] = W[ids[id_idx[j]]] x B, the expert
// index read on the device.  Q4_K / Q6_K stacks; Btype F32 or Q8_K.
// W[j]: the expert tensor matrix j picks from (ffn_gate_exps and ffn_up_exps may share one launch: same activations)
extern "C" hipError_t lfamd_launch_gemv_ids(int Atype, int count, const void *const *W, long expert_bytes, int experts,
                                            const int32_t *ids, const int *id_idx, long m, long k, int Btype, const void *B,
                                            size_t b_row_bytes, float *const *C, hipStream_t s) {
    if (count <= 0 || count > GEMV_MAX_MATS || m <= 0 ||
        (Atype != LFAMD_TYPE_Q4_K && Atype != LFAMD_TYPE_Q5_K && Atype != LFAMD_TYPE_Q6_K))
        return hipErrorInvalidValue;
    if ((size_t)(k / 256) * XBLK > 150 * 1024)
        return hipErrorInvalidValue;
    gemv_mats mats;
    int n_ht = 0;
    mats.count = count;
    mats.ids = ids, mats.expert_bytes = expert_bytes, mats.experts = experts;
    for (int i = 0; i < GEMV_MAX_MATS; i++) {
        const int j = i < count ? i : 0;
        mats.A[i] = (const uint8_t *)W[j], mats.C[i] = C[j], mats.m[i] = i < count ? m : 0, mats.ldc[i] = m;
        mats.id_idx[i] = id_idx[j];
        if (i < count)
            n_ht += (int)(((m + 31) / 32) * 2);
        mats.ht_end[i] = n_ht;
    }
    const int f = Btype == LFAMD_TYPE_F32 ? 1 : 0;
    if (Atype == LFAMD_TYPE_Q4_K)
        return lfamd_gemv_ids_go_q4k(f, mats, n_ht, k, B, b_row_bytes, s);
    if (Atype == LFAMD_TYPE_Q5_K)
        return lfamd_gemv_ids_go_q5k(f, mats, n_ht, k, B, b_row_bytes, s);
    return lfamd_gemv_ids_go_q6k(f, mats, n_ht, k, B, b_row_bytes, s);
}

// two experts of one tensor, each against its own activation row (ffn_down_exps at decode): one launch
extern "C" hipError_t lfamd_launch_gemv_ids_pair(int Atype, const void *W, long expert_bytes, int experts, const int32_t *ids, int idx_a,
                                                 int idx_b, long m, long k, int Btype, const void *Ba, const void *Bb, size_t b_row_bytes,
                                                 float *Ca, float *Cb, hipStream_t s) {
    if (m <= 0 || (Atype != LFAMD_TYPE_Q4_K && Atype != LFAMD_TYPE_Q5_K && Atype != LFAMD_TYPE_Q6_K) || (size_t)(k / 256) * XBLK > 150 * 1024)
        return hipErrorInvalidValue;
    gemv_mats ma, mb;
    const int n_ht = (int)(((m + 31) / 32) * 2);
    for (gemv_mats *mm : {&ma, &mb}) {
        const bool a = mm == &ma;
        mm->count = 1, mm->ids = ids, mm->expert_bytes = expert_bytes, mm->experts = experts;
        for (int i = 0; i < GEMV_MAX_MATS; i++) {
            mm->A[i] = (const uint8_t *)W, mm->C[i] = a ? Ca : Cb, mm->m[i] = i == 0 ? m : 0, mm->ldc[i] = m;
            mm->id_idx[i] = a ? idx_a : idx_b, mm->ht_end[i] = n_ht;
        }
    }
    const int f = Btype == LFAMD_TYPE_F32 ? 1 : 0;
    if (Atype == LFAMD_TYPE_Q4_K)
        return lfamd_gemv_ids_pair_go_q4k(f, ma, mb, n_ht, k, Ba, Bb, b_row_bytes, s);
    if (Atype == LFAMD_TYPE_Q5_K)
        return lfamd_gemv_ids_pair_go_q5k(f, ma, mb, n_ht, k, Ba, Bb, b_row_bytes, s);
    return lfamd_gemv_ids_pair_go_q6k(f, ma, mb, n_ht, k, Ba, Bb, b_row_bytes, s);
}

extern "C" hipError_t lfamd_launch_gemv(int Atype, const void *A, long m, long k, int Btype, const void *B,
                                        size_t b_row_bytes, long n, float *C, long ldc, int vregs32, int precise,
                                        hipStream_t s) {
    return lfamd_launch_gemv_multi(Atype, 1, &A, &m, k, Btype, B, b_row_bytes, n, &C, &ldc, vregs32, precise, s);
}
