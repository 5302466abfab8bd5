// moe.hip — GGML_OP_MUL_MAT_ID (mixture of experts): llamafile_mixmul / iqk_mul_mat_moe on the GPU.
//
//   result[token][thinker][:] = W[plan[token][thinker]] x thought[token][thinker % tasks][:]
//   (tinyblas_cpu_mixmul.inc:39-50; row mapping of iqk_mul_mat_moe, iqk_mul_mat.inc:84-101)
//
// v1 follows the reference GPU path's structure (ggml_cuda_mul_mat_id, ggml-cuda.cu.patch:18499-18635):
// the routing table is read back to the host, rows routed to each expert are gathered
// (k_copy_src1_to_contiguous, :18450-18475), one mat-mul per expert runs on the gathered rows, and
// the results are scattered back (k_copy_dst_from_contiguous, :18477-18497).  Experts are visited
// in index order and rows in (token, thinker) order, like build_row_pointers
// (tinyblas_cpu_mixmul.inc:297-320).  Decode (one token) needs no gather: the GEMV reads its rows in
// place.
#include "lfamd_device.h"
#include "../../include/lfamd_hip.h"

#include <vector>

extern "C" int lfamd_mul_mat(int, const void *, long, long, int, const void *, size_t, long, float *, long, void *, size_t,
                             unsigned, void *);
extern "C" size_t lfamd_mul_mat_workspace(int, long, long, long);
extern "C" size_t lfamd_mul_mat_workspace_upto(int, long, long, long);
extern "C" hipError_t lfamd_launch_gemv_ids_pair(int, const void *, long, int, const int32_t *, int, int, long, long, int, const void *,
                                                 const void *, size_t, float *, float *, hipStream_t);
extern "C" hipError_t lfamd_launch_gemv_ids(int, int, const void *const *, long, int, const int32_t *, const int *, long, long, int,
                                            const void *, size_t, float *const *, hipStream_t);

__global__ void moe_gather_kernel(const uint8_t *__restrict__ src, size_t src_stride, size_t row_bytes,
                                  const int32_t *__restrict__ src_idx, uint8_t *__restrict__ dst, long nrows) {
    const long r = blockIdx.x;
    if (r >= nrows)
        return;
    const uint8_t *s = src + (size_t)src_idx[r] * src_stride;
    uint8_t *d = dst + (size_t)r * row_bytes;
    // rows are multiples of 2 bytes (34 / 36 / 292-byte blocks)
    for (size_t o = 2 * threadIdx.x; o < row_bytes; o += 2 * blockDim.x)
        *(uint16_t *)(d + o) = *(const uint16_t *)(s + o);
}

__global__ void moe_scatter_kernel(const float *__restrict__ src, long rows, const int32_t *__restrict__ dst_idx,
                                   float *__restrict__ dst, long nrows) {
    const long r = blockIdx.x;
    if (r >= nrows)
        return;
    const float *s = src + (size_t)r * rows;
    float *d = dst + (size_t)dst_idx[r] * rows;
    for (long o = threadIdx.x; o < rows; o += blockDim.x)
        d[o] = s[o];
}

static inline size_t align_up_(size_t x, size_t a) {
    return (x + a - 1) / a * a;
}

// ---------------------------------------------------------------------------------------------
// Device-side routing for batches of K-quant experts: no read-back (the reference synchronises the stream and sorts on
// the host, ggml-cuda.cu.patch:18528-18531).  One work-group: count the rows of every expert, give expert e the slot
// range [poff[e], poff[e] + cnt[e]) with poff a multiple of the 128-token tile, and fill the ranges in (token, thinker)
// order like build_row_pointers (tinyblas_cpu_mixmul.inc:297-320).  slot_row = result row of a slot, src_row = activation
// row it reads (-1 for padding slots).  Rows with an out-of-range expert id get no slot (their result stays untouched).
__global__ __launch_bounds__(256) void moe_route_kernel(const int32_t *__restrict__ plan, long tokens, int thinkers, int tasks,
                                                        int experts, int n_slots, int *__restrict__ cnt, int *__restrict__ poff,
                                                        int *__restrict__ slot_row, int *__restrict__ src_row) {
    __shared__ int s_cnt[256], s_run[256]; // rows per expert; next free slot of every expert during the ordered fill
    __shared__ int s_wave[4][256];         // rows of expert e held by wave w in the current block of 256 rows
    extern __shared__ uint8_t s_plan[];    // expert id per row (0xFF: none)
    const long R = tokens * thinkers;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    s_cnt[t] = 0;
    for (int sl = t; sl < n_slots; sl += 256)
        slot_row[sl] = -1, src_row[sl] = -1;
    __syncthreads();
    for (long r = t; r < R; r += 256) {
        const int e = plan[r];
        const bool ok = e >= 0 && e < experts;
        s_plan[r] = ok ? (uint8_t)e : (uint8_t)0xFF;
        if (ok)
            atomicAdd(&s_cnt[e], 1);
    }
    __syncthreads();
    if (t == 0) {
        int off = 0;
        for (int e = 0; e < experts; e++) {
            s_run[e] = off;
            off += (s_cnt[e] + 127) / 128 * 128;
        }
    }
    __syncthreads();
    if (t < experts) {
        cnt[t] = s_cnt[t];
        poff[t] = s_run[t];
    }
    // Ordered fill, 256 rows at a time: a row's slot = its expert's running base + the rows of that expert in earlier
    // waves of the block + those in lower lanes of its own wave (one ballot per expert) — the stable (token, thinker)
    // order of build_row_pointers without a serial scan (the scan by `experts` threads took 109 us for 1024 rows).
    for (long base = 0; base < R; base += 256) {
        const long r = base + t;
        const int e = r < R ? (int)s_plan[r] : 0xFF;
        int rank = 0;
        for (int ex = 0; ex < experts; ex++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(e == ex);
            if (e == ex)
                rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (lane == 0)
                s_wave[wave][ex] = __builtin_popcountll(m);
        }
        __syncthreads();
        if (e != 0xFF) {
            int sl = s_run[e] + rank;
            for (int w = 0; w < wave; w++)
                sl += s_wave[w][e];
            const long tok = r / thinkers;
            const int th = (int)(r - tok * thinkers);
            slot_row[sl] = (int)r;
            src_row[sl] = (int)(tok * tasks + th % tasks);
        }
        __syncthreads();
        if (t < experts)
            s_run[t] += s_wave[0][t] + s_wave[1][t] + s_wave[2][t] + s_wave[3][t];
        __syncthreads();
    }
}

static bool moe_grouped_ok(int type, long tokens, int thinkers, int experts) {
    return (type == LFAMD_TYPE_Q4_K || type == LFAMD_TYPE_Q5_K || type == LFAMD_TYPE_Q6_K) && experts < 255 &&
           tokens * thinkers <= 60 * 1024; // the routing kernel keeps one byte per row in LDS
}

static size_t moe_grouped_slots(long tokens, int thinkers, int experts) {
    return align_up_((size_t)tokens * thinkers, 128) + (size_t)experts * 128;
}

extern "C" hipError_t lfamd_launch_prep_q8k(const void *, size_t, long, long, long, void *, void *, void *, int, const int32_t *,
                                            hipStream_t);
extern "C" hipError_t lfamd_launch_prep_f32(const void *, size_t, long, long, long, void *, void *, void *, int, const int32_t *,
                                            hipStream_t);
extern "C" int lfamd_gemm_wide_scaled_ok(int, int);
extern "C" hipError_t lfamd_launch_gemm_wide_moe(int, const void *, long, int, long, long, const void *, const void *, const void *,
                                                 long, const int *, const int *, const int *, int, float *, long, int, hipStream_t);

static size_t moe_grouped_ws(long cols, long tokens, int thinkers, int experts) {
    const size_t n_pad = moe_grouped_slots(tokens, thinkers, experts), nb = (size_t)(cols / 256);
    return align_up_((2 * (size_t)experts + 2 * n_pad) * 4, 256) + align_up_(n_pad * (size_t)cols * 2, 256) +
           align_up_(nb * n_pad * 4, 256) + align_up_(n_pad * nb * 32, 256);
}

extern "C" size_t lfamd_moe_workspace(int type, long rows, long cols, int experts, long tokens, int thinkers) {
    (void)experts;
    const size_t nr = (size_t)tokens * thinkers;
    const size_t brb = lfamd_row_size(lfamd_vec_dot_type(type), cols);
    const size_t inner = lfamd_mul_mat_workspace_upto(type, rows, cols, (long)nr); // (an expert's batch is any n <= nr)
    size_t host_path = align_up_(nr * brb, 256) + align_up_(nr * (size_t)rows * 4, 256) + align_up_(nr * 4, 256) * 2 +
                       align_up_(inner, 256);
    if (moe_grouped_ok(type, tokens, thinkers, experts) && cols % 256 == 0) {
        size_t g = moe_grouped_ws(cols, tokens, thinkers, experts);
        return g > host_path ? g : host_path;
    }
    return host_path;
}

// Decode, several expert tensors of one type and shape that consume the SAME activation row (ffn_gate_exps + ffn_up_exps,
// tasks == 1): all (tensor, thinker) pairs of a token in launches of up to four GEMVs — one launch for two tensors x two thinkers
// (Mixtral) instead of one per tensor.  false: the caller runs lfamd_launch_moe per tensor.
extern "C" bool lfamd_moe_decode_multi_ok(int type, long cols, int Btype, int tasks, long tokens, unsigned flags) {
    return tokens <= 4 && tasks == 1 && (type == LFAMD_TYPE_Q4_K || type == LFAMD_TYPE_Q5_K || type == LFAMD_TYPE_Q6_K) &&
           !(flags & LFAMD_FLAG_FORCE_GENERIC) && (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_K) && cols % 256 == 0 &&
           (size_t)(cols / 256) * 384 <= 150 * 1024;
}
extern "C" hipError_t lfamd_launch_moe_decode_multi(int type, int count, const void *const *W, long rows, long cols, int experts,
                                                    size_t expert_bytes, int Btype, const void *thought, size_t b_row_bytes, long tokens,
                                                    const int32_t *plan, int thinkers, float *const *result, hipStream_t s) {
    for (long t = 0; t < tokens; t++) {
        const uint8_t *Brow = (const uint8_t *)thought + (size_t)t * b_row_bytes;
        const void *Ws[4];
        int idx[4];
        float *Cs[4];
        int cnt = 0;
        for (int j = 0; j < count; j++)
            for (int th = 0; th < thinkers; th++) {
                Ws[cnt] = W[j], idx[cnt] = (int)(t * thinkers + th), Cs[cnt] = result[j] + (size_t)(t * thinkers + th) * rows;
                if (++cnt == 4 || (j == count - 1 && th == thinkers - 1)) {
                    hipError_t e = lfamd_launch_gemv_ids(type, cnt, Ws, (long)expert_bytes, experts, plan, idx, rows, cols, Btype, Brow,
                                                         b_row_bytes, Cs, s);
                    if (e != hipSuccess)
                        return e;
                    cnt = 0;
                }
            }
    }
    return hipSuccess;
}

extern "C" hipError_t lfamd_launch_moe(int type, const void *W, long rows, long cols, int experts, size_t expert_bytes,
                                       int Btype, const void *thought, size_t b_row_bytes, int tasks, long tokens,
                                       const int32_t *plan, int thinkers, float *result, void *ws, size_t ws_bytes,
                                       unsigned flags, hipStream_t s) {
    const size_t nr = (size_t)tokens * thinkers;
    // ---- decode (a few tokens), K-quant experts: no read-back, no gather.  Each (token, thinker) row is a GEMV whose
    // kernel picks the expert from the device-resident routing table; thinkers that share their activations
    // (tasks == 1: ffn_gate / ffn_up) are fused into one launch.  Asynchronous and graph-capturable, unlike the
    // reference's host round trip (ggml-cuda.cu.patch:18528-18531).
    if (tokens <= 4 && (type == LFAMD_TYPE_Q4_K || type == LFAMD_TYPE_Q5_K || type == LFAMD_TYPE_Q6_K) &&
        !(flags & LFAMD_FLAG_FORCE_GENERIC) &&
        (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_K) && (size_t)(cols / 256) * 384 <= 150 * 1024) {
        static const bool no_pair = getenv("LFAMD_MOE_NO_PAIR") != nullptr; // development: A/B
        for (long t = 0; t < tokens; t++) {
            int th = 0;
            while (th < thinkers) {
                if (tasks > 1 && th + 1 < thinkers && (th + 1) % tasks != th % tasks && !no_pair) {
                    // two thinkers with their own activation rows (ffn_down_exps): one launch, half of the work-groups each
                    const uint8_t *Ba = (const uint8_t *)thought + (size_t)(t * tasks + th % tasks) * b_row_bytes;
                    const uint8_t *Bb = (const uint8_t *)thought + (size_t)(t * tasks + (th + 1) % tasks) * b_row_bytes;
                    hipError_t e3 = lfamd_launch_gemv_ids_pair(type, W, (long)expert_bytes, experts, plan, (int)(t * thinkers + th),
                                                               (int)(t * thinkers + th + 1), rows, cols, Btype, Ba, Bb, b_row_bytes,
                                                               result + (size_t)(t * thinkers + th) * rows,
                                                               result + (size_t)(t * thinkers + th + 1) * rows, s);
                    if (e3 != hipSuccess)
                        return e3;
                    th += 2;
                    continue;
                }
                int cnt = 1;
                if (tasks == 1)
                    cnt = thinkers - th < 4 ? thinkers - th : 4;
                int idx[4];
                float *Cs[4];
                for (int q = 0; q < cnt; q++) {
                    idx[q] = (int)(t * thinkers + th + q);
                    Cs[q] = result + (size_t)(t * thinkers + th + q) * rows;
                }
                const uint8_t *Brow = (const uint8_t *)thought + (size_t)(t * tasks + th % tasks) * b_row_bytes;
                const void *Ws[4] = {W, W, W, W};
                hipError_t e2 = lfamd_launch_gemv_ids(type, cnt, Ws, (long)expert_bytes, experts, plan, idx, rows, cols, Btype, Brow,
                                                      b_row_bytes, Cs, s);
                if (e2 != hipSuccess)
                    return e2;
                th += cnt;
            }
        }
        return hipSuccess;
    }
    // ---- batches of K-quant experts: device-side routing + ONE grouped launch of the 128 x 128 MFMA body over
    // (expert, row block, token tile); asynchronous and graph-capturable.
    if (tokens > 4 && moe_grouped_ok(type, tokens, thinkers, experts) && !(flags & LFAMD_FLAG_FORCE_GENERIC) &&
        (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_K) && cols % 256 == 0 &&
        ws_bytes >= moe_grouped_ws(cols, tokens, thinkers, experts)) {
        const size_t n_pad = moe_grouped_slots(tokens, thinkers, experts), nbk = (size_t)(cols / 256);
        uint8_t *p = (uint8_t *)ws;
        int *cnt = (int *)p, *poff = cnt + experts, *slot_row = poff + experts, *src_row = slot_row + n_pad;
        p += align_up_((2 * (size_t)experts + 2 * n_pad) * 4, 256);
        void *Xh = p;
        p += align_up_(n_pad * (size_t)cols * 2, 256);
        void *d8T = p;
        p += align_up_(nbk * n_pad * 4, 256);
        void *Xm = p;
        moe_route_kernel<<<1, 256, align_up_(nr, 16), s>>>(plan, tokens, thinkers, tasks, experts, (int)n_pad, cnt, poff, slot_row,
                                                          src_row);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess)
            return e2;
        // scaled operands like lfamd_mul_mat's batches, unless LFAMD_FLAG_PRECISE
        const int plain = (flags & LFAMD_FLAG_GEMM_PLAIN) ? 1 : 0;
        const int scaled = (flags & LFAMD_FLAG_PRECISE) ? 0 : lfamd_gemm_wide_scaled_ok(type, plain);
        const int pmode = scaled ? 2 : 0;
        e2 = Btype == LFAMD_TYPE_F32
                 ? lfamd_launch_prep_f32(thought, b_row_bytes, (long)n_pad, (long)n_pad, cols, Xh, d8T, Xm, pmode, src_row, s)
                 : lfamd_launch_prep_q8k(thought, b_row_bytes, (long)n_pad, (long)n_pad, cols, Xh, d8T, Xm, pmode, src_row, s);
        if (e2 != hipSuccess)
            return e2;
        const int ct_max = (int)((nr + 127) / 128);
        return lfamd_launch_gemm_wide_moe(type, W, (long)expert_bytes, experts, rows, cols, Xh, d8T, Xm, (long)n_pad, cnt, poff,
                                          slot_row, ct_max, result, rows, plain | (scaled << 1), s);
    }
    std::vector<int32_t> hplan(nr);
    hipError_t e = hipMemcpyAsync(hplan.data(), plan, nr * 4, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess)
        return e;
    e = hipStreamSynchronize(s); // the reference syncs here too (ggml-cuda.cu.patch:18528-18531)
    if (e != hipSuccess)
        return e;

    const size_t brb = lfamd_row_size(Btype, cols);
    uint8_t *p = (uint8_t *)ws;
    uint8_t *Bg = p;
    p += align_up_(nr * brb, 256);
    float *Cg = (float *)p;
    p += align_up_(nr * (size_t)rows * 4, 256);
    int32_t *d_src = (int32_t *)p;
    p += align_up_(nr * 4, 256);
    int32_t *d_dst = (int32_t *)p;
    p += align_up_(nr * 4, 256);
    void *inner_ws = p;
    size_t inner_bytes = ws_bytes - (size_t)(p - (uint8_t *)ws);

    // rows per expert in (token, thinker) order
    std::vector<int32_t> src_idx, dst_idx;
    std::vector<long> start(experts + 1, 0);
    src_idx.reserve(nr);
    dst_idx.reserve(nr);
    for (int ex = 0; ex < experts; ex++) {
        start[ex] = (long)src_idx.size();
        for (long t = 0; t < tokens; t++)
            for (int th = 0; th < thinkers; th++)
                if (hplan[t * thinkers + th] == ex) {
                    src_idx.push_back((int32_t)(t * tasks + th % tasks));
                    dst_idx.push_back((int32_t)(t * thinkers + th));
                }
    }
    start[experts] = (long)src_idx.size();
    const long routed = (long)src_idx.size(); // rows whose expert id is in range
    if (routed == 0)
        return hipSuccess;
    e = hipMemcpyAsync(d_src, src_idx.data(), routed * 4, hipMemcpyHostToDevice, s);
    if (e != hipSuccess)
        return e;
    e = hipMemcpyAsync(d_dst, dst_idx.data(), routed * 4, hipMemcpyHostToDevice, s);
    if (e != hipSuccess)
        return e;
    // the host vectors must outlive the async copies
    e = hipStreamSynchronize(s);
    if (e != hipSuccess)
        return e;

    moe_gather_kernel<<<(unsigned)routed, 128, 0, s>>>((const uint8_t *)thought, b_row_bytes, brb, d_src, Bg, routed);
    for (int ex = 0; ex < experts; ex++) {
        long cnt = start[ex + 1] - start[ex];
        if (cnt == 0)
            continue;
        int r = lfamd_mul_mat(type, (const uint8_t *)W + (size_t)ex * expert_bytes, rows, cols, Btype,
                              Bg + (size_t)start[ex] * brb, brb, cnt, Cg + (size_t)start[ex] * rows, rows, inner_ws,
                              inner_bytes, flags, (void *)s);
        if (r != LFAMD_OK)
            return hipErrorUnknown;
    }
    moe_scatter_kernel<<<(unsigned)routed, 256, 0, s>>>(Cg, rows, d_dst, result, routed);
    return hipGetLastError();
}
