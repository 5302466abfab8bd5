// ffn_block.hip — the decode feed-forward block as ONE launch: ffn_gate and ffn_up (one activation row), silu(gate) * up,
// Q8_K quantisation, ffn_down (+ residual).  SURVEY.md section 8 f-3 ("the step either side of the path").
//
// What it replaces: GGML_OP_MUL_MAT (ffn_gate), GGML_OP_MUL_MAT (ffn_up), GGML_OP_SILU + GGML_OP_MUL, the activation
// quantiser and GGML_OP_MUL_MAT (ffn_down) [+ GGML_OP_ADD] of a llama feed-forward at batch 1 (reference kernels:
// mul_mat_vec_q + quantize_q8_1, ggml-cuda.cu.patch:14428-14575, 15259-15293; silu_f32 :16172-16179).  Same arithmetic as
// the separate calls: exact integer block dots (iqk_mul_mat.inc:601-643), silu(x) = x / (1 + expf(-x)) in f32, activations
// quantised bit for bit like quantize_row_q8_K.
//
// Why one launch: at batch 1 a decode pass is 4.6 GB at what HBM gives plus ~2.6 us per launch (boundary + XCD start skew +
// first-byte latency) — a third of the pass.  Here ffn_down's weights do not wait for a launch
// boundary: every work-group issues the loads of its ffn_down half-tile into REGISTERS (they do not depend on the
// activations) and only then arrives at a grid-wide barrier; the barrier's ~4 us overlap the 33-48 MB of ffn_down streaming in.
//   phase 1   gemv_kq_body1 over {gate, up} (the decode GEMV body, unchanged): G, U (f32) -> workspace
//   prefetch  waves 0..NWC-1: the CH2 super-blocks each of them owns of work-group b's ffn_down half-tile b -> registers
//   barrier   every wave drains its stores; one lane per work-group: release, count in, poll the generation word, acquire
//             (cdna_hip_programming.md Guideline 16; bounded by wall time: a stuck grid sets an error flag, never hangs)
//   phase 2   each computing wave stages exactly the blocks it consumes: h = silu(G) * U -> Q8_K image in LDS (no work-group
//             barrier before the dots), integer dots against the prefetched weights, 4 lanes per row, waves through LDS,
//             out[row] = sum (+ residual[row])
// The waves of a work-group: NWC compute in phase 2 (14 x 4 super-blocks = the 56 of k = 14336), wave 15 runs the barrier
// (a wave with weight loads in flight cannot poll: its vmcnt waits would wait for the weights).
#include "gemv_impl.h"
#include "../../include/lfamd_hip.h"

extern "C" void lfamd_set_error(const char *msg);

// grid barrier state: [0] arrivals of the current launch (back to 0 when it completes), [1] generation, [2] error flag
__device__ unsigned g_ffn_sync[4];

template <typename TRGU, typename TRD, int NWC, int CH2>
__global__ __launch_bounds__(1024) void ffn_block_kernel(const uint8_t *__restrict__ x, int nb1, int n_ht_gu, int gdim, int nb2,
                                                         int n_ht_d, const float *__restrict__ G, const float *__restrict__ U,
                                                         const float *__restrict__ residual, long timeout_ticks, int dbg,
                                                         const gemv_mats mats_gu, const gemv_mats mats_d) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    static_assert(TRD::ACT == LFAMD_TYPE_Q8_K && CH2 <= GEMV_CH_MAX && NWC <= 15, "K-quant ffn_down; wave 15 runs the barrier");
    const int bid = (int)blockIdx.x;
    // ---- phase 1: gate and up (16 waves, one super-block per wave and item: k <= 4096; deeper rows two per chunk)
    if (!(dbg & 4))
        gemv_kq_body1<TRGU, LFAMD_TYPE_F32, 16, 1, false>(mats_gu, nb1, x, (size_t)nb1 * 1024, 0, n_ht_gu, bid, gdim, lds, nullptr, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's G / U stores have left (and nothing of phase 1 is in flight)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i16 = lane & 15, h = (lane >> 4) & 1, gsel = lane >> 5;
    const uint8_t *Ad = mats_d.A[0];
    float *out = mats_d.C[0];
    const long m_out = mats_d.m[0];
    const uint32_t rtb = (uint32_t)nb2 * TRD::TILE;
    typename TRD::chunk wbuf;
    auto issue_item = [&](int ht) __attribute__((always_inline)) {
        const int hh = ht & 1;
        const lfamd_rsrc r = make_rsrc(Ad + (size_t)(ht >> 1) * rtb, ht < n_ht_d ? rtb : 0u);
#pragma unroll
        for (int s = 0; s < CH2; s++) // (a super-block past the row lands past the descriptor: zeros)
            TRD::load(wbuf, s, r, (uint32_t)(wave + NWC * s) * TRD::TILE, gsel, h * 32 + hh * 16 + i16, hh * 16 + i16);
    };
    // ---- ffn_down's weights of this work-group's first half-tile: in flight across the barrier
    if (wave < NWC)
        issue_item(bid);
    __builtin_amdgcn_s_barrier(); // every wave of the work-group has drained its stores (no vmcnt wait is implied here)

    // ---- grid barrier (wave 15, lane 0): sense reversal on a generation word, arrivals counted and reset by the last one in
    if (wave == 15 && lane == 0 && !(dbg & 1)) { // (dbg: development timing switches, results void)
        const unsigned gen = __hip_atomic_load(&g_ffn_sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (kept explicitly: ROCm 7.2 can drop the fence's own wait)
        const unsigned old = __hip_atomic_fetch_add(&g_ffn_sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)gdim - 1u) {
            __hip_atomic_store(&g_ffn_sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&g_ffn_sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(&g_ffn_sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)timeout_ticks) { // the grid is not co-resident
                    __hip_atomic_store(&g_ffn_sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier(); // the CU's L1 has been invalidated: G and U of every work-group are visible to plain loads

    // ---- phase 2
    float *red = (float *)(lds + (size_t)nb2 * XBLK); // [16][16]
    uint8_t *dummy = (uint8_t *)(red + 16 * 16) + (size_t)wave * XBLK;
    if (wave < NWC) {
        // h = silu(G) * U for the blocks this wave consumes, quantised like quantize_row_q8_K into the wave's image slots
        const lfamd_rsrc rg = make_rsrc(G, (uint32_t)nb2 * 1024u), ru = make_rsrc(U, (uint32_t)nb2 * 1024u);
        uint4 gv[CH2], uv[CH2];
#pragma unroll
        for (int s = 0; s < CH2; s++) {
            const uint32_t off = (uint32_t)(wave + NWC * s) * 1024u + (uint32_t)lane * 16u; // past the row: zeros
            gv[s] = buf_ld16(rg, off), uv[s] = buf_ld16(ru, off);
        }
#pragma unroll
        for (int s = 0; s < CH2; s++) {
            const int b = wave + NWC * s;
            const float g[4] = {__builtin_bit_cast(float, gv[s].x), __builtin_bit_cast(float, gv[s].y), __builtin_bit_cast(float, gv[s].z),
                                __builtin_bit_cast(float, gv[s].w)};
            const float u[4] = {__builtin_bit_cast(float, uv[s].x), __builtin_bit_cast(float, uv[s].y), __builtin_bit_cast(float, uv[s].z),
                                __builtin_bit_cast(float, uv[s].w)};
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
                y[e] = (g[e] / (1.0f + expf(-g[e]))) * u[e]; // (the expression of swiglu_q8k_kernel, norm_quant.hip)
            stage_f32_q8k_wave(b < nb2 ? lds + (size_t)b * XBLK : dummy, make_float4(y[0], y[1], y[2], y[3]), lane);
        }
    }
    // the image blocks a wave reads are the ones it has just written (cf. gemv_kq_body1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (dbg & 2)
        return;
    for (int ht = bid; ht < n_ht_d; ht += gdim) {
        if (ht != bid && wave < NWC)
            issue_item(ht);
        float acc = 0.0f;
        if (wave < NWC) {
#pragma unroll
            for (int s = 0; s < CH2; s++) {
                const int b = wave + NWC * s;
                const float t = TRD::dot(wbuf, s, lds + (size_t)(b < nb2 ? b : 0) * XBLK, gsel, h);
                acc += b < nb2 ? t : 0.0f;
            }
        }
        const float v = kq_sum_rows(acc);
        if (lane < 16)
            red[wave * 16 + lane] = v;
        __syncthreads();
        if (threadIdx.x < 16) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < NWC; w++)
                t += red[w * 16 + threadIdx.x];
            const long row = (long)(ht >> 1) * 32 + (ht & 1) * 16 + threadIdx.x;
            if (row < m_out)
                ((__attribute__((address_space(1))) float *)out)[row] = residual ? t + residual[row] : t;
        }
        __syncthreads();
    }
}

static int num_cus_ffn() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            n = p.multiProcessorCount;
        if (n <= 0)
            n = 256;
    }
    return n;
}

extern "C" size_t lfamd_ffn_block_workspace(long n_ff) {
    return n_ff > 0 ? (size_t)2 * (size_t)n_ff * sizeof(float) : 0;
}

// 0 = the grid barrier of every lfamd_ffn_block launch so far completed; 1 = one timed out (results void).  Synchronises.
extern "C" int lfamd_ffn_block_check(void) {
    unsigned v[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_ffn_sync), sizeof v) != hipSuccess)
        return -1;
    return (int)v[2];
}

extern "C" int lfamd_ffn_block(int type_gu, const void *d_Wgate, const void *d_Wup, long n_ff, long k, int type_down, const void *d_Wdown,
                               long m_out, const float *d_x, const float *d_residual, float *d_out, void *d_workspace,
                               size_t workspace_bytes, void *stream) {
    const long nb1 = k / 256, nb2 = n_ff / 256;
    if (!d_Wgate || !d_Wup || !d_Wdown || !d_x || !d_out || k <= 0 || n_ff <= 0 || m_out <= 0 || k % 256 || n_ff % 256 ||
        ((uintptr_t)d_x & 15) || ((uintptr_t)d_workspace & 15)) {
        lfamd_set_error("lfamd_ffn_block: bad arguments (k and n_ff multiples of 256, 16-byte aligned x / workspace)");
        return LFAMD_ERR_INVALID;
    }
    // what this launch form covers; everything else: the separate calls (lfamd_mul_mat_multi, lfamd_swiglu_quantize, lfamd_mul_mat)
    if (type_gu != LFAMD_TYPE_Q4_K || (type_down != LFAMD_TYPE_Q4_K && type_down != LFAMD_TYPE_Q6_K) || nb1 > 16 || nb2 > 60) {
        lfamd_set_error("lfamd_ffn_block: Q4_K gate / up with k <= 4096 and Q4_K / Q6_K down with n_ff <= 15360 only");
        return LFAMD_ERR_UNSUPPORTED;
    }
    if (workspace_bytes < lfamd_ffn_block_workspace(n_ff)) {
        lfamd_set_error("lfamd_ffn_block: workspace too small (lfamd_ffn_block_workspace)");
        return LFAMD_ERR_INVALID;
    }
    float *G = (float *)d_workspace, *U = G + n_ff;
    gemv_mats mgu, md;
    auto fill = [](gemv_mats &mm, int count, const void *const *A, const long *m, float *const *C) {
        int n_ht = 0;
        mm.count = count;
        mm.ids = nullptr, mm.expert_bytes = 0, mm.experts = 0;
        for (int i = 0; i < GEMV_MAX_MATS; i++) {
            const int j = i < count ? i : 0;
            mm.A[i] = (const uint8_t *)A[j], mm.C[i] = C[j], mm.m[i] = i < count ? m[j] : 0, mm.ldc[i] = i < count ? m[j] : 0;
            mm.id_idx[i] = 0;
            if (i < count)
                n_ht += (int)(((m[j] + 31) / 32) * 2);
            mm.ht_end[i] = n_ht;
        }
        return n_ht;
    };
    const void *Agu[2] = {d_Wgate, d_Wup};
    const long mgu_[2] = {n_ff, n_ff};
    float *Cgu[2] = {G, U};
    const int n_ht_gu = fill(mgu, 2, Agu, mgu_, Cgu);
    const void *Ad[1] = {d_Wdown};
    const long md_[1] = {m_out};
    float *Cd[1] = {d_out};
    const int n_ht_d = fill(md, 1, Ad, md_, Cd);
    // one persistent work-group per CU: all of them must be resident for the grid barrier (1024 threads: one per CU fits)
    const int max_wg = num_cus_ffn();
    const int per_wg = (n_ht_gu + max_wg - 1) / max_wg;
    const int grid = (n_ht_gu + per_wg - 1) / per_wg;
    const size_t smem1 = (size_t)nb1 * XBLK + 2 * 16 * 16 * sizeof(float) + (size_t)16 * XBLK;
    const size_t smem2 = (size_t)nb2 * XBLK + 16 * 16 * sizeof(float) + (size_t)16 * XBLK;
    const size_t smem = smem1 > smem2 ? smem1 : smem2;
    long timeout_ticks = 200000000; // 2 s of the 100 MHz wall clock
    if (const char *t = getenv("LFAMD_FFN_TIMEOUT_S")) {
        const double sec = atof(t);
        if (sec > 0.0 && sec < 3600.0)
            timeout_ticks = (long)(sec * 1e8);
    }
    hipStream_t s = (hipStream_t)stream;
    auto go = [&](auto kernel) {
        if (smem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess)
                return e;
        }
        static const int dbg = getenv("LFAMD_FFN_DEBUG") ? atoi(getenv("LFAMD_FFN_DEBUG")) : 0;
        kernel<<<grid, 1024, smem, s>>>((const uint8_t *)d_x, (int)nb1, n_ht_gu, grid, (int)nb2, n_ht_d, G, U, d_residual, timeout_ticks, dbg, mgu, md);
        return hipGetLastError();
    };
    hipError_t e;
    const bool c14 = nb2 <= 56; // 14 waves x 4 super-blocks (k = 14336), else 15 x 4
    if (type_down == LFAMD_TYPE_Q4_K)
        e = c14 ? go(ffn_block_kernel<q4k_traits, q4k_traits, 14, 4>) : go(ffn_block_kernel<q4k_traits, q4k_traits, 15, 4>);
    else
        e = c14 ? go(ffn_block_kernel<q4k_traits, q6k_traits, 14, 4>) : go(ffn_block_kernel<q4k_traits, q6k_traits, 15, 4>);
    if (e != hipSuccess) {
        lfamd_set_error(hipGetErrorString(e));
        return LFAMD_ERR_HIP;
    }
    return LFAMD_OK;
}
