// gemm_q80.hip — Q8_0 x Q8_0 for batches (n > 8), bit-exact restatement of tinyBLAS_Q0_AVX2::gemm
// (tinyblas_cpu.h:934-971), register-tiled.
//
// The reference keeps, for every output (i, j), EIGHT f32 partial sums Cv[0..7] (the lanes of a __m256): lane jj
// accumulates, block after block,  Cv[jj] = fma(f32(dA)*f32(dB), f32(int dot of bytes 4jj..4jj+3), Cv[jj])
// (or the Kahan form `madder` on PRECISE tiles, with GCC's contraction fma(a, b, -e)), and the result is
// hsum(Cv) = ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)).  Every rounding depends on the block order, so K cannot be
// split or re-associated and the 4-byte integer dots cannot be merged into a 32-byte MFMA dot: this is VALU work
// (v_dot4_i32_i8 + cvt + mul + fma per block and lane), about 32 lane-ops per output and block.  What CAN be
// done is reuse: a GPU lane owns reference lane jj of a 4 x 8 patch of outputs (rows r, r+8, r+16, r+24 of a
// 32-row slab, 8 consecutive tokens), so one 16-byte weight load and one 16-byte activation read feed 32 and 16
// chain steps; the 32 independent chains per lane hide the f32 latency.
//
//   work-group = 4 waves on a 32-row x 32-token tile; wave w owns tokens 8w..8w+7, lane = (r = lane>>3, jj = lane&7)
//   weights    : P80 tiles straight from HBM/L2 into registers (lane's dword jj of four consecutive blocks of row
//                r per 16-byte load; the four waves of a work-group read the same tiles: L1 hits)
//   activations: the quad image written by prep_q80 (per token and quad: for each jj the four blocks' dwords side
//                by side, then the four f32 scales), staged through LDS in chunks of 16 quads
//   summation mode per output from the mnpack geometry (q0_is_kahan): wave-uniform fast paths for all-plain and
//   all-Kahan patches, per-output select otherwise.
#include "lfamd_device.h"

#define GQ_QUAD 144 // bytes per (token, quad) in the activation image
#define GQ_QD 128
#define GQ_CHUNK 16 // quads per LDS stage
#define GQ_ROWS 32
#define GQ_COLS 32

// ---------------------------------------------------------------------------------------------
// activation image.  One thread per (token, 32-block).
template <bool F32IN>
__global__ void prep_q80_kernel(const uint8_t *__restrict__ B, size_t b_row_bytes, long n, long n_pad, int nblocks, int nquads,
                                uint8_t *__restrict__ img) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = n_pad * (long)nquads * 4;
    if (idx >= total)
        return;
    const long tok = idx / (nquads * 4);
    const int l = (int)(idx - tok * (nquads * 4));
    uint8_t *dst = img + ((size_t)tok * nquads + (l >> 2)) * GQ_QUAD + (l & 3) * 4;
    uint32_t y[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float d = 0.0f;
    if (tok < n && l < nblocks) {
        if constexpr (F32IN) {
            // quantize_row_q8_0 (upstream ggml-quants.c, restated in quantize.hip): d = amax/127, id = 1/d, roundf
            const float4 *x = (const float4 *)(B + tok * b_row_bytes) + (size_t)l * 8;
            float v[32];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float4 f = x[e];
                v[4 * e] = f.x, v[4 * e + 1] = f.y, v[4 * e + 2] = f.z, v[4 * e + 3] = f.w;
            }
            float amax = 0.0f;
#pragma unroll
            for (int e = 0; e < 32; e++)
                amax = fmaxf(amax, fabsf(v[e]));
            const float dd = amax / 127.0f;
            const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
#pragma unroll
            for (int e = 0; e < 32; e++) {
                const int q = (int)roundf(v[e] * id);
                y[e >> 2] |= (uint32_t)(q & 0xff) << (8 * (e & 3));
            }
            d = h2f(f2h_bits(dd)); // the block stores d as f16
        } else {
            const uint8_t *blk = B + tok * b_row_bytes + (size_t)l * 34; // 34-byte blocks: 2-byte aligned
            d = h2f(*(const uint16_t *)blk);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint16_t *p = (const uint16_t *)(blk + 2 + 4 * e);
                y[e] = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++)
        *(uint32_t *)(dst + e * 16) = y[e];
    *(float *)(dst + GQ_QD) = d;
}

// ---------------------------------------------------------------------------------------------
template <int MODE> // 0: all plain, 1: all Kahan, 2: per-output select
__device__ static inline void gq_update(float &cv, float &ce, float a, float bq, bool kahan) {
    if constexpr (MODE == 0) {
        cv = __builtin_fmaf(a, bq, cv);
    } else if constexpr (MODE == 1) {
        const float y = __builtin_fmaf(a, bq, -ce);
        const float t = cv + y;
        ce = (t - cv) - y;
        cv = t;
    } else {
        const float plain = __builtin_fmaf(a, bq, cv);
        const float y = __builtin_fmaf(a, bq, -ce);
        const float t = cv + y;
        const float e2 = (t - cv) - y;
        cv = kahan ? t : plain;
        ce = kahan ? e2 : 0.0f;
    }
}

struct gq_w { // one quad of weights for the lane's four rows
    uint4 q[4];
    uint2 d[4];
};

template <int MODE>
__device__ static inline void gq_body(const lfamd_rsrc (&rA)[4], int nblocks, int nquads, const uint8_t *__restrict__ img,
                                      long col_base, uint8_t *lds, int lane, int wave, uint32_t kmask, float (&cv)[4][8]) {
    const int r = lane >> 3, jj = lane & 7;
    float ce[4][8];
#pragma unroll
    for (int ri = 0; ri < 4; ri++)
#pragma unroll
        for (int c = 0; c < 8; c++)
            cv[ri][c] = ce[ri][c] = 0.0f;

    auto loadw = [&](gq_w &w, int L) { // past the row group / row: zero records, no traffic
#pragma unroll
        for (int ri = 0; ri < 4; ri++) {
            w.q[ri] = buf_ld16(rA[ri], (uint32_t)L * P80_TILE + lane * 16);
            w.d[ri] = buf_ld8(rA[ri], (uint32_t)L * P80_TILE + P80_D + r * 8);
        }
    };
    // blocks [0, nb_run) of the quad run; padding blocks must not (a Kahan step with a*b = 0 still folds the
    // pending compensation into the sum)
    auto quad = [&](const gq_w &w, const uint8_t *xl, int nb_run) {
        uint4 xq[8];
        float4 xd[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            xq[c] = *(const uint4 *)(xl + c * (GQ_CHUNK * GQ_QUAD) + jj * 16);
            xd[c] = *(const float4 *)(xl + c * (GQ_CHUNK * GQ_QUAD) + GQ_QD);
        }
#pragma unroll
        for (int dd = 0; dd < 4; dd++) {
            if (dd < nb_run) {
#pragma unroll
                for (int ri = 0; ri < 4; ri++) {
                    const uint32_t wq = dd == 0 ? w.q[ri].x : dd == 1 ? w.q[ri].y : dd == 2 ? w.q[ri].z : w.q[ri].w;
                    const uint32_t dw = dd < 2 ? w.d[ri].x : w.d[ri].y;
                    const float da = h2f((uint16_t)((dd & 1) ? (dw >> 16) : (dw & 0xffff)));
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const uint32_t xqd = dd == 0 ? xq[c].x : dd == 1 ? xq[c].y : dd == 2 ? xq[c].z : xq[c].w;
                        const float xdd = dd == 0 ? xd[c].x : dd == 1 ? xd[c].y : dd == 2 ? xd[c].z : xd[c].w;
                        const float a = da * xdd;
                        const float bq = (float)sdot4(wq, xqd, 0);
                        gq_update<MODE>(cv[ri][c], ce[ri][c], a, bq, (kmask >> (ri * 8 + c)) & 1);
                    }
                }
            }
        }
    };

    const int nq_full = nblocks >> 2;
    gq_w wa, wb;
    loadw(wa, 0);
    for (int L0 = 0; L0 < nquads; L0 += GQ_CHUNK) {
        // ---- stage this chunk of the activation image: 32 tokens x up to 16 quads, token-major in LDS
        const int qn = min(GQ_CHUNK, nquads - L0);
        __syncthreads(); // everybody is done reading the previous chunk
        for (int v = threadIdx.x; v < GQ_COLS * qn * 9; v += blockDim.x) {
            const int tok = v / (qn * 9), rem = v - tok * (qn * 9);
            const uint4 val = *(const uint4 *)(img + ((size_t)(col_base + tok) * nquads + L0) * GQ_QUAD + rem * 16);
            *(uint4 *)(lds + (size_t)tok * (GQ_CHUNK * GQ_QUAD) + rem * 16) = val;
        }
        __syncthreads();
        const uint8_t *xw = lds + (size_t)(wave * 8) * (GQ_CHUNK * GQ_QUAD);
        for (int s = 0; s < qn; s += 2) { // ping-pong weight registers, next quad in flight
            const int L = L0 + s;
            loadw(wb, L + 1);
            quad(wa, xw + s * GQ_QUAD, L < nq_full ? 4 : nblocks - 4 * L);
            loadw(wa, L + 2);
            if (s + 1 < qn)
                quad(wb, xw + (s + 1) * GQ_QUAD, L + 1 < nq_full ? 4 : nblocks - 4 * (L + 1));
        }
    }
}

__global__ __launch_bounds__(256) void gemm_q80_kernel(const uint8_t *__restrict__ A, long m, long n, int nblocks, int nquads,
                                                       const uint8_t *__restrict__ img, float *__restrict__ C, long ldc,
                                                       int vregs32, int precise) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane >> 3, jj = lane & 7;
    const long n_rg = (m + 7) / 8;
    const long rg0 = (long)blockIdx.x * 4;
    const long col_base = (long)blockIdx.y * GQ_COLS;
    const uint32_t rg_bytes = (uint32_t)nquads * P80_TILE;
    lfamd_rsrc rA[4];
#pragma unroll
    for (int ri = 0; ri < 4; ri++) {
        const long rg = rg0 + ri;
        rA[ri] = make_rsrc(A + (size_t)(rg < n_rg ? rg : 0) * rg_bytes, rg < n_rg ? rg_bytes : 0u);
    }
    // summation mode of this lane's 32 outputs (tinyblas_cpu.h:794-931 via q0_is_kahan); out-of-range outputs
    // are never stored: give them the mode of the last valid one so they do not force the mixed path
    uint32_t kmask = 0;
#pragma unroll
    for (int ri = 0; ri < 4; ri++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
            long row = (rg0 + ri) * 8 + r, col = col_base + wave * 8 + c;
            row = row < m ? row : m - 1;
            col = col < n ? col : n - 1;
            if (q0_is_kahan(row, col, m, n, vregs32 != 0, precise != 0))
                kmask |= 1u << (ri * 8 + c);
        }
    const bool all_plain = __builtin_amdgcn_ballot_w64(kmask != 0) == 0;
    const bool all_kahan = __builtin_amdgcn_ballot_w64(kmask != 0xffffffffu) == 0;

    float cv[4][8];
    // the three bodies contain __syncthreads(): every wave of the work-group runs one of them the same number of times
    if (all_plain)
        gq_body<0>(rA, nblocks, nquads, img, col_base, lds, lane, wave, kmask, cv);
    else if (all_kahan)
        gq_body<1>(rA, nblocks, nquads, img, col_base, lds, lane, wave, kmask, cv);
    else
        gq_body<2>(rA, nblocks, nquads, img, col_base, lds, lane, wave, kmask, cv);

    // hsum(__m256), tinyblas_cpu.h:277-296: ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7))
#pragma unroll
    for (int ri = 0; ri < 4; ri++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
            float v = cv[ri][c];
            v = v + __shfl_xor(v, 4, 64);
            v = v + __shfl_xor(v, 2, 64);
            v = v + __shfl_xor(v, 1, 64);
            const long row = (rg0 + ri) * 8 + r, col = col_base + wave * 8 + c;
            if (jj == 0 && row < m && col < n)
                C[col * ldc + row] = v;
        }
}

// ---------------------------------------------------------------------------------------------
extern "C" size_t lfamd_gemm_q80_workspace(long k, long n) {
    const size_t n_pad = ((size_t)n + GQ_COLS - 1) / GQ_COLS * GQ_COLS;
    const size_t nquads = ((size_t)(k / 32) + 3) / 4;
    return n_pad * nquads * GQ_QUAD;
}

// Btype: LFAMD_TYPE_F32 (quantised here like quantize_row_q8_0) or LFAMD_TYPE_Q8_0 blocks
extern "C" hipError_t lfamd_launch_gemm_q80(const void *A, long m, long k, int Btype, const void *B, size_t b_row_bytes, long n,
                                            float *C, long ldc, void *ws, int vregs32, int precise, hipStream_t s) {
    if (m <= 0 || n <= 0)
        return hipSuccess;
    const int nblocks = (int)(k / 32), nquads = (nblocks + 3) / 4;
    const long n_pad = (n + GQ_COLS - 1) / GQ_COLS * GQ_COLS;
    const long total = n_pad * (long)nquads * 4;
    if (Btype == LFAMD_TYPE_F32)
        prep_q80_kernel<true><<<(unsigned)((total + 127) / 128), 128, 0, s>>>((const uint8_t *)B, b_row_bytes, n, n_pad, nblocks,
                                                                             nquads, (uint8_t *)ws);
    else
        prep_q80_kernel<false><<<(unsigned)((total + 127) / 128), 128, 0, s>>>((const uint8_t *)B, b_row_bytes, n, n_pad, nblocks,
                                                                              nquads, (uint8_t *)ws);
    const size_t smem = (size_t)GQ_COLS * GQ_CHUNK * GQ_QUAD; // 72 KiB
    hipError_t e = hipFuncSetAttribute((const void *)gemm_q80_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess)
        return e;
    const long n_rg = (m + 7) / 8;
    dim3 grid((unsigned)((n_rg + 3) / 4), (unsigned)(n_pad / GQ_COLS));
    gemm_q80_kernel<<<grid, 256, smem, s>>>((const uint8_t *)A, m, n, nblocks, nquads, (const uint8_t *)ws, C, ldc, vregs32,
                                            precise);
    return hipGetLastError();
}
