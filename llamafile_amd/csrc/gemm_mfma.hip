// gemm_mfma.hip — prefill GEMM: packed K-quant weights dequantised per wave straight into MFMA
// fragments, activations as exact-integer f16 tiles in LDS, v_mfma_f32_32x32x16_f16.
//
// Replaces the reference's mul_mat_q (ggml-cuda.cu.patch:13829-14130; dp4a, 32-wide warps, no MFMA
// on gfx950 — SURVEY.md F5) and follows the CPU path's arithmetic, mul_mat_qX_K_q8_K_T
// (iqk_mul_mat.inc:601-643):
//     C[n][i] = sum_b d8[n][b] * ( d[i][b] * sum_j sc[i][b][j] * <q[i][b][j], q8[n][b][j]>
//                                  - dmin[i][b] * sum_j mn[i][b][j] * bsum[n][b][j] )
// The integer parts are computed EXACTLY on the matrix cores: the weight operand is sc*q (<= 63*15,
// exact in f16), the activation operand is the int8 code (exact in f16), products and their sums over
// one 256-wide super-block stay below 2^24 in the f32 accumulator.  The two f32 scales are applied
// once per super-block on the VALU.  The mins term is one extra MFMA per super-block with the pair
// sums of bsums split into two exact f16 parts (pack.hip, prep_q8k_kernel).
//
// Orientation: MFMA A operand = activations (rows = tokens), B operand = weights (cols = weight
// rows), so a lane owns one weight row (its d/dmin are per-lane scalars) and the C store is 32
// consecutive floats per half-wave.
//
// Structure: work-group = 4 waves; wave w owns weight row-tile 4*blockIdx.x + w (32 rows) and all 64
// tokens of the block's token tile; weights go HBM -> VGPR -> fragment (no LDS, one super-block
// prefetched ahead), activations go global -> LDS directly (global_load_lds) into two XOR-swizzled
// tiles, one barrier per super-block.
#include "lfamd_device.h"

#define TOK_TILE 64
#define XT_ROW_BYTES 512 // 256 f16 codes of one token for one super-block

__device__ static inline half2_t as_half2(uint32_t u) {
    return __builtin_bit_cast(half2_t, u);
}

__device__ static inline half2_t pk_fma(half2_t a, half2_t b, half2_t c) {
    return __builtin_elementwise_fma(a, b, c);
}

__device__ static inline half2_t bcast_h2(float v) {
    _Float16 h = (_Float16)v;
    half2_t r = {h, h};
    return r;
}

union frag_u {
    half8_t v;
    half2_t p[4];
    uint4 u;
};

// One K-step (8 nibbles of this lane) of a Q4_K-family dword -> f16x8 of sc*q.
// S = (sc,sc), O = (-1024 sc), S16 = sc/16, O16 = -64 sc.
__device__ static inline half8_t dequant_q4(uint32_t x, half2_t S, half2_t O, half2_t S16, half2_t O16) {
    frag_u f;
    const uint32_t y = x >> 8;
    f.p[0] = pk_fma(as_half2((x & 0x000F000Fu) | 0x64006400u), S, O);
    f.p[1] = pk_fma(as_half2((x & 0x00F000F0u) | 0x64006400u), S16, O16);
    f.p[2] = pk_fma(as_half2((y & 0x000F000Fu) | 0x64006400u), S, O);
    f.p[3] = pk_fma(as_half2((y & 0x00F000F0u) | 0x64006400u), S16, O16);
    return f.v;
}

// Q6_K: codes are 6 bit (ql nibble | qh field), value sc*(code-32).  (code-32) is formed exactly,
// the product with the int8 scale is rounded to f16 (exact up to 2048; RNE to even above).
__device__ static inline half8_t dequant_q6(uint32_t x, uint32_t H, half2_t S) {
    frag_u f;
    const uint32_t y = x >> 8;
    const half2_t m1056 = {(_Float16)-1056.0f, (_Float16)-1056.0f};
    const half2_t m96 = {(_Float16)-96.0f, (_Float16)-96.0f};
    const half2_t r16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    half2_t c0 = as_half2((x & 0x000F000Fu) | (H & 0x00300030u) | 0x64006400u) + m1056;
    half2_t c1 = pk_fma(as_half2((x & 0x00F000F0u) | (H & 0x03000300u) | 0x64006400u), r16, m96);
    half2_t c2 = as_half2((y & 0x000F000Fu) | ((H >> 8) & 0x00300030u) | 0x64006400u) + m1056;
    half2_t c3 = pk_fma(as_half2((y & 0x00F000F0u) | ((H << 8) & 0x03000300u) | 0x64006400u), r16, m96);
    f.p[0] = c0 * S;
    f.p[1] = c1 * S;
    f.p[2] = c2 * S;
    f.p[3] = c3 * S;
    return f.v;
}

// per-super-block operands a wave keeps in registers
template <int TYPE>
struct wregs {
    uint4 qs[4];
    uint4 hd;    // Q4_K: {d, dmin, scales[12]};  Q6_K: 16 int8 scales
    uint4 qh[2]; // Q6_K only
    float dw;    // Q6_K only
    float4_t_ d8[2][4];
    half8_t xm[2]; // Q4_K only
};

template <int TYPE>
__global__ __launch_bounds__(256) void gemm_kq_kernel(const uint8_t *__restrict__ A, long m, int nb,
                                                      const _Float16 *__restrict__ Xh, const float *__restrict__ d8T,
                                                      const _Float16 *__restrict__ Xm, long n, long n_pad,
                                                      float *__restrict__ C, long ldc) {
    // two activation tiles (64 tokens x 256 codes, f16) so the global->LDS copy of super-block b+1 runs
    // under the MFMAs of super-block b
    __shared__ __attribute__((aligned(16))) uint8_t xt[2][TOK_TILE * XT_ROW_BYTES]; // 64 KiB
    constexpr int TILE = TYPE == LFAMD_TYPE_Q4_K ? P4K_TILE : P6K_TILE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const long n_row_tiles = (m + 31) / 32;
    const long rt = (long)blockIdx.x * 4 + wave;
    const bool active = rt < n_row_tiles;
    const long n0 = (long)blockIdx.y * TOK_TILE;
    const long k = (long)nb * 256;
    const uint8_t *tile0 = A + (size_t)(active ? rt : 0) * nb * TILE;

    // ---- issue everything super-block b needs: activations straight into LDS (global_load_lds, the XOR
    // swizzle applied on the SOURCE address since the LDS side is lane-linear), weights / scales to VGPRs
    auto prefetch = [&](int b, wregs<TYPE> &w) {
        uint8_t *dst = xt[b & 1];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int wi = wave * 8 + e;      // wave-instruction: rows 2wi, 2wi+1 of the tile (1 KiB)
            const int nn = 2 * wi + h, p = i; // this lane fills slot p of row nn with logical chunk p ^ (nn & 15)
            const uint8_t *src = (const uint8_t *)Xh + ((size_t)(n0 + nn) * k + (size_t)b * 256) * 2 + ((p ^ (nn & 15)) * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + wi * 1024), 16, 0, 0);
        }
        const uint8_t *tile = tile0 + (size_t)b * TILE;
#pragma unroll
        for (int g = 0; g < 4; g++)
            w.qs[g] = *(const uint4 *)(tile + g * 1024 + lane * 16);
        if constexpr (TYPE == LFAMD_TYPE_Q4_K) {
            w.hd = *(const uint4 *)(tile + P4K_HDR + i * 16);
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
                w.xm[nt] = *(const half8_t *)(Xm + ((size_t)(n0 + nt * 32 + i) * nb + b) * 16 + 8 * h);
        } else {
            w.qh[0] = *(const uint4 *)(tile + P6K_QH + 0 * 1024 + lane * 16);
            w.qh[1] = *(const uint4 *)(tile + P6K_QH + 1 * 1024 + lane * 16);
            w.hd = *(const uint4 *)(tile + P6K_SC + i * 16);
            w.dw = h2f(*(const uint16_t *)(tile + P6K_D + i * 2));
        }
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++)
                w.d8[nt][r4] = *(const float4_t_ *)(d8T + (size_t)b * n_pad + n0 + nt * 32 + 8 * r4 + 4 * h);
    };

    float16_t_ acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[nt][r] = 0.0f;

    auto compute = [&](int b, const wregs<TYPE> &w) {
        const uint8_t *xb = xt[b & 1];
        float16_t_ tmp[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                tmp[nt][r] = 0.0f;
        const uint32_t qw[16] = {w.qs[0].x, w.qs[0].y, w.qs[0].z, w.qs[0].w, w.qs[1].x, w.qs[1].y, w.qs[1].z, w.qs[1].w,
                                 w.qs[2].x, w.qs[2].y, w.qs[2].z, w.qs[2].w, w.qs[3].x, w.qs[3].y, w.qs[3].z, w.qs[3].w};
        if constexpr (TYPE == LFAMD_TYPE_Q4_K) {
            const uint4 hd = w.hd;
            const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
            uint32_t sc03, sc47, mn03, mn47;
            q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
#pragma unroll
            for (int j = 0; j < 8; j++) { // 32-wide sub-block = K-steps 2j, 2j+1
                const float scf = (float)(((j < 4 ? sc03 : sc47) >> (8 * (j & 3))) & 0xff);
                const half2_t S = bcast_h2(scf), O = bcast_h2(-1024.0f * scf);
                const half2_t S16 = bcast_h2(scf * 0.0625f), O16 = bcast_h2(-64.0f * scf);
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int t = 2 * j + e;
                    const half8_t wf = dequant_q4(qw[t], S, O, S16, O16);
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) {
                        const int c = 2 * t + h;
                        const half8_t xf = *(const half8_t *)(xb + (nt * 32 + i) * XT_ROW_BYTES + ((c ^ (i & 15)) * 16));
                        tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, wf, tmp[nt], 0, 0, 0);
                    }
                }
            }
            // ---- mins: one MFMA per token tile, K = 16 = {lo parts | hi parts} of the 8 pair sums
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
                float16_t_ tm;
#pragma unroll
                for (int r = 0; r < 16; r++)
                    tm[r] = 0.0f;
                tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.xm[nt], wm.v, tm, 0, 0, 0);
                // ---- per-super-block scaling: acc += d8[n] * (d * tmp - dmin * tm)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        const float u = fmaf(-dmin, tm[r], d * tmp[nt][r]);
                        acc[nt][r] = fmaf(u, w.d8[nt][r4][e], acc[nt][r]);
                    }
            }
        } else { // Q6_K: 16-wide sub-blocks, one per K-step; no mins
            const uint32_t hw[8] = {w.qh[0].x, w.qh[0].y, w.qh[0].z, w.qh[0].w, w.qh[1].x, w.qh[1].y, w.qh[1].z, w.qh[1].w};
            const uint32_t scw[4] = {w.hd.x, w.hd.y, w.hd.z, w.hd.w};
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const float scf = (float)(int)(int8_t)((scw[t >> 2] >> (8 * (t & 3))) & 0xff);
                const half2_t S = bcast_h2(scf);
                uint32_t H = hw[t >> 1];
                if (t & 1)
                    H >>= 2;
                const half8_t wf = dequant_q6(qw[t], H, S);
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    const int c = 2 * t + h;
                    const half8_t xf = *(const half8_t *)(xb + (nt * 32 + i) * XT_ROW_BYTES + ((c ^ (i & 15)) * 16));
                    tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, wf, tmp[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        acc[nt][r] = fmaf(w.dw * tmp[nt][r], w.d8[nt][r4][e], acc[nt][r]);
                    }
        }
    };

    // ---- software pipeline, two register sets (no copies): super-block b+1 is in flight while b computes.
    // __syncthreads() drains the LDS-DMA (vmcnt(0)) before the tile is read one iteration later.
    wregs<TYPE> wa, wb;
    prefetch(0, wa);
    __syncthreads();
    for (int b = 0; b < nb; b += 2) {
        if (b + 1 < nb)
            prefetch(b + 1, wb);
        compute(b, wa);
        __syncthreads();
        if (b + 1 < nb) {
            if (b + 2 < nb)
                prefetch(b + 2, wa);
            compute(b + 1, wb);
            __syncthreads();
        }
    }

    // ---- store: reg r of token tile nt is token n0 + 32nt + (r&3) + 8(r>>2) + 4h, weight row 32rt + i
    if (active) {
        const long row = rt * 32 + i;
        if (row < m) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const long tok = n0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (tok < n)
                        C[tok * ldc + row] = acc[nt][r];
                }
        }
    }
}

extern "C" hipError_t lfamd_launch_gemm_kq(int Atype, const void *A, long m, long k, const void *Xh, const void *d8T,
                                           const void *Xm, long n, long n_pad, float *C, long ldc, hipStream_t s) {
    if (m <= 0 || n <= 0)
        return hipSuccess;
    int nb = (int)(k / 256);
    long n_row_tiles = (m + 31) / 32;
    dim3 grid((unsigned)((n_row_tiles + 3) / 4), (unsigned)(n_pad / TOK_TILE));
    if (Atype == LFAMD_TYPE_Q4_K)
        gemm_kq_kernel<LFAMD_TYPE_Q4_K><<<grid, 256, 0, s>>>((const uint8_t *)A, m, nb, (const _Float16 *)Xh,
                                                               (const float *)d8T, (const _Float16 *)Xm, n, n_pad, C, ldc);
    else if (Atype == LFAMD_TYPE_Q6_K)
        gemm_kq_kernel<LFAMD_TYPE_Q6_K><<<grid, 256, 0, s>>>((const uint8_t *)A, m, nb, (const _Float16 *)Xh,
                                                               (const float *)d8T, (const _Float16 *)Xm, n, n_pad, C, ldc);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
