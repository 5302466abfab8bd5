// gemm_sb.hip — SMALL BATCHES (a handful of tokens: speculative decoding, a few sequences decoding together) of Q4_K / Q5_K /
// Q6_K on the matrix cores.  Replaces, for n >= SB_MIN tokens, the multi-column VALU GEMVs (whose integer dots grow with n:
// 3.6x the n = 1 time at n = 8) and, up to 32 tokens, the detour through the prefill GEMM (128-token tiles, 15-45 us).
//
// Reference behaviour: ggml-cuda.cu.patch:14359, 14506-14575, 18384-18386 (mul_mat_vec_q up to 8 columns, mul_mat_q above);
// numerics of the exact-code bodies (gemm_wide_impl.h): integer codes are exact f16 numbers, sc * q is exact in f16 (Q4_K /
// Q5_K; Q6_K: the int8 scale is split in two 4-bit halves so that it is as well), a super-block's sum is exact in the f32 accumulator, then
// acc += (d * sum - dmin * mins) * d8[token] in f32: <= 2e-6 of the oracle.
//
// Shape of the work: the weights are streamed ONCE, like the GEMV; the activations (n x k codes, a few dozen KiB) come out of
// L2.  One MFMA 32x32x16 covers a 32-row tile x 16 weights x 32 token slots, so every token count up to 32 costs the same.
//   launch 1  sb_prep_kernel : one wave per (token slot, super-block): quantize_row_q8_K arithmetic (or the given Q8_K blocks)
//                              -> Xh [nb][32][256] f16 codes, d8T [nb][32] f32, Xm [nb][32][16] f16 (pair sums split 64 hi + lo);
//                              the same launch zeroes the n x m result.
//   launch 2  gemm_sb_kernel : work-group = (32-row tile, K HALF); wave w takes super-blocks b0 + w, b0 + w + NW, ...: five
//                              16-byte weight loads per lane straight from HBM (the P4K / P5K / P6K lattice IS the MFMA fragment
//                              order), the token fragments 16 bytes per lane from L2, dequantisation in registers, 16 + 1 MFMAs,
//                              the f32 scaling per super-block; the waves' partial tiles are summed through LDS in a fixed order
//                              and ADDED to the result with one float atomic per element.  Exactly two work-groups add to an
//                              element of the zeroed result, and a + b == b + a: the outcome does not depend on their order
//                              (bit-identical from run to run), while 4096 x 4096 offers 256 work-groups instead of 128.
#include "gemm_common.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#define SB_COLS 32 // token slots (one MFMA tile)

// ---------------------------------------------------------------------------------------------------------------------
// activation staging + zeroing of the result
// I8: the codes stay int8 and are written in the byte order of the int8 MFMA's weight operand (gemm_sb16i_kernel): sub-block
// jb, K half h -> 16 bytes = K-step 2 jb: elements (0,4,1,5,2,6,3,7) of k = 32 jb + 8 h + j, then K-step 2 jb + 1 the same.
template <bool F32IN, bool MINS, bool I8 = false>
__global__ __launch_bounds__(64) void sb_prep_kernel(const uint8_t *__restrict__ B, size_t b_row_bytes, int n, int nb,
                                                      _Float16 *__restrict__ Xh, float *__restrict__ d8T, _Float16 *__restrict__ Xm,
                                                      float *__restrict__ C, long m, long ldc) {
    const int blk = blockIdx.x, tok = blk / nb, b = blk - tok * nb, t = threadIdx.x;
    if (C) { // the result: n rows of m floats, spread over the grid (null: the consumer stores whole rows itself)
        const long total = (long)n * m, stride = (long)gridDim.x * 64;
        for (long e = (long)blk * 64 + t; e < total; e += stride) {
            const long r = e / m;
            C[r * ldc + (e - r * m)] = 0.0f;
        }
    }
    if (!B)
        return; // (zero only: the staged codes of an earlier matrix of the same launch group are reused)
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    _Float16 *xo = Xh + ((size_t)b * SB_COLS + tok) * 256;
    _Float16 *mo = Xm + ((size_t)b * SB_COLS + tok) * 16;
    int q[4];
    float d;
    if constexpr (F32IN) {
        // quantize_row_q8_K: the FIRST element of largest magnitude gives the sign of iscale = -128 / max; nearest-even codes
        // clamped at 127; d = 1 / iscale  (pack.hip: prep_f32_kernel, the same arithmetic)
        const float4 f = *(const float4 *)((const float *)(B + (size_t)tok * b_row_bytes) + (size_t)b * 256 + 4 * t);
        const float v[4] = {f.x, f.y, f.z, f.w};
        // block maximum by DPP (no LDS round trips), then the FIRST lane / element that reaches it (gemv_impl.h: stage_f32_q8k_wave)
        const float a0 = fabsf(v[0]), a1 = fabsf(v[1]), a2 = fabsf(v[2]), a3 = fabsf(v[3]);
        float am = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
        am = fmaxf(am, dpp_f32<DPP_XOR1>(am));
        am = fmaxf(am, dpp_f32<DPP_XOR2>(am));
        am = fmaxf(am, dpp_f32<DPP_HALF_MIRROR>(am));
        am = fmaxf(am, dpp_f32<DPP_MIRROR>(am));
        const float amax = fmaxf(fmaxf(readlane_f32(am, 0), readlane_f32(am, 16)), fmaxf(readlane_f32(am, 32), readlane_f32(am, 48)));
        const bool m0 = a0 == amax, m1 = a1 == amax, m2 = a2 == amax, m3 = a3 == amax;
        const unsigned long long ball = __builtin_amdgcn_ballot_w64(m0 || m1 || m2 || m3);
        const float cand = m0 ? v[0] : (m1 ? v[1] : (m2 ? v[2] : v[3]));
        const float val = readlane_f32(cand, ball ? __builtin_ctzll(ball) : 0);
        q[0] = q[1] = q[2] = q[3] = 0;
        d = 0.0f;
        if (amax != 0.0f) {
            const float iscale = -128.0f / val;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = (int)rintf(iscale * v[e]);
                q[e] = c > 127 ? 127 : c;
            }
            d = 1.0f / iscale;
        }
    } else {
        const lfamd_block_q8_K *y = (const lfamd_block_q8_K *)(B + (size_t)tok * b_row_bytes) + b;
        const uint32_t w = *(const uint32_t *)((const uint8_t *)y->qs + 4 * t); // 292-byte blocks are 4-aligned
#pragma unroll
        for (int e = 0; e < 4; e++)
            q[e] = (int)(int8_t)(w >> (8 * e));
        d = y->d;
    }
    if constexpr (I8) {
        // thread t holds k = 4 t + c: sub-block t >> 3, K-step half (t >> 2) & 1, K half (t >> 1) & 1, element j = 4 (t & 1) + c
        int8_t *xq = (int8_t *)Xh + ((size_t)b * SB_COLS + tok) * 256 + (t >> 3) * 32 + ((t >> 1) & 1) * 16 + ((t >> 2) & 1) * 8 + (t & 1);
#pragma unroll
        for (int c = 0; c < 4; c++)
            xq[2 * c] = (int8_t)q[c];
    } else {
        const half4_t h4 = {(_Float16)(float)q[0], (_Float16)(float)q[1], (_Float16)(float)q[2], (_Float16)(float)q[3]};
        *(half4_t *)(xo + 4 * t) = h4;
    }
    if (t == 0)
        d8T[(size_t)b * SB_COLS + tok] = d;
    if constexpr (MINS) { // pair sum j = codes 32j .. 32j+31 = lanes 8j .. 8j+7; |S| <= 4096: S = 64 hi + lo, lo in [0, 63]
        int S = q[0] + q[1] + q[2] + q[3];
        S += (int)dpp_u32<DPP_XOR1>((uint32_t)S);
        S += (int)dpp_u32<DPP_XOR2>((uint32_t)S);
        S += (int)dpp_u32<DPP_HALF_MIRROR>((uint32_t)S); // lanes 8j .. 8j+7
        if ((t & 7) == 0) {
            const int lo = S & 63, hi = (S - lo) / 64;
            mo[t >> 3] = (_Float16)(float)lo;
            mo[8 + (t >> 3)] = (_Float16)(float)hi;
        }
    }
}

// Q6_K: the K-step's eight codes minus 32 as f16 (exact), no scale yet (cf. dequant_q6, gemm_common.h)
__device__ static inline void q6_codes(uint32_t x, uint32_t H, half2_t (&c)[4]) {
    const uint32_t y = x >> 8;
    const half2_t m1056 = {(_Float16)-1056.0f, (_Float16)-1056.0f};
    const half2_t m96 = {(_Float16)-96.0f, (_Float16)-96.0f};
    const half2_t r16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    c[0] = as_half2((x & 0x000F000Fu) | (H & 0x00300030u) | 0x64006400u) + m1056;
    c[1] = pk_fma(as_half2((x & 0x00F000F0u) | (H & 0x03000300u) | 0x64006400u), r16, m96);
    c[2] = as_half2((y & 0x000F000Fu) | ((H >> 8) & 0x00300030u) | 0x64006400u) + m1056;
    c[3] = pk_fma(as_half2((y & 0x00F000F0u) | ((H << 8) & 0x03000300u) | 0x64006400u), r16, m96);
}

// ---------------------------------------------------------------------------------------------------------------------
// Work-group = (K half kh, tiles wg, wg + n_wg, ...); wave w owns super-blocks b0 + w, b0 + w + NW, ... of every tile.  ROLLING
// PREFETCH: a unit's registers (four nibble groups, the header, sixteen token fragments, the token scales) are refilled for the
// wave's NEXT unit as soon as their last use in the current one has been issued — one register set, every load a whole unit
// ahead, also across the reduction at the end of a tile.  No inline asm here: hipcc keeps the counts.
template <int TYPE, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_sb_kernel(const uint8_t *__restrict__ A, long m, int nb, const _Float16 *__restrict__ Xh,
                                                          const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, int n,
                                                          float *__restrict__ C, long ldc, int n_rt) {
    constexpr bool Q5 = TYPE == LFAMD_TYPE_Q5_K, Q6 = TYPE == LFAMD_TYPE_Q6_K;
    constexpr int TILE = Q5 ? P5K_TILE : Q6 ? P6K_TILE : P4K_TILE;
    __shared__ float red[NW][16][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 31, h = lane >> 5;
    const int kh = blockIdx.x & 1, wg = blockIdx.x >> 1, n_wg = gridDim.x >> 1;
    const int half = (nb + 1) >> 1;
    const int b0 = kh ? half : 0, b1 = kh ? nb : half;
    const int first = b0 + wave;
    const int nmine = first < b1 ? (b1 - first + NW - 1) / NW : 0; // this wave's super-blocks in every tile
    // token slot of this lane as the MFMA's A row; slots past n are never stored: they re-read slot 0 (same cache lines, no mask)
    const int slot = i < n ? i : 0;
    const uint32_t xoff = (uint32_t)(slot * 256 + 8 * h), xmoff = (uint32_t)(slot * 16 + 8 * h);

    const float16_t_ zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t magic = opaque_magic();

    // ---- the register set of one unit
    u32x4_t qs[4], hd, qh0 = {0, 0, 0, 0}, qh1 = {0, 0, 0, 0};
    uint16_t dw16 = 0;
    half8_t F[16], xm = {0, 0, 0, 0, 0, 0, 0, 0};
    float4_t_ d8[4];
    (void)qh0, (void)qh1, (void)dw16, (void)xm;
    // cursor of the unit being LOADED (one ahead of the one being computed); past the end it stays on the last unit
    int rt_n = wg, b_n = first;
    const uint8_t *tile_n = A;
    const _Float16 *xr_n = Xh, *xm_n = Xm;
    const float *d8_n = d8T;
    auto point = [&]() {
        tile_n = A + ((size_t)rt_n * nb + b_n) * TILE;
        xr_n = Xh + (size_t)b_n * SB_COLS * 256 + xoff;
        xm_n = Xm + (size_t)b_n * SB_COLS * 16 + xmoff;
        d8_n = d8T + (size_t)b_n * SB_COLS + 4 * h;
    };
    auto advance = [&]() {
        int b2 = b_n + NW, rt2 = rt_n;
        if (b2 >= b1)
            b2 = first, rt2 = rt_n + n_wg;
        if (rt2 < n_rt) // (the last unit reloads itself: valid addresses, dead values)
            b_n = b2, rt_n = rt2;
        point();
    };
    auto ld_qs = [&](int g) { qs[g] = __builtin_nontemporal_load((const u32x4_t *)(tile_n + g * 1024 + lane * 16)); };
    auto ld_hd = [&]() { hd = __builtin_nontemporal_load((const u32x4_t *)(tile_n + (Q6 ? P6K_SC : P4K_HDR) + i * 16)); };
    auto ld_F = [&](int t) { F[t] = *(const half8_t *)(xr_n + 16 * t); };
    auto ld_d8 = [&]() {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++)
            d8[r4] = *(const float4_t_ *)(d8_n + 8 * r4); // reg r of the tile is token slot 8 (r >> 2) + 4 h + (r & 3)
    };
    if (nmine > 0 && wg < n_rt) {
        point();
#pragma unroll
        for (int g = 0; g < 4; g++)
            ld_qs(g);
        ld_hd();
        if constexpr (Q5)
            qh0 = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P5K_QH + lane * 16));
        if constexpr (Q6) {
            qh0 = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P6K_QH + lane * 16));
            qh1 = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P6K_QH + 1024 + lane * 16));
            dw16 = *(const uint16_t *)(tile_n + P6K_D + i * 2);
        } else {
            xm = *(const half8_t *)xm_n;
        }
#pragma unroll
        for (int t = 0; t < 16; t++)
            ld_F(t);
        ld_d8();
    }

    for (int rt = wg; rt < n_rt; rt += n_wg) {
        float16_t_ acc = zero16;
        for (int j = 0; j < nmine; j++) {
            advance(); // tile_n .. d8_n: the unit after this one
            float16_t_ tmp = zero16;
            if constexpr (!Q6) {
                const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
                uint32_t sc03, sc47, mn03, mn47;
                q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
                ld_hd();
                uint32_t hq5[4] = {qh0.x, qh0.y, qh0.z, qh0.w};
                (void)hq5;
                if constexpr (Q5)
                    qh0 = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P5K_QH + lane * 16));
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t qw[4] = {qs[g].x, qs[g].y, qs[g].z, qs[g].w};
                    ld_qs(g);
#pragma unroll
                    for (int dd = 0; dd < 4; dd++) {
                        const int t = 4 * g + dd, jb = t >> 1; // sub-block jb
                        const q4_consts2 cp = q4_consts_pair(jb < 4 ? sc03 : sc47, (jb & 2) ? 2 : 0);
                        const int hsel = jb & 1;
                        const half2_t S = {cp.S[hsel], cp.S[hsel]}, O = {cp.O[hsel], cp.O[hsel]};
                        const half2_t S16 = {cp.S16[hsel], cp.S16[hsel]}, O16 = {cp.O16[hsel], cp.O16[hsel]};
                        half8_t wf;
                        if constexpr (Q5)
                            wf = dequant_q5(qw[dd], hq5[g] >> dd, S, O, S16, O16, magic);
                        else
                            wf = dequant_q4(qw[dd], S, O, S16, O16, magic);
                        tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t], wf, tmp, 0, 0, 0);
                        ld_F(t);
                    }
                }
                // mins: one MFMA, K = 16 = {lo parts | hi parts} of the eight pair sums against {m_j | 64 m_j}
                frag_u wm;
                const float mscale = h ? 64.0f : 1.0f;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t mw = p < 2 ? mn03 : mn47;
                    const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
                    const half2_t v = {(_Float16)(m0 * mscale), (_Float16)(m1 * mscale)};
                    wm.p[p] = v;
                }
                const float16_t_ tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wm.v, zero16, 0, 0, 0);
                xm = *(const half8_t *)xm_n;
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        const float u = fmaf(-dmin, tm[r], d * tmp[r]);
                        acc[r] = fmaf(u, d8[r4][e], acc[r]);
                    }
                ld_d8();
            } else {
                const uint32_t scw[4] = {hd.x, hd.y, hd.z, hd.w};
                ld_hd();
                const float dw = h2f(dw16);
                dw16 = *(const uint16_t *)(tile_n + P6K_D + i * 2);
                // sc * (code - 32) reaches 128 * 32 = 4096: not an f16 integer above 2048.  The int8 scale is split sc = 16 hi + lo
                // (lo in [0, 15], hi in [-8, 7]): both products are exact (<= 480), each half sums exactly in its own f32 tile
                // (<= 1.6e7 < 2^24), and 16 * hi-tile + lo-tile is formed once per super-block.
                float16_t_ thi = zero16;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t qw[4] = {qs[g].x, qs[g].y, qs[g].z, qs[g].w};
                    ld_qs(g);
                    const u32x4_t qh = g < 2 ? qh0 : qh1;
                    const uint32_t hw[4] = {qh.x, qh.y, qh.z, qh.w};
                    if (g == 1)
                        qh0 = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P6K_QH + lane * 16));
                    if (g == 3)
                        qh1 = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P6K_QH + 1024 + lane * 16));
#pragma unroll
                    for (int dd = 0; dd < 4; dd++) {
                        const int t = 4 * g + dd;
                        const int sc = (int)(int8_t)((scw[g] >> (8 * dd)) & 0xff);
                        const int lo = sc & 15, hi = (sc - lo) >> 4;
                        const half2_t Slo = bcast_h2((float)lo), Shi = bcast_h2((float)hi);
                        uint32_t H = hw[(t >> 1) & 3];
                        if (t & 1)
                            H >>= 2;
                        half2_t c[4];
                        q6_codes(qw[dd], H, c);
                        frag_u flo, fhi;
#pragma unroll
                        for (int p = 0; p < 4; p++)
                            flo.p[p] = c[p] * Slo, fhi.p[p] = c[p] * Shi;
                        tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t], flo.v, tmp, 0, 0, 0);
                        thi = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t], fhi.v, thi, 0, 0, 0);
                        ld_F(t);
                    }
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        acc[r] = fmaf(dw * fmaf(16.0f, thi[r], tmp[r]), d8[r4][e], acc[r]);
                    }
                ld_d8();
            }
        }
        // ---- the waves' partial tiles, summed in wave order; reg r = token slot 8 (r >> 2) + 4 h + (r & 3), lane's row 32 rt + i
#pragma unroll
        for (int r = 0; r < 16; r++)
            red[wave][r][lane] = acc[r];
        __syncthreads();
        for (int v = threadIdx.x; v < 16 * 64; v += NW * 64) {
            const int r = v >> 6, l = v & 63;
            const int tok = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
            const long row = (long)rt * 32 + (l & 31);
            if (tok < n && row < m) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; w++)
                    s += red[w][r][l];
                unsafeAtomicAdd(C + (long)tok * ldc + row, s);
            }
        }
        __syncthreads(); // (the next tile's partial tiles overwrite `red`)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------------
// SHALLOW rows (at most NW super-blocks per K half: k <= 4096 with NW = 8): a wave's super-block is the same in every tile, so its
// token fragments, pair sums and token scales are loaded ONCE and only the weights stream — two register sets, the second tile's
// weights in flight while the first is multiplied (two units per wave in flight: what the weight stream needs to stay near the
// HBM rate with one 8-wave work-group per CU).
template <int TYPE>
struct sb_wset {
    u32x4_t qs[4], hd, qh0, qh1;
    uint32_t dw;
};

template <int TYPE>
__device__ __forceinline__ void sb_load_set(sb_wset<TYPE> &w, const uint8_t *tile, int lane, int i) {
    constexpr bool Q5 = TYPE == LFAMD_TYPE_Q5_K, Q6 = TYPE == LFAMD_TYPE_Q6_K;
#pragma unroll
    for (int g = 0; g < 4; g++)
        w.qs[g] = __builtin_nontemporal_load((const u32x4_t *)(tile + g * 1024 + lane * 16));
    w.hd = __builtin_nontemporal_load((const u32x4_t *)(tile + (Q6 ? P6K_SC : P4K_HDR) + i * 16));
    if constexpr (Q5)
        w.qh0 = __builtin_nontemporal_load((const u32x4_t *)(tile + P5K_QH + lane * 16));
    if constexpr (Q6) {
        w.qh0 = __builtin_nontemporal_load((const u32x4_t *)(tile + P6K_QH + lane * 16));
        w.qh1 = __builtin_nontemporal_load((const u32x4_t *)(tile + P6K_QH + 1024 + lane * 16));
        w.dw = *(const uint16_t *)(tile + P6K_D + i * 2);
    }
}

// one (tile, super-block) unit: returns the tile's contribution (the arithmetic of gemm_sb_kernel's loop body)
template <int TYPE>
__device__ __forceinline__ float16_t_ sb_unit(const sb_wset<TYPE> &w, const half8_t (&F)[16], const half8_t xm, const float4_t_ (&d8)[4], int h,
                                              uint32_t magic) {
    constexpr bool Q5 = TYPE == LFAMD_TYPE_Q5_K, Q6 = TYPE == LFAMD_TYPE_Q6_K;
    const float16_t_ zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t qw[16] = {w.qs[0].x, w.qs[0].y, w.qs[0].z, w.qs[0].w, w.qs[1].x, w.qs[1].y, w.qs[1].z, w.qs[1].w,
                             w.qs[2].x, w.qs[2].y, w.qs[2].z, w.qs[2].w, w.qs[3].x, w.qs[3].y, w.qs[3].z, w.qs[3].w};
    float16_t_ tmp = zero16, out;
    if constexpr (!Q6) {
        const uint32_t hq5[4] = {w.qh0.x, w.qh0.y, w.qh0.z, w.qh0.w};
        (void)hq5;
        const float d = h2f((uint16_t)(w.hd.x & 0xffff)), dmin = h2f((uint16_t)(w.hd.x >> 16));
        uint32_t sc03, sc47, mn03, mn47;
        q4k_scales_bytes(w.hd.y, w.hd.z, w.hd.w, sc03, sc47, mn03, mn47);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const q4_consts2 cp = q4_consts_pair(j < 4 ? sc03 : sc47, (j & 2) ? 2 : 0);
            const int hsel = j & 1;
            const half2_t S = {cp.S[hsel], cp.S[hsel]}, O = {cp.O[hsel], cp.O[hsel]};
            const half2_t S16 = {cp.S16[hsel], cp.S16[hsel]}, O16 = {cp.O16[hsel], cp.O16[hsel]};
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int t = 2 * j + e;
                half8_t wf;
                if constexpr (Q5)
                    wf = dequant_q5(qw[t], hq5[t >> 2] >> (t & 3), S, O, S16, O16, magic);
                else
                    wf = dequant_q4(qw[t], S, O, S16, O16, magic);
                tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t], wf, tmp, 0, 0, 0);
            }
        }
        frag_u wm;
        const float mscale = h ? 64.0f : 1.0f;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t mw = p < 2 ? mn03 : mn47;
            const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
            const half2_t v = {(_Float16)(m0 * mscale), (_Float16)(m1 * mscale)};
            wm.p[p] = v;
        }
        const float16_t_ tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wm.v, zero16, 0, 0, 0);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int r = 4 * r4 + e;
                out[r] = fmaf(-dmin, tm[r], d * tmp[r]) * d8[r4][e];
            }
    } else {
        const uint32_t hw[8] = {w.qh0.x, w.qh0.y, w.qh0.z, w.qh0.w, w.qh1.x, w.qh1.y, w.qh1.z, w.qh1.w};
        const uint32_t scw[4] = {w.hd.x, w.hd.y, w.hd.z, w.hd.w};
        const float dw = h2f((uint16_t)w.dw);
        float16_t_ thi = zero16; // (the int8 scale in two 4-bit halves: see gemm_sb_kernel)
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int sc = (int)(int8_t)((scw[t >> 2] >> (8 * (t & 3))) & 0xff);
            const int lo = sc & 15, hi = (sc - lo) >> 4;
            const half2_t Slo = bcast_h2((float)lo), Shi = bcast_h2((float)hi);
            uint32_t H = hw[t >> 1];
            if (t & 1)
                H >>= 2;
            half2_t c[4];
            q6_codes(qw[t], H, c);
            frag_u flo, fhi;
#pragma unroll
            for (int p = 0; p < 4; p++)
                flo.p[p] = c[p] * Slo, fhi.p[p] = c[p] * Shi;
            tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t], flo.v, tmp, 0, 0, 0);
            thi = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[t], fhi.v, thi, 0, 0, 0);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int r = 4 * r4 + e;
                out[r] = (dw * fmaf(16.0f, thi[r], tmp[r])) * d8[r4][e];
            }
    }
    return out;
}

template <int TYPE, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_sb_shallow_kernel(const uint8_t *__restrict__ A, long m, int nb, const _Float16 *__restrict__ Xh,
                                                                  const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, int n,
                                                                  float *__restrict__ C, long ldc, int n_rt) {
    constexpr bool Q5 = TYPE == LFAMD_TYPE_Q5_K, Q6 = TYPE == LFAMD_TYPE_Q6_K;
    constexpr int TILE = Q5 ? P5K_TILE : Q6 ? P6K_TILE : P4K_TILE;
    __shared__ float red[NW][16][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 31, h = lane >> 5;
    const int kh = blockIdx.x & 1, wg = blockIdx.x >> 1, n_wg = gridDim.x >> 1;
    const int half = (nb + 1) >> 1;
    const int b0 = kh ? half : 0, b1 = kh ? nb : half;
    const int b = b0 + wave; // at most NW super-blocks per half: one per wave
    const bool mine = b < b1;
    const int bc = mine ? b : b0; // (a wave without a super-block loads valid bytes and contributes zeros)
    const int slot = i < n ? i : 0;
    const uint32_t magic = opaque_magic();

    half8_t F[16], xm = {0, 0, 0, 0, 0, 0, 0, 0};
    float4_t_ d8[4];
    {
        const _Float16 *xr = Xh + ((size_t)bc * SB_COLS + slot) * 256 + 8 * h;
#pragma unroll
        for (int t = 0; t < 16; t++)
            F[t] = *(const half8_t *)(xr + 16 * t);
        if constexpr (!Q6)
            xm = *(const half8_t *)(Xm + ((size_t)bc * SB_COLS + slot) * 16 + 8 * h);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++)
            d8[r4] = *(const float4_t_ *)(d8T + (size_t)bc * SB_COLS + 8 * r4 + 4 * h);
    }
    const uint8_t *col = A + (size_t)bc * TILE; // + rt * nb * TILE
    const size_t rstep = (size_t)nb * TILE;
    const int last = wg + ((n_rt - 1 - wg) / n_wg) * n_wg; // this work-group's last tile
    auto tile_at = [&](int rt) { return col + (size_t)(rt < last ? rt : last) * rstep; };
    sb_wset<TYPE> w0, w1;
    sb_load_set<TYPE>(w0, tile_at(wg), lane, i);
    sb_load_set<TYPE>(w1, tile_at(wg + n_wg), lane, i);

    auto reduce = [&](int rt, const float16_t_ &part) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            red[wave][r][lane] = mine ? part[r] : 0.0f;
        __syncthreads();
        for (int v = threadIdx.x; v < 16 * 64; v += NW * 64) {
            const int r = v >> 6, l = v & 63;
            const int tok = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
            const long row = (long)rt * 32 + (l & 31);
            if (tok < n && row < m) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; w++)
                    s += red[w][r][l];
                unsafeAtomicAdd(C + (long)tok * ldc + row, s);
            }
        }
        __syncthreads();
    };
    constexpr bool HOLD_D8 = !Q5 && !Q6; // (Q5_K / Q6_K carry more weight registers: their token scales are re-read from L2 per tile)
    auto fresh_d8 = [&]() {
        if constexpr (!HOLD_D8) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++)
                d8[r4] = *(const volatile float4_t_ *)(d8T + (size_t)bc * SB_COLS + 8 * r4 + 4 * h);
        }
    };
    for (int rt = wg; rt < n_rt; rt += 2 * n_wg) {
        fresh_d8();
        const float16_t_ p0 = sb_unit<TYPE>(w0, F, xm, d8, h, magic);
        sb_load_set<TYPE>(w0, tile_at(rt + 2 * n_wg), lane, i);
        reduce(rt, p0);
        if (rt + n_wg < n_rt) {
            fresh_d8();
            const float16_t_ p1 = sb_unit<TYPE>(w1, F, xm, d8, h, magic);
            sb_load_set<TYPE>(w1, tile_at(rt + 3 * n_wg), lane, i);
            reduce(rt + n_wg, p1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// UP TO 16 TOKENS, Q4_K / Q5_K: sixteen waves per work-group with the token fragments in LDS instead of registers.  The two bodies
// above keep a unit's sixteen 16-byte token fragments in 64 VGPRs, which leaves two waves per SIMD and one unit of weights in
// flight per wave (≈ 37 KB per CU: a third of what the HBM stream needs on matrices with several tiles per CU).  Here the
// work-group copies its share of the staged codes (n rows x its super-blocks, 512 B each, 16-byte chunks XOR-swizzled by the
// row) into LDS once, a wave reads a fragment right in front of the MFMA that consumes it, and 16 waves x one unit of weights
// (rolling refill) are in flight.  KSPLIT = 1: a work-group owns whole rows of the result and stores them; KSPLIT = 2 (deep rows,
// or more than 8 tokens: half the codes per work-group): two work-groups add into the zeroed result as above.
template <int TYPE, int KSPLIT, int RL>
__global__ __launch_bounds__(1024) void gemm_sb16_kernel(const uint8_t *__restrict__ A, long m, int nb, const _Float16 *__restrict__ Xh,
                                                         const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, int n,
                                                         float *__restrict__ C, long ldc, int n_rt) {
    constexpr bool Q5 = TYPE == LFAMD_TYPE_Q5_K;
    constexpr int TILE = Q5 ? P5K_TILE : P4K_TILE, NW = 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[]; // codes [nbl][n][512 B], then red [NW][RL][64] f32
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 31, h = lane >> 5;
    const int kh = KSPLIT == 2 ? (int)(blockIdx.x & 1) : 0, wg = KSPLIT == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int n_wg = KSPLIT == 2 ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int half = KSPLIT == 2 ? (nb + 1) >> 1 : nb;
    const int b0 = kh ? half : 0, b1 = kh ? nb : half, nbl = b1 - b0;
    float *red = (float *)(lds + (size_t)nbl * n * 512);
    // ---- this work-group's codes: chunk c (16 B = K-step t, half h: c = 2 t + h) of row (bl, slot) at c ^ (slot & 15)
    for (int e = threadIdx.x; e < nbl * n * 32; e += 1024) {
        const int c = e & 31, row = e >> 5, bl = row / n, slot = row - bl * n;
        const uint4 v = *(const uint4 *)(Xh + ((size_t)(b0 + bl) * SB_COLS + slot) * 256 + c * 8);
        *(uint4 *)(lds + (size_t)row * 512 + ((c ^ (slot & 15)) << 4)) = v;
    }
    __syncthreads();
    const int first = b0 + wave;
    const int nmine = first < b1 ? (b1 - first + NW - 1) / NW : 0;
    const int slot = i < n ? i : 0; // (token slots past n are never stored: they re-read slot 0)
    const uint32_t frag0 = (uint32_t)(slot * 512), sw = (uint32_t)(slot & 15);
    const uint32_t xmoff = (uint32_t)(slot * 16 + 8 * h);
    const float16_t_ zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t magic = opaque_magic();

    u32x4_t qs[4], hd, qh0 = {0, 0, 0, 0};
    half8_t xm;
    float4_t_ d8[(RL + 3) / 4];
    (void)qh0;
    int rt_n = wg, b_n = first;
    const uint8_t *tile_n = A;
    const _Float16 *xm_n = Xm;
    const float *d8_n = d8T;
    auto point = [&]() {
        tile_n = A + ((size_t)rt_n * nb + b_n) * TILE;
        xm_n = Xm + (size_t)b_n * SB_COLS * 16 + xmoff;
        d8_n = d8T + (size_t)b_n * SB_COLS + 4 * h;
    };
    auto advance = [&]() {
        int b2 = b_n + NW, rt2 = rt_n;
        if (b2 >= b1)
            b2 = first, rt2 = rt_n + n_wg;
        if (rt2 < n_rt)
            b_n = b2, rt_n = rt2;
        point();
    };
    auto ld_qs = [&](int g) { qs[g] = __builtin_nontemporal_load((const u32x4_t *)(tile_n + g * 1024 + lane * 16)); };
    auto ld_hd = [&]() { hd = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P4K_HDR + i * 16)); };
    auto ld_d8 = [&]() {
#pragma unroll
        for (int r4 = 0; r4 < (RL + 3) / 4; r4++)
            d8[r4] = *(const float4_t_ *)(d8_n + 8 * r4); // reg r of the tile is token slot 8 (r >> 2) + 4 h + (r & 3)
    };
    if (nmine > 0) {
        point();
#pragma unroll
        for (int g = 0; g < 4; g++)
            ld_qs(g);
        ld_hd();
        if constexpr (Q5)
            qh0 = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P5K_QH + lane * 16));
        xm = *(const half8_t *)xm_n;
        ld_d8();
    }
    for (int rt = wg; rt < n_rt; rt += n_wg) {
        float acc[RL];
#pragma unroll
        for (int r = 0; r < RL; r++)
            acc[r] = 0.0f;
        int b = first;
        for (int j = 0; j < nmine; j++, b += NW) {
            const uint8_t *fr = lds + (size_t)(b - b0) * n * 512 + frag0; // this lane's code row of super-block b
            advance();
            float16_t_ tmp = zero16;
            const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
            uint32_t sc03, sc47, mn03, mn47;
            q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
            ld_hd();
            uint32_t hq5[4] = {qh0.x, qh0.y, qh0.z, qh0.w};
            (void)hq5;
            if constexpr (Q5)
                qh0 = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P5K_QH + lane * 16));
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t qw[4] = {qs[g].x, qs[g].y, qs[g].z, qs[g].w};
                ld_qs(g);
#pragma unroll
                for (int dd = 0; dd < 4; dd++) {
                    const int t = 4 * g + dd, jb = t >> 1;
                    const q4_consts2 cp = q4_consts_pair(jb < 4 ? sc03 : sc47, (jb & 2) ? 2 : 0);
                    const int hsel = jb & 1;
                    const half2_t S = {cp.S[hsel], cp.S[hsel]}, O = {cp.O[hsel], cp.O[hsel]};
                    const half2_t S16 = {cp.S16[hsel], cp.S16[hsel]}, O16 = {cp.O16[hsel], cp.O16[hsel]};
                    half8_t wf;
                    if constexpr (Q5)
                        wf = dequant_q5(qw[dd], hq5[g] >> dd, S, O, S16, O16, magic);
                    else
                        wf = dequant_q4(qw[dd], S, O, S16, O16, magic);
                    const half8_t f = *(const half8_t *)(fr + ((((uint32_t)(2 * t) + (uint32_t)h) ^ sw) << 4));
                    tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, wf, tmp, 0, 0, 0);
                }
            }
            frag_u wm;
            const float mscale = h ? 64.0f : 1.0f;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t mw = p < 2 ? mn03 : mn47;
                const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
                const half2_t v = {(_Float16)(m0 * mscale), (_Float16)(m1 * mscale)};
                wm.p[p] = v;
            }
            const float16_t_ tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wm.v, zero16, 0, 0, 0);
            xm = *(const half8_t *)xm_n;
#pragma unroll
            for (int r = 0; r < RL; r++) {
                const float u = fmaf(-dmin, tm[r], d * tmp[r]);
                acc[r] = fmaf(u, d8[r >> 2][r & 3], acc[r]);
            }
            ld_d8();
        }
        // ---- the waves' partial tiles (the RL registers that hold token slots below 16), summed in wave order
#pragma unroll
        for (int r = 0; r < RL; r++)
            red[(wave * RL + r) * 64 + lane] = acc[r];
        __syncthreads();
        for (int v = threadIdx.x; v < RL * 64; v += 1024) {
            const int r = v >> 6, l = v & 63;
            const int tok = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
            const long row = (long)rt * 32 + (l & 31);
            if (tok < n && row < m) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; w++)
                    s += red[(w * RL + r) * 64 + l];
                if constexpr (KSPLIT == 2)
                    unsafeAtomicAdd(C + (long)tok * ldc + row, s);
                else
                    C[(long)tok * ldc + row] = s;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The 16-wave body on the INT8 matrix cores (Q4_K, up to 16 tokens): `v_mfma_i32_32x32x32_i8` takes one 32-weight sub-block per
// instruction, so the nibbles only have to become bytes (two ANDs and a shift per eight weights instead of the eleven VALU that
// build an f16 fragment), the activation codes stay int8 (half the LDS and L2 bytes), a sub-block's sum is an exact int32 and
// its 6-bit scale is applied with one `v_mad_i32_i24` per LIVE accumulator register (RL = 4 up to 8 tokens).  ≈ 110 VALU per
// (tile, super-block) unit against ≈ 270 of the f16 body; same results class (exact integer dots, f32 scales: <= 2e-6).
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
template <int KSPLIT, int RL>
__global__ __launch_bounds__(1024) void gemm_sb16i_kernel(const uint8_t *__restrict__ A, long m, int nb, const int8_t *__restrict__ Xq,
                                                          const float *__restrict__ d8T, const _Float16 *__restrict__ Xm, int n,
                                                          float *__restrict__ C, long ldc, int n_rt) {
    constexpr int TILE = P4K_TILE, NW = 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[]; // codes [nbl][n][256 B], then red [NW][RL][64] f32
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 31, h = lane >> 5;
    const int kh = KSPLIT == 2 ? (int)(blockIdx.x & 1) : 0, wg = KSPLIT == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int n_wg = KSPLIT == 2 ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int half = KSPLIT == 2 ? (nb + 1) >> 1 : nb;
    const int b0 = kh ? half : 0, b1 = kh ? nb : half, nbl = b1 - b0;
    float *red = (float *)(lds + (size_t)nbl * n * 256);
    for (int e = threadIdx.x; e < nbl * n * 16; e += 1024) { // 16-byte chunk c = 2 jb + h of row (bl, slot) at c ^ (slot & 15)
        const int c = e & 15, row = e >> 4, bl = row / n, slot = row - bl * n;
        const uint4 v = *(const uint4 *)(Xq + ((size_t)(b0 + bl) * SB_COLS + slot) * 256 + c * 16);
        *(uint4 *)(lds + (size_t)row * 256 + ((c ^ (slot & 15)) << 4)) = v;
    }
    __syncthreads();
    const int first = b0 + wave;
    const int nmine = first < b1 ? (b1 - first + NW - 1) / NW : 0;
    const int slot = i < n ? i : 0;
    const uint32_t frag0 = (uint32_t)(slot * 256), sw = (uint32_t)(slot & 15);
    const uint32_t xmoff = (uint32_t)(slot * 16 + 8 * h);
    const float16_t_ zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const v16i_t zero16i = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    u32x4_t qs[4], hd;
    half8_t xm;
    float4_t_ d8[(RL + 3) / 4];
    int rt_n = wg, b_n = first;
    const uint8_t *tile_n = A;
    const _Float16 *xm_n = Xm;
    const float *d8_n = d8T;
    auto point = [&]() {
        tile_n = A + ((size_t)rt_n * nb + b_n) * TILE;
        xm_n = Xm + (size_t)b_n * SB_COLS * 16 + xmoff;
        d8_n = d8T + (size_t)b_n * SB_COLS + 4 * h;
    };
    auto advance = [&]() {
        int b2 = b_n + NW, rt2 = rt_n;
        if (b2 >= b1)
            b2 = first, rt2 = rt_n + n_wg;
        if (rt2 < n_rt)
            b_n = b2, rt_n = rt2;
        point();
    };
    auto ld_qs = [&](int g) { qs[g] = __builtin_nontemporal_load((const u32x4_t *)(tile_n + g * 1024 + lane * 16)); };
    auto ld_hd = [&]() { hd = __builtin_nontemporal_load((const u32x4_t *)(tile_n + P4K_HDR + i * 16)); };
    auto ld_d8 = [&]() {
#pragma unroll
        for (int r4 = 0; r4 < (RL + 3) / 4; r4++)
            d8[r4] = *(const float4_t_ *)(d8_n + 8 * r4);
    };
    if (nmine > 0) {
        point();
#pragma unroll
        for (int g = 0; g < 4; g++)
            ld_qs(g);
        ld_hd();
        xm = *(const half8_t *)xm_n;
        ld_d8();
    }
    for (int rt = wg; rt < n_rt; rt += n_wg) {
        float acc[RL];
#pragma unroll
        for (int r = 0; r < RL; r++)
            acc[r] = 0.0f;
        int b = first;
        for (int j = 0; j < nmine; j++, b += NW) {
            const uint8_t *fr = lds + (size_t)(b - b0) * n * 256 + frag0; // this lane's code row of super-block b
            advance();
            const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
            uint32_t sc03, sc47, mn03, mn47;
            q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
            ld_hd();
            int sumi[RL];
#pragma unroll
            for (int r = 0; r < RL; r++)
                sumi[r] = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t qw[4] = {qs[g].x, qs[g].y, qs[g].z, qs[g].w};
                ld_qs(g);
#pragma unroll
                for (int e = 0; e < 2; e++) { // sub-block jb = 2 g + e: K-steps 4 g + 2 e, + 1
                    const int jb = 2 * g + e;
                    const uint32_t x0 = qw[2 * e], x1 = qw[2 * e + 1];
                    const v4i_t wv = {(int)(x0 & 0x0F0F0F0Fu), (int)((x0 >> 4) & 0x0F0F0F0Fu), (int)(x1 & 0x0F0F0F0Fu), (int)((x1 >> 4) & 0x0F0F0F0Fu)};
                    const v4i_t av = *(const v4i_t *)(fr + ((((uint32_t)(2 * jb) + (uint32_t)h) ^ sw) << 4));
                    const v16i_t p = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, wv, zero16i, 0, 0, 0);
                    const int sc = (int)(((jb < 4 ? sc03 : sc47) >> (8 * (jb & 3))) & 0xff);
#pragma unroll
                    for (int r = 0; r < RL; r++)
                        sumi[r] += __mul24(p[r], sc); // (|p| <= 15 * 127 * 32, sc <= 63: 24-bit operands)
                }
            }
            frag_u wm;
            const float mscale = h ? 64.0f : 1.0f;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t mw = p < 2 ? mn03 : mn47;
                const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
                const half2_t v = {(_Float16)(m0 * mscale), (_Float16)(m1 * mscale)};
                wm.p[p] = v;
            }
            const float16_t_ tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wm.v, zero16, 0, 0, 0);
            xm = *(const half8_t *)xm_n;
#pragma unroll
            for (int r = 0; r < RL; r++) {
                const float u = fmaf(-dmin, tm[r], d * (float)sumi[r]);
                acc[r] = fmaf(u, d8[r >> 2][r & 3], acc[r]);
            }
            ld_d8();
        }
#pragma unroll
        for (int r = 0; r < RL; r++)
            red[(wave * RL + r) * 64 + lane] = acc[r];
        __syncthreads();
        for (int v = threadIdx.x; v < RL * 64; v += 1024) {
            const int r = v >> 6, l = v & 63;
            const int tok = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
            const long row = (long)rt * 32 + (l & 31);
            if (tok < n && row < m) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; w++)
                    s += red[(w * RL + r) * 64 + l];
                if constexpr (KSPLIT == 2)
                    unsafeAtomicAdd(C + (long)tok * ldc + row, s);
                else
                    C[(long)tok * ldc + row] = s;
            }
        }
        __syncthreads();
    }
}

static int lfamd_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            cus = p.multiProcessorCount;
        if (cus <= 0)
            cus = 256;
    }
    return cus;
}

extern "C" {

size_t lfamd_gemm_sb_workspace(long k) { // Xh, d8T, Xm
    const size_t nb = (size_t)(k / 256);
    return nb * SB_COLS * 512 + nb * SB_COLS * 4 + nb * SB_COLS * 32;
}

bool lfamd_gemm_sb_ok(int Atype, long k, long n) {
    static const bool off = getenv("LFAMD_NO_SMALL_BATCH") != nullptr;
    return !off && (Atype == LFAMD_TYPE_Q4_K || Atype == LFAMD_TYPE_Q5_K || Atype == LFAMD_TYPE_Q6_K) && k > 0 && k % 256 == 0 && n >= 2 &&
           n <= SB_COLS;
}

// reuse_stage != 0: the workspace already holds these activations staged by the previous call (sibling matrices of one launch
// group: ffn_gate / ffn_up, attn_q / k / v): only the zeroing of this matrix's result, where its body needs it, is launched.
hipError_t lfamd_launch_gemm_sb(int Atype, const void *A, long m, long k, int Btype, const void *B, size_t b_row_bytes, long n, float *C,
                                long ldc, void *ws, int reuse_stage, hipStream_t s) {
    const int nb = (int)(k / 256);
    uint8_t *w8 = (uint8_t *)ws;
    _Float16 *Xh = (_Float16 *)w8;
    float *d8T = (float *)(w8 + (size_t)nb * SB_COLS * 512);
    _Float16 *Xm = (_Float16 *)(w8 + (size_t)nb * SB_COLS * 512 + (size_t)nb * SB_COLS * 4);
    const unsigned pg = (unsigned)(n * nb); // (token slots past n are never read: gemm_sb_kernel re-reads slot 0 for them)
    const bool mins = Atype != LFAMD_TYPE_Q6_K;
    // up to 16 tokens of Q4_K / Q5_K: the 16-wave body with the codes in LDS; K halves when one work-group's codes would not fit
    // beside the reduction buffer, and on deep rows (twice the work-groups for the 128 row tiles of a 4096-row matrix)
    static const bool no16 = getenv("LFAMD_SB_NO16") != nullptr;
    const int rl = n <= 8 ? 4 : 8;
    const size_t red_bytes = 16 * (size_t)rl * 64 * 4;
    int ksplit = 0;
    // (measured, profiles/r03_small_batch16.txt: a gain only where a work-group walks several tiles — 14336 x 4096 n = 8: 20.5 -> 16.4 us;
    //  with one tile per work-group the two-launch latency chain dominates and the 8-wave bodies are as fast)
    static const bool no_i8 = getenv("LFAMD_SB_NO_I8") != nullptr; // development: A/B against the f16 bodies
    // Q4_K up to 8 tokens: the int8 form of the 16-wave body, on every shape (14336 x 4096: 16.6 -> 13.2 us, 4096 x 14336: 16.4 -> 13.4,
    // 8192 x 4096: 15.1 -> 10.6, 4096 x 4096: 9.1 -> 8.5); the f16 form (Q5_K, 9 .. 16 tokens) only where a work-group walks several
    // tiles (profiles/r03_small_batch16.txt)
    const bool i8_ok = Atype == LFAMD_TYPE_Q4_K && n <= 8 && !no_i8;
    if (mins && n <= 16 && !no16 && ((m + 31) / 32 > lfamd_num_cus() || i8_ok)) {
        if ((size_t)nb * n * 512 + red_bytes <= 150 * 1024 && nb <= 16)
            ksplit = 1;
        else if ((size_t)((nb + 1) / 2) * n * 512 + red_bytes <= 150 * 1024)
            ksplit = 2;
    }
    const bool i8 = ksplit && i8_ok; // (the 16-token form of the int8 body spills at 128 VGPRs: f16 body there)
    if (i8) { // int8 codes need half the LDS: whole rows of K fit where the f16 body splits
        if ((size_t)nb * n * 256 + red_bytes <= 150 * 1024 && nb <= 16)
            ksplit = 1;
        else if ((size_t)((nb + 1) / 2) * n * 256 + red_bytes <= 150 * 1024)
            ksplit = 2;
    }
    float *Czero = ksplit == 1 ? nullptr : C;
    if (reuse_stage) {
        B = nullptr;
        if (!Czero)
            goto staged;
    }
    if (i8) {
        if (Btype == LFAMD_TYPE_F32)
            sb_prep_kernel<true, true, true><<<pg, 64, 0, s>>>((const uint8_t *)B, b_row_bytes, (int)n, nb, Xh, d8T, Xm, Czero, m, ldc);
        else
            sb_prep_kernel<false, true, true><<<pg, 64, 0, s>>>((const uint8_t *)B, b_row_bytes, (int)n, nb, Xh, d8T, Xm, Czero, m, ldc);
    } else if (Btype == LFAMD_TYPE_F32) {
        if (mins)
            sb_prep_kernel<true, true><<<pg, 64, 0, s>>>((const uint8_t *)B, b_row_bytes, (int)n, nb, Xh, d8T, Xm, Czero, m, ldc);
        else
            sb_prep_kernel<true, false><<<pg, 64, 0, s>>>((const uint8_t *)B, b_row_bytes, (int)n, nb, Xh, d8T, Xm, Czero, m, ldc);
    } else {
        if (mins)
            sb_prep_kernel<false, true><<<pg, 64, 0, s>>>((const uint8_t *)B, b_row_bytes, (int)n, nb, Xh, d8T, Xm, Czero, m, ldc);
        else
            sb_prep_kernel<false, false><<<pg, 64, 0, s>>>((const uint8_t *)B, b_row_bytes, (int)n, nb, Xh, d8T, Xm, Czero, m, ldc);
    }
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
staged:
    const int n_rt = (int)((m + 31) / 32);
    if (ksplit) {
        const int cus = lfamd_num_cus();
        const size_t smem = (size_t)(ksplit == 1 ? nb : (nb + 1) / 2) * n * 512 + red_bytes;
        const unsigned g16 = (unsigned)ksplit * (unsigned)(n_rt < cus ? n_rt : cus);
        auto go = [&](auto kernel) {
            if (smem > 64 * 1024) {
                hipError_t e2 = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                if (e2 != hipSuccess)
                    return e2;
            }
            kernel<<<g16, 1024, smem, s>>>((const uint8_t *)A, m, nb, Xh, d8T, Xm, (int)n, C, ldc, n_rt);
            return hipGetLastError();
        };
        if (i8) {
            const size_t smem8 = (size_t)(ksplit == 1 ? nb : (nb + 1) / 2) * n * 256 + red_bytes;
            auto go8 = [&](auto kernel) {
                if (smem8 > 64 * 1024) {
                    hipError_t e2 = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem8);
                    if (e2 != hipSuccess)
                        return e2;
                }
                kernel<<<g16, 1024, smem8, s>>>((const uint8_t *)A, m, nb, (const int8_t *)Xh, d8T, Xm, (int)n, C, ldc, n_rt);
                return hipGetLastError();
            };
            return ksplit == 1 ? go8(gemm_sb16i_kernel<1, 4>) : go8(gemm_sb16i_kernel<2, 4>);
        }
        const bool q5 = Atype == LFAMD_TYPE_Q5_K;
        if (ksplit == 1)
            return rl == 4 ? (q5 ? go(gemm_sb16_kernel<LFAMD_TYPE_Q5_K, 1, 4>) : go(gemm_sb16_kernel<LFAMD_TYPE_Q4_K, 1, 4>))
                           : (q5 ? go(gemm_sb16_kernel<LFAMD_TYPE_Q5_K, 1, 8>) : go(gemm_sb16_kernel<LFAMD_TYPE_Q4_K, 1, 8>));
        return rl == 4 ? (q5 ? go(gemm_sb16_kernel<LFAMD_TYPE_Q5_K, 2, 4>) : go(gemm_sb16_kernel<LFAMD_TYPE_Q4_K, 2, 4>))
                       : (q5 ? go(gemm_sb16_kernel<LFAMD_TYPE_Q5_K, 2, 8>) : go(gemm_sb16_kernel<LFAMD_TYPE_Q4_K, 2, 8>));
    }
    const unsigned grid = 2u * (unsigned)(n_rt < lfamd_num_cus() ? n_rt : lfamd_num_cus()); // one 8-wave work-group per CU, K halves adjacent
    constexpr int NW = 8;
    static const bool no_shallow = getenv("LFAMD_SB_ROLLING") != nullptr; // development: A/B of the two bodies
    if ((nb + 1) / 2 <= NW && !no_shallow && Atype != LFAMD_TYPE_Q6_K) { // (Q6_K: more weight registers than the second set leaves room for) // one super-block per wave and K half: token fragments held in registers across the tiles
        switch (Atype) {
        case LFAMD_TYPE_Q4_K:
            gemm_sb_shallow_kernel<LFAMD_TYPE_Q4_K, NW><<<grid, NW * 64, 0, s>>>((const uint8_t *)A, m, nb, Xh, d8T, Xm, (int)n, C, ldc, n_rt);
            break;
        case LFAMD_TYPE_Q5_K:
            gemm_sb_shallow_kernel<LFAMD_TYPE_Q5_K, NW><<<grid, NW * 64, 0, s>>>((const uint8_t *)A, m, nb, Xh, d8T, Xm, (int)n, C, ldc, n_rt);
            break;
        case LFAMD_TYPE_Q6_K:
            gemm_sb_shallow_kernel<LFAMD_TYPE_Q6_K, NW><<<grid, NW * 64, 0, s>>>((const uint8_t *)A, m, nb, Xh, d8T, Xm, (int)n, C, ldc, n_rt);
            break;
        default:
            return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (Atype) {
    case LFAMD_TYPE_Q4_K:
        gemm_sb_kernel<LFAMD_TYPE_Q4_K, NW><<<grid, NW * 64, 0, s>>>((const uint8_t *)A, m, nb, Xh, d8T, Xm, (int)n, C, ldc, n_rt);
        break;
    case LFAMD_TYPE_Q5_K:
        gemm_sb_kernel<LFAMD_TYPE_Q5_K, NW><<<grid, NW * 64, 0, s>>>((const uint8_t *)A, m, nb, Xh, d8T, Xm, (int)n, C, ldc, n_rt);
        break;
    case LFAMD_TYPE_Q6_K:
        gemm_sb_kernel<LFAMD_TYPE_Q6_K, NW><<<grid, NW * 64, 0, s>>>((const uint8_t *)A, m, nb, Xh, d8T, Xm, (int)n, C, ldc, n_rt);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
}
