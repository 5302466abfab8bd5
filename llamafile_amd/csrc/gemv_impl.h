// gemv.hip — wave-reduction GEMV kernels for the batch-1 (decode) case and small batches n <= 8.
//
// Replaces the reference's mul_mat_vec_q + quantize_q8_1 pair (ggml-cuda.cu.patch:14428-14575,
// 15259-15293: 32-wide warps, dp4a with a scalar fallback on gfx950) and follows the CPU
// path's arithmetic:
//   Q4_K / Q6_K x Q8_K : mul_mat_qX_K_q8_K_T (iqk_mul_mat.inc:601-643) — exact int8 block dots,
//                        f32 scales; the -32 offset of Q6_K is folded into a sums term like
//                        DequantizerQ6K does (iqk_mul_mat.inc:570-599).
//   Q8_0 x Q8_0        : tinyBLAS_Q0_AVX2::gemm (tinyblas_cpu.h:934-971) BIT-EXACT: 8 f32 lanes per
//                        output, blocks accumulated sequentially with fma (or Kahan), same hsum tree.
//
// Activations may arrive already quantised (the llamafile_sgemm boundary: Btype = Q8_K / Q8_0) or as
// f32 (the GGML_OP_MUL_MAT boundary): then every work-group quantises the (tiny) activation vector
// itself in its prologue, bit-identically to quantize_row_q8_K / q8_0, which removes one launch and
// one HBM round trip per mat-mul.
//
// These are HBM-bandwidth kernels: weights stream once from the packed layout with one
// global_load_dwordx4 per lane (1 KiB per wave instruction) straight into VGPRs — no LDS round
// trip for the weights (cdna_hip_programming.md §5 "GEMV / M <= 16" row).  The first chunk of
// weight loads is issued BEFORE the activation staging so HBM latency overlaps it, and the next
// chunk is always in flight while the current one is consumed.
#pragma once
#include "lfamd_device.h"
#include <stdlib.h>

// LDS image of one Q8_K activation block for the K-quant GEMVs.  A lane = (gsel, h) reads its 64 code
// bytes (two groups g = 2gsel+gi, four K-steps dd each) with four ds_read_b128, its 8 half-sums with
// one more, its 4 sub-block sums with a ds_read_b64.
//   [0,256)    codes: 8-byte groups at position pos = 16 gsel + 8 h + 4 gi + dd
//   [256,320)  hb  : int16 sum of each 8-byte group, same position order
//   [320,352)  ps  : int16 sum of K-step pairs (dd = 2e, 2e+1): position 8 gsel + 4 h + 2 gi + e
//   [352,384)  d   : f32 block scale (Q8_K: one; Q8_0-quantised activations for the legacy 32-block types: eight;
//                    Q8_1: eight dwords {f16 d, f16 s = d * sum(q)} like the block_q8_1 header)
#define XBLK 384
#define XBLK_HB 256
#define XBLK_PS 320
#define XBLK_D 352

// Within every 8-byte group the codes (y0..y7) are stored as (y0,y4,y1,y5 | y2,y6,y3,y7): the order
// in which (x & 0x0F0F0F0F) and ((x>>4) & 0x0F0F0F0F) expose the nibbles of a packed K-step dword.
// grp = (k offset)/8 = 16 gsel + 8 gi + 2 dd + h.  Returns the group's code sum.
__device__ static inline int put_group(uint8_t *dst, int grp, uint32_t y0, uint32_t y1) {
    const uint32_t p0 = __builtin_amdgcn_perm(y1, y0, 0x05010400);
    const uint32_t p1 = __builtin_amdgcn_perm(y1, y0, 0x07030602);
    const int pos = (grp & 16) | ((grp & 1) << 3) | ((grp & 8) >> 1) | ((grp >> 1) & 3);
    *(uint2 *)(dst + 8 * pos) = make_uint2(p0, p1);
    int hs = sdot4(y0, 0x01010101u, 0);
    hs = sdot4(y1, 0x01010101u, hs);
    *(int16_t *)(dst + XBLK_HB + 2 * pos) = (int16_t)hs;
    return hs;
}

// pair sum of groups grp (dd even) and grp+2 (dd odd), written by the even one
__device__ static inline void put_pair(uint8_t *dst, int grp, int hs_even_plus_odd) {
    const int pp = ((grp & 16) >> 1) | ((grp & 1) << 2) | ((grp & 8) >> 2) | ((grp >> 2) & 1);
    *(int16_t *)(dst + XBLK_PS + 2 * pp) = (int16_t)hs_even_plus_odd;
}

// already-quantised Q8_K rows (llamafile field order {d, bsums[16], qs[256]}); blockDim % 32 == 0
__device__ static inline void stage_q8k(uint8_t *lds, const uint8_t *B, size_t b_row_bytes, long col0, int nc, int nb) {
    const int groups = nc * nb * 32; // 8-byte groups; a block's 32 groups sit in 32 consecutive lanes
    for (int gidx = threadIdx.x; gidx < groups; gidx += blockDim.x) {
        int c = gidx / (nb * 32), r = gidx % (nb * 32);
        int b = r >> 5, grp = r & 31;
        const uint8_t *y = B + (col0 + c) * b_row_bytes + (size_t)b * 292;
        const uint32_t *src = (const uint32_t *)(y + 36 + 8 * grp);
        uint8_t *dst = lds + (size_t)(c * nb + b) * XBLK;
        const int hs = put_group(dst, grp, src[0], src[1]);
        const int other = __shfl_xor(hs, 2, 64); // group grp ^ 2: the other K-step of the pair
        if ((grp & 2) == 0)
            put_pair(dst, grp, hs + other);
        if (grp == 0)
            *(float *)(dst + XBLK_D) = *(const float *)y;
    }
}

// f32 rows, quantised here exactly like quantize_row_q8_K (upstream ggml-quants.c; restated in
// quantize.hip): first index of the largest |x| fixes the sign of iscale = -128/max, codes are
// nearest_int (round-half-even) clamped at 127, d = 1/iscale.  One "piece" = 16 consecutive floats;
// 16 consecutive lanes share one 256-block.
__device__ static inline void quantise_piece_q8k(uint8_t *dst, const float (&v)[16], int l16) {
    float amax = 0.0f, val = 0.0f;
    int idx = l16 * 16;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        float ax = fabsf(v[e]);
        if (ax > amax) {
            amax = ax;
            val = v[e];
            idx = l16 * 16 + e;
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        float oa = __shfl_xor(amax, off, 64);
        int oi = __shfl_xor(idx, off, 64);
        float ov = __shfl_xor(val, off, 64);
        if (oa > amax || (oa == amax && oi < idx)) {
            amax = oa;
            idx = oi;
            val = ov;
        }
    }
    uint32_t y[4] = {0, 0, 0, 0};
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -128.0f / val;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            int q = (int)rintf(iscale * v[e]);
            q = q > 127 ? 127 : q;
            y[e >> 2] |= (uint32_t)(q & 0xff) << (8 * (e & 3));
        }
        d = 1.0f / iscale;
    }
    // this lane's 16 codes = groups 2*l16 (h = 0) and 2*l16+1 (h = 1) of K-step dd = l16 & 3
    const int hs0 = put_group(dst, 2 * l16 + 0, y[0], y[1]);
    const int hs1 = put_group(dst, 2 * l16 + 1, y[2], y[3]);
    const int o0 = __shfl_xor(hs0, 1, 64), o1 = __shfl_xor(hs1, 1, 64); // K-step dd ^ 1
    if ((l16 & 1) == 0) {
        put_pair(dst, 2 * l16 + 0, hs0 + o0);
        put_pair(dst, 2 * l16 + 1, hs1 + o1);
    }
    if (l16 == 0)
        *(float *)(dst + XBLK_D) = d;
}

__device__ static inline void load_piece(float (&v)[16], const float *x, int p) {
    const float4 *src = (const float4 *)(x + (size_t)p * 16);
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float4 f = src[e];
        v[4 * e + 0] = f.x, v[4 * e + 1] = f.y, v[4 * e + 2] = f.z, v[4 * e + 3] = f.w;
    }
}

__device__ static inline void stage_f32_as_q8k(uint8_t *lds, const uint8_t *X, size_t x_row_bytes, long col0, int nc,
                                               int nb) {
    const int pieces = nb * 16;
    const int l16 = threadIdx.x & 15;
    for (int c = 0; c < nc; c++) {
        const float *x = (const float *)(X + (col0 + c) * x_row_bytes);
        for (int p = threadIdx.x; p < pieces; p += blockDim.x) {
            float v[16];
            load_piece(v, x, p);
            quantise_piece_q8k(lds + (size_t)(c * nb + (p >> 4)) * XBLK, v, l16);
        }
    }
}

// scale slot `blk` (0..7) of a Q8_0 / Q8_1 activation image: Q8_0 keeps the f16-rounded d as f32; Q8_1 keeps {d, s} as
// the two f16 of block_q8_1, s = f16(d * sum) from the UNROUNDED d like upstream quantize_row_q8_1
template <bool S1>
__device__ static inline void put_scale_q80(uint8_t *dst, int blk, float d, int sum) {
    if constexpr (S1)
        *(uint32_t *)(dst + XBLK_D + 4 * blk) = (uint32_t)f2h_bits(d) | ((uint32_t)f2h_bits((float)sum * d) << 16);
    else
        *(float *)(dst + XBLK_D + 4 * blk) = h2f(f2h_bits(d));
}

// ---- activations of the legacy 32-block weight types (Q4_0 ...): Q8_0 quantisation (d = amax/127 per 32 values,
// stored as f16; q = roundf(x/d): upstream quantize_row_q8_0), same code / group-sum / pair-sum image, eight scales.
template <bool S1 = false>
__device__ static inline void quantise_piece_q80(uint8_t *dst, const float (&v)[16], int l16) {
    float amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; e++)
        amax = fmaxf(amax, fabsf(v[e]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64)); // lanes 2b, 2b+1 hold the two halves of 32-block b
    const float d = amax / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    uint32_t y[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int q = (int)roundf(v[e] * id);
        y[e >> 2] |= (uint32_t)(q & 0xff) << (8 * (e & 3));
    }
    const int hs0 = put_group(dst, 2 * l16 + 0, y[0], y[1]);
    const int hs1 = put_group(dst, 2 * l16 + 1, y[2], y[3]);
    const int o0 = __shfl_xor(hs0, 1, 64), o1 = __shfl_xor(hs1, 1, 64);
    if ((l16 & 1) == 0) {
        put_pair(dst, 2 * l16 + 0, hs0 + o0);
        put_pair(dst, 2 * l16 + 1, hs1 + o1);
        put_scale_q80<S1>(dst, l16 >> 1, d, hs0 + o0 + hs1 + o1);
    }
}

template <bool S1>
__device__ static inline void stage_f32_as_q80(uint8_t *lds, const uint8_t *X, size_t x_row_bytes, long col0, int nc,
                                               int nb) {
    const int pieces = nb * 16;
    const int l16 = threadIdx.x & 15;
    for (int c = 0; c < nc; c++) {
        const float *x = (const float *)(X + (col0 + c) * x_row_bytes);
        for (int p = threadIdx.x; p < pieces; p += blockDim.x) {
            float v[16];
            load_piece(v, x, p);
            quantise_piece_q80<S1>(lds + (size_t)(c * nb + (p >> 4)) * XBLK, v, l16);
        }
    }
}

// already-quantised Q8_0 (34-byte blocks) / Q8_1 (36-byte blocks {d, s, qs}) rows: one 8-byte group per lane, 32
// consecutive lanes per 256 codes
template <bool S1>
__device__ static inline void stage_q80_blocks(uint8_t *lds, const uint8_t *B, size_t b_row_bytes, long col0, int nc, int nb) {
    constexpr int BS = S1 ? 36 : 34, QS = S1 ? 4 : 2;
    const int groups = nc * nb * 32;
    for (int gidx = threadIdx.x; gidx < groups; gidx += blockDim.x) {
        int c = gidx / (nb * 32), r = gidx % (nb * 32);
        int b = r >> 5, grp = r & 31;
        const uint8_t *blk = B + (col0 + c) * b_row_bytes + (size_t)(b * 8 + (grp >> 2)) * BS;
        const uint16_t *src = (const uint16_t *)(blk + QS + 8 * (grp & 3));
        const uint32_t y0 = (uint32_t)src[0] | ((uint32_t)src[1] << 16), y1 = (uint32_t)src[2] | ((uint32_t)src[3] << 16);
        uint8_t *dst = lds + (size_t)(c * nb + b) * XBLK;
        const int hs = put_group(dst, grp, y0, y1);
        const int other = __shfl_xor(hs, 2, 64);
        if ((grp & 2) == 0)
            put_pair(dst, grp, hs + other);
        if ((grp & 3) == 0) {
            if constexpr (S1)
                *(uint32_t *)(dst + XBLK_D + 4 * (grp >> 2)) = (uint32_t)((const uint16_t *)blk)[0] | ((uint32_t)((const uint16_t *)blk)[1] << 16);
            else
                *(float *)(dst + XBLK_D + 4 * (grp >> 2)) = h2f(*(const uint16_t *)blk);
        }
    }
}

template <int BT, int ACT>
__device__ static inline void stage_x(uint8_t *lds, const uint8_t *B, size_t b_row_bytes, long col0, int nc, int nb) {
    if constexpr (ACT == LFAMD_TYPE_Q8_K) {
        if constexpr (BT == LFAMD_TYPE_F32)
            stage_f32_as_q8k(lds, B, b_row_bytes, col0, nc, nb);
        else
            stage_q8k(lds, B, b_row_bytes, col0, nc, nb);
    } else {
        if constexpr (BT == LFAMD_TYPE_F32)
            stage_f32_as_q80<ACT == LFAMD_TYPE_Q8_1>(lds, B, b_row_bytes, col0, nc, nb);
        else
            stage_q80_blocks<ACT == LFAMD_TYPE_Q8_1>(lds, B, b_row_bytes, col0, nc, nb);
    }
}

// ---------------------------------------------------------------------------------------------
// K-quant GEMV skeleton.  PERSISTENT work-groups: the grid is sized to about two work-groups per CU and
// each work-group walks half-tiles (16 weight rows) ht = blockIdx.x, +gridDim.x, ...  The activation
// vector is quantised / staged ONCE per work-group and reused for all its tiles, and the weight
// stream is software-pipelined across tiles: the loads of work item f+1 are issued before item f is
// consumed, so every wave keeps 12+ KiB of HBM reads in flight for its whole life.
//   work item = (tile, chunk of GEMV_CH super-blocks of this wave); wave w owns super-blocks w, w+NW, ...
//   NW = 16 waves for a single activation row: a k = 4096 row (16 super-blocks) then costs each wave ONE
//   super-block per tile, so the integer-dot phase that follows the arrival of the data is as short as it
//   can be; unused chunk slots are never touched, so the register allocator drops them.
//   lane = (i16 = lane&15, h = (lane>>4)&1, gsel = lane>>5) covers groups g = 2*gsel + gi (gi = 0,1).

#define GEMV_CH_MAX 4

struct q4k_traits {
    static constexpr int ACT = LFAMD_TYPE_Q8_K; // activation quantisation the reference uses for this type
    static constexpr int TILE = P4K_TILE;
    struct chunk {
        uint4 q0[GEMV_CH_MAX], q1[GEMV_CH_MAX], hd[GEMV_CH_MAX];
    };
    // `off` = byte offset of the super-block's tile inside the row-tile the descriptor covers
    __device__ static inline void load(chunk &ch, int s, lfamd_rsrc r, uint32_t off, int gsel, int slot, int hrow) {
        ch.q0[s] = buf_ld16_nt(r, off + (2 * gsel + 0) * 1024 + slot * 16);
        ch.q1[s] = buf_ld16_nt(r, off + (2 * gsel + 1) * 1024 + slot * 16);
        ch.hd[s] = buf_ld16_nt(r, off + P4K_HDR + hrow * 16);
    }
    // one super-block of this lane against one staged activation block; returns the f32 contribution
    __device__ static inline float dot(const chunk &ch, int s, const uint8_t *xb, int gsel, int h) {
        const uint4 q0 = ch.q0[s], q1 = ch.q1[s], hd = ch.hd[s];
        const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
        uint32_t sc03, sc47, mn03, mn47;
        q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
        // this lane's four sub-blocks: j = 2g + e, g = 2*gsel + gi  ->  j = 4*gsel + 2*gi + e
        const uint32_t scw = gsel ? sc47 : sc03, mnw = gsel ? mn47 : mn03;
        const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const uint4 *yq = (const uint4 *)(xb + 128 * gsel + 64 * h);
        const uint4 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3];
        const uint32_t yw[16] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w,
                                 yc.x, yc.y, yc.z, yc.w, yd.x, yd.y, yd.z, yd.w};
        const uint2 psw = *(const uint2 *)(xb + XBLK_PS + 16 * gsel + 8 * h);
        const int ps[4] = {(int)(int16_t)(psw.x & 0xffff), (int)(int16_t)(psw.x >> 16), (int)(int16_t)(psw.y & 0xffff),
                           (int)(int16_t)(psw.y >> 16)};
        int sumi = 0, summ = 0;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) { // jj = 2 gi + e
            int isum = 0;
#pragma unroll
            for (int d2 = 0; d2 < 2; d2++) {
                const int t8 = 2 * jj + d2; // = 4 gi + dd
                const uint32_t x = qw[t8];
                isum = sdot4(x & 0x0F0F0F0F, yw[2 * t8], isum);
                isum = sdot4((x >> 4) & 0x0F0F0F0F, yw[2 * t8 + 1], isum);
            }
            sumi += (int)((scw >> (8 * jj)) & 0xff) * isum;
            summ += (int)((mnw >> (8 * jj)) & 0xff) * ps[jj];
        }
        const float d8 = *(const float *)(xb + XBLK_D);
        // d*d8*sumi - dmin*d8*summ  (iqk_mul_mat.inc:284-291, 632)
        return fmaf(d * d8, (float)sumi, -(dmin * d8) * (float)summ);
    }
};

// Q4_0 (legacy 32-blocks, activations Q8_0; mul_mat_qX_0_q8_0_T, iqk_mul_mat.inc:998-1349): the P4K nibble image with
// eight f16 block scales per row as header; w = d*(q - 8).  Per 32-block: d*d8*(<q, q8> - 8*sum(q8)), the sum taken
// from the staged pair sums like the Q4_K mins.
struct q40_traits {
    static constexpr int ACT = LFAMD_TYPE_Q8_0;
    static constexpr int TILE = P4K_TILE;
    struct chunk {
        uint4 q0[GEMV_CH_MAX], q1[GEMV_CH_MAX];
        uint2 hd[GEMV_CH_MAX];
    };
    __device__ static inline void load(chunk &ch, int s, lfamd_rsrc r, uint32_t off, int gsel, int slot, int hrow) {
        ch.q0[s] = buf_ld16_nt(r, off + (2 * gsel + 0) * 1024 + slot * 16);
        ch.q1[s] = buf_ld16_nt(r, off + (2 * gsel + 1) * 1024 + slot * 16);
        ch.hd[s] = buf_ld8(r, off + P4K_HDR + hrow * 16 + gsel * 8); // scales of blocks 4 gsel .. 4 gsel + 3
    }
    __device__ static inline float dot(const chunk &ch, int s, const uint8_t *xb, int gsel, int h) {
        const uint4 q0 = ch.q0[s], q1 = ch.q1[s];
        const uint2 hd = ch.hd[s];
        const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const uint4 *yq = (const uint4 *)(xb + 128 * gsel + 64 * h);
        const uint4 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3];
        const uint32_t yw[16] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w,
                                 yc.x, yc.y, yc.z, yc.w, yd.x, yd.y, yd.z, yd.w};
        const uint2 psw = *(const uint2 *)(xb + XBLK_PS + 16 * gsel + 8 * h);
        const int ps[4] = {(int)(int16_t)(psw.x & 0xffff), (int)(int16_t)(psw.x >> 16), (int)(int16_t)(psw.y & 0xffff),
                           (int)(int16_t)(psw.y >> 16)};
        const float4 d8 = *(const float4 *)(xb + XBLK_D + 16 * gsel);
        const float d8v[4] = {d8.x, d8.y, d8.z, d8.w};
        float acc = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) { // block 4 gsel + jj
            int isum = 0;
#pragma unroll
            for (int d2 = 0; d2 < 2; d2++) {
                const int t8 = 2 * jj + d2;
                const uint32_t x = qw[t8];
                isum = sdot4(x & 0x0F0F0F0F, yw[2 * t8], isum);
                isum = sdot4((x >> 4) & 0x0F0F0F0F, yw[2 * t8 + 1], isum);
            }
            const uint32_t dw = jj < 2 ? hd.x : hd.y;
            const float d = h2f((uint16_t)((jj & 1) ? (dw >> 16) : (dw & 0xffff)));
            acc = fmaf(d * d8v[jj], (float)(isum - 8 * ps[jj]), acc);
        }
        return acc;
    }
};

// Q5_K: Q4_K with a fifth bit per weight (DequantizerQ5K, iqk_mul_mat.inc:496-511): codes 0..31, same scales / mins.
struct q5k_traits {
    static constexpr int ACT = LFAMD_TYPE_Q8_K; // activation quantisation the reference uses for this type
    static constexpr int TILE = P5K_TILE;
    struct chunk {
        uint4 q0[GEMV_CH_MAX], q1[GEMV_CH_MAX], hd[GEMV_CH_MAX];
        uint2 hq[GEMV_CH_MAX];
    };
    __device__ static inline void load(chunk &ch, int s, lfamd_rsrc r, uint32_t off, int gsel, int slot, int hrow) {
        ch.q0[s] = buf_ld16_nt(r, off + (2 * gsel + 0) * 1024 + slot * 16);
        ch.q1[s] = buf_ld16_nt(r, off + (2 * gsel + 1) * 1024 + slot * 16);
        ch.hd[s] = buf_ld16_nt(r, off + P5K_HDR + hrow * 16);
        ch.hq[s] = buf_ld8(r, off + P5K_QH + slot * 16 + gsel * 8); // fifth bits of groups 2 gsel, 2 gsel + 1
    }
    __device__ static inline float dot(const chunk &ch, int s, const uint8_t *xb, int gsel, int h) {
        const uint4 q0 = ch.q0[s], q1 = ch.q1[s], hd = ch.hd[s];
        const uint32_t hw[2] = {ch.hq[s].x, ch.hq[s].y};
        const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
        uint32_t sc03, sc47, mn03, mn47;
        q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
        const uint32_t scw = gsel ? sc47 : sc03, mnw = gsel ? mn47 : mn03;
        const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const uint4 *yq = (const uint4 *)(xb + 128 * gsel + 64 * h);
        const uint4 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3];
        const uint32_t yw[16] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w,
                                 yc.x, yc.y, yc.z, yc.w, yd.x, yd.y, yd.z, yd.w};
        const uint2 psw = *(const uint2 *)(xb + XBLK_PS + 16 * gsel + 8 * h);
        const int ps[4] = {(int)(int16_t)(psw.x & 0xffff), (int)(int16_t)(psw.x >> 16), (int)(int16_t)(psw.y & 0xffff),
                           (int)(int16_t)(psw.y >> 16)};
        int sumi = 0, summ = 0;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) { // jj = 2 gi + e
            int isum = 0;
#pragma unroll
            for (int d2 = 0; d2 < 2; d2++) {
                const int t8 = 2 * jj + d2; // = 4 gi + dd
                const uint32_t x = qw[t8];
                const uint32_t Hd = hw[t8 >> 2] >> (t8 & 3);
                const uint32_t c0 = (x & 0x0F0F0F0F) | (Hd & 0x10101010);
                const uint32_t c1 = ((x >> 4) & 0x0F0F0F0F) | ((Hd >> 4) & 0x00100010) | ((Hd << 12) & 0x10001000);
                isum = sdot4(c0, yw[2 * t8], isum);
                isum = sdot4(c1, yw[2 * t8 + 1], isum);
            }
            sumi += (int)((scw >> (8 * jj)) & 0xff) * isum;
            summ += (int)((mnw >> (8 * jj)) & 0xff) * ps[jj];
        }
        const float d8 = *(const float *)(xb + XBLK_D);
        return fmaf(d * d8, (float)sumi, -(dmin * d8) * (float)summ);
    }
};

// Q6_K: sub-blocks are 16 wide (one per K-step), codes are 6 bit, offset -32 handled as
// sum sc*(dot(code,q8) - 32*sum(q8)) like DequantizerQ6K (iqk_mul_mat.inc:570-599).
struct q6k_traits {
    static constexpr int ACT = LFAMD_TYPE_Q8_K; // activation quantisation the reference uses for this type
    static constexpr int TILE = P6K_TILE;
    struct chunk {
        uint4 l0[GEMV_CH_MAX], l1[GEMV_CH_MAX], hq[GEMV_CH_MAX];
        uint2 sc[GEMV_CH_MAX];
        uint32_t d[GEMV_CH_MAX];
    };
    __device__ static inline void load(chunk &ch, int s, lfamd_rsrc r, uint32_t off, int gsel, int slot, int hrow) {
        ch.l0[s] = buf_ld16_nt(r, off + (2 * gsel + 0) * 1024 + slot * 16);
        ch.l1[s] = buf_ld16_nt(r, off + (2 * gsel + 1) * 1024 + slot * 16);
        ch.hq[s] = buf_ld16_nt(r, off + P6K_QH + gsel * 1024 + slot * 16);
        ch.sc[s] = buf_ld8(r, off + P6K_SC + hrow * 16 + gsel * 8); // scales of K-steps 8*gsel..+7
        ch.d[s] = buf_ld2(r, off + P6K_D + hrow * 2);
    }
    __device__ static inline float dot(const chunk &ch, int s, const uint8_t *xb, int gsel, int h) {
        const uint4 l0 = ch.l0[s], l1 = ch.l1[s], hq = ch.hq[s];
        const uint2 scb = ch.sc[s];
        const float d = h2f((uint16_t)ch.d[s]);
        const uint32_t lw[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        const uint32_t hw[4] = {hq.x, hq.y, hq.z, hq.w}; // [gi*2 + e]
        const uint4 *yq = (const uint4 *)(xb + 128 * gsel + 64 * h);
        const uint4 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3];
        const uint32_t yw[16] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w,
                                 yc.x, yc.y, yc.z, yc.w, yd.x, yd.y, yd.z, yd.w};
        const uint4 hbw = *(const uint4 *)(xb + XBLK_HB + 32 * gsel + 16 * h);
        const uint32_t hbv[4] = {hbw.x, hbw.y, hbw.z, hbw.w};
        int sumi = 0;
#pragma unroll
        for (int t8 = 0; t8 < 8; t8++) { // t8 = 4 gi + dd
            const uint32_t x = lw[t8];
            uint32_t H = hw[t8 >> 1];
            if (t8 & 1)
                H >>= 2;
            // lo bytes (j0,j4,j1,j5): high fields at bits 4-5 of each byte already
            const uint32_t clo = (x & 0x0F0F0F0F) | (H & 0x30303030);
            // hi bytes (j2,j6,j3,j7): fields at bits 8-9 / 0-1 / 24-25 / 16-17
            const uint32_t chi = ((x >> 4) & 0x0F0F0F0F) | ((H >> 4) & 0x00300030) | ((H << 12) & 0x30003000);
            int isum = sdot4(clo, yw[2 * t8], 0);
            isum = sdot4(chi, yw[2 * t8 + 1], isum);
            const int hs = (int)(int16_t)((hbv[t8 >> 1] >> (16 * (t8 & 1))) & 0xffff);
            const int sc = (int)(int8_t)(((t8 < 4 ? scb.x : scb.y) >> (8 * (t8 & 3))) & 0xff);
            sumi += sc * (isum - 32 * hs);
        }
        const float d8 = *(const float *)(xb + XBLK_D);
        return (d * d8) * (float)sumi;
    }
};

// Q4_1 / Q5_0 / Q5_1 on the PCL image (rows of whole 256-weight groups): the P4K nibble lattice, eight f16 d and eight
// f16 m per row, fifth bits on the P5K lattice.  Per 32-block (iqk_mul_mat.inc:1241-1349, ggml_vec_dot_q5_0_q8_0):
//   Q5_0: d*d8*(<q5, q8> - 16*sum(q8))          Q4_1 / Q5_1: d*d8*<q, q8> + m*s,  s = the activation block's d8*sum(q8)
// A lane holds half of a block's weights (K-half h): the m*s term is added by the h = 0 lane.
template <int TYPE>
struct pcl_traits {
    static constexpr bool HAS_M = TYPE == LFAMD_TYPE_Q4_1 || TYPE == LFAMD_TYPE_Q5_1;
    static constexpr bool HAS_H = TYPE == LFAMD_TYPE_Q5_0 || TYPE == LFAMD_TYPE_Q5_1;
    static constexpr int ACT = HAS_M ? LFAMD_TYPE_Q8_1 : LFAMD_TYPE_Q8_0;
    static constexpr int TILE = PCL_TILE;
    struct chunk {
        uint4 q0[GEMV_CH_MAX], q1[GEMV_CH_MAX];
        uint2 hd[GEMV_CH_MAX], hm[GEMV_CH_MAX], hq[GEMV_CH_MAX];
    };
    __device__ static inline void load(chunk &ch, int s, lfamd_rsrc r, uint32_t off, int gsel, int slot, int hrow) {
        ch.q0[s] = buf_ld16_nt(r, off + (2 * gsel + 0) * 1024 + slot * 16);
        ch.q1[s] = buf_ld16_nt(r, off + (2 * gsel + 1) * 1024 + slot * 16);
        ch.hd[s] = buf_ld8(r, off + PCL_D + hrow * 16 + gsel * 8); // scales of blocks 4 gsel .. 4 gsel + 3
        if constexpr (HAS_M)
            ch.hm[s] = buf_ld8(r, off + PCL_M + hrow * 16 + gsel * 8);
        if constexpr (HAS_H)
            ch.hq[s] = buf_ld8(r, off + PCL_QH + slot * 16 + gsel * 8); // fifth bits of groups 2 gsel, 2 gsel + 1
    }
    __device__ static inline float dot(const chunk &ch, int s, const uint8_t *xb, int gsel, int h) {
        const uint4 q0 = ch.q0[s], q1 = ch.q1[s];
        const uint2 hd = ch.hd[s];
        const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const uint4 *yq = (const uint4 *)(xb + 128 * gsel + 64 * h);
        const uint4 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3];
        const uint32_t yw[16] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w,
                                 yc.x, yc.y, yc.z, yc.w, yd.x, yd.y, yd.z, yd.w};
        const uint4 dsw = *(const uint4 *)(xb + XBLK_D + 16 * gsel); // Q8_0: four f32 d8; Q8_1: four {f16 d8, f16 s}
        const uint32_t dsv[4] = {dsw.x, dsw.y, dsw.z, dsw.w};
        int ps[4] = {0, 0, 0, 0};
        if constexpr (!HAS_M) {
            const uint2 psw = *(const uint2 *)(xb + XBLK_PS + 16 * gsel + 8 * h);
            ps[0] = (int)(int16_t)(psw.x & 0xffff), ps[1] = (int)(int16_t)(psw.x >> 16);
            ps[2] = (int)(int16_t)(psw.y & 0xffff), ps[3] = (int)(int16_t)(psw.y >> 16);
        }
        float acc = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) { // block 4 gsel + jj
            int isum = 0;
#pragma unroll
            for (int d2 = 0; d2 < 2; d2++) {
                const int t8 = 2 * jj + d2;
                const uint32_t x = qw[t8];
                uint32_t c0 = x & 0x0F0F0F0F, c1 = (x >> 4) & 0x0F0F0F0F;
                if constexpr (HAS_H) {
                    const uint32_t Hd = (t8 < 4 ? ch.hq[s].x : ch.hq[s].y) >> (t8 & 3);
                    c0 |= Hd & 0x10101010;
                    c1 |= ((Hd >> 4) & 0x00100010) | ((Hd << 12) & 0x10001000);
                }
                isum = sdot4(c0, yw[2 * t8], isum);
                isum = sdot4(c1, yw[2 * t8 + 1], isum);
            }
            const uint32_t dw = jj < 2 ? hd.x : hd.y;
            const float d = h2f((uint16_t)((jj & 1) ? (dw >> 16) : (dw & 0xffff)));
            if constexpr (HAS_M) {
                const uint32_t mw = jj < 2 ? ch.hm[s].x : ch.hm[s].y;
                const float m = h2f((uint16_t)((jj & 1) ? (mw >> 16) : (mw & 0xffff)));
                const float d8 = h2f((uint16_t)(dsv[jj] & 0xffff)), s8 = h2f((uint16_t)(dsv[jj] >> 16));
                acc = fmaf(d * d8, (float)isum, acc);
                acc = fmaf(m, h ? 0.0f : s8, acc);
            } else {
                acc = fmaf(d * __builtin_bit_cast(float, dsv[jj]), (float)(isum - 16 * ps[jj]), acc);
            }
        }
        return acc;
    }
};

// Q2_K / Q3_K on the RESIDENT compact images (lfamd_device.h: PK2 / PK3): half the bytes of the canonical image per row; two
// K-steps come out of one code dword with an AND and a shift-AND, Q3_K's third bit with a shift, an AND and a shift-OR.
// Per 16-wide sub-block: sc * (<q + OFF, y> - OFF * sum(y)) with the staged group sums; Q2_K's mins against the same sums
// (iqk_mul_mat.inc:432-470, 533-568).
template <int TYPE>
struct pk_traits {
    static constexpr int ACT = LFAMD_TYPE_Q8_K;
    static constexpr bool Q3 = TYPE == LFAMD_TYPE_Q3_K;
    static constexpr int TILE = Q3 ? PK3_TILE : PK2_TILE;
    static constexpr bool MINS = !Q3;
    static constexpr int OFF = Q3 ? 4 : 0;
    struct chunk {
        uint4 q[GEMV_CH_MAX];
        uint2 hb[GEMV_CH_MAX], sc[GEMV_CH_MAX];
        uint32_t dd[GEMV_CH_MAX];
    };
    __device__ static inline void load(chunk &ch, int s, lfamd_rsrc r, uint32_t off, int gsel, int slot, int hrow) {
        ch.q[s] = buf_ld16_nt(r, off + gsel * 1024 + slot * 16);
        if constexpr (Q3)
            ch.hb[s] = buf_ld8(r, off + PK3_HB + gsel * 512 + slot * 8);
        ch.sc[s] = buf_ld8(r, off + (Q3 ? PK3_SC : PK2_SC) + hrow * 16 + gsel * 8); // scale bytes of K-steps 8*gsel..+7
        ch.dd[s] = __builtin_amdgcn_raw_buffer_load_b32(r, off + (Q3 ? PK3_D : PK2_D) + hrow * 4, 0, 0); // {d, dmin}
    }
    __device__ static inline float dot(const chunk &ch, int s, const uint8_t *xb, int gsel, int h) {
        const uint4 qc = ch.q[s];
        const uint2 scb = ch.sc[s];
        const uint32_t dw = ch.dd[s];
        const float d = h2f((uint16_t)(dw & 0xffff));
        const uint32_t cw[4] = {qc.x, qc.y, qc.z, qc.w};
        const uint4 *yq = (const uint4 *)(xb + 128 * gsel + 64 * h);
        const uint4 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3];
        const uint32_t yw[16] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w,
                                 yc.x, yc.y, yc.z, yc.w, yd.x, yd.y, yd.z, yd.w};
        const uint4 hbw = *(const uint4 *)(xb + XBLK_HB + 32 * gsel + 16 * h);
        const uint32_t hbv[4] = {hbw.x, hbw.y, hbw.z, hbw.w};
        int sumi = 0, summ = 0;
#pragma unroll
        for (int t8 = 0; t8 < 8; t8++) { // t8 = 4 gi + dd
            uint32_t x = ((t8 & 1) ? (cw[t8 >> 1] >> 2) : cw[t8 >> 1]) & 0x33333333u;
            if constexpr (Q3) {
                const uint32_t H = (t8 < 4 ? ch.hb[s].x : ch.hb[s].y) >> (t8 & 3);
                x |= (H & 0x11111111u) << 2;
            }
            int isum = sdot4(x & 0x0F0F0F0F, yw[2 * t8], 0);
            isum = sdot4((x >> 4) & 0x0F0F0F0F, yw[2 * t8 + 1], isum);
            const int hs = (int)(int16_t)((hbv[t8 >> 1] >> (16 * (t8 & 1))) & 0xffff);
            const uint32_t sbyte = ((t8 < 4 ? scb.x : scb.y) >> (8 * (t8 & 3))) & 0xff;
            if constexpr (Q3) {
                sumi += (int)(int8_t)sbyte * (isum - OFF * hs);
            } else {
                sumi += (int)(sbyte & 15) * isum;
                summ += (int)(sbyte >> 4) * hs;
            }
        }
        const float d8 = *(const float *)(xb + XBLK_D);
        if constexpr (MINS)
            return fmaf(d * d8, (float)sumi, -(h2f((uint16_t)(dw >> 16)) * d8) * (float)summ);
        return (d * d8) * (float)sumi;
    }
};

// IQ4_XS on the RESIDENT compact image (generic.hip: pk4x_pack_kernel): codebook indices on the P4K nibble lattice, so a dword's
// nibbles already sit in the order of the staged activation codes; the 16-entry int8 codebook (kvalues_iq4nl,
// iqk_mul_mat.inc:601-628 looks it up with a byte shuffle too) is four registers and three v_perm per four codes: the lower
// and the upper eight entries by the index's low three bits, then byte i of one or the other by its bit 3.
struct iq4c_traits {
    static constexpr int ACT = LFAMD_TYPE_Q8_K;
    static constexpr int TILE = P4K_TILE;
    struct chunk {
        uint4 q0[GEMV_CH_MAX], q1[GEMV_CH_MAX], hd[GEMV_CH_MAX];
    };
    __device__ static inline void load(chunk &ch, int s, lfamd_rsrc r, uint32_t off, int gsel, int slot, int hrow) {
        ch.q0[s] = buf_ld16_nt(r, off + (2 * gsel + 0) * 1024 + slot * 16);
        ch.q1[s] = buf_ld16_nt(r, off + (2 * gsel + 1) * 1024 + slot * 16);
        ch.hd[s] = buf_ld16(r, off + P4K_HDR + hrow * 16);
    }
    __device__ static inline uint32_t lut4(uint32_t n) { // four indices (one per byte) -> four codebook values
        constexpr uint32_t T0 = 0xBFAD9881u, T1 = 0xF6EADDCFu, T2 = 0x26190D01u, T3 = 0x71594535u; // kvalues_iq4nl, 4 entries each
        const uint32_t sel = n & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(T1, T0, sel), hi = __builtin_amdgcn_perm(T3, T2, sel);
        return __builtin_amdgcn_perm(hi, lo, ((n >> 1) & 0x04040404u) | 0x03020100u);
    }
    __device__ static inline float dot(const chunk &ch, int s, const uint8_t *xb, int gsel, int h) {
        const uint4 q0 = ch.q0[s], q1 = ch.q1[s], hd = ch.hd[s];
        const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const uint4 *yq = (const uint4 *)(xb + 128 * gsel + 64 * h);
        const uint4 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3];
        const uint32_t yw[16] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w,
                                 yc.x, yc.y, yc.z, yc.w, yd.x, yd.y, yd.z, yd.w};
        const uint32_t scw = gsel ? hd.y : hd.x; // scales of sub-blocks 4 gsel .. 4 gsel + 3
        int sumi = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) { // sub-block 4 gsel + u = K-steps t8 = 2u, 2u + 1 of this lane
            int isum = 0;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const uint32_t x = qw[2 * u + e];
                isum = sdot4(lut4(x & 0x0F0F0F0Fu), yw[2 * (2 * u + e)], isum);
                isum = sdot4(lut4((x >> 4) & 0x0F0F0F0Fu), yw[2 * (2 * u + e) + 1], isum);
            }
            sumi += (int)(int8_t)((scw >> (8 * u)) & 0xff) * isum;
        }
        const float d8 = *(const float *)(xb + XBLK_D);
        return (h2f((uint16_t)(hd.z & 0xffff)) * d8) * (float)sumi;
    }
};

// Up to GEMV_MAX_MATS weight matrices of one type and row length that consume the SAME activations
// (attn_q/k/v, ffn_gate/up) are served by one launch: their half-tiles are concatenated.
#define GEMV_MAX_MATS 4
struct gemv_mats {
    const uint8_t *A[GEMV_MAX_MATS];
    float *C[GEMV_MAX_MATS];
    long m[GEMV_MAX_MATS];
    long ldc[GEMV_MAX_MATS];
    int ht_end[GEMV_MAX_MATS]; // exclusive prefix of half-tile counts
    int count;
    // GGML_OP_MUL_MAT_ID at decode (IDS kernels only): matrix j is expert ids[id_idx[j]] of the stack at A[j]
    const int32_t *ids;
    long expert_bytes;
    int id_idx[GEMV_MAX_MATS];
    int experts;
};

#ifndef GEMV_DIAG
#define GEMV_DIAG 0
#endif
#if GEMV_DIAG // development: in-kernel s_memtime stamps of two work-groups (never in the product build)
static __device__ unsigned long long g_gemv_stamps[4 * 16 * 16];
extern "C" __attribute__((weak)) int lfamd_debug_gemv_stamps(unsigned long long *dst) { // per TU; dev only
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemv_stamps), sizeof(g_gemv_stamps));
}
// entry / exit time and placement (HW_ID, XCC_ID) of wave 0 of every work-group
static __device__ unsigned long long g_gemv_wg[512 * 4];
extern "C" __attribute__((weak)) int lfamd_debug_gemv_wgs(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemv_wg), sizeof(g_gemv_wg));
}
#define GWG(slot)                                                                                                \
    do {                                                                                                         \
        if (blockIdx.x < 512 && threadIdx.x == 0) {                                                              \
            g_gemv_wg[blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memrealtime();                               \
            if ((slot) == 0) {                                                                                   \
                g_gemv_wg[blockIdx.x * 4 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |  \
                                                ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); \
                gwg_clk0 = __builtin_amdgcn_s_memtime();                                                         \
            } else { /* shader-clock cycles of this work-group's life: with the 100 MHz stamps, the clock it ran at */ \
                g_gemv_wg[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - gwg_clk0;                         \
            }                                                                                                    \
        }                                                                                                        \
    } while (0)
#define GSTAMP()                                                                                                 \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if ((blockIdx.x == 0 || blockIdx.x == 100) && lane == 0 && stamp_n < 16)                                  \
            g_gemv_stamps[((blockIdx.x ? 1 : 0) * 16 + wave) * 16 + stamp_n++] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
// (diagnostic only: drain the loads first, so the stamp is the arrival time of the item's weights)
#define GSTAMP_ARRIVAL()                                                                                         \
    do {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
        GSTAMP();                                                                                                \
    } while (0)
#else
#define GSTAMP()
#define GSTAMP_ARRIVAL()
#define GWG(slot)
#endif

// ---------------------------------------------------------------------------------------------
// Decode body (ONE activation row).  What the stamps of the body above showed on 4096 x 4096 (tools/gemv_stamps.py):
// 0.8 us of dependent kernel-argument rounds before the first load, then 1.4 us in which four of the sixteen waves
// fetch the row as 64-byte-strided pieces (32 cache-line lookups per wave instruction — more L1 work than the whole
// weight stream of the CU), quantise it for everybody and release a work-group barrier.  Here instead:
//   * every kernel argument is read with static indices (one round of s_load, selects instead of indexed reads);
//   * wave w quantises exactly the super-blocks it will consume (w, w + NW, ...): one coalesced float4 per lane and
//     block (1 KiB per wave instruction), the block maximum by DPP, the LDS image written with one dword per lane;
//     a wave only ever reads image blocks it wrote itself (LDS operations of one wave execute in order), so there
//     is NO work-group barrier before the dots;
//   * the activation loads go out first and the weight loads right behind them (vmcnt retires in order);
//   * no integer division: tiles and chunks are walked with cursors.
// Arithmetic (per-lane dot, wave reduction, fixed-order sum over the waves) is the body above's: results are
// bit-identical.

// lane l holds codes y = (c[4l] .. c[4l+3]) of one 256-code block: write the code image, the group sums and the pair
// sums (layout: XBLK above).  Two lanes form an 8-code group.
__device__ static inline void put_codes_wave(uint8_t *dst, uint32_t y, int lane) {
    const int hf = lane & 1, grp = lane >> 1;
    const uint32_t other = dpp_u32<DPP_XOR1>(y);
    const uint32_t lo = hf ? other : y, hi = hf ? y : other;
    const uint32_t word = hf ? __builtin_amdgcn_perm(hi, lo, 0x07030602) : __builtin_amdgcn_perm(hi, lo, 0x05010400);
    const int pos = (grp & 16) | ((grp & 1) << 3) | ((grp & 8) >> 1) | ((grp >> 1) & 3);
    *(uint32_t *)(dst + 8 * pos + 4 * hf) = word;
    const int s4 = sdot4(y, 0x01010101u, 0);
    const int hs = s4 + (int)dpp_u32<DPP_XOR1>((uint32_t)s4);
    if (hf == 0)
        *(int16_t *)(dst + XBLK_HB + 2 * pos) = (int16_t)hs;
    const int hs2 = hs + (int)dpp_u32<DPP_ROW_SHL4>((uint32_t)hs); // group grp + 2 = lane + 4 (the pair's odd K-step)
    if ((lane & 5) == 0)
        put_pair(dst, grp, hs2);
}

// quantize_row_q8_K on one block held as a float4 per lane (element 4 lane + e): first index of the largest |x| gives
// the sign of iscale = -128/max; codes nearest_int (half-even) clamped at 127; d = 1/iscale.
__device__ static inline void stage_f32_q8k_wave(uint8_t *dst, const float4 v, int lane) {
    const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
    float am = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
    am = fmaxf(am, dpp_f32<DPP_XOR1>(am));
    am = fmaxf(am, dpp_f32<DPP_XOR2>(am));
    am = fmaxf(am, dpp_f32<DPP_HALF_MIRROR>(am));
    am = fmaxf(am, dpp_f32<DPP_MIRROR>(am));
    const float amax = fmaxf(fmaxf(readlane_f32(am, 0), readlane_f32(am, 16)), fmaxf(readlane_f32(am, 32), readlane_f32(am, 48)));
    // (branch-free: an all-zero block goes through the same arithmetic with a harmless divisor and is zeroed by selects)
    const bool m0 = a0 == amax, m1 = a1 == amax, m2 = a2 == amax, m3 = a3 == amax;
    const unsigned long long ball = __builtin_amdgcn_ballot_w64(m0 || m1 || m2 || m3);
    const float cand = m0 ? v.x : (m1 ? v.y : (m2 ? v.z : v.w));
    const int first = ball ? __builtin_ctzll(ball) : 0;
    const bool nz = amax != 0.0f;
    const float val = nz ? readlane_f32(cand, first) : 1.0f;
    const float iscale = -128.0f / val;
    int q0 = (int)rintf(iscale * v.x), q1 = (int)rintf(iscale * v.y), q2 = (int)rintf(iscale * v.z), q3 = (int)rintf(iscale * v.w);
    q0 = q0 > 127 ? 127 : q0, q1 = q1 > 127 ? 127 : q1, q2 = q2 > 127 ? 127 : q2, q3 = q3 > 127 ? 127 : q3;
    uint32_t y = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
    y = nz ? y : 0u;
    const float d = nz ? 1.0f / iscale : 0.0f;
    put_codes_wave(dst, y, lane);
    if (lane == 0)
        *(float *)(dst + XBLK_D) = d;
}

// quantize_row_q8_0 on eight 32-blocks held as a float4 per lane (8 lanes per block): d = amax/127 (stored as f16),
// q = roundf(x / d).
template <bool S1>
__device__ static inline void stage_f32_q80_wave(uint8_t *dst, const float4 v, int lane) {
    float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    am = fmaxf(am, dpp_f32<DPP_XOR1>(am));
    am = fmaxf(am, dpp_f32<DPP_XOR2>(am));
    am = fmaxf(am, dpp_f32<DPP_HALF_MIRROR>(am));
    const float d = am / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    const int q0 = (int)roundf(v.x * id), q1 = (int)roundf(v.y * id), q2 = (int)roundf(v.z * id), q3 = (int)roundf(v.w * id);
    const uint32_t y = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
    put_codes_wave(dst, y, lane);
    int sum = 0;
    if constexpr (S1) { // the block's code sum over its eight lanes
        sum = sdot4(y, 0x01010101u, 0);
        sum += (int)dpp_u32<DPP_XOR1>((uint32_t)sum);
        sum += (int)dpp_u32<DPP_XOR2>((uint32_t)sum);
        sum += (int)dpp_u32<DPP_HALF_MIRROR>((uint32_t)sum);
    }
    if ((lane & 7) == 0)
        put_scale_q80<S1>(dst, lane >> 3, d, sum);
}

// Two blocks per pass (deep rows: a wave owns several super-blocks): lane l holds the EIGHT values 8 (l & 31) + e of
// block (l >> 5) — one whole 8-code group — so the fixed part of the quantiser (DPP maximum, the two IEEE divisions, the
// image addressing) is paid once per two blocks: ~70 VALU per block instead of ~120.  `dst` is this lane's block image.
__device__ static inline void stage_f32_q8k_wave2(uint8_t *dst, const float4 va, const float4 vb, int lane) {
    const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; e++)
        a[e] = fabsf(v[e]);
    float am = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7])));
    am = fmaxf(am, dpp_f32<DPP_XOR1>(am));
    am = fmaxf(am, dpp_f32<DPP_XOR2>(am));
    am = fmaxf(am, dpp_f32<DPP_HALF_MIRROR>(am));
    am = fmaxf(am, dpp_f32<DPP_MIRROR>(am));
    const bool hi = lane >= 32;
    const float amax_lo = fmaxf(readlane_f32(am, 0), readlane_f32(am, 16)), amax_hi = fmaxf(readlane_f32(am, 32), readlane_f32(am, 48));
    const float amax = hi ? amax_hi : amax_lo;
    bool any = false;
    float cand = v[7];
#pragma unroll
    for (int e = 7; e >= 0; e--) { // the FIRST element that reaches the maximum wins
        const bool me = a[e] == amax;
        cand = me ? v[e] : cand;
        any = any || me;
    }
    const unsigned long long ball = __builtin_amdgcn_ballot_w64(any);
    const uint32_t blo = (uint32_t)ball, bhi = (uint32_t)(ball >> 32);
    const int first_lo = blo ? __builtin_ctz(blo) : 0, first_hi = 32 + (bhi ? __builtin_ctz(bhi) : 0);
    const float val_lo = readlane_f32(cand, first_lo), val_hi = readlane_f32(cand, first_hi);
    const bool nz = amax != 0.0f;
    const float val = nz ? (hi ? val_hi : val_lo) : 1.0f;
    const float iscale = -128.0f / val;
    uint32_t y[2] = {0, 0};
#pragma unroll
    for (int e = 0; e < 8; e++) {
        int q = (int)rintf(iscale * v[e]);
        q = q > 127 ? 127 : q;
        y[e >> 2] |= (uint32_t)(q & 0xff) << (8 * (e & 3));
    }
    const uint32_t y0 = nz ? y[0] : 0u, y1 = nz ? y[1] : 0u;
    const float d = nz ? 1.0f / iscale : 0.0f;
    const int grp = lane & 31;
    const int hs = put_group(dst, grp, y0, y1);
    const int other = (int)dpp_u32<DPP_XOR2>((uint32_t)hs); // group grp ^ 2
    if ((grp & 2) == 0)
        put_pair(dst, grp, hs + other);
    if (grp == 0)
        *(float *)(dst + XBLK_D) = d;
}

template <bool S1>
__device__ static inline void stage_f32_q80_wave2(uint8_t *dst, const float4 va, const float4 vb, int lane) {
    const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
    float am = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; e++)
        am = fmaxf(am, fabsf(v[e]));
    am = fmaxf(am, dpp_f32<DPP_XOR1>(am)); // four lanes per 32-block
    am = fmaxf(am, dpp_f32<DPP_XOR2>(am));
    const float d = am / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    uint32_t y[2] = {0, 0};
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int q = (int)roundf(v[e] * id);
        y[e >> 2] |= (uint32_t)(q & 0xff) << (8 * (e & 3));
    }
    const int grp = lane & 31;
    const int hs = put_group(dst, grp, y[0], y[1]);
    const int other = (int)dpp_u32<DPP_XOR2>((uint32_t)hs);
    if ((grp & 2) == 0)
        put_pair(dst, grp, hs + other);
    const int pair = hs + other; // groups grp, grp ^ 2; with the neighbour pair: the four groups of the 32-block
    const int sum = pair + (int)dpp_u32<DPP_XOR1>((uint32_t)pair); // (all lanes active: a DPP source lane must be)
    if ((grp & 3) == 0)
        put_scale_q80<S1>(dst, grp >> 2, d, sum);
}

// one block of already-quantised activations, staged by one wave (lanes 0..31: one 8-code group each)
template <int ACT>
__device__ static inline void stage_quantised_wave(uint8_t *dst, const uint8_t *row_generic, int b, int lane) {
    // (explicitly GLOBAL: the pointer went through an SGPR pin and would otherwise be read with FLAT loads, which make
    // hipcc's wait counts conservative for the whole kernel — tests/test_isa_hazards.py)
    typedef const __attribute__((address_space(1))) uint8_t *gptr;
    const gptr row = (gptr)row_generic;
    if (lane < 32) {
        const int grp = lane;
        uint32_t y0, y1;
        if constexpr (ACT == LFAMD_TYPE_Q8_K) {
            const gptr y = row + (size_t)b * 292;
            const __attribute__((address_space(1))) uint32_t *src = (const __attribute__((address_space(1))) uint32_t *)(y + 36 + 8 * grp);
            y0 = src[0], y1 = src[1];
            if (grp == 0)
                *(float *)(dst + XBLK_D) = *(const __attribute__((address_space(1))) float *)y;
        } else {
            constexpr bool S1 = ACT == LFAMD_TYPE_Q8_1;
            const gptr blk = row + (size_t)(b * 8 + (grp >> 2)) * (S1 ? 36 : 34);
            const __attribute__((address_space(1))) uint16_t *src = (const __attribute__((address_space(1))) uint16_t *)(blk + (S1 ? 4 : 2) + 8 * (grp & 3));
            y0 = (uint32_t)src[0] | ((uint32_t)src[1] << 16), y1 = (uint32_t)src[2] | ((uint32_t)src[3] << 16);
            if ((grp & 3) == 0) {
                const __attribute__((address_space(1))) uint16_t *hdr = (const __attribute__((address_space(1))) uint16_t *)blk;
                if constexpr (S1)
                    *(uint32_t *)(dst + XBLK_D + 4 * (grp >> 2)) = (uint32_t)hdr[0] | ((uint32_t)hdr[1] << 16);
                else
                    *(float *)(dst + XBLK_D + 4 * (grp >> 2)) = h2f(hdr[0]);
            }
        }
        const int hs = put_group(dst, grp, y0, y1);
        const int other = (int)dpp_u32<DPP_XOR2>((uint32_t)hs); // group grp ^ 2
        if ((grp & 2) == 0)
            put_pair(dst, grp, hs + other);
    }
}

// The body of a GEMV work-group: work-group `bid` of `gdim` over the half-tiles of `mats` (the plain kernel passes its
// block index and grid size; the two-type kernel gives each type its own sub-grid).
template <typename TR, int NC, int BT, int NW, int GEMV_CH, bool IDS>
__device__ __forceinline__ void gemv_kq_body(const gemv_mats &mats, int nb, const uint8_t *__restrict__ B, size_t b_row_bytes,
                                             long col0, int n_ht, const int bid, const int gdim, uint8_t *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, h = (lane >> 4) & 1, gsel = lane >> 5;
    float *red = (float *)(lds + (size_t)NC * nb * XBLK); // [2][NW][NC][16]
#if GEMV_DIAG
    int stamp_n = 0;
#endif
    GSTAMP();

    const int sb_per_wave = (nb + NW - 1) / NW;
    const int cpt = (sb_per_wave + GEMV_CH - 1) / GEMV_CH; // chunks per tile
    const int ntile = (n_ht - bid + gdim - 1) / gdim;
    const int total = ntile * cpt;

    // Loads are issued UNCONDITIONALLY through a bounds-checked buffer descriptor: a branch around a load
    // makes hipcc fall back to s_waitcnt vmcnt(0) (the prefetch of the next item is lost), whereas a
    // descriptor with zero records ("no item f") or an offset past the row-tile ("no super-block b")
    // returns zeros without touching memory and keeps the counted waits exact.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t rt_bytes = (uint32_t)nb * TR::TILE;
    auto issue = [&](typename TR::chunk &ch, int f) {
        const int tile_i = f / cpt, chunk_i = f - tile_i * cpt;
        long ht = (long)bid + (long)tile_i * gdim;
        int j = 0;
#pragma unroll
        for (int jj = 1; jj < GEMV_MAX_MATS; jj++)
            if (jj < mats.count && ht >= mats.ht_end[jj - 1])
                j = jj;
        const uint8_t *A = mats.A[j];
        bool have = f < total;
        if constexpr (IDS) { // expert picked on the device: no routing-table read-back, graph-capturable
            const int ex = mats.ids[mats.id_idx[j]];
            const bool ok = ex >= 0 && ex < mats.experts;
            A += (size_t)(ok ? ex : 0) * mats.expert_bytes;
            have = have && ok;
        }
        if (j > 0)
            ht -= mats.ht_end[j - 1];
        const int hh = (int)(ht & 1);
        const lfamd_rsrc r = make_rsrc(A + (size_t)(ht >> 1) * rt_bytes, have ? rt_bytes : 0u);
        const int slot = h * 32 + hh * 16 + i16, hrow = hh * 16 + i16;
#pragma unroll
        for (int s = 0; s < GEMV_CH; s++) {
            const int b = wave_u + NW * (chunk_i * GEMV_CH + s); // b >= nb lands past the descriptor: zeros
            TR::load(ch, s, r, (uint32_t)b * TR::TILE, gsel, slot, hrow);
        }
    };

    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
        acc[c] = 0.0f;

    auto consume = [&](const typename TR::chunk &ch, int f) {
        const int tile_i = f / cpt, chunk_i = f - tile_i * cpt;
#pragma unroll
        for (int s = 0; s < GEMV_CH; s++) {
            const int b = wave + NW * (chunk_i * GEMV_CH + s);
            const int bc = b < nb ? b : nb - 1;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                // a super-block beyond the row was loaded as zeros (d = 0): contributes exactly 0
                acc[c] += TR::dot(ch, s, lds + (size_t)(c * nb + bc) * XBLK, gsel, h);
            }
        }
        if (chunk_i == cpt - 1) { // tile finished: 4 lanes per row (h, gsel), then the waves through LDS
            float *rb = red + (tile_i & 1) * (NW * NC * 16);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                float v = acc[c];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (lane < 16)
                    rb[(wave * NC + c) * 16 + lane] = v;
                acc[c] = 0.0f;
            }
            GSTAMP();
            __syncthreads();
            GSTAMP();
            if (threadIdx.x < 16 * NC) {
                const int c = threadIdx.x >> 4, i = threadIdx.x & 15;
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; w++)
                    v += rb[(w * NC + c) * 16 + i];
                long ht = (long)bid + (long)tile_i * gdim;
                int j = 0;
#pragma unroll
                for (int jj = 1; jj < GEMV_MAX_MATS; jj++)
                    if (jj < mats.count && ht >= mats.ht_end[jj - 1])
                        j = jj;
                if (j > 0)
                    ht -= mats.ht_end[j - 1];
                const long row = (ht >> 1) * 32 + (ht & 1) * 16 + i;
                bool ok = true;
                if constexpr (IDS) { // an out-of-range expert id leaves its result row untouched (the host path skips it)
                    const int ex = mats.ids[mats.id_idx[j]];
                    ok = ex >= 0 && ex < mats.experts;
                }
                if (row < mats.m[j] && ok)
                    mats.C[j][(col0 + c) * mats.ldc[j] + row] = v;
            }
        }
    };

    typename TR::chunk bufA, bufB;
    if (total <= 0)
        return; // (whole work-group: total is uniform)
    // vmcnt retires in order: whatever is loaded first is waited for first.  In the decode case (one f32
    // row, at most one piece per thread) fetch the activations BEFORE the first weight chunk, so the
    // quantisation below only waits for them and the weights keep flying.
    if (BT == LFAMD_TYPE_F32 && NC == 1 && nb * 16 <= NW * 64) {
        float v[16];
        const bool mine = (int)threadIdx.x < nb * 16;
        if (mine)
            load_piece(v, (const float *)(B + col0 * b_row_bytes), threadIdx.x);
        issue(bufA, 0);
        GSTAMP();
        if (mine) {
            if constexpr (TR::ACT == LFAMD_TYPE_Q8_K)
                quantise_piece_q8k(lds + (size_t)(threadIdx.x >> 4) * XBLK, v, threadIdx.x & 15);
            else
                quantise_piece_q80<TR::ACT == LFAMD_TYPE_Q8_1>(lds + (size_t)(threadIdx.x >> 4) * XBLK, v, threadIdx.x & 15);
        }
        GSTAMP();
    } else if (BT == LFAMD_TYPE_F32 && GEMV_CH <= 2 && nb <= GEMV_CH * NW) {
        // several f32 rows, shallow: wave w quantises the blocks it consumes (w, w + NW) of every column from one
        // coalesced float4 per lane and block (the per-thread 64-byte pieces of stage_x cost 32 cache-line lookups per
        // wave instruction); the loads go out before the weights, blocks past the row arrive as zeros and land in the
        // wave's dummy slot
        uint4 xv[NC][GEMV_CH];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const lfamd_rsrc rx = make_rsrc(B + (col0 + c) * b_row_bytes, (uint32_t)nb * 1024u);
#pragma unroll
            for (int u = 0; u < GEMV_CH; u++)
                xv[c][u] = buf_ld16(rx, (uint32_t)(wave_u + NW * u) * 1024u + (uint32_t)lane * 16u);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue(bufA, 0);
        __builtin_amdgcn_sched_barrier(0);
        uint8_t *dummy = (uint8_t *)(red + 2 * NW * NC * 16) + (size_t)wave * XBLK;
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int u = 0; u < GEMV_CH; u++) {
                const int b = wave + NW * u;
                uint8_t *dst = b < nb ? lds + (size_t)(c * nb + b) * XBLK : dummy;
                const float4 v = make_float4(__builtin_bit_cast(float, xv[c][u].x), __builtin_bit_cast(float, xv[c][u].y),
                                             __builtin_bit_cast(float, xv[c][u].z), __builtin_bit_cast(float, xv[c][u].w));
                if constexpr (TR::ACT == LFAMD_TYPE_Q8_K)
                    stage_f32_q8k_wave(dst, v, lane);
                else
                    stage_f32_q80_wave<TR::ACT == LFAMD_TYPE_Q8_1>(dst, v, lane);
            }
    } else {
        issue(bufA, 0); // in flight during the staging below
        stage_x<BT, TR::ACT>(lds, B, b_row_bytes, col0, NC, nb);
    }
    __syncthreads();
    GSTAMP();

    // (issuing bufB before the staging as well, and re-issuing each buffer right after its consume, measured
    // 10-25 % SLOWER on every decode shape: the counted waits degrade and the activation loads queue behind more
    // weight traffic)
    for (int f = 0; f < total; f += 2) {
        issue(bufB, f + 1);
        consume(bufA, f);
        GSTAMP();
        issue(bufA, f + 2);
        if (f + 1 < total)
            consume(bufB, f + 1);
    }
}


// the matrices of a launch as SSA values: every kernel argument is fetched in ONE round of scalar loads at the top of
// the kernel (the empty asm pins each value in SGPRs there: left alone, hipcc turns a select of two kernel arguments
// into a load from a selected ADDRESS — a second, dependent round trip of ~0.4 us before the first weight load and
// another one before the stores)
struct kq_tab {
    const uint8_t *A[GEMV_MAX_MATS];
    float *C[GEMV_MAX_MATS];
    long m[GEMV_MAX_MATS], ldc[GEMV_MAX_MATS];
    int e[GEMV_MAX_MATS];
    int idx[GEMV_MAX_MATS];
    int cnt;
};

__device__ __forceinline__ kq_tab kq_table(const gemv_mats &mats) {
    kq_tab t;
#pragma unroll
    for (int i = 0; i < GEMV_MAX_MATS; i++) {
        t.A[i] = mats.A[i], t.C[i] = mats.C[i], t.m[i] = mats.m[i], t.ldc[i] = mats.ldc[i], t.e[i] = mats.ht_end[i];
        t.idx[i] = mats.id_idx[i];
        asm volatile("" : "+s"(t.A[i]), "+s"(t.C[i]), "+s"(t.m[i]), "+s"(t.ldc[i]), "+s"(t.e[i]), "+s"(t.idx[i]));
    }
    t.cnt = mats.count;
    asm volatile("" : "+s"(t.cnt));
    return t;
}

struct kq_sel {
    const uint8_t *A;
    float *C;
    long m, ldc;
    int ht; // half-tile inside the matrix
    int idx; // id_idx of the matrix (IDS)
};

__device__ __forceinline__ kq_sel kq_pick(const kq_tab t, int ht) {
    const bool s1 = t.cnt > 1 && ht >= t.e[0], s2 = t.cnt > 2 && ht >= t.e[1], s3 = t.cnt > 3 && ht >= t.e[2];
    kq_sel r;
    r.A = s3 ? t.A[3] : (s2 ? t.A[2] : (s1 ? t.A[1] : t.A[0]));
    r.C = s3 ? t.C[3] : (s2 ? t.C[2] : (s1 ? t.C[1] : t.C[0]));
    r.m = s3 ? t.m[3] : (s2 ? t.m[2] : (s1 ? t.m[1] : t.m[0]));
    r.ldc = s3 ? t.ldc[3] : (s2 ? t.ldc[2] : (s1 ? t.ldc[1] : t.ldc[0]));
    r.idx = s3 ? t.idx[3] : (s2 ? t.idx[2] : (s1 ? t.idx[1] : t.idx[0]));
    r.ht = ht - (s3 ? t.e[2] : (s2 ? t.e[1] : (s1 ? t.e[0] : 0)));
    return r;
}

struct kq_cursor {
    int ht, chunk;
};

// loads are issued UNCONDITIONALLY through bounds-checked descriptors (zero records = nothing to fetch), see above
template <typename TR, int NW, int GEMV_CH, bool IDS>
__device__ __forceinline__ void kq_issue(typename TR::chunk &ch, const gemv_mats &mats, const kq_tab tab, const kq_cursor c, const int n_ht,
                                         const uint32_t rt_bytes, const int wave, const int i16, const int h, const int gsel) {
    const kq_sel p = kq_pick(tab, c.ht);
    const uint8_t *A = p.A;
    bool have = c.ht < n_ht;
    if constexpr (IDS) { // expert picked on the device: no routing-table read-back, graph-capturable
        const int ex = mats.ids[p.idx];
        const bool ok = ex >= 0 && ex < mats.experts;
        A += (size_t)(ok ? ex : 0) * mats.expert_bytes;
        have = have && ok;
    }
    const int hh = p.ht & 1;
    const lfamd_rsrc r = make_rsrc(A + (size_t)(p.ht >> 1) * rt_bytes, have ? rt_bytes : 0u);
    const int slot = h * 32 + hh * 16 + i16, hrow = hh * 16 + i16;
#pragma unroll
    for (int s = 0; s < GEMV_CH; s++) {
        const int b = wave + NW * (c.chunk * GEMV_CH + s); // b >= nb lands past the descriptor: zeros
        TR::load(ch, s, r, (uint32_t)b * TR::TILE, gsel, slot, hrow);
    }
}

// v[l] + v[l ^ 16] + v[l ^ 32] + v[l ^ 48] in every lane, with gfx950's row / half-wave swaps (two VALU instead of two
// ds_bpermute round trips through the LDS at the end of every tile)
__device__ static inline float kq_sum_rows(float v) {
    // (inline asm: with __builtin_amdgcn_permlane16_swap hipcc 7.2 loses the instruction's SECOND result in this kernel
    // and adds the first to itself — tests/test_golden.py caught it; s_nop 1 = the VALU-write -> permlane-read hazard)
    float x = v, y = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); // rows (r0,r0,r2,r2) and (r1,r1,r3,r3)
    float p = x + y, q;
    q = p;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q)); // halves (lo,lo) and (hi,hi)
    return p + q;
}

template <typename TR, int BT, int NW, int GEMV_CH, bool IDS, bool EARLY = false, bool PAIR = false>
__device__ __forceinline__ void gemv_kq_body1(const gemv_mats &mats, int nb, const uint8_t *__restrict__ B,
                                              size_t b_row_bytes, long col0, int n_ht, const int bid, int gdim,
                                              uint8_t *lds, const uint8_t *A0_pre, int cnt_pre) {
    asm volatile("" : "+s"(nb), "+s"(B), "+s"(b_row_bytes), "+s"(col0), "+s"(n_ht), "+s"(gdim)); // (one s_load round)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i16 = lane & 15, h = (lane >> 4) & 1, gsel = lane >> 5;
    constexpr int RW = PAIR ? 32 : 16; // result rows per item: a half-tile, or (PAIR) both half-tiles of a 32-row tile
    float *red = (float *)(lds + (size_t)nb * XBLK); // [2][NW][RW]
#if GEMV_DIAG
    int stamp_n = 0;
    unsigned long long gwg_clk0 = 0;
#endif
    GWG(0);
    GSTAMP();
    // the activation loads need only the leading (preloaded) kernel arguments: they go out before the rest of the
    // arguments has even arrived (-amdgpu-kernarg-preload-count, Makefile)
    // GEMV_CH == 1 (launched for nb <= NW only): ONE block per wave, a float4 per lane.  Deeper rows: FOUR blocks per
    // group as two passes of two (lane l: eight values of block (l >> 5) of the pass), four loads per lane.
    constexpr int JX = GEMV_CH == 1 ? 1 : (NW == 8 ? 2 : 4);
    const uint8_t *xrow = B + col0 * b_row_bytes;
    const lfamd_rsrc rx = make_rsrc(xrow, (uint32_t)nb * 1024u);
    uint4 xv[JX];
    auto x_off = [&](int j0, int j) __attribute__((always_inline)) { // byte offset of load j of the group starting at block slot j0
        if constexpr (JX == 1)
            return (uint32_t)(wave + NW * j0) * 1024u + (uint32_t)lane * 16u;
        else
            return (uint32_t)(wave + NW * (j0 + 2 * (j >> 1) + (lane >> 5))) * 1024u + (uint32_t)(lane & 31) * 32u + 16u * (j & 1);
    };
    if constexpr (BT == LFAMD_TYPE_F32) {
#pragma unroll
        for (int j = 0; j < JX; j++)
            xv[j] = buf_ld16(rx, x_off(0, j));
        __builtin_amdgcn_sched_barrier(0); // (the scheduler would put the weight loads first)
    }
    // pre-quantised Q8_K rows, one block per wave: the block's codes (8 bytes per lane of the lower half-wave) and its scale
    // are fetched here too, ahead of the weights
    // (two blocks per wave in the 8-wave form: the upper half-wave fetches block wave + NW)
    constexpr bool PRE1 = BT == LFAMD_TYPE_Q8_K && TR::ACT == LFAMD_TYPE_Q8_K && JX <= 2;
    uint2 pq = make_uint2(0, 0);
    uint32_t pd = 0;
    const int pre_b = JX == 1 ? wave : wave + NW * (lane >> 5);
    if constexpr (PRE1) {
        const lfamd_rsrc rq = make_rsrc(xrow, (uint32_t)nb * 292u);
        pq = buf_ld8(rq, (uint32_t)pre_b * 292u + 36u + (uint32_t)(lane & 31) * 8u);
        pd = __builtin_amdgcn_raw_buffer_load_b32(rq, (uint32_t)pre_b * 292u, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // a single-matrix launch (attn_output, ffn_down) knows its weights from the preloaded arguments: its first item goes
    // out BEFORE the argument table's scalar loads have returned (~0.4 us earlier); both branches issue the same loads
    typename TR::chunk bufA, bufB;
    typename TR::chunk bufA2, bufB2; // (PAIR: the second half-tile of the item; dead otherwise)
    static_assert(!(PAIR && (EARLY || IDS)), "the paired item is a variant of the plain launch");
    constexpr bool early = EARLY && !IDS; // (a kernel variant, not a run-time branch: hipcc merges its wait counts at a join)
    (void)cnt_pre;
    if constexpr (early) {
        const int hh = bid & 1;
        const uint32_t rtb = (uint32_t)nb * TR::TILE;
        const lfamd_rsrc r = make_rsrc(A0_pre + (size_t)(bid >> 1) * rtb, bid < n_ht ? rtb : 0u);
#pragma unroll
        for (int s = 0; s < GEMV_CH; s++)
            TR::load(bufA, s, r, (uint32_t)(wave + NW * s) * TR::TILE, gsel, h * 32 + hh * 16 + i16, hh * 16 + i16);
    }
    const kq_tab tab = kq_table(mats);

    const int sb_per_wave = (nb + NW - 1) / NW;
    const int cpt = (sb_per_wave + GEMV_CH - 1) / GEMV_CH; // chunks per tile
    const uint32_t rt_bytes = (uint32_t)nb * TR::TILE;

    float acc = 0.0f, acc2 = 0.0f;
    int par = 0;
    constexpr int HT_STEP = PAIR ? 2 : 1; // (PAIR: a cursor points at the even half-tile of its tile)
    kq_cursor ci{HT_STEP * bid, 0}, cc{HT_STEP * bid, 0};
#define KQ_ADVANCE(c)                                                                                                  \
    do {                                                                                                               \
        if (++(c).chunk == cpt)                                                                                        \
            (c).chunk = 0, (c).ht += HT_STEP * gdim;                                                                   \
    } while (0)
    // (measured and rejected: s_setprio 3 around the load issue, and a work-group barrier between every wave's activation
    // loads and the first weight loads — the youngest waves of a k = 14336 row still issue 1.5-3 us after the oldest, the
    // CU's vector-memory queue is simply full — both 2-7 % slower on every decode shape)
#define KQ_ISSUE(buf)                                                                                                  \
    do {                                                                                                               \
        kq_issue<TR, NW, GEMV_CH, IDS>(buf, mats, tab, ci, n_ht, rt_bytes, wave, i16, h, gsel);                            \
        if constexpr (PAIR) {                                                                                          \
            const kq_cursor c2{ci.ht + 1, ci.chunk};                                                                   \
            kq_issue<TR, NW, GEMV_CH, IDS>(buf##2, mats, tab, c2, n_ht, rt_bytes, wave, i16, h, gsel);                 \
        }                                                                                                              \
        KQ_ADVANCE(ci);                                                                                                \
    } while (0)
    // one chunk of this wave against its own image blocks; at the end of a tile: 4 lanes per row (h, gsel), then the
    // waves through LDS in a fixed order
#define KQ_CONSUME(buf)                                                                                                \
    do {                                                                                                               \
        GSTAMP_ARRIVAL();                                                                                              \
        _Pragma("unroll") for (int s = 0; s < GEMV_CH; s++) {                                                          \
            const int b = wave + NW * (cc.chunk * GEMV_CH + s);                                                        \
            /* a super-block beyond the row was loaded as zeros; its image slot belongs to nobody: discard */          \
            const float t = TR::dot(buf, s, lds + (size_t)(b < nb ? b : 0) * XBLK, gsel, h);                           \
            acc += b < nb ? t : 0.0f;                                                                                  \
            if constexpr (PAIR) {                                                                                      \
                const float t2 = TR::dot(buf##2, s, lds + (size_t)(b < nb ? b : 0) * XBLK, gsel, h);                   \
                acc2 += b < nb ? t2 : 0.0f;                                                                            \
            }                                                                                                          \
        }                                                                                                              \
        if (cc.chunk == cpt - 1) {                                                                                     \
            float *rb = red + par * (NW * RW);                                                                         \
            par ^= 1;                                                                                                  \
            float v = kq_sum_rows(acc); /* lanes i16, i16 + 16, i16 + 32, i16 + 48 */                                  \
            if (lane < 16)                                                                                             \
                rb[wave * RW + lane] = v;                                                                              \
            acc = 0.0f;                                                                                                \
            if constexpr (PAIR) {                                                                                      \
                float v2 = kq_sum_rows(acc2);                                                                          \
                if (lane < 16)                                                                                         \
                    rb[wave * RW + 16 + lane] = v2;                                                                    \
                acc2 = 0.0f;                                                                                           \
            }                                                                                                          \
            GSTAMP();                                                                                                  \
            __syncthreads();                                                                                           \
            GSTAMP();                                                                                                  \
            if (threadIdx.x < RW) {                                                                                    \
                const int i = threadIdx.x;                                                                             \
                float t = 0.0f;                                                                                        \
                _Pragma("unroll") for (int w = 0; w < NW; w++) t += rb[w * RW + i];                                    \
                const kq_sel p = kq_pick(tab, cc.ht);                                                               \
                const long row = (long)(p.ht >> 1) * 32 + (p.ht & 1) * 16 + i; /* (PAIR: p.ht even, i < 32) */         \
                bool ok = true;                                                                                        \
                if constexpr (IDS) { /* an out-of-range expert id leaves its result row untouched */                   \
                    const int ex = mats.ids[p.idx];                                                                    \
                    ok = ex >= 0 && ex < mats.experts;                                                                 \
                }                                                                                                      \
                if (row < p.m && ok)                                                                                   \
                    ((__attribute__((address_space(1))) float *)p.C)[col0 * p.ldc + row] = t; /* (not FLAT) */                                                                     \
            }                                                                                                          \
        }                                                                                                              \
        KQ_ADVANCE(cc);                                                                                                \
    } while (0)

    if constexpr (BT == LFAMD_TYPE_F32) {
        // this wave's blocks, JX at a time; the first JX went out BEFORE the first weight chunk (in-order vmcnt).  The
        // staging is straight-line code (a block past the row is loaded as zeros through the descriptor and staged
        // into the wave's dummy slot): with a branch per block hipcc merges its vmcnt bookkeeping at the joins and
        // makes the second block wait for the WEIGHTS.
        uint8_t *dummy = (uint8_t *)(red + 2 * NW * RW) + (size_t)wave * XBLK;
        if constexpr (early)
            KQ_ADVANCE(ci);
        else
            KQ_ISSUE(bufA);
        __builtin_amdgcn_sched_barrier(0);
        GSTAMP();
#define KQ_F4(u) make_float4(__builtin_bit_cast(float, (u).x), __builtin_bit_cast(float, (u).y), __builtin_bit_cast(float, (u).z), \
                           __builtin_bit_cast(float, (u).w))
#define KQ_STAGE_GROUP(j0)                                                                                             \
    if constexpr (JX == 1) {                                                                                           \
        const int b = wave + NW * (j0);                                                                                \
        uint8_t *dst = b < nb ? lds + (size_t)b * XBLK : dummy;                                                        \
        if constexpr (TR::ACT == LFAMD_TYPE_Q8_K)                                                                      \
            stage_f32_q8k_wave(dst, KQ_F4(xv[0]), lane);                                                               \
        else                                                                                                           \
            stage_f32_q80_wave<TR::ACT == LFAMD_TYPE_Q8_1>(dst, KQ_F4(xv[0]), lane);                                   \
    } else {                                                                                                           \
        _Pragma("unroll") for (int ps = 0; ps < JX / 2; ps++) {                                                        \
            const int b = wave + NW * ((j0) + 2 * ps + (lane >> 5));                                                   \
            uint8_t *dst = b < nb ? lds + (size_t)b * XBLK : dummy;                                                    \
            if constexpr (TR::ACT == LFAMD_TYPE_Q8_K)                                                                  \
                stage_f32_q8k_wave2(dst, KQ_F4(xv[2 * ps]), KQ_F4(xv[2 * ps + 1]), lane);                              \
            else                                                                                                       \
                stage_f32_q80_wave2<TR::ACT == LFAMD_TYPE_Q8_1>(dst, KQ_F4(xv[2 * ps]), KQ_F4(xv[2 * ps + 1]), lane);  \
        }                                                                                                              \
    }
        KQ_STAGE_GROUP(0) // (peeled: inside the loop below the back edge would hide the weight loads from the wait counts)
        for (int j0 = JX; j0 < sb_per_wave; j0 += JX) { // rows longer than JX * NW super-blocks
#pragma unroll
            for (int j = 0; j < JX; j++)
                xv[j] = buf_ld16(rx, x_off(j0, j));
            KQ_STAGE_GROUP(j0)
        }
#undef KQ_F4
#undef KQ_STAGE_GROUP
    } else {
        if constexpr (early)
            KQ_ADVANCE(ci);
        else
            KQ_ISSUE(bufA);
        GSTAMP();
        if constexpr (PRE1 && JX == 1) {
            if (wave < nb && lane < 32) {
                uint8_t *dst = lds + (size_t)wave * XBLK;
                const int hs = put_group(dst, lane, pq.x, pq.y);
                const int other = (int)dpp_u32<DPP_XOR2>((uint32_t)hs);
                if ((lane & 2) == 0)
                    put_pair(dst, lane, hs + other);
                if (lane == 0)
                    *(uint32_t *)(dst + XBLK_D) = pd;
            }
        } else if constexpr (PRE1) {
            { // both half-waves, straight-line (a block past the row was fetched as zeros and lands in the dummy slot)
                uint8_t *dummy = (uint8_t *)(red + 2 * NW * RW) + (size_t)wave * XBLK;
                uint8_t *dst = pre_b < nb ? lds + (size_t)pre_b * XBLK : dummy;
                const int grp = lane & 31;
                const int hs = put_group(dst, grp, pq.x, pq.y);
                const int other = (int)dpp_u32<DPP_XOR2>((uint32_t)hs);
                if ((grp & 2) == 0)
                    put_pair(dst, grp, hs + other);
                if (grp == 0)
                    *(uint32_t *)(dst + XBLK_D) = pd;
            }
            for (int b = wave + 2 * NW; b < nb; b += NW)
                stage_quantised_wave<TR::ACT>(lds + (size_t)b * XBLK, xrow, b, lane);
        } else {
            for (int b = wave; b < nb; b += NW)
                stage_quantised_wave<TR::ACT>(lds + (size_t)b * XBLK, xrow, b, lane);
        }
    }
    // the image blocks this wave reads are the ones it has just written: program order in the LDS queue is enough;
    // only the compiler must not move the reads above the (differently typed) writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    GSTAMP();

    // (measured and rejected, round 2, twice: TWO items ahead for deep rows — the second buffer issued before the staging, or
    // right after this wave's activations have landed and before the quantisation arithmetic; pre-quantised rows with both
    // up front: 9.5 -> 10.5 us Q4_K, 14.9 -> 16.4 Q6_K, 11.0 / 17.6 pre-quantised.  A CU's loads return in issue order and
    // the 256 CUs together already take what HBM gives (33 MB in ~5.2 us of a ~6.8 us kernel body): putting wave w's second
    // item ahead of wave w+1's first only delays the moment the LAST wave can start its first dot, and both its dots
    // then run after the stream has ended)
    // pairs of items without a branch inside (a conditionally skipped consume leaves its buffer's loads "pending" at
    // the loop header for hipcc's wait-count pass, which then drains vmcnt(0) before every issue), then the odd one
    // (measured and rejected, round 3: FOUR items in flight per wave for launches whose work-groups walk many small items — the
    // Mixtral gate + up experts, 14 half-tiles of 37 KB per work-group: decode pass 1.820 -> 1.883 ms)
    for (;;) {
        kq_cursor cn = cc;
        KQ_ADVANCE(cn);
        if (cn.ht >= n_ht)
            break;
        KQ_ISSUE(bufB);
        KQ_CONSUME(bufA);
        GSTAMP();
        KQ_ISSUE(bufA);
        KQ_CONSUME(bufB);
    }
    if (cc.ht < n_ht)
        KQ_CONSUME(bufA);
    GWG(1);
#undef KQ_ADVANCE
#undef KQ_ISSUE
#undef KQ_CONSUME
}

template <typename TR, int NC, int BT, int NW, int GEMV_CH, bool IDS = false, bool EARLY = false, bool PAIR = false>
__global__ __launch_bounds__(NW * 64) void gemv_kq_kernel(const uint8_t *__restrict__ B, size_t b_row_bytes, long col0, int nb,
                                                          int n_ht, int gdim, const uint8_t *__restrict__ A0, int cnt,
                                                          const gemv_mats mats) {
    // (A0 = mats.A[0], cnt = mats.count once more, among the PRELOADED leading arguments: a single-matrix launch issues its
    // first weight loads without waiting for the argument table)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if constexpr (NC == 1)
        gemv_kq_body1<TR, BT, NW, GEMV_CH, IDS, EARLY, PAIR>(mats, nb, B, b_row_bytes, col0, n_ht, (int)blockIdx.x, gdim, lds, A0, cnt);
    else
        gemv_kq_body<TR, NC, BT, NW, GEMV_CH, IDS>(mats, nb, B, b_row_bytes, col0, n_ht, (int)blockIdx.x, gdim, lds);
}

// Two expert GEMVs on DIFFERENT activation rows in ONE decode launch (GGML_OP_MUL_MAT_ID ffn_down_exps: every chosen expert
// multiplies its own row): work-groups [0, grid_a) run mats_a against row Ba, the rest mats_b against row Bb.  A work-group
// serves one of the two, so it stages one row, as in a single launch.
template <typename TR, int BT, int NW, int GEMV_CH>
__global__ __launch_bounds__(NW * 64) void gemv_kq_ids_pair_kernel(const uint8_t *__restrict__ Ba, const uint8_t *__restrict__ Bb,
                                                                   size_t b_row_bytes, int nb, int n_ht_a, int n_ht_b, int grid_a, int grid_b,
                                                                   const gemv_mats mats_a, const gemv_mats mats_b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if ((int)blockIdx.x < grid_a)
        gemv_kq_body1<TR, BT, NW, GEMV_CH, true>(mats_a, nb, Ba, b_row_bytes, 0, n_ht_a, (int)blockIdx.x, grid_a, lds, nullptr, 0);
    else
        gemv_kq_body1<TR, BT, NW, GEMV_CH, true>(mats_b, nb, Bb, b_row_bytes, 0, n_ht_b, (int)blockIdx.x - grid_a, grid_b, lds, nullptr, 0);
}

// Two weight types in ONE decode launch (sibling mat-muls on the same activations whose types differ: attn_q/k in Q4_K
// with attn_v in Q6_K in a Q4_K_M file): work-groups [0, grid_a) run type A's body over mats_a, the rest type B's over
// mats_b.  A work-group is of one type, so it stages the activations once, in the (shared) Q8_K image.
template <typename TRA, typename TRB, int BT, int NW, int GEMV_CH>
__global__ __launch_bounds__(NW * 64) void gemv_kq_dual_kernel(const uint8_t *__restrict__ B, size_t b_row_bytes, int nb, int n_ht_a,
                                                               int n_ht_b, int grid_a, int grid_b, const gemv_mats mats_a,
                                                               const gemv_mats mats_b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    static_assert(TRA::ACT == TRB::ACT, "both types must share the activation image");
    if ((int)blockIdx.x < grid_a)
        gemv_kq_body1<TRA, BT, NW, GEMV_CH, false>(mats_a, nb, B, b_row_bytes, 0, n_ht_a, (int)blockIdx.x, grid_a, lds, nullptr, 0);
    else
        gemv_kq_body1<TRB, BT, NW, GEMV_CH, false>(mats_b, nb, B, b_row_bytes, 0, n_ht_b, (int)blockIdx.x - grid_a, grid_b, lds, nullptr,
                                                   0);
}

// ---------------------------------------------------------------------------------------------
// Q8_0 x Q8_0, bit-exact restatement of tinyBLAS_Q0_AVX2::gemm (tinyblas_cpu.h:934-971).
// One wave = 8 weight rows, lane = (r = lane>>3, j = lane&7) owns f32 lane j of row r's accumulator
// Cv; blocks are visited in order l = 0..nblocks-1 exactly like the reference's loop, so every
// rounding is the same:  a = f32(dA)*f32(dB);  b = f32(int dot of bytes 4j..4j+3);
// Cv = fma(a, b, Cv)   or, on a PRECISE tile, madder (tinyblas_cpu.h:203-209) with the compiler's
// contraction of sub(mul(a,b),e) into fma(a,b,-e).

// LDS image of the Q8_0 activations, per QUAD of four 32-blocks (the unit a weight tile covers): for each of the
// eight dword positions j the four blocks' dwords side by side (a lane = (row r, position j) takes its four
// activation dwords with ONE ds_read_b128; the eight rows of a wave read the same 128 bytes: broadcast), then the
// four block scales as f32 (one more ds_read_b128, uniform).  8 ds_read_b32 per quad became 2 ds_read_b128.
#define X80_QUAD 144
#define X80_QD 128
// quads (1 KiB per wave each) kept in flight per wave: the row's blocks MUST be visited in order by one lane
// (bit-exact f32 chain), so a matrix offers only m/8 waves (2 per CU at m = 4096) and memory-level parallelism
// has to come from depth.  n = 1: 32 (a whole k = 4096 row group in flight, 192 ring VGPRs); batches: 16.
#define Q80_WAVES 2  // waves per work-group (8 rows each) sharing one staged activation image

// MODE: 0 = every output plain fma, 1 = every output Kahan (uniform for n = 1: tinyblas_cpu.h:797-925),
// 2 = per-output choice from the mnpack geometry (small batches n > 1)
// sibling matrices that share the activations (attn_q/k/v, ffn_gate/up) run as ONE launch over their concatenated
// 8-row groups: a 1024-row matrix alone is 128 waves, each a serial k-long chain — three such launches cost three times
// the chain latency, one launch costs it once.  The mnpack geometry (Kahan choice) stays per matrix.
struct q80_mats {
    const uint8_t *A[GEMV_MAX_MATS];
    float *C[GEMV_MAX_MATS];
    long m[GEMV_MAX_MATS];
    long ldc[GEMV_MAX_MATS];
    long rg_end[GEMV_MAX_MATS]; // exclusive prefix of row-group counts
    int count;
};

template <int NC, int BT, int MODE, int Q80_DEPTH>
__global__ __launch_bounds__(Q80_WAVES * 64) void gemv_q80_kernel(const uint8_t *__restrict__ B, size_t b_row_bytes, long col0, int nblocks,
                                                                 int nquads, long n_total, int vregs32, int precise,
                                                                 const q80_mats mats) {
    // (activation pointer and sizes lead the argument list: they arrive preloaded in SGPRs — Makefile,
    // -amdgpu-kernarg-preload-count — and the activation loads below need nothing else)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane >> 3, j = lane & 7;
#if GEMV_DIAG
    int stamp_n = 0;
#endif
    GSTAMP();
    // the first two activation pieces of this thread go out before anything else: they need only the preloaded leading
    // arguments, while the matrix pick below waits for three dependent rounds of scalar loads
    float va0[16], vb0[16];
    if constexpr (BT == LFAMD_TYPE_F32) {
        const float *x0 = (const float *)(B + col0 * b_row_bytes);
        const int pieces0 = nblocks * 2;
        if ((int)threadIdx.x < pieces0)
            load_piece(va0, x0, threadIdx.x);
        if ((int)threadIdx.x + Q80_WAVES * 64 < pieces0)
            load_piece(vb0, x0, threadIdx.x + Q80_WAVES * 64);
        __builtin_amdgcn_sched_barrier(0);
    }
    long rg = (long)blockIdx.x * Q80_WAVES + wave;
    int mj = 0;
#pragma unroll
    for (int jj = 1; jj < GEMV_MAX_MATS; jj++)
        if (jj < mats.count && rg >= mats.rg_end[jj - 1])
            mj = jj;
    if (mj > 0)
        rg -= mats.rg_end[mj - 1];
    const uint8_t *__restrict__ A = mats.A[mj];
    float *__restrict__ C = mats.C[mj];
    const long m = mats.m[mj], ldc = mats.ldc[mj];
    const long n_rg = (m + 7) / 8;
    const long row = rg * 8 + r;
    // bounds-checked, unconditional weight loads (zeros past the row group / for an idle wave): keeps
    // hipcc's counted vmcnt exact so Q80_DEPTH KiB per wave really stay in flight
    const uint32_t rg_bytes = (uint32_t)nquads * P80_TILE;
    const lfamd_rsrc rA = make_rsrc(A + (size_t)(rg < n_rg ? rg : 0) * rg_bytes, rg < n_rg ? rg_bytes : 0u);

    uint4 qa[Q80_DEPTH];
    uint2 ds[Q80_DEPTH];
    auto issue = [&](int s, int L) {
        qa[s] = buf_ld16_nt(rA, (uint32_t)L * P80_TILE + lane * 16);
        ds[s] = buf_ld8(rA, (uint32_t)L * P80_TILE + P80_D + r * 8);
    };

    if constexpr (BT == LFAMD_TYPE_F32) {
        // quantize_row_q8_0 (upstream): d = amax/127, id = 1/d, q = roundf(x*id); 16 floats per lane,
        // two lanes per 32-block.  The first piece of each thread is fetched BEFORE the weights (vmcnt
        // retires in order), the weights are issued, then the activations are quantised under their flight.
        const int pieces = nblocks * 2, nthr = Q80_WAVES * 64;
        for (int c = 0; c < NC; c++) {
            const float *x = (const float *)(B + (col0 + c) * b_row_bytes);
            // two pieces per thread and round: both loads go out together (one memory latency per round, not two)
            for (int p0 = 0; p0 < pieces; p0 += 2 * nthr) {
                const int pa = p0 + threadIdx.x, pb = pa + nthr;
                float va[16], vb[16];
                if (c == 0 && p0 == 0) { // (fetched at the top of the kernel)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        va[e] = va0[e], vb[e] = vb0[e];
                } else {
                    if (pa < pieces)
                        load_piece(va, x, pa);
                    if (pb < pieces)
                        load_piece(vb, x, pb);
                }
                if (c == 0 && p0 == 0) {
#pragma unroll
                    for (int s = 0; s < Q80_DEPTH; s++)
                        issue(s, s);
                }
                auto quantise = [&](const float (&v)[16], int p) {
                    float amax = 0.0f;
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        amax = fmaxf(amax, fabsf(v[e]));
                    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                    const float d = amax / 127.0f;
                    const float id = d != 0.0f ? 1.0f / d : 0.0f;
                    uint32_t y[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        int q = (int)roundf(v[e] * id);
                        y[e >> 2] |= (uint32_t)(q & 0xff) << (8 * (e & 3));
                    }
                    const int l = p >> 1, hf = p & 1;
                    uint8_t *dst = lds + (size_t)(c * nquads + (l >> 2)) * X80_QUAD + (l & 3) * 4;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        *(uint32_t *)(dst + (4 * hf + e) * 16) = y[e];
                    if (hf == 0)
                        *(float *)(dst + X80_QD) = h2f(f2h_bits(d)); // the block stores d as f16
                };
                // (pieces is even and nthr a multiple of 64: the lane pair (2i, 2i+1) of a block is either both in or out)
                if (pa < pieces)
                    quantise(va, pa);
                if (pb < pieces)
                    quantise(vb, pb);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < Q80_DEPTH; s++)
            issue(s, s);
        for (int idx = threadIdx.x; idx < NC * nblocks * 9; idx += Q80_WAVES * 64) {
            int c = idx / (nblocks * 9), rem = idx % (nblocks * 9);
            int l = rem / 9, w = rem % 9;
            const uint8_t *y = B + (col0 + c) * b_row_bytes + (size_t)l * 34;
            uint32_t v;
            if (w < 8) {
                const uint16_t *p = (const uint16_t *)(y + 2 + 4 * w); // 34-byte blocks: 2-byte aligned
                v = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
            } else {
                v = __builtin_bit_cast(uint32_t, h2f(*(const uint16_t *)y));
            }
            *(uint32_t *)(lds + (size_t)(c * nquads + (l >> 2)) * X80_QUAD + (l & 3) * 4 + (w < 8 ? w * 16 : X80_QD)) = v;
        }
    }
    GSTAMP();
    __syncthreads();
    GSTAMP();

    bool kahan[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
        kahan[c] = MODE == 2 ? q0_is_kahan(row < m ? row : m - 1, col0 + c, m, n_total, vregs32 != 0, precise != 0) : MODE == 1;

    float Cv[NC], Ce[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
        Cv[c] = Ce[c] = 0.0f;

    // One block: a = f32(dA)*f32(dB), b = f32(int dot of bytes 4j..4j+3), then the reference's update.  The update is a
    // chain of dependent f32 ops (four per block under Kahan) that ONE lane must run in block order; everything else
    // (scale products, integer dots) is independent of it.  A row offers a single wave no other work, so the loop is
    // software-pipelined by hand: the products of quad L+1 are prepared (prep) before the chain of quad L runs and
    // fill its latency bubbles (measured: the fused form spent ~80 cycles per block, 4-5 us per k = 4096 row).
    struct prepd {
        float a[4][NC], b[4][NC];
    };
    auto prep = [&](int sl, int L, prepd &P) {
        const uint4 q4 = qa[sl];
        const uint2 d2 = ds[sl];
        const uint32_t qw[4] = {q4.x, q4.y, q4.z, q4.w};
        const float da[4] = {h2f((uint16_t)(d2.x & 0xffff)), h2f((uint16_t)(d2.x >> 16)), h2f((uint16_t)(d2.y & 0xffff)),
                             h2f((uint16_t)(d2.y >> 16))};
        const int Lc = L < nquads ? L : nquads - 1; // clamped: the LDS reads are unconditional
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint8_t *xb = lds + (size_t)(c * nquads + Lc) * X80_QUAD;
            const uint4 xq4 = *(const uint4 *)(xb + j * 16);
            const float4 xd4 = *(const float4 *)(xb + X80_QD);
            const uint32_t xq[4] = {xq4.x, xq4.y, xq4.z, xq4.w};
            const float xd[4] = {xd4.x, xd4.y, xd4.z, xd4.w};
#pragma unroll
            for (int dd = 0; dd < 4; dd++) {
                P.a[dd][c] = da[dd] * xd[dd];
                P.b[dd][c] = (float)sdot4(qw[dd], xq[dd], 0);
            }
        }
    };
    auto chain = [&](const prepd &P, int dd) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float a = P.a[dd][c], bq = P.b[dd][c];
            if constexpr (MODE == 0) {
                Cv[c] = __builtin_fmaf(a, bq, Cv[c]);
            } else if constexpr (MODE == 1) {
                const float y = __builtin_fmaf(a, bq, -Ce[c]);
                const float t = Cv[c] + y;
                Ce[c] = (t - Cv[c]) - y;
                Cv[c] = t;
            } else { // branch-free select
                const float plain = __builtin_fmaf(a, bq, Cv[c]);
                const float y = __builtin_fmaf(a, bq, -Ce[c]);
                const float t = Cv[c] + y;
                const float e2 = (t - Cv[c]) - y;
                Cv[c] = kahan[c] ? t : plain;
                Ce[c] = kahan[c] ? e2 : 0.0f;
            }
        }
    };

    // Blocks past the row (zero padding of the last quad, zero-filled prefetch slots) must NOT run: a Kahan
    // step with a*b = 0 still folds the pending compensation into the sum.  Full quads run unguarded.
    const int nq_full = nblocks >> 2;
    prepd P[2];
    prep(0, 0, P[0]);
    issue(0, Q80_DEPTH);
    // rounds of Q80_DEPTH full quads run without a branch in the body (k = 4096 and 14336: every round);
    // the remainder round carries the guards
    int L0 = 0;
    for (; L0 + Q80_DEPTH <= nq_full; L0 += Q80_DEPTH) {
#pragma unroll
        for (int s = 0; s < Q80_DEPTH; s++) {
            const int sn = (s + 1) % Q80_DEPTH;
            prep(sn, L0 + s + 1, P[(s + 1) & 1]); // slot sn holds quad L+1
            issue(sn, L0 + s + 1 + Q80_DEPTH);    // and is refilled as soon as its registers are read
            const prepd &Pc = P[s & 1];
            chain(Pc, 0);
            chain(Pc, 1);
            chain(Pc, 2);
            chain(Pc, 3);
        }
    }
    if (L0 < nquads) {
#pragma unroll
        for (int s = 0; s < Q80_DEPTH; s++) {
            const int L = L0 + s;
            const int sn = (s + 1) % Q80_DEPTH;
            prep(sn, L + 1, P[(s + 1) & 1]);
            const prepd &Pc = P[s & 1];
            if (L < nq_full) {
                chain(Pc, 0);
                chain(Pc, 1);
                chain(Pc, 2);
                chain(Pc, 3);
            } else if (L == nq_full) {
                if (4 * L + 0 < nblocks)
                    chain(Pc, 0);
                if (4 * L + 1 < nblocks)
                    chain(Pc, 1);
                if (4 * L + 2 < nblocks)
                    chain(Pc, 2);
            }
        }
    }
    GSTAMP();
    // hsum(__m256), tinyblas_cpu.h:277-296: ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7))
#pragma unroll
    for (int c = 0; c < NC; c++) {
        float v = Cv[c];
        v = v + __shfl_xor(v, 4, 64);
        v = v + __shfl_xor(v, 2, 64);
        v = v + __shfl_xor(v, 1, 64);
        if (j == 0 && row < m)
            C[(col0 + c) * ldc + row] = v;
    }
}

// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------

static int g_num_cus = 0;

static int num_cus() {
    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            g_num_cus = p.multiProcessorCount;
        if (g_num_cus <= 0)
            g_num_cus = 256;
    }
    return g_num_cus;
}

// which types take 32-row items on long walks (launch_kq): the ones whose dot is long enough for a second, independent one to
// fill its gaps.  128256 x 4096, 16-row -> 32-row items: Q6_K 82.0 -> 73.7 us, Q2_K 44.6 -> 40.8, Q3_K 50.3 -> 44.9, IQ4_XS
// 58.1 -> 55.4; the light dots lose a little: Q4_K 47.4 -> 47.7, Q5_K 58.1 -> 58.5, Q4_0 46.0 -> 47.6 (65536 rows: 27.0 -> 28.2)
template <typename TR>
struct kq_pair_items {
    static constexpr bool value = false;
};
template <>
struct kq_pair_items<q6k_traits> {
    static constexpr bool value = true;
};
template <int TYPE>
struct kq_pair_items<pk_traits<TYPE>> {
    static constexpr bool value = true;
};
template <>
struct kq_pair_items<iq4c_traits> {
    static constexpr bool value = true;
};

template <typename TR, int NC, int BT, int NW, int CH>
static hipError_t launch_kq(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    int nb = (int)(k / 256);
    size_t smem = (size_t)NC * nb * XBLK + 2 * NW * NC * 16 * sizeof(float) + (size_t)NW * XBLK; // (+ dummy slots)
    // persistent grid: 16 waves per CU, every work-group the same number of half-tiles
    const int max_wg = (16 / NW) * num_cus();
    const int per_wg = (n_ht + max_wg - 1) / max_wg;
    const int grid = (n_ht + per_wg - 1) / per_wg;
    auto go = [&](auto kernel) {
        if (smem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess)
                return e;
        }
        kernel<<<grid, NW * 64, smem, s>>>((const uint8_t *)B, brb, col0, nb, n_ht, grid, mats.A[0], mats.count, mats);
        return hipGetLastError();
    };
    if constexpr (NC == 1 && NW == 16 && CH == 1 && kq_pair_items<TR>::value) {
        // Long walks (output.weight: 63 half-tiles per work-group) take items of a full 32-row tile — both half-tiles in
        // flight together, ONE barrier + reduce + store per 74 KB instead of per 37 KB; same arithmetic per row, same bits.
        // 128256 x 4096 Q6_K: 82.8 -> 73.6 us (5.2 -> 5.9 TB/s); 32000 x 4096: 23.6 -> 22.8 (types: kq_pair_items).  Short walks lose to the
        // coarser division of the tiles over the work-groups (28672 x 4096, 3.5 tiles each: 13.1 -> 14.3 us), hence the bound.
        // LFAMD_GEMV_PAIR_MIN=<half-tiles per work-group> moves it (0 = never).
        static const int pair_min = getenv("LFAMD_GEMV_PAIR_MIN") ? atoi(getenv("LFAMD_GEMV_PAIR_MIN")) : 16;
        if (pair_min > 0 && per_wg >= pair_min) {
            const int n_t = n_ht / 2, per_t = (n_t + max_wg - 1) / max_wg, grid_t = (n_t + per_t - 1) / per_t;
            const size_t smem_t = smem + 2 * NW * 16 * sizeof(float);
            auto kernel = gemv_kq_kernel<TR, NC, BT, NW, CH, false, false, true>;
            if (smem_t > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_t);
                if (e != hipSuccess)
                    return e;
            }
            kernel<<<grid_t, NW * 64, smem_t, s>>>((const uint8_t *)B, brb, col0, nb, n_ht, grid_t, mats.A[0], mats.count, mats);
            return hipGetLastError();
        }
    }
    if constexpr (NC == 1) {
        if (mats.count == 1) // one matrix: the variant that issues its first weight loads from the preloaded arguments
            return go(gemv_kq_kernel<TR, NC, BT, NW, CH, false, true>);
    }
    return go(gemv_kq_kernel<TR, NC, BT, NW, CH>);
}

template <typename TRA, typename TRB, int BT, int NW, int CH>
static hipError_t launch_kq_dual_nw(const gemv_mats &ma, int n_ht_a, const gemv_mats &mb, int n_ht_b, int nb, const void *B,
                                    size_t brb, hipStream_t s) {
    const size_t smem = (size_t)nb * XBLK + 2 * NW * 16 * sizeof(float) + (size_t)NW * XBLK; // (+ dummy slots)
    // one persistent grid of at most one work-group per CU, split between the types so that the slower side finishes
    // first: a Q6_K half-tile costs about 1.35 Q4_K / Q5_K ones (dot instructions and bytes), and with equal tiles per
    // work-group the few Q6_K work-groups of attn_v set the launch's length (7.06 -> see DESIGN §4)
    const int max_wg = num_cus();
    int grid_a = 0, grid_b = 0;
    long best = -1;
    for (int pb = 1; pb <= n_ht_b; pb++) {
        const int gb = (n_ht_b + pb - 1) / pb;
        if (gb >= max_wg)
            continue;
        const int pa = (n_ht_a + (max_wg - gb) - 1) / (max_wg - gb);
        const long cost = (long)pa * 100 > (long)pb * 135 ? (long)pa * 100 : (long)pb * 135;
        if (best < 0 || cost < best)
            best = cost, grid_b = gb, grid_a = (n_ht_a + pa - 1) / pa;
    }
    if (best < 0) { // (more Q6_K half-tiles than CUs can never be one per work-group: equal shares)
        const int per_wg = (n_ht_a + n_ht_b + max_wg - 1) / max_wg;
        grid_a = (n_ht_a + per_wg - 1) / per_wg, grid_b = (n_ht_b + per_wg - 1) / per_wg;
    }
    auto kernel = gemv_kq_dual_kernel<TRA, TRB, BT, NW, CH>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess)
            return e;
    }
    kernel<<<grid_a + grid_b, NW * 64, smem, s>>>((const uint8_t *)B, brb, nb, n_ht_a, n_ht_b, grid_a, grid_b, ma, mb);
    return hipGetLastError();
}

template <typename TRA, typename TRB, int BT>
static hipError_t launch_kq_dual(const gemv_mats &ma, int n_ht_a, const gemv_mats &mb, int n_ht_b, long k, const void *B,
                                 size_t brb, hipStream_t s) {
    const int nb = (int)(k / 256);
    if (nb <= 16)
        return launch_kq_dual_nw<TRA, TRB, BT, 16, 1>(ma, n_ht_a, mb, n_ht_b, nb, B, brb, s);
    return launch_kq_dual_nw<TRA, TRB, BT, 16, 2>(ma, n_ht_a, mb, n_ht_b, nb, B, brb, s);
}

// both launches' half-tiles (n_ht each) on one grid: half of the CUs' work-groups per expert
template <typename TR, int BT>
static hipError_t launch_kq_ids_pair(const gemv_mats &ma, const gemv_mats &mb, int n_ht, long k, const void *Ba, const void *Bb, size_t brb,
                                     hipStream_t s) {
    const int nb = (int)(k / 256);
    constexpr int NW = 16;
    const size_t smem = (size_t)nb * XBLK + 2 * NW * 16 * sizeof(float) + (size_t)NW * XBLK;
    const int max_wg = num_cus() / 2 > 0 ? num_cus() / 2 : 1;
    const int per_wg = (n_ht + max_wg - 1) / max_wg;
    const int g1 = (n_ht + per_wg - 1) / per_wg;
    auto go = [&](auto kernel) {
        if (smem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess)
                return e;
        }
        kernel<<<2 * g1, NW * 64, smem, s>>>((const uint8_t *)Ba, (const uint8_t *)Bb, brb, nb, n_ht, n_ht, g1, g1, ma, mb);
        return hipGetLastError();
    };
    // (round 3, 8 waves x two work-groups per CU for this launch: Mixtral decode pass 1.712 -> 1.919 ms; 16 waves stay)
    if (nb <= 16)
        return go(gemv_kq_ids_pair_kernel<TR, BT, NW, 1>);
    return go(gemv_kq_ids_pair_kernel<TR, BT, NW, 2>);
}

template <typename TR, int BT>
static hipError_t launch_kq_ids(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, hipStream_t s) {
    const int nb = (int)(k / 256);
    constexpr int NW = 16;
    const size_t smem = (size_t)nb * XBLK + 2 * NW * 16 * sizeof(float) + (size_t)NW * XBLK; // (+ dummy slots)
    const int max_wg = num_cus();
    const int per_wg = (n_ht + max_wg - 1) / max_wg;
    const int grid = (n_ht + per_wg - 1) / per_wg;
    // gate + up experts in one launch (4 x 896 half-tiles of 16 super-blocks, 14 per 16-wave work-group): every item ends in a
    // work-group barrier, and two independent 8-wave work-groups per CU hide each other's — Mixtral decode pass 1.823 ->
    // 1.706 ms (548 -> 586 tokens/s).  Shorter walks keep the 16-wave form (see launch_kq_pick).
    static const bool no_nw8 = getenv("LFAMD_IDS_NO_NW8") && atoi(getenv("LFAMD_IDS_NO_NW8"));
    // (the 16-wave form with 32-row items, seven per work-group, was measured too: 1.748 ms per pass against 1.702 for this one)
    if (nb <= 16 && !no_nw8 && n_ht >= 8 * num_cus()) {
        constexpr int NW8 = 8;
        const size_t smem8 = (size_t)nb * XBLK + 2 * NW8 * 16 * sizeof(float) + (size_t)NW8 * XBLK;
        const int max8 = 2 * num_cus(), per8 = (n_ht + max8 - 1) / max8, grid8 = (n_ht + per8 - 1) / per8;
        auto kernel = gemv_kq_kernel<TR, 1, BT, NW8, 2, true>;
        kernel<<<grid8, NW8 * 64, smem8, s>>>((const uint8_t *)B, brb, 0, nb, n_ht, grid8, mats.A[0], 0, mats);
    } else if (nb <= 16) {
        auto kernel = gemv_kq_kernel<TR, 1, BT, NW, 1, true>;
        kernel<<<grid, NW * 64, smem, s>>>((const uint8_t *)B, brb, 0, nb, n_ht, grid, mats.A[0], 0 /* expert picked on the device: no early issue */, mats);
    } else {
        auto kernel = gemv_kq_kernel<TR, 1, BT, NW, 2, true>;
        if (smem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess)
                return e;
        }
        kernel<<<grid, NW * 64, smem, s>>>((const uint8_t *)B, brb, 0, nb, n_ht, grid, mats.A[0], 0 /* expert picked on the device: no early issue */, mats);
    }
    return hipGetLastError();
}

template <typename TR, int NC, int BT>
static hipError_t launch_kq_pick(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0,
                                 hipStream_t s) {
    const long nb = k / 256;
    if constexpr (NC == 1) {
        // a launch of at most one half-tile per CU (attn_output, attn_k/v alone: the whole kernel is one prologue + one
        // item) runs 8 waves of two super-blocks each: half as many waves contend for a SIMD while the row is quantised
        // (two blocks per pass cost 140 VALU against 2 x 120) — 4096 x 4096: 4.35 -> 3.88 us, 1024 x 4096: 3.65 -> 3.27,
        // 4096 x 8192: 6.83 -> 6.49.  With more tiles per work-group the 16-wave form streams better (14336 x 4096:
        // 8.4 vs 9.1 us), and rows of 56 super-blocks lose too (9.4 vs 10.3).
        // (the same form for f32 and pre-quantised rows: the two launches stay bit-identical)
        if (n_ht <= num_cus() && nb <= 32)
            return launch_kq<TR, NC, BT, 8, 2>(mats, n_ht, k, B, brb, col0, s);
        // (round 3, 8 waves for launches of MANY items too: 14336 x 4096 8.30 vs 9.10 us, 28672 13.3 vs 14.2, 32000 (Q6_K) 23.5 vs
        // 24.3, 128256 83.7 vs 84.2 — the 16-wave form keeps them; only at 57344 rows, 14 half-tiles per work-group, does
        // the 8-wave form win (24.4 -> 23.5), which is the expert launch below)
        if (nb <= 16)
            return launch_kq<TR, NC, BT, 16, 1>(mats, n_ht, k, B, brb, col0, s);
        return launch_kq<TR, NC, BT, 16, 2>(mats, n_ht, k, B, brb, col0, s);
    } else {
        // several columns (n = 2..8): 16 waves when a work-group walks several half-tiles (14336 x 4096, n = 4: 24.0 ->
        // 17.5 us), 8 waves of two blocks for single-tile launches (4096 x 4096: the same either way) and for deep rows
        // (4096 x 14336, n = 4: 22.6 vs 23.7 us with 16)
        if (nb <= 16 && n_ht > num_cus())
            return launch_kq<TR, NC, BT, 16, 1>(mats, n_ht, k, B, brb, col0, s);
        if (nb <= 16)
            return launch_kq<TR, NC, BT, 8, 2>(mats, n_ht, k, B, brb, col0, s);
        return launch_kq<TR, NC, BT, 8, 4>(mats, n_ht, k, B, brb, col0, s);
    }
}

template <int NC, int BT>
static hipError_t launch_q4k(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<q4k_traits, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q40(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<q40_traits, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q5k(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<q5k_traits, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q6k(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<q6k_traits, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q2k(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<pk_traits<LFAMD_TYPE_Q2_K>, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q3k(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<pk_traits<LFAMD_TYPE_Q3_K>, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q41(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<pcl_traits<LFAMD_TYPE_Q4_1>, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q50(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<pcl_traits<LFAMD_TYPE_Q5_0>, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q51(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<pcl_traits<LFAMD_TYPE_Q5_1>, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_iq4xs(const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s) {
    return launch_kq_pick<iq4c_traits, NC, BT>(mats, n_ht, k, B, brb, col0, s);
}

template <int NC, int BT>
static hipError_t launch_q80(const q80_mats &mats, long n_total, long k, const void *B, size_t brb, long col0, int vregs32,
                             int precise, hipStream_t s) {
    int nblocks = (int)(k / 32), nquads = (nblocks + 3) / 4;
    size_t smem = (size_t)NC * nquads * X80_QUAD;
    const long rgs = mats.rg_end[GEMV_MAX_MATS - 1];
    unsigned grid = (unsigned)((rgs + Q80_WAVES - 1) / Q80_WAVES);
    // n = 1: the whole problem is one column of 2x1 / 1x1 tiles, so the summation mode is uniform
    const int mode = n_total == 1 ? ((vregs32 || precise) ? 1 : 0) : 2;
#define Q80_GO(MODE)                                                                                                   \
    do {                                                                                                               \
        auto kernel = gemv_q80_kernel<NC, BT, MODE, 16>;                                                               \
        if (smem > 64 * 1024) {                                                                                        \
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (e != hipSuccess)                                                                                       \
                return e;                                                                                              \
        }                                                                                                              \
        kernel<<<grid, Q80_WAVES * 64, smem, s>>>((const uint8_t *)B, brb, col0, nblocks, nquads, n_total, vregs32, precise, \
                                                  mats);                                                               \
    } while (0)
    if (mode == 0)
        Q80_GO(0);
    else if (mode == 1)
        Q80_GO(1);
    else
        Q80_GO(2);
#undef Q80_GO
    return hipGetLastError();
}


#define DISPATCH_NC(FN, BT, nc, ...)                                                                                   \
    switch (nc) {                                                                                                      \
    case 1:                                                                                                            \
        e = FN<1, BT>(__VA_ARGS__);                                                                                    \
        break;                                                                                                         \
    case 2:                                                                                                            \
        e = FN<2, BT>(__VA_ARGS__);                                                                                    \
        break;                                                                                                         \
    case 3:                                                                                                            \
        e = FN<3, BT>(__VA_ARGS__);                                                                                    \
        break;                                                                                                         \
    case 4:                                                                                                            \
        e = FN<4, BT>(__VA_ARGS__);                                                                                    \
        break;                                                                                                         \
    case 5:                                                                                                            \
        e = FN<5, BT>(__VA_ARGS__);                                                                                    \
        break;                                                                                                         \
    case 6:                                                                                                            \
        e = FN<6, BT>(__VA_ARGS__);                                                                                    \
        break;                                                                                                         \
    case 7:                                                                                                            \
        e = FN<7, BT>(__VA_ARGS__);                                                                                    \
        break;                                                                                                         \
    default:                                                                                                           \
        e = FN<8, BT>(__VA_ARGS__);                                                                                    \
        break;                                                                                                         \
    }


// ---- one translation unit per weight type (gemv_*.hip) instantiates its kernels through these stamps (the 8 column
// counts x 2 activation types x chunk variants of every type used to compile serially in one 140-second file)
#define GEMV_GO_ARGS int nc, int f32in, const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb, long col0, hipStream_t s
#define GEMV_INSTANTIATE(NAME, TRAITS, QTYPE)                                                                          \
    hipError_t lfamd_gemv_go_##NAME(GEMV_GO_ARGS) {                                                                    \
        hipError_t e = hipSuccess;                                                                                     \
        if (f32in) {                                                                                                   \
            DISPATCH_NC(launch_##NAME, LFAMD_TYPE_F32, nc, mats, n_ht, k, B, brb, col0, s)                             \
        } else {                                                                                                       \
            DISPATCH_NC(launch_##NAME, QTYPE, nc, mats, n_ht, k, B, brb, col0, s)                                      \
        }                                                                                                              \
        return e;                                                                                                      \
    }
#define GEMV_INSTANTIATE_DUAL(NAME, TRA, TRB)                                                                           \
    hipError_t lfamd_gemv_dual_go_##NAME(int f32in, const gemv_mats &ma, int n_ht_a, const gemv_mats &mb, int n_ht_b, long k, \
                                         const void *B, size_t brb, hipStream_t s) {                                   \
        return f32in ? launch_kq_dual<TRA, TRB, LFAMD_TYPE_F32>(ma, n_ht_a, mb, n_ht_b, k, B, brb, s)                  \
                     : launch_kq_dual<TRA, TRB, LFAMD_TYPE_Q8_K>(ma, n_ht_a, mb, n_ht_b, k, B, brb, s);                \
    }
#define GEMV_INSTANTIATE_IDS_PAIR(NAME, TRAITS)                                                                        \
    hipError_t lfamd_gemv_ids_pair_go_##NAME(int f32in, const gemv_mats &ma, const gemv_mats &mb, int n_ht, long k, const void *Ba, \
                                             const void *Bb, size_t brb, hipStream_t s) {                                \
        return f32in ? launch_kq_ids_pair<TRAITS, LFAMD_TYPE_F32>(ma, mb, n_ht, k, Ba, Bb, brb, s)                       \
                     : launch_kq_ids_pair<TRAITS, LFAMD_TYPE_Q8_K>(ma, mb, n_ht, k, Ba, Bb, brb, s);                     \
    }
#define GEMV_INSTANTIATE_IDS(NAME, TRAITS)                                                                             \
    hipError_t lfamd_gemv_ids_go_##NAME(int f32in, const gemv_mats &mats, int n_ht, long k, const void *B, size_t brb,  \
                                        hipStream_t s) {                                                               \
        return f32in ? launch_kq_ids<TRAITS, LFAMD_TYPE_F32>(mats, n_ht, k, B, brb, s)                                 \
                     : launch_kq_ids<TRAITS, LFAMD_TYPE_Q8_K>(mats, n_ht, k, B, brb, s);                               \
    }
