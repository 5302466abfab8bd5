// gemv.hip — wave-reduction GEMV kernels for the batch-1 (decode) case and small batches n <= 8.
//
// Replaces the reference's mul_mat_vec_q (ggml-cuda.cu.patch:14428-14575: 32-wide warps, dp4a with a
// scalar fallback on gfx950) and follows the CPU path's arithmetic:
//   Q4_K / Q6_K x Q8_K : mul_mat_qX_K_q8_K_T (iqk_mul_mat.inc:601-643) — exact int8 block dots,
//                        f32 scales; the -32 offset of Q6_K is folded into a bsums term like
//                        DequantizerQ6K does (iqk_mul_mat.inc:570-599).
//   Q8_0 x Q8_0        : tinyBLAS_Q0_AVX2::gemm (tinyblas_cpu.h:934-971) BIT-EXACT: 8 f32 lanes per
//                        output, blocks accumulated sequentially with fma (or Kahan), same hsum tree.
//
// These are HBM-bandwidth kernels: weights stream once from the packed layout with one
// global_load_dwordx4 per lane (1 KiB per wave instruction) straight into VGPRs — no LDS round
// trip for the weights (cdna_hip_programming.md §5 "GEMV / M <= 16" row); only the tiny activation
// vector is staged in LDS, byte-permuted once so the int8 dot needs no shuffles.
#include "lfamd_device.h"

// LDS image of one Q8_K activation block for the K-quant GEMVs
#define XBLK 336 // 256 permuted q8 + 32 half-sums (i16) + f32 d + pad
#define XBLK_HB 256
#define XBLK_D 320

// Stage `nc` activation rows (Q8_K, llamafile order) into LDS.  Within every aligned 8-byte group
// the bytes (y0..y7) are stored as (y0,y4,y1,y5 | y2,y6,y3,y7): the order in which
// (x & 0x0F0F0F0F) and ((x>>4) & 0x0F0F0F0F) expose the nibbles of a packed K-step dword.
__device__ static inline void stage_q8k(uint8_t *lds, const uint8_t *B, size_t b_row_bytes, long col0, int nc, int nb) {
    const int groups = nc * nb * 32; // 8-byte groups
    for (int gidx = threadIdx.x; gidx < groups; gidx += blockDim.x) {
        int c = gidx / (nb * 32), r = gidx % (nb * 32);
        int b = r >> 5, grp = r & 31;
        const uint8_t *y = B + (col0 + c) * b_row_bytes + (size_t)b * 292;
        const uint32_t *src = (const uint32_t *)(y + 36 + 8 * grp);
        uint32_t y0 = src[0], y1 = src[1];
        uint32_t p0 = __builtin_amdgcn_perm(y1, y0, 0x05010400);
        uint32_t p1 = __builtin_amdgcn_perm(y1, y0, 0x07030602);
        uint8_t *dst = lds + (size_t)(c * nb + b) * XBLK;
        *(uint2 *)(dst + 8 * grp) = make_uint2(p0, p1);
        // half-sum of these 8 codes
        int hs = sdot4(y0, 0x01010101u, 0);
        hs = sdot4(y1, 0x01010101u, hs);
        *(int16_t *)(dst + XBLK_HB + 2 * grp) = (int16_t)hs;
        if (grp == 0)
            *(float *)(dst + XBLK_D) = *(const float *)y;
    }
}

// ---------------------------------------------------------------------------------------------
// Q4_K.  Work-group = 256 threads = 4 waves, 16 weight rows (half a packed tile); wave w takes
// super-blocks b = w, w+4, ...  lane = (i16 = lane&15, h = (lane>>4)&1, gsel = lane>>5) covers
// groups g = 2*gsel + gi (gi = 0,1) of its row.

template <int NC>
__global__ __launch_bounds__(256) void gemv_q4k_kernel(const uint8_t *__restrict__ A, long m, int nb,
                                                       const uint8_t *__restrict__ B, size_t b_row_bytes, long col0,
                                                       float *__restrict__ C, long ldc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, h = (lane >> 4) & 1, gsel = lane >> 5;
    const long ht = blockIdx.x; // half tile
    const long rt = ht >> 1;
    const int hh = (int)(ht & 1);
    const int slot = h * 32 + hh * 16 + i16; // lane slot inside a 64-lane group image
    const uint8_t *tile0 = A + (size_t)rt * nb * P4K_TILE;

    stage_q8k(lds, B, b_row_bytes, col0, NC, nb);
    __syncthreads();

    float accd[NC], accm[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
        accd[c] = accm[c] = 0.0f;

#pragma unroll 2
    for (int b = wave; b < nb; b += 4) {
        const uint8_t *tile = tile0 + (size_t)b * P4K_TILE;
        uint4 q0 = *(const uint4 *)(tile + (2 * gsel + 0) * 1024 + slot * 16);
        uint4 q1 = *(const uint4 *)(tile + (2 * gsel + 1) * 1024 + slot * 16);
        uint4 hd = *(const uint4 *)(tile + P4K_HDR + (hh * 16 + i16) * 16);
        const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
        uint32_t sc03, sc47, mn03, mn47;
        q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
        // this lane's four sub-blocks: j = 2g + e, g = 2*gsel + gi  ->  j = 4*gsel + 2*gi + e
        const uint32_t scw = gsel ? sc47 : sc03, mnw = gsel ? mn47 : mn03;
        const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint8_t *xb = lds + (size_t)(c * nb + b) * XBLK;
            int sumi = 0, summ = 0;
#pragma unroll
            for (int gi = 0; gi < 2; gi++) {
                const int g = 2 * gsel + gi;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int jj = 2 * gi + e; // byte index inside scw / mnw
                    int isum = 0, hsum = 0;
#pragma unroll
                    for (int d2 = 0; d2 < 2; d2++) {
                        const int dd = 2 * e + d2;
                        const uint32_t x = qw[4 * gi + dd];
                        const int koff = 64 * g + 16 * dd + 8 * h;
                        const uint2 y = *(const uint2 *)(xb + koff);
                        isum = sdot4(x & 0x0F0F0F0F, y.x, isum);
                        isum = sdot4((x >> 4) & 0x0F0F0F0F, y.y, isum);
                        hsum += *(const int16_t *)(xb + XBLK_HB + 2 * (koff >> 3));
                    }
                    sumi += (int)((scw >> (8 * jj)) & 0xff) * isum;
                    summ += (int)((mnw >> (8 * jj)) & 0xff) * hsum;
                }
            }
            const float d8 = *(const float *)(xb + XBLK_D);
            accd[c] = fmaf(d * d8, (float)sumi, accd[c]);
            accm[c] = fmaf(dmin * d8, (float)summ, accm[c]);
        }
    }

    // reduce: 4 lanes per row (h, gsel), then 4 waves through LDS
    __syncthreads(); // all waves done reading the activation image
    float *red = (float *)lds;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        float v = accd[c] - accm[c];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16)
            red[(wave * NC + c) * 16 + lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16 * NC) {
        int c = threadIdx.x >> 4, i = threadIdx.x & 15;
        float v = red[(0 * NC + c) * 16 + i] + red[(1 * NC + c) * 16 + i] + red[(2 * NC + c) * 16 + i] +
                  red[(3 * NC + c) * 16 + i];
        long row = rt * 32 + hh * 16 + i;
        if (row < m)
            C[(col0 + c) * ldc + row] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Q6_K.  Same decomposition; sub-blocks are 16 wide (one per K-step), codes are 6 bit, offset -32
// handled as  sum sc*(dot(code,q8) - 32*sum(q8)).

template <int NC>
__global__ __launch_bounds__(256) void gemv_q6k_kernel(const uint8_t *__restrict__ A, long m, int nb,
                                                       const uint8_t *__restrict__ B, size_t b_row_bytes, long col0,
                                                       float *__restrict__ C, long ldc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, h = (lane >> 4) & 1, gsel = lane >> 5;
    const long ht = blockIdx.x;
    const long rt = ht >> 1;
    const int hh = (int)(ht & 1);
    const int slot = h * 32 + hh * 16 + i16;
    const uint8_t *tile0 = A + (size_t)rt * nb * P6K_TILE;

    stage_q8k(lds, B, b_row_bytes, col0, NC, nb);
    __syncthreads();

    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
        acc[c] = 0.0f;

#pragma unroll 2
    for (int b = wave; b < nb; b += 4) {
        const uint8_t *tile = tile0 + (size_t)b * P6K_TILE;
        uint4 l0 = *(const uint4 *)(tile + (2 * gsel + 0) * 1024 + slot * 16);
        uint4 l1 = *(const uint4 *)(tile + (2 * gsel + 1) * 1024 + slot * 16);
        uint4 hq = *(const uint4 *)(tile + P6K_QH + gsel * 1024 + slot * 16);
        uint2 scb = *(const uint2 *)(tile + P6K_SC + (hh * 16 + i16) * 16 + gsel * 8); // scales of K-steps 8*gsel..+7
        const float d = h2f(*(const uint16_t *)(tile + P6K_D + (hh * 16 + i16) * 2));
        const uint32_t lw[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        const uint32_t hw[4] = {hq.x, hq.y, hq.z, hq.w}; // [gi*2 + e]
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint8_t *xb = lds + (size_t)(c * nb + b) * XBLK;
            int sumi = 0;
#pragma unroll
            for (int gi = 0; gi < 2; gi++) {
                const int g = 2 * gsel + gi;
#pragma unroll
                for (int dd = 0; dd < 4; dd++) {
                    const uint32_t x = lw[4 * gi + dd];
                    uint32_t H = hw[2 * gi + (dd >> 1)];
                    if (dd & 1)
                        H >>= 2;
                    // lo bytes (j0,j4,j1,j5): high fields at bits 4-5 of each byte already
                    const uint32_t clo = (x & 0x0F0F0F0F) | (H & 0x30303030);
                    // hi bytes (j2,j6,j3,j7): fields at bits 8-9 / 0-1 / 24-25 / 16-17
                    const uint32_t chi = ((x >> 4) & 0x0F0F0F0F) | ((H >> 4) & 0x00300030) | ((H << 12) & 0x30003000);
                    const int koff = 64 * g + 16 * dd + 8 * h;
                    const uint2 y = *(const uint2 *)(xb + koff);
                    int isum = sdot4(clo, y.x, 0);
                    isum = sdot4(chi, y.y, isum);
                    const int hs = *(const int16_t *)(xb + XBLK_HB + 2 * (koff >> 3));
                    const int t8 = 4 * gi + dd; // K-step index inside this lane's 8
                    const int sc = (int)(int8_t)(((t8 < 4 ? scb.x : scb.y) >> (8 * (t8 & 3))) & 0xff);
                    sumi += sc * (isum - 32 * hs);
                }
            }
            const float d8 = *(const float *)(xb + XBLK_D);
            acc[c] = fmaf(d * d8, (float)sumi, acc[c]);
        }
    }

    __syncthreads();
    float *red = (float *)lds;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        float v = acc[c];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16)
            red[(wave * NC + c) * 16 + lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16 * NC) {
        int c = threadIdx.x >> 4, i = threadIdx.x & 15;
        float v = red[(0 * NC + c) * 16 + i] + red[(1 * NC + c) * 16 + i] + red[(2 * NC + c) * 16 + i] +
                  red[(3 * NC + c) * 16 + i];
        long row = rt * 32 + hh * 16 + i;
        if (row < m)
            C[(col0 + c) * ldc + row] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Q8_0 x Q8_0, bit-exact restatement of tinyBLAS_Q0_AVX2::gemm (tinyblas_cpu.h:934-971).
// One wave = 8 weight rows, lane = (r = lane>>3, j = lane&7) owns f32 lane j of row r's accumulator
// Cv; blocks are visited in order l = 0..nblocks-1 exactly like the reference's loop, so every
// rounding is the same:  a = f32(dA)*f32(dB);  b = f32(int dot of bytes 4j..4j+3);
// Cv = fma(a, b, Cv)   or, on a PRECISE tile, madder (tinyblas_cpu.h:203-209) with the compiler's
// contraction of sub(mul(a,b),e) into fma(a,b,-e).

#define X80_BLK 36 // LDS: 8 dwords of q8 + f32 d

template <int NC>
__global__ __launch_bounds__(64) void gemv_q80_kernel(const uint8_t *__restrict__ A, long m, long n_total, int nblocks,
                                                      int nquads, const uint8_t *__restrict__ B, size_t b_row_bytes,
                                                      long col0, float *__restrict__ C, long ldc, int vregs32,
                                                      int precise) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x;
    const int r = lane >> 3, j = lane & 7;
    const long rg = blockIdx.x;
    const long row = rg * 8 + r;

    // stage activations: [c][l] -> 8 dwords + f32 scale
    for (int idx = lane; idx < NC * nblocks * 9; idx += 64) {
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
        *(uint32_t *)(lds + (size_t)(c * nblocks + l) * X80_BLK + 4 * w) = v;
    }
    __syncthreads();

    bool kahan[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
        kahan[c] = q0_is_kahan(row < m ? row : m - 1, col0 + c, m, n_total, vregs32 != 0, precise != 0);

    float Cv[NC], Ce[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
        Cv[c] = Ce[c] = 0.0f;

    const uint8_t *tile0 = A + (size_t)rg * nquads * P80_TILE;
#pragma unroll 4
    for (int L = 0; L < nquads; L++) {
        const uint8_t *tile = tile0 + (size_t)L * P80_TILE;
        const uint4 qa = *(const uint4 *)(tile + lane * 16);
        const uint2 ds = *(const uint2 *)(tile + P80_D + r * 8);
        const uint32_t qw[4] = {qa.x, qa.y, qa.z, qa.w};
        const float da[4] = {h2f((uint16_t)(ds.x & 0xffff)), h2f((uint16_t)(ds.x >> 16)), h2f((uint16_t)(ds.y & 0xffff)),
                             h2f((uint16_t)(ds.y >> 16))};
#pragma unroll
        for (int dd = 0; dd < 4; dd++) {
            const int l = 4 * L + dd;
            if (l < nblocks) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const uint8_t *xb = lds + (size_t)(c * nblocks + l) * X80_BLK;
                    const float a = da[dd] * *(const float *)(xb + 32);
                    const float bq = (float)sdot4(qw[dd], *(const uint32_t *)(xb + 4 * j), 0);
                    if (kahan[c]) {
                        const float y = __builtin_fmaf(a, bq, -Ce[c]);
                        const float t = Cv[c] + y;
                        Ce[c] = (t - Cv[c]) - y;
                        Cv[c] = t;
                    } else {
                        Cv[c] = __builtin_fmaf(a, bq, Cv[c]);
                    }
                }
            }
        }
    }
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

template <int NC>
static hipError_t launch_q4k(const void *A, long m, long k, const void *B, size_t brb, long col0, float *C, long ldc,
                             hipStream_t s) {
    int nb = (int)(k / 256);
    size_t smem = (size_t)NC * nb * XBLK;
    if (smem < 4 * NC * 16 * sizeof(float))
        smem = 4 * NC * 16 * sizeof(float);
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)gemv_q4k_kernel<NC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess)
            return e;
    }
    unsigned grid = (unsigned)(((m + 31) / 32) * 2);
    gemv_q4k_kernel<NC><<<grid, 256, smem, s>>>((const uint8_t *)A, m, nb, (const uint8_t *)B, brb, col0, C, ldc);
    return hipGetLastError();
}

template <int NC>
static hipError_t launch_q6k(const void *A, long m, long k, const void *B, size_t brb, long col0, float *C, long ldc,
                             hipStream_t s) {
    int nb = (int)(k / 256);
    size_t smem = (size_t)NC * nb * XBLK;
    if (smem < 4 * NC * 16 * sizeof(float))
        smem = 4 * NC * 16 * sizeof(float);
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)gemv_q6k_kernel<NC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess)
            return e;
    }
    unsigned grid = (unsigned)(((m + 31) / 32) * 2);
    gemv_q6k_kernel<NC><<<grid, 256, smem, s>>>((const uint8_t *)A, m, nb, (const uint8_t *)B, brb, col0, C, ldc);
    return hipGetLastError();
}

template <int NC>
static hipError_t launch_q80(const void *A, long m, long n_total, long k, const void *B, size_t brb, long col0, float *C,
                             long ldc, int vregs32, int precise, hipStream_t s) {
    int nblocks = (int)(k / 32), nquads = (nblocks + 3) / 4;
    size_t smem = (size_t)NC * nblocks * X80_BLK;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)gemv_q80_kernel<NC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess)
            return e;
    }
    unsigned grid = (unsigned)((m + 7) / 8);
    gemv_q80_kernel<NC><<<grid, 64, smem, s>>>((const uint8_t *)A, m, n_total, nblocks, nquads, (const uint8_t *)B, brb,
                                                col0, C, ldc, vregs32, precise);
    return hipGetLastError();
}

// LDS budget: keep one launch's activation image under 160 KiB; otherwise split the columns.
static int max_cols_for(size_t per_col_bytes) {
    size_t cap = 150 * 1024;
    int nc = (int)(cap / (per_col_bytes ? per_col_bytes : 1));
    return nc < 1 ? 0 : (nc > 8 ? 8 : nc);
}

#define DISPATCH_NC(FN, nc, ...)                                                                                       \
    switch (nc) {                                                                                                      \
    case 1:                                                                                                            \
        e = FN<1>(__VA_ARGS__);                                                                                        \
        break;                                                                                                         \
    case 2:                                                                                                            \
        e = FN<2>(__VA_ARGS__);                                                                                        \
        break;                                                                                                         \
    case 3:                                                                                                            \
        e = FN<3>(__VA_ARGS__);                                                                                        \
        break;                                                                                                         \
    case 4:                                                                                                            \
        e = FN<4>(__VA_ARGS__);                                                                                        \
        break;                                                                                                         \
    case 5:                                                                                                            \
        e = FN<5>(__VA_ARGS__);                                                                                        \
        break;                                                                                                         \
    case 6:                                                                                                            \
        e = FN<6>(__VA_ARGS__);                                                                                        \
        break;                                                                                                         \
    case 7:                                                                                                            \
        e = FN<7>(__VA_ARGS__);                                                                                        \
        break;                                                                                                         \
    default:                                                                                                           \
        e = FN<8>(__VA_ARGS__);                                                                                        \
        break;                                                                                                         \
    }

extern "C" hipError_t lfamd_launch_gemv(int Atype, const void *A, long m, long k, const void *B, size_t b_row_bytes, long n,
                                        float *C, long ldc, int vregs32, int precise, hipStream_t s) {
    if (m <= 0 || n <= 0)
        return hipSuccess;
    size_t per_col;
    if (Atype == LFAMD_TYPE_Q8_0)
        per_col = (size_t)(k / 32) * X80_BLK;
    else
        per_col = (size_t)(k / 256) * XBLK;
    int step = max_cols_for(per_col);
    if (step == 0)
        return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    for (long col0 = 0; col0 < n && e == hipSuccess; col0 += step) {
        int nc = (int)((n - col0) < step ? (n - col0) : step);
        if (Atype == LFAMD_TYPE_Q4_K) {
            DISPATCH_NC(launch_q4k, nc, A, m, k, B, b_row_bytes, col0, C, ldc, s)
        } else if (Atype == LFAMD_TYPE_Q6_K) {
            DISPATCH_NC(launch_q6k, nc, A, m, k, B, b_row_bytes, col0, C, ldc, s)
        } else if (Atype == LFAMD_TYPE_Q8_0) {
            DISPATCH_NC(launch_q80, nc, A, m, n, k, B, b_row_bytes, col0, C, ldc, vregs32, precise, s)
        } else {
            return hipErrorInvalidValue;
        }
    }
    return e;
}
