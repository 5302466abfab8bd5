// generic.hip — mat-mul on RAW-layout weights, and the upload-time / per-call canonical images.
//
// The untuned kernels serve what has no resident packed layout: the legacy 32-block types when a row is not a whole
// number of 256-weight groups (Q4_0, Q4_1, Q5_0, Q5_1; tinyblas_cpu_sgemm.inc:45-240, iqk_mul_mat.inc:998-1349) and
// float weights outside the MFMA body's shapes, with the reference's arithmetic: exact integer block dot products, f32
// scales.  Every K-quant and IQ4_XS is packed (lfamd_device.h) and never comes here.
//
// One wave per (weight row, tile of up to 8 activation rows).  Lanes split the row into 16-weight
// units (32 for the legacy 32-blocks), each lane unpacks its unit once and dots it against up to 8
// activation rows, then the wave reduces with DPP/shuffles.
#include "lfamd_device.h"

__device__ static const int8_t kvalues_iq4nl_dev[16] = {-127, -104, -83, -65, -49, -35, -22, -10,
                                                        1,    13,   25,  38,  53,  69,  89,  113};

// Unpack 16 consecutive weights (unit s of 16 in a super-block) to integer codes q, and the unit's
// integer scale / min:  w = d*sc*q - dmin*mn.  Formulas: ggml-cuda.cu.patch:3217-3471, 3684-3699.
template <int TYPE>
__device__ static inline void unpack16(const uint8_t *blk, int s, int q[16], int &sc, int &mn, float &d, float &dmin) {
    if constexpr (TYPE == LFAMD_TYPE_Q2_K) {
        const uint8_t *scales = blk, *qs = blk + 16;
        d = h2f(*(const uint16_t *)(blk + 80));
        dmin = h2f(*(const uint16_t *)(blk + 82));
        sc = scales[s] & 0xF;
        mn = scales[s] >> 4;
        int n = s >> 3, quarter = (s >> 1) & 3, l0 = (s & 1) * 16;
        for (int l = 0; l < 16; l++)
            q[l] = (qs[32 * n + l0 + l] >> (2 * quarter)) & 3;
    } else if constexpr (TYPE == LFAMD_TYPE_Q3_K) {
        const uint8_t *hmask = blk, *qs = blk + 32, *scales = blk + 96;
        d = h2f(*(const uint16_t *)(blk + 108));
        dmin = 0.0f;
        int is = s;
        int us = is < 4    ? (scales[is] & 0xF) | (((scales[is + 8] >> 0) & 3) << 4)
                 : is < 8  ? (scales[is] & 0xF) | (((scales[is + 4] >> 2) & 3) << 4)
                 : is < 12 ? (scales[is - 8] >> 4) | (((scales[is] >> 4) & 3) << 4)
                           : (scales[is - 8] >> 4) | (((scales[is - 4] >> 6) & 3) << 4);
        sc = us - 32;
        mn = 0;
        int n = s >> 3, j = (s >> 1) & 3, l0 = (s & 1) * 16;
        uint8_t m = (uint8_t)(1 << (4 * n + j));
        for (int l = 0; l < 16; l++) {
            int v = (qs[32 * n + l0 + l] >> (2 * j)) & 3;
            q[l] = v - ((hmask[l0 + l] & m) ? 0 : 4);
        }
    }
}

// legacy 32-blocks x Q8_0 / Q8_1 (iqk_mul_mat.inc:998-1349)
template <int TYPE, int TS, bool TYPE1>
__global__ __launch_bounds__(256) void generic_legacy_kernel(const uint8_t *__restrict__ A, long m, int nb,
                                                             const uint8_t *__restrict__ B, size_t b_row_bytes, long n,
                                                             float *__restrict__ C, long ldc) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long col0 = (long)blockIdx.y * 8;
    if (row >= m)
        return;
    const int nc = (int)((n - col0) < 8 ? (n - col0) : 8);
    const uint8_t *arow = A + (size_t)row * nb * TS;
    float acc[8];
    for (int c = 0; c < 8; c++)
        acc[c] = 0.0f;
    for (int u = lane; u < nb; u += 64) {
        const uint8_t *blk = arow + (size_t)u * TS;
        int q[32];
        float d = h2f(*(const uint16_t *)blk), mval = 0.0f;
        if constexpr (TYPE == LFAMD_TYPE_Q4_0) {
            for (int j = 0; j < 16; j++) {
                q[j] = (blk[2 + j] & 15) - 8;
                q[j + 16] = (blk[2 + j] >> 4) - 8;
            }
        } else if constexpr (TYPE == LFAMD_TYPE_Q4_1) {
            mval = h2f(*(const uint16_t *)(blk + 2));
            for (int j = 0; j < 16; j++) {
                q[j] = blk[4 + j] & 15;
                q[j + 16] = blk[4 + j] >> 4;
            }
        } else if constexpr (TYPE == LFAMD_TYPE_Q5_0) {
            uint32_t qh = blk[2] | (blk[3] << 8) | (blk[4] << 16) | ((uint32_t)blk[5] << 24);
            for (int j = 0; j < 16; j++) {
                q[j] = ((blk[6 + j] & 15) | (((qh >> j) & 1) << 4)) - 16;
                q[j + 16] = ((blk[6 + j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) - 16;
            }
        } else if constexpr (TYPE == LFAMD_TYPE_Q5_1) {
            mval = h2f(*(const uint16_t *)(blk + 2));
            uint32_t qh = blk[4] | (blk[5] << 8) | (blk[6] << 16) | ((uint32_t)blk[7] << 24);
            for (int j = 0; j < 16; j++) {
                q[j] = (blk[8 + j] & 15) | (((qh >> j) & 1) << 4);
                q[j + 16] = (blk[8 + j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
            }
        }
        for (int c = 0; c < nc; c++) {
            const uint8_t *yb = B + (col0 + c) * b_row_bytes + (size_t)u * (TYPE1 ? 36 : 34);
            float dy = h2f(*(const uint16_t *)yb);
            const int8_t *q8 = (const int8_t *)(yb + (TYPE1 ? 4 : 2));
            int dot = 0;
            for (int l = 0; l < 32; l++)
                dot += q[l] * (int)q8[l];
            acc[c] = fmaf(d * dy, (float)dot, acc[c]);
            if constexpr (TYPE1)
                acc[c] += mval * h2f(*(const uint16_t *)(yb + 2));
        }
    }
    for (int c = 0; c < nc; c++) {
        float v = acc[c];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_xor(v, off, 64);
        if (lane == 0)
            C[(col0 + c) * ldc + row] = v;
    }
}

// float types (tinyBLAS<> F32/F16/BF16, tinyblas_cpu.h:419-613): f32 accumulate
template <int ATYPE, int BTYPE>
__global__ __launch_bounds__(256) void generic_float_kernel(const uint8_t *__restrict__ A, long m, long k,
                                                            const uint8_t *__restrict__ B, size_t b_row_bytes, long n,
                                                            float *__restrict__ C, long ldc) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long col0 = (long)blockIdx.y * 8;
    if (row >= m)
        return;
    const int nc = (int)((n - col0) < 8 ? (n - col0) : 8);
    constexpr int AS = ATYPE == LFAMD_TYPE_F32 ? 4 : 2;
    const uint8_t *arow = A + (size_t)row * k * AS;
    float acc[8];
    for (int c = 0; c < 8; c++)
        acc[c] = 0.0f;
    auto ld = [](int type, const uint8_t *p, long idx) -> float {
        if (type == LFAMD_TYPE_F32)
            return ((const float *)p)[idx];
        uint16_t h = ((const uint16_t *)p)[idx];
        if (type == LFAMD_TYPE_F16)
            return h2f(h);
        return __builtin_bit_cast(float, (uint32_t)h << 16);
    };
    for (long l = lane; l < k; l += 64) {
        float a = ld(ATYPE, arow, l);
        for (int c = 0; c < nc; c++)
            acc[c] = fmaf(a, ld(BTYPE, B + (col0 + c) * b_row_bytes, l), acc[c]);
    }
    for (int c = 0; c < nc; c++) {
        float v = acc[c];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_xor(v, off, 64);
        if (lane == 0)
            C[(col0 + c) * ldc + row] = v;
    }
}

extern "C" hipError_t lfamd_launch_generic(int Atype, const void *A, long m, long k, int Btype, const void *B,
                                           size_t b_row_bytes, long n, float *C, long ldc, hipStream_t s) {
    if (m <= 0 || n <= 0)
        return hipSuccess;
    dim3 grid((unsigned)((m + 3) / 4), (unsigned)((n + 7) / 8));
    const uint8_t *a = (const uint8_t *)A, *b = (const uint8_t *)B;
    int nb32 = (int)(k / 32);
#define LG(T, TS, T1) generic_legacy_kernel<T, TS, T1><<<grid, 256, 0, s>>>(a, m, nb32, b, b_row_bytes, n, C, ldc)
#define FL(TA, TB) generic_float_kernel<TA, TB><<<grid, 256, 0, s>>>(a, m, k, b, b_row_bytes, n, C, ldc)
    switch (Atype) {
    case LFAMD_TYPE_Q4_0:
        LG(LFAMD_TYPE_Q4_0, 18, false);
        break;
    case LFAMD_TYPE_Q4_1:
        LG(LFAMD_TYPE_Q4_1, 20, true);
        break;
    case LFAMD_TYPE_Q5_0:
        LG(LFAMD_TYPE_Q5_0, 22, false);
        break;
    case LFAMD_TYPE_Q5_1:
        LG(LFAMD_TYPE_Q5_1, 24, true);
        break;
    case LFAMD_TYPE_F32:
        if (Btype != LFAMD_TYPE_F32)
            return hipErrorInvalidValue;
        FL(LFAMD_TYPE_F32, LFAMD_TYPE_F32);
        break;
    case LFAMD_TYPE_F16:
        if (Btype == LFAMD_TYPE_F32)
            FL(LFAMD_TYPE_F16, LFAMD_TYPE_F32);
        else if (Btype == LFAMD_TYPE_F16)
            FL(LFAMD_TYPE_F16, LFAMD_TYPE_F16);
        else
            return hipErrorInvalidValue;
        break;
    case LFAMD_TYPE_BF16:
        if (Btype == LFAMD_TYPE_F32)
            FL(LFAMD_TYPE_BF16, LFAMD_TYPE_F32);
        else if (Btype == LFAMD_TYPE_BF16)
            FL(LFAMD_TYPE_BF16, LFAMD_TYPE_BF16);
        else
            return hipErrorInvalidValue;
        break;
    default:
        return hipErrorInvalidValue;
    }
#undef LG
#undef FL
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Upload-time canonicalisation of Q2_K / Q3_K (lfamd_pack_weights): RAW blocks -> PCK tiles (lfamd_device.h), the
// resident layout both the MFMA GEMM and the decode GEMV read.  One thread per output dword; reads 84 / 110 bytes per
// 256 weights, writes 164.
template <int TYPE>
__global__ void wprep16_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb, uint8_t *__restrict__ out,
                               long n_tiles) {
    constexpr int OFF = TYPE == LFAMD_TYPE_Q3_K ? 4 : 0; // code = q + OFF in 0..15
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tile = tid / 1312; // 1024 code dwords + 128 scale + 128 min + 32 {d, dmin}
    const int w = (int)(tid % 1312);
    if (tile >= n_tiles)
        return;
    const long rt = tile / nb;
    const int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * PCK_TILE);
    constexpr size_t bs = TYPE == LFAMD_TYPE_Q2_K ? sizeof(lfamd_block_q2_K) : sizeof(lfamd_block_q3_K);
    int q[16], sc, mn;
    float d, dmin;
    if (w < 1024) {
        const int g = w >> 8, lane = (w >> 2) & 63, dd = w & 3;
        const int i = lane & 31, h = lane >> 5;
        const long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            unpack16<TYPE>(raw + row * raw_row_bytes + (size_t)b * bs, 4 * g + dd, q, sc, mn, d, dmin);
            for (int j = 0; j < 8; j++)
                v |= (uint32_t)(q[8 * h + j] + OFF) << (4 * NIBPOS(j));
        }
        dst[w] = v;
    } else if (w < 1280) {
        const int s4 = w - 1024, mins = s4 >= 128;
        const int i = (s4 & 127) >> 2, u = s4 & 3;
        const long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows)
            for (int e = 0; e < 4; e++) {
                unpack16<TYPE>(raw + row * raw_row_bytes + (size_t)b * bs, 4 * u + e, q, sc, mn, d, dmin);
                v |= (uint32_t)((mins ? mn : sc) & 0xff) << (8 * e);
            }
        dst[w] = v;
    } else {
        const int i = w - 1280;
        const long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            unpack16<TYPE>(raw + row * raw_row_bytes + (size_t)b * bs, 0, q, sc, mn, d, dmin);
            v = (uint32_t)f2h_bits(d) | ((uint32_t)f2h_bits(dmin) << 16); // both were f16 in the block: exact round trip
        }
        dst[w] = v;
    }
}

// Legacy 32-block types (Q4_1, Q5_0, Q5_1) -> PCL tiles.  Weight l of a block: low nibble of qs[l] (l < 16) or high
// nibble of qs[l - 16], fifth bit = bit l of qh (iqk_mul_mat.inc:1241-1283).
template <int TYPE>
__global__ void wprep32_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb, uint8_t *__restrict__ out,
                               long n_tiles) {
    constexpr bool HAS_M = TYPE == LFAMD_TYPE_Q4_1 || TYPE == LFAMD_TYPE_Q5_1;
    constexpr bool HAS_H = TYPE == LFAMD_TYPE_Q5_0 || TYPE == LFAMD_TYPE_Q5_1;
    constexpr int BS = TYPE == LFAMD_TYPE_Q4_1 ? 20 : TYPE == LFAMD_TYPE_Q5_0 ? 22 : 24;
    constexpr int QH_OFF = HAS_M ? 4 : 2, QS_OFF = QH_OFF + (HAS_H ? 4 : 0);
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tile = tid / 1536; // 1024 code dwords + 128 d + 128 m + 256 fifth-bit dwords
    const int w = (int)(tid % 1536);
    if (tile >= n_tiles)
        return;
    const long rt = tile / nb;
    const int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * PCL_TILE);
    uint32_t v = 0;
    if (w < 1024) {
        const int g = w >> 8, lane = (w >> 2) & 63, dd = w & 3;
        const int i = lane & 31, h = lane >> 5;
        const long row = rt * 32 + i;
        if (row < rows) {
            const uint8_t *blk0 = raw + row * raw_row_bytes + (size_t)b * 8 * BS;
            const int t = 4 * g + dd;
            for (int j = 0; j < 8; j++) {
                const int k = 16 * t + 8 * h + j, bl = k >> 5, l = k & 31;
                const uint8_t byte = blk0[bl * BS + QS_OFF + (l & 15)];
                v |= (uint32_t)(l < 16 ? (byte & 15) : (byte >> 4)) << (4 * NIBPOS(j));
            }
        }
    } else if (w < 1280) {
        const int s4 = w - 1024, is_m = s4 >= 128;
        const int i = (s4 & 127) >> 2, q = s4 & 3;
        const long row = rt * 32 + i;
        if (row < rows && (!is_m || HAS_M)) {
            const uint8_t *blk0 = raw + row * raw_row_bytes + (size_t)b * 8 * BS;
            const uint16_t lo = *(const uint16_t *)(blk0 + (2 * q) * BS + (is_m ? 2 : 0));
            const uint16_t hi = *(const uint16_t *)(blk0 + (2 * q + 1) * BS + (is_m ? 2 : 0));
            v = (uint32_t)lo | ((uint32_t)hi << 16);
        }
    } else {
        const int s = w - 1280, lane = s >> 2, g = s & 3;
        const int i = lane & 31, h = lane >> 5;
        const long row = rt * 32 + i;
        if (row < rows && HAS_H) {
            const uint8_t *blk0 = raw + row * raw_row_bytes + (size_t)b * 8 * BS;
            for (int dd = 0; dd < 4; dd++) {
                const int t = 4 * g + dd;
                for (int j = 0; j < 8; j++) {
                    const int k = 16 * t + 8 * h + j, bl = k >> 5, l = k & 31;
                    const uint8_t *qh = blk0 + bl * BS + QH_OFF;
                    const uint32_t bit = (qh[l >> 3] >> (l & 7)) & 1u;
                    v |= bit << (4 * q5hpos(j) + dd);
                }
            }
        }
    }
    dst[w] = v;
}

extern "C" size_t lfamd_wprep32_bytes(long rows, long cols) {
    return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * PCL_TILE;
}

extern "C" hipError_t lfamd_launch_wprep32(int type, const void *raw, size_t raw_row_bytes, long rows, long cols, void *out,
                                           hipStream_t s) {
    const int nb = (int)(cols / 256);
    const long n_tiles = ((rows + 31) / 32) * nb;
    const long threads = n_tiles * 1536;
    const size_t rrb = raw_row_bytes;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (type == LFAMD_TYPE_Q4_1)
        wprep32_kernel<LFAMD_TYPE_Q4_1><<<grid, 256, 0, s>>>((const uint8_t *)raw, rrb, rows, nb, (uint8_t *)out, n_tiles);
    else if (type == LFAMD_TYPE_Q5_0)
        wprep32_kernel<LFAMD_TYPE_Q5_0><<<grid, 256, 0, s>>>((const uint8_t *)raw, rrb, rows, nb, (uint8_t *)out, n_tiles);
    else if (type == LFAMD_TYPE_Q5_1)
        wprep32_kernel<LFAMD_TYPE_Q5_1><<<grid, 256, 0, s>>>((const uint8_t *)raw, rrb, rows, nb, (uint8_t *)out, n_tiles);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// IQ4_XS -> PC8 tiles (kvalues_iq4nl applied here, so the GEMM sees plain integers)
__global__ void wprep8_iq4xs_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb, uint8_t *__restrict__ out,
                                    long n_tiles) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tile = tid / 2176; // 2048 code dwords + 128 header dwords
    const int w = (int)(tid % 2176);
    if (tile >= n_tiles)
        return;
    const long rt = tile / nb;
    const int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * PC8_TILE);
    uint32_t v = 0;
    if (w < 2048) {
        const int g2 = w >> 8, lane = (w >> 2) & 63, e = w & 3;
        const int i = lane & 31, h = lane >> 5;
        const long row = rt * 32 + i;
        if (row < rows) {
            const lfamd_block_iq4_xs *blk = (const lfamd_block_iq4_xs *)(raw + row * raw_row_bytes) + b;
            const int t = 2 * g2 + (e >> 1);
            for (int jj = 0; jj < 4; jj++) {
                const int k = 16 * t + 8 * h + 4 * (e & 1) + jj, ib = k >> 5, l = k & 31;
                const uint8_t byte = blk->qs[16 * ib + (l & 15)];
                const int val = kvalues_iq4nl_dev[l < 16 ? (byte & 15) : (byte >> 4)];
                v |= (uint32_t)((val + 128) & 0xff) << (8 * jj);
            }
        }
    } else {
        const int s = w - 2048, i = s >> 2, q = s & 3;
        const long row = rt * 32 + i;
        if (row < rows && q < 3) {
            const lfamd_block_iq4_xs *blk = (const lfamd_block_iq4_xs *)(raw + row * raw_row_bytes) + b;
            if (q < 2) {
                for (int e = 0; e < 4; e++) {
                    const int ib = 4 * q + e;
                    const int ls = ((blk->scales_l[ib / 2] >> (4 * (ib % 2))) & 0xf) | (((blk->scales_h >> (2 * ib)) & 3) << 4);
                    v |= (uint32_t)((ls - 32) & 0xff) << (8 * e);
                }
            } else {
                v = blk->d;
            }
        }
    }
    dst[w] = v;
}

extern "C" size_t lfamd_wprep8_bytes(long rows, long cols) {
    return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * PC8_TILE;
}

extern "C" hipError_t lfamd_launch_wprep8(int type, const void *raw, size_t raw_row_bytes, long rows, long cols, void *out,
                                          hipStream_t s) {
    if (type != LFAMD_TYPE_IQ4_XS)
        return hipErrorInvalidValue;
    const int nb = (int)(cols / 256);
    const long n_tiles = ((rows + 31) / 32) * nb;
    const long threads = n_tiles * 2176;
    wprep8_iq4xs_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nb, (uint8_t *)out,
                                                                        n_tiles);
    return hipGetLastError();
}

// Resident compact images of Q2_K / Q3_K (lfamd_device.h: PK2 / PK3): RAW blocks -> compact tiles.  One thread per output dword.
template <int TYPE>
__global__ void pk_pack_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb, uint8_t *__restrict__ out,
                               long n_tiles) {
    constexpr bool Q3 = TYPE == LFAMD_TYPE_Q3_K;
    constexpr int OFF = Q3 ? 4 : 0, TILE = Q3 ? PK3_TILE : PK2_TILE, NDW = TILE / 4;
    constexpr int SC0 = (Q3 ? PK3_SC : PK2_SC) / 4, D0 = (Q3 ? PK3_D : PK2_D) / 4;
    constexpr size_t bs = Q3 ? sizeof(lfamd_block_q3_K) : sizeof(lfamd_block_q2_K);
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tile = tid / NDW;
    const int w = (int)(tid % NDW);
    if (tile >= n_tiles)
        return;
    const long rt = tile / nb;
    const int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * TILE);
    int q[16], sc, mn;
    float d, dmin;
    uint32_t v = 0;
    if (w < 512) { // codes, low two bits: K-steps 8 gsel + 2 u (bits 0-1 of a nibble) and + 1 (bits 2-3)
        const int gsel = w >> 8, lane = (w >> 2) & 63, u = w & 3;
        const int i = lane & 31, h = lane >> 5;
        const long row = rt * 32 + i;
        if (row < rows)
            for (int e = 0; e < 2; e++) {
                unpack16<TYPE>(raw + row * raw_row_bytes + (size_t)b * bs, 8 * gsel + 2 * u + e, q, sc, mn, d, dmin);
                for (int j = 0; j < 8; j++)
                    v |= (uint32_t)((q[8 * h + j] + OFF) & 3) << (4 * NIBPOS(j) + 2 * e);
            }
    } else if (Q3 && w < 768) { // third bits: dword x of [gsel][lane] = K-steps 8 gsel + 4 x + s at bit 4 NIBPOS(j) + s
        const int s8 = w - 512, gsel = s8 >> 7, lane = (s8 >> 1) & 63, x = s8 & 1;
        const int i = lane & 31, h = lane >> 5;
        const long row = rt * 32 + i;
        if (row < rows)
            for (int e = 0; e < 4; e++) {
                unpack16<TYPE>(raw + row * raw_row_bytes + (size_t)b * bs, 8 * gsel + 4 * x + e, q, sc, mn, d, dmin);
                for (int j = 0; j < 8; j++)
                    v |= (uint32_t)(((q[8 * h + j] + OFF) >> 2) & 1) << (4 * NIBPOS(j) + e);
            }
    } else if (w >= SC0 && w < SC0 + 128) { // 16 scale bytes per row
        const int s4 = w - SC0, i = s4 >> 2, u = s4 & 3;
        const long row = rt * 32 + i;
        if (row < rows)
            for (int e = 0; e < 4; e++) {
                unpack16<TYPE>(raw + row * raw_row_bytes + (size_t)b * bs, 4 * u + e, q, sc, mn, d, dmin);
                v |= (uint32_t)((Q3 ? sc : (sc | (mn << 4))) & 0xff) << (8 * e);
            }
    } else if (w >= D0) {
        const int i = w - D0;
        const long row = rt * 32 + i;
        if (row < rows) {
            unpack16<TYPE>(raw + row * raw_row_bytes + (size_t)b * bs, 0, q, sc, mn, d, dmin);
            v = (uint32_t)f2h_bits(d) | ((uint32_t)f2h_bits(dmin) << 16);
        }
    }
    dst[w] = v;
}

// Batches: the compact image -> the canonical PCK image the MFMA body reads, into the caller's workspace (per call).
template <int TYPE>
__global__ void pk_expand_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long n_tiles) {
    constexpr bool Q3 = TYPE == LFAMD_TYPE_Q3_K;
    constexpr int TILE = Q3 ? PK3_TILE : PK2_TILE, SC0 = Q3 ? PK3_SC : PK2_SC, D0 = Q3 ? PK3_D : PK2_D;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tile = tid / 1312; // (wprep16_kernel's dword order)
    const int w = (int)(tid % 1312);
    if (tile >= n_tiles)
        return;
    const uint8_t *src = in + tile * TILE;
    uint32_t *dst = (uint32_t *)(out + tile * PCK_TILE);
    uint32_t v;
    if (w < 1024) {
        const int g = w >> 8, lane = (w >> 2) & 63, dd = w & 3;
        const int t = 4 * g + dd, gsel = t >> 3, t8 = t & 7;
        const uint32_t c = *(const uint32_t *)(src + gsel * 1024 + lane * 16 + (t8 >> 1) * 4);
        v = ((t8 & 1) ? (c >> 2) : c) & 0x33333333u;
        if constexpr (Q3) {
            const uint32_t hb = *(const uint32_t *)(src + PK3_HB + gsel * 512 + lane * 8 + (t8 >> 2) * 4);
            v |= ((hb >> (t8 & 3)) & 0x11111111u) << 2;
        }
    } else if (w < 1280) {
        const int s4 = w - 1024, mins = s4 >= 128;
        const int i = (s4 & 127) >> 2, u = s4 & 3;
        const uint32_t sb = *(const uint32_t *)(src + SC0 + i * 16 + u * 4);
        v = Q3 ? (mins ? 0u : sb) : (mins ? (sb >> 4) & 0x0F0F0F0Fu : sb & 0x0F0F0F0Fu);
    } else {
        v = *(const uint32_t *)(src + D0 + (w - 1280) * 4);
    }
    dst[w] = v;
}

// IQ4_XS resident image (PK4X = P4K_TILE bytes per 32 x 256): the codebook INDICES on the P4K nibble lattice (K-step t = 4 g + dd of
// lane (i, h): element j at bit 4 NIBPOS(j)), then per row {8 int8 sub-block scales (ls - 32), f16 d, pad} — 144 bytes per 256
// weights against 136 in the file; the decode GEMV looks the 16-entry codebook up in registers.
__global__ void pk4x_pack_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb, uint8_t *__restrict__ out,
                                 long n_tiles) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tile = tid / 1152; // 1024 index dwords + 128 header dwords
    const int w = (int)(tid % 1152);
    if (tile >= n_tiles)
        return;
    const long rt = tile / nb;
    const int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * P4K_TILE);
    uint32_t v = 0;
    if (w < 1024) {
        const int g = w >> 8, lane = (w >> 2) & 63, dd = w & 3;
        const int i = lane & 31, h = lane >> 5;
        const long row = rt * 32 + i;
        if (row < rows) {
            const lfamd_block_iq4_xs *blk = (const lfamd_block_iq4_xs *)(raw + row * raw_row_bytes) + b;
            const int t = 4 * g + dd;
            for (int j = 0; j < 8; j++) {
                const int k = 16 * t + 8 * h + j, ib = k >> 5, l = k & 31;
                const uint8_t byte = blk->qs[16 * ib + (l & 15)];
                v |= (uint32_t)(l < 16 ? (byte & 15) : (byte >> 4)) << (4 * NIBPOS(j));
            }
        }
    } else {
        const int s4 = w - 1024, i = s4 >> 2, q = s4 & 3; // header dwords as in the PC8 image: scales 0-3, 4-7, d, pad
        const long row = rt * 32 + i;
        if (row < rows && q < 3) {
            const lfamd_block_iq4_xs *blk = (const lfamd_block_iq4_xs *)(raw + row * raw_row_bytes) + b;
            if (q < 2) {
                for (int e = 0; e < 4; e++) {
                    const int ib = 4 * q + e;
                    const int ls = ((blk->scales_l[ib / 2] >> (4 * (ib % 2))) & 0xf) | (((blk->scales_h >> (2 * ib)) & 3) << 4);
                    v |= (uint32_t)((ls - 32) & 0xff) << (8 * e);
                }
            } else {
                v = blk->d;
            }
        }
    }
    dst[w] = v;
}

// batches: compact image -> the PC8 byte image (codebook value + 128) the MFMA body reads, per call, into the workspace
__global__ void pk4x_expand_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long n_tiles, long rows, int nb) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tile = tid / 2176; // (wprep8_iq4xs_kernel's dword order)
    const int w = (int)(tid % 2176);
    if (tile >= n_tiles)
        return;
    const uint8_t *src = in + tile * P4K_TILE;
    uint32_t *dst = (uint32_t *)(out + tile * PC8_TILE);
    uint32_t v = 0;
    if (w < 2048) {
        const int g2 = w >> 8, lane = (w >> 2) & 63, e = w & 3;
        const int t = 2 * g2 + (e >> 1);
        const uint32_t x = *(const uint32_t *)(src + (t >> 2) * 1024 + lane * 16 + (t & 3) * 4);
        if ((tile / nb) * 32 + (lane & 31) < rows) // (rows past the matrix: zero bytes, like the builder from GGUF rows)
        for (int jj = 0; jj < 4; jj++) {
            const int j = 4 * (e & 1) + jj;
            const int val = kvalues_iq4nl_dev[(x >> (4 * NIBPOS(j))) & 15];
            v |= (uint32_t)((val + 128) & 0xff) << (8 * jj);
        }
    } else {
        v = *(const uint32_t *)(src + P4K_HDR + (w - 2048) * 4);
    }
    dst[w] = v;
}

extern "C" hipError_t lfamd_launch_pk4x_pack(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s) {
    const int nb = (int)(cols / 256);
    const long n_tiles = ((rows + 31) / 32) * nb;
    if (n_tiles == 0)
        return hipSuccess;
    const long threads = n_tiles * 1152;
    pk4x_pack_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nb, (uint8_t *)out, n_tiles);
    return hipGetLastError();
}

extern "C" hipError_t lfamd_launch_pk4x_expand(const void *packed, long rows, long cols, void *out, hipStream_t s) {
    const long n_tiles = ((rows + 31) / 32) * (cols / 256);
    if (n_tiles == 0)
        return hipSuccess;
    const long threads = n_tiles * 2176;
    pk4x_expand_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)packed, (uint8_t *)out, n_tiles, rows, (int)(cols / 256));
    return hipGetLastError();
}

extern "C" size_t lfamd_pk_bytes(int type, long rows, long cols) {
    return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * (type == LFAMD_TYPE_Q3_K ? PK3_TILE : PK2_TILE);
}

extern "C" hipError_t lfamd_launch_pk_pack(int type, const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s) {
    const int nb = (int)(cols / 256);
    const long n_tiles = ((rows + 31) / 32) * nb;
    if (n_tiles == 0)
        return hipSuccess;
    if (type == LFAMD_TYPE_Q2_K) {
        const long threads = n_tiles * (PK2_TILE / 4);
        pk_pack_kernel<LFAMD_TYPE_Q2_K><<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nb,
                                                                                         (uint8_t *)out, n_tiles);
    } else if (type == LFAMD_TYPE_Q3_K) {
        const long threads = n_tiles * (PK3_TILE / 4);
        pk_pack_kernel<LFAMD_TYPE_Q3_K><<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nb,
                                                                                         (uint8_t *)out, n_tiles);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" hipError_t lfamd_launch_pk_expand(int type, const void *packed, long rows, long cols, void *out, hipStream_t s) {
    const long n_tiles = ((rows + 31) / 32) * (cols / 256);
    if (n_tiles == 0)
        return hipSuccess;
    const long threads = n_tiles * 1312;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (type == LFAMD_TYPE_Q2_K)
        pk_expand_kernel<LFAMD_TYPE_Q2_K><<<grid, 256, 0, s>>>((const uint8_t *)packed, (uint8_t *)out, n_tiles);
    else if (type == LFAMD_TYPE_Q3_K)
        pk_expand_kernel<LFAMD_TYPE_Q3_K><<<grid, 256, 0, s>>>((const uint8_t *)packed, (uint8_t *)out, n_tiles);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

extern "C" size_t lfamd_wprep16_bytes(long rows, long cols) {
    return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * PCK_TILE;
}

extern "C" hipError_t lfamd_launch_wprep16(int type, const void *raw, size_t raw_row_bytes, long rows, long cols, void *out,
                                           hipStream_t s) {
    const int nb = (int)(cols / 256);
    const long n_tiles = ((rows + 31) / 32) * nb;
    const long threads = n_tiles * 1312;
    const size_t rrb = raw_row_bytes;
    if (type == LFAMD_TYPE_Q2_K)
        wprep16_kernel<LFAMD_TYPE_Q2_K><<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, rrb, rows, nb,
                                                                                         (uint8_t *)out, n_tiles);
    else if (type == LFAMD_TYPE_Q3_K)
        wprep16_kernel<LFAMD_TYPE_Q3_K><<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, rrb, rows, nb,
                                                                                         (uint8_t *)out, n_tiles);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
