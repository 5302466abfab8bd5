// pack.hip — device-side re-layout of GGUF weight tensors into the packed tile formats
// (lfamd_device.h) and preparation of quantised activations for the MFMA GEMM.
//
// Counterpart of the reference's weight upload, ggml_backend_cuda_buffer_set_tensor
// (ggml-cuda.cu.patch:16971-16977): the backend owns the device copy, so it may choose its layout.
#include "lfamd_device.h"

// ---------------------------------------------------------------------------------------------
// Q4_K -> P4K.  One thread per output dword of the qs part, one per 16-byte header.

__global__ void pack_q4k_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb,
                                uint8_t *__restrict__ out, long n_tiles) {
    long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // per tile: 1024 qs dwords + 32 header slots (4 dwords each) = 1152 dwords
    long tile = tid / 1152;
    int w = (int)(tid % 1152);
    if (tile >= n_tiles)
        return;
    long rt = tile / nb;
    int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * P4K_TILE);
    if (w < 1024) {
        int g = w >> 8, lane = (w >> 2) & 63, dd = w & 3;
        int i = lane & 31, h = lane >> 5;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const lfamd_block_q4_K *blk = (const lfamd_block_q4_K *)(raw + row * raw_row_bytes) + b;
            int t = 4 * g + dd;
            for (int j = 0; j < 8; j++) {
                int k = 16 * t + 8 * h + j;
                int c = k >> 6, wi = k & 63;
                uint8_t byte = blk->qs[32 * c + (wi & 31)];
                uint32_t nib = wi < 32 ? (byte & 15u) : (uint32_t)(byte >> 4);
                v |= nib << (4 * NIBPOS(j));
            }
        }
        dst[w] = v;
    } else {
        int s = w - 1024; // 0..127: header dword
        int i = s >> 2, q = s & 3;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const uint8_t *blk = raw + row * raw_row_bytes + (size_t)b * sizeof(lfamd_block_q4_K);
            // raw blocks are only 2-byte aligned in general (row strides are multiples of 144, so
            // 4-byte here, but stay safe)
            const uint16_t *p = (const uint16_t *)(blk + 4 * q);
            v = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
        }
        dst[1024 + s] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Q4_0 -> P40 (cols % 256 == 0): the P4K nibble image (K-step dword = the 8 codes of k = 16t + 8h + j) and, as
// header, the eight f16 block scales of the row's 256 weights.  block_q4_0 = {d, qs[16]}: weight l of a block is the
// low nibble of qs[l] (l < 16) or the high nibble of qs[l - 16]; value d*(q - 8).

__global__ void pack_q40_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb,
                                uint8_t *__restrict__ out, long n_tiles) {
    long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long tile = tid / 1152;
    int w = (int)(tid % 1152);
    if (tile >= n_tiles)
        return;
    long rt = tile / nb;
    int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * P4K_TILE);
    if (w < 1024) {
        int g = w >> 8, lane = (w >> 2) & 63, dd = w & 3;
        int i = lane & 31, h = lane >> 5;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const lfamd_block_q4_0 *blk = (const lfamd_block_q4_0 *)(raw + row * raw_row_bytes) + (size_t)b * 8;
            int t = 4 * g + dd;
            for (int j = 0; j < 8; j++) {
                int k = 16 * t + 8 * h + j;
                int bl = k >> 5, l = k & 31;
                uint8_t byte = blk[bl].qs[l & 15];
                uint32_t nib = l < 16 ? (byte & 15u) : (uint32_t)(byte >> 4);
                v |= nib << (4 * NIBPOS(j));
            }
        }
        dst[w] = v;
    } else {
        int s = w - 1024; // row i, scale pair q: blocks 2q, 2q+1
        int i = s >> 2, q = s & 3;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const lfamd_block_q4_0 *blk = (const lfamd_block_q4_0 *)(raw + row * raw_row_bytes) + (size_t)b * 8;
            v = (uint32_t)blk[2 * q].d | ((uint32_t)blk[2 * q + 1].d << 16);
        }
        dst[1024 + s] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Q5_K -> P5K: the P4K image of the low nibbles and the header, then one dword of fifth bits per (lane, group).
// block_q5_K = {d, dmin, scales[12], qh[32], qs[128]}: weight l of sub-block j has its fifth bit at bit j of qh[l].

__global__ void pack_q5k_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb,
                                uint8_t *__restrict__ out, long n_tiles) {
    long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long tile = tid / 1408; // 1024 qs dwords + 128 header dwords + 256 fifth-bit dwords
    int w = (int)(tid % 1408);
    if (tile >= n_tiles)
        return;
    long rt = tile / nb;
    int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * P5K_TILE);
    if (w < 1024) {
        int g = w >> 8, lane = (w >> 2) & 63, dd = w & 3;
        int i = lane & 31, h = lane >> 5;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const lfamd_block_q5_K *blk = (const lfamd_block_q5_K *)(raw + row * raw_row_bytes) + b;
            int t = 4 * g + dd;
            for (int j = 0; j < 8; j++) {
                int k = 16 * t + 8 * h + j;
                int c = k >> 6, wi = k & 63;
                uint8_t byte = blk->qs[32 * c + (wi & 31)];
                uint32_t nib = wi < 32 ? (byte & 15u) : (uint32_t)(byte >> 4);
                v |= nib << (4 * NIBPOS(j));
            }
        }
        dst[w] = v;
    } else if (w < 1152) {
        int s = w - 1024; // header dword: the block's first 16 bytes {d, dmin, scales[12]}
        int i = s >> 2, q = s & 3;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const uint8_t *blk = raw + row * raw_row_bytes + (size_t)b * sizeof(lfamd_block_q5_K);
            const uint16_t *p = (const uint16_t *)(blk + 4 * q);
            v = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
        }
        dst[1024 + s] = v;
    } else {
        int s = w - 1152; // lane * 4 + g
        int lane = s >> 2, g = s & 3;
        int i = lane & 31, h = lane >> 5;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const lfamd_block_q5_K *blk = (const lfamd_block_q5_K *)(raw + row * raw_row_bytes) + b;
            for (int dd = 0; dd < 4; dd++) {
                int t = 4 * g + dd, sub = t >> 1;
                for (int j = 0; j < 8; j++) {
                    int l = 16 * (t & 1) + 8 * h + j;
                    uint32_t bit = (blk->qh[l] >> sub) & 1u;
                    v |= bit << (4 * q5hpos(j) + dd);
                }
            }
        }
        dst[1152 + s] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Q6_K -> P6K

__device__ static inline int q6k_code(const lfamd_block_q6_K *blk, int k) { // 0..63
    int p = k >> 7, wi = k & 127, l = wi & 31, quarter = wi >> 5;
    uint8_t qlb = blk->ql[64 * p + (quarter & 1) * 32 + l];
    int nib = quarter < 2 ? (qlb & 15) : (qlb >> 4);
    int hi = (blk->qh[32 * p + l] >> (2 * quarter)) & 3;
    return nib | (hi << 4);
}

__global__ void pack_q6k_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nb,
                                uint8_t *__restrict__ out, long n_tiles) {
    long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // per tile: 1024 ql dwords + 512 qh dwords + 128 scale dwords + 16 d dwords = 1680 dwords
    long tile = tid / 1680;
    int w = (int)(tid % 1680);
    if (tile >= n_tiles)
        return;
    long rt = tile / nb;
    int b = (int)(tile % nb);
    uint32_t *dst = (uint32_t *)(out + tile * P6K_TILE);
    if (w < 1024) {
        int g = w >> 8, lane = (w >> 2) & 63, dd = w & 3;
        int i = lane & 31, h = lane >> 5;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const lfamd_block_q6_K *blk = (const lfamd_block_q6_K *)(raw + row * raw_row_bytes) + b;
            int t = 4 * g + dd;
            for (int j = 0; j < 8; j++)
                v |= (uint32_t)(q6k_code(blk, 16 * t + 8 * h + j) & 15) << (4 * NIBPOS(j));
        }
        dst[w] = v;
    } else if (w < 1536) {
        int s = w - 1024;
        int gg = s >> 8, lane = (s >> 2) & 63, q = s & 3; // dword q of the lane's 16 B: group 2gg+(q>>1), pair e=q&1
        int i = lane & 31, h = lane >> 5;
        int g = 2 * gg + (q >> 1), e = q & 1;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const lfamd_block_q6_K *blk = (const lfamd_block_q6_K *)(raw + row * raw_row_bytes) + b;
            for (int ab = 0; ab < 2; ab++) {
                int dd = 2 * e + ab, t = 4 * g + dd;
                for (int j = 0; j < 8; j++)
                    v |= (uint32_t)(q6k_code(blk, 16 * t + 8 * h + j) >> 4) << qhbit(dd, j);
            }
        }
        dst[w] = v;
    } else if (w < 1664) {
        int s = w - 1536;
        int i = s >> 2, q = s & 3;
        long row = rt * 32 + i;
        uint32_t v = 0;
        if (row < rows) {
            const lfamd_block_q6_K *blk = (const lfamd_block_q6_K *)(raw + row * raw_row_bytes) + b;
            const uint8_t *sc = (const uint8_t *)blk->scales + 4 * q;
            v = sc[0] | (sc[1] << 8) | (sc[2] << 16) | ((uint32_t)sc[3] << 24);
        }
        dst[w] = v;
    } else {
        int s = w - 1664; // 16 dwords = 32 f16
        uint32_t v = 0;
        for (int e = 0; e < 2; e++) {
            long row = rt * 32 + 2 * s + e;
            if (row < rows) {
                const lfamd_block_q6_K *blk = (const lfamd_block_q6_K *)(raw + row * raw_row_bytes) + b;
                v |= (uint32_t)blk->d << (16 * e);
            }
        }
        dst[w] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Q8_0 -> P80

__global__ void pack_q80_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, int nblocks,
                                int nquads, uint8_t *__restrict__ out, long n_tiles) {
    long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // per tile: 256 qs dwords + 16 scale dwords = 272 dwords
    long tile = tid / 272;
    int w = (int)(tid % 272);
    if (tile >= n_tiles)
        return;
    long rg = tile / nquads;
    int L = (int)(tile % nquads);
    uint32_t *dst = (uint32_t *)(out + tile * P80_TILE);
    if (w < 256) {
        int r = w >> 5, j = (w >> 2) & 7, dd = w & 3;
        long row = rg * 8 + r;
        int blk = 4 * L + dd;
        uint32_t v = 0;
        if (row < rows && blk < nblocks) {
            const lfamd_block_q8_0 *bp = (const lfamd_block_q8_0 *)(raw + row * raw_row_bytes) + blk;
            const uint8_t *q = (const uint8_t *)bp->qs + 4 * j;
            v = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
        }
        dst[w] = v;
    } else {
        int s = w - 256; // 16 dwords = 32 f16 = d[r][dd]
        uint32_t v = 0;
        for (int e = 0; e < 2; e++) {
            int idx = 2 * s + e;
            int r = idx >> 2, dd = idx & 3;
            long row = rg * 8 + r;
            int blk = 4 * L + dd;
            if (row < rows && blk < nblocks) {
                const lfamd_block_q8_0 *bp = (const lfamd_block_q8_0 *)(raw + row * raw_row_bytes) + blk;
                v |= (uint32_t)bp->d << (16 * e);
            }
        }
        dst[w] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// RAW passthrough (row compaction): bytes copied row by row to stride = row_bytes.

__global__ void pack_raw_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, size_t row_bytes,
                                uint8_t *__restrict__ out) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)rows * row_bytes;
    for (; tid < total; tid += (size_t)gridDim.x * blockDim.x) {
        size_t r = tid / row_bytes, c = tid % row_bytes;
        out[tid] = raw[r * raw_row_bytes + c];
    }
}

// ---------------------------------------------------------------------------------------------
// Activation preparation for the MFMA GEMM: llamafile-order Q8_K rows ->
//   Xh  [nb][n_pad][256] f16  integer codes q8 (exact in f16), zero rows beyond n.  Super-block major: a token
//                             tile's codes of one super-block are ONE contiguous run (a [n_pad][k] matrix puts the
//                             tile's 512-byte row pieces a power-of-two stride apart -> they camp on 2 of the 16
//                             L2 channels and the GEMM's operand stream tops out near 10 TB/s)
//   d8T [nb][n_pad]     f32   block scales, transposed so a token tile's scales are contiguous
//   Xm  [nb][n_pad][16] f16   mins operand: for the 8 pair sums S_j = bsums[2j]+bsums[2j+1]
//                             (|S_j| <= 4096) the split S_j = 64*hi_j + lo_j, lo in [0,63]:
//                             elements 0..7 = lo_j, 8..15 = hi_j (both exact in f16)

// mins16 == 1: Xm holds the 16 bsums themselves (exact in f16, |sum| <= 2032) for the types with 16-wide sub-blocks (Q2_K)
// mins16 == 2: SCALED staging for the scaled-operand body (gemm_lw.hip, FAST): Xh = f16(d8 * code), Xm = f16(d8 * S_j) in
//              elements 0..7 and zeros in 8..15 — one f16 rounding per operand, no per-super-block scaling in the GEMM
__device__ static inline _Float16 sat_f16(float v) {
    return (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f);
}
__global__ void prep_q8k_kernel(const uint8_t *__restrict__ B, size_t b_row_bytes, long n, long n_pad, int nb,
                                _Float16 *__restrict__ Xh, float *__restrict__ d8T, _Float16 *__restrict__ Xm, int mins16,
                                const int32_t *__restrict__ src_idx) {
    long blk = blockIdx.x; // (token, super-block)
    long tok = blk / nb;
    int b = (int)(blk % nb);
    int t = threadIdx.x; // 64 threads: 4 codes each
    _Float16 *xo = Xh + ((size_t)b * n_pad + tok) * 256;
    // src_idx (MUL_MAT_ID batches): token slot -> activation row, -1 = padding slot
    const long src = src_idx ? (long)src_idx[tok] : (tok < n ? tok : -1);
    if (src >= 0) {
        const lfamd_block_q8_K *y = (const lfamd_block_q8_K *)(B + src * b_row_bytes) + b;
        uint32_t q = *(const uint32_t *)((const uint8_t *)y->qs + 4 * t); // 292-byte blocks are 4-aligned
        const float xs = mins16 == 2 ? y->d : 1.0f;
        for (int e = 0; e < 4; e++)
            xo[4 * t + e] = sat_f16((float)(int)(int8_t)(q >> (8 * e)) * xs);
        if (t == 0)
            d8T[(size_t)b * n_pad + tok] = y->d;
        if (mins16 == 2) {
            if (t < 8) {
                _Float16 *mo = Xm + ((size_t)b * n_pad + tok) * 16;
                mo[t] = sat_f16((float)((int)y->bsums[2 * t] + (int)y->bsums[2 * t + 1]) * xs);
                mo[8 + t] = (_Float16)0;
            }
        } else if (mins16) {
            if (t < 16)
                Xm[((size_t)b * n_pad + tok) * 16 + t] = (_Float16)(int)y->bsums[t];
        } else if (t < 8) {
            int S = (int)y->bsums[2 * t] + (int)y->bsums[2 * t + 1];
            int lo = S & 63, hi = (S - lo) / 64;
            _Float16 *mo = Xm + ((size_t)b * n_pad + tok) * 16;
            mo[t] = (_Float16)lo;
            mo[8 + t] = (_Float16)hi;
        }
    } else {
        for (int e = 0; e < 4; e++)
            xo[4 * t + e] = (_Float16)0;
        if (t == 0)
            d8T[(size_t)b * n_pad + tok] = 0.0f;
        if (t < 16)
            Xm[((size_t)b * n_pad + tok) * 16 + t] = (_Float16)0;
    }
}

// Same outputs straight from f32 activations: quantise exactly like quantize_row_q8_K (first index of
// the largest |x|, iscale = -128/max, nearest-even, clamp 127, d = 1/iscale) without materialising
// the Q8_K blocks.  One wave per (token, super-block), 4 values per lane.
__global__ __launch_bounds__(64) void prep_f32_kernel(const uint8_t *__restrict__ X, size_t x_row_bytes, long n, long n_pad,
                                                      int nb, _Float16 *__restrict__ Xh, float *__restrict__ d8T,
                                                      _Float16 *__restrict__ Xm, int mins16, const int32_t *__restrict__ src_idx) {
    long blk = blockIdx.x;
    long tok = blk / nb;
    int b = (int)(blk % nb);
    int t = threadIdx.x;
    _Float16 *xo = Xh + ((size_t)b * n_pad + tok) * 256;
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    const long src = src_idx ? (long)src_idx[tok] : (tok < n ? tok : -1); // as prep_q8k_kernel
    if (src < 0) {
        half4_t z = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
        *(half4_t *)(xo + 4 * t) = z;
        if (t == 0)
            d8T[(size_t)b * n_pad + tok] = 0.0f;
        if (t < 16)
            Xm[((size_t)b * n_pad + tok) * 16 + t] = (_Float16)0;
        return;
    }
    const float4 f = *(const float4 *)((const float *)(X + src * x_row_bytes) + (size_t)b * 256 + 4 * t);
    const float v[4] = {f.x, f.y, f.z, f.w};
    float amax = 0.0f, val = 0.0f;
    int idx = 4 * t;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float ax = fabsf(v[e]);
        if (ax > amax) {
            amax = ax;
            val = v[e];
            idx = 4 * t + e;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float oa = __shfl_xor(amax, off, 64);
        int oi = __shfl_xor(idx, off, 64);
        float ov = __shfl_xor(val, off, 64);
        if (oa > amax || (oa == amax && oi < idx)) {
            amax = oa;
            idx = oi;
            val = ov;
        }
    }
    int q[4] = {0, 0, 0, 0};
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -128.0f / val;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            int c = (int)rintf(iscale * v[e]);
            q[e] = c > 127 ? 127 : c;
        }
        d = 1.0f / iscale;
    }
    const float xs = mins16 == 2 ? d : 1.0f;
    half4_t h4 = {sat_f16((float)q[0] * xs), sat_f16((float)q[1] * xs), sat_f16((float)q[2] * xs), sat_f16((float)q[3] * xs)};
    *(half4_t *)(xo + 4 * t) = h4;
    int S = q[0] + q[1] + q[2] + q[3]; // pair sum j = t/8 covers codes 32j..32j+31 = lanes 8j..8j+7
    S += __shfl_xor(S, 1, 64);
    S += __shfl_xor(S, 2, 64);
    if (mins16 == 1 && (t & 3) == 0) // bsums[t/4]: codes 16(t/4) .. +15
        Xm[((size_t)b * n_pad + tok) * 16 + (t >> 2)] = (_Float16)S;
    S += __shfl_xor(S, 4, 64);
    if (mins16 == 2) {
        if ((t & 7) == 0) {
            _Float16 *mo = Xm + ((size_t)b * n_pad + tok) * 16;
            mo[t >> 3] = sat_f16((float)S * xs);
            mo[8 + (t >> 3)] = (_Float16)0;
        }
    } else if (!mins16 && (t & 7) == 0) {
        int j = t >> 3;
        int lo = S & 63, hi = (S - lo) / 64;
        _Float16 *mo = Xm + ((size_t)b * n_pad + tok) * 16;
        mo[j] = (_Float16)lo;
        mo[8 + j] = (_Float16)hi;
    }
    if (t == 0)
        d8T[(size_t)b * n_pad + tok] = d;
}

// SCALED staging with a per-token power-of-two normalisation (mode 2, the scaled-operand GEMM of gemm_lw.hip):
//   Xh = f16(d8 * code * 2^-e(token)),  Xm[0..7] = f16(d8 * S_j * 2^-e), Xm[8..15] = 0,  tok_scale[token] = 2^e
// with e chosen so that the token's largest |d8 * 128| lands in [512, 1024): no f16 overflow for huge activations, no
// subnormals for tiny ones, and — a power of two commutes with the rounding — the same bits as the unnormalised staging
// wherever that one is in range.  The GEMM multiplies its output column by tok_scale when it stores.
// One work-group per token: pass 1 finds the largest block scale (|d8| = amax / 128), pass 2 quantises exactly like
// prep_f32_kernel / prep_q8k_kernel (the second read of the row hits the caches).
// MAXJ > 0: the wave's super-blocks (b = wave + 16 j, j < MAXJ) are loaded ONCE, all loads in flight together, and both
// passes run from registers (one memory round trip per token; 16 waves per token keep as many waves in flight as the per-super-block kernels);
// MAXJ == 0: any nb, the row is read twice (the second time from the caches).
template <bool F32IN, int MAXJ, int NW>
__global__ __launch_bounds__(NW * 64) void prep_scaled_kernel(const uint8_t *__restrict__ X, size_t row_bytes, long n, long n_pad, int nb,
                                                          _Float16 *__restrict__ Xh, float *__restrict__ tok_scale,
                                                          _Float16 *__restrict__ Xm, const int32_t *__restrict__ src_idx) {
    __shared__ float wmax[NW]; // NW = 4 or 16 waves per token
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    constexpr int NJ = MAXJ > 0 ? MAXJ : 1;
    const long tok = blockIdx.x;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const long src = src_idx ? (long)src_idx[tok] : (tok < n ? tok : -1);
    if (src < 0) { // padding slot: zero codes, never stored
        const half4_t z = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
        for (int b = wave; b < nb; b += NW) {
            *(half4_t *)(Xh + ((size_t)b * n_pad + tok) * 256 + 4 * lane) = z;
            if (lane < 16)
                Xm[((size_t)b * n_pad + tok) * 16 + lane] = (_Float16)0;
        }
        if (t == 0)
            tok_scale[tok] = 0.0f;
        return;
    }
    const uint8_t *row = X + src * row_bytes;
    // what a lane holds of super-block b: four f32 values, or four codes and the block scale
    float4 fv[NJ];
    uint32_t qv[NJ];
    float dv[NJ];
    auto load = [&](int b, float4 &f, uint32_t &w, float &d) {
        if constexpr (F32IN) {
            f = *(const float4 *)((const float *)row + (size_t)b * 256 + 4 * lane);
        } else {
            const lfamd_block_q8_K *y = (const lfamd_block_q8_K *)row + b;
            w = *(const uint32_t *)((const uint8_t *)y->qs + 4 * lane);
            d = y->d;
        }
    };
    auto block_amax = [&](const float4 &f, float d) -> float { // |d8| * 128 (lane-local part for f32 input)
        if constexpr (F32IN)
            return fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w)));
        else
            return fabsf(d) * 128.0f;
    };
    float dmax = 0.0f;
    if constexpr (MAXJ > 0) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            fv[j] = make_float4(0.f, 0.f, 0.f, 0.f), qv[j] = 0, dv[j] = 0.0f;
            if (wave + NW * j < nb)
                load(wave + NW * j, fv[j], qv[j], dv[j]);
        }
#pragma unroll
        for (int j = 0; j < NJ; j++)
            dmax = fmaxf(dmax, block_amax(fv[j], dv[j]));
    } else {
        for (int b = wave; b < nb; b += NW) {
            load(b, fv[0], qv[0], dv[0]);
            dmax = fmaxf(dmax, block_amax(fv[0], dv[0]));
        }
    }
    // (DPP + readlane reductions: the ds_bpermute butterflies of the first version — about twenty LDS round trips per
    // thread — made this 12 MB conversion take 7.5 us)
    dmax = wave_max_f32(dmax);
    if (lane == 0)
        wmax[wave] = dmax;
    __syncthreads();
    dmax = wmax[lane & (NW - 1)]; // (every wave reduces the NW partial maxima itself)
    dmax = fmaxf(dmax, dpp_f32<DPP_XOR1>(dmax));
    dmax = fmaxf(dmax, dpp_f32<DPP_XOR2>(dmax));
    dmax = fmaxf(dmax, dpp_f32<DPP_HALF_MIRROR>(dmax));
    dmax = fmaxf(dmax, dpp_f32<DPP_MIRROR>(dmax));
    const bool ok = dmax > 0.0f && dmax < 3.0e38f; // (zero / non-finite rows: no normalisation)
    const float scale = ok ? ldexpf(1.0f, 9 - ilogbf(dmax)) : 1.0f;
    if (t == 0)
        tok_scale[tok] = ok ? ldexpf(1.0f, ilogbf(dmax) - 9) : 1.0f;

    auto emit = [&](int b, const float4 &f, uint32_t w, float dq) {
        _Float16 *xo = Xh + ((size_t)b * n_pad + tok) * 256;
        int q[4] = {0, 0, 0, 0};
        float d = 0.0f;
        if constexpr (F32IN) { // quantize_row_q8_K: first index of the largest |x|, iscale = -128 / max, nearest-even, clamp 127
            const float v[4] = {f.x, f.y, f.z, f.w};
            float amax = 0.0f, val = 0.0f;
            int idx = 4 * lane;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float ax = fabsf(v[e]);
                if (ax > amax)
                    amax = ax, val = v[e], idx = 4 * lane + e;
            }
            // the block's largest |x| (one value through the butterfly), then the FIRST lane holding it: lanes are in index
            // order and `val` is already the lane's first such element, so this is quantize_row_q8_K's tie-break
            const float bmax = wave_max_f32(amax);
            const unsigned long long holders = __builtin_amdgcn_ballot_w64(amax == bmax);
            val = readlane_f32(val, holders ? __builtin_ctzll(holders) : 0);
            amax = bmax;
            (void)idx;
            // (branch-free: an all-zero block gives iscale = 0 -> codes 0, d = 0, like the reference's early return; with a
            // branch per block the four blocks of a wave cannot be scheduled into each other)
            const bool nz = amax != 0.0f;
            const float iscale = nz ? -128.0f / val : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = (int)rintf(iscale * v[e]);
                q[e] = c > 127 ? 127 : c;
            }
            d = nz ? 1.0f / iscale : 0.0f;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                q[e] = (int)(int8_t)(w >> (8 * e));
            d = dq;
        }
        const float xs = d * scale;
        const half4_t h4 = {sat_f16((float)q[0] * xs), sat_f16((float)q[1] * xs), sat_f16((float)q[2] * xs), sat_f16((float)q[3] * xs)};
        *(half4_t *)(xo + 4 * lane) = h4;
        int S = q[0] + q[1] + q[2] + q[3]; // pair sum j = lane / 8 covers codes 32j .. 32j+31
        S += (int)dpp_u32<DPP_XOR1>((uint32_t)S);
        S += (int)dpp_u32<DPP_XOR2>((uint32_t)S);
        S += (int)dpp_u32<DPP_HALF_MIRROR>((uint32_t)S);
        if ((lane & 7) == 0) {
            _Float16 *mo = Xm + ((size_t)b * n_pad + tok) * 16;
            mo[lane >> 3] = sat_f16((float)S * xs);
            mo[8 + (lane >> 3)] = (_Float16)0;
        }
    };
    if constexpr (MAXJ > 0) {
        if (nb == NW * NJ) { // every wave has all NJ blocks: straight-line code, the blocks' dependent chains interleave
#pragma unroll
            for (int j = 0; j < NJ; j++)
                emit(wave + NW * j, fv[j], qv[j], dv[j]);
        } else {
#pragma unroll
            for (int j = 0; j < NJ; j++)
                if (wave + NW * j < nb) // (wave-uniform)
                    emit(wave + NW * j, fv[j], qv[j], dv[j]);
        }
    } else {
        for (int b = wave; b < nb; b += NW) {
            load(b, fv[0], qv[0], dv[0]);
            emit(b, fv[0], qv[0], dv[0]);
        }
    }
}

// Activation preparation for the legacy 32-block weight types (Q4_0 ...): Q8_0 quantisation (upstream
// quantize_row_q8_0: d = amax/127 stored as f16, q = roundf(x/d)) ->
//   Xh  [nb][n_pad][256] f16 codes (as above);  d8T [nb*8][n_pad] f32 block scales.  One wave per (super-block, token).
// Q81: Q8_1 activations (Q4_1 / Q5_1 weights): additionally sT [nb*8][n_pad] = the block's s = f16(d * sum(q)).
template <bool F32IN, bool Q81>
__global__ __launch_bounds__(256) void prep80_kernel(const uint8_t *__restrict__ X, size_t x_row_bytes, long n, long n_pad, int nb,
                                                    _Float16 *__restrict__ Xh, float *__restrict__ d8T, float *__restrict__ sT) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= (long)nb * n_pad)
        return;
    int b = (int)(blk / n_pad);
    long tok = blk - (long)b * n_pad;
    int t = threadIdx.x & 63; // codes 4t..4t+3 of the super-block: 32-block t >> 3
    int q[4] = {0, 0, 0, 0};
    float d = 0.0f, sv = 0.0f;
    if (tok < n) {
        if constexpr (F32IN) {
            const float4 f = *(const float4 *)((const float *)(X + tok * x_row_bytes) + (size_t)b * 256 + 4 * t);
            const float v[4] = {f.x, f.y, f.z, f.w};
            float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
            const float dd = amax / 127.0f;
            const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; e++)
                q[e] = (int)roundf(v[e] * id);
            d = h2f(f2h_bits(dd));
            if constexpr (Q81) { // upstream quantize_row_q8_1: s = f16(sum * d), d not yet rounded
                int sum = q[0] + q[1] + q[2] + q[3];
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                sum += __shfl_xor(sum, 4, 64);
                sv = h2f(f2h_bits((float)sum * dd));
            }
        } else {
            constexpr int BSZ = Q81 ? 36 : 34, QOFF = Q81 ? 4 : 2;
            const uint8_t *y = X + tok * x_row_bytes + (size_t)(b * 8 + (t >> 3)) * BSZ; // 2-byte aligned blocks
            const uint16_t *p = (const uint16_t *)(y + QOFF + 4 * (t & 7));
            if constexpr (Q81)
                sv = h2f(*(const uint16_t *)(y + 2));
            const uint32_t w = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
#pragma unroll
            for (int e = 0; e < 4; e++)
                q[e] = (int)(int8_t)(w >> (8 * e));
            d = h2f(*(const uint16_t *)y);
        }
    }
    half4_t h4 = {(_Float16)q[0], (_Float16)q[1], (_Float16)q[2], (_Float16)q[3]};
    *(half4_t *)(Xh + ((size_t)b * n_pad + tok) * 256 + 4 * t) = h4;
    if ((t & 7) == 0) {
        d8T[((size_t)b * 8 + (t >> 3)) * n_pad + tok] = d;
        if constexpr (Q81)
            sT[((size_t)b * 8 + (t >> 3)) * n_pad + tok] = sv;
    }
}

// Activation preparation for the float weight types (F16 / BF16): Xh [nb][n_pad][256] of 2-byte values — f32 rows
// converted like ggml does before calling sgemm (f16: round to nearest even; bf16: ggml_compute_fp32_to_bf16, nearest even
// with NaN quieting) or rows already in the weight's type copied.  One wave per (super-block, token).
template <int OUT, bool F32IN>
__global__ __launch_bounds__(256) void prep_float_kernel(const uint8_t *__restrict__ X, size_t x_row_bytes, long n, long n_pad,
                                                        int nb, uint16_t *__restrict__ Xh) {
    long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= (long)nb * n_pad)
        return;
    int b = (int)(blk / n_pad);
    long tok = blk - (long)b * n_pad;
    int t = threadIdx.x & 63;
    uint16_t o[4] = {0, 0, 0, 0};
    if (tok < n) {
        if constexpr (F32IN) {
            const float4 f = *(const float4 *)((const float *)(X + tok * x_row_bytes) + (size_t)b * 256 + 4 * t);
            const float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if constexpr (OUT == LFAMD_TYPE_F16) {
                    o[e] = f2h_bits(v[e]);
                } else {
                    const uint32_t u = __builtin_bit_cast(uint32_t, v[e]);
                    o[e] = (u & 0x7fffffffu) > 0x7f800000u ? (uint16_t)((u >> 16) | 64) : (uint16_t)((u + (0x7fffu + ((u >> 16) & 1))) >> 16);
                }
            }
        } else {
            const uint2 w = *(const uint2 *)(X + tok * x_row_bytes + ((size_t)b * 256 + 4 * t) * 2);
            o[0] = (uint16_t)w.x, o[1] = (uint16_t)(w.x >> 16), o[2] = (uint16_t)w.y, o[3] = (uint16_t)(w.y >> 16);
        }
    }
    *(uint2 *)(Xh + ((size_t)b * n_pad + tok) * 256 + 4 * t) =
        make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers (used by api.hip)

extern "C" {

hipError_t lfamd_launch_pack_q4k(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s) {
    int nb = (int)(cols / 256);
    long n_tiles = ((rows + 31) / 32) * nb;
    long threads = n_tiles * 1152;
    pack_q4k_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nb,
                                                                        (uint8_t *)out, n_tiles);
    return hipGetLastError();
}

hipError_t lfamd_launch_pack_q40(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s) {
    int nb = (int)(cols / 256);
    long n_tiles = ((rows + 31) / 32) * nb;
    long threads = n_tiles * 1152;
    pack_q40_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nb,
                                                                        (uint8_t *)out, n_tiles);
    return hipGetLastError();
}

hipError_t lfamd_launch_pack_q5k(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s) {
    int nb = (int)(cols / 256);
    long n_tiles = ((rows + 31) / 32) * nb;
    long threads = n_tiles * 1408;
    pack_q5k_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nb,
                                                                        (uint8_t *)out, n_tiles);
    return hipGetLastError();
}

hipError_t lfamd_launch_pack_q6k(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s) {
    int nb = (int)(cols / 256);
    long n_tiles = ((rows + 31) / 32) * nb;
    long threads = n_tiles * 1680;
    pack_q6k_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nb,
                                                                        (uint8_t *)out, n_tiles);
    return hipGetLastError();
}

hipError_t lfamd_launch_pack_q80(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s) {
    int nblocks = (int)(cols / 32);
    int nquads = (nblocks + 3) / 4;
    long n_tiles = ((rows + 7) / 8) * nquads;
    long threads = n_tiles * 272;
    pack_q80_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, nblocks,
                                                                        nquads, (uint8_t *)out, n_tiles);
    return hipGetLastError();
}

hipError_t lfamd_launch_pack_raw(const void *raw, size_t raw_row_bytes, long rows, size_t row_bytes, void *out,
                                 hipStream_t s) {
    size_t total = (size_t)rows * row_bytes;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65536)
        blocks = 65536;
    if (blocks == 0)
        return hipSuccess;
    pack_raw_kernel<<<(unsigned)blocks, 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, row_bytes, (uint8_t *)out);
    return hipGetLastError();
}

// the register-resident forms up to 16 / 64 super-blocks (k <= 4096 / 16384: 1 / 4 per wave), the looping one beyond
// register-resident forms: 4 waves per token with 4 / 16 super-blocks each (k <= 4096 / 16384) — a quarter of the waves of
// the 16-wave form to launch and to meet at the barrier, four loads in flight per lane instead of one; the looping form beyond
#define PREP_SCALED_GO(F32IN, SRC, RB)                                                                                 \
    do {                                                                                                               \
        if (nb <= 16)                                                                                                  \
            prep_scaled_kernel<F32IN, 4, 4><<<(unsigned)n_pad, 256, 0, s>>>((const uint8_t *)SRC, RB, n, n_pad, nb, (_Float16 *)Xh, \
                                                                           (float *)d8T, (_Float16 *)Xm, src_idx);      \
        else if (nb <= 64)                                                                                             \
            prep_scaled_kernel<F32IN, 16, 4><<<(unsigned)n_pad, 256, 0, s>>>((const uint8_t *)SRC, RB, n, n_pad, nb, (_Float16 *)Xh, \
                                                                            (float *)d8T, (_Float16 *)Xm, src_idx);     \
        else                                                                                                           \
            prep_scaled_kernel<F32IN, 0, 16><<<(unsigned)n_pad, 1024, 0, s>>>((const uint8_t *)SRC, RB, n, n_pad, nb, (_Float16 *)Xh, \
                                                                             (float *)d8T, (_Float16 *)Xm, src_idx);    \
    } while (0)

hipError_t lfamd_launch_prep_f32(const void *X, size_t x_row_bytes, long n, long n_pad, long cols, void *Xh, void *d8T,
                                 void *Xm, int mins16, const int32_t *src_idx, hipStream_t s) {
    int nb = (int)(cols / 256);
    long blocks = n_pad * nb;
    if (blocks == 0)
        return hipSuccess;
    if (mins16 == 2) // scaled staging: d8T receives the per-token output scales [n_pad]
        PREP_SCALED_GO(true, X, x_row_bytes);
    else
        prep_f32_kernel<<<(unsigned)blocks, 64, 0, s>>>((const uint8_t *)X, x_row_bytes, n, n_pad, nb, (_Float16 *)Xh,
                                                         (float *)d8T, (_Float16 *)Xm, mins16, src_idx);
    return hipGetLastError();
}

hipError_t lfamd_launch_prep80(int Btype, const void *B, size_t b_row_bytes, long n, long n_pad, long cols, void *Xh, void *d8T,
                               void *sT, hipStream_t s) {
    int nb = (int)(cols / 256);
    long blocks = n_pad * nb;
    if (blocks == 0)
        return hipSuccess;
    const unsigned grid = (unsigned)((blocks + 3) / 4);
    const uint8_t *X = (const uint8_t *)B;
    if (sT) { // Q8_1 activations
        if (Btype == LFAMD_TYPE_F32)
            prep80_kernel<true, true><<<grid, 256, 0, s>>>(X, b_row_bytes, n, n_pad, nb, (_Float16 *)Xh, (float *)d8T, (float *)sT);
        else
            prep80_kernel<false, true><<<grid, 256, 0, s>>>(X, b_row_bytes, n, n_pad, nb, (_Float16 *)Xh, (float *)d8T, (float *)sT);
    } else {
        if (Btype == LFAMD_TYPE_F32)
            prep80_kernel<true, false><<<grid, 256, 0, s>>>(X, b_row_bytes, n, n_pad, nb, (_Float16 *)Xh, (float *)d8T, nullptr);
        else
            prep80_kernel<false, false><<<grid, 256, 0, s>>>(X, b_row_bytes, n, n_pad, nb, (_Float16 *)Xh, (float *)d8T, nullptr);
    }
    return hipGetLastError();
}

hipError_t lfamd_launch_prep_float(int Atype, int Btype, const void *B, size_t b_row_bytes, long n, long n_pad, long cols, void *Xh,
                                   hipStream_t s) {
    int nb = (int)(cols / 256);
    long blocks = n_pad * nb;
    if (blocks == 0)
        return hipSuccess;
    const unsigned grid = (unsigned)((blocks + 3) / 4);
    const uint8_t *X = (const uint8_t *)B;
    if (Atype == LFAMD_TYPE_F16) {
        if (Btype == LFAMD_TYPE_F32)
            prep_float_kernel<LFAMD_TYPE_F16, true><<<grid, 256, 0, s>>>(X, b_row_bytes, n, n_pad, nb, (uint16_t *)Xh);
        else
            prep_float_kernel<LFAMD_TYPE_F16, false><<<grid, 256, 0, s>>>(X, b_row_bytes, n, n_pad, nb, (uint16_t *)Xh);
    } else {
        if (Btype == LFAMD_TYPE_F32)
            prep_float_kernel<LFAMD_TYPE_BF16, true><<<grid, 256, 0, s>>>(X, b_row_bytes, n, n_pad, nb, (uint16_t *)Xh);
        else
            prep_float_kernel<LFAMD_TYPE_BF16, false><<<grid, 256, 0, s>>>(X, b_row_bytes, n, n_pad, nb, (uint16_t *)Xh);
    }
    return hipGetLastError();
}

hipError_t lfamd_launch_prep_q8k(const void *B, size_t b_row_bytes, long n, long n_pad, long cols, void *Xh, void *d8T,
                                 void *Xm, int mins16, const int32_t *src_idx, hipStream_t s) {
    int nb = (int)(cols / 256);
    long blocks = n_pad * nb;
    if (blocks == 0)
        return hipSuccess;
    if (mins16 == 2)
        PREP_SCALED_GO(false, B, b_row_bytes);
    else
        prep_q8k_kernel<<<(unsigned)blocks, 64, 0, s>>>((const uint8_t *)B, b_row_bytes, n, n_pad, nb, (_Float16 *)Xh,
                                                         (float *)d8T, (_Float16 *)Xm, mins16, src_idx);
    return hipGetLastError();
}
}

// ---------------------------------------------------------------------------------------------
// Range check for the scaled-operand GEMM (gemm_lw.hip FAST): every row header of a P4K / P5K / P6K image.

__global__ void scaled_ok_kernel(const uint8_t *__restrict__ img, long tiles, int tile_bytes, int q6, int *__restrict__ bad) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; // (tile, row in tile)
    if (idx >= tiles * 32)
        return;
    union {
        uint16_t u;
        _Float16 h;
    } d, dm;
    const uint8_t *tile = img + (size_t)(idx >> 5) * tile_bytes;
    if (q6) { // f16(d * sc) * (code - 32): |d| * 127 * 32 must stay inside f16
        d.u = *(const uint16_t *)(tile + P6K_D + (idx & 31) * 2);
        if (!(fabsf((float)d.h) * (127.0f * 32.0f) <= 65504.0f))
            atomicOr(bad, 1);
        return;
    }
    const uint32_t dd = *(const uint32_t *)(tile + P4K_HDR + (idx & 31) * 16);
    d.u = (uint16_t)(dd & 0xffff), dm.u = (uint16_t)(dd >> 16);
    const float fd = fabsf((float)d.h), fm = fabsf((float)dm.h);
    if (!(fd * 63.0f < 64.0f) || !(fm * 63.0f <= 65504.0f)) // also catches NaN / inf
        atomicOr(bad, 1);
}

extern "C" hipError_t lfamd_launch_scaled_ok(int type, long rows, long cols, const void *packed, int *d_flag, hipStream_t s) {
    const long tiles = ((rows + 31) / 32) * (cols / 256);
    const int tile_bytes = type == LFAMD_TYPE_Q5_K ? P5K_TILE : type == LFAMD_TYPE_Q6_K ? P6K_TILE : P4K_TILE;
    const long threads = tiles * 32;
    scaled_ok_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)packed, tiles, tile_bytes,
                                                                       type == LFAMD_TYPE_Q6_K ? 1 : 0, d_flag);
    return hipGetLastError();
}
