// norm_quant.hip — the steps in front of the path, fused: RMS-norm x weight, and silu(gate) * up, ->
// the reference's Q8_K activation blocks in ONE kernel each, so that the mat-muls behind a norm (attn_q/k/v, ffn_gate/up, output) receive
// Btype = Q8_K and their prologue is a copy instead of a quantisation.
//
// Reference counterparts: ggml_compute_forward_rms_norm_f32 + the MUL by the norm weight (upstream ggml.c; GPU:
// rms_norm_f32, ggml-cuda.cu.patch:14926-14960) followed by quantize_row_q8_K inside ggml_compute_forward_mul_mat
// (llamafile order {d, bsums, qs}: ggml-common.h.patch:25-35).  Arithmetic restated:
//     sum  = sum_i (double)(x[i] * x[i])          (f32 products, f64 accumulation)
//     mean = (float)(sum / k);  scale = 1.0f / sqrtf(mean + eps)
//     y[i] = (x[i] * scale) * w[i]                (two f32 roundings: the norm, then the MUL node)
//     Q8_K per 256 values: first index of the largest |y| -> iscale = -128 / max, nearest-even codes clamped at 127,
//     d = 1 / iscale, bsums of 16
// The f64 sum is a tree here and a sequential loop on the CPU: they differ only below 1e-15 relative, i.e. the f32 `mean`
// (and everything after it) is identical unless the sum sits within 1e-9 relative of a rounding boundary.
#include "lfamd_device.h"
#include "../../include/lfamd_hip.h"

extern "C" void lfamd_set_error(const char *msg);

namespace {

__device__ static inline double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// quantize_row_q8_K on one 256-block held as four values per lane (element 4 lane + e), written in the reference's block
// format {d, bsums[16], qs[256]} (cf. gemv_impl.h stage_f32_q8k_wave: same arithmetic)
__device__ static inline void put_q8k_block(uint8_t *blk, const float (&y)[4], int lane) {
    const float a0 = fabsf(y[0]), a1 = fabsf(y[1]), a2 = fabsf(y[2]), a3 = fabsf(y[3]);
    const float amax = wave_max_f32(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    const bool m0 = a0 == amax, m1 = a1 == amax, m2 = a2 == amax, m3 = a3 == amax;
    const unsigned long long ball = __builtin_amdgcn_ballot_w64(m0 || m1 || m2 || m3);
    const float cand = m0 ? y[0] : (m1 ? y[1] : (m2 ? y[2] : y[3]));
    const bool nz = amax != 0.0f;
    const float val = nz ? readlane_f32(cand, ball ? __builtin_ctzll(ball) : 0) : 1.0f;
    const float iscale = -128.0f / val;
    int q[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int c = (int)rintf(iscale * y[e]);
        q[e] = nz ? (c > 127 ? 127 : c) : 0;
    }
    *(uint32_t *)(blk + 36 + 4 * lane) = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) |
                                        ((uint32_t)(q[3] & 0xff) << 24);
    int bs = q[0] + q[1] + q[2] + q[3];
    bs += (int)dpp_u32<DPP_XOR1>((uint32_t)bs);
    bs += (int)dpp_u32<DPP_XOR2>((uint32_t)bs);
    if ((lane & 3) == 0)
        *(int16_t *)(blk + 4 + 2 * (lane >> 2)) = (int16_t)bs;
    if (lane == 0)
        *(float *)blk = nz ? 1.0f / iscale : 0.0f;
}

// The same block written straight into the staged image of the int8 batch body (csrc/gemm_i8.hip: Xq [nb][n_pad][256] int8 in the
// byte order of its unpacked nibbles, d8T [nb][n_pad] f32, Xs [nb][n_pad][16] f16 bsums): the codes, scale and sums are those of
// put_q8k_block bit for bit — only where they go differs, so the mat-mul behind gives the bits it gives on the f32 row.
// Lane l holds codes 4 l .. 4 l + 3; group g = l >> 1 of eight codes c0..c7 is stored as (c0,c4,c1,c5 | c2,c6,c3,c7) at
// (g >> 2) * 32 + (g & 1) * 16 + ((g >> 1) & 1) * 8 of the token's 256 bytes (prep_i8_kernel).
struct staged_image {
    int8_t *Xq;
    float *d8T;
    _Float16 *Xs;
    long n_pad;
};
__device__ static inline void put_staged_block(const staged_image &im, int b, long tok, const float (&y)[4], int lane) {
    const float a0 = fabsf(y[0]), a1 = fabsf(y[1]), a2 = fabsf(y[2]), a3 = fabsf(y[3]);
    const float amax = wave_max_f32(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    const bool m0 = a0 == amax, m1 = a1 == amax, m2 = a2 == amax, m3 = a3 == amax;
    const unsigned long long ball = __builtin_amdgcn_ballot_w64(m0 || m1 || m2 || m3);
    const float cand = m0 ? y[0] : (m1 ? y[1] : (m2 ? y[2] : y[3]));
    const bool nz = amax != 0.0f;
    const float val = nz ? readlane_f32(cand, ball ? __builtin_ctzll(ball) : 0) : 1.0f;
    const float iscale = -128.0f / val;
    int q[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int c = (int)rintf(iscale * y[e]);
        q[e] = nz ? (c > 127 ? 127 : c) : 0;
    }
    const uint32_t mine = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
    const uint32_t other = dpp_u32<DPP_XOR1>(mine);
    const size_t o = (size_t)b * im.n_pad + tok;
    if ((lane & 1) == 0) { // mine = c0..c3, other = c4..c7
        const int g = lane >> 1;
        const uint32_t w0 = __builtin_amdgcn_perm(other, mine, 0x05010400u); // (c0, c4, c1, c5)
        const uint32_t w1 = __builtin_amdgcn_perm(other, mine, 0x07030602u); // (c2, c6, c3, c7)
        *(uint2 *)(im.Xq + o * 256 + (g >> 2) * 32 + (g & 1) * 16 + ((g >> 1) & 1) * 8) = make_uint2(w0, w1);
    }
    int bs = q[0] + q[1] + q[2] + q[3];
    bs += (int)dpp_u32<DPP_XOR1>((uint32_t)bs);
    bs += (int)dpp_u32<DPP_XOR2>((uint32_t)bs);
    if ((lane & 3) == 0)
        im.Xs[o * 16 + (lane >> 2)] = (_Float16)(float)bs;
    if (lane == 0)
        im.d8T[o] = nz ? 1.0f / iscale : 0.0f;
}
__device__ static inline void put_staged_zero(const staged_image &im, int b, long tok, int lane) { // a padding token of the image
    const size_t o = (size_t)b * im.n_pad + tok;
    *(uint32_t *)(im.Xq + o * 256 + 4 * lane) = 0u;
    if (lane < 8)
        *(uint32_t *)(im.Xs + o * 16 + 2 * lane) = 0u;
    if (lane == 0)
        im.d8T[o] = 0.0f;
}

// The staged image of the scaled-operand f16 batch bodies (gemm_lw / gemm_ks / gemm_kr; what prep_scaled_kernel of pack.hip writes):
//   Xh [nb][n_pad][256] f16 = q8 * d8 * 2^-e(token), d8T [n_pad] f32 = 2^e(token) (the store multiplies the column by it),
//   Xm [nb][n_pad][16] f16 = the eight 32-code sums times d8 * 2^-e (the mins operand), then eight zeros;
// e(token) from the largest |y| of the whole row, so that every operand sits in f16's normal range: that maximum must be known
// before the first code is emitted — two passes over the row inside the producer.  Same arithmetic as prep_scaled_kernel, so the
// mat-mul behind gives the bits it gives on the f32 row.
struct scaled_image {
    _Float16 *Xh;
    float *tok_scale;
    _Float16 *Xm;
    long n_pad;
};
__device__ static inline _Float16 sat_f16_(float v) {
    return (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f);
}
__device__ static inline void put_scaled_block(const scaled_image &im, int b, long tok, const float (&y)[4], float scale, int lane) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    float amax = 0.0f, val = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float ax = fabsf(y[e]);
        if (ax > amax)
            amax = ax, val = y[e];
    }
    const float bmax = wave_max_f32(amax);
    const unsigned long long holders = __builtin_amdgcn_ballot_w64(amax == bmax);
    val = readlane_f32(val, holders ? __builtin_ctzll(holders) : 0);
    const bool nz = bmax != 0.0f;
    const float iscale = nz ? -128.0f / val : 0.0f;
    int q[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int c = (int)rintf(iscale * y[e]);
        q[e] = c > 127 ? 127 : c;
    }
    const float d = nz ? 1.0f / iscale : 0.0f;
    const float xs = d * scale;
    const half4_t h4 = {sat_f16_((float)q[0] * xs), sat_f16_((float)q[1] * xs), sat_f16_((float)q[2] * xs), sat_f16_((float)q[3] * xs)};
    const size_t o = (size_t)b * im.n_pad + tok;
    *(half4_t *)(im.Xh + o * 256 + 4 * lane) = h4;
    int S = q[0] + q[1] + q[2] + q[3]; // sum j = lane / 8 covers codes 32 j .. 32 j + 31
    S += (int)dpp_u32<DPP_XOR1>((uint32_t)S);
    S += (int)dpp_u32<DPP_XOR2>((uint32_t)S);
    S += (int)dpp_u32<DPP_HALF_MIRROR>((uint32_t)S);
    if ((lane & 7) == 0) {
        _Float16 *mo = im.Xm + o * 16;
        mo[lane >> 3] = sat_f16_((float)S * xs);
        mo[8 + (lane >> 3)] = (_Float16)0;
    }
}
__device__ static inline void put_scaled_zero(const scaled_image &im, int b, long tok, int lane) {
    const size_t o = (size_t)b * im.n_pad + tok;
    *(uint2 *)(im.Xh + o * 256 + 4 * lane) = make_uint2(0u, 0u);
    if (lane < 8)
        *(uint32_t *)(im.Xm + o * 16 + 2 * lane) = 0u;
}
// the row's power-of-two normalisation from the largest |y| each wave saw (every wave reduces the partial maxima itself)
template <int NWV>
__device__ static inline float row_scale(float dmax, float *wmax, int wave, int lane, float *tok_scale_out) {
    dmax = wave_max_f32(dmax);
    if (lane == 0)
        wmax[wave] = dmax;
    __syncthreads();
    float m = wmax[lane & (NWV - 1)];
    m = fmaxf(m, dpp_f32<DPP_XOR1>(m));
    m = fmaxf(m, dpp_f32<DPP_XOR2>(m));
    const bool ok = m > 0.0f && m < 3.0e38f; // (zero / non-finite rows: no normalisation)
    if (tok_scale_out && wave == 0 && lane == 0)
        *tok_scale_out = ok ? ldexpf(1.0f, ilogbf(m) - 9) : 1.0f;
    return ok ? ldexpf(1.0f, 9 - ilogbf(m)) : 1.0f;
}

// one work-group (4 waves) per row; wave w owns the 256-blocks w, w + 4, ...
template <bool STAGED>
__global__ __launch_bounds__(256) void rms_norm_q8k_kernel(const float *__restrict__ x, size_t x_row_bytes, const float *__restrict__ w,
                                                           float eps, long k, uint8_t *__restrict__ yq, size_t yq_row_bytes,
                                                           float *__restrict__ yf, size_t yf_row_bytes, long nrows, staged_image im) {
    __shared__ double part[4];
    const long row = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (STAGED && row >= nrows) { // (uniform) the image's padding tokens
        for (int b = wave; b < (int)(k / 256); b += 4)
            put_staged_zero(im, b, row, lane);
        return;
    }
    const float *xr = (const float *)((const uint8_t *)x + row * x_row_bytes);
    const int nb = (int)(k / 256);
    double s = 0.0;
    for (int b = wave; b < nb; b += 4) {
        const float4 v = *(const float4 *)(xr + (size_t)b * 256 + 4 * lane);
        s += (double)(v.x * v.x) + (double)(v.y * v.y) + (double)(v.z * v.z) + (double)(v.w * v.w);
    }
    s = wave_sum_f64(s);
    if (lane == 0)
        part[wave] = s;
    __syncthreads();
    const double sum = (part[0] + part[1]) + (part[2] + part[3]);
    const float mean = (float)(sum / (double)k);
    const float scale = 1.0f / sqrtf(mean + eps);
    uint8_t *qrow = yq ? yq + row * yq_row_bytes : nullptr;
    float *frow = yf ? (float *)((uint8_t *)yf + row * yf_row_bytes) : nullptr;
    for (int b = wave; b < nb; b += 4) {
        const float4 v = *(const float4 *)(xr + (size_t)b * 256 + 4 * lane); // (second read: L1 / L2)
        const float4 g = w ? *(const float4 *)(w + (size_t)b * 256 + 4 * lane) : make_float4(1.f, 1.f, 1.f, 1.f);
        float y[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
        if (w)
            y[0] *= g.x, y[1] *= g.y, y[2] *= g.z, y[3] *= g.w;
        if (frow)
            *(float4 *)(frow + (size_t)b * 256 + 4 * lane) = make_float4(y[0], y[1], y[2], y[3]);
        if constexpr (STAGED) {
            put_staged_block(im, b, row, y, lane);
        } else {
            if (!qrow)
                continue;
            put_q8k_block(qrow + (size_t)b * 292, y, lane);
        }
    }
}

// SwiGLU in front of ffn_down: y = silu(gate) * up with silu(x) = x / (1 + expf(-x)) (silu_f32, ggml-cuda.cu.patch:16172-16179;
// ggml_silu_f32, ggml-vector.inc:1662-1664) and the MUL node, then Q8_K.  One wave per 256-block.
template <bool STAGED>
__global__ __launch_bounds__(256) void swiglu_q8k_kernel(const float *__restrict__ gate, size_t gate_row_bytes, const float *__restrict__ up,
                                                         size_t up_row_bytes, long k, uint8_t *__restrict__ yq, size_t yq_row_bytes,
                                                         float *__restrict__ yf, size_t yf_row_bytes, long nrows, staged_image im) {
    const long row = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= (int)(k / 256))
        return;
    if (STAGED && row >= nrows) { // (uniform) the image's padding tokens
        put_staged_zero(im, b, row, lane);
        return;
    }
    const float4 g = *(const float4 *)((const float *)((const uint8_t *)gate + row * gate_row_bytes) + (size_t)b * 256 + 4 * lane);
    const float4 u = *(const float4 *)((const float *)((const uint8_t *)up + row * up_row_bytes) + (size_t)b * 256 + 4 * lane);
    const float gv[4] = {g.x, g.y, g.z, g.w}, uv[4] = {u.x, u.y, u.z, u.w};
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
        y[e] = (gv[e] / (1.0f + expf(-gv[e]))) * uv[e];
    if (yf)
        *(float4 *)((float *)((uint8_t *)yf + row * yf_row_bytes) + (size_t)b * 256 + 4 * lane) = make_float4(y[0], y[1], y[2], y[3]);
    if constexpr (STAGED)
        put_staged_block(im, b, row, y, lane);
    else if (yq)
        put_q8k_block(yq + row * yq_row_bytes + (size_t)b * 292, y, lane);
}

// RMS-norm x weight -> the scaled image: pass 1 the sum of squares, pass 2 the largest |y| of the row, pass 3 the blocks (the row is
// re-read from L1 / L2; one work-group of 4 waves per row like rms_norm_q8k_kernel, the same y bit for bit)
__global__ __launch_bounds__(256) void rms_norm_scaled_kernel(const float *__restrict__ x, size_t x_row_bytes, const float *__restrict__ w,
                                                              float eps, long k, float *__restrict__ yf, size_t yf_row_bytes, long nrows,
                                                              scaled_image im) {
    __shared__ double part[4];
    __shared__ float wmax[4];
    const long row = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = (int)(k / 256);
    if (row >= nrows) { // (uniform) the image's padding tokens
        for (int b = wave; b < nb; b += 4)
            put_scaled_zero(im, b, row, lane);
        if (threadIdx.x == 0)
            im.tok_scale[row] = 0.0f;
        return;
    }
    const float *xr = (const float *)((const uint8_t *)x + row * x_row_bytes);
    double s = 0.0;
    for (int b = wave; b < nb; b += 4) {
        const float4 v = *(const float4 *)(xr + (size_t)b * 256 + 4 * lane);
        s += (double)(v.x * v.x) + (double)(v.y * v.y) + (double)(v.z * v.z) + (double)(v.w * v.w);
    }
    s = wave_sum_f64(s);
    if (lane == 0)
        part[wave] = s;
    __syncthreads();
    const double sum = (part[0] + part[1]) + (part[2] + part[3]);
    const float mean = (float)(sum / (double)k);
    const float scale = 1.0f / sqrtf(mean + eps);
    auto y_of = [&](int b, float (&y)[4]) {
        const float4 v = *(const float4 *)(xr + (size_t)b * 256 + 4 * lane);
        y[0] = v.x * scale, y[1] = v.y * scale, y[2] = v.z * scale, y[3] = v.w * scale;
        if (w) {
            const float4 g = *(const float4 *)(w + (size_t)b * 256 + 4 * lane);
            y[0] *= g.x, y[1] *= g.y, y[2] *= g.z, y[3] *= g.w;
        }
    };
    float dmax = 0.0f;
    for (int b = wave; b < nb; b += 4) {
        float y[4];
        y_of(b, y);
        dmax = fmaxf(dmax, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
    }
    const float nscale = row_scale<4>(dmax, wmax, wave, lane, im.tok_scale + row);
    float *frow = yf ? (float *)((uint8_t *)yf + row * yf_row_bytes) : nullptr;
    for (int b = wave; b < nb; b += 4) {
        float y[4];
        y_of(b, y);
        if (frow)
            *(float4 *)(frow + (size_t)b * 256 + 4 * lane) = make_float4(y[0], y[1], y[2], y[3]);
        put_scaled_block(im, b, row, y, nscale, lane);
    }
}

// silu(gate) * up -> the scaled image: one work-group of 4 waves per row; pass 1 the largest |y| (the products are kept in LDS when
// the row fits — 16 KiB per 4096 values — else recomputed), pass 2 the blocks
__global__ __launch_bounds__(256) void swiglu_scaled_kernel(const float *__restrict__ gate, size_t gate_row_bytes, const float *__restrict__ up,
                                                            size_t up_row_bytes, long k, float *__restrict__ yf, size_t yf_row_bytes,
                                                            long nrows, scaled_image im) {
    __shared__ float wmax[4];
    const long row = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = (int)(k / 256);
    if (row >= nrows) {
        for (int b = wave; b < nb; b += 4)
            put_scaled_zero(im, b, row, lane);
        if (threadIdx.x == 0)
            im.tok_scale[row] = 0.0f;
        return;
    }
    const float *gr = (const float *)((const uint8_t *)gate + row * gate_row_bytes), *ur = (const float *)((const uint8_t *)up + row * up_row_bytes);
    auto y_of = [&](int b, float (&y)[4]) {
        const float4 g = *(const float4 *)(gr + (size_t)b * 256 + 4 * lane);
        const float4 u = *(const float4 *)(ur + (size_t)b * 256 + 4 * lane);
        const float gv[4] = {g.x, g.y, g.z, g.w}, uv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; e++)
            y[e] = (gv[e] / (1.0f + expf(-gv[e]))) * uv[e];
    };
    float dmax = 0.0f;
    for (int b = wave; b < nb; b += 4) {
        float y[4];
        y_of(b, y);
        dmax = fmaxf(dmax, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
    }
    const float nscale = row_scale<4>(dmax, wmax, wave, lane, im.tok_scale + row);
    float *frow = yf ? (float *)((uint8_t *)yf + row * yf_row_bytes) : nullptr;
    for (int b = wave; b < nb; b += 4) {
        float y[4];
        y_of(b, y); // (the same instructions on the same inputs: the same bits as in pass 1)
        if (frow)
            *(float4 *)(frow + (size_t)b * 256 + 4 * lane) = make_float4(y[0], y[1], y[2], y[3]);
        put_scaled_block(im, b, row, y, nscale, lane);
    }
}

} // namespace

static scaled_image scaled_of(void *image, long k, long nrows) {
    const size_t nb = (size_t)(k / 256), n_pad = ((size_t)nrows + 127) / 128 * 128;
    auto up256 = [](size_t v) { return (v + 255) / 256 * 256; };
    scaled_image im;
    im.Xh = (_Float16 *)image;
    im.tok_scale = (float *)((uint8_t *)image + up256(n_pad * (size_t)k * 2));
    im.Xm = (_Float16 *)((uint8_t *)im.tok_scale + up256(nb * n_pad * 4));
    im.n_pad = (long)n_pad;
    return im;
}

extern "C" size_t lfamd_staged_scaled_size(long k, long nrows) { // = the staging part of lfamd_mul_mat_workspace for these bodies
    if (k <= 0 || k % 256 || nrows < 0)
        return 0;
    const size_t nb = (size_t)(k / 256), n_pad = ((size_t)nrows + 127) / 128 * 128;
    auto up256 = [](size_t v) { return (v + 255) / 256 * 256; };
    return up256(n_pad * (size_t)k * 2) + up256(nb * n_pad * 4) + up256(n_pad * nb * 32);
}

static staged_image staged_of(void *image, long k, long nrows) {
    const long nb = k / 256, n_pad = (nrows + 127) / 128 * 128;
    staged_image im;
    im.Xq = (int8_t *)image;
    im.d8T = (float *)((uint8_t *)image + (size_t)n_pad * nb * 256);
    im.Xs = (_Float16 *)((uint8_t *)im.d8T + (size_t)n_pad * nb * 4);
    im.n_pad = n_pad;
    return im;
}

extern "C" size_t lfamd_staged_q8k_size(long k, long nrows) {
    if (k <= 0 || k % 256 || nrows < 0)
        return 0;
    const size_t n_pad = ((size_t)nrows + 127) / 128 * 128, nb = (size_t)(k / 256);
    return n_pad * nb * 256 + n_pad * nb * 4 + n_pad * nb * 32; // = lfamd_gemm_i8_workspace(k, nrows)
}

extern "C" int lfamd_rms_norm_quantize(const float *d_x, size_t x_row_bytes, const float *d_weight, float eps, long nrows, long k,
                                       int vec_dot_type, void *d_yq, size_t yq_row_bytes, float *d_yf, size_t yf_row_bytes,
                                       void *stream) {
    const bool scaled = d_yq && vec_dot_type == LFAMD_TYPE_STAGED_SCALED; // d_yq = an image of lfamd_staged_scaled_size(k, nrows) bytes
    const bool staged = scaled || (d_yq && vec_dot_type == LFAMD_TYPE_STAGED_Q8K); // ... of lfamd_staged_q8k_size(k, nrows) bytes
    if (nrows < 0 || k <= 0 || k % 256 || (d_yq && vec_dot_type != LFAMD_TYPE_Q8_K && !staged) || (!d_yq && !d_yf) ||
        ((uintptr_t)d_x & 15) || (x_row_bytes & 15) || ((uintptr_t)d_weight & 15) || ((uintptr_t)d_yf & 15) || (yf_row_bytes & 15) ||
        ((uintptr_t)d_yq & (staged ? 15 : 3)) || (!staged && (yq_row_bytes & 3))) {
        lfamd_set_error("lfamd_rms_norm_quantize: k must be a multiple of 256, output format Q8_K (or the staged image, 16-byte aligned), "
                        "16-byte aligned f32 rows");
        return LFAMD_ERR_INVALID;
    }
    if (nrows == 0)
        return LFAMD_OK;
    if (scaled) {
        const scaled_image im = scaled_of(d_yq, k, nrows);
        rms_norm_scaled_kernel<<<(unsigned)im.n_pad, 256, 0, (hipStream_t)stream>>>(d_x, x_row_bytes, d_weight, eps, k, d_yf, yf_row_bytes, nrows,
                                                                                   im);
    } else if (staged) {
        const staged_image im = staged_of(d_yq, k, nrows);
        rms_norm_q8k_kernel<true><<<(unsigned)im.n_pad, 256, 0, (hipStream_t)stream>>>(d_x, x_row_bytes, d_weight, eps, k, nullptr, 0, d_yf,
                                                                                       yf_row_bytes, nrows, im);
    } else {
        rms_norm_q8k_kernel<false><<<(unsigned)nrows, 256, 0, (hipStream_t)stream>>>(d_x, x_row_bytes, d_weight, eps, k, (uint8_t *)d_yq,
                                                                                     yq_row_bytes, d_yf, yf_row_bytes, nrows, staged_image{});
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lfamd_set_error(hipGetErrorString(e));
        return LFAMD_ERR_HIP;
    }
    return LFAMD_OK;
}

extern "C" int lfamd_swiglu_quantize(const float *d_gate, size_t gate_row_bytes, const float *d_up, size_t up_row_bytes, long nrows,
                                     long k, int vec_dot_type, void *d_yq, size_t yq_row_bytes, float *d_yf, size_t yf_row_bytes,
                                     void *stream) {
    const bool scaled = d_yq && vec_dot_type == LFAMD_TYPE_STAGED_SCALED; // d_yq = an image of lfamd_staged_scaled_size(k, nrows) bytes
    const bool staged = scaled || (d_yq && vec_dot_type == LFAMD_TYPE_STAGED_Q8K); // ... of lfamd_staged_q8k_size(k, nrows) bytes
    if (nrows < 0 || k <= 0 || k % 256 || (d_yq && vec_dot_type != LFAMD_TYPE_Q8_K && !staged) || (!d_yq && !d_yf) || !d_gate || !d_up ||
        ((uintptr_t)d_gate & 15) || (gate_row_bytes & 15) || ((uintptr_t)d_up & 15) || (up_row_bytes & 15) || ((uintptr_t)d_yf & 15) ||
        (yf_row_bytes & 15) || ((uintptr_t)d_yq & (staged ? 15 : 3)) || (!staged && (yq_row_bytes & 3)) || nrows > 65535 - 127) {
        lfamd_set_error("lfamd_swiglu_quantize: k must be a multiple of 256, output format Q8_K (or the staged image, 16-byte aligned), "
                        "16-byte aligned f32 rows, <= 65408 rows");
        return LFAMD_ERR_INVALID;
    }
    if (nrows == 0)
        return LFAMD_OK;
    if (scaled) {
        const scaled_image im = scaled_of(d_yq, k, nrows);
        swiglu_scaled_kernel<<<(unsigned)im.n_pad, 256, 0, (hipStream_t)stream>>>(d_gate, gate_row_bytes, d_up, up_row_bytes, k, d_yf, yf_row_bytes,
                                                                                 nrows, im);
    } else if (staged) {
        const staged_image im = staged_of(d_yq, k, nrows);
        const dim3 grid((unsigned)((k / 256 + 3) / 4), (unsigned)im.n_pad);
        swiglu_q8k_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(d_gate, gate_row_bytes, d_up, up_row_bytes, k, nullptr, 0, d_yf,
                                                                      yf_row_bytes, nrows, im);
    } else {
        const dim3 grid((unsigned)((k / 256 + 3) / 4), (unsigned)nrows);
        swiglu_q8k_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(d_gate, gate_row_bytes, d_up, up_row_bytes, k, (uint8_t *)d_yq,
                                                                       yq_row_bytes, d_yf, yf_row_bytes, nrows, staged_image{});
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lfamd_set_error(hipGetErrorString(e));
        return LFAMD_ERR_HIP;
    }
    return LFAMD_OK;
}
