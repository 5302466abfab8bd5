// gemv_float.hip — decode (n <= 8) mat-vec for float weights: F16 / BF16 / F32 rows in RAW layout.
//
// Reference: tinyBLAS<..., ggml_fp16_t | ggml_bf16_t | float, float, float> (tinyblas_cpu.h:419-613 through
// tinyblas_cpu_sgemm.inc:45-150): weights widened to f32, f32 activations, f32 fused multiply-add, per-tile horizontal
// sums.  Same products here; only the order of the f32 additions differs (tests: 1e-3 normwise like every float path).
//
// HBM-bound: 2 or 4 bytes per weight streamed once.  A wave owns two weight rows at a time and walks K in chunks of
// 64 lanes x 16 bytes; the loads of a whole group of chunks (2 rows x GF_GROUP x 16 B per lane) are in flight before
// the first is consumed, and the first group goes out BEFORE the work-group stages the activations into LDS as f32.
// Rows are fetched through one bounds-checked descriptor each (row bytes = records), so the K tail of a row whose
// length is not a whole group reads zeros, and the zero-padded LDS image keeps those products at 0.
//
// LDS image of one activation column: kpad f32.  For 2-byte weights a lane needs 8 consecutive activations per chunk;
// they are stored as two 16-byte halves at ((2 chunk + half) * 64 + lane) * 16 so both ds_read_b128 are conflict-free.
#include "lfamd_device.h"

#define GF_WAVES 8
#define GF_GROUP 8

template <int ATYPE>
struct gf_type {
    static constexpr int AS = ATYPE == LFAMD_TYPE_F32 ? 4 : 2;
    static constexpr int EPL = 16 / AS;   // elements per lane per 16-byte load
    static constexpr int CHUNK = 64 * EPL; // elements per wave-wide load
};

template <int ATYPE>
__device__ static inline void gf_widen(const uint4 w, float (&f)[8]) {
    const uint32_t v[4] = {w.x, w.y, w.z, w.w};
    if constexpr (ATYPE == LFAMD_TYPE_F32) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            f[e] = __builtin_bit_cast(float, v[e]);
    } else if constexpr (ATYPE == LFAMD_TYPE_F16) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            f[2 * e] = h2f((uint16_t)(v[e] & 0xffff));
            f[2 * e + 1] = h2f((uint16_t)(v[e] >> 16));
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            f[2 * e] = __builtin_bit_cast(float, v[e] << 16);
            f[2 * e + 1] = __builtin_bit_cast(float, v[e] & 0xffff0000u);
        }
    }
}

__device__ static inline float gf_wave_sum(float v) {
    v += dpp_f32<DPP_XOR1>(v);
    v += dpp_f32<DPP_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v);
    v += dpp_f32<DPP_MIRROR>(v);
    return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}

// activations of columns col0 .. col0 + NC - 1 -> f32 image (zero padded to kpad)
template <int ATYPE, int BTYPE, int NC>
__device__ static inline void gf_stage(uint8_t *lds, const uint8_t *B, size_t b_row_bytes, long col0, long k, int kpad) {
    constexpr int EPL = gf_type<ATYPE>::EPL;
    const int quads = kpad / 4; // float4 slots per column
    for (int c = 0; c < NC; c++) {
        const uint8_t *x = B + (col0 + c) * b_row_bytes;
        float *img = (float *)lds + (size_t)c * kpad;
        for (int i = threadIdx.x; i < quads; i += blockDim.x) {
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if ((long)i * 4 < k) { // (k % 4 == 0: checked by the launcher)
                if constexpr (BTYPE == LFAMD_TYPE_F32) {
                    v = *(const float4 *)(x + (size_t)i * 16);
                } else {
                    const uint2 h = *(const uint2 *)(x + (size_t)i * 8);
                    if constexpr (BTYPE == LFAMD_TYPE_F16) {
                        v = make_float4(h2f((uint16_t)(h.x & 0xffff)), h2f((uint16_t)(h.x >> 16)), h2f((uint16_t)(h.y & 0xffff)),
                                        h2f((uint16_t)(h.y >> 16)));
                    } else {
                        v = make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                                        __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u));
                    }
                }
            }
            int slot = i;
            if constexpr (EPL == 8) { // quad i = elements 4i..4i+3 = chunk i/128, lane (i%128)/2, half i&1
                const int chunk = i >> 7, lane = (i & 127) >> 1, half = i & 1;
                slot = (2 * chunk + half) * 64 + lane;
            }
            *(float4 *)(img + (size_t)slot * 4) = v;
        }
    }
}

template <int ATYPE, int BTYPE, int NC>
__global__ __launch_bounds__(GF_WAVES * 64) void gemv_float_kernel(const uint8_t *__restrict__ A, long m, long k,
                                                                    const uint8_t *__restrict__ B, size_t b_row_bytes,
                                                                    long col0, float *__restrict__ C, long ldc, int kpad) {
    extern __shared__ __attribute__((aligned(16))) uint8_t gf_lds[];
    constexpr int AS = gf_type<ATYPE>::AS, EPL = gf_type<ATYPE>::EPL, CHUNK = gf_type<ATYPE>::CHUNK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long npairs = (m + 1) / 2, stride = (long)gridDim.x * GF_WAVES;
    const uint32_t row_bytes = (uint32_t)(k * AS);
    const int ngroups = kpad / (CHUNK * GF_GROUP);
    const long first = (long)blockIdx.x * GF_WAVES + wave;

    uint4 w[2][GF_GROUP];
    lfamd_rsrc r0 = make_rsrc(A + (size_t)(2 * first) * row_bytes, 2 * first < m ? row_bytes : 0);
    lfamd_rsrc r1 = make_rsrc(A + (size_t)(2 * first + 1) * row_bytes, 2 * first + 1 < m ? row_bytes : 0);
#pragma unroll
    for (int j = 0; j < GF_GROUP; j++) {
        const uint32_t off = (uint32_t)((j * 64 + lane) * 16);
        w[0][j] = buf_ld16_nt(r0, off);
        w[1][j] = buf_ld16_nt(r1, off);
    }
    gf_stage<ATYPE, BTYPE, NC>(gf_lds, B, b_row_bytes, col0, k, kpad);
    __syncthreads();

    for (long pair = first; pair < npairs; pair += stride) {
        if (pair != first) {
            r0 = make_rsrc(A + (size_t)(2 * pair) * row_bytes, row_bytes);
            r1 = make_rsrc(A + (size_t)(2 * pair + 1) * row_bytes, 2 * pair + 1 < m ? row_bytes : 0);
        }
        float acc[2][NC];
#pragma unroll
        for (int c = 0; c < NC; c++)
            acc[0][c] = acc[1][c] = 0.0f;
        for (int g = 0; g < ngroups; g++) {
            if (pair != first || g != 0) {
#pragma unroll
                for (int j = 0; j < GF_GROUP; j++) {
                    const uint32_t off = (uint32_t)(((g * GF_GROUP + j) * 64 + lane) * 16);
                    w[0][j] = buf_ld16_nt(r0, off);
                    w[1][j] = buf_ld16_nt(r1, off);
                }
            }
#pragma unroll
            for (int j = 0; j < GF_GROUP; j++) {
                const int chunk = g * GF_GROUP + j;
                float f0[8], f1[8];
                gf_widen<ATYPE>(w[0][j], f0);
                gf_widen<ATYPE>(w[1][j], f1);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const float *img = (const float *)gf_lds + (size_t)c * kpad;
                    float x[8];
                    if constexpr (EPL == 8) {
                        const float4 xa = *(const float4 *)(img + ((2 * chunk + 0) * 64 + lane) * 4);
                        const float4 xb = *(const float4 *)(img + ((2 * chunk + 1) * 64 + lane) * 4);
                        x[0] = xa.x, x[1] = xa.y, x[2] = xa.z, x[3] = xa.w, x[4] = xb.x, x[5] = xb.y, x[6] = xb.z, x[7] = xb.w;
                    } else {
                        const float4 xa = *(const float4 *)(img + (chunk * 64 + lane) * 4);
                        x[0] = xa.x, x[1] = xa.y, x[2] = xa.z, x[3] = xa.w;
                    }
#pragma unroll
                    for (int e = 0; e < EPL; e++) {
                        acc[0][c] = fmaf(f0[e], x[e], acc[0][c]);
                        acc[1][c] = fmaf(f1[e], x[e], acc[1][c]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float s0 = gf_wave_sum(acc[0][c]), s1 = gf_wave_sum(acc[1][c]);
            if (lane == 0) {
                C[(col0 + c) * ldc + 2 * pair] = s0;
                if (2 * pair + 1 < m)
                    C[(col0 + c) * ldc + 2 * pair + 1] = s1;
            }
        }
    }
}

static int gf_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            n = p.multiProcessorCount;
        if (n <= 0)
            n = 256;
    }
    return n;
}

// elements per group of chunks; the LDS image of a column is k rounded up to it
static int gf_group_elems(int Atype) {
    return (Atype == LFAMD_TYPE_F32 ? 256 : 512) * GF_GROUP;
}

// Can this shape take the float GEMV?  (16-byte aligned rows, one column's image within the LDS budget)
extern "C" int lfamd_gemv_float_ok(int Atype, long k, long n) {
    if (Atype != LFAMD_TYPE_F32 && Atype != LFAMD_TYPE_F16 && Atype != LFAMD_TYPE_BF16)
        return 0;
    if (n < 1 || n > 8 || k <= 0 || k % 8)
        return 0;
    const long ge = gf_group_elems(Atype), kpad = (k + ge - 1) / ge * ge;
    return kpad * 4 <= 150 * 1024;
}

template <int ATYPE, int BTYPE>
static hipError_t gf_go(const void *A, long m, long k, const void *B, size_t brb, long n, float *C, long ldc, hipStream_t s) {
    const long ge = gf_group_elems(ATYPE);
    const int kpad = (int)((k + ge - 1) / ge * ge);
    int step = (int)((150 * 1024) / ((size_t)kpad * 4));
    step = step > 8 ? 8 : step;
    const long npairs = (m + 1) / 2;
    for (long col0 = 0; col0 < n; col0 += step) {
        const int nc = (int)((n - col0) < step ? (n - col0) : step);
        const size_t smem = (size_t)nc * kpad * 4;
        const long per_cu = smem <= 72 * 1024 ? 2 : 1;
        long grid = (npairs + GF_WAVES - 1) / GF_WAVES;
        if (grid > gf_num_cus() * per_cu)
            grid = gf_num_cus() * per_cu;
#define GF_CASE(NC)                                                                                                    \
    case NC: {                                                                                                         \
        auto kernel = gemv_float_kernel<ATYPE, BTYPE, NC>;                                                             \
        if (smem > 64 * 1024) {                                                                                        \
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (e != hipSuccess)                                                                                       \
                return e;                                                                                              \
        }                                                                                                              \
        kernel<<<(unsigned)grid, GF_WAVES * 64, smem, s>>>((const uint8_t *)A, m, k, (const uint8_t *)B, brb, col0, C, ldc, kpad); \
        break;                                                                                                         \
    }
        switch (nc) {
            GF_CASE(1)
            GF_CASE(2)
            GF_CASE(3)
            GF_CASE(4)
            GF_CASE(5)
            GF_CASE(6)
            GF_CASE(7)
            GF_CASE(8)
        }
#undef GF_CASE
    }
    return hipGetLastError();
}

// Btype: LFAMD_TYPE_F32 or the weight type itself
extern "C" hipError_t lfamd_launch_gemv_float(int Atype, const void *A, long m, long k, int Btype, const void *B,
                                              size_t b_row_bytes, long n, float *C, long ldc, hipStream_t s) {
    if (m <= 0 || n <= 0)
        return hipSuccess;
    if (!lfamd_gemv_float_ok(Atype, k, n) || (Btype != LFAMD_TYPE_F32 && Btype != Atype))
        return hipErrorInvalidValue;
    const bool f = Btype == LFAMD_TYPE_F32;
    switch (Atype) {
    case LFAMD_TYPE_F32:
        return gf_go<LFAMD_TYPE_F32, LFAMD_TYPE_F32>(A, m, k, B, b_row_bytes, n, C, ldc, s);
    case LFAMD_TYPE_F16:
        return f ? gf_go<LFAMD_TYPE_F16, LFAMD_TYPE_F32>(A, m, k, B, b_row_bytes, n, C, ldc, s)
                 : gf_go<LFAMD_TYPE_F16, LFAMD_TYPE_F16>(A, m, k, B, b_row_bytes, n, C, ldc, s);
    default:
        return f ? gf_go<LFAMD_TYPE_BF16, LFAMD_TYPE_F32>(A, m, k, B, b_row_bytes, n, C, ldc, s)
                 : gf_go<LFAMD_TYPE_BF16, LFAMD_TYPE_BF16>(A, m, k, B, b_row_bytes, n, C, ldc, s);
    }
}
