// quantize.hip — f32 activations -> the reference's activation block formats, on the device.
//
// The reference quantises src1 on the CPU inside ggml_compute_forward_mul_mat (SURVEY.md §3.1 step 2,
// upstream quantize_row_q8_0 / q8_1 / q8_K) and its GPU backend has quantize_q8_1
// (ggml-cuda.cu.patch:15259-15293).  These kernels reproduce the scalar reference quantisers bit for
// bit (same divisions, roundf / nearest-even, first-maximum rule of q8_K), and emit llamafile's
// Q8_K field order {d, bsums, qs} (ggml-common.h.patch:25-35).
#include "lfamd_device.h"

// one wave (64 threads) handles two 32-blocks
template <bool Q81>
__global__ __launch_bounds__(64) void quantize_q8_01_kernel(const float *__restrict__ x, size_t x_row_bytes, long cols,
                                                            uint8_t *__restrict__ y, size_t y_row_bytes) {
    const long row = blockIdx.y;
    const long blk = (long)blockIdx.x * 2 + (threadIdx.x >> 5);
    const int l = threadIdx.x & 31;
    const long nblk = cols / 32;
    const bool valid = blk < nblk;
    const float *xr = (const float *)((const uint8_t *)x + row * x_row_bytes);
    const float v = valid ? xr[blk * 32 + l] : 0.0f;
    float amax = fabsf(v);
    for (int off = 16; off > 0; off >>= 1)
        amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    const float d = amax / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    const int q = (int)roundf(v * id);
    if (!valid)
        return;
    constexpr int BS = Q81 ? 36 : 34;
    uint8_t *yb = y + row * y_row_bytes + (size_t)blk * BS;
    ((int8_t *)yb)[(Q81 ? 4 : 2) + l] = (int8_t)q;
    if constexpr (Q81) {
        int sum = q;
        for (int off = 16; off > 0; off >>= 1)
            sum += __shfl_xor(sum, off, 64);
        if (l == 0) {
            *(uint16_t *)yb = f2h_bits(d);
            *(uint16_t *)(yb + 2) = f2h_bits((float)sum * d);
        }
    } else if (l == 0) {
        *(uint16_t *)yb = f2h_bits(d);
    }
}

// one work-group of 256 threads per 256-block
__global__ __launch_bounds__(256) void quantize_q8_K_kernel(const float *__restrict__ x, size_t x_row_bytes, long cols,
                                                            uint8_t *__restrict__ y, size_t y_row_bytes) {
    __shared__ float s_amax[4];
    __shared__ int s_idx[4];
    __shared__ float s_val[4];
    __shared__ int s_bs[16];
    const long row = blockIdx.y, blk = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *xr = (const float *)((const uint8_t *)x + row * x_row_bytes) + blk * 256;
    const float v = xr[t];
    // first index of the largest |x| (upstream: `if (ax > amax) { amax = ax; max = x[j]; }`)
    float amax = fabsf(v);
    int idx = t;
    float val = v;
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
    if (lane == 0) {
        s_amax[wave] = amax;
        s_idx[wave] = idx;
        s_val[wave] = val;
    }
    if (t < 16)
        s_bs[t] = 0;
    __syncthreads();
    amax = s_amax[0];
    val = s_val[0];
    idx = s_idx[0];
    for (int w = 1; w < 4; w++)
        if (s_amax[w] > amax || (s_amax[w] == amax && s_idx[w] < idx)) {
            amax = s_amax[w];
            idx = s_idx[w];
            val = s_val[w];
        }
    uint8_t *yb = y + row * y_row_bytes + (size_t)blk * 292;
    if (amax == 0.0f) {
        ((int8_t *)yb)[36 + t] = 0;
        if (t < 16)
            ((int16_t *)(yb + 4))[t] = 0;
        if (t == 0)
            *(float *)yb = 0.0f;
        return;
    }
    const float iscale = -128.0f / val;
    int q = (int)rintf(iscale * v); // nearest_int(): round-half-even
    q = q > 127 ? 127 : q;
    ((int8_t *)yb)[36 + t] = (int8_t)q;
    // bsums: 16 consecutive codes
    int s = q;
    for (int off = 8; off > 0; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if ((t & 15) == 0)
        ((int16_t *)(yb + 4))[t >> 4] = (int16_t)s;
    if (t == 0)
        *(float *)yb = 1.0f / iscale;
}

extern "C" hipError_t lfamd_launch_quantize(int vdt, const float *x, long nrows, long cols, size_t x_row_bytes, void *y,
                                            size_t y_row_bytes, hipStream_t s) {
    if (nrows <= 0 || cols <= 0)
        return hipSuccess;
    if (vdt == LFAMD_TYPE_Q8_K) {
        dim3 grid((unsigned)(cols / 256), (unsigned)nrows);
        quantize_q8_K_kernel<<<grid, 256, 0, s>>>(x, x_row_bytes, cols, (uint8_t *)y, y_row_bytes);
    } else {
        dim3 grid((unsigned)((cols / 32 + 1) / 2), (unsigned)nrows);
        if (vdt == LFAMD_TYPE_Q8_1)
            quantize_q8_01_kernel<true><<<grid, 64, 0, s>>>(x, x_row_bytes, cols, (uint8_t *)y, y_row_bytes);
        else
            quantize_q8_01_kernel<false><<<grid, 64, 0, s>>>(x, x_row_bytes, cols, (uint8_t *)y, y_row_bytes);
    }
    return hipGetLastError();
}
