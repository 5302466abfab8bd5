// stream_probe.hip — development probe: what does "launch + stream N MB once" cost on this GPU for the
// GEMV's decomposition (WGs of 256 threads, 16 B per lane, non-temporal loads)?  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_empty(float *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && out == nullptr) out[0] = 0; }

// each WG reads `per_wg` bytes contiguous: U loads of 16 B per lane in flight at once
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stream(const unsigned char *src, size_t per_wg, float *out) {
    const unsigned char *p = src + (size_t)blockIdx.x * per_wg + threadIdx.x * 16;
    unsigned acc = 0;
    for (size_t off = 0; off < per_wg; off += (size_t)U * 4096) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (off + (size_t)u * 4096 < per_wg) {
                const u32x4 *q = (const u32x4 *)(p + off + (size_t)u * 4096);
                v[u] = NT ? __builtin_nontemporal_load(q) : *q;
            } else v[u] = u32x4{0,0,0,0};
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = 1.0f;  // keeps the loads live, practically never taken
}

template <typename F>
float time_graph(F launch, int copies, int iters) {
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int c = 0; c < copies; c++) launch(c, s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / (iters * copies);
}

int main() {
    const int copies = 48;
    float *out; CK(hipMalloc(&out, 1 << 20));
    size_t sizes[] = {2359296, 9437184, 33030144};
    printf("empty kernel in graph: %.2f us/launch\n", time_graph([&](int, hipStream_t s) { k_empty<<<256, 256, 0, s>>>(out); }, copies, 20));
    for (size_t bytes : sizes) {
        unsigned char *buf; CK(hipMalloc(&buf, bytes * copies)); CK(hipMemset(buf, 1, bytes * copies));
        for (int wgs : {256, 512, 1024}) {
            size_t per_wg = bytes / wgs;
            if (per_wg % 4096) continue;
            float t9 = time_graph([&](int c, hipStream_t s) { k_stream<9, true><<<wgs, 256, 0, s>>>(buf + (size_t)c * bytes, per_wg, out); }, copies, 20);
            float t3 = time_graph([&](int c, hipStream_t s) { k_stream<3, true><<<wgs, 256, 0, s>>>(buf + (size_t)c * bytes, per_wg, out); }, copies, 20);
            float t9p = time_graph([&](int c, hipStream_t s) { k_stream<9, false><<<wgs, 256, 0, s>>>(buf + (size_t)c * bytes, per_wg, out); }, copies, 20);
            printf("bytes %9zu wgs %5d per_wg %7zu: U=9 nt %.2f us (%.0f GB/s) | U=3 nt %.2f us | U=9 plain %.2f us\n", bytes, wgs, per_wg, t9, bytes / t9 / 1e3, t3, t9p);
        }
        CK(hipFree(buf));
    }
    return 0;
}
