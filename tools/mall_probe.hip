// development probe: does a side-stream "touch" of the NEXT op's weights (into the 256 MB Infinity Cache) shorten a
// chain of HBM-bound streaming kernels?  Build: hipcc -O3 --offload-arch=gfx950 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// consumer: every byte read once (16 B per lane, grid-stride), like the GEMV's weight stream
__global__ __launch_bounds__(1024) void consume(const u32x4 *p, size_t n16, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = __builtin_nontemporal_load(p + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// toucher: one dword per 64 bytes
__global__ __launch_bounds__(256) void touch(const unsigned *p, size_t n64, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += (size_t)gridDim.x * blockDim.x)
        acc ^= __builtin_nontemporal_load(p + i * 16);
    if (acc == 0x12345678u) out[1] = acc;
}

int main(int argc, char **argv) {
    size_t chunk = (argc > 1 ? atol(argv[1]) : 33) << 20;
    int nchunk = argc > 2 ? atoi(argv[2]) : 60;
    int tgrid = argc > 3 ? atoi(argv[3]) : 256;
    char *buf; unsigned *out;
    CHK(hipMalloc(&buf, chunk * nchunk)); CHK(hipMalloc(&out, 64));
    CHK(hipMemset(buf, 1, chunk * nchunk));
    hipStream_t s1, s2; CHK(hipStreamCreate(&s1)); CHK(hipStreamCreate(&s2));
    hipEvent_t t0, t1; CHK(hipEventCreate(&t0)); CHK(hipEventCreate(&t1));
    auto C = [&](int i, hipStream_t s) { consume<<<256, 1024, 0, s>>>((const u32x4 *)(buf + chunk * i), chunk / 16, out); };
    auto P = [&](int i, hipStream_t s) { touch<<<tgrid, 256, 0, s>>>((const unsigned *)(buf + chunk * i), chunk / 64, out); };
    for (int mode = 0; mode < 3; mode++) {
        // mode 0: chain of consumers; mode 1: touch(i) then consume(i) on one stream; mode 2: side stream touches i+1 while i runs
        hipGraph_t g; hipGraphExec_t ge;
        std::vector<hipEvent_t> ev(nchunk + 1), ev2(nchunk + 1);
        for (size_t q = 0; q < ev.size(); q++) CHK(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
        for (size_t q = 0; q < ev2.size(); q++) CHK(hipEventCreateWithFlags(&ev2[q], hipEventDisableTiming));
        CHK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        if (mode == 2) {
            CHK(hipEventRecord(ev[0], s1));
            CHK(hipStreamWaitEvent(s2, ev[0], 0));
        }
        for (int i = 0; i < nchunk; i++) {
            if (mode == 1) P(i, s1);
            if (mode == 2 && i + 1 < nchunk) { // touch(i+1) may start once consume(i-1) is done
                P(i + 1, s2);
            }
            C(i, s1);
            if (mode == 2) {
                CHK(hipEventRecord(ev[i + 1], s1));
                CHK(hipStreamWaitEvent(s2, ev[i + 1], 0)); // next touch waits for this consumer: run-ahead of one op
            }
        }
        if (mode == 2) {
            CHK(hipEventRecord(ev2[0], s2));
            CHK(hipStreamWaitEvent(s1, ev2[0], 0));
        }
        CHK(hipStreamEndCapture(s1, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 2; r++) CHK(hipGraphLaunch(ge, s1));
        CHK(hipStreamSynchronize(s1));
        CHK(hipEventRecord(t0, s1));
        const int reps = 5;
        for (int r = 0; r < reps; r++) CHK(hipGraphLaunch(ge, s1));
        CHK(hipEventRecord(t1, s1));
        CHK(hipStreamSynchronize(s1));
        float ms; CHK(hipEventElapsedTime(&ms, t0, t1));
        double us = ms * 1e3 / reps / nchunk;
        printf("mode %d: chunk %zu MB x %d: %.2f us per op  -> %.2f TB/s effective\n", mode, chunk >> 20, nchunk, us, chunk / us / 1e6);
    }
    return 0;
}
