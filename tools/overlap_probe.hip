// development probe: what a chain of DEPENDENT weight-streaming launches (the decode GEMVs of one layer after another) costs
// per launch boundary on MI355X, and what is left of that cost when launch i + 1 is allowed to start early — resident beside
// launch i, its first weight loads in flight — and waits on a device-side completion counter of launch i instead of on the
// command processor's barrier.
//
//   mode seq   : hipGraph, kernel node i depends on node i - 1                       (what bench.py's decode graph is today)
//   mode early : kernel node i depends on node i - 2 only; in the kernel: first loads, then poll done[i - 1] == gridDim
//                (two launches resident at a time: both must fit on a CU together; the poll has a wall-clock timeout)
//   mode free  : node i depends on node i - 2, no flag wait                          (upper bound: no dependency at all)
//
//   hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/overlap_probe.bin && tools/overlap_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <vector>

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int DEPTH = 8; // dwordx4 loads in flight per lane

__device__ unsigned g_timeouts;

// every work-group streams `per_wg` bytes (a multiple of blockDim * 16 * DEPTH) and folds them into one value per lane
template <int NT>
__global__ __launch_bounds__(NT) void stream_k(const u32x4 *__restrict__ w, long per_wg16, float *out, const unsigned *wait_flag,
                                               unsigned expect, unsigned *done_flag, const float *xin) {
    const u32x4 *p = w + (long)blockIdx.x * per_wg16 + threadIdx.x;
    const long steps = per_wg16 / NT; // loads per lane
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        v[d] = d < steps ? __builtin_nontemporal_load(p + (long)d * NT) : (u32x4)0;
    float x = 1.0f;
    if (wait_flag) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
            while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 500000ull) { // 5 ms
                    atomicAdd(&g_timeouts, 1u);
                    break;
                }
            }
        }
        __syncthreads();
    }
    if (xin) // the activation: written by the launch before, read past its L2
        x = __hip_atomic_load(xin + (threadIdx.x & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t acc = 0;
    for (long s = DEPTH; s < steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
            v[d] = s + d < steps ? __builtin_nontemporal_load(p + (s + d) * NT) : (u32x4)0;
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
    // result: 256 floats from work-group 0, written through to memory (agent scope)
    if (blockIdx.x == 0 && threadIdx.x < 256)
        __hip_atomic_store(out + threadIdx.x, x * 0.5f + (float)(acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done_flag) {
        __builtin_amdgcn_s_waitcnt(0); // the stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add(done_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one persistent launch: `phases` streaming phases, a grid barrier between them (monotonic arrival counter: work-group adds 1,
// waits for phase * gridDim), the first loads of the next phase issued BEFORE the barrier.  PREF = false: loads after it.
template <int NT, bool PREF>
__global__ __launch_bounds__(NT) void pers_k(const u32x4 *__restrict__ w, long per_wg16, long phase_stride16, int phases, float *out,
                                             unsigned *counter, const float *xin) {
    const long steps = per_wg16 / NT;
    u32x4 v[DEPTH];
    uint32_t acc = 0;
    float x = 1.0f;
    const u32x4 *p = w + (long)blockIdx.x * per_wg16 + threadIdx.x;
    if (PREF) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            v[d] = d < steps ? __builtin_nontemporal_load(p + (long)d * NT) : (u32x4)0;
    }
    for (int ph = 0; ph < phases; ph++) {
        if (ph) { // barrier: everything of phase ph - 1 has been written through
            if (threadIdx.x == 0) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                const unsigned want = (unsigned)ph * gridDim.x;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 500000ull) {
                        atomicAdd(&g_timeouts, 1u);
                        break;
                    }
                }
            }
            __syncthreads();
            x += __hip_atomic_load(xin + ph * 256 + (threadIdx.x & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!PREF) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
                v[d] = d < steps ? __builtin_nontemporal_load(p + (long)d * NT) : (u32x4)0;
        }
        for (long s = DEPTH; s < steps; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
                v[d] = s + d < steps ? __builtin_nontemporal_load(p + (s + d) * NT) : (u32x4)0;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
        p += phase_stride16;
        if (PREF && ph + 1 < phases) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
                v[d] = d < steps ? __builtin_nontemporal_load(p + (long)d * NT) : (u32x4)0;
        }
        if (blockIdx.x == 0 && threadIdx.x < 256)
            __hip_atomic_store(out + (ph + 1) * 256 + threadIdx.x, x * 0.5f + (float)(acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ph + 1 < phases) {
            if (blockIdx.x == 0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 0x12345u)
        out[0] = x;
}

template <int NT, bool PREF>
static void run_pers(int phases, size_t bytes_per_phase, uint8_t *pool, size_t pool_bytes, unsigned *flags, float *acts, hipStream_t s) {
    const int grid = 256;
    const long per_wg16 = (long)(bytes_per_phase / grid / 16 / NT) * NT;
    const size_t real_bytes = (size_t)per_wg16 * 16 * grid;
    if (real_bytes * phases > pool_bytes)
        phases = (int)(pool_bytes / real_bytes);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        CHECK(hipMemsetAsync(flags, 0, 4, s));
        CHECK(hipEventRecord(e0, s));
        pers_k<NT, PREF><<<grid, NT, 0, s>>>((const u32x4 *)pool, per_wg16, (long)(real_bytes / 16), phases, acts, flags, acts);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best)
            best = ms;
    }
    unsigned to = 0;
    CHECK(hipMemcpyFromSymbol(&to, HIP_SYMBOL(g_timeouts), 4));
    const double us = best * 1e3 / phases;
    printf("pers  NT=%4d grid= 256 %6.1f MB/phase  x %3d phases  : %7.2f us/phase   %6.2f TB/s  prefetch-before-barrier=%d%s\n", NT, real_bytes / 1e6,
           phases, us, real_bytes / us / 1e6, (int)PREF, to ? "  (POLL TIMEOUTS!)" : "");
}

// the same persistent kernel with the runtime's own grid barrier (cooperative launch, cooperative_groups::this_grid().sync())
template <int NT>
__global__ __launch_bounds__(NT) void pers_cg_k(const u32x4 *__restrict__ w, long per_wg16, long phase_stride16, int phases, float *out) {
    cooperative_groups::grid_group grid = cooperative_groups::this_grid();
    const long steps = per_wg16 / NT;
    u32x4 v[DEPTH];
    uint32_t acc = 0;
    const u32x4 *p = w + (long)blockIdx.x * per_wg16 + threadIdx.x;
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        v[d] = d < steps ? __builtin_nontemporal_load(p + (long)d * NT) : (u32x4)0;
    for (int ph = 0; ph < phases; ph++) {
        if (ph)
            grid.sync();
        for (long s = DEPTH; s < steps; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
                v[d] = s + d < steps ? __builtin_nontemporal_load(p + (s + d) * NT) : (u32x4)0;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
        p += phase_stride16;
        if (ph + 1 < phases) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
                v[d] = d < steps ? __builtin_nontemporal_load(p + (long)d * NT) : (u32x4)0;
        }
    }
    if (acc == 0x12345u)
        out[0] = 1.0f;
}

template <int NT>
static void run_pers_cg(int phases, size_t bytes_per_phase, uint8_t *pool, size_t pool_bytes, float *acts, hipStream_t s) {
    const int grid = 256;
    long per_wg16 = (long)(bytes_per_phase / grid / 16 / NT) * NT;
    const size_t real_bytes = (size_t)per_wg16 * 16 * grid;
    if (real_bytes * phases > pool_bytes)
        phases = (int)(pool_bytes / real_bytes);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const u32x4 *w = (const u32x4 *)pool;
    long stride = (long)(real_bytes / 16);
    void *args[] = {&w, &per_wg16, &stride, &phases, &acts};
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        CHECK(hipEventRecord(e0, s));
        CHECK(hipLaunchCooperativeKernel((const void *)pers_cg_k<NT>, dim3(grid), dim3(NT), args, 0, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best)
            best = ms;
    }
    const double us = best * 1e3 / phases;
    printf("coop  NT=%4d grid= 256 %6.1f MB/phase  x %3d phases  : %7.2f us/phase   %6.2f TB/s  (cooperative_groups grid.sync)\n", NT, real_bytes / 1e6,
           phases, us, real_bytes / us / 1e6);
}

struct Args {
    const u32x4 *w;
    long per_wg16;
    float *out;
    const unsigned *wait_flag;
    unsigned expect;
    unsigned *done_flag;
    const float *xin;
};

template <int NT>
static double run(const char *mode, int n_kernels, size_t bytes_per_kernel, int grid, uint8_t *pool, size_t pool_bytes, unsigned *flags, float *acts,
                  hipStream_t s) {
    const long per_wg16 = (long)(bytes_per_kernel / grid / 16 / NT) * NT;
    const size_t real_bytes = (size_t)per_wg16 * 16 * grid;
    const bool seq = !strcmp(mode, "seq"), early = !strcmp(mode, "early");
    hipGraph_t g;
    CHECK(hipGraphCreate(&g, 0));
    hipGraphNode_t root;
    hipMemsetParams mp = {};
    mp.dst = flags, mp.value = 0, mp.elementSize = 4, mp.width = n_kernels + 1, mp.height = 1, mp.pitch = (n_kernels + 1) * 4;
    CHECK(hipGraphAddMemsetNode(&root, g, nullptr, 0, &mp));
    std::vector<hipGraphNode_t> nodes(n_kernels);
    std::vector<Args> args(n_kernels);
    std::vector<std::array<void *, 7>> argv(n_kernels);
    const size_t n_slots = pool_bytes / real_bytes;
    for (int i = 0; i < n_kernels; i++) {
        Args &a = args[i];
        a.w = (const u32x4 *)(pool + (size_t)(i % n_slots) * real_bytes);
        a.per_wg16 = per_wg16;
        a.out = acts + (size_t)(i + 1) * 256;
        a.wait_flag = early && i > 0 ? flags + (i - 1) : nullptr;
        a.expect = grid;
        a.done_flag = early ? flags + i : nullptr;
        a.xin = acts + (size_t)i * 256;
        void **av = argv[i].data();
        av[0] = &a.w, av[1] = &a.per_wg16, av[2] = &a.out, av[3] = &a.wait_flag, av[4] = &a.expect, av[5] = &a.done_flag, av[6] = &a.xin;
        hipKernelNodeParams kp = {};
        kp.func = (void *)stream_k<NT>;
        kp.gridDim = dim3(grid), kp.blockDim = dim3(NT), kp.sharedMemBytes = 0, kp.kernelParams = av, kp.extra = nullptr;
        hipGraphNode_t deps[2];
        int nd = 0;
        if (seq)
            deps[nd++] = i ? nodes[i - 1] : root;
        else {
            deps[nd++] = i >= 2 ? nodes[i - 2] : root;
        }
        CHECK(hipGraphAddKernelNode(&nodes[i], g, deps, nd, &kp));
    }
    hipGraphExec_t ge;
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 2; r++)
        CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    const int reps = 5;
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; r++)
        CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned to = 0;
    CHECK(hipMemcpyFromSymbol(&to, HIP_SYMBOL(g_timeouts), 4));
    const double us = ms * 1e3 / reps / n_kernels;
    printf("%-5s NT=%4d grid=%4d %6.1f MB/launch x %3d launches: %7.2f us/launch  %6.2f TB/s%s\n", mode, NT, grid, real_bytes / 1e6, n_kernels, us,
           real_bytes / us / 1e6, to ? "  (POLL TIMEOUTS!)" : "");
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    return us;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const size_t pool_bytes = 2ull << 30; // beyond the 256 MiB MALL
    uint8_t *pool;
    unsigned *flags;
    float *acts;
    CHECK(hipMalloc(&pool, pool_bytes));
    CHECK(hipMemset(pool, 0x5a, pool_bytes));
    CHECK(hipMalloc(&flags, 4096 * 4));
    CHECK(hipMalloc(&acts, 4096 * 256 * 4));
    CHECK(hipMemset(acts, 0, 4096 * 256 * 4));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    const int n = 128;
    if (argc < 2)
    for (const char *mode : {"seq", "free", "early"})
        for (size_t mb : {9, 33, 66}) {
            run<1024>(mode, n, mb << 20, 256, pool, pool_bytes, flags, acts, s);
            run<512>(mode, n, mb << 20, 256, pool, pool_bytes, flags, acts, s);
            run<512>(mode, n, mb << 20, 512, pool, pool_bytes, flags, acts, s);
        }
    for (size_t mb : {4, 9, 33, 66}) {
        run_pers_cg<1024>(n, mb << 20, pool, pool_bytes, acts, s);
        run_pers<1024, false>(n, mb << 20, pool, pool_bytes, flags, acts, s);
        run_pers<1024, true>(n, mb << 20, pool, pool_bytes, flags, acts, s);
        run_pers<512, true>(n, mb << 20, pool, pool_bytes, flags, acts, s);
    }
    return 0;
}
