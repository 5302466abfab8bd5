// oneshot_impl.h — the one-shot peer exchange protocol (comm.hip), as device code that other kernels can end with:
// the decode GEMV of attn_output / ffn_down runs it in its LAST work-group to finish (gemv_impl.h: gemv_kq_fx_kernel),
// so that product and all-reduce are one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ONESHOT_MAX_WORLD 8
#define ONESHOT_WGS 8      // work-groups per launch; WG w owns chunk w of the message and flag column w
#define ONESHOT_THREADS 256

// exchange block: [flags: ONESHOT_MAX_WORLD x ONESHOT_WGS x 64 B][slot 0][slot 1]
#define ONESHOT_FLAGS_BYTES (ONESHOT_MAX_WORLD * ONESHOT_WGS * 64)

struct oneshot_args {
    uint8_t *peer[ONESHOT_MAX_WORLD];
    int rank, world;
    size_t slot_bytes;
    long count; // floats
    long timeout_ticks; // of the 100 MHz wall clock
    int *state; // [0] error flag, [1 + w] the number of calls work-group w has served: the call's sequence number lives on
                // the DEVICE, so a captured launch advances it on every graph replay (every rank issues the same calls);
                // [1 + ONESHOT_WGS] arrival counter of the fused GEMV launch
};

typedef float oneshot_f4 __attribute__((ext_vector_type(4)));

// system-scope accesses (write-through stores, cache-bypassing loads): the only forms another GPU can observe / that
// observe another GPU's stores inside a running kernel
__device__ static inline void st_sys16(void *p, float4 f) {
    const oneshot_f4 v = {f.x, f.y, f.z, f.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ static inline float4 ld_sys16(const void *p) {
    oneshot_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
// four system-scope loads in flight together, one wait (issue and wait in ONE statement: the compiler does not track loads
// made by inline asm, so no result register may be visible to it before the wait)
__device__ static inline void ld_sys16x4(oneshot_f4 &a, oneshot_f4 &b, oneshot_f4 &c, oneshot_f4 &d, const void *pa, const void *pb,
                                         const void *pc, const void *pd) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(pa), "v"(pb), "v"(pc), "v"(pd)
                 : "memory");
}
__device__ static inline void st_sys4(void *p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ static inline uint32_t ld_sys4(const void *p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// The whole exchange by ONE work-group of NT threads (all ONESHOT_WGS chunks: it advances every chunk's sequence number and
// raises / awaits every flag column, so launches of the 8-work-group kernel and fused launches can alternate freely):
//   out[i] = (residual ? residual[i] : 0) + sum over ranks, in rank order, of partial_r[i]       (count % 4 == 0)
// `partial` was written with system-scope stores by the work-groups of this launch (all retired: the caller is the last
// arrival) and is read with system-scope loads; NULL = the work-groups stored their rows straight into slot (seq & 1) of
// this rank's block, seq being the number this call takes.  `sh` = two ints of LDS.
template <int NT>
__device__ static inline void oneshot_whole(const oneshot_args &a, const float *partial, const float *residual, float *out, int *sh) {
    const int tid = (int)threadIdx.x;
    const long quads = a.count / 4;
    uint8_t *mine = a.peer[a.rank];
    if (tid == 0) {
        int seq = 0;
        for (int w = 0; w < ONESHOT_WGS; w++)
            seq = ++a.state[1 + w];
        sh[0] = seq;
        sh[1] = a.state[0];
    }
    __syncthreads();
    const uint32_t seq = (uint32_t)sh[0];
    const int dead = sh[1];
    const size_t slot_off = ONESHOT_FLAGS_BYTES + (size_t)(seq & 1) * a.slot_bytes;
    // (four quads per thread in flight, one wait: a single work-group must not pay a memory round trip per quad)
    for (long base = 0; partial && base < quads; base += 4L * NT) {
        oneshot_f4 v[4];
        const void *src[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const long q = base + (long)i * NT + tid;
            src[i] = (const float4 *)partial + (q < quads ? q : quads - 1);
        }
        ld_sys16x4(v[0], v[1], v[2], v[3], src[0], src[1], src[2], src[3]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const long q = base + (long)i * NT + tid;
            if (q < quads)
                st_sys16(mine + slot_off + q * 16, make_float4(v[i].x, v[i].y, v[i].z, v[i].w));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int r_of = tid / ONESHOT_WGS, w_of = tid % ONESHOT_WGS;
    if (r_of < a.world)
        st_sys4(a.peer[r_of] + ((size_t)a.rank * ONESHOT_WGS + w_of) * 64, seq);
    if (r_of < a.world && !dead) {
        const uint8_t *f = mine + ((size_t)r_of * ONESHOT_WGS + w_of) * 64;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(ld_sys4(f) - seq) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)a.timeout_ticks) {
                atomicExch(a.state, 1 + r_of);
                break;
            }
        }
    }
    __syncthreads();
    for (long base = 0; base < quads; base += 4L * NT) {
        float4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const long q = base + (long)i * NT + tid;
            acc[i] = residual && q < quads ? ((const float4 *)residual)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int r = 0; r < a.world; r++) { // rank order; the four quads of a rank share one round trip
            oneshot_f4 v[4];
            const void *src[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const long q = base + (long)i * NT + tid;
                src[i] = a.peer[r] + slot_off + (q < quads ? q : quads - 1) * 16;
            }
            ld_sys16x4(v[0], v[1], v[2], v[3], src[0], src[1], src[2], src[3]);
#pragma unroll
            for (int i = 0; i < 4; i++)
                acc[i].x += v[i].x, acc[i].y += v[i].y, acc[i].z += v[i].z, acc[i].w += v[i].w;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const long q = base + (long)i * NT + tid;
            if (q < quads)
                ((float4 *)out)[q] = acc[i];
        }
    }
}
