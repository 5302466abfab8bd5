// comm.hip — the exchange step of the tensor-parallel path behind the C ABI (include/lfamd_hip.h, "collectives").
//
// The reference has no collective: its multi-GPU mode splits weight ROWS over the devices and gathers every result on a
// main GPU with peer copies (ggml_cuda_op_mul_mat, ggml-cuda.cu.patch:17853-18153; peer copies :17781-17851,
// 18077-18121).  Here attn_output / ffn_down are split by input columns and the f32 partial sums of
// the residual stream are ALL-REDUCED, one process per GPU:
//   * RCCL (ncclAllReduce / ncclAllGather over xGMI) for any size — the 8 MB prefill tensors; librccl is dlopen()ed
//     when the first communicator is made, so a single-GPU host needs no RCCL at all;
//   * a ONE-SHOT peer kernel for decode-sized messages (16-32 KB): every rank publishes its partial in an IPC-shared
//     slot (write-through stores), raises a flag in every peer's flag block, waits for the peers' flags and sums the
//     world's partials straight out of the peers' memory in RANK ORDER (deterministic, identical on every rank),
//     adding the residual in the same pass.  One launch, no ring: xGMI is point-to-point, a 16 KB ring all-reduce is
//     14 latency-bound hops at world 8.  Every spin is bounded (a lost peer sets an error flag instead of hanging).
#include "lfamd_device.h"
#include "../../include/lfamd_hip.h"
#include "oneshot_impl.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

extern "C" void lfamd_set_error(const char *msg);
hipError_t lfamd_launch_add_f32(float *y, const float *r, long n, hipStream_t s);
namespace {

// ---- the few RCCL entry points, by name (rccl.h's types restated as opaque handles; 128-byte unique id)
typedef struct ncclComm *ncclComm_t;
struct rccl_api {
    void *dso = nullptr;
    int (*GetUniqueId)(void *);
    int (*CommInitRank)(ncclComm_t *, int, const void *, int); // (the id is passed BY VALUE in C: see init_rank)
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
    int (*CommDestroy)(ncclComm_t);
    const char *(*GetErrorString)(int);
    int (*CommInitAll)(ncclComm_t *, int, const int *); // optional (single-process form)
} R;
struct unique_id {
    char internal[128];
};
typedef int (*init_rank_fn)(ncclComm_t *, int, unique_id, int);
enum { NCCL_FLOAT32 = 7, NCCL_INT8 = 0, NCCL_SUM = 0 };

bool load_rccl() {
    if (R.dso)
        return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!h) {
        lfamd_set_error("librccl not found (dlopen)");
        return false;
    }
#define SYM(field, name)                                                                                               \
    *(void **)&R.field = dlsym(h, name);                                                                               \
    if (!R.field) {                                                                                                    \
        lfamd_set_error("librccl: missing symbol " name);                                                              \
        return false;                                                                                                  \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    *(void **)&R.CommInitAll = dlsym(h, "ncclCommInitAll");
    R.dso = h;
    return true;
}

int nccl_fail(int rc, const char *where) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", where, R.GetErrorString ? R.GetErrorString(rc) : "RCCL error");
    lfamd_set_error(buf);
    return LFAMD_ERR_HIP;
}

} // namespace

#include "oneshot_impl.h"

// exchange blocks handed out by lfamd_oneshot_alloc (fine-grained / uncached device memory): attach accepts no other
static std::mutex g_blocks_mu;
static std::unordered_map<void *, size_t> g_blocks;

struct lfamd_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1;
    // one-shot state
    uint8_t *local = nullptr;               // this rank's exchange block (IPC-exported by the host)
    uint8_t *peer[ONESHOT_MAX_WORLD] = {};  // every rank's block in this process' address space (peer[rank] == local)
    size_t slot_bytes = 0;                  // one message slot (two slots: consecutive calls alternate)
    int *d_state = nullptr;                 // [0] error flag, [1 .. ONESHOT_WGS] per-work-group call counters
    long timeout_ticks = 400000000;         // 4 s of the 100 MHz wall clock (LFAMD_ONESHOT_TIMEOUT_S)
    int device = -1;                        // >= 0: made by lfamd_comm_init_all (one process, several devices): the HIP device this
                                            // rank's calls run on, the exchange block is the communicator's own, peers are plain pointers
};

// the single-process form launches each rank's kernels on that rank's device, whatever device the host thread has current
struct comm_device {
    int prev = -1;
    explicit comm_device(const lfamd_comm *c) {
        if (c->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != c->device)
            (void)hipSetDevice(c->device);
        else
            prev = -1;
    }
    ~comm_device() {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// out[i] = (residual ? residual[i] : 0) + sum over ranks r = 0 .. world-1 of partial_r[i]     (count % 4 == 0)
// GATHER: out[r * count + i] = partial_r[i] instead (the vocabulary shards of the logits; same publish / flag protocol)
template <bool GATHER>
__global__ __launch_bounds__(ONESHOT_THREADS) void oneshot_allreduce_kernel(const oneshot_args a, const float *__restrict__ partial,
                                                                            const float *__restrict__ residual,
                                                                            float *__restrict__ out) {
    const int w = blockIdx.x;
    const long quads = a.count / 4, per = (quads + ONESHOT_WGS - 1) / ONESHOT_WGS;
    const long q0 = (long)w * per, q1 = q0 + per < quads ? q0 + per : quads;
    uint8_t *mine = a.peer[a.rank];
    __shared__ uint32_t s_seq;
    __shared__ int s_dead;
    if (threadIdx.x == 0) {
        s_seq = (uint32_t)++a.state[1 + w];
        s_dead = a.state[0]; // a peer was lost earlier: no more waiting (results are void, the host sees lfamd_comm_check)
    }
    __syncthreads();
    const uint32_t seq = s_seq;
    const size_t slot_off = ONESHOT_FLAGS_BYTES + (size_t)(seq & 1) * a.slot_bytes;
    // 1. publish this rank's chunk: write-through, drained, then the flag in EVERY rank's block (its own included)
    for (long q = q0 + threadIdx.x; q < q1; q += ONESHOT_THREADS)
        st_sys16(mine + slot_off + q * 16, ((const float4 *)partial)[q]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x < a.world)
        st_sys4(a.peer[threadIdx.x] + ((size_t)a.rank * ONESHOT_WGS + w) * 64, seq);
    // 2. wait for chunk w of every rank.  Bounded by WALL time (s_memrealtime, 100 MHz): a rank may be late by seconds
    // (first-touch page-in, a slow host thread) without being lost; only after ONESHOT_TIMEOUT_S is the peer declared lost,
    // the error latched (later calls do not wait again: results are void until lfamd_comm_clear_error) and reported by
    // lfamd_comm_check.
    if ((int)threadIdx.x < a.world && !s_dead) {
        const uint8_t *f = mine + ((size_t)threadIdx.x * ONESHOT_WGS + w) * 64;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(ld_sys4(f) - seq) < 0) { // (sequence numbers wrap: compare as a signed difference)
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)a.timeout_ticks) {
                atomicExch(a.state, 1 + (int)threadIdx.x);
                break;
            }
        }
    }
    __syncthreads();
    // 3. sum in rank order out of the ranks' slots
    for (long q = q0 + threadIdx.x; q < q1; q += ONESHOT_THREADS) {
        if constexpr (GATHER) {
            for (int r = 0; r < a.world; r++)
                ((float4 *)out)[(long)r * quads + q] = ld_sys16(a.peer[r] + slot_off + q * 16);
        } else {
            float4 s = residual ? ((const float4 *)residual)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < a.world; r++) {
                const float4 v = ld_sys16(a.peer[r] + slot_off + q * 16);
                s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
            }
            ((float4 *)out)[q] = s;
        }
    }
}

extern "C" {

int lfamd_oneshot_alloc(void **d_block, size_t bytes);
int lfamd_oneshot_free(void *d_block);
int lfamd_comm_allreduce_add_f32(lfamd_comm *c, const float *d_partial, const float *d_residual, float *d_out, long count,
                                 void *stream);
size_t lfamd_oneshot_bytes(size_t max_message_bytes);
int lfamd_comm_destroy(lfamd_comm *c);

int lfamd_comm_unique_id(void *id128) {
    if (!load_rccl())
        return LFAMD_ERR_UNSUPPORTED;
    int rc = R.GetUniqueId(id128);
    return rc ? nccl_fail(rc, "ncclGetUniqueId") : LFAMD_OK;
}

int lfamd_comm_init(lfamd_comm **out, int rank, int world, const void *id128) {
    if (!out || world < 1 || rank < 0 || rank >= world) {
        lfamd_set_error("lfamd_comm_init: bad rank / world");
        return LFAMD_ERR_INVALID;
    }
    lfamd_comm *c = new lfamd_comm;
    c->rank = rank, c->world = world;
    if (id128) { // RCCL communicator (optional for world == 1 and for one-shot-only use)
        if (!load_rccl()) {
            delete c;
            return LFAMD_ERR_UNSUPPORTED;
        }
        unique_id id;
        memcpy(&id, id128, sizeof id);
        int rc = ((init_rank_fn)R.CommInitRank)(&c->nccl, world, id, rank);
        if (rc) {
            delete c;
            return nccl_fail(rc, "ncclCommInitRank");
        }
    }
    *out = c;
    return LFAMD_OK;
}

int lfamd_comm_destroy(lfamd_comm *c) {
    if (!c)
        return LFAMD_OK;
    comm_device on(c);
    for (int r = 0; r < c->world && r < ONESHOT_MAX_WORLD; r++)
        if (c->peer[r] && r != c->rank && c->device < 0)
            (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->device >= 0 && c->local)
        (void)lfamd_oneshot_free(c->local);
    if (c->d_state)
        (void)hipFree(c->d_state);
    if (c->nccl)
        R.CommDestroy(c->nccl);
    delete c;
    return LFAMD_OK;
}

size_t lfamd_oneshot_bytes(size_t max_message_bytes) {
    const size_t slot = (max_message_bytes + 255) / 256 * 256;
    return ONESHOT_FLAGS_BYTES + 2 * slot;
}

// The exchange block must be memory whose stores another GPU observes INSIDE a running kernel and whose loads are not
// served from a stale local cache: fine-grained (uncached) device memory.  Ordinary hipMalloc memory is coarse-grained —
// coherent only at kernel boundaries — and the protocol's write-through / cache-bypassing accesses alone are not a
// documented guarantee on it across xGMI.  LFAMD_ONESHOT_ANY_MEMORY=1 lifts the check in attach (single-GPU rehearsals
// with framework-owned buffers).
int lfamd_oneshot_alloc(void **d_block, size_t bytes) {
    if (!d_block || !bytes) {
        lfamd_set_error("lfamd_oneshot_alloc: bad arguments");
        return LFAMD_ERR_INVALID;
    }
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        lfamd_set_error(hipGetErrorString(e));
        return LFAMD_ERR_HIP;
    }
    {
        std::lock_guard<std::mutex> lk(g_blocks_mu);
        g_blocks[p] = bytes;
    }
    *d_block = p;
    return LFAMD_OK;
}

int lfamd_oneshot_free(void *d_block) {
    if (!d_block)
        return LFAMD_OK;
    {
        std::lock_guard<std::mutex> lk(g_blocks_mu);
        g_blocks.erase(d_block);
    }
    return hipFree(d_block) == hipSuccess ? LFAMD_OK : LFAMD_ERR_HIP;
}

int lfamd_oneshot_export(void *d_block, void *handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, d_block);
    if (e != hipSuccess) {
        lfamd_set_error(hipGetErrorString(e));
        return LFAMD_ERR_HIP;
    }
    memcpy(handle64, &h, 64);
    return LFAMD_OK;
}

int lfamd_oneshot_attach(lfamd_comm *c, void *d_local_block, size_t block_bytes, const void *handles, size_t max_message_bytes) {
    if (!c || c->world > ONESHOT_MAX_WORLD || block_bytes < lfamd_oneshot_bytes(max_message_bytes)) {
        lfamd_set_error("lfamd_oneshot_attach: bad communicator / block too small / world > 8");
        return LFAMD_ERR_INVALID;
    }
    {
        static const bool any_memory = getenv("LFAMD_ONESHOT_ANY_MEMORY") && atoi(getenv("LFAMD_ONESHOT_ANY_MEMORY"));
        std::lock_guard<std::mutex> lk(g_blocks_mu);
        auto it = g_blocks.find(d_local_block);
        if (!any_memory && (it == g_blocks.end() || it->second < block_bytes)) {
            lfamd_set_error("lfamd_oneshot_attach: the exchange block must come from lfamd_oneshot_alloc (fine-grained memory)");
            return LFAMD_ERR_INVALID;
        }
    }
    if (const char *t = getenv("LFAMD_ONESHOT_TIMEOUT_S")) {
        const double sec = atof(t);
        if (sec > 0.0 && sec < 3600.0)
            c->timeout_ticks = (long)(sec * 1e8);
    }
    hipError_t e = hipMemset(d_local_block, 0, ONESHOT_FLAGS_BYTES); // (the host barriers between this and the first all-reduce)
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess && !c->d_state)
        e = hipMalloc((void **)&c->d_state, (2 + ONESHOT_WGS) * sizeof(int));
    if (e == hipSuccess)
        e = hipMemset(c->d_state, 0, (2 + ONESHOT_WGS) * sizeof(int));
    for (int r = 0; r < c->world && e == hipSuccess; r++) {
        if (r == c->rank) {
            c->peer[r] = (uint8_t *)d_local_block;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const uint8_t *)handles + (size_t)r * 64, 64);
        void *p = nullptr;
        e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        c->peer[r] = (uint8_t *)p;
    }
    if (e != hipSuccess) {
        lfamd_set_error(hipGetErrorString(e));
        return LFAMD_ERR_HIP;
    }
    c->local = (uint8_t *)d_local_block;
    c->slot_bytes = (max_message_bytes + 255) / 256 * 256;
    return LFAMD_OK;
}

// d_out = (d_residual ? d_residual : 0) + sum over ranks of d_partial.  One-shot peer kernel when attached and the
// message fits a slot (count % 4 == 0, 16-byte aligned pointers), else RCCL all-reduce followed by the residual add.
int lfamd_comm_allreduce_add_f32(lfamd_comm *c, const float *d_partial, const float *d_residual, float *d_out, long count,
                                 void *stream) {
    if (!c || count < 0) {
        lfamd_set_error("lfamd_comm_allreduce_add_f32: bad arguments");
        return LFAMD_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    comm_device on(c);
    const bool aligned = (count & 3) == 0 && ((((uintptr_t)d_partial) | ((uintptr_t)d_out) | ((uintptr_t)d_residual)) & 15) == 0;
    if (c->local && aligned && (size_t)count * 4 <= c->slot_bytes) {
        oneshot_args a;
        for (int r = 0; r < ONESHOT_MAX_WORLD; r++)
            a.peer[r] = c->peer[r < c->world ? r : 0];
        a.rank = c->rank, a.world = c->world;
        a.slot_bytes = c->slot_bytes;
        a.count = count;
        a.timeout_ticks = c->timeout_ticks;
        a.state = c->d_state;
        oneshot_allreduce_kernel<false><<<ONESHOT_WGS, ONESHOT_THREADS, 0, s>>>(a, d_partial, d_residual, d_out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            lfamd_set_error(hipGetErrorString(e));
            return LFAMD_ERR_HIP;
        }
        return LFAMD_OK;
    }
    if (c->world == 1 && !c->nccl) {
        if (d_out != d_partial) {
            hipError_t e = hipMemcpyAsync(d_out, d_partial, (size_t)count * 4, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) {
                lfamd_set_error(hipGetErrorString(e));
                return LFAMD_ERR_HIP;
            }
        }
    } else {
        if (!c->nccl) {
            lfamd_set_error("lfamd_comm_allreduce_add_f32: no RCCL communicator and the message does not fit the one-shot slot");
            return LFAMD_ERR_INVALID;
        }
        int rc = R.AllReduce(d_partial, d_out, (size_t)count, NCCL_FLOAT32, NCCL_SUM, c->nccl, s);
        if (rc)
            return nccl_fail(rc, "ncclAllReduce");
    }
    if (d_residual) {
        hipError_t e = lfamd_launch_add_f32(d_out, d_residual, count, s);
        if (e != hipSuccess) {
            lfamd_set_error(hipGetErrorString(e));
            return LFAMD_ERR_HIP;
        }
    }
    return LFAMD_OK;
}

int lfamd_comm_allreduce_sum_f32(lfamd_comm *c, float *d_inout, long count, void *stream) {
    return lfamd_comm_allreduce_add_f32(c, d_inout, nullptr, d_inout, count, stream);
}

// every rank's `bytes_per_rank` bytes, in rank order, into d_recv (world * bytes_per_rank): the vocabulary-row shards of
// the logits
int lfamd_comm_allgather(lfamd_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream) {
    if (!c) {
        lfamd_set_error("lfamd_comm_allgather: no communicator");
        return LFAMD_ERR_INVALID;
    }
    comm_device on(c);
    if (c->world == 1 && !c->nccl) {
        if (d_recv != d_send) {
            hipError_t e = hipMemcpyAsync(d_recv, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, (hipStream_t)stream);
            if (e != hipSuccess) {
                lfamd_set_error(hipGetErrorString(e));
                return LFAMD_ERR_HIP;
            }
        }
        return LFAMD_OK;
    }
    if (!c->nccl) { // (rehearsals without RCCL: the one-shot protocol, when the shard fits a slot)
        if (c->local && bytes_per_rank % 16 == 0 && bytes_per_rank <= c->slot_bytes &&
            ((((uintptr_t)d_send) | ((uintptr_t)d_recv)) & 15) == 0) {
            oneshot_args a;
            for (int r = 0; r < ONESHOT_MAX_WORLD; r++)
                a.peer[r] = c->peer[r < c->world ? r : 0];
            a.rank = c->rank, a.world = c->world;
            a.slot_bytes = c->slot_bytes;
            a.count = (long)(bytes_per_rank / 4);
            a.timeout_ticks = c->timeout_ticks;
            a.state = c->d_state;
            oneshot_allreduce_kernel<true><<<ONESHOT_WGS, ONESHOT_THREADS, 0, (hipStream_t)stream>>>(a, (const float *)d_send, nullptr,
                                                                                                    (float *)d_recv);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                lfamd_set_error(hipGetErrorString(e));
                return LFAMD_ERR_HIP;
            }
            return LFAMD_OK;
        }
        lfamd_set_error("lfamd_comm_allgather: no RCCL communicator and the shard does not fit the one-shot slot");
        return LFAMD_ERR_INVALID;
    }
    int rc = R.AllGather(d_send, d_recv, bytes_per_rank, NCCL_INT8, c->nccl, (hipStream_t)stream);
    return rc ? nccl_fail(rc, "ncclAllGather") : LFAMD_OK;
}

// One process, several devices (one host thread, hipSetDevice per shard — ncclCommInitAll's shape): comms[i] is rank i of
// world ndev on HIP device devices[i].  Every rank's exchange block is allocated here (fine-grained memory on its device) and
// the peers hold each other's blocks as plain pointers (one address space: no IPC handles, no host barrier — the blocks are
// zeroed and visible before this returns).  RCCL communicators are added when the library has ncclCommInitAll and the devices
// are distinct; without them only messages that fit the one-shot slot are served.  A device may repeat (rehearsal on one GPU:
// the ranks' kernels then run side by side on it, so issue every rank's call before waiting for any).
int lfamd_comm_init_all(lfamd_comm **comms, int ndev, const int *devices, size_t max_message_bytes) {
    if (!comms || !devices || ndev < 1 || ndev > ONESHOT_MAX_WORLD || !max_message_bytes) {
        lfamd_set_error("lfamd_comm_init_all: 1 .. 8 devices, a list and a message size are needed");
        return LFAMD_ERR_INVALID;
    }
    int visible = 0, prev = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        lfamd_set_error("lfamd_comm_init_all: no HIP device");
        return LFAMD_ERR_HIP;
    }
    bool distinct = true;
    for (int i = 0; i < ndev; i++) {
        if (devices[i] < 0 || devices[i] >= visible) {
            lfamd_set_error("lfamd_comm_init_all: device ordinal out of range");
            return LFAMD_ERR_INVALID;
        }
        for (int j = 0; j < i; j++)
            distinct = distinct && devices[j] != devices[i];
    }
    (void)hipGetDevice(&prev);
    const size_t block = lfamd_oneshot_bytes(max_message_bytes);
    hipError_t e = hipSuccess;
    for (int i = 0; i < ndev; i++)
        comms[i] = nullptr;
    for (int i = 0; i < ndev && e == hipSuccess; i++) {
        lfamd_comm *c = new lfamd_comm;
        comms[i] = c;
        c->rank = i, c->world = ndev, c->device = devices[i];
        e = hipSetDevice(devices[i]);
        void *p = nullptr;
        if (e == hipSuccess && lfamd_oneshot_alloc(&p, block) != LFAMD_OK)
            e = hipErrorOutOfMemory;
        c->local = (uint8_t *)p;
        if (e == hipSuccess)
            e = hipMemset(p, 0, ONESHOT_FLAGS_BYTES);
        if (e == hipSuccess)
            e = hipMalloc((void **)&c->d_state, (2 + ONESHOT_WGS) * sizeof(int));
        if (e == hipSuccess)
            e = hipMemset(c->d_state, 0, (2 + ONESHOT_WGS) * sizeof(int));
        if (e == hipSuccess)
            e = hipDeviceSynchronize();
        for (int j = 0; j < ndev && e == hipSuccess; j++) // peers' blocks are read and written from this device's kernels
            if (devices[j] != devices[i]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) {
                    lfamd_set_error("lfamd_comm_init_all: the devices cannot access each other's memory");
                    e = hipErrorPeerAccessUnsupported;
                } else {
                    const hipError_t pe = hipDeviceEnablePeerAccess(devices[j], 0);
                    (void)hipGetLastError();
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { // (anything else would fault later, inside the kernel)
                        lfamd_set_error("lfamd_comm_init_all: hipDeviceEnablePeerAccess failed");
                        e = pe;
                    }
                }
            }
        c->slot_bytes = (max_message_bytes + 255) / 256 * 256;
        if (const char *t = getenv("LFAMD_ONESHOT_TIMEOUT_S")) {
            const double sec = atof(t);
            if (sec > 0.0 && sec < 3600.0)
                c->timeout_ticks = (long)(sec * 1e8);
        }
    }
    if (e == hipSuccess) {
        for (int i = 0; i < ndev; i++)
            for (int r = 0; r < ndev; r++)
                comms[i]->peer[r] = comms[r]->local;
        if (distinct && ndev > 1 && load_rccl() && R.CommInitAll) {
            ncclComm_t all[ONESHOT_MAX_WORLD] = {};
            if (R.CommInitAll(all, ndev, devices) == 0)
                for (int i = 0; i < ndev; i++)
                    comms[i]->nccl = all[i];
        }
    }
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (e != hipErrorPeerAccessUnsupported)
            lfamd_set_error(hipGetErrorString(e));
        for (int i = 0; i < ndev; i++) {
            if (comms[i])
                (void)lfamd_comm_destroy(comms[i]);
            comms[i] = nullptr;
        }
        return LFAMD_ERR_HIP;
    }
    return LFAMD_OK;
}

// forget a latched peer-lost error (after the host has dealt with it): the next one-shot calls wait for their peers again
int lfamd_comm_clear_error(lfamd_comm *c) {
    if (!c || !c->d_state)
        return LFAMD_OK;
    comm_device on(c); // (a single-process communicator lives on its own device)
    return hipMemset(c->d_state, 0, sizeof(int)) == hipSuccess ? LFAMD_OK : LFAMD_ERR_HIP;
}

// 0 = no one-shot all-reduce has timed out waiting for a peer; else 1 + the rank that never arrived (synchronises)
int lfamd_comm_check(lfamd_comm *c) {
    if (!c || !c->d_state)
        return 0;
    comm_device on(c);
    int v = 0;
    if (hipMemcpy(&v, c->d_state, sizeof v, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return v;
}
}

__global__ void add_f32_kernel(float *__restrict__ y, const float *__restrict__ r, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        y[i] += r[i];
}

hipError_t lfamd_launch_add_f32(float *y, const float *r, long n, hipStream_t s) {
    if (n > 0)
        add_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(y, r, n);
    return hipGetLastError();
}
