// sgemm_host.cpp — libllamafile_sgemm.so: the reference's CPU mat-mul plug-in ABI
// (include/llamafile_sgemm.h), served by the MI355X HIP module through dlopen.
//
// Mirrors, on the host side:
//   llamafile/sgemm.cpp:26-145      the dispatcher (here: "is the HIP module loaded?" instead of CPUID)
//   llamafile/cuda.c:701-753        dlopen + symbol import of the GPU module
//   tinyblas_cpu_sgemm.inc:45-331   which (Atype, Btype, Ctype) combinations are serviced
//   tinyblas_cpu_mixmul.inc:77-398  llamafile_mixmul's tensor walking
// It contains NO arithmetic: every product is computed by libllamafile_amd_hip.so.  When the module
// or the GPU is missing every entry point answers `false` ("not serviced"), never a CPU result.
#include "../../include/llamafile_sgemm.h"
#include "../../include/lfamd_blocks.h"
#include "../../include/lfamd_hip.h"

#include <assert.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <list>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

// ---- the imported device ABI (one pointer per include/lfamd_hip.h function) ----
struct HipApi {
    void *dso = nullptr;
    decltype(&lfamd_abi_version) abi_version;
    decltype(&lfamd_last_error) last_error;
    decltype(&lfamd_device_count) device_count;
    decltype(&lfamd_init) init;
    decltype(&lfamd_malloc) malloc_;
    decltype(&lfamd_free) free_;
    decltype(&lfamd_memcpy_h2d) h2d;
    decltype(&lfamd_memcpy_d2h) d2h;
    decltype(&lfamd_stream_sync) sync;
    decltype(&lfamd_host_alloc) host_alloc;
    decltype(&lfamd_host_free) host_free;
    decltype(&lfamd_packed_size) packed_size;
    decltype(&lfamd_pack_weights) pack_weights;
    decltype(&lfamd_scaled_gemm_ok) scaled_ok;
    decltype(&lfamd_quantize_rows) quantize_rows;
    decltype(&lfamd_mul_mat_workspace) mul_mat_workspace;
    decltype(&lfamd_mul_mat) mul_mat;
    decltype(&lfamd_mul_mat_id_workspace) mul_mat_id_workspace;
    decltype(&lfamd_mul_mat_id) mul_mat_id;
};

struct DevBuf { // grow-only device scratch
    void *p = nullptr;
    size_t cap = 0;
};

// Where cached host bytes came from, re-checked on every cache hit: the mapping's identity (device, inode and file offset of
// the first byte, from /proc/self/maps; all zero for a registered range or anonymous memory) and a fingerprint of sampled
// bytes.  A model that was munmap()ed and whose address range now holds another file, another part of the same file or
// writable memory no longer matches and is packed again.
struct Origin {
    bool registered = false;
    uint64_t dev = 0, ino = 0, off = 0;
    uint64_t fp = 0;
    bool same(const Origin &o) const { return registered == o.registered && dev == o.dev && ino == o.ino && off == o.off && fp == o.fp; }
};

struct CachedWeights {
    int type;
    long rows, cols;
    size_t row_bytes;
    void *d_packed;
    size_t bytes;
    bool exact_only = false; // block scales outside the scaled-operand GEMM's range (lfamd_scaled_gemm_ok)
    std::list<const void *>::iterator lru;
    Origin origin;
};

// Which host bytes may be kept on the device across calls?  Only bytes the host cannot change behind our back:
//   * ranges the host registered with llamafile_sgemm_amd_register_weights (it promises they stay put), and
//   * addresses inside a mapping WITHOUT write permission (an mmap'd GGUF: llama.cpp maps model files PROT_READ),
//     read from /proc/self/maps.  The snapshot holds EVERY mapping with its permissions and file identity; it is read again
//     when an address is not covered by it or when it is older than MAPS_MAX_AGE_NS, so a writable address (ggml calls
//     llamafile_sgemm with the KV cache as `A` for KQ / KQV) is answered from the snapshot instead of re-parsing the file on
//     every call, and a read-only answer is never older than that age (the Origin check covers the window).
// Everything else is uploaded on every call.
struct Range {
    uintptr_t lo, hi;
    bool ro;
    uint64_t dev, ino, off;
};
constexpr int64_t MAPS_MAX_AGE_NS = 200 * 1000 * 1000;

struct State {
    std::once_flag once;
    bool ok = false;
    std::string error = "not initialised";
    HipApi api;
    std::mutex mu;
    std::unordered_map<const void *, CachedWeights> cache;
    std::list<const void *> lru; // front = most recently used
    size_t cache_bytes = 0, cache_budget = (size_t)200 << 30;
    std::map<uintptr_t, uintptr_t> registered; // lo -> hi
    std::vector<Range> ro_maps;                // every mapping of /proc/self/maps, sorted by address
    int64_t maps_time = 0;                     // CLOCK_MONOTONIC of the snapshot (0: none)
    unsigned long maps_reads = 0;              // how often the file was parsed (tests)
    DevBuf raw, b, c, ws, plan, x, a_scratch;
    std::vector<uint8_t> h_gather; // host staging reused across MoE calls
    std::vector<float> h_out;
    std::vector<int32_t> h_plan;
    unsigned flags = 0;
    std::atomic<int> precise{0};
} g;

template <typename T>
bool import(void *dso, const char *name, T &fn, std::string &err) {
    fn = (T)dlsym(dso, name);
    if (!fn) {
        err = std::string("missing symbol ") + name;
        return false;
    }
    return true;
}

std::string module_path() {
    if (const char *e = getenv("LFAMD_HIP_MODULE"))
        return e;
    Dl_info info;
    if (dladdr((void *)&module_path, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t slash = p.find_last_of('/');
        return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/libllamafile_amd_hip.so";
    }
    return "libllamafile_amd_hip.so";
}

void load_module() {
    std::string path = module_path();
    void *dso = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dso) {
        g.error = std::string("dlopen failed: ") + dlerror();
        return;
    }
    HipApi &a = g.api;
    a.dso = dso;
    std::string err;
    bool ok = import(dso, "lfamd_abi_version", a.abi_version, err) && import(dso, "lfamd_last_error", a.last_error, err) &&
              import(dso, "lfamd_device_count", a.device_count, err) && import(dso, "lfamd_init", a.init, err) &&
              import(dso, "lfamd_malloc", a.malloc_, err) && import(dso, "lfamd_free", a.free_, err) &&
              import(dso, "lfamd_memcpy_h2d", a.h2d, err) && import(dso, "lfamd_memcpy_d2h", a.d2h, err) &&
              import(dso, "lfamd_stream_sync", a.sync, err) && import(dso, "lfamd_host_alloc", a.host_alloc, err) &&
              import(dso, "lfamd_host_free", a.host_free, err) && import(dso, "lfamd_packed_size", a.packed_size, err) &&
              import(dso, "lfamd_pack_weights", a.pack_weights, err) &&
              import(dso, "lfamd_scaled_gemm_ok", a.scaled_ok, err) &&
              import(dso, "lfamd_quantize_rows", a.quantize_rows, err) &&
              import(dso, "lfamd_mul_mat_workspace", a.mul_mat_workspace, err) &&
              import(dso, "lfamd_mul_mat", a.mul_mat, err) &&
              import(dso, "lfamd_mul_mat_id_workspace", a.mul_mat_id_workspace, err) &&
              import(dso, "lfamd_mul_mat_id", a.mul_mat_id, err);
    if (!ok) {
        g.error = err;
        return;
    }
    if (a.abi_version() != LFAMD_ABI_VERSION) {
        g.error = "HIP module ABI version mismatch";
        return;
    }
    if (a.device_count() <= 0) {
        g.error = "no HIP device";
        return;
    }
    int dev = 0;
    if (const char *e = getenv("LFAMD_DEVICE"))
        dev = atoi(e);
    if (a.init(dev) != LFAMD_OK) {
        g.error = std::string("lfamd_init: ") + a.last_error();
        return;
    }
    // which build of tinyBLAS_Q0 the reference would run on this host (sgemm.cpp:26-102)
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx512f"))
        g.flags |= LFAMD_FLAG_Q0_VREGS32;
#endif
    if (const char *e = getenv("LFAMD_Q80_EXACT")) // Q8_0 batches bit for bit like tinyBLAS_Q0 (default: library f16 GEMM, <= 1e-3; MFMA body, 2e-6, where hipBLASLt does not load)
        if (atoi(e))
            g.flags |= LFAMD_FLAG_Q80_EXACT;
    g.ok = true;
    g.error.clear();
}

bool available() {
    std::call_once(g.once, load_module);
    return g.ok;
}

// pinned + device-mapped staging (single-column calls: the kernel reads / writes it in place over PCIe)
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
};
bool reserve_host(HostBuf &b, size_t bytes);

bool reserve(DevBuf &b, size_t bytes) {
    if (bytes <= b.cap)
        return true;
    if (b.p)
        g.api.free_(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    if (g.api.malloc_(&b.p, want) != LFAMD_OK)
        return false;
    b.cap = want;
    return true;
}

HostBuf g_hb, g_hc;
bool reserve_host(HostBuf &b, size_t bytes) {
    if (bytes <= b.cap)
        return true;
    if (b.p)
        g.api.host_free(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (g.api.host_alloc(&b.p, want) != LFAMD_OK)
        return false;
    b.cap = want;
    return true;
}

unsigned flags_now() {
    return g.flags | (g.precise.load(std::memory_order_relaxed) ? LFAMD_FLAG_PRECISE : 0u);
}

int64_t now_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}

void read_ro_maps() {
    g.ro_maps.clear();
    g.maps_time = now_ns();
    g.maps_reads++;
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f)
        return;
    char line[512];
    while (fgets(line, sizeof line, f)) {
        unsigned long lo, hi, off, ino;
        unsigned maj, mnr;
        char perms[8];
        if (sscanf(line, "%lx-%lx %7s %lx %x:%x %lu", &lo, &hi, perms, &off, &maj, &mnr, &ino) == 7)
            g.ro_maps.push_back({(uintptr_t)lo, (uintptr_t)hi, perms[0] == 'r' && perms[1] == '-',
                                 ((uint64_t)maj << 32) | mnr, (uint64_t)ino, (uint64_t)off});
    }
    fclose(f);
}

// 1 = [lo, hi) lies in read-only mappings (o = identity of its first byte), 0 = part of it is writable or unmapped in the
// snapshot, -1 = the snapshot does not know the first address at all
int lookup_maps(uintptr_t lo, uintptr_t hi, Origin *o) {
    // adjacent read-only mappings of one file may be split: walk them
    uintptr_t at = lo;
    bool first = true;
    for (const Range &r : g.ro_maps) {
        if (r.lo <= at && at < r.hi) {
            if (!r.ro)
                return 0;
            if (first) {
                o->dev = r.dev, o->ino = r.ino, o->off = r.off + (at - r.lo);
                first = false;
            }
            at = r.hi;
            if (at >= hi)
                return 1;
        }
    }
    return first ? -1 : 0;
}

// sampled bytes of [p, p + bytes): 64 words spread over the range (first and last included)
uint64_t fingerprint(const void *p, size_t bytes) {
    uint64_t h = 0xcbf29ce484222325ull ^ bytes;
    if (bytes < 8) {
        for (size_t i = 0; i < bytes; i++)
            h = (h ^ ((const uint8_t *)p)[i]) * 0x100000001b3ull;
        return h;
    }
    const size_t last = bytes - 8, steps = 63;
    for (size_t i = 0; i <= steps; i++) {
        uint64_t w;
        memcpy(&w, (const uint8_t *)p + (size_t)((unsigned __int128)last * i / steps), 8);
        h = (h ^ w) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    return h;
}

// may [p, p + bytes) be cached on the device?  Fills *o for the answer "yes".  Caller holds g.mu.
bool is_immutable(const void *p, size_t bytes, Origin *o) {
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    *o = Origin{};
    auto it = g.registered.upper_bound(lo);
    if (it != g.registered.begin()) {
        --it;
        if (it->first <= lo && hi <= it->second) {
            o->registered = true;
            o->fp = fingerprint(p, bytes);
            return true;
        }
    }
    int r = -1;
    if (g.maps_time != 0 && now_ns() - g.maps_time <= MAPS_MAX_AGE_NS)
        r = lookup_maps(lo, hi, o);
    if (r < 0) { // no snapshot, an old one, or one that predates the mapping
        read_ro_maps();
        r = lookup_maps(lo, hi, o);
    }
    if (r != 1)
        return false;
    o->fp = fingerprint(p, bytes);
    return true;
}

void drop(std::unordered_map<const void *, CachedWeights>::iterator it) {
    g.api.free_(it->second.d_packed);
    g.cache_bytes -= it->second.bytes;
    g.lru.erase(it->second.lru);
    g.cache.erase(it);
}

// upload + pack `rows` x `cols` of `type` from host rows `row_bytes` apart into d_packed
bool upload_packed(int type, const void *A, long rows, long cols, size_t row_bytes, void *d_packed, bool *exact_only) {
    const size_t total = (size_t)rows * row_bytes;
    if (!reserve(g.raw, total) || g.api.h2d(g.raw.p, A, total, nullptr) != LFAMD_OK)
        return false;
    if (g.api.pack_weights(type, rows, cols, g.raw.p, row_bytes, d_packed, nullptr) != LFAMD_OK)
        return false;
    const int in_range = g.api.scaled_ok(type, rows, cols, d_packed, nullptr); // (synchronises the stream)
    if (in_range < 0)
        return false;
    *exact_only = in_range == 0;
    return true;
}

// Device copy of a weight matrix (rows x cols of `type`, host rows `row_bytes` apart).  Immutable host bytes are packed
// once and kept (LRU within a byte budget, LFAMD_CACHE_BYTES); anything else is packed into a scratch buffer per call.
struct DevWeights {
    const void *d_packed;
    bool exact_only;
};

bool get_weights(int type, const void *A, long rows, long cols, size_t row_bytes, DevWeights *out) {
    const size_t total = (size_t)rows * row_bytes;
    const size_t packed = g.api.packed_size(type, rows, cols);
    Origin origin;
    if (!is_immutable(A, total, &origin)) {
        auto stale = g.cache.find(A); // (a range that was unregistered or remapped writable since)
        if (stale != g.cache.end())
            drop(stale);
        bool eo = false;
        if (!reserve(g.a_scratch, packed) || !upload_packed(type, A, rows, cols, row_bytes, g.a_scratch.p, &eo))
            return false;
        *out = {g.a_scratch.p, eo};
        return true;
    }
    auto it = g.cache.find(A);
    if (it != g.cache.end()) {
        CachedWeights &w = it->second;
        if (w.type == type && w.rows == rows && w.cols == cols && w.row_bytes == row_bytes && w.origin.same(origin)) {
            g.lru.splice(g.lru.begin(), g.lru, w.lru);
            *out = {w.d_packed, w.exact_only};
            return true;
        }
        drop(it); // same address, another view of it — or other bytes behind it (remapped since)
    }
    while (!g.lru.empty() && g.cache_bytes + packed > g.cache_budget)
        drop(g.cache.find(g.lru.back()));
    CachedWeights w{type, rows, cols, row_bytes, nullptr, packed};
    w.origin = origin;
    while (g.api.malloc_(&w.d_packed, w.bytes) != LFAMD_OK) { // device full: evict and retry
        if (g.lru.empty()) { // nothing left to evict: serve this call from the per-call scratch image instead of failing it
            bool eo = false;
            if (!reserve(g.a_scratch, packed) || !upload_packed(type, A, rows, cols, row_bytes, g.a_scratch.p, &eo))
                return false;
            *out = {g.a_scratch.p, eo};
            return true;
        }
        drop(g.cache.find(g.lru.back()));
    }
    if (!upload_packed(type, A, rows, cols, row_bytes, w.d_packed, &w.exact_only)) {
        g.api.free_(w.d_packed);
        return false;
    }
    g.lru.push_front(A);
    w.lru = g.lru.begin();
    g.cache_bytes += w.bytes;
    const CachedWeights &kept = (g.cache[A] = w);
    *out = {kept.d_packed, kept.exact_only};
    return true;
}

bool is_quant(int t) {
    return lfamd_blck_size(t) > 1;
}

// Which requests llamafile_sgemm services — same table as the reference's x86 builds
// (tinyblas_cpu_sgemm.inc:45-240 + the iqk pre-dispatch :286-304).
bool sgemm_supported(long n, int Atype, int Btype, int Ctype) {
    if (Ctype != LFAMD_TYPE_F32)
        return false;
    switch (Atype) {
    case LFAMD_TYPE_F32:
        return Btype == LFAMD_TYPE_F32;
    case LFAMD_TYPE_F16:
    case LFAMD_TYPE_BF16:
        if (Btype == LFAMD_TYPE_F32)
            return n <= 2; // else WANT_QUANTIZATION (:70-76, :123-129)
        return Btype == Atype;
    case LFAMD_TYPE_Q8_0:
    case LFAMD_TYPE_Q4_0:
    case LFAMD_TYPE_Q5_0:
    case LFAMD_TYPE_Q4_1:
    case LFAMD_TYPE_Q5_1:
    case LFAMD_TYPE_Q2_K:
    case LFAMD_TYPE_Q3_K:
    case LFAMD_TYPE_Q4_K:
    case LFAMD_TYPE_Q5_K:
    case LFAMD_TYPE_Q6_K:
    case LFAMD_TYPE_IQ4_XS:
        return Btype == lfamd_vec_dot_type(Atype);
    default:
        return false;
    }
}

// C[j*ldc + i] for j < n from host A, B -> host C.  Caller holds g.mu.
bool run_mul_mat(int Atype, const void *A, long m, long kelems, size_t a_row_bytes, int Btype, const void *B,
                 size_t b_row_bytes, long n, float *C, long ldc) {
    if (m == 0 || n == 0)
        return true;
    DevWeights w;
    if (!get_weights(Atype, A, m, kelems, a_row_bytes, &w))
        return false;
    // C spans (n-1)*ldc + m floats (the last column is not padded to ldc)
    size_t bbytes = (size_t)n * b_row_bytes, cbytes = ((size_t)(n - 1) * (size_t)ldc + (size_t)m) * 4;
    size_t wsb = g.api.mul_mat_workspace(Atype, m, kelems, n);
    if (!reserve(g.ws, wsb))
        return false;
    // One column (decode): no DMA.  The activation row goes into a pinned, device-mapped buffer with a CPU copy (16 KiB), the
    // GEMV reads it and writes the result row in place over PCIe, one synchronisation, a CPU copy out: two transfer set-ups
    // (~10 us each) less per call.  (Batches keep the DMA path: their kernels may add into C with float atomics.)
    static const bool zero_copy = getenv("LFAMD_HOST_NO_ZERO_COPY") == nullptr;
    if (n == 1 && zero_copy && reserve_host(g_hb, bbytes) && reserve_host(g_hc, cbytes)) {
        memcpy(g_hb.p, B, bbytes);
        if (g.api.mul_mat(Atype, w.d_packed, m, kelems, Btype, g_hb.p, b_row_bytes, n, (float *)g_hc.p, ldc, g.ws.p, g.ws.cap,
                          flags_now() | (w.exact_only ? LFAMD_FLAG_PRECISE : 0u), nullptr) != LFAMD_OK)
            return false;
        if (g.api.sync(nullptr) != LFAMD_OK)
            return false;
        memcpy(C, g_hc.p, cbytes); // (n = 1: exactly the m floats of the column)
        return true;
    }
    if (!reserve(g.b, bbytes) || !reserve(g.c, cbytes))
        return false;
    if (g.api.h2d(g.b.p, B, bbytes, nullptr) != LFAMD_OK)
        return false;
    if (ldc != m) // keep the caller's bytes in the gaps of C
        if (g.api.h2d(g.c.p, C, cbytes, nullptr) != LFAMD_OK)
            return false;
    if (g.api.mul_mat(Atype, w.d_packed, m, kelems, Btype, g.b.p, b_row_bytes, n, (float *)g.c.p, ldc, g.ws.p, g.ws.cap,
                      flags_now() | (w.exact_only ? LFAMD_FLAG_PRECISE : 0u), nullptr) != LFAMD_OK)
        return false;
    if (g.api.d2h(C, g.c.p, cbytes, nullptr) != LFAMD_OK)
        return false;
    return g.api.sync(nullptr) == LFAMD_OK;
}

[[noreturn]] void die(const char *what) {
    fprintf(stderr, "llamafile_sgemm (MI355X): %s: %s\n", what, g.ok ? g.api.last_error() : g.error.c_str());
    abort();
}

} // namespace

extern "C" {

int llamafile_sgemm_amd_available(void) {
    return available() ? 1 : 0;
}

const char *llamafile_sgemm_amd_error(void) {
    return g.error.c_str();
}

void llamafile_sgemm_amd_set_precise(int precise) {
    g.precise.store(precise, std::memory_order_relaxed);
}

void llamafile_sgemm_amd_register_weights(const void *p, size_t bytes) {
    if (!p || !bytes)
        return;
    std::lock_guard<std::mutex> lk(g.mu);
    g.registered[(uintptr_t)p] = (uintptr_t)p + bytes;
}

void llamafile_sgemm_amd_unregister_weights(const void *p) {
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g.registered.find((uintptr_t)p);
    if (it == g.registered.end())
        return;
    const uintptr_t lo = it->first, hi = it->second;
    g.registered.erase(it);
    if (!g.ok)
        return;
    for (auto c = g.cache.begin(); c != g.cache.end();) { // device copies of tensors inside the range go with it
        auto next = std::next(c);
        if ((uintptr_t)c->first >= lo && (uintptr_t)c->first < hi)
            drop(c);
        c = next;
    }
}

void llamafile_sgemm_amd_set_cache_budget(size_t bytes) {
    std::lock_guard<std::mutex> lk(g.mu);
    g.cache_budget = bytes;
}

size_t llamafile_sgemm_amd_cached_bytes(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    return g.cache_bytes;
}

void llamafile_sgemm_amd_forget(const void *A) {
    if (!available())
        return;
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g.cache.find(A);
    if (it != g.cache.end())
        drop(it);
}

void llamafile_sgemm_amd_reset(void) {
    if (!available())
        return;
    std::lock_guard<std::mutex> lk(g.mu);
    while (!g.cache.empty())
        drop(g.cache.begin());
    g.ro_maps.clear();
    g.maps_time = 0;
}

unsigned long llamafile_sgemm_amd_maps_reads(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    return g.maps_reads;
}

bool llamafile_sgemm(long m, long n, long k, const void *A, long lda, const void *B, long ldb, void *C, long ldc,
                     int ith, int nth, int Atype, int Btype, int Ctype) {
    assert(m >= 0);
    assert(n >= 0);
    assert(k >= 0);
    assert(lda >= k);
    assert(ldb >= k);
    assert(ldc >= m);
    assert(nth > 0);
    assert(ith < nth);
    if (!available() || !sgemm_supported(n, Atype, Btype, Ctype))
        return false; // same answer on every thread: a pure function of the arguments
    if (ith != 0)
        return true;
    std::lock_guard<std::mutex> lk(g.mu);
    const long kelems = k * lfamd_blck_size(Atype);
    const size_t a_row = (size_t)lda * lfamd_type_size(Atype), b_row = (size_t)ldb * lfamd_type_size(Btype);
    if (!run_mul_mat(Atype, A, m, kelems, a_row, Btype, B, b_row, n, (float *)C, ldc))
        die("device mat-mul failed after the request was accepted");
    return true;
}

bool iqk_mul_mat(long Nx, long Ny, long ne00, int typeA, const void *A, const void *B, float *C, long stride_C, int ith,
                 int nth) {
    if (!available() || !is_quant(typeA) || typeA == LFAMD_TYPE_Q8_0 || ne00 % lfamd_blck_size(typeA))
        return false; // x86 set_mul_mat has no Q8_0 case (iqk_mul_mat.inc:1408-1463)
    const int bt = lfamd_vec_dot_type(typeA);
    if (!sgemm_supported(Ny, typeA, bt, LFAMD_TYPE_F32))
        return false;
    if (ith != 0)
        return true;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!run_mul_mat(typeA, A, Nx, ne00, lfamd_row_size(typeA, ne00), bt, B, lfamd_row_size(bt, ne00), Ny, C, stride_C))
        die("device mat-mul failed after the request was accepted");
    return true;
}

bool iqk_mul_mat_moe(long Nx, long Ny, long ne00, int ne11, int typeA, const void *A, const void *B, float *C, long nb1,
                     long nb2, const void *vrow_mapping, int ith, int nth) {
    assert(vrow_mapping != nullptr);
    if (!available() || !is_quant(typeA) || typeA == LFAMD_TYPE_Q8_0 || ne00 % lfamd_blck_size(typeA))
        return false;
    const int bt = lfamd_vec_dot_type(typeA);
    if (!sgemm_supported(Ny, typeA, bt, LFAMD_TYPE_F32))
        return false;
    if (ith != 0)
        return true;
    std::lock_guard<std::mutex> lk(g.mu);
    const lfamd_mmid_row_mapping *map = (const lfamd_mmid_row_mapping *)vrow_mapping;
    const size_t brow = lfamd_row_size(bt, ne00);
    // gather the mapped activation rows (DataInfo::src1_row, iqk_mul_mat.inc:84-88)
    std::vector<uint8_t> &bg = g.h_gather; // (grow-only staging, reused across calls)
    bg.resize((size_t)Ny * brow);
    for (long iy = 0; iy < Ny; iy++) {
        size_t src = ((size_t)(map[iy].i1 % ne11) + (size_t)map[iy].i2 * ne11) * brow;
        memcpy(bg.data() + (size_t)iy * brow, (const uint8_t *)B + src, brow);
    }
    std::vector<float> &cg = g.h_out;
    cg.resize((size_t)Ny * (size_t)Nx);
    if (!run_mul_mat(typeA, A, Nx, ne00, lfamd_row_size(typeA, ne00), bt, bg.data(), brow, Ny, cg.data(), Nx))
        die("device mat-mul failed after the request was accepted");
    for (long iy = 0; iy < Ny; iy++) { // DataInfo::dst_row, iqk_mul_mat.inc:94-101
        float *dst = C + (size_t)map[iy].i1 * (nb1 / sizeof(float)) + (size_t)map[iy].i2 * (nb2 / sizeof(float));
        memcpy(dst, cg.data() + (size_t)iy * Nx, (size_t)Nx * sizeof(float));
    }
    return true;
}

bool llamafile_mixmul_iqk(long Nx, long Ny, long ne00, int ne11, int typeA, const void *A, const void *B, float *C,
                          long nb1, long nb2, const void *vrow_mapping, int ith, int nth) {
    return iqk_mul_mat_moe(Nx, Ny, ne00, ne11, typeA, A, B, C, nb1, nb2, vrow_mapping, ith, nth);
}

size_t llamafile_mixmul_needs(const struct ggml_tensor *, const struct ggml_tensor *, const struct ggml_tensor *) {
    return 0; // the device path needs none of the caller's params->wdata scratch
}

bool llamafile_mixmul(const struct ggml_compute_params *params, const struct ggml_tensor *weights,
                      const struct ggml_tensor *thought, const struct ggml_tensor *plan, struct ggml_tensor *result) {
    const long rows = weights->ne[1], cols = weights->ne[0], tokens = thought->ne[2];
    const int experts = (int)weights->ne[2], thinkers = (int)plan->ne[0], tasks = (int)thought->ne[1];
    // invariants asserted by the reference (tinyblas_cpu_mixmul.inc:112-133)
    assert(tasks <= thinkers);
    assert(thinkers <= experts);
    assert(tokens == plan->ne[1]);
    assert(rows == result->ne[0]);
    assert(cols == thought->ne[0]);
    assert(tokens == result->ne[2]);
    assert(thinkers == result->ne[1]);
    assert(params->nth > 0 && params->ith < params->nth);
    const int wt = weights->type;
    if (!available() || !is_quant(wt) || plan->type != LFAMD_TYPE_I32 || result->type != LFAMD_TYPE_F32)
        return false;
    const int bt = lfamd_vec_dot_type(wt);
    if (bt < 0 || cols % lfamd_blck_size(wt))
        return false;
    if (thought->type != LFAMD_TYPE_F32 && thought->type != bt)
        return false;
    // no column strides (:140-146)
    if (weights->nb[0] != lfamd_type_size(wt) || thought->nb[0] != lfamd_type_size(thought->type) ||
        result->nb[0] != sizeof(float) || weights->nb[1] % lfamd_type_size(wt))
        return false;
    if (params->ith != 0)
        return true;

    std::lock_guard<std::mutex> lk(g.mu);
    const size_t packed = g.api.packed_size(wt, rows, cols);
    // experts packed back to back in one device allocation; kept across calls only when the host bytes are immutable
    // (registered, or a read-only mapping), otherwise packed into scratch on every call — like get_weights
    DevWeights w{nullptr, false};
    {
        const size_t span = (size_t)(experts - 1) * weights->nb[2] + (size_t)rows * weights->nb[1];
        Origin origin;
        const bool keep = is_immutable(weights->data, span, &origin);
        auto it = g.cache.find(weights->data);
        if (it != g.cache.end() && !(keep && it->second.type == wt && it->second.rows == rows * experts && it->second.cols == cols &&
                                     it->second.row_bytes == weights->nb[1] && it->second.origin.same(origin))) {
            drop(it);
            it = g.cache.end();
        }
        if (it != g.cache.end()) {
            g.lru.splice(g.lru.begin(), g.lru, it->second.lru);
            w = {it->second.d_packed, it->second.exact_only};
        } else {
            void *dst = nullptr;
            const size_t bytes = packed * experts;
            if (keep) {
                while (!g.lru.empty() && g.cache_bytes + bytes > g.cache_budget)
                    drop(g.cache.find(g.lru.back()));
                while (g.api.malloc_(&dst, bytes) != LFAMD_OK) {
                    if (g.lru.empty())
                        die("device allocation for expert weights failed");
                    drop(g.cache.find(g.lru.back()));
                }
            } else {
                if (!reserve(g.a_scratch, bytes))
                    die("device allocation for expert weights failed");
                dst = g.a_scratch.p;
            }
            for (int e = 0; e < experts; e++) {
                size_t ebytes = (size_t)rows * weights->nb[1];
                if (!reserve(g.raw, ebytes) ||
                    g.api.h2d(g.raw.p, (const uint8_t *)weights->data + (size_t)e * weights->nb[2], ebytes, nullptr) ||
                    g.api.pack_weights(wt, rows, cols, g.raw.p, weights->nb[1], (uint8_t *)dst + (size_t)e * packed, nullptr) ||
                    g.api.sync(nullptr))
                    die("expert weight upload failed");
            }
            const int in_range = g.api.scaled_ok(wt, (long)experts * ((rows + 31) / 32) * 32, cols, dst, nullptr);
            if (in_range < 0)
                die("expert weight range check failed");
            w = {dst, in_range == 0};
            if (keep) {
                CachedWeights nw{wt, rows * experts, cols, weights->nb[1], dst, bytes, in_range == 0};
                nw.origin = origin;
                g.lru.push_front(weights->data);
                nw.lru = g.lru.begin();
                g.cache_bytes += bytes;
                g.cache[weights->data] = nw;
            }
        }
    }
    // activations: contiguous [tokens][tasks] rows on the device, quantised there if given as f32
    const size_t brow = lfamd_row_size(bt, cols);
    const size_t nrows = (size_t)tokens * tasks;
    if (!reserve(g.b, nrows * brow))
        die("device allocation failed");
    if (thought->type == LFAMD_TYPE_F32) {
        if (!reserve(g.x, nrows * (size_t)cols * 4))
            die("device allocation failed");
        for (long t = 0; t < tokens; t++)
            for (int k = 0; k < tasks; k++)
                if (g.api.h2d((uint8_t *)g.x.p + ((size_t)t * tasks + k) * cols * 4,
                              (const uint8_t *)thought->data + (size_t)t * thought->nb[2] + (size_t)k * thought->nb[1],
                              (size_t)cols * 4, nullptr))
                    die("activation upload failed");
        if (g.api.quantize_rows(bt, (const float *)g.x.p, (long)nrows, cols, (size_t)cols * 4, g.b.p, brow, nullptr))
            die("activation quantisation failed");
    } else {
        for (long t = 0; t < tokens; t++)
            for (int k = 0; k < tasks; k++)
                if (g.api.h2d((uint8_t *)g.b.p + ((size_t)t * tasks + k) * brow,
                              (const uint8_t *)thought->data + (size_t)t * thought->nb[2] + (size_t)k * thought->nb[1], brow,
                              nullptr))
                    die("activation upload failed");
    }
    // routing table, contiguous [tokens][thinkers]
    std::vector<int32_t> &hplan = g.h_plan; // (grow-only host staging, reused across calls)
    hplan.resize((size_t)tokens * thinkers);
    for (long t = 0; t < tokens; t++)
        for (int th = 0; th < thinkers; th++)
            hplan[(size_t)t * thinkers + th] =
                *(const int32_t *)((const uint8_t *)plan->data + (size_t)t * plan->nb[1] + (size_t)th * plan->nb[0]);
    size_t rbytes = (size_t)tokens * thinkers * rows * 4;
    size_t wsb = g.api.mul_mat_id_workspace(wt, rows, cols, experts, tokens, thinkers);
    if (!reserve(g.plan, hplan.size() * 4) || !reserve(g.c, rbytes) || !reserve(g.ws, wsb))
        die("device allocation failed");
    if (g.api.h2d(g.plan.p, hplan.data(), hplan.size() * 4, nullptr) || g.api.sync(nullptr))
        die("plan upload failed");
    if (g.api.mul_mat_id(wt, w.d_packed, rows, cols, experts, bt, g.b.p, brow, tasks, tokens, (const int32_t *)g.plan.p,
                         thinkers, (float *)g.c.p, g.ws.p, g.ws.cap, flags_now() | (w.exact_only ? LFAMD_FLAG_PRECISE : 0u), nullptr))
        die("device mul_mat_id failed");
    std::vector<float> &hres = g.h_out;
    hres.resize((size_t)tokens * thinkers * rows);
    if (g.api.d2h(hres.data(), g.c.p, rbytes, nullptr) || g.api.sync(nullptr))
        die("result download failed");
    for (long t = 0; t < tokens; t++)
        for (int th = 0; th < thinkers; th++)
            memcpy((uint8_t *)result->data + (size_t)t * result->nb[2] + (size_t)th * result->nb[1],
                   hres.data() + ((size_t)t * thinkers + th) * rows, (size_t)rows * 4);
    return true;
}
}
