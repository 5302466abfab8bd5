// gguf_reader.cpp — minimal GGUF (v2 / v3) reader for the host library: the tensor directory and the bytes of a model
// file, so that BASELINE's configs can run on real files.
//
// Reference counterpart: gguf_init_from_file (upstream ggml.c; llamafile's changes to it are
// llama.cpp.patches/patches/ggml.c.patch:2503-2612 — it reads through `struct llamafile` instead of a FILE).  This is not
// that code: the file is mapped read-only once and every tensor is handed out as a pointer into the mapping, which is
// exactly what the weight cache of llamafile_sgemm treats as immutable (a mapping without write permission,
// include/llamafile_sgemm.h), so tensors read this way are uploaded and packed once.
// Format (public GGUF specification): "GGUF" | u32 version | u64 n_tensors | u64 n_kv | kv pairs | tensor infos
// {name, n_dims, dims[], type, offset} | padding to general.alignment (default 32) | tensor data.
#include "../../include/lfamd_blocks.h"
#include "../../include/llamafile_sgemm.h"

#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {

struct kv_entry {
    std::string key;
    uint32_t type;   // gguf value type
    uint64_t u = 0;  // integer / bool value
    double f = 0.0;  // float value
    std::string s;   // string value
    uint64_t arr_n = 0;
};

struct tensor_entry {
    std::string name;
    int type;
    int n_dims;
    int64_t ne[4];
    uint64_t offset; // from the start of the data section
    size_t nbytes;
};

} // namespace

struct lfamd_gguf {
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t size = 0;
    uint32_t version = 0;
    size_t alignment = 32, data_offset = 0;
    std::vector<kv_entry> kv;
    std::vector<tensor_entry> tensors;
};

namespace {

struct cursor {
    const uint8_t *p, *end;
    bool ok = true;
    template <typename T>
    T get() {
        T v{};
        if (!ok || (size_t)(end - p) < sizeof(T)) {
            ok = false;
            return v;
        }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (!ok || n > (uint64_t)(end - p)) {
            ok = false;
            return {};
        }
        std::string s((const char *)p, (size_t)n);
        p += n;
        return s;
    }
    void skip(uint64_t n) {
        if (!ok || n > (uint64_t)(end - p))
            ok = false;
        else
            p += n;
    }
};

size_t scalar_size(uint32_t t) {
    switch (t) {
    case 0: case 1: case 7: return 1;  // u8 i8 bool
    case 2: case 3: return 2;          // u16 i16
    case 4: case 5: case 6: return 4;  // u32 i32 f32
    case 10: case 11: case 12: return 8; // u64 i64 f64
    default: return 0;
    }
}

bool read_value(cursor &c, uint32_t t, kv_entry &e) {
    switch (t) {
    case 0: e.u = c.get<uint8_t>(); break;
    case 1: e.u = (uint64_t)(int64_t)c.get<int8_t>(); break;
    case 2: e.u = c.get<uint16_t>(); break;
    case 3: e.u = (uint64_t)(int64_t)c.get<int16_t>(); break;
    case 4: e.u = c.get<uint32_t>(); break;
    case 5: e.u = (uint64_t)(int64_t)c.get<int32_t>(); break;
    case 6: e.f = c.get<float>(); break;
    case 7: e.u = c.get<uint8_t>() != 0; break;
    case 8: e.s = c.str(); break;
    case 10: e.u = c.get<uint64_t>(); break;
    case 11: e.u = (uint64_t)c.get<int64_t>(); break;
    case 12: e.f = c.get<double>(); break;
    case 9: { // array: only its length is kept (tokenizer tables are of no use to the mat-mul path)
        const uint32_t at = c.get<uint32_t>();
        const uint64_t n = c.get<uint64_t>();
        e.arr_n = n;
        if (at == 8) {
            for (uint64_t i = 0; i < n && c.ok; i++)
                c.skip(c.get<uint64_t>());
        } else {
            const size_t es = scalar_size(at);
            if (!es || (n && n > (uint64_t)(c.end - c.p) / es))
                c.ok = false;
            else
                c.skip(n * es);
        }
        break;
    }
    default: c.ok = false;
    }
    return c.ok;
}

void set_err(char *err, size_t n, const char *msg) {
    if (err && n)
        snprintf(err, n, "%s", msg);
}

} // namespace

extern "C" {

lfamd_gguf *lfamd_gguf_open(const char *path, char *err, size_t errlen) {
    lfamd_gguf *g = new lfamd_gguf;
    g->fd = open(path, O_RDONLY | O_CLOEXEC);
    struct stat st;
    if (g->fd < 0 || fstat(g->fd, &st) != 0 || st.st_size < 24) {
        set_err(err, errlen, "cannot open file (or shorter than a GGUF header)");
        if (g->fd >= 0)
            close(g->fd);
        delete g;
        return nullptr;
    }
    g->size = (size_t)st.st_size;
    void *m = mmap(nullptr, g->size, PROT_READ, MAP_SHARED, g->fd, 0);
    if (m == MAP_FAILED) {
        set_err(err, errlen, "mmap failed");
        close(g->fd);
        delete g;
        return nullptr;
    }
    g->map = (const uint8_t *)m;
    cursor c{g->map, g->map + g->size};
    auto fail = [&](const char *msg) -> lfamd_gguf * {
        set_err(err, errlen, msg);
        munmap((void *)g->map, g->size);
        close(g->fd);
        delete g;
        return nullptr;
    };
    if (memcmp(c.p, "GGUF", 4) != 0)
        return fail("invalid magic characters");
    c.skip(4);
    g->version = c.get<uint32_t>();
    if (g->version < 2 || g->version > 3)
        return fail("unsupported GGUF version (2 and 3 are read; v1 is no longer supported)");
    const uint64_t nt = c.get<uint64_t>(), nkv = c.get<uint64_t>();
    if (!c.ok || nt > (1u << 24) || nkv > (1u << 24))
        return fail("failed to read header");
    for (uint64_t i = 0; i < nkv; i++) {
        kv_entry e;
        e.key = c.str();
        e.type = c.get<uint32_t>();
        if (!c.ok || !read_value(c, e.type, e))
            return fail("failed to read key-value pairs");
        if (e.key == "general.alignment" && e.u >= 1 && e.u <= (1u << 20))
            g->alignment = (size_t)e.u;
        g->kv.push_back(std::move(e));
    }
    for (uint64_t i = 0; i < nt; i++) {
        tensor_entry t;
        t.name = c.str();
        t.n_dims = (int)c.get<uint32_t>();
        if (!c.ok || t.n_dims < 1 || t.n_dims > 4)
            return fail("failed to read tensor info");
        int64_t ne = 1;
        for (int d = 0; d < 4; d++) {
            t.ne[d] = d < t.n_dims ? (int64_t)c.get<uint64_t>() : 1;
            if (t.ne[d] < 0 || (t.ne[d] && ne > INT64_MAX / (t.ne[d] ? t.ne[d] : 1)))
                return fail("tensor dimensions overflow");
            ne *= t.ne[d];
        }
        t.type = (int)c.get<uint32_t>();
        t.offset = c.get<uint64_t>();
        if (!c.ok)
            return fail("failed to read tensor info");
        const int blck = lfamd_blck_size(t.type);
        const size_t ts = lfamd_type_size(t.type);
        if (blck <= 0 || ts == 0) {
            t.nbytes = 0; // a type this module has no block format for: listed, not sized
        } else {
            if (t.ne[0] % blck)
                return fail("tensor row length is not a multiple of the type's block size");
            // ne / blck blocks of ts bytes each: checked, a crafted header must not wrap the byte count
            unsigned long long nbytes = 0;
            if (__builtin_mul_overflow((unsigned long long)(ne / blck), (unsigned long long)ts, &nbytes) || nbytes > (unsigned long long)SIZE_MAX)
                return fail("tensor byte size overflows");
            t.nbytes = (size_t)nbytes;
        }
        g->tensors.push_back(std::move(t));
    }
    const size_t pos = (size_t)(c.p - g->map);
    g->data_offset = (pos + g->alignment - 1) / g->alignment * g->alignment;
    if (g->data_offset > g->size)
        return fail("tensor data lies outside the file (or is misaligned)");
    // t.offset is an unvalidated u64 from the file: compare without forming sums that can wrap
    const uint64_t room = (uint64_t)(g->size - g->data_offset);
    for (const tensor_entry &t : g->tensors)
        if (t.offset % g->alignment || t.offset > room || (uint64_t)t.nbytes > room - t.offset)
            return fail("tensor data lies outside the file (or is misaligned)");
    return g;
}

void lfamd_gguf_close(lfamd_gguf *g) {
    if (!g)
        return;
    munmap((void *)g->map, g->size);
    close(g->fd);
    delete g;
}

int lfamd_gguf_version(const lfamd_gguf *g) { return (int)g->version; }
long lfamd_gguf_n_tensors(const lfamd_gguf *g) { return (long)g->tensors.size(); }
long lfamd_gguf_n_kv(const lfamd_gguf *g) { return (long)g->kv.size(); }
size_t lfamd_gguf_alignment(const lfamd_gguf *g) { return g->alignment; }

// tensor i: name, ggml type, dims (ne[0] = row length), pointer into the read-only mapping, bytes (0: unknown block format)
int lfamd_gguf_tensor(const lfamd_gguf *g, long i, const char **name, int *type, int *n_dims, int64_t ne[4], const void **data,
                      size_t *nbytes) {
    if (i < 0 || i >= (long)g->tensors.size())
        return -1;
    const tensor_entry &t = g->tensors[i];
    if (name) *name = t.name.c_str();
    if (type) *type = t.type;
    if (n_dims) *n_dims = t.n_dims;
    if (ne) memcpy(ne, t.ne, sizeof t.ne);
    if (data) *data = g->map + g->data_offset + t.offset;
    if (nbytes) *nbytes = t.nbytes;
    return 0;
}

long lfamd_gguf_find_tensor(const lfamd_gguf *g, const char *name) {
    for (size_t i = 0; i < g->tensors.size(); i++)
        if (g->tensors[i].name == name)
            return (long)i;
    return -1;
}

// metadata: 0 = found.  Integers (any width, bool) through _u64, floats through _f64, strings through _str.
static const kv_entry *find_kv(const lfamd_gguf *g, const char *key) {
    for (const kv_entry &e : g->kv)
        if (e.key == key)
            return &e;
    return nullptr;
}
int lfamd_gguf_get_u64(const lfamd_gguf *g, const char *key, uint64_t *v) {
    const kv_entry *e = find_kv(g, key);
    if (!e || e->type == 6 || e->type == 8 || e->type == 9 || e->type == 12)
        return -1;
    *v = e->u;
    return 0;
}
int lfamd_gguf_get_f64(const lfamd_gguf *g, const char *key, double *v) {
    const kv_entry *e = find_kv(g, key);
    if (!e || (e->type != 6 && e->type != 12))
        return -1;
    *v = e->f;
    return 0;
}
const char *lfamd_gguf_get_str(const lfamd_gguf *g, const char *key) {
    const kv_entry *e = find_kv(g, key);
    return e && e->type == 8 ? e->s.c_str() : nullptr;
}
}
