// ggml_backend_lfamd.hip — the 12 GGML_CALL (ms_abi) symbols llamafile/cuda.c:726-737 imports from ggml-rocm.so, served by
// this module: a ggml backend whose supports_op says yes to GGML_OP_MUL_MAT / GGML_OP_MUL_MAT_ID only.
//
// Reference counterparts: ggml_cuda_link (ggml-cuda.cu.patch:468), the buffer interface (:16890-17027, set_tensor :16971),
// the buffer type (:17040-17100, row padding :17072-17081), supports_op (:19221-19262), graph_compute (:18945),
// ggml_cuda_mul_mat / _mul_mat_id (:18377-18443, 18499-18635), reg_devices (:19594-19610), device queries (:19532-19575).
// Layouts: include/ggml_backend_lfamd.h.  Arithmetic: none here — every product goes through include/lfamd_hip.h.
//
// Weights: the host's set_tensor delivers raw GGUF rows into the buffer; the first mat-mul that uses a tensor of a
// WEIGHTS buffer packs it once into the module's layout (kept until the buffer is freed or the tensor rewritten); src0
// of any other buffer (the KV cache as KQ / KQV operand) is packed per call.
//
// Devices: one host process drives every gfx950 device it sees (LFAMD_BACKEND_DEVICES = comma list of HIP ordinals, or "all" =
// the default; an ordinal may repeat, which is how one GPU rehearses two).  Each logical device has its own buffer type and
// backend (llama.cpp --split-mode layer); ggml_backend_cuda_split_buffer_type is the ROW split of the reference
// (ggml-cuda.cu.patch:17123-17450): a matrix's rows are cut at the tensor_split fractions, each device keeps and packs its
// slice, and a MUL_MAT whose src0 lives there runs one product per device on that device's slice (the activations are copied
// over, every device's columns of the result are copied back into dst on the device of the backend that runs the graph).
#include "lfamd_device.h"
#include "../../include/ggml_backend_lfamd.h"
#include "../../include/lfamd_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

const ggml_backend_api *g_api = nullptr;
int g_op_mul_mat = -1, g_op_mul_mat_id = -1;
bool g_linked = false;

#define LFAMD_MAX_DEVS 16 // GGML_CUDA_MAX_DEVICES (ggml-cuda.h.patch: the length of llama.cpp's tensor_split array)
int g_ndev = 0;
int g_phys[LFAMD_MAX_DEVS]; // logical device -> HIP ordinal

// makes a logical device current for the scope (and puts the caller's device back: ggml's threads own no device state here)
struct on_device {
    int prev = 0;
    bool ok;
    explicit on_device(int logical) {
        (void)hipGetDevice(&prev);
        ok = logical >= 0 && logical < g_ndev && hipSetDevice(g_phys[logical]) == hipSuccess;
    }
    ~on_device() { (void)hipSetDevice(prev); }
};

void logf(const char *fmt, const char *a = "", const char *b = "") {
    if (g_api && g_api->FLAG_log_disable && *g_api->FLAG_log_disable)
        return;
    fprintf(stderr, fmt, a, b);
}

struct type_row {
    int id;
    const char *name;
};
// the types this module has kernels for, with upstream's names (ggml.c type_traits[].type_name)
const type_row k_types[] = {{LFAMD_TYPE_F32, "f32"},   {LFAMD_TYPE_F16, "f16"},   {LFAMD_TYPE_Q4_0, "q4_0"}, {LFAMD_TYPE_Q4_1, "q4_1"},
                            {LFAMD_TYPE_Q5_0, "q5_0"}, {LFAMD_TYPE_Q5_1, "q5_1"}, {LFAMD_TYPE_Q8_0, "q8_0"}, {LFAMD_TYPE_Q2_K, "q2_K"},
                            {LFAMD_TYPE_Q3_K, "q3_K"}, {LFAMD_TYPE_Q4_K, "q4_K"}, {LFAMD_TYPE_Q5_K, "q5_K"}, {LFAMD_TYPE_Q6_K, "q6_K"},
                            {LFAMD_TYPE_IQ4_XS, "iq4_xs"}, {LFAMD_TYPE_BF16, "bf16"}};

bool type_ok(int t) {
    for (const type_row &r : k_types)
        if (r.id == t)
            return true;
    return false;
}

// ---- packed copies of weight tensors
struct packed {
    void *d = nullptr;
    size_t bytes = 0;
    int type;
    long rows, cols;
    size_t row_bytes;
    bool exact_only = false;
};
std::mutex g_mu;
std::unordered_map<const void *, packed> g_packed; // key: device address of the raw slice

void drop_range(const uint8_t *lo, const uint8_t *hi) {
    for (auto it = g_packed.begin(); it != g_packed.end();) {
        const uint8_t *p = (const uint8_t *)it->first;
        if (p >= lo && p < hi) {
            (void)hipFree(it->second.d);
            it = g_packed.erase(it);
        } else {
            ++it;
        }
    }
}

struct buffer_ctx {
    void *base;
    size_t size;
    int dev; // logical device
};

// what a row-split product needs on a device other than the one that runs the graph
struct peer_scratch {
    void *x = nullptr; // the activations, dense f32 rows
    size_t x_cap = 0;
    void *c = nullptr; // this device's columns of the result, dense
    size_t c_cap = 0;
    void *ws = nullptr;
    size_t ws_cap = 0;
};

struct backend_ctx {
    int device;
    void *scratch = nullptr; // per-call packed src0 of non-weight buffers
    size_t scratch_cap = 0;
    void *ws = nullptr;
    size_t ws_cap = 0;
    void *plan = nullptr; // contiguous copy of a strided ids tensor
    size_t plan_cap = 0;
    peer_scratch peer[LFAMD_MAX_DEVS];
    long sibling_calls = 0; // calls that served more than one node (LFAMD_BACKEND_STATS=1 prints it when the backend is freed)
};

bool grow(void *&p, size_t &cap, size_t need) {
    if (need <= cap)
        return true;
    if (p)
        (void)hipFree(p);
    p = nullptr, cap = 0;
    if (hipMalloc(&p, need + need / 4 + 256) != hipSuccess)
        return false;
    cap = need + need / 4 + 256;
    return true;
}

// ------------------------------------------------------------------ buffer interface
GGML_CALL const char *buf_get_name(ggml_backend_buffer_t) {
    return "ROCm-lfamd";
}
GGML_CALL void buf_free(ggml_backend_buffer_t buffer) {
    buffer_ctx *c = (buffer_ctx *)buffer->context;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        drop_range((const uint8_t *)c->base, (const uint8_t *)c->base + c->size);
    }
    {
        on_device d(c->dev);
        (void)hipFree(c->base);
    }
    delete c;
}
GGML_CALL void *buf_get_base(ggml_backend_buffer_t buffer) {
    return ((buffer_ctx *)buffer->context)->base;
}
GGML_CALL void buf_init_tensor(ggml_backend_buffer_t, struct ggml_tensor *) {}
GGML_CALL void buf_set_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor *tensor, const void *data, size_t offset, size_t size) {
    on_device d(((buffer_ctx *)buffer->context)->dev);
    (void)hipMemcpy((uint8_t *)tensor->data + offset, data, size, hipMemcpyHostToDevice);
    std::lock_guard<std::mutex> lk(g_mu); // the tensor's packed copies are stale now
    drop_range((const uint8_t *)tensor->data, (const uint8_t *)tensor->data + g_api->ggml_nbytes(tensor));
}
GGML_CALL void buf_get_tensor(ggml_backend_buffer_t buffer, const struct ggml_tensor *tensor, void *data, size_t offset, size_t size) {
    on_device d(((buffer_ctx *)buffer->context)->dev);
    (void)hipMemcpy(data, (const uint8_t *)tensor->data + offset, size, hipMemcpyDeviceToHost);
}
GGML_CALL bool buf_cpy_tensor(ggml_backend_buffer_t buffer, const struct ggml_tensor *src, struct ggml_tensor *dst) {
    if (src->buffer && src->buffer->iface.get_name == buf_get_name) { // (either device's memory: unified addressing)
        on_device d(((buffer_ctx *)buffer->context)->dev);
        (void)hipMemcpy(dst->data, src->data, g_api->ggml_nbytes(src), hipMemcpyDefault);
        std::lock_guard<std::mutex> lk(g_mu);
        drop_range((const uint8_t *)dst->data, (const uint8_t *)dst->data + g_api->ggml_nbytes(dst));
        return true;
    }
    return false;
}
GGML_CALL void buf_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    buffer_ctx *c = (buffer_ctx *)buffer->context;
    on_device d(c->dev);
    (void)hipMemset(c->base, value, c->size);
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_mu); // every packed copy made from this buffer's bytes is stale now
    drop_range((const uint8_t *)c->base, (const uint8_t *)c->base + c->size);
}
const ggml_backend_buffer_i k_buffer_iface = {buf_get_name, buf_free, buf_get_base, buf_init_tensor, buf_set_tensor,
                                              buf_get_tensor, buf_cpy_tensor, buf_clear, nullptr};

// ------------------------------------------------------------------ buffer type
GGML_CALL const char *buft_get_name(ggml_backend_buffer_type_t) {
    return "ROCm-lfamd";
}
GGML_CALL ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    const int dev = (int)(intptr_t)buft->context;
    on_device d(dev);
    void *p = nullptr;
    if (!d.ok || hipMalloc(&p, size ? size : 256) != hipSuccess) {
        logf("%s: allocating a device buffer failed\n", "ggml_backend_lfamd");
        return nullptr;
    }
    buffer_ctx *c = new buffer_ctx{p, size, dev};
    return g_api->ggml_backend_buffer_init(buft, k_buffer_iface, c, size);
}
GGML_CALL size_t buft_alignment(ggml_backend_buffer_type_t) {
    return 128;
}
// the reference pads quantised tensors so that the last row can be read to a multiple of 512 elements
// (MATRIX_ROW_PADDING, ggml-cuda.cu.patch:17072-17081); kept, so that allocation sizes match what llama.cpp budgets
GGML_CALL size_t buft_alloc_size(ggml_backend_buffer_type_t, const struct ggml_tensor *tensor) {
    size_t size = g_api->ggml_nbytes(tensor);
    const int64_t ne0 = tensor->ne[0];
    if (g_api->ggml_is_quantized(tensor->type) && ne0 % 512 != 0)
        size += g_api->ggml_row_size(tensor->type, 512 - ne0 % 512);
    return size;
}
ggml_backend_buffer_type g_bufts[LFAMD_MAX_DEVS]; // [logical device], context = its number (filled by ggml_cuda_link)


// ------------------------------------------------------------------ row-split buffer type (--split-mode row)
// Reference: ggml_backend_cuda_split_buffer_* (ggml-cuda.cu.patch:17123-17450).  Tensor data pointers inside such a buffer are
// placeholders (the reference's 0x1000 base); the bytes live in one allocation per device, found through tensor->extra.
struct split_buft_ctx {
    float start[LFAMD_MAX_DEVS + 1]; // cumulative fractions: device d owns rows [start[d], start[d + 1]) x nrows
};
struct split_extra {
    void *d[LFAMD_MAX_DEVS] = {};
    long lo[LFAMD_MAX_DEVS + 1] = {}; // device d owns rows lo[d] .. lo[d + 1]
    size_t row_bytes = 0;
    size_t bytes[LFAMD_MAX_DEVS] = {};
};
struct split_buffer_ctx {
    std::vector<split_extra *> extras;
};
std::mutex g_split_mu;
std::vector<ggml_backend_buffer_type *> g_split_bufts; // one per distinct tensor_split (kept for the process' life, like the reference's map)

// Matrices are cut by rows in multiples of 256 (whole row tiles of every packed layout); a tensor with more than one slice in
// dims 2 / 3 (an expert stack) stays whole on the first device: MUL_MAT_ID routes per token, not per row range (the reference
// refuses MUL_MAT_ID on split buffers outright, ggml-cuda.cu.patch:18501).
void split_rows(const split_buft_ctx *sc, const struct ggml_tensor *t, long lo[LFAMD_MAX_DEVS + 1]) {
    const long rows = (long)g_api->ggml_nrows(t);
    const bool whole = t->ne[2] * t->ne[3] != 1;
    lo[0] = 0;
    for (int d = 1; d <= LFAMD_MAX_DEVS; d++) {
        long r = d >= g_ndev || whole ? rows : (long)((double)sc->start[d] * (double)rows) / 256 * 256;
        if (r > rows)
            r = rows;
        lo[d] = r < lo[d - 1] ? lo[d - 1] : r;
    }
}

GGML_CALL const char *split_buf_get_name(ggml_backend_buffer_t) {
    return "ROCm-lfamd_Split";
}
void split_free_extra(split_extra *e) {
    for (int d = 0; d < g_ndev; d++)
        if (e->d[d]) {
            drop_range((const uint8_t *)e->d[d], (const uint8_t *)e->d[d] + e->bytes[d]);
            on_device g(d);
            (void)hipFree(e->d[d]);
        }
    delete e;
}
GGML_CALL void split_buf_free(ggml_backend_buffer_t buffer) {
    split_buffer_ctx *c = (split_buffer_ctx *)buffer->context;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (split_extra *e : c->extras)
            split_free_extra(e);
    }
    delete c;
}
GGML_CALL void *split_buf_get_base(ggml_backend_buffer_t) {
    return (void *)0x1000; // placeholders for ggml-alloc's offsets; never dereferenced
}
GGML_CALL void split_buf_init_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor *tensor) {
    if (tensor->view_src) { // (the reference asserts: views of split tensors are not supported)
        logf("%s: a view inside a row-split buffer is not supported\n", "ggml_backend_lfamd");
        return;
    }
    split_buffer_ctx *c = (split_buffer_ctx *)buffer->context;
    split_extra *e = new split_extra;
    split_rows((const split_buft_ctx *)buffer->buft->context, tensor, e->lo);
    e->row_bytes = tensor->nb[1];
    for (int d = 0; d < g_ndev; d++) {
        const long rows = e->lo[d + 1] - e->lo[d];
        if (rows <= 0)
            continue;
        on_device g(d);
        e->bytes[d] = (size_t)rows * e->row_bytes;
        if (!g.ok || hipMalloc(&e->d[d], e->bytes[d] + 512) != hipSuccess) {
            logf("%s: allocating a row slice failed\n", "ggml_backend_lfamd");
            e->d[d] = nullptr;
        }
    }
    c->extras.push_back(e);
    tensor->extra = e;
}
GGML_CALL void split_buf_set_tensor(ggml_backend_buffer_t, struct ggml_tensor *tensor, const void *data, size_t offset, size_t size) {
    split_extra *e = (split_extra *)tensor->extra;
    if (!e || offset != 0 || size != g_api->ggml_nbytes(tensor)) { // whole tensors only (the reference asserts the same)
        logf("%s: partial writes into a row-split tensor are not supported\n", "ggml_backend_lfamd");
        return;
    }
    for (int d = 0; d < g_ndev; d++)
        if (e->d[d]) {
            on_device g(d);
            (void)hipMemcpy(e->d[d], (const uint8_t *)data + (size_t)e->lo[d] * e->row_bytes, e->bytes[d], hipMemcpyHostToDevice);
        }
    std::lock_guard<std::mutex> lk(g_mu); // packed copies of the slices are stale now
    for (int d = 0; d < g_ndev; d++)
        if (e->d[d])
            drop_range((const uint8_t *)e->d[d], (const uint8_t *)e->d[d] + e->bytes[d]);
}
GGML_CALL void split_buf_get_tensor(ggml_backend_buffer_t, const struct ggml_tensor *tensor, void *data, size_t offset, size_t size) {
    const split_extra *e = (const split_extra *)tensor->extra;
    if (!e || offset != 0 || size != g_api->ggml_nbytes(tensor)) {
        logf("%s: partial reads of a row-split tensor are not supported\n", "ggml_backend_lfamd");
        return;
    }
    for (int d = 0; d < g_ndev; d++)
        if (e->d[d]) {
            on_device g(d);
            (void)hipMemcpy((uint8_t *)data + (size_t)e->lo[d] * e->row_bytes, e->d[d], e->bytes[d], hipMemcpyDeviceToHost);
        }
}
GGML_CALL void split_buf_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    split_buffer_ctx *c = (split_buffer_ctx *)buffer->context;
    std::lock_guard<std::mutex> lk(g_mu);
    for (split_extra *e : c->extras)
        for (int d = 0; d < g_ndev; d++)
            if (e->d[d]) {
                on_device g(d);
                (void)hipMemset(e->d[d], value, e->bytes[d]);
                (void)hipDeviceSynchronize();
                drop_range((const uint8_t *)e->d[d], (const uint8_t *)e->d[d] + e->bytes[d]);
            }
}
const ggml_backend_buffer_i k_split_buffer_iface = {split_buf_get_name,   split_buf_free,       split_buf_get_base,
                                                    split_buf_init_tensor, split_buf_set_tensor, split_buf_get_tensor,
                                                    nullptr,              split_buf_clear,      nullptr};

GGML_CALL const char *split_buft_get_name(ggml_backend_buffer_type_t) {
    return "ROCm-lfamd_Split";
}
GGML_CALL ggml_backend_buffer_t split_buft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    // the size ggml-alloc asks for is the sum of get_alloc_size over the tensors; the memory is allocated per tensor and
    // device in init_tensor (like the reference, :17381-17389)
    return g_api->ggml_backend_buffer_init(buft, k_split_buffer_iface, new split_buffer_ctx, size);
}
GGML_CALL size_t split_buft_alloc_size(ggml_backend_buffer_type_t, const struct ggml_tensor *tensor) {
    return buft_alloc_size(nullptr, tensor) + 512 * (size_t)g_ndev; // (each slice carries its own tail padding)
}
bool is_split(const struct ggml_tensor *t) {
    return t && t->buffer && t->buffer->iface.get_name == split_buf_get_name;
}
bool is_split_buft(ggml_backend_buffer_type_t buft) {
    return buft && buft->iface.get_name == split_buft_get_name;
}

// ------------------------------------------------------------------ mat-mul nodes
bool row_major(const struct ggml_tensor *t) { // elements of a row contiguous, rows / slices at any stride
    return t->nb[0] == g_api->ggml_type_size(t->type);
}

// where a weight tensor's bytes are: its data pointer, or (a tensor of a row-split buffer that stayed whole) its first slice
const uint8_t *weight_bytes(const struct ggml_tensor *a) {
    return is_split(a) ? (a->extra ? (const uint8_t *)((const split_extra *)a->extra)->d[0] : nullptr) : (const uint8_t *)a->data;
}

// packed device copy of `rows` raw rows at `raw` (on the CURRENT device); kept under the raw address when `keep`
const packed *get_packed_rows(void *&scratch, size_t &scratch_cap, int type, const uint8_t *raw, long rows, long cols, size_t row_bytes,
                              bool keep, packed *tmp) {
    const size_t need = lfamd_packed_size(type, rows, cols);
    if (keep) {
        auto it = g_packed.find(raw);
        if (it != g_packed.end() && it->second.type == type && it->second.rows == rows && it->second.cols == cols &&
            it->second.row_bytes == row_bytes)
            return &it->second;
    }
    packed p;
    p.bytes = need, p.type = type, p.rows = rows, p.cols = cols, p.row_bytes = row_bytes;
    if (keep) {
        if (hipMalloc(&p.d, need ? need : 16) != hipSuccess)
            return nullptr;
    } else {
        if (!grow(scratch, scratch_cap, need))
            return nullptr;
        p.d = scratch;
    }
    if (lfamd_pack_weights(type, rows, cols, raw, row_bytes, p.d, nullptr) != LFAMD_OK) {
        if (keep)
            (void)hipFree(p.d);
        return nullptr;
    }
    // (only the K-quant batch bodies have a scaled-operand form whose range must be checked; the check synchronises the stream)
    const bool scaled_form = type == LFAMD_TYPE_Q4_K || type == LFAMD_TYPE_Q5_K || type == LFAMD_TYPE_Q6_K;
    p.exact_only = scaled_form && lfamd_scaled_gemm_ok(type, rows, cols, p.d, nullptr) == 0;
    if (keep)
        return &(g_packed[raw] = p);
    *tmp = p;
    return tmp;
}

// packed device copy of the (i02, i03) slice of src0
const packed *get_packed(backend_ctx *ctx, const struct ggml_tensor *a, int64_t i02, int64_t i03, packed *tmp) {
    const uint8_t *base = weight_bytes(a);
    if (!base)
        return nullptr;
    const bool keep = a->buffer && g_api->ggml_backend_buffer_get_usage(a->buffer) == GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
    return get_packed_rows(ctx->scratch, ctx->scratch_cap, a->type, base + i02 * a->nb[2] + i03 * a->nb[3], (long)a->ne[1], (long)a->ne[0],
                           a->nb[1], keep, tmp);
}

// The struct layouts this module reads (include/ggml_backend_lfamd.h, marked RECALLED: llama.cpp is not vendored in the reference
// tree) are cross-checked against the HOST's own accessors on every tensor an operator touches: type / ne / nb of
// `struct ggml_tensor` through ggml_nbytes, ggml_nelements, ggml_element_size and ggml_is_contiguous (each a function of exactly
// those fields), the buffer's `usage` field through ggml_backend_buffer_get_usage.  A host built against other layouts makes
// them disagree: supports_op then answers no for everything (llamafile's CPU fallback) instead of computing on garbage.
bool layout_agrees(const struct ggml_tensor *t) {
    if (!t)
        return true;
    if (t->type < 0 || t->type >= 64 || lfamd_blck_size(t->type) <= 0)
        return false;
    const int64_t ne = t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3];
    if (g_api->ggml_nelements(t) != ne || g_api->ggml_element_size(t) != (size_t)lfamd_type_size(t->type))
        return false;
    size_t nbytes; // ggml_nbytes: the span the strides cover
    const int64_t blck = lfamd_blck_size(t->type);
    if (blck == 1) {
        nbytes = lfamd_type_size(t->type);
        for (int i = 0; i < 4; i++)
            nbytes += (size_t)(t->ne[i] - 1) * t->nb[i];
    } else {
        nbytes = (size_t)t->ne[0] * t->nb[0] / (size_t)blck;
        for (int i = 1; i < 4; i++)
            nbytes += (size_t)(t->ne[i] - 1) * t->nb[i];
    }
    if (ne > 0 && g_api->ggml_nbytes(t) != nbytes)
        return false;
    const bool contiguous = t->nb[0] == (size_t)lfamd_type_size(t->type) && t->nb[1] == t->nb[0] * (size_t)t->ne[0] / (size_t)blck &&
                            t->nb[2] == t->nb[1] * (size_t)t->ne[1] && t->nb[3] == t->nb[2] * (size_t)t->ne[2];
    if (g_api->ggml_is_contiguous(t) != contiguous)
        return false;
    if (t->buffer && g_api->ggml_backend_buffer_get_usage(t->buffer) != t->buffer->usage)
        return false;
    return true;
}
bool op_layout_agrees(const struct ggml_tensor *op) {
    static bool warned = false;
    const bool ok = layout_agrees(op) && layout_agrees(op->src[0]) && layout_agrees(op->src[1]) && layout_agrees(op->src[2]);
    if (!ok && !warned) {
        warned = true;
        logf("%s: the host's ggml structs do not match this module's layouts (%s): declining every operator\n", "ggml_backend_lfamd",
             "include/ggml_backend_lfamd.h");
    }
    return ok;
}

bool mul_mat_supported(const struct ggml_tensor *op) {
    const struct ggml_tensor *a = op->src[0], *b = op->src[1];
    if (!a || !b || !type_ok(a->type) || b->type != LFAMD_TYPE_F32 || op->type != LFAMD_TYPE_F32)
        return false;
    if (!row_major(a) || !row_major(b) || !g_api->ggml_is_contiguous(op))
        return false;
    if (a->ne[0] != b->ne[0] || a->ne[0] % lfamd_blck_size(a->type))
        return false;
    if (lfamd_packed_size(a->type, (long)a->ne[1], (long)a->ne[0]) == 0 && a->ne[1] && a->ne[0])
        return false;
    return true;
}

// MUL_MAT with a row-split src0 (ggml_cuda_op_mul_mat with split = true, ggml-cuda.cu.patch:18060-18330): device d computes
// columns lo[d] .. lo[d + 1] of every result row from ITS rows of the matrix.  The backend's own device reads the activations
// and writes dst in place; every other device gets the activations copied over (f32: no device re-quantises differently), its
// product is launched before anything is waited for (the devices run side by side), and its columns are then copied into dst.
enum ggml_status run_mul_mat_split(backend_ctx *ctx, struct ggml_tensor *dst) {
    const struct ggml_tensor *a = dst->src[0], *b = dst->src[1];
    const split_extra *e = (const split_extra *)a->extra;
    if (!e)
        return GGML_STATUS_FAILED;
    const long k = (long)a->ne[0], n = (long)b->ne[1];
    const long ldc = (long)(dst->nb[1] / sizeof(float));
    if (hipDeviceSynchronize() != hipSuccess) // the activations are complete before another device reads them
        return GGML_STATUS_FAILED;
    std::lock_guard<std::mutex> lk(g_mu);
    for (int64_t i13 = 0; i13 < b->ne[3]; i13++)
        for (int64_t i12 = 0; i12 < b->ne[2]; i12++) {
            const uint8_t *bp = (const uint8_t *)b->data + i12 * b->nb[2] + i13 * b->nb[3];
            float *cp = (float *)((uint8_t *)dst->data + i12 * dst->nb[2] + i13 * dst->nb[3]);
            for (int d = 0; d < g_ndev; d++) {
                const long rows = e->lo[d + 1] - e->lo[d];
                if (rows <= 0)
                    continue;
                if (!e->d[d])
                    return GGML_STATUS_ALLOC_FAILED;
                on_device g(d);
                const bool own = d == ctx->device;
                peer_scratch &ps = ctx->peer[d];
                const void *B = bp;
                size_t brb = b->nb[1];
                float *C = cp + e->lo[d];
                long c_ld = ldc;
                if (!own) {
                    if (!grow(ps.x, ps.x_cap, (size_t)n * k * 4) || !grow(ps.c, ps.c_cap, (size_t)n * rows * 4))
                        return GGML_STATUS_ALLOC_FAILED;
                    if (hipMemcpy2DAsync(ps.x, (size_t)k * 4, bp, b->nb[1], (size_t)k * 4, n, hipMemcpyDefault, nullptr) != hipSuccess)
                        return GGML_STATUS_FAILED;
                    B = ps.x, brb = (size_t)k * 4, C = (float *)ps.c, c_ld = rows;
                }
                packed tmp;
                void *no_scratch = nullptr; // (split buffers hold weights: always kept)
                size_t no_cap = 0;
                const packed *w = get_packed_rows(no_scratch, no_cap, a->type, (const uint8_t *)e->d[d], rows, k, e->row_bytes, true, &tmp);
                if (!w)
                    return GGML_STATUS_ALLOC_FAILED;
                void *&ws = own ? ctx->ws : ps.ws;
                size_t &ws_cap = own ? ctx->ws_cap : ps.ws_cap;
                if (!grow(ws, ws_cap, lfamd_mul_mat_workspace(a->type, rows, k, n)))
                    return GGML_STATUS_ALLOC_FAILED;
                if (lfamd_mul_mat(a->type, w->d, rows, k, LFAMD_TYPE_F32, B, brb, n, C, c_ld, ws, ws_cap,
                                  (w->exact_only ? LFAMD_FLAG_PRECISE : 0u) | LFAMD_FLAG_Q0_VREGS32, nullptr) != LFAMD_OK) {
                    logf("%s: lfamd_mul_mat (row slice): %s\n", "ggml_backend_lfamd", lfamd_last_error());
                    return GGML_STATUS_FAILED;
                }
            }
            for (int d = 0; d < g_ndev; d++) {
                const long rows = e->lo[d + 1] - e->lo[d];
                if (rows <= 0 || d == ctx->device)
                    continue;
                on_device g(d);
                if (hipDeviceSynchronize() != hipSuccess ||
                    hipMemcpy2D(cp + e->lo[d], (size_t)ldc * 4, ctx->peer[d].c, (size_t)rows * 4, (size_t)rows * 4, n, hipMemcpyDefault) !=
                        hipSuccess)
                    return GGML_STATUS_FAILED;
            }
        }
    return GGML_STATUS_SUCCESS;
}

enum ggml_status run_mul_mat(backend_ctx *ctx, struct ggml_tensor *dst) {
    const struct ggml_tensor *a = dst->src[0], *b = dst->src[1];
    const long m = (long)a->ne[1], k = (long)a->ne[0], n = (long)b->ne[1];
    if (m == 0 || n == 0)
        return GGML_STATUS_SUCCESS;
    // broadcast over dims 2 / 3 like ggml_compute_forward_mul_mat (upstream; ggml.c.patch:1942-2022): src1 slice (i12, i13)
    // uses src0 slice (i12 / r2, i13 / r3)
    const int64_t r2 = b->ne[2] / (a->ne[2] ? a->ne[2] : 1), r3 = b->ne[3] / (a->ne[3] ? a->ne[3] : 1);
    if (is_split(a) && a->ne[2] * a->ne[3] == 1)
        return run_mul_mat_split(ctx, dst);
    const size_t wsb = lfamd_mul_mat_workspace(a->type, m, k, n);
    if (!grow(ctx->ws, ctx->ws_cap, wsb))
        return GGML_STATUS_ALLOC_FAILED;
    std::lock_guard<std::mutex> lk(g_mu);
    packed tmp;
    const packed *w = nullptr;
    int64_t have02 = -1, have03 = -1; // (broadcast heads share a src0 slice: packed once, not once per head)
    for (int64_t i13 = 0; i13 < b->ne[3]; i13++)
        for (int64_t i12 = 0; i12 < b->ne[2]; i12++) {
            const int64_t i02 = i12 / (r2 ? r2 : 1), i03 = i13 / (r3 ? r3 : 1);
            if (!w || i02 != have02 || i03 != have03) {
                w = get_packed(ctx, a, i02, i03, &tmp);
                have02 = i02, have03 = i03;
            }
            if (!w)
                return GGML_STATUS_ALLOC_FAILED;
            const uint8_t *bp = (const uint8_t *)b->data + i12 * b->nb[2] + i13 * b->nb[3];
            float *cp = (float *)((uint8_t *)dst->data + i12 * dst->nb[2] + i13 * dst->nb[3]);
            if (lfamd_mul_mat(a->type, w->d, m, k, LFAMD_TYPE_F32, bp, b->nb[1], n, cp, (long)(dst->nb[1] / sizeof(float)), ctx->ws,
                              ctx->ws_cap, (w->exact_only ? LFAMD_FLAG_PRECISE : 0u) | LFAMD_FLAG_Q0_VREGS32, nullptr) != LFAMD_OK) {
                logf("%s: lfamd_mul_mat: %s\n", "ggml_backend_lfamd", lfamd_last_error());
                return GGML_STATUS_FAILED;
            }
        }
    return GGML_STATUS_SUCCESS;
}

bool mul_mat_id_supported(const struct ggml_tensor *op) {
    const struct ggml_tensor *as = op->src[0], *b = op->src[1], *ids = op->src[2];
    if (!as || !b || !ids || !type_ok(as->type) || lfamd_blck_size(as->type) == 1 || b->type != LFAMD_TYPE_F32 ||
        op->type != LFAMD_TYPE_F32 || ids->type != LFAMD_TYPE_I32)
        return false;
    if (!g_api->ggml_is_contiguous(as) || !g_api->ggml_is_contiguous(b) || !g_api->ggml_is_contiguous(op) || as->ne[3] != 1)
        return false;
    if (b->ne[1] != 1 && b->ne[1] != ids->ne[0])
        return false;
    return as->ne[0] % lfamd_blck_size(as->type) == 0 && lfamd_packed_size(as->type, (long)as->ne[1], (long)as->ne[0]) != 0;
}

// the whole expert stack packed back to back under the stack's address (g_mu held)
const packed *get_packed_stack(backend_ctx *ctx, const struct ggml_tensor *as, packed *tmp) {
    const long rows = (long)as->ne[1], cols = (long)as->ne[0];
    const int experts = (int)as->ne[2];
    const size_t one = lfamd_packed_size(as->type, rows, cols);
    const uint8_t *as_bytes = weight_bytes(as); // (an expert stack of a row-split buffer stays whole on the first device)
    if (!as_bytes)
        return nullptr;
    const bool keep = as->buffer && g_api->ggml_backend_buffer_get_usage(as->buffer) == GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
    auto it = g_packed.find(as_bytes);
    if (keep && it != g_packed.end() && it->second.type == as->type && it->second.rows == rows * experts && it->second.cols == cols)
        return &it->second;
    packed p;
    p.bytes = one * experts, p.type = as->type, p.rows = rows * experts, p.cols = cols, p.row_bytes = as->nb[1];
    if (keep) {
        if (hipMalloc(&p.d, p.bytes) != hipSuccess)
            return nullptr;
    } else {
        if (!grow(ctx->scratch, ctx->scratch_cap, p.bytes))
            return nullptr;
        p.d = ctx->scratch;
    }
    for (int e = 0; e < experts; e++)
        if (lfamd_pack_weights(as->type, rows, cols, as_bytes + (size_t)e * as->nb[2], as->nb[1], (uint8_t *)p.d + (size_t)e * one,
                               nullptr) != LFAMD_OK) {
            if (keep)
                (void)hipFree(p.d);
            return nullptr;
        }
    p.exact_only = lfamd_scaled_gemm_ok(as->type, (long)experts * ((rows + 31) / 32) * 32, cols, p.d, nullptr) == 0;
    if (keep)
        return &(g_packed[as_bytes] = p);
    *tmp = p;
    return tmp;
}

// `count` MUL_MAT_ID nodes over the same activations and routing table (count == 1: the plain node; 2: ffn_gate_exps +
// ffn_up_exps of a layer, which the graph holds back to back — one launch at decode, lfamd_mul_mat_id_multi)
enum ggml_status run_mul_mat_id(backend_ctx *ctx, struct ggml_tensor *const *dsts, int count) {
    struct ggml_tensor *dst = dsts[0];
    const struct ggml_tensor *as = dst->src[0], *b = dst->src[1], *ids = dst->src[2];
    const long rows = (long)as->ne[1], cols = (long)as->ne[0], tokens = (long)b->ne[2];
    const int experts = (int)as->ne[2], thinkers = (int)ids->ne[0], tasks = (int)b->ne[1];
    if (!rows || !tokens || !thinkers)
        return GGML_STATUS_SUCCESS;
    std::lock_guard<std::mutex> lk(g_mu);
    const void *wp[4];
    float *rp[4];
    unsigned flags = LFAMD_FLAG_Q0_VREGS32;
    for (int j = 0; j < count; j++) {
        packed tmp;
        const packed *w = get_packed_stack(ctx, dsts[j]->src[0], &tmp);
        if (!w)
            return GGML_STATUS_ALLOC_FAILED;
        wp[j] = w->d, rp[j] = (float *)dsts[j]->data;
        if (w->exact_only)
            flags |= LFAMD_FLAG_PRECISE;
    }
    // routing table as contiguous int32 [tokens][thinkers]
    const int32_t *plan = (const int32_t *)ids->data;
    if (ids->nb[1] != (size_t)thinkers * 4) {
        if (!grow(ctx->plan, ctx->plan_cap, (size_t)tokens * thinkers * 4))
            return GGML_STATUS_ALLOC_FAILED;
        if (hipMemcpy2D(ctx->plan, (size_t)thinkers * 4, ids->data, ids->nb[1], (size_t)thinkers * 4, tokens, hipMemcpyDeviceToDevice) !=
            hipSuccess)
            return GGML_STATUS_FAILED;
        plan = (const int32_t *)ctx->plan;
    }
    const size_t wsb = lfamd_mul_mat_id_workspace(as->type, rows, cols, experts, tokens, thinkers);
    if (!grow(ctx->ws, ctx->ws_cap, wsb))
        return GGML_STATUS_ALLOC_FAILED;
    const int rc = count == 1 ? lfamd_mul_mat_id(as->type, wp[0], rows, cols, experts, LFAMD_TYPE_F32, b->data, b->nb[1], tasks, tokens, plan,
                                                 thinkers, rp[0], ctx->ws, ctx->ws_cap, flags, nullptr)
                              : lfamd_mul_mat_id_multi(as->type, count, wp, rows, cols, experts, LFAMD_TYPE_F32, b->data, b->nb[1], tasks,
                                                       tokens, plan, thinkers, rp, ctx->ws, ctx->ws_cap, flags, nullptr);
    if (rc != LFAMD_OK) {
        logf("%s: lfamd_mul_mat_id: %s\n", "ggml_backend_lfamd", lfamd_last_error());
        return GGML_STATUS_FAILED;
    }
    return GGML_STATUS_SUCCESS;
}

// Sibling nodes (ggml-cuda.cu.patch:18945 walks the node list one by one; llama.cpp builds attn_q / attn_k / attn_v and
// ffn_gate / ffn_up back to back over the same src1): how many nodes from `i` on can run as ONE call.
//   MUL_MAT: plain 2-D weights of this device, the same src1 tensor, at most 8 activation rows -> lfamd_mul_mat_multi_types
//   MUL_MAT_ID: the same src1 and ids, one type and shape, at most 4 tokens                    -> lfamd_mul_mat_id_multi
int sibling_run(const struct ggml_cgraph *g, int i) {
    static const bool off = getenv("LFAMD_BACKEND_NO_SIBLING_FUSION") && atoi(getenv("LFAMD_BACKEND_NO_SIBLING_FUSION"));
    const struct ggml_tensor *first = g->nodes[i];
    const struct ggml_tensor *a0 = first->src[0], *b0 = first->src[1];
    if (off || !a0 || !b0)
        return 1;
    const bool id = first->op == g_op_mul_mat_id;
    const int limit = id ? 2 : 4;
    if (id ? (b0->ne[2] > 4 || b0->ne[1] != 1 || !mul_mat_id_supported(first))
           : (b0->ne[1] > 8 || b0->ne[2] * b0->ne[3] != 1 || a0->ne[2] * a0->ne[3] != 1 || is_split(a0) || !mul_mat_supported(first) ||
              b0->nb[1] % 16 || first->nb[1] != (size_t)a0->ne[1] * 4))
        return 1;
    int n = 1;
    while (n < limit && i + n < g->n_nodes) {
        const struct ggml_tensor *t = g->nodes[i + n];
        if (t->op != first->op || t->src[1] != b0 || !t->src[0])
            break;
        const struct ggml_tensor *a = t->src[0];
        if (id) {
            // (stacks outside a weights buffer are packed into the context's ONE scratch image per call: two of them cannot be live
            // together, so such nodes run one by one)
            const auto kept = [](const struct ggml_tensor *w) {
                return w->buffer && g_api->ggml_backend_buffer_get_usage(w->buffer) == GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
            };
            if (t->src[2] != first->src[2] || a->type != a0->type || a->ne[0] != a0->ne[0] || a->ne[1] != a0->ne[1] || a->ne[2] != a0->ne[2] ||
                !mul_mat_id_supported(t) || !kept(a0) || !kept(a))
                break;
        } else if (a->ne[2] * a->ne[3] != 1 || is_split(a) || !mul_mat_supported(t) || t->nb[1] != (size_t)a->ne[1] * 4 ||
                   lfamd_vec_dot_type(a->type) != lfamd_vec_dot_type(a0->type)) {
            break;
        }
        n++;
    }
    return n;
}

enum ggml_status run_mul_mat_siblings(backend_ctx *ctx, struct ggml_tensor *const *dsts, int count) {
    const struct ggml_tensor *b = dsts[0]->src[1];
    const long k = (long)b->ne[0], n = (long)b->ne[1];
    if (n == 0)
        return GGML_STATUS_SUCCESS;
    int types[4];
    const void *A[4];
    long m[4], ldc[4];
    float *C[4];
    size_t wsb = 0;
    unsigned flags = LFAMD_FLAG_Q0_VREGS32;
    std::lock_guard<std::mutex> lk(g_mu);
    for (int j = 0; j < count; j++) {
        const struct ggml_tensor *a = dsts[j]->src[0];
        packed tmp;
        // (a matrix outside a weights buffer is packed into the one scratch area: it cannot share a call with another)
        const bool keep = a->buffer && g_api->ggml_backend_buffer_get_usage(a->buffer) == GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
        if (!keep)
            return GGML_STATUS_ABORTED;
        const packed *w = get_packed(ctx, a, 0, 0, &tmp);
        if (!w)
            return GGML_STATUS_ALLOC_FAILED;
        types[j] = a->type, A[j] = w->d, m[j] = (long)a->ne[1], ldc[j] = (long)a->ne[1], C[j] = (float *)dsts[j]->data;
        if (w->exact_only)
            flags |= LFAMD_FLAG_PRECISE;
        const size_t need = lfamd_mul_mat_workspace(a->type, m[j], k, n);
        wsb = need > wsb ? need : wsb;
    }
    if (!grow(ctx->ws, ctx->ws_cap, wsb))
        return GGML_STATUS_ALLOC_FAILED;
    if (lfamd_mul_mat_multi_types(count, types, A, m, k, LFAMD_TYPE_F32, b->data, b->nb[1], n, C, ldc, ctx->ws, ctx->ws_cap, flags,
                                  nullptr) != LFAMD_OK) {
        logf("%s: lfamd_mul_mat_multi_types: %s\n", "ggml_backend_lfamd", lfamd_last_error());
        return GGML_STATUS_FAILED;
    }
    return GGML_STATUS_SUCCESS;
}

// ------------------------------------------------------------------ backend interface
GGML_CALL const char *be_get_name(ggml_backend_t) {
    return "ROCm-lfamd";
}
GGML_CALL void be_free(ggml_backend_t backend) {
    backend_ctx *c = (backend_ctx *)backend->context;
    if (g_api && g_api->getenv && g_api->getenv("LFAMD_BACKEND_STATS"))
        fprintf(stderr, "ggml_backend_lfamd: %ld sibling calls\n", c->sibling_calls);
    {
        on_device d(c->device);
        (void)hipDeviceSynchronize();
        if (c->scratch)
            (void)hipFree(c->scratch);
        if (c->ws)
            (void)hipFree(c->ws);
        if (c->plan)
            (void)hipFree(c->plan);
    }
    for (int i = 0; i < g_ndev; i++) {
        on_device d(i);
        for (void *p : {c->peer[i].x, c->peer[i].c, c->peer[i].ws})
            if (p)
                (void)hipFree(p);
    }
    delete c;
    delete backend;
}
GGML_CALL ggml_backend_buffer_type_t be_default_buft(ggml_backend_t backend) {
    return &g_bufts[((backend_ctx *)backend->context)->device];
}
GGML_CALL void be_synchronize(ggml_backend_t backend) {
    on_device d(((backend_ctx *)backend->context)->device);
    (void)hipDeviceSynchronize();
}
GGML_CALL enum ggml_status be_graph_compute(ggml_backend_t backend, struct ggml_cgraph *cgraph) {
    backend_ctx *c = (backend_ctx *)backend->context;
    on_device on(c->device);
    if (!on.ok)
        return GGML_STATUS_FAILED;
    for (int i = 0; i < cgraph->n_nodes; i++) {
        struct ggml_tensor *node = cgraph->nodes[i];
        enum ggml_status st = GGML_STATUS_SUCCESS;
        if (node->op == g_op_mul_mat || node->op == g_op_mul_mat_id) {
            const int run = sibling_run(cgraph, i);
            if (node->op == g_op_mul_mat_id) {
                st = run_mul_mat_id(c, &cgraph->nodes[i], run);
                i += run - 1;
                c->sibling_calls += run > 1;
            } else {
                st = run > 1 ? run_mul_mat_siblings(c, &cgraph->nodes[i], run) : GGML_STATUS_ABORTED;
                if (st == GGML_STATUS_ABORTED) // (not a sibling run after all: node by node)
                    st = run_mul_mat(c, node);
                else
                    i += run - 1, c->sibling_calls += 1;
            }
        } else if (g_api->ggml_is_empty(node))
            continue;
        else {
            const char *name = g_api->ggml_op_name(node->op);
            // views of this backend's tensors carry no work (the scheduler keeps them with their source)
            if (!strcmp(name, "NONE") || !strcmp(name, "RESHAPE") || !strcmp(name, "VIEW") || !strcmp(name, "PERMUTE") ||
                !strcmp(name, "TRANSPOSE"))
                continue;
            logf("%s: op %s is not supported by this backend\n", "ggml_backend_lfamd", name);
            return GGML_STATUS_FAILED;
        }
        if (st != GGML_STATUS_SUCCESS)
            return st;
    }
    return hipDeviceSynchronize() == hipSuccess ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
}
GGML_CALL bool be_supports_op(ggml_backend_t, const struct ggml_tensor *op) {
    if (!op_layout_agrees(op))
        return false;
    if (op->op == g_op_mul_mat)
        return mul_mat_supported(op);
    if (op->op == g_op_mul_mat_id)
        return mul_mat_id_supported(op);
    const char *name = g_api->ggml_op_name(op->op);
    return !strcmp(name, "NONE") || !strcmp(name, "RESHAPE") || !strcmp(name, "VIEW") || !strcmp(name, "PERMUTE") ||
           !strcmp(name, "TRANSPOSE");
}
GGML_CALL bool be_supports_buft(ggml_backend_t backend, ggml_backend_buffer_type_t buft) {
    return buft == &g_bufts[((backend_ctx *)backend->context)->device] || is_split_buft(buft);
}
GGML_CALL bool be_offload_op(ggml_backend_t, const struct ggml_tensor *) {
    return false; // weights live in this backend's buffers; nothing is pulled over per batch
}
const ggml_backend_i k_backend_iface = {be_get_name, be_free, be_default_buft, nullptr, nullptr, nullptr, be_synchronize,
                                        nullptr,     nullptr, nullptr,         nullptr, be_graph_compute, be_supports_op,
                                        be_supports_buft, be_offload_op, nullptr, nullptr, nullptr, nullptr, nullptr};
ggml_guid g_guid = {0x6c, 0x66, 0x61, 0x6d, 0x64, 0x2d, 0x6d, 0x69, 0x33, 0x35, 0x35, 0x78, 0x2d, 0x72, 0x30, 0x32};

GGML_CALL ggml_backend_t reg_init(const char *, void *user_data) {
    return ggml_backend_cuda_init((int)(intptr_t)user_data);
}

} // namespace

extern "C" {

GGML_CALL bool ggml_cuda_link(const struct ggml_backend_api *backend_api) {
    g_api = backend_api;
    g_linked = false;
    if (!backend_api)
        return false;
    // operator numbers from the host's own table (un-vendored enum: never assumed)
    g_op_mul_mat = g_op_mul_mat_id = -1;
    for (int i = 0; i < 40 && (g_op_mul_mat < 0 || g_op_mul_mat_id < 0); i++) {
        const char *n = backend_api->ggml_op_name(i);
        if (!n)
            continue;
        if (!strcmp(n, "MUL_MAT"))
            g_op_mul_mat = i;
        else if (!strcmp(n, "MUL_MAT_ID"))
            g_op_mul_mat_id = i;
    }
    if (g_op_mul_mat < 0 || g_op_mul_mat_id < 0) {
        logf("%s: the host's ggml_op_name table has no MUL_MAT / MUL_MAT_ID below 40: refusing to link\n", "ggml_cuda_link");
        return false;
    }
    // type numbers, block and element sizes must be the ones this module was built for
    for (const type_row &r : k_types) {
        const char *n = backend_api->ggml_type_name(r.id);
        if (!n || strcmp(n, r.name) || backend_api->ggml_type_size(r.id) != lfamd_type_size(r.id) ||
            backend_api->ggml_blck_size(r.id) != lfamd_blck_size(r.id)) {
            logf("%s: ggml type %s does not match this module's block formats: refusing to link\n", "ggml_cuda_link", r.name);
            return false;
        }
    }
    // the devices this process drives: LFAMD_BACKEND_DEVICES = "all" (default) or a comma list of HIP ordinals (an ordinal may
    // repeat: two logical devices on one GPU, the single-GPU rehearsal of the multi-device paths); gfx950 only
    const int visible = lfamd_device_count();
    g_ndev = 0;
    const char *want = backend_api->getenv ? backend_api->getenv("LFAMD_BACKEND_DEVICES") : nullptr;
    auto add = [&](int ordinal) {
        hipDeviceProp_t p;
        if (ordinal < 0 || ordinal >= visible || g_ndev >= LFAMD_MAX_DEVS || hipGetDeviceProperties(&p, ordinal) != hipSuccess)
            return;
        if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
            return;
        g_phys[g_ndev++] = ordinal;
    };
    if (want && *want && strcmp(want, "all") != 0) {
        for (const char *q = want; *q;) {
            char *end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q)
                break;
            add((int)v);
            q = *end == ',' ? end + 1 : end;
            if (*end && *end != ',')
                break;
        }
    } else {
        for (int i = 0; i < visible; i++)
            add(i);
    }
    if (g_ndev <= 0 || lfamd_init(g_phys[0]) != LFAMD_OK) {
        g_ndev = 0;
        logf("%s: no MI355X (gfx950) device: %s\n", "ggml_cuda_link", lfamd_last_error());
        return false;
    }
    for (int i = 0; i < g_ndev; i++) {
        g_bufts[i] = {{buft_get_name, buft_alloc, buft_alignment, nullptr, buft_alloc_size, nullptr}, (void *)(intptr_t)i};
        on_device d(i); // peers read each other's buffers directly where the fabric allows (xGMI); copies work either way
        for (int j = 0; j < g_ndev; j++)
            if (g_phys[j] != g_phys[i]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, g_phys[i], g_phys[j]) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(g_phys[j], 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) // (copies between these two then go through the host)
                        fprintf(stderr, "ggml_backend_lfamd: peer access %d -> %d not enabled (%s)\n", g_phys[i], g_phys[j], hipGetErrorString(pe));
                }
                (void)hipGetLastError(); // (already enabled is not an error worth keeping)
            }
    }
    g_linked = true;
    return true;
}

GGML_CALL int ggml_backend_cuda_get_device_count(void) {
    return g_linked ? g_ndev : 0;
}

GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_buffer_type(int device) {
    if (!g_linked || device < 0 || device >= g_ndev)
        return nullptr;
    // LFAMD_BACKEND_MATRICES_ONLY=1: the per-layer ("offload") buffer type is host memory, so that norm weights, the KV cache
    // and compute buffers stay with the CPU backend; matrices reach the device through the split buffer type below
    // (llama.cpp --split-mode row: buft_matrix).  INTEGRATION.md section 2.
    static const bool matrices_only = [] {
        const char *e = getenv("LFAMD_BACKEND_MATRICES_ONLY");
        return e && *e && *e != '0';
    }();
    return matrices_only ? g_api->ggml_backend_cpu_buffer_type() : &g_bufts[device];
}

GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_host_buffer_type(void) {
    return g_api ? g_api->ggml_backend_cpu_buffer_type() : nullptr; // (plain host memory; uploads go through set_tensor)
}

GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_split_buffer_type(const float *tensor_split) {
    if (!g_linked)
        return nullptr;
    if (g_ndev == 1)
        return &g_bufts[0]; // nothing to split over
    // cumulative, normalised fractions (ggml-cuda.cu.patch:17432-17447); all zeros / NULL = equal shares
    split_buft_ctx sc;
    float sum = 0.0f;
    for (int d = 0; tensor_split && d < g_ndev; d++)
        sum += tensor_split[d] > 0.0f ? tensor_split[d] : 0.0f;
    float run = 0.0f;
    for (int d = 0; d <= LFAMD_MAX_DEVS; d++) {
        sc.start[d] = d >= g_ndev ? 1.0f : sum > 0.0f ? run / sum : (float)d / (float)g_ndev;
        if (d < g_ndev && sum > 0.0f)
            run += tensor_split[d] > 0.0f ? tensor_split[d] : 0.0f;
    }
    std::lock_guard<std::mutex> lk(g_split_mu);
    for (ggml_backend_buffer_type *t : g_split_bufts)
        if (!memcmp(t->context, &sc, sizeof sc))
            return t;
    ggml_backend_buffer_type *t = new ggml_backend_buffer_type{
        {split_buft_get_name, split_buft_alloc, buft_alignment, nullptr, split_buft_alloc_size, nullptr}, new split_buft_ctx(sc)};
    g_split_bufts.push_back(t);
    return t;
}

GGML_CALL ggml_backend_t ggml_backend_cuda_init(int device) {
    if (!g_linked || device < 0 || device >= g_ndev)
        return nullptr;
    backend_ctx *c = new backend_ctx;
    c->device = device;
    return new ggml_backend{&g_guid, k_backend_iface, c};
}

GGML_CALL int ggml_backend_cuda_reg_devices(void) {
    const int n = ggml_backend_cuda_get_device_count();
    for (int i = 0; i < n; i++) {
        char name[32];
        snprintf(name, sizeof name, "ROCm%d", i);
        g_api->ggml_backend_register(name, reg_init, ggml_backend_cuda_buffer_type(i), (void *)(intptr_t)i);
    }
    return n;
}

GGML_CALL void ggml_backend_cuda_get_device_properties(int device, struct ggml_cuda_device_properties *properties) {
    memset(properties, 0, sizeof *properties);
    hipDeviceProp_t p;
    if (device < 0 || device >= g_ndev || hipGetDeviceProperties(&p, g_phys[device]) != hipSuccess)
        return;
    strncpy(properties->name, p.name, sizeof(properties->name) - 1);
    properties->totalGlobalMem = p.totalGlobalMem;
    properties->multiProcessorCount = p.multiProcessorCount;
    properties->major = p.major;
    properties->minor = p.minor;
    strncpy(properties->compute, p.gcnArchName, sizeof(properties->compute) - 1);
}

GGML_CALL void ggml_backend_cuda_get_device_memory(int device, size_t *free, size_t *total) {
    *free = *total = 0;
    int cur = 0;
    if (device < 0 || device >= g_ndev || hipGetDevice(&cur) != hipSuccess || hipSetDevice(g_phys[device]) != hipSuccess)
        return;
    (void)hipMemGetInfo(free, total);
    (void)hipSetDevice(cur);
}

GGML_CALL bool ggml_backend_cuda_register_host_buffer(void *buffer, size_t size) {
    return hipHostRegister(buffer, size, hipHostRegisterPortable | hipHostRegisterReadOnly) == hipSuccess ||
           hipHostRegister(buffer, size, hipHostRegisterPortable) == hipSuccess;
}

GGML_CALL void ggml_backend_cuda_unregister_host_buffer(void *buffer) {
    (void)hipHostUnregister(buffer);
}

GGML_CALL void ggml_backend_cuda_get_device_description(int device, char *description, size_t description_size) {
    hipDeviceProp_t p;
    if (device >= 0 && device < g_ndev && hipGetDeviceProperties(&p, g_phys[device]) == hipSuccess)
        snprintf(description, description_size, "%s", p.name);
    else if (description_size)
        description[0] = 0;
}
}
