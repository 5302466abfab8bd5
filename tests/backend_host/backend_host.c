/*
 * backend_host.c — TEST INFRASTRUCTURE: plays llamafile's side of the GPU-module boundary (llamafile/cuda.c:701-753):
 * dlopen()s the module, imports the 12 GGML_CALL (ms_abi) symbols by name, hands it a ggml_backend_api callback table
 * (llama.cpp.patches/patches/ggml-backend-impl.h.patch:20-58) and drives one GGML_OP_MUL_MAT / GGML_OP_MUL_MAT_ID node
 * through ggml_backend_i.graph_compute, with tensors laid out in the module's buffer like ggml-alloc would.
 * The callbacks are this file's own minimal restatement of the ggml helpers the module calls (sizes, names,
 * contiguity); operator numbers are deliberately NOT upstream's, to prove the module resolves them by name.
 *
 *   backend_host <module.so> exports                      -> checks the 12 symbols, link() on a box without a GPU: "nolink"
 *   backend_host <module.so> mulmat <type> <m> <k> <n> <nb2> <W.bin> <X.bin> <out.bin>   (nb2 = batch slices of X per W)
 *   backend_host <module.so> mulmatid <type> <m> <k> <experts> <thinkers> <tasks> <tokens> <W.bin> <X.bin> <ids.bin> <out.bin>
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/ggml_backend_lfamd.h"
#include "../../include/lfamd_blocks.h"

#define OP_NONE 0
#define OP_MUL_MAT 31    /* (upstream's number differs: the module must not care) */
#define OP_MUL_MAT_ID 33

static bool log_disable = false;
static int registered = 0;

static GGML_CALL void h_exit(int c) { exit(c); }
static GGML_CALL void h_free(void *p) { free(p); }
static GGML_CALL void *h_malloc(size_t n) { return malloc(n); }
static GGML_CALL char *h_getenv(const char *n) { return getenv(n); }
static GGML_CALL long h_write(int fd, const void *p, long n) { return write(fd, p, n); }
static GGML_CALL void h_register(const char *name, ggml_backend_init_fn fn, ggml_backend_buffer_type_t buft, void *ud) {
    (void)name, (void)fn, (void)buft, (void)ud;
    registered++;
}
static GGML_CALL ggml_backend_buffer_t h_buffer_init(ggml_backend_buffer_type_t buft, struct ggml_backend_buffer_i iface,
                                                     ggml_backend_buffer_context_t ctx, size_t size) {
    struct ggml_backend_buffer *b = calloc(1, sizeof *b);
    b->iface = iface, b->buft = buft, b->context = ctx, b->size = size, b->usage = GGML_BACKEND_BUFFER_USAGE_ANY;
    return b;
}
static GGML_CALL ggml_backend_buffer_t h_cpu_from_ptr(void *p, size_t n) { (void)p, (void)n; return NULL; }
static struct ggml_backend_buffer_type cpu_buft_obj;
static GGML_CALL ggml_backend_buffer_type_t h_cpu_buft(void) { return &cpu_buft_obj; }
static GGML_CALL size_t h_nbytes(const struct ggml_tensor *t) {
    size_t blck = lfamd_blck_size(t->type);
    size_t n = t->ne[0] * t->nb[0] / blck;
    for (int i = 1; i < 4; i++)
        n += (t->ne[i] - 1) * t->nb[i];
    return n;
}
static GGML_CALL size_t h_buft_alloc_size(ggml_backend_buffer_type_t b, struct ggml_tensor *t) { (void)b; return h_nbytes(t); }
static GGML_CALL ggml_backend_buffer_t h_buft_alloc(ggml_backend_buffer_type_t b, size_t n) { return b->iface.alloc_buffer(b, n); }
static GGML_CALL bool h_is_cpu(ggml_backend_t b) { (void)b; return false; }
static GGML_CALL void h_tensor_get(const struct ggml_tensor *t, void *d, size_t o, size_t n) { t->buffer->iface.get_tensor(t->buffer, t, d, o, n); }
static GGML_CALL void h_tensor_set(struct ggml_tensor *t, const void *d, size_t o, size_t n) { t->buffer->iface.set_tensor(t->buffer, t, d, o, n); }
static GGML_CALL bool h_is_quantized(int t) { return lfamd_blck_size(t) > 1; }
static GGML_CALL size_t h_type_size(int t) { return lfamd_type_size(t); }
static GGML_CALL int64_t h_blck_size(int t) { return lfamd_blck_size(t); }
static GGML_CALL bool h_is_transposed(const struct ggml_tensor *t) { return t->nb[0] > t->nb[1]; }
static GGML_CALL int h_unary(const struct ggml_tensor *t) { (void)t; return 0; }
static GGML_CALL int64_t h_nelements(const struct ggml_tensor *t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
static GGML_CALL int64_t h_nrows(const struct ggml_tensor *t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
static GGML_CALL bool h_is_permuted(const struct ggml_tensor *t) { return t->nb[0] > t->nb[1] || t->nb[1] > t->nb[2] || t->nb[2] > t->nb[3]; }
static GGML_CALL bool h_is_contiguous(const struct ggml_tensor *t) {
    return t->nb[0] == lfamd_type_size(t->type) && t->nb[1] == t->nb[0] * t->ne[0] / lfamd_blck_size(t->type) &&
           t->nb[2] == t->nb[1] * t->ne[1] && t->nb[3] == t->nb[2] * t->ne[2];
}
static GGML_CALL const char *h_op_name(int op) {
    return op == OP_NONE ? "NONE" : op == OP_MUL_MAT ? "MUL_MAT" : op == OP_MUL_MAT_ID ? "MUL_MAT_ID" : "OTHER";
}
static GGML_CALL const char *h_type_name(int t) {
    switch (t) {
    case LFAMD_TYPE_F32: return "f32";
    case LFAMD_TYPE_F16: return "f16";
    case LFAMD_TYPE_Q4_0: return "q4_0";
    case LFAMD_TYPE_Q4_1: return "q4_1";
    case LFAMD_TYPE_Q5_0: return "q5_0";
    case LFAMD_TYPE_Q5_1: return "q5_1";
    case LFAMD_TYPE_Q8_0: return "q8_0";
    case LFAMD_TYPE_Q2_K: return "q2_K";
    case LFAMD_TYPE_Q3_K: return "q3_K";
    case LFAMD_TYPE_Q4_K: return "q4_K";
    case LFAMD_TYPE_Q5_K: return "q5_K";
    case LFAMD_TYPE_Q6_K: return "q6_K";
    case LFAMD_TYPE_IQ4_XS: return "iq4_xs";
    case LFAMD_TYPE_I32: return "i32";
    case LFAMD_TYPE_BF16: return "bf16";
    default: return "?";
    }
}
/* BACKEND_HOST_SKEW=1: a host whose struct ggml_tensor differs from the module's header would make the accessors disagree with
   what the module reads from the struct: simulated by an element size that is off by one */
static GGML_CALL size_t h_element_size(const struct ggml_tensor *t) { return lfamd_type_size(t->type) + (getenv("BACKEND_HOST_SKEW") ? 1 : 0); }
static GGML_CALL size_t h_row_size(int t, int64_t ne) { return lfamd_type_size(t) * ne / lfamd_blck_size(t); }
static GGML_CALL void h_rope(int a, int b, float c, float d, float e, float f[2]) { (void)a, (void)b, (void)c, (void)d, (void)e, (void)f; }
static GGML_CALL const char *h_op_desc(const struct ggml_tensor *t) { return h_op_name(t->op); }
static GGML_CALL bool h_buffer_is_host(ggml_backend_buffer_t b) { (void)b; return false; }
static GGML_CALL bool h_guid_matches(ggml_guid_t a, ggml_guid_t b) { return !memcmp(a, b, 16); }
static GGML_CALL bool h_is_empty(const struct ggml_tensor *t) { return !t->ne[0] || !t->ne[1] || !t->ne[2] || !t->ne[3]; }
static GGML_CALL enum ggml_backend_buffer_usage h_usage(ggml_backend_buffer_t b) { return b->usage; }
static GGML_CALL bool h_same_shape(const struct ggml_tensor *a, const struct ggml_tensor *b) { return !memcmp(a->ne, b->ne, sizeof a->ne); }
static GGML_CALL bool h_contig1(const struct ggml_tensor *t) { return t->nb[2] == t->nb[1] * t->ne[1] && t->nb[3] == t->nb[2] * t->ne[2]; }
static GGML_CALL bool h_contig2(const struct ggml_tensor *t) { return t->nb[3] == t->nb[2] * t->ne[2]; }

static struct ggml_backend_api api = {
    &log_disable, h_exit, h_free, h_malloc, h_getenv, h_write, h_register, h_buffer_init, h_cpu_from_ptr, h_cpu_buft, h_buft_alloc_size,
    h_buft_alloc, h_is_cpu, h_tensor_get, h_tensor_set, h_is_quantized, h_type_size, h_blck_size, h_is_transposed, h_nbytes, h_unary,
    h_nelements, h_nrows, h_is_permuted, h_is_contiguous, h_op_name, h_type_name, h_element_size, h_row_size, h_rope, h_op_desc,
    h_buffer_is_host, h_guid_matches, h_is_empty, h_usage, h_same_shape, h_contig1, h_contig2};

static void *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    *n = ftell(f);
    rewind(f);
    void *p = malloc(*n ? *n : 1);
    if (fread(p, 1, *n, f) != *n) { perror("read"); exit(2); }
    fclose(f);
    return p;
}

static void shape(struct ggml_tensor *t, int type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    memset(t, 0, sizeof *t);
    t->type = type;
    t->ne[0] = ne0, t->ne[1] = ne1, t->ne[2] = ne2, t->ne[3] = ne3;
    t->nb[0] = lfamd_type_size(type);
    t->nb[1] = t->nb[0] * ne0 / lfamd_blck_size(type);
    t->nb[2] = t->nb[1] * ne1;
    t->nb[3] = t->nb[2] * ne2;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage\n"); return 2; }
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    /* llamafile/cuda.c:726-737 */
    const char *names[12] = {"ggml_cuda_link", "ggml_backend_cuda_host_buffer_type", "ggml_backend_cuda_buffer_type", "ggml_backend_cuda_init",
                             "ggml_backend_cuda_split_buffer_type", "ggml_backend_cuda_reg_devices", "ggml_backend_cuda_get_device_properties",
                             "ggml_backend_cuda_get_device_memory", "ggml_backend_cuda_get_device_count", "ggml_backend_cuda_unregister_host_buffer",
                             "ggml_backend_cuda_register_host_buffer", "ggml_backend_cuda_get_device_description"};
    void *sym[12];
    for (int i = 0; i < 12; i++)
        if (!(sym[i] = dlsym(lib, names[i]))) { fprintf(stderr, "missing symbol %s\n", names[i]); return 4; }
    bool GGML_CALL (*link)(const struct ggml_backend_api *) = sym[0];
    ggml_backend_buffer_type_t GGML_CALL (*buffer_type)(int) = sym[2];
    ggml_backend_t GGML_CALL (*backend_init)(int) = sym[3];
    int GGML_CALL (*reg_devices)(void) = sym[5];
    void GGML_CALL (*get_props)(int, struct ggml_cuda_device_properties *) = sym[6];
    void GGML_CALL (*get_mem)(int, size_t *, size_t *) = sym[7];
    int GGML_CALL (*get_count)(void) = sym[8];
    void GGML_CALL (*get_desc)(int, char *, size_t) = sym[11];

    const bool linked = link(&api);
    if (!strcmp(argv[2], "exports")) {
        printf("%s count=%d\n", linked ? "linked" : "nolink", get_count());
        return 0;
    }
    if (!strcmp(argv[2], "bufts")) { /* which buffer type a layer's small tensors / its matrices get (llama.cpp buft vs buft_matrix) */
        ggml_backend_buffer_type_t GGML_CALL (*split_type)(const float *) = sym[4];
        if (!linked) return 5;
        printf("layer=%s matrix=%s\n", buffer_type(0) == h_cpu_buft() ? "host" : "device", split_type(NULL) == h_cpu_buft() ? "host" : "device");
        return 0;
    }
    if (!linked) { fprintf(stderr, "link failed\n"); return 5; }
    const int ndev = get_count();
    if (ndev < 1 || reg_devices() != ndev || registered != ndev) { fprintf(stderr, "reg_devices\n"); return 6; }
    struct ggml_cuda_device_properties pr;
    get_props(0, &pr);
    size_t fr = 0, tot = 0;
    get_mem(0, &fr, &tot);
    char desc[128];
    get_desc(0, desc, sizeof desc);
    fprintf(stderr, "device: %s (%s) CUs=%d mem %zu / %zu MiB free; %s\n", pr.name, pr.compute, pr.multiProcessorCount, fr >> 20, tot >> 20, desc);
    if (strncmp(pr.compute, "gfx950", 6) || !tot) return 7;

    /* BACKEND_HOST_MAIN: the logical device whose backend runs the graph (llama.cpp's main_gpu);
       BACKEND_HOST_SPLIT="f0,f1,...": the weights go into ggml_backend_cuda_split_buffer_type(tensor_split) (--split-mode row) */
    const int main_dev = getenv("BACKEND_HOST_MAIN") ? atoi(getenv("BACKEND_HOST_MAIN")) : 0;
    if (buffer_type(ndev) || backend_init(ndev) || buffer_type(-1)) { fprintf(stderr, "a device past the count was served\n"); return 8; }
    ggml_backend_buffer_type_t buft = buffer_type(main_dev);
    ggml_backend_t be = backend_init(main_dev);
    if (!buft || !be) return 8;
    ggml_backend_buffer_type_t wbuft = buft;
    if (getenv("BACKEND_HOST_SPLIT")) {
        ggml_backend_buffer_type_t GGML_CALL (*split_type)(const float *) = sym[4];
        float fr[16] = {0};
        int i = 0;
        for (const char *q = getenv("BACKEND_HOST_SPLIT"); *q && i < 16; i++) {
            char *end;
            fr[i] = strtof(q, &end);
            q = *end == ',' ? end + 1 : end;
        }
        wbuft = split_type(fr);
        if (!wbuft || wbuft != split_type(fr)) { fprintf(stderr, "split buffer type\n"); return 8; }
        if (ndev > 1 && (wbuft == buft || !be->iface.supports_buft(be, wbuft))) { fprintf(stderr, "split buffer type not served\n"); return 8; }
        fprintf(stderr, "weights in %s\n", wbuft->iface.get_name(wbuft));
    }

    struct ggml_tensor W, X, IDS, OUT;
    size_t nw, nx, ni = 0;
    void *hw, *hx, *hi = NULL;
    const char *outpath;
    const int type = atoi(argv[3]);
    if (!strcmp(argv[2], "mulmat")) {
        const long m = atol(argv[4]), k = atol(argv[5]), n = atol(argv[6]), nb2 = atol(argv[7]);
        hw = slurp(argv[8], &nw), hx = slurp(argv[9], &nx), outpath = argv[10];
        shape(&W, type, k, m, 1, 1);
        shape(&X, LFAMD_TYPE_F32, k, n, nb2, 1); /* nb2 slices of X broadcast against the one W (r2 = nb2) */
        shape(&OUT, LFAMD_TYPE_F32, m, n, nb2, 1);
        OUT.op = OP_MUL_MAT, OUT.src[0] = &W, OUT.src[1] = &X;
    } else {
        const long m = atol(argv[4]), k = atol(argv[5]), experts = atol(argv[6]), thinkers = atol(argv[7]), tasks = atol(argv[8]),
                   tokens = atol(argv[9]);
        hw = slurp(argv[10], &nw), hx = slurp(argv[11], &nx), hi = slurp(argv[12], &ni), outpath = argv[13];
        shape(&W, type, k, m, experts, 1);
        shape(&X, LFAMD_TYPE_F32, k, tasks, tokens, 1);
        shape(&IDS, LFAMD_TYPE_I32, thinkers, tokens, 1, 1);
        shape(&OUT, LFAMD_TYPE_F32, m, thinkers, tokens, 1);
        OUT.op = OP_MUL_MAT_ID, OUT.src[0] = &W, OUT.src[1] = &X, OUT.src[2] = &IDS;
    }
    if (h_nbytes(&W) != nw || h_nbytes(&X) != nx) { fprintf(stderr, "input size mismatch %zu %zu / %zu %zu\n", h_nbytes(&W), nw, h_nbytes(&X), nx); return 9; }
    /* a weights buffer and a compute buffer, tensors placed like ggml-alloc would (aligned offsets) */
    const size_t align = buft->iface.get_alignment(buft);
    ggml_backend_buffer_t wbuf = wbuft->iface.alloc_buffer(wbuft, wbuft->iface.get_alloc_size(wbuft, &W) + align);
    const size_t xo = (h_nbytes(&X) + align - 1) / align * align, io = (ni + align - 1) / align * align;
    ggml_backend_buffer_t cbuf = buft->iface.alloc_buffer(buft, xo + io + h_nbytes(&OUT) + align);
    if (!wbuf || !cbuf) return 10;
    /* BACKEND_HOST_NO_WEIGHTS_USAGE: the matrices live in an ordinary buffer (no packed copy is kept: the module packs per call) */
    const int weights_usage = getenv("BACKEND_HOST_NO_WEIGHTS_USAGE") == NULL;
    if (weights_usage)
        wbuf->usage = GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
    W.buffer = wbuf, W.data = wbuf->iface.get_base(wbuf);
    if (wbuf->iface.init_tensor)
        wbuf->iface.init_tensor(wbuf, &W); /* (ggml-alloc calls it for every tensor it places) */
    uint8_t *cb = cbuf->iface.get_base(cbuf);
    X.buffer = cbuf, X.data = cb;
    IDS.buffer = cbuf, IDS.data = cb + xo;
    OUT.buffer = cbuf, OUT.data = cb + xo + io;
    wbuf->iface.set_tensor(wbuf, &W, hw, 0, nw);
    if (getenv("BACKEND_HOST_SPLIT")) { /* a row-split tensor reads back as the bytes that were written */
        void *back = malloc(nw);
        wbuf->iface.get_tensor(wbuf, &W, back, 0, nw);
        if (memcmp(back, hw, nw)) { fprintf(stderr, "split tensor read-back differs\n"); return 14; }
        free(back);
    }
    cbuf->iface.set_tensor(cbuf, &X, hx, 0, nx);
    if (hi)
        cbuf->iface.set_tensor(cbuf, &IDS, hi, 0, ni);
    if (!be->iface.supports_op(be, &OUT)) { fprintf(stderr, "supports_op says no\n"); return 11; }
    /* BACKEND_HOST_PAIR: a second node over the SAME src1 (and ids) with its own copy of the weights, right behind the first
       (attn_q / attn_k, ffn_gate / ffn_up): the module may run the two as one call; both results must be the same bytes */
    struct ggml_tensor W2, OUT2;
    struct ggml_tensor *nodes[2] = {&OUT, &OUT2};
    const int pair = getenv("BACKEND_HOST_PAIR") != NULL;
    ggml_backend_buffer_t wbuf2 = NULL, cbuf2 = NULL;
    if (pair) {
        W2 = W, OUT2 = OUT;
        wbuf2 = buft->iface.alloc_buffer(buft, buft->iface.get_alloc_size(buft, &W2) + align);
        cbuf2 = buft->iface.alloc_buffer(buft, h_nbytes(&OUT) + align);
        if (!wbuf2 || !cbuf2) return 10;
        if (weights_usage)
            wbuf2->usage = GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
        W2.buffer = wbuf2, W2.data = wbuf2->iface.get_base(wbuf2), W2.extra = NULL;
        wbuf2->iface.set_tensor(wbuf2, &W2, hw, 0, nw);
        OUT2.buffer = cbuf2, OUT2.data = cbuf2->iface.get_base(cbuf2), OUT2.src[0] = &W2;
    }
    struct ggml_cgraph g = {2, pair ? 2 : 1, 0, nodes, NULL, NULL};
    for (int rep = 0; rep < 2; rep++) /* the second run uses the kept packed weights */
        if (be->iface.graph_compute(be, &g) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); return 12; }
    be->iface.synchronize(be);
    size_t no = h_nbytes(&OUT);
    void *ho = malloc(no);
    /* clear() rewrites the weights behind the kept packed copy: zero bytes are zero weights in every block format, so the
       product must come out as exact zeros; set_tensor() afterwards must bring the real product back */
    wbuf->iface.clear(wbuf, 0);
    if (be->iface.graph_compute(be, &g) != GGML_STATUS_SUCCESS) return 12;
    be->iface.synchronize(be);
    cbuf->iface.get_tensor(cbuf, &OUT, ho, 0, no);
    for (size_t i = 0; i < no / 4; i++)
        if (((const uint32_t *)ho)[i] & 0x7fffffffu) { fprintf(stderr, "stale packed weights after clear()\n"); return 13; }
    wbuf->iface.set_tensor(wbuf, &W, hw, 0, nw);
    if (be->iface.graph_compute(be, &g) != GGML_STATUS_SUCCESS) return 12;
    be->iface.synchronize(be);
    cbuf->iface.get_tensor(cbuf, &OUT, ho, 0, no);
    if (pair) {
        void *ho2 = malloc(no);
        cbuf2->iface.get_tensor(cbuf2, &OUT2, ho2, 0, no);
        if (memcmp(ho, ho2, no)) { fprintf(stderr, "sibling nodes over the same weights and activations differ\n"); return 15; }
        free(ho2);
        cbuf2->iface.free_buffer(cbuf2);
        wbuf2->iface.free_buffer(wbuf2);
    }
    FILE *f = fopen(outpath, "wb");
    fwrite(ho, 1, no, f);
    fclose(f);
    cbuf->iface.free_buffer(cbuf);
    wbuf->iface.free_buffer(wbuf);
    be->iface.free(be);
    printf("ok\n");
    return 0;
}
