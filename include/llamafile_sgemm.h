/*
 * llamafile_sgemm.h — the drop-in boundary: the C ABI of llamafile's CPU mat-mul plug-in, served by
 * MI355X kernels.  libllamafile_sgemm.so exports exactly these symbols; each replaces the reference
 * symbol of the same name (paths relative to /root/reference):
 *
 *   llamafile_sgemm        llamafile/sgemm.h:23-24  (dispatcher llamafile/sgemm.cpp:128-131,
 *                          body llamafile/tinyblas_cpu_sgemm.inc:274-331)
 *   llamafile_mixmul       llamafile/sgemm.h:25-26  (llamafile/tinyblas_cpu_mixmul.inc:395-398)
 *   llamafile_mixmul_needs llamafile/sgemm.h:27-28  (llamafile/tinyblas_cpu_mixmul_amd_avx.cpp:8-18)
 *   llamafile_mixmul_iqk   llamafile/sgemm.h:76-77  (llamafile/sgemm.cpp:141-145)
 *   iqk_mul_mat            llamafile/sgemm.h:10     (llamafile/iqk_mul_mat.inc:182-202)
 *   iqk_mul_mat_moe        llamafile/sgemm.h:14-15  (llamafile/iqk_mul_mat.inc:204-221)
 *
 * Same signatures (SysV calling convention, `long`/`int` arguments), same units (k, lda, ldb in BLOCKS
 * for quantised types), same return convention: `true` = "I serviced this request", `false` = the
 * caller falls back to its generic path (llama.cpp.patches/patches/ggml.c.patch:1964-1965,
 * 2010-2016); precondition violations abort like the reference's asserts
 * (tinyblas_cpu_sgemm.inc:277-284).  All pointers are HOST pointers borrowed for the call, as in the
 * reference.  Threading contract: all `nth` ggml worker threads call with the same
 * arguments; every thread gets the same boolean; thread ith == 0 performs the device work and
 * returns when C is complete, the others return at once (the executor's per-node barrier,
 * ggml.c.patch:2325, orders their next node after it).
 *
 * The library dlopen()s the HIP module (include/lfamd_hip.h) on first use — like llamafile/cuda.c
 * dlopen()s ggml-rocm.so — and returns `false` from every entry point when no MI355X / module is
 * present, which is precisely the reference's "not supported here" answer.  It never computes on
 * the CPU.
 */
#ifndef LLAMAFILE_SGEMM_AMD_H_
#define LLAMAFILE_SGEMM_AMD_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ggml structs the ABI touches (un-vendored upstream llama.cpp @ 8b3befc, RECALLED — kept in this
 * one place so it can be regenerated when the submodule is available) ---- */
#define LFAMD_GGML_MAX_DIMS 4
#define LFAMD_GGML_MAX_SRC 10
#define LFAMD_GGML_MAX_OP_PARAMS 64
#define LFAMD_GGML_MAX_NAME 128 /* llamafile raises it from 64: ggml.h.patch:27-29 */

struct ggml_backend_buffer;
struct ggml_compute_state_shared;

struct ggml_tensor {
    int type;    /* enum ggml_type */
    int backend; /* deprecated enum ggml_backend_type */
    struct ggml_backend_buffer *buffer;
    int64_t ne[LFAMD_GGML_MAX_DIMS];
    size_t nb[LFAMD_GGML_MAX_DIMS];
    int op; /* enum ggml_op */
    int32_t op_params[LFAMD_GGML_MAX_OP_PARAMS / sizeof(int32_t)];
    int32_t flags;
    struct ggml_tensor *grad;
    struct ggml_tensor *src[LFAMD_GGML_MAX_SRC];
    struct ggml_tensor *view_src;
    size_t view_offs;
    void *data;
    char name[LFAMD_GGML_MAX_NAME];
    void *extra;
};

struct ggml_compute_params { /* ggml.h.patch:48-57 */
    int ith, nth;
    size_t wsize;
    void *wdata;
    struct ggml_compute_state_shared *shared;
};

/* {expert slot, token}: iqk_mul_mat.inc:69-72 */
struct lfamd_mmid_row_mapping {
    int32_t i1;
    int32_t i2;
};

/* ---- the reference ABI ---- */
bool llamafile_sgemm(long m, long n, long k, const void *A, long lda, const void *B, long ldb, void *C,
                     long ldc, int ith, int nth, int Atype, int Btype, int Ctype);
bool llamafile_mixmul(const struct ggml_compute_params *params, const struct ggml_tensor *weights,
                      const struct ggml_tensor *thought, const struct ggml_tensor *plan,
                      struct ggml_tensor *result);
size_t llamafile_mixmul_needs(const struct ggml_tensor *weights, const struct ggml_tensor *thought,
                              const struct ggml_tensor *plan);
bool llamafile_mixmul_iqk(long Nx, long Ny, long ne00, int ne11, int typeA, const void *A, const void *B,
                          float *C, long nb1, long nb2, const void *vrow_mapping, int ith, int nth);
bool iqk_mul_mat(long Nx, long Ny, long ne00, int typeA, const void *A, const void *B, float *C,
                 long stride_C, int ith, int nth);
bool iqk_mul_mat_moe(long Nx, long Ny, long ne00, int ne11, int typeA, const void *A, const void *B,
                     float *C, long nb1, long nb2, const void *vrow_mapping, int ith, int nth);

/* ---- management additions (not in the reference; a host may ignore them) ---- */
/* 1 if the HIP module is loaded and a gfx950 device initialised, else 0 (+ reason via _error). */
int llamafile_sgemm_amd_available(void);
const char *llamafile_sgemm_amd_error(void);
/* Device copies of weights are kept across calls ONLY for host bytes that cannot change behind the library's back:
 * ranges registered here (the host promises the bytes stay put until it unregisters them), and addresses inside a
 * mapping without write permission (an mmap'd GGUF file; read from /proc/self/maps).  Any other `A` — ggml also enters
 * llamafile_sgemm with the mutable KV cache as `A` for KQ / KQV — is uploaded and packed on every call.  The kept
 * copies are evicted least-recently-used within a byte budget (default 200 GiB). */
void llamafile_sgemm_amd_register_weights(const void *p, size_t bytes);
void llamafile_sgemm_amd_unregister_weights(const void *p); /* also frees the device copies inside the range */
void llamafile_sgemm_amd_set_cache_budget(size_t bytes);
size_t llamafile_sgemm_amd_cached_bytes(void);
/* Drop one tensor's device copy / everything (e.g. before unmapping a model). */
void llamafile_sgemm_amd_forget(const void *A);
void llamafile_sgemm_amd_reset(void);
/* A kept copy is re-validated on every use: the mapping's identity (device, inode, file offset of the tensor's first byte)
 * and a fingerprint of sampled bytes must still match, so a model that was unmapped and whose address range now holds
 * other bytes is packed again without the host calling _forget / _reset.  The /proc/self/maps snapshot is parsed again
 * when an address is unknown to it or it is older than 200 ms; this counts the parses (diagnostic, tests). */
unsigned long llamafile_sgemm_amd_maps_reads(void);
/* FLAG_precise of the reference (--precise): Kahan summation in the Q8_0/Q4_0 kernels. */
void llamafile_sgemm_amd_set_precise(int precise);

/* ---- model files ----
 * Minimal GGUF v2 / v3 reader (reference: gguf_init_from_file, upstream ggml.c; llamafile's changes at
 * llama.cpp.patches/patches/ggml.c.patch:2503-2612).  The file is mapped read-only; a tensor is handed out as a pointer into
 * that mapping — immutable bytes in the sense above, so llamafile_sgemm uploads and packs it once.  Tensor i: ne[0] = row
 * length (k), ne[1] = rows (m); `nbytes` is 0 for a type without a block format here. */
typedef struct lfamd_gguf lfamd_gguf;
lfamd_gguf *lfamd_gguf_open(const char *path, char *err, size_t errlen);
void lfamd_gguf_close(lfamd_gguf *g);
int lfamd_gguf_version(const lfamd_gguf *g);
long lfamd_gguf_n_tensors(const lfamd_gguf *g);
long lfamd_gguf_n_kv(const lfamd_gguf *g);
size_t lfamd_gguf_alignment(const lfamd_gguf *g);
int lfamd_gguf_tensor(const lfamd_gguf *g, long i, const char **name, int *type, int *n_dims, int64_t ne[4], const void **data,
                      size_t *nbytes);
long lfamd_gguf_find_tensor(const lfamd_gguf *g, const char *name);
int lfamd_gguf_get_u64(const lfamd_gguf *g, const char *key, uint64_t *v);
int lfamd_gguf_get_f64(const lfamd_gguf *g, const char *key, double *v);
const char *lfamd_gguf_get_str(const lfamd_gguf *g, const char *key);

#ifdef __cplusplus
}
#endif
#endif
