// api.hip — the C ABI of libllamafile_amd_hip.so (include/lfamd_hip.h).
//
// Dispatch policy mirrors ggml_cuda_mul_mat (ggml-cuda.cu.patch:18377-18443): n <= 8 -> GEMV
// kernels (MMVQ_MAX_BATCH_SIZE = 8, :14359), otherwise the MFMA GEMM; types without a tuned kernel
// run the generic kernel.  There is no CPU fallback anywhere in this module.
#include "lfamd_device.h"
#include "../../include/lfamd_hip.h"

#include <stdio.h>
#include <string.h>

extern "C" {
hipError_t lfamd_launch_pack_q4k(const void *, size_t, long, long, void *, hipStream_t);
hipError_t lfamd_launch_pack_q40(const void *, size_t, long, long, void *, hipStream_t);
hipError_t lfamd_launch_prep80(int, const void *, size_t, long, long, long, void *, void *, void *, hipStream_t);
hipError_t lfamd_launch_wprep32(int, const void *, size_t, long, long, void *, hipStream_t);
size_t lfamd_wprep32_bytes(long, long);
hipError_t lfamd_launch_prep_float(int, int, const void *, size_t, long, long, long, void *, hipStream_t);
hipError_t lfamd_launch_wprep8(int, const void *, size_t, long, long, void *, hipStream_t);
size_t lfamd_wprep8_bytes(long, long);
hipError_t lfamd_launch_pack_q5k(const void *, size_t, long, long, void *, hipStream_t);
hipError_t lfamd_launch_pack_q6k(const void *, size_t, long, long, void *, hipStream_t);
hipError_t lfamd_launch_pack_q80(const void *, size_t, long, long, void *, hipStream_t);
hipError_t lfamd_launch_pack_raw(const void *, size_t, long, size_t, void *, hipStream_t);
hipError_t lfamd_launch_prep_q8k(const void *, size_t, long, long, long, void *, void *, void *, int, const int32_t *, hipStream_t);
hipError_t lfamd_launch_prep_f32(const void *, size_t, long, long, long, void *, void *, void *, int, const int32_t *, hipStream_t);
bool lfamd_moe_decode_multi_ok(int type, long cols, int Btype, int tasks, long tokens, unsigned flags);
hipError_t lfamd_launch_moe_decode_multi(int type, int count, const void *const *W, long rows, long cols, int experts, size_t expert_bytes,
                                         int Btype, const void *thought, size_t b_row_bytes, long tokens, const int32_t *plan, int thinkers,
                                         float *const *result, hipStream_t s);
bool lfamd_blaslt_ok();
size_t lfamd_blaslt_workspace();
hipError_t lfamd_blaslt_gemm(int dtype, const void *W, long ldw, const void *X, long ldx, long m, long n, long k, float *C, long ldc, void *ws,
                             size_t ws_bytes, hipStream_t s);
hipError_t lfamd_launch_rows_to_16(int dtype, const void *X, size_t x_row_bytes, long n, long k, void *out, hipStream_t s);
hipError_t lfamd_launch_q80_rows_to_f16(int Btype, const void *X, size_t x_row_bytes, long n, long k, void *out, hipStream_t s);
hipError_t lfamd_launch_q80_image(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s);
size_t lfamd_pk_bytes(int type, long rows, long cols);
hipError_t lfamd_launch_pk4x_pack(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s);
hipError_t lfamd_launch_pk4x_expand(const void *packed, long rows, long cols, void *out, hipStream_t s);
hipError_t lfamd_launch_pk_pack(int type, const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s);
hipError_t lfamd_launch_pk_expand(int type, const void *packed, long rows, long cols, void *out, hipStream_t s);
size_t lfamd_gemm_sb_workspace(long k);
bool lfamd_gemm_sb_ok(int Atype, long k, long n);
hipError_t lfamd_launch_gemm_sb(int Atype, const void *A, long m, long k, int Btype, const void *B, size_t b_row_bytes, long n, float *C,
                                long ldc, void *ws, int reuse_stage, hipStream_t s);
hipError_t lfamd_launch_wprep16(int, const void *, size_t, long, long, void *, hipStream_t);
size_t lfamd_wprep16_bytes(long, long);
hipError_t lfamd_launch_generic(int, const void *, long, long, int, const void *, size_t, long, float *, long, hipStream_t);
hipError_t lfamd_launch_gemv_float(int, const void *, long, long, int, const void *, size_t, long, float *, long, hipStream_t);
int lfamd_gemv_float_ok(int, long, long);
hipError_t lfamd_launch_gemv(int, const void *, long, long, int, const void *, size_t, long, float *, long, int, int,
                             hipStream_t);
hipError_t lfamd_launch_gemv_multi(int, int, const void *const *, const long *, long, int, const void *, size_t, long,
                                   float *const *, const long *, int, int, hipStream_t);
hipError_t lfamd_launch_gemm_q80(const void *, long, long, int, const void *, size_t, long, float *, long, void *, int, int,
                                 hipStream_t);
size_t lfamd_gemm_q80_workspace(long, long);
#define LW_MIN_TILES 1 // the 128 x 64 loader-wave tile beats the split-K body at every grid measured (8 .. 128 tiles: 5-10 %)
hipError_t lfamd_launch_gemv_dual(int, int, const void *const *, const long *, float *const *, const long *, int, int,
                                  const void *const *, const long *, float *const *, const long *, long, int, const void *, size_t,
                                  hipStream_t);
hipError_t lfamd_launch_scaled_ok(int, long, long, const void *, int *, hipStream_t);
int lfamd_gemm_wide_scaled_ok(int, int);
// (the wide launchers take `mode`: bit 0 plain body, bit 1 activations staged scaled — gemm_wide.hip)
hipError_t lfamd_launch_gemm_wide_multi(int, int, const void *const *, const long *, long, const void *, const void *,
                                        const void *, long, long, float *const *, const long *, int, void *, size_t, hipStream_t);
hipError_t lfamd_launch_gemm_wide(int, const void *, long, long, const void *, const void *, const void *, long, long,
                                  float *, long, int, void *, size_t, hipStream_t);
size_t lfamd_gemm_lw_ksplit_bytes(long, long);
hipError_t lfamd_launch_gemm_wide_dual(int, int, const void *const *, const long *, float *const *, const long *, int, int,
                                       const void *const *, const long *, float *const *, const long *, long, const void *,
                                       const void *, const void *, long, long, int, hipStream_t);
hipError_t lfamd_launch_gemm_kq(int, const void *, long, long, const void *, const void *, const void *, long, long,
                                float *, long, hipStream_t);
hipError_t lfamd_launch_quantize(int, const float *, long, long, size_t, void *, size_t, hipStream_t);
hipError_t lfamd_launch_moe(int, const void *, long, long, int, size_t, int, const void *, size_t, int, long,
                            const int32_t *, int, float *, void *, size_t, unsigned, hipStream_t);
size_t lfamd_moe_workspace(int, long, long, int, long, int);
int lfamd_gemm_i8_ok(int Atype, long row_blocks128, long n);
size_t lfamd_gemm_i8_workspace(long k, long n);
hipError_t lfamd_launch_gemm_lf_q80(int count, const void *const *A, const long *m, long k, int Btype, const void *B, size_t b_row_bytes, long n,
                                    float *const *C, const long *ldc, void *ws, hipStream_t s);
size_t lfamd_gemm_lf_workspace(long k, long n);
hipError_t lfamd_launch_gemm_lf_float(int Atype, const void *A, size_t a_row_bytes, long m, long k, const void *Xh, long n, long n_pad, float *C,
                                      long ldc, hipStream_t s);
hipError_t lfamd_launch_gemm_i8_staged(int count, const void *const *A, const long *m, long k, const void *image, long n, float *const *C,
                                       const long *ldc, hipStream_t s);
hipError_t lfamd_launch_gemm_i8(int count, const void *const *A, const long *m, long k, int Btype, const void *B, size_t b_row_bytes, long n,
                                float *const *C, const long *ldc, void *ws, const int32_t *src_idx, hipStream_t s);
}

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail) {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

static int hip_fail(hipError_t e, const char *where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return LFAMD_ERR_HIP;
}

#define HIPCHK(expr, where)                                                                                            \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return hip_fail(e_, where);                                                                                \
    } while (0)

static inline size_t align_up(size_t x, size_t a) {
    return (x + a - 1) / a * a;
}

extern "C" {

int lfamd_abi_version(void) {
    return LFAMD_ABI_VERSION;
}

const char *lfamd_last_error(void) {
    return g_err;
}

void lfamd_set_error(const char *msg) { // (for the module's other translation units: comm.hip, backend glue)
    snprintf(g_err, sizeof(g_err), "%s", msg);
}

int lfamd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int lfamd_device_name(int device, char *buf, size_t len) {
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device), "hipGetDeviceProperties");
    snprintf(buf, len, "%s (%s)", p.name, p.gcnArchName);
    return LFAMD_OK;
}

int lfamd_init(int device) {
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device), "hipGetDeviceProperties");
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(LFAMD_ERR_UNSUPPORTED, "device arch %s is not gfx950 (MI355X); this module has no other code objects",
                    p.gcnArchName);
    HIPCHK(hipSetDevice(device), "hipSetDevice");
    return LFAMD_OK;
}

int lfamd_malloc(void **dptr, size_t bytes) {
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 16), "hipMalloc");
    return LFAMD_OK;
}
int lfamd_free(void *dptr) {
    HIPCHK(hipFree(dptr), "hipFree");
    return LFAMD_OK;
}
int lfamd_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), "hipMemcpyAsync h2d");
    return LFAMD_OK;
}
int lfamd_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync d2h");
    return LFAMD_OK;
}
// Pinned, device-mapped host memory (the address is valid on the device too): lets a decode-sized call read its activations and
// write its result in place over PCIe instead of through two DMA transfers (sgemm_host.cpp, n = 1).
int lfamd_host_alloc(void **p, size_t bytes) {
    (void)hipGetLastError();
    HIPCHK(hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocPortable), "hipHostMalloc");
    return LFAMD_OK;
}
int lfamd_host_free(void *p) {
    if (p)
        HIPCHK(hipHostFree(p), "hipHostFree");
    return LFAMD_OK;
}
int lfamd_memset(void *dst, int value, size_t bytes, void *stream) {
    HIPCHK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream), "hipMemsetAsync");
    return LFAMD_OK;
}
int lfamd_stream_sync(void *stream) {
    HIPCHK(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
    return LFAMD_OK;
}

// ---------------------------------------------------------------------------------------------

static bool type_known(int t) {
    switch (t) {
    case LFAMD_TYPE_F32:
    case LFAMD_TYPE_F16:
    case LFAMD_TYPE_BF16:
    case LFAMD_TYPE_Q4_0:
    case LFAMD_TYPE_Q4_1:
    case LFAMD_TYPE_Q5_0:
    case LFAMD_TYPE_Q5_1:
    case LFAMD_TYPE_Q8_0:
    case LFAMD_TYPE_Q2_K:
    case LFAMD_TYPE_Q3_K:
    case LFAMD_TYPE_Q4_K:
    case LFAMD_TYPE_Q5_K:
    case LFAMD_TYPE_Q6_K:
    case LFAMD_TYPE_IQ4_XS:
        return true;
    default:
        return false;
    }
}

// Q8_0: bytes of the P80 image (256-aligned: the PC8 image starts behind it)
static size_t q80_p80_bytes(long rows, long cols) {
    return align_up((size_t)((rows + 7) / 8) * (size_t)((cols / 32 + 3) / 4) * P80_TILE, 256);
}

size_t lfamd_packed_size(int type, long rows, long cols) {
    if (!type_known(type) || rows < 0 || cols < 0 || cols % lfamd_blck_size(type))
        return 0;
    switch (type) {
    case LFAMD_TYPE_Q4_K:
        return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * P4K_TILE;
    case LFAMD_TYPE_Q4_0:
        if (cols % 256 == 0) // P40; other row lengths stay RAW (generic kernels)
            return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * P4K_TILE;
        return (size_t)rows * lfamd_row_size(type, cols);
    case LFAMD_TYPE_Q5_K:
        return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * P5K_TILE;
    case LFAMD_TYPE_Q6_K:
        return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * P6K_TILE;
    case LFAMD_TYPE_Q8_0: // P80, the ONE resident image (1.0625 bytes per weight, the file's): the bit-exact vecdot GEMV, the exact
                          // batch kernel and the f16 MFMA batch body (gemm_lf.hip) all read it.  A host that opted into the vendor GEMM
                          // (LFAMD_USE_BLASLT=1) also keeps the plain f16(d * q) rows that library needs (3.1 bytes per weight)
        if (lfamd_blaslt_ok())
            return q80_p80_bytes(rows, cols) + (size_t)rows * (size_t)cols * 2;
        return q80_p80_bytes(rows, cols);
    case LFAMD_TYPE_Q2_K:
    case LFAMD_TYPE_Q3_K: // PK2 / PK3: compact images (84 / 116 bytes per 256 weights) the decode GEMV reads; batches expand them
        return lfamd_pk_bytes(type, rows, cols); // into the canonical PCK image in the workspace, per call
    case LFAMD_TYPE_IQ4_XS: // codebook indices on the P4K nibble lattice + 16 header bytes per row (144 bytes per 256 weights, 1.06 x
                            // the file); batches expand it per call into the PC8 byte image in the workspace
        return (size_t)((rows + 31) / 32) * (size_t)(cols / 256) * P4K_TILE;
    case LFAMD_TYPE_Q4_1:
    case LFAMD_TYPE_Q5_0:
    case LFAMD_TYPE_Q5_1:
        if (cols % 256 == 0) // PCL; other row lengths stay RAW (generic kernels)
            return lfamd_wprep32_bytes(rows, cols);
        return (size_t)rows * lfamd_row_size(type, cols);
    default:
        return (size_t)rows * lfamd_row_size(type, cols);
    }
}

int lfamd_pack_weights(int type, long rows, long cols, const void *d_raw, size_t raw_row_bytes, void *d_packed,
                       void *stream) {
    (void)hipGetLastError(); // a stale error of an earlier call (e.g. an invalidated stream capture) must not fail this one
    if (!type_known(type))
        return fail(LFAMD_ERR_UNSUPPORTED, "pack_weights: unsupported ggml type%s", "");
    if (rows < 0 || cols < 0 || cols % lfamd_blck_size(type) || raw_row_bytes < lfamd_row_size(type, cols))
        return fail(LFAMD_ERR_INVALID, "pack_weights: bad shape%s", "");
    if (rows == 0 || cols == 0)
        return LFAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (type) {
    case LFAMD_TYPE_Q4_K:
        HIPCHK(lfamd_launch_pack_q4k(d_raw, raw_row_bytes, rows, cols, d_packed, s), "pack_q4k");
        break;
    case LFAMD_TYPE_Q4_0:
        if (cols % 256 == 0) {
            HIPCHK(lfamd_launch_pack_q40(d_raw, raw_row_bytes, rows, cols, d_packed, s), "pack_q40");
        } else {
            HIPCHK(lfamd_launch_pack_raw(d_raw, raw_row_bytes, rows, lfamd_row_size(type, cols), d_packed, s), "pack_raw");
        }
        break;
    case LFAMD_TYPE_Q5_K:
        HIPCHK(lfamd_launch_pack_q5k(d_raw, raw_row_bytes, rows, cols, d_packed, s), "pack_q5k");
        break;
    case LFAMD_TYPE_Q6_K:
        HIPCHK(lfamd_launch_pack_q6k(d_raw, raw_row_bytes, rows, cols, d_packed, s), "pack_q6k");
        break;
    case LFAMD_TYPE_Q8_0:
        HIPCHK(lfamd_launch_pack_q80(d_raw, raw_row_bytes, rows, cols, d_packed, s), "pack_q80");
        if (lfamd_blaslt_ok())
            HIPCHK(lfamd_launch_q80_image(d_raw, raw_row_bytes, rows, cols, (uint8_t *)d_packed + q80_p80_bytes(rows, cols), s), "pack_f16 (Q8_0)");
        break;
    case LFAMD_TYPE_Q2_K:
    case LFAMD_TYPE_Q3_K:
        HIPCHK(lfamd_launch_pk_pack(type, d_raw, raw_row_bytes, rows, cols, d_packed, s), "pack_pk");
        break;
    case LFAMD_TYPE_IQ4_XS:
        HIPCHK(lfamd_launch_pk4x_pack(d_raw, raw_row_bytes, rows, cols, d_packed, s), "pack_pk4x");
        break;
    case LFAMD_TYPE_Q4_1:
    case LFAMD_TYPE_Q5_0:
    case LFAMD_TYPE_Q5_1:
        if (cols % 256 == 0) {
            HIPCHK(lfamd_launch_wprep32(type, d_raw, raw_row_bytes, rows, cols, d_packed, s), "pack_pcl");
        } else {
            HIPCHK(lfamd_launch_pack_raw(d_raw, raw_row_bytes, rows, lfamd_row_size(type, cols), d_packed, s), "pack_raw");
        }
        break;
    default:
        HIPCHK(lfamd_launch_pack_raw(d_raw, raw_row_bytes, rows, lfamd_row_size(type, cols), d_packed, s), "pack_raw");
    }
    return LFAMD_OK;
}

int lfamd_scaled_gemm_ok(int type, long rows, long cols, const void *d_packed, void *stream) {
    (void)hipGetLastError();
    if (!type_known(type))
        return fail(LFAMD_ERR_UNSUPPORTED, "scaled_gemm_ok: unsupported ggml type%s", "");
    if ((type != LFAMD_TYPE_Q4_K && type != LFAMD_TYPE_Q5_K && type != LFAMD_TYPE_Q6_K) || rows <= 0 || cols <= 0)
        return 1;
    if (cols % 256 || !d_packed)
        return fail(LFAMD_ERR_INVALID, "scaled_gemm_ok: bad shape%s", "");
    hipStream_t s = (hipStream_t)stream;
    int *d_flag = nullptr, h_flag = 0;
    HIPCHK(hipMalloc(&d_flag, sizeof(int)), "hipMalloc");
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(int), s);
    if (e == hipSuccess)
        e = lfamd_launch_scaled_ok(type, rows, cols, d_packed, d_flag, s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)hipFree(d_flag);
    HIPCHK(e, "scaled_gemm_ok");
    return h_flag ? 0 : 1;
}

int lfamd_quantize_rows(int vec_dot_type, const float *d_x, long nrows, long cols, size_t x_row_bytes, void *d_y,
                        size_t y_row_bytes, void *stream) {
    (void)hipGetLastError(); // a stale error of an earlier call (e.g. an invalidated stream capture) must not fail this one
    if (vec_dot_type != LFAMD_TYPE_Q8_0 && vec_dot_type != LFAMD_TYPE_Q8_1 && vec_dot_type != LFAMD_TYPE_Q8_K)
        return fail(LFAMD_ERR_UNSUPPORTED, "quantize_rows: unsupported activation type%s", "");
    if (cols % lfamd_blck_size(vec_dot_type) || y_row_bytes < lfamd_row_size(vec_dot_type, cols))
        return fail(LFAMD_ERR_INVALID, "quantize_rows: bad shape%s", "");
    HIPCHK(lfamd_launch_quantize(vec_dot_type, d_x, nrows, cols, x_row_bytes, d_y, y_row_bytes, (hipStream_t)stream),
           "quantize_rows");
    return LFAMD_OK;
}

// ---------------------------------------------------------------------------------------------

// Q4_0 rows that are whole 256-weight groups are kept in the P40 layout and served by the tuned kernels
static bool packed40(int Atype, long k) {
    return Atype == LFAMD_TYPE_Q4_0 && k % 256 == 0;
}

static bool use_gemm(int Atype, long n, unsigned flags, long k) {
    if (flags & LFAMD_FLAG_FORCE_GENERIC)
        return false;
    return n > 8 && (Atype == LFAMD_TYPE_Q4_K || Atype == LFAMD_TYPE_Q5_K || Atype == LFAMD_TYPE_Q6_K || packed40(Atype, k));
}

// K-quants whose resident layout is the canonical image the MFMA body reads (Q2_K, Q3_K: PCK; IQ4_XS: PC8)
static bool use_gemm_canon(int Atype, long n, unsigned flags) {
    return !(flags & LFAMD_FLAG_FORCE_GENERIC) && n > 8 &&
           (Atype == LFAMD_TYPE_Q2_K || Atype == LFAMD_TYPE_Q3_K || Atype == LFAMD_TYPE_IQ4_XS);
}

static size_t gemm_act_ws(long k, long n) { // Xh + d8T + Xm of the K-quant GEMM
    size_t n_pad = align_up((size_t)n, 128), nb = (size_t)(k / 256);
    return align_up(n_pad * (size_t)k * 2, 256) + align_up(nb * n_pad * 4, 256) + align_up(n_pad * nb * 32, 256);
}

// legacy 32-block types whose rows are whole 256-weight groups (Q4_1, Q5_0, Q5_1): resident PCL image
static bool packed_pcl(int Atype, long k) {
    return k % 256 == 0 && (Atype == LFAMD_TYPE_Q4_1 || Atype == LFAMD_TYPE_Q5_0 || Atype == LFAMD_TYPE_Q5_1);
}
static bool use_gemm_canon32(int Atype, long n, unsigned flags, long k) {
    return !(flags & LFAMD_FLAG_FORCE_GENERIC) && n > 8 && packed_pcl(Atype, k);
}

// F16 / BF16 weights, batches, rows of whole 256-element groups: MFMA body straight on the RAW rows
static bool use_gemm_float(int Atype, long n, unsigned flags, long k) {
    return !(flags & LFAMD_FLAG_FORCE_GENERIC) && n > 8 && k % 256 == 0 && (Atype == LFAMD_TYPE_F16 || Atype == LFAMD_TYPE_BF16);
}

// Q8_0 batches.  Default (rows of whole 128-weight quads): this module's f16 MFMA body on the resident P80 image (gemm_lf.hip) —
// f16(d * q) x f16(d8 * q8), what the reference's GPU path computes for such a batch, <= 1e-3 (the north star's tolerance for
// f16 MFMA paths).  The north star asks for bit-exactness of the Q8_0 VECDOT (n <= 8: the GEMV), not for replaying tinyBLAS's
// 8-lane chains at n = 512; LFAMD_FLAG_PRECISE / LFAMD_FLAG_Q80_EXACT (or other row lengths): the register-tiled BIT-EXACT kernel
// (gemm_q80.hip), an order of magnitude slower.  A host that opted into the vendor library (LFAMD_USE_BLASLT=1) gets its f16 GEMM
// on a second resident image instead (use_gemm_q80_lt).
static bool use_gemm_q80_lt(int Atype, long n, unsigned flags, long k) {
    return !(flags & (LFAMD_FLAG_FORCE_GENERIC | LFAMD_FLAG_PRECISE | LFAMD_FLAG_Q80_EXACT)) && n > 8 && Atype == LFAMD_TYPE_Q8_0 && k % 32 == 0 &&
           lfamd_blaslt_ok();
}
static bool use_gemm_q80_lf(int Atype, long n, unsigned flags, long k) {
    return !(flags & (LFAMD_FLAG_FORCE_GENERIC | LFAMD_FLAG_PRECISE | LFAMD_FLAG_Q80_EXACT)) && n > 8 && Atype == LFAMD_TYPE_Q8_0 && k % 128 == 0 &&
           !lfamd_blaslt_ok();
}
static size_t gemm_lt_ws(long k, long n) { // the 16-bit activation rows, then the library's workspace
    return align_up((size_t)n * (size_t)k * 2, 256) + lfamd_blaslt_workspace();
}
static bool use_gemm_q80(int Atype, long n, unsigned flags, long k) {
    return !(flags & LFAMD_FLAG_FORCE_GENERIC) && n > 8 && Atype == LFAMD_TYPE_Q8_0 && !use_gemm_q80_lf(Atype, n, flags, k) &&
           !use_gemm_q80_lt(Atype, n, flags, k);
}

// Small batches of Q4_K / Q5_K / Q6_K (up to 32 tokens) on gemm_sb.hip, where it is the fastest route (MI355X; old -> new, us).
// Q4_K up to 8 tokens runs the int8-MFMA body, whose time barely depends on the token count (profiles/r03_small_batch_i8.txt):
//   4096 x 4096 7.8 .. 8.5 (GEMV: 6.6 at 2 tokens, 14.8 at 8), 14336 x 4096 ~ 13 (11.1 .. 24.6), 4096 x 14336 11.9 .. 13.3
//   (15.4 .. 45.4), 128256 x 4096 51 .. 54 (58 .. 138): from 2 tokens on deep rows and tall matrices, 3 with several row tiles
//   per CU, 4 with one.
// The f16 bodies (Q5_K, Q6_K, 9 .. 32 tokens; profiles/r03_small_batch.txt):
//   deep rows (k > 8192, ffn_down)          : n >= 3   (n = 8: 66 -> 20.3 Q6_K; n = 32: 28 -> 25.8)
//   at most one row tile per CU (m <= 8192) : n >= 5   (4096 x 4096: n = 32 16.9 -> 12.2)
//   up to four row tiles per CU             : 6 <= n <= 24, not Q6_K
// Below that the multi-column GEMV is faster (one launch, no staging pass); taller matrices keep the GEMV / the 128-token GEMM
// tiles.  The testing flags that force a GEMM body or the generic kernels keep their meaning.
static bool use_gemm_sb(int Atype, long n, unsigned flags, long k, long m) {
    if (flags & (LFAMD_FLAG_FORCE_GENERIC | LFAMD_FLAG_GEMM_NARROW | LFAMD_FLAG_GEMM_WIDE | LFAMD_FLAG_GEMM_PLAIN))
        return false;
    if (!lfamd_gemm_sb_ok(Atype, k, n))
        return false;
    static const bool force = getenv("LFAMD_SB_FORCE") != nullptr; // development: threshold sweeps
    if (force)
        return true;
    const long tiles_per_cu = ((m + 31) / 32 + 255) / 256;
    if (Atype == LFAMD_TYPE_Q4_K && n <= 8) // the int8 body
        return k > 8192 || tiles_per_cu > 4 ? n >= 2 : tiles_per_cu > 1 ? n >= 3 : n >= 4;
    if (k > 8192)
        return n >= 3;
    if (tiles_per_cu <= 1)
        return n >= 5;
    return tiles_per_cu <= 4 && Atype != LFAMD_TYPE_Q6_K && n >= 6 && n <= 24;
}

// Q4_K batches on the int8 matrix cores (gemm_i8.hip): exact integer dots; every launch whose 128 x 64 tiles fill at least half
// the CUs, unless a testing flag asks for one of the f16 bodies
static bool use_gemm_i8(int Atype, long n, unsigned flags, long k, long row_blocks) {
    if (flags & (LFAMD_FLAG_FORCE_GENERIC | LFAMD_FLAG_GEMM_NARROW | LFAMD_FLAG_GEMM_WIDE | LFAMD_FLAG_GEMM_PLAIN | LFAMD_FLAG_PRECISE))
        return false;
    return n > 8 && k > 0 && k % 256 == 0 && lfamd_gemm_i8_ok(Atype, row_blocks, n);
}

static bool use_gemv(int Atype, long n, unsigned flags, long k) {
    if (flags & LFAMD_FLAG_FORCE_GENERIC)
        return false;
    return n <= 8 && (Atype == LFAMD_TYPE_Q4_K || Atype == LFAMD_TYPE_Q5_K || Atype == LFAMD_TYPE_Q6_K ||
                      Atype == LFAMD_TYPE_Q8_0 || Atype == LFAMD_TYPE_Q2_K || Atype == LFAMD_TYPE_Q3_K ||
                      Atype == LFAMD_TYPE_IQ4_XS || packed40(Atype, k) || packed_pcl(Atype, k));
}

static bool gemv_quantise_separately(int Atype, long m) {
    (void)Atype;
    (void)m;
    return false; // persistent GEMV work-groups stage the activations once each: fused is always cheaper
}

int lfamd_mul_mat_is_exact(int Atype, long m, long k, long n, unsigned flags) {
    if (!type_known(Atype) || m <= 0 || k <= 0 || n <= 0)
        return 0;
    const bool kq = Atype == LFAMD_TYPE_Q4_K || Atype == LFAMD_TYPE_Q5_K || Atype == LFAMD_TYPE_Q6_K;
    if (n <= 8 || !kq)
        return Atype != LFAMD_TYPE_Q8_0 || n <= 8 || !(use_gemm_q80_lt(Atype, n, flags, k) || use_gemm_q80_lf(Atype, n, flags, k)) ? 1 : 0;
    if (use_gemm_sb(Atype, n, flags, k, m))
        return 1;
    if (Atype == LFAMD_TYPE_Q6_K)
        return 0; // (both batch bodies round sc * (q - 32) above 2048)
    if (flags & (LFAMD_FLAG_PRECISE | LFAMD_FLAG_GEMM_NARROW | LFAMD_FLAG_GEMM_PLAIN))
        return 1;
    return use_gemm_i8(Atype, n, flags, k, (m + 127) / 128) ? 1 : (lfamd_gemm_wide_scaled_ok(Atype, 0) ? 0 : 1);
}

// Which body a K-quant batch runs when it is not the int8 one.  Two families: 128 x 128 / 256 x 128 tiles with K streamed once
// (gemm_wide.hip and its loader-wave / K-split-wave / row-split descendants) when that grid fills the 256 CUs; the 128 x 64 split-K body
// (gemm_mfma.hip, exact codes) for smaller grids; LFAMD_GEMM_BODY=narrow|wide forces one.  The wide family runs scaled operands (one
// f16 rounding each, ~1e-4 relative) unless the caller wants the exact integer-code arithmetic (LFAMD_FLAG_PRECISE); it has a 128 x 64
// tile for the grids the 128 x 128 tile cannot fill, so it also replaces the split-K body down to LW_MIN_TILES.  Returns `scaled`.
static bool gemm_body_choice(int Atype, long m, long n, unsigned flags, bool &narrow) {
    static const char *body = getenv("LFAMD_GEMM_BODY");
    const int plain = (flags & LFAMD_FLAG_GEMM_PLAIN) ? 1 : 0;
    const long tiles128 = ((m + 127) / 128) * (long)(align_up((size_t)n, 128) / 128);
    const int can_scale = !(flags & LFAMD_FLAG_PRECISE) && lfamd_gemm_wide_scaled_ok(Atype, plain);
    narrow = (flags & LFAMD_FLAG_GEMM_NARROW) ? true
             : (flags & LFAMD_FLAG_GEMM_WIDE) ? false
             : body                           ? body[0] == 'n'
                                              : (tiles128 < 192 && !(can_scale && tiles128 >= LW_MIN_TILES));
    return !narrow && can_scale;
}

// Does a call accept the scaled-operand staged image a fused producer wrote (LFAMD_TYPE_STAGED_SCALED)?  The K-quant batches whose
// body reads it: what lfamd_mul_mat would stage with prep_scaled_kernel itself.
int lfamd_mul_mat_takes_staged_scaled(int Atype, long m, long k, long n, unsigned flags) {
    if (!type_known(Atype) || m <= 0 || k <= 0 || n <= 0 || Atype == LFAMD_TYPE_Q4_0)
        return 0;
    bool narrow;
    return !(flags & LFAMD_FLAG_FORCE_GENERIC) && !use_gemm_sb(Atype, n, flags, k, m) && use_gemm(Atype, n, flags, k) &&
                   !use_gemm_i8(Atype, n, flags, k, (m + 127) / 128) && gemm_body_choice(Atype, m, n, flags, narrow)
               ? 1
               : 0;
}

struct scaled_image_ptrs {
    const void *Xh, *d8T, *Xm;
    size_t n_pad;
};
static scaled_image_ptrs scaled_image_of(const void *image, long k, long n) {
    scaled_image_ptrs p;
    p.n_pad = align_up((size_t)n, 128);
    const size_t nb = (size_t)(k / 256);
    p.Xh = image;
    p.d8T = (const uint8_t *)image + align_up(p.n_pad * (size_t)k * 2, 256);
    p.Xm = (const uint8_t *)p.d8T + align_up(nb * p.n_pad * 4, 256);
    return p;
}

// sibling matrices of one type on one launch of the int8 body: their row blocks together must make a grid it takes
static bool multi_i8_ok(int Atype, int count, const long *m, const long *ldc, long k, long n, unsigned flags) {
    long rbs = 0;
    for (int j = 0; j < count; j++) {
        if (m[j] < 0 || ldc[j] < m[j])
            return false;
        rbs += (m[j] + 127) / 128;
    }
    return rbs > 0 && use_gemm(Atype, n, flags, k) && use_gemm_i8(Atype, n, flags, k, rbs);
}

// Does a call accept the staged image a fused producer wrote (LFAMD_TYPE_STAGED_Q8K)?  Exactly the calls that run the int8 body.
int lfamd_mul_mat_takes_staged(int Atype, long m, long k, long n, unsigned flags) {
    if (!type_known(Atype) || m <= 0 || k <= 0 || n <= 0)
        return 0;
    return !use_gemm_sb(Atype, n, flags, k, m) && use_gemm(Atype, n, flags, k) && use_gemm_i8(Atype, n, flags, k, (m + 127) / 128) ? 1 : 0;
}

static size_t mul_mat_workspace_base(int Atype, long m, long k, long n) {
    if (use_gemv(Atype, n, 0, k) && gemv_quantise_separately(Atype, m))
        return align_up((size_t)n * lfamd_row_size(lfamd_vec_dot_type(Atype), k), 256);
    if (use_gemm(Atype, n, 0, k)) {
        size_t n_pad = align_up((size_t)n, 128), nb = (size_t)(k / 256);
        return align_up(n_pad * (size_t)k * 2, 256) + align_up(nb * n_pad * 4, 256) + align_up(n_pad * nb * 32, 256) +
               (Atype == LFAMD_TYPE_Q4_0 ? 0 : lfamd_gemm_lw_ksplit_bytes(m, n)); // partial tiles of a K-split launch
    }
    if (Atype == LFAMD_TYPE_Q8_0 && n > 8) { // (either body may be asked for through the flags: the largest)
        const size_t exact = align_up(lfamd_gemm_q80_workspace(k, n), 256);
        const size_t mfma = use_gemm_q80_lt(Atype, n, 0, k) ? gemm_lt_ws(k, n) : k % 128 == 0 ? align_up(lfamd_gemm_lf_workspace(k, n), 256) : 0;
        return exact > mfma ? exact : mfma;
    }
    if (use_gemm_float(Atype, n, 0, k)) {
        const size_t own = align_up(align_up((size_t)n, 128) * (size_t)k * 2, 256);
        const size_t lt = lfamd_blaslt_ok() ? gemm_lt_ws(k, n) : 0;
        return own > lt ? own : lt;
    }
    if (use_gemm_canon(Atype, n, 0)) // (+ the canonical image of a Q2_K / Q3_K matrix, rebuilt from the compact one per call)
        return gemm_act_ws(k, n) + align_up(Atype == LFAMD_TYPE_IQ4_XS ? lfamd_wprep8_bytes(m, k) : lfamd_wprep16_bytes(m, k), 256);
    if (use_gemm_canon32(Atype, n, 0, k)) { // Xh, d8T [nb*8][n_pad], sT [nb*8][n_pad], image
        size_t n_pad = align_up((size_t)n, 128), nb = (size_t)(k / 256);
        return align_up(n_pad * (size_t)k * 2, 256) + 2 * align_up(nb * 8 * n_pad * 4, 256);
    }
    if (use_gemv(Atype, n, 0, k) || !type_known(Atype) || lfamd_blck_size(Atype) == 1)
        return 0;
    // generic kernels given f32 activations quantise them into the workspace first
    return align_up((size_t)n * lfamd_row_size(lfamd_vec_dot_type(Atype), k), 256);
}

size_t lfamd_mul_mat_workspace(int Atype, long m, long k, long n) {
    const size_t base = mul_mat_workspace_base(Atype, m, k, n); // (the body a testing flag may force)
    const size_t sb = use_gemm_sb(Atype, n, 0, k, m) ? align_up(lfamd_gemm_sb_workspace(k), 256) : 0;
    return base > sb ? base : sb;
}

// The largest workspace any batch of 1 .. n rows can ask for (which body serves a batch depends on n, so the size is not monotonic
// in n: a K-split launch of few tiles keeps partial tiles a larger batch does not need).  For callers that size ONE buffer for
// per-group calls of varying n (csrc/moe.hip).  Evaluated at n and at every point where a body's choice or tile count can change.
size_t lfamd_mul_mat_workspace_upto(int Atype, long m, long k, long n) {
    size_t best = 0;
    auto take = [&](long v) {
        if (v >= 1 && v <= n) {
            const size_t w = lfamd_mul_mat_workspace(Atype, m, k, v);
            best = w > best ? w : best;
        }
    };
    take(n);
    for (long v : {1L, 2L, 3L, 4L, 5L, 6L, 7L, 8L, 9L, 16L, 17L, 32L, 33L})
        take(v);
    for (long v = 64; v <= n + 63; v += 64) // token tiles of 64 and 128: the last batch of a tile count and the first of the next
        take(v), take(v + 1);
    return best;
}

int lfamd_mul_mat(int Atype, const void *d_A, long m, long k, int Btype, const void *d_B, size_t b_row_bytes, long n,
                  float *d_C, long ldc, void *d_ws, size_t ws_bytes, unsigned flags, void *stream) {
    (void)hipGetLastError(); // a stale error of an earlier call (e.g. an invalidated stream capture) must not fail this one
    const int plain = (flags & LFAMD_FLAG_GEMM_PLAIN) ? 1 : 0;
    if (!type_known(Atype))
        return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat: unsupported weight type%s", "");
    if (m < 0 || n < 0 || k < 0 || ldc < m || k % lfamd_blck_size(Atype))
        return fail(LFAMD_ERR_INVALID, "mul_mat: bad shape%s", "");
    if (Btype == LFAMD_TYPE_STAGED_Q8K) { // a fused producer wrote the int8 body's staged image: the GEMM alone, no staging launch
        if (m == 0 || n == 0)
            return LFAMD_OK;
        if (!lfamd_mul_mat_takes_staged(Atype, m, k, n, flags))
            return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat: this call does not run the int8 batch body (lfamd_mul_mat_takes_staged)%s", "");
        const void *A1 = d_A;
        HIPCHK(lfamd_launch_gemm_i8_staged(1, &A1, &m, k, d_B, n, &d_C, &ldc, (hipStream_t)stream), "gemm_i8 (staged input)");
        return LFAMD_OK;
    }
    if (Btype == LFAMD_TYPE_STAGED_SCALED) { // a fused producer wrote the scaled-operand bodies' staged image
        if (m == 0 || n == 0)
            return LFAMD_OK;
        if (!lfamd_mul_mat_takes_staged_scaled(Atype, m, k, n, flags) || !d_B || ((uintptr_t)d_B & 15))
            return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat: this call does not run a scaled-operand batch body (lfamd_mul_mat_takes_staged_scaled)%s", "");
        const size_t part = lfamd_gemm_lw_ksplit_bytes(m, n); // partial tiles of a K-split launch: the only workspace left
        if (part && (ws_bytes < part || !d_ws))
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        const scaled_image_ptrs im = scaled_image_of(d_B, k, n);
        HIPCHK(lfamd_launch_gemm_wide(Atype, d_A, m, k, im.Xh, im.d8T, im.Xm, n, (long)im.n_pad, d_C, ldc, plain | 2, d_ws, ws_bytes, (hipStream_t)stream),
               "gemm_wide (staged input)");
        return LFAMD_OK;
    }
    const int vdt = lfamd_vec_dot_type(Atype);
    const bool float_a = Atype == LFAMD_TYPE_F32 || Atype == LFAMD_TYPE_F16 || Atype == LFAMD_TYPE_BF16;
    if (float_a) {
        if (!(Btype == LFAMD_TYPE_F32 || Btype == Atype))
            return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat: float weights need F32 or same-type activations%s", "");
    } else if (Btype != vdt && Btype != LFAMD_TYPE_F32) {
        // f32 activations (the GGML_OP_MUL_MAT boundary) are quantised on the device to vdt
        return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat: activations must be F32 or the weight type's vec_dot format%s", "");
    }
    if (b_row_bytes < lfamd_row_size(Btype, k))
        return fail(LFAMD_ERR_INVALID, "mul_mat: activation row stride too small%s", "");
    if (m == 0 || n == 0)
        return LFAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const int vregs32 = (flags & LFAMD_FLAG_Q0_VREGS32) ? 1 : 0, precise = (flags & LFAMD_FLAG_PRECISE) ? 1 : 0;

    if (use_gemm_sb(Atype, n, flags, k, m)) { // a handful of tokens: weights streamed once, MFMA tile of 32 token slots (gemm_sb.hip)
        if (ws_bytes < align_up(lfamd_gemm_sb_workspace(k), 256) || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        HIPCHK(lfamd_launch_gemm_sb(Atype, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, d_ws, 0, s), "gemm_sb");
        return LFAMD_OK;
    }
    if (use_gemm(Atype, n, flags, k)) {
        size_t need = lfamd_mul_mat_workspace(Atype, m, k, n);
        if (ws_bytes < need || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        if (use_gemm_i8(Atype, n, flags, k, (m + 127) / 128) && (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_K)) {
            const void *A1 = d_A;
            HIPCHK(lfamd_launch_gemm_i8(1, &A1, &m, k, Btype, d_B, b_row_bytes, n, &d_C, &ldc, d_ws, nullptr, s), "gemm_i8");
            return LFAMD_OK;
        }
        size_t n_pad = align_up((size_t)n, 128), nb = (size_t)(k / 256);
        uint8_t *ws = (uint8_t *)d_ws;
        void *Xh = ws;
        void *d8T = ws + align_up(n_pad * (size_t)k * 2, 256);
        void *Xm = (uint8_t *)d8T + align_up(nb * n_pad * 4, 256);
        if (Atype == LFAMD_TYPE_Q4_0) { // Q8_0-quantised activations, eight scales per 256 (they take the Xm area too)
            HIPCHK(lfamd_launch_prep80(Btype, d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, nullptr, s), "prep80");
            HIPCHK(lfamd_launch_gemm_wide(Atype, d_A, m, k, Xh, d8T, nullptr, n, (long)n_pad, d_C, ldc, plain, nullptr, 0, s), "gemm_wide");
            return LFAMD_OK;
        }
        // two bodies: 128 x 128 tiles, K streamed once (gemm_wide.hip) when that grid fills the 256 CUs; the
        // 128 x 64 split-K body (gemm_mfma.hip) for smaller grids.  LFAMD_GEMM_BODY=narrow|wide forces one.
        bool narrow;
        const int scaled = gemm_body_choice(Atype, m, n, flags, narrow) ? 1 : 0;
        if (Btype == LFAMD_TYPE_F32)
            HIPCHK(lfamd_launch_prep_f32(d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, Xm, scaled ? 2 : 0, nullptr, s), "prep_f32");
        else
            HIPCHK(lfamd_launch_prep_q8k(d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, Xm, scaled ? 2 : 0, nullptr, s), "prep_q8k");
        if (narrow) {
            HIPCHK(lfamd_launch_gemm_kq(Atype, d_A, m, k, Xh, d8T, Xm, n, (long)n_pad, d_C, ldc, s), "gemm_kq");
        } else {
            uint8_t *Pp = (uint8_t *)Xm + align_up(n_pad * nb * 32, 256); // after Xh, d8T, Xm (lfamd_mul_mat_workspace)
            HIPCHK(lfamd_launch_gemm_wide(Atype, d_A, m, k, Xh, d8T, Xm, n, (long)n_pad, d_C, ldc, plain | (scaled << 1), Pp,
                                          (size_t)((uint8_t *)d_ws + ws_bytes - Pp), s),
                   "gemm_wide");
        }
        return LFAMD_OK;
    }
    if (use_gemm_float(Atype, n, flags, k)) {
        size_t need = lfamd_mul_mat_workspace(Atype, m, k, n);
        if (ws_bytes < need || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        // plain 16-bit float weights: the vendor's GEMM (blaslt.hip) unless a testing flag asks for this module's body
        if (lfamd_blaslt_ok() && !(flags & (LFAMD_FLAG_GEMM_WIDE | LFAMD_FLAG_GEMM_NARROW | LFAMD_FLAG_GEMM_PLAIN)) && ((uintptr_t)d_A & 15) == 0) {
            uint8_t *ws8 = (uint8_t *)d_ws;
            const void *X16 = d_B;
            long ldx = (long)(b_row_bytes / 2);
            if (Btype == LFAMD_TYPE_F32 || ((uintptr_t)d_B & 15) || (b_row_bytes & 15)) {
                if (Btype != LFAMD_TYPE_F32)
                    goto own_float_body; // (unaligned 16-bit rows: rare; the module's kernel takes them)
                HIPCHK(lfamd_launch_rows_to_16(Atype, d_B, b_row_bytes, n, k, ws8, s), "rows_to_16");
                X16 = ws8, ldx = k;
            }
            uint8_t *ltws = ws8 + align_up((size_t)n * (size_t)k * 2, 256);
            if (lfamd_blaslt_gemm(Atype, d_A, k, X16, ldx, m, n, k, d_C, ldc, ltws, lfamd_blaslt_workspace(), s) == hipSuccess)
                return LFAMD_OK;
            (void)hipGetLastError(); // the library declined this shape: this module's body
        }
    own_float_body:
        size_t n_pad = align_up((size_t)n, 128);
        HIPCHK(lfamd_launch_prep_float(Atype, Btype, d_B, b_row_bytes, n, (long)n_pad, k, d_ws, s), "prep_float");
        // default: the loader-wave body on the RAW rows (gemm_lf.hip); the 128 x 128 wide body on request (testing flags), for
        // unaligned tensors and for matrices beyond a 32-bit byte offset
        const size_t a_row = lfamd_row_size(Atype, k);
        if (!(flags & (LFAMD_FLAG_GEMM_WIDE | LFAMD_FLAG_GEMM_NARROW | LFAMD_FLAG_GEMM_PLAIN)) && ((uintptr_t)d_A & 15) == 0 &&
            (size_t)m * a_row < ((size_t)1 << 32)) {
            HIPCHK(lfamd_launch_gemm_lf_float(Atype, d_A, a_row, m, k, d_ws, n, (long)n_pad, d_C, ldc, s), "gemm_lf (float)");
            return LFAMD_OK;
        }
        HIPCHK(lfamd_launch_gemm_wide(Atype, d_A, m, k, d_ws, d_ws, d_ws, n, (long)n_pad, d_C, ldc, plain, nullptr, 0, s), "gemm_wide");
        return LFAMD_OK;
    }
    if (float_a && n <= 8 && !(flags & LFAMD_FLAG_FORCE_GENERIC) && lfamd_gemv_float_ok(Atype, k, n) &&
        ((uintptr_t)d_A & 15) == 0 && ((uintptr_t)d_B & 15) == 0 && (b_row_bytes & 15) == 0) { // decode on float weights (16-byte loads)
        HIPCHK(lfamd_launch_gemv_float(Atype, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, s), "gemv_float");
        return LFAMD_OK;
    }
    if (use_gemm_canon32(Atype, n, flags, k)) {
        size_t need = lfamd_mul_mat_workspace(Atype, m, k, n);
        if (ws_bytes < need || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        size_t n_pad = align_up((size_t)n, 128), nb = (size_t)(k / 256);
        uint8_t *ws = (uint8_t *)d_ws;
        void *Xh = ws;
        void *d8T = ws + align_up(n_pad * (size_t)k * 2, 256);
        void *sT = (uint8_t *)d8T + align_up(nb * 8 * n_pad * 4, 256);
        const bool q81 = vdt == LFAMD_TYPE_Q8_1;
        HIPCHK(lfamd_launch_prep80(Btype, d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, q81 ? sT : nullptr, s), "prep80");
        HIPCHK(lfamd_launch_gemm_wide(Atype, d_A, m, k, Xh, d8T, q81 ? sT : nullptr, n, (long)n_pad, d_C, ldc, plain, nullptr, 0, s), "gemm_wide");
        return LFAMD_OK;
    }
    if (use_gemm_canon(Atype, n, flags)) {
        size_t need = lfamd_mul_mat_workspace(Atype, m, k, n);
        if (ws_bytes < need || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        size_t n_pad = align_up((size_t)n, 128), nb = (size_t)(k / 256);
        uint8_t *ws = (uint8_t *)d_ws;
        void *Xh = ws;
        void *d8T = ws + align_up(n_pad * (size_t)k * 2, 256);
        void *Xm = (uint8_t *)d8T + align_up(nb * n_pad * 4, 256);
        const int mins16 = Atype == LFAMD_TYPE_Q2_K;
        if (Btype == LFAMD_TYPE_F32)
            HIPCHK(lfamd_launch_prep_f32(d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, Xm, mins16, nullptr, s), "prep_f32");
        else
            HIPCHK(lfamd_launch_prep_q8k(d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, Xm, mins16, nullptr, s), "prep_q8k");
        // the resident image is the compact one; the MFMA body reads the canonical form, rebuilt here per call
        void *img = ws + gemm_act_ws(k, n);
        if (Atype == LFAMD_TYPE_IQ4_XS)
            HIPCHK(lfamd_launch_pk4x_expand(d_A, m, k, img, s), "pk4x_expand");
        else
            HIPCHK(lfamd_launch_pk_expand(Atype, d_A, m, k, img, s), "pk_expand");
        HIPCHK(lfamd_launch_gemm_wide(Atype, img, m, k, Xh, d8T, Xm, n, (long)n_pad, d_C, ldc, plain, nullptr, 0, s), "gemm_wide");
        return LFAMD_OK;
    }
    if (use_gemm_q80_lt(Atype, n, flags, k)) {
        if (ws_bytes < gemm_lt_ws(k, n) || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        uint8_t *ws8 = (uint8_t *)d_ws;
        const void *img = (const uint8_t *)d_A + q80_p80_bytes(m, k); // f16(d * q) rows, built once by lfamd_pack_weights
        HIPCHK(lfamd_launch_q80_rows_to_f16(Btype, d_B, b_row_bytes, n, k, ws8, s), "q80_rows_to_f16");
        if (lfamd_blaslt_gemm(LFAMD_TYPE_F16, img, k, ws8, k, m, n, k, d_C, ldc, ws8 + align_up((size_t)n * (size_t)k * 2, 256),
                              lfamd_blaslt_workspace(), s) == hipSuccess)
            return LFAMD_OK;
        // the library declined this shape or this device: the bit-exact kernel on the resident P80 image (the workspace is sized for
        // either, mul_mat_workspace_base)
        (void)hipGetLastError();
        if (ws_bytes < align_up(lfamd_gemm_q80_workspace(k, n), 256))
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        HIPCHK(lfamd_launch_gemm_q80(d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, d_ws, vregs32, precise, s), "gemm_q80 (library declined)");
        return LFAMD_OK;
    }
    if (use_gemm_q80_lf(Atype, n, flags, k) && (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_0)) {
        if (ws_bytes < lfamd_gemm_lf_workspace(k, n) || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        const void *A1 = d_A;
        HIPCHK(lfamd_launch_gemm_lf_q80(1, &A1, &m, k, Btype, d_B, b_row_bytes, n, &d_C, &ldc, d_ws, s), "gemm_lf (Q8_0)");
        return LFAMD_OK;
    }
    if (use_gemm_q80(Atype, n, flags, k)) {
        size_t need = align_up(lfamd_gemm_q80_workspace(k, n), 256);
        if (ws_bytes < need || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        HIPCHK(lfamd_launch_gemm_q80(d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, d_ws, vregs32, precise, s), "gemm_q80");
        return LFAMD_OK;
    }
    if (use_gemv(Atype, n, flags, k)) {
        if (Btype == LFAMD_TYPE_F32 && gemv_quantise_separately(Atype, m)) {
            // very tall matrices (output.weight): thousands of work-groups would each re-quantise the same
            // activation vector; quantise it once into the workspace instead
            size_t qrow = lfamd_row_size(vdt, k), need = align_up((size_t)n * qrow, 256);
            if (ws_bytes < need || !d_ws)
                return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
            HIPCHK(lfamd_launch_quantize(vdt, (const float *)d_B, n, k, b_row_bytes, d_ws, qrow, s), "quantize_rows");
            HIPCHK(lfamd_launch_gemv(Atype, d_A, m, k, vdt, d_ws, qrow, n, d_C, ldc, vregs32, precise, s), "gemv");
            return LFAMD_OK;
        }
        HIPCHK(lfamd_launch_gemv(Atype, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, vregs32, precise, s), "gemv");
        return LFAMD_OK;
    }
    if (Atype == LFAMD_TYPE_Q4_K || Atype == LFAMD_TYPE_Q5_K || Atype == LFAMD_TYPE_Q6_K || Atype == LFAMD_TYPE_Q8_0 ||
        Atype == LFAMD_TYPE_Q2_K || Atype == LFAMD_TYPE_Q3_K || Atype == LFAMD_TYPE_IQ4_XS || packed40(Atype, k) || packed_pcl(Atype, k))
        return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat: FORCE_GENERIC needs RAW-layout weights; this type is packed%s", "");
    if (!float_a && Btype == LFAMD_TYPE_F32) {
        size_t qrow = lfamd_row_size(vdt, k), need = align_up((size_t)n * qrow, 256);
        if (ws_bytes < need || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat: workspace too small%s", "");
        HIPCHK(lfamd_launch_quantize(vdt, (const float *)d_B, n, k, b_row_bytes, d_ws, qrow, s), "quantize_rows");
        HIPCHK(lfamd_launch_generic(Atype, d_A, m, k, vdt, d_ws, qrow, n, d_C, ldc, s), "generic");
        return LFAMD_OK;
    }
    HIPCHK(lfamd_launch_generic(Atype, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, s), "generic");
    return LFAMD_OK;
}

// Sibling mat-muls on the same activations whose weight types may differ (a backend's graph_compute sees attn_q/k/v as
// three MUL_MAT nodes with one src1; in a Q4_K_M file q and k are Q4_K, v is Q6_K).  Decode (n = 1) with exactly two
// K-quant types {Q4_K | Q5_K, Q6_K}: ONE launch (gemv_kq_dual_kernel).  Everything else: one lfamd_mul_mat_multi per
// run of equal types.
int lfamd_mul_mat_multi_types(int count, const int *Atype, const void *const *d_A, const long *m, long k, int Btype,
                              const void *d_B, size_t b_row_bytes, long n, float *const *d_C, const long *ldc, void *d_ws,
                              size_t ws_bytes, unsigned flags, void *stream) {
    (void)hipGetLastError();
    if (count <= 0)
        return LFAMD_OK;
    if (!Atype || !d_A || !m || !d_C || !ldc)
        return fail(LFAMD_ERR_INVALID, "mul_mat_multi_types: null argument%s", "");
    int ta = -1, tb = -1, na = 0, nb_ = 0;
    const void *Aa[4], *Ab[4];
    long ma[4], mb[4], la[4], lb[4];
    float *Ca[4], *Cb[4];
    bool dual = n == 1 && count <= 8 && k > 0 && k % 256 == 0 && !(flags & LFAMD_FLAG_FORCE_GENERIC) &&
                (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_K) && b_row_bytes >= lfamd_row_size(Btype, k);
    for (int j = 0; j < count && dual; j++) {
        const int t = Atype[j];
        if (m[j] <= 0 || ldc[j] < m[j]) {
            dual = false;
        } else if (t == LFAMD_TYPE_Q6_K) {
            if (nb_ == 4)
                dual = false;
            else
                Ab[nb_] = d_A[j], mb[nb_] = m[j], lb[nb_] = ldc[j], Cb[nb_] = d_C[j], nb_++, tb = t;
        } else if ((t == LFAMD_TYPE_Q4_K || t == LFAMD_TYPE_Q5_K) && (ta < 0 || ta == t)) {
            if (na == 4)
                dual = false;
            else
                Aa[na] = d_A[j], ma[na] = m[j], la[na] = ldc[j], Ca[na] = d_C[j], na++, ta = t;
        } else {
            dual = false;
        }
    }
    if (dual && na > 0 && nb_ > 0) {
        HIPCHK(lfamd_launch_gemv_dual(ta, na, Aa, ma, Ca, la, tb, nb_, Ab, mb, Cb, lb, k, Btype, d_B, b_row_bytes,
                                      (hipStream_t)stream),
               "gemv_dual");
        return LFAMD_OK;
    }
    // Batches whose nodes are all K-quants with a resident layout (attn_q/k = Q4_K with attn_v = Q6_K at prefill): the
    // scaled-operand GEMM of every type reads the SAME staged activations, so they are prepared once; then one launch of
    // the loader-wave body per run of equal types.
    {
        const bool staged_in = Btype == LFAMD_TYPE_STAGED_SCALED; // (a fused producer wrote the scaled image: d_B, 16-byte aligned)
        bool share = n > 8 && count > 1 && k > 0 && k % 256 == 0 && !(flags & (LFAMD_FLAG_PRECISE | LFAMD_FLAG_FORCE_GENERIC |
                                                                                LFAMD_FLAG_GEMM_NARROW)) &&
                     (staged_in ? d_B && ((uintptr_t)d_B & 15) == 0
                                : (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_K) && b_row_bytes >= lfamd_row_size(Btype, k));
        const int plain = (flags & LFAMD_FLAG_GEMM_PLAIN) ? 1 : 0;
        bool mixed = false;
        for (int j = 0; j < count && share; j++) {
            const int t = Atype[j];
            share = (t == LFAMD_TYPE_Q4_K || t == LFAMD_TYPE_Q5_K || t == LFAMD_TYPE_Q6_K) && lfamd_gemm_wide_scaled_ok(t, plain) &&
                    m[j] >= 0 && ldc[j] >= m[j];
            mixed = mixed || t != Atype[0];
        }
        const size_t n_pad = align_up((size_t)n, 128), nbk = (size_t)(k / 256);
        const size_t need = align_up(n_pad * (size_t)k * 2, 256) + align_up(nbk * n_pad * 4, 256) + align_up(n_pad * nbk * 32, 256);
        if (share && mixed && (staged_in || (d_ws && ws_bytes >= need))) {
            hipStream_t s = (hipStream_t)stream;
            uint8_t *ws = staged_in ? (uint8_t *)const_cast<void *>(d_B) : (uint8_t *)d_ws; // (the image has the workspace's layout)
            void *Xh = ws;
            void *d8T = ws + align_up(n_pad * (size_t)k * 2, 256);
            void *Xm = (uint8_t *)d8T + align_up(nbk * n_pad * 4, 256);
            if (staged_in)
                ; // nothing to stage
            else if (Btype == LFAMD_TYPE_F32)
                HIPCHK(lfamd_launch_prep_f32(d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, Xm, 2, nullptr, s), "prep_f32");
            else
                HIPCHK(lfamd_launch_prep_q8k(d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, Xm, 2, nullptr, s), "prep_q8k");
            { // {Q4_K | Q5_K, Q6_K} whose tiles fill more than half the chip in one round: one launch for both types
                const void *Aa[4], *Ab[4];
                long ma_[4], mb_[4], la[4], lb[4];
                float *Ca[4], *Cb[4];
                int ta = -1, na = 0, nbq = 0;
                bool two = true;
                for (int j = 0; j < count && two; j++) {
                    if (Atype[j] == LFAMD_TYPE_Q6_K) {
                        two = nbq < 4;
                        if (two)
                            Ab[nbq] = d_A[j], mb_[nbq] = m[j], lb[nbq] = ldc[j], Cb[nbq] = d_C[j], nbq++;
                    } else if (ta < 0 || ta == Atype[j]) {
                        two = na < 4;
                        if (two)
                            Aa[na] = d_A[j], ma_[na] = m[j], la[na] = ldc[j], Ca[na] = d_C[j], na++, ta = Atype[j];
                    } else {
                        two = false;
                    }
                }
                if (two && na > 0 && nbq > 0) {
                    hipError_t e = lfamd_launch_gemm_wide_dual(ta, na, Aa, ma_, Ca, la, LFAMD_TYPE_Q6_K, nbq, Ab, mb_, Cb, lb, k, Xh, d8T,
                                                               Xm, n, (long)n_pad, plain | 2, s);
                    if (e == hipSuccess)
                        return LFAMD_OK;
                    if (e != hipErrorNotSupported)
                        HIPCHK(e, "gemm_wide_dual");
                }
            }
            for (int j0 = 0; j0 < count;) {
                int j1 = j0 + 1;
                while (j1 < count && Atype[j1] == Atype[j0] && j1 - j0 < 4)
                    j1++;
                HIPCHK(lfamd_launch_gemm_wide_multi(Atype[j0], j1 - j0, d_A + j0, m + j0, k, Xh, d8T, Xm, n, (long)n_pad, d_C + j0,
                                                    ldc + j0, plain | 2, nullptr, 0, s),
                       "gemm_wide_multi");
                j0 = j1;
            }
            return LFAMD_OK;
        }
    }
    for (int j0 = 0; j0 < count;) { // runs of equal types
        int j1 = j0 + 1;
        while (j1 < count && Atype[j1] == Atype[j0] && j1 - j0 < 4)
            j1++;
        int r = lfamd_mul_mat_multi(Atype[j0], j1 - j0, d_A + j0, m + j0, k, Btype, d_B, b_row_bytes, n, d_C + j0, ldc + j0, d_ws,
                                    ws_bytes, flags, stream);
        if (r != LFAMD_OK)
            return r;
        j0 = j1;
    }
    return LFAMD_OK;
}

int lfamd_mul_mat_multi(int Atype, int count, const void *const *d_A, const long *m, long k, int Btype, const void *d_B,
                        size_t b_row_bytes, long n, float *const *d_C, const long *ldc, void *d_ws, size_t ws_bytes,
                        unsigned flags, void *stream) {
    (void)hipGetLastError(); // a stale error of an earlier call (e.g. an invalidated stream capture) must not fail this one
    const int plain = (flags & LFAMD_FLAG_GEMM_PLAIN) ? 1 : 0;
    if (count <= 0)
        return LFAMD_OK;
    if (Btype == LFAMD_TYPE_STAGED_Q8K) { // sibling matrices on one staged image: every one of them must take it
        if (n == 0)
            return LFAMD_OK;
        if (!(count > 1 && count <= 4 && multi_i8_ok(Atype, count, m, ldc, k, n, flags))) // (else: each matrix by itself)
            for (int j = 0; j < count; j++)
                if (m[j] > 0 && (ldc[j] < m[j] || !lfamd_mul_mat_takes_staged(Atype, m[j], k, n, flags)))
                    return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat_multi: a matrix of this call does not run the int8 batch body%s", "");
        for (int j0 = 0; j0 < count; j0 += 4) {
            const int c = count - j0 < 4 ? count - j0 : 4;
            HIPCHK(lfamd_launch_gemm_i8_staged(c, d_A + j0, m + j0, k, d_B, n, d_C + j0, ldc + j0, (hipStream_t)stream), "gemm_i8 (staged input, multi)");
        }
        return LFAMD_OK;
    }
    if (Btype == LFAMD_TYPE_STAGED_SCALED) { // sibling matrices on one scaled image: the route the same call takes on f32 rows
        if (n == 0)
            return LFAMD_OK;
        if (!d_B || ((uintptr_t)d_B & 15))
            return fail(LFAMD_ERR_INVALID, "mul_mat_multi: the staged image must be 16-byte aligned%s", "");
        if (count > 1 && count <= 4 && multi_i8_ok(Atype, count, m, ldc, k, n, flags))
            return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat_multi: these matrices run the int8 batch body together (LFAMD_TYPE_STAGED_Q8K)%s", "");
        bool fuse = count > 1 && count <= 4 && use_gemm(Atype, n, flags, k) && Atype != LFAMD_TYPE_Q4_0 && k > 0 && k % 256 == 0 &&
                    !(flags & (LFAMD_FLAG_GEMM_NARROW | LFAMD_FLAG_PRECISE | LFAMD_FLAG_FORCE_GENERIC)) && lfamd_gemm_wide_scaled_ok(Atype, plain);
        long rbs = 0;
        for (int j = 0; j < count && fuse; j++) {
            fuse = m[j] >= 0 && ldc[j] >= m[j];
            rbs += (m[j] + 127) / 128;
        }
        const scaled_image_ptrs im = scaled_image_of(d_B, k, n);
        if (fuse && (rbs * (long)(im.n_pad / 128) >= 192 || (flags & LFAMD_FLAG_GEMM_WIDE))) { // (one launch over the concatenated row blocks)
            HIPCHK(lfamd_launch_gemm_wide_multi(Atype, count, d_A, m, k, im.Xh, im.d8T, im.Xm, n, (long)im.n_pad, d_C, ldc, plain | 2, nullptr, 0,
                                                (hipStream_t)stream),
                   "gemm_wide_multi (staged input)");
            return LFAMD_OK;
        }
        for (int j = 0; j < count; j++) { // one call per matrix: each must take the image itself
            const int r = lfamd_mul_mat(Atype, d_A[j], m[j], k, Btype, d_B, 0, n, d_C[j], ldc[j], d_ws, ws_bytes, flags, stream);
            if (r != LFAMD_OK)
                return r;
        }
        return LFAMD_OK;
    }
    if (count == 1 && use_gemm_sb(Atype, n, flags, k, m[0])) // a handful of tokens on one matrix (attn_output, ffn_down): gemm_sb.hip
        return lfamd_mul_mat(Atype, d_A[0], m[0], k, Btype, d_B, b_row_bytes, n, d_C[0], ldc[0], d_ws, ws_bytes, flags, stream);
    // several tokens (6 and more) on sibling matrices that all take the small-batch MFMA kernel (ffn_gate + ffn_up): the activations
    // are staged once, then one launch per matrix — 14336 x 4096 x 2 at 8 tokens: 39.8 us on the multi-column GEMV, 32.7 us as two
    // separate calls, less with the shared staging
    const bool i8_body = Atype == LFAMD_TYPE_Q4_K && n <= 8;
    if (count > 1 && n >= (i8_body ? 4 : 6) && (Btype == LFAMD_TYPE_F32 || Btype == lfamd_vec_dot_type(Atype)) &&
        b_row_bytes >= lfamd_row_size(Btype, k)) {
        bool all_sb = true;
        for (int j = 0; j < count && all_sb; j++) // (small siblings — attn_k / attn_v — are faster on the fused GEMV below 8 tokens)
            all_sb = (m[j] > 8192 || (i8_body && n >= 8)) && ldc[j] >= m[j] && use_gemm_sb(Atype, n, flags, k, m[j]);
        if (all_sb) {
            if (ws_bytes < align_up(lfamd_gemm_sb_workspace(k), 256) || !d_ws)
                return fail(LFAMD_ERR_WORKSPACE, "mul_mat_multi: workspace too small%s", "");
            for (int j = 0; j < count; j++)
                HIPCHK(lfamd_launch_gemm_sb(Atype, d_A[j], m[j], k, Btype, d_B, b_row_bytes, n, d_C[j], ldc[j], d_ws, j > 0, (hipStream_t)stream),
                       "gemm_sb (multi)");
            return LFAMD_OK;
        }
    }
    // one fused launch when the GEMV path applies to every matrix; otherwise one mul_mat per matrix
    bool fuse = count <= 4 && use_gemv(Atype, n, flags, k) && (Btype == LFAMD_TYPE_F32 || Btype == lfamd_vec_dot_type(Atype)) &&
                k > 0 && k % lfamd_blck_size(Atype) == 0 && (Atype == LFAMD_TYPE_Q8_0 || k % 256 == 0) &&
                b_row_bytes >= lfamd_row_size(Btype, k);
    for (int j = 0; j < count && fuse; j++)
        fuse = m[j] >= 0 && ldc[j] >= m[j];
    if (fuse) {
        if (n == 0)
            return LFAMD_OK;
        HIPCHK(lfamd_launch_gemv_multi(Atype, count, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc,
                                       (flags & LFAMD_FLAG_Q0_VREGS32) ? 1 : 0, (flags & LFAMD_FLAG_PRECISE) ? 1 : 0,
                                       (hipStream_t)stream),
               "gemv_multi");
        return LFAMD_OK;
    }
    // Q4_K siblings whose tiles TOGETHER make a grid the int8 body takes (attn_q/k/v of an all-Q4_K layer: 48 row blocks at 512
    // tokens): one staging, one launch over the concatenated row blocks, exact integer dots (6144 x 4096 x 512: 47.5 us against 56.8)
    if (count > 1 && count <= 4 && (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_K) && k > 0 && b_row_bytes >= lfamd_row_size(Btype, k) &&
        multi_i8_ok(Atype, count, m, ldc, k, n, flags)) {
        if (ws_bytes < lfamd_gemm_i8_workspace(k, n) || !d_ws)
            return fail(LFAMD_ERR_WORKSPACE, "mul_mat_multi: workspace too small%s", "");
        HIPCHK(lfamd_launch_gemm_i8(count, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, d_ws, nullptr, (hipStream_t)stream), "gemm_i8 (multi)");
        return LFAMD_OK;
    }
    // K-quant batches: ONE activation prep for all the matrices, and one launch of the 128 x 128 body over their
    // concatenated row blocks when that grid fills the chip (attn_q/k/v: 48 + 8 + 8 row blocks instead of three
    // launches of which two fill a quarter of the CUs)
    bool gfuse = count > 1 && count <= 4 && use_gemm(Atype, n, flags, k) && Atype != LFAMD_TYPE_Q4_0 && k > 0 && k % 256 == 0 &&
                 (Btype == LFAMD_TYPE_F32 || Btype == lfamd_vec_dot_type(Atype)) && b_row_bytes >= lfamd_row_size(Btype, k) &&
                 !(flags & LFAMD_FLAG_GEMM_NARROW);
    long rbs = 0;
    for (int j = 0; j < count && gfuse; j++) {
        gfuse = m[j] >= 0 && ldc[j] >= m[j];
        rbs += (m[j] + 127) / 128;
    }
    if (gfuse) {
        hipStream_t s = (hipStream_t)stream;
        size_t n_pad = align_up((size_t)n, 128), nb = (size_t)(k / 256);
        gfuse = rbs * (long)(n_pad / 128) >= 192 || (flags & LFAMD_FLAG_GEMM_WIDE);
        if (gfuse) {
            size_t need = align_up(n_pad * (size_t)k * 2, 256) + align_up(nb * n_pad * 4, 256) + align_up(n_pad * nb * 32, 256);
            if (ws_bytes < need || !d_ws)
                return fail(LFAMD_ERR_WORKSPACE, "mul_mat_multi: workspace too small%s", "");
            uint8_t *ws = (uint8_t *)d_ws;
            void *Xh = ws;
            void *d8T = ws + align_up(n_pad * (size_t)k * 2, 256);
            void *Xm = (uint8_t *)d8T + align_up(nb * n_pad * 4, 256);
            const int scaled = (flags & LFAMD_FLAG_PRECISE) ? 0 : lfamd_gemm_wide_scaled_ok(Atype, plain);
                if (Btype == LFAMD_TYPE_F32)
                HIPCHK(lfamd_launch_prep_f32(d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, Xm, scaled ? 2 : 0, nullptr, s), "prep_f32");
            else
                HIPCHK(lfamd_launch_prep_q8k(d_B, b_row_bytes, n, (long)n_pad, k, Xh, d8T, Xm, scaled ? 2 : 0, nullptr, s), "prep_q8k");
            HIPCHK(lfamd_launch_gemm_wide_multi(Atype, count, d_A, m, k, Xh, d8T, Xm, n, (long)n_pad, d_C, ldc, plain | (scaled << 1), nullptr, 0, s),
                   "gemm_wide_multi");
            return LFAMD_OK;
        }
    }
    // Q8_0 batches on sibling matrices: one staging of the activations, one launch over the concatenated row blocks
    if (count > 1 && count <= 4 && use_gemm_q80_lf(Atype, n, flags, k) && (Btype == LFAMD_TYPE_F32 || Btype == LFAMD_TYPE_Q8_0) && k > 0 &&
        b_row_bytes >= lfamd_row_size(Btype, k)) {
        bool ok = true;
        for (int j = 0; j < count && ok; j++)
            ok = m[j] >= 0 && ldc[j] >= m[j];
        if (ok) {
            if (ws_bytes < lfamd_gemm_lf_workspace(k, n) || !d_ws)
                return fail(LFAMD_ERR_WORKSPACE, "mul_mat_multi: workspace too small%s", "");
            HIPCHK(lfamd_launch_gemm_lf_q80(count, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, d_ws, (hipStream_t)stream), "gemm_lf (Q8_0, multi)");
            return LFAMD_OK;
        }
    }
    for (int j = 0; j < count; j++) {
        int r = lfamd_mul_mat(Atype, d_A[j], m[j], k, Btype, d_B, b_row_bytes, n, d_C[j], ldc[j], d_ws, ws_bytes, flags, stream);
        if (r != LFAMD_OK)
            return r;
    }
    return LFAMD_OK;
}

size_t lfamd_mul_mat_id_workspace(int type, long rows, long cols, int experts, long tokens, int thinkers) {
    return lfamd_moe_workspace(type, rows, cols, experts, tokens, thinkers);
}

int lfamd_mul_mat_id(int type, const void *d_W, long rows, long cols, int experts, int Btype, const void *d_thought,
                     size_t b_row_bytes, int tasks, long tokens, const int32_t *d_plan, int thinkers, float *d_result,
                     void *d_ws, size_t ws_bytes, unsigned flags, void *stream) {
    (void)hipGetLastError(); // a stale error of an earlier call (e.g. an invalidated stream capture) must not fail this one
    if (!type_known(type))
        return fail(LFAMD_ERR_UNSUPPORTED, "mul_mat_id: unsupported weight type%s", "");
    if (rows < 0 || cols < 0 || cols % lfamd_blck_size(type) || experts <= 0 || tasks <= 0 || thinkers <= 0 ||
        tasks > thinkers || thinkers > experts)
        return fail(LFAMD_ERR_INVALID, "mul_mat_id: bad shape%s", "");
    // f32 activations (the GGML_OP_MUL_MAT_ID boundary) are served by the decode path, which quantises in-kernel
    const bool f32_decode = Btype == LFAMD_TYPE_F32 && !(flags & LFAMD_FLAG_FORCE_GENERIC) &&
                            ((tokens <= 4 && (type == LFAMD_TYPE_Q4_K || type == LFAMD_TYPE_Q5_K || type == LFAMD_TYPE_Q6_K)) ||
                             (tokens > 4 && cols % 256 == 0 && experts < 255 && tokens * thinkers <= 60 * 1024 &&
                              (type == LFAMD_TYPE_Q4_K || type == LFAMD_TYPE_Q5_K || type == LFAMD_TYPE_Q6_K)));
    if (Btype != lfamd_vec_dot_type(type) && !f32_decode)
        return fail(LFAMD_ERR_UNSUPPORTED,
                    "mul_mat_id: activations must be in the weight type's vec_dot format (F32 only for Q4_K / Q5_K / Q6_K experts)%s", "");
    if (tokens == 0 || rows == 0)
        return LFAMD_OK;
    size_t need = lfamd_moe_workspace(type, rows, cols, experts, tokens, thinkers);
    if (need && (ws_bytes < need || !d_ws))
        return fail(LFAMD_ERR_WORKSPACE, "mul_mat_id: workspace too small%s", "");
    HIPCHK(lfamd_launch_moe(type, d_W, rows, cols, experts, lfamd_packed_size(type, rows, cols), Btype, d_thought,
                            b_row_bytes, tasks, tokens, d_plan, thinkers, d_result, d_ws, ws_bytes, flags,
                            (hipStream_t)stream),
           "mul_mat_id");
    return LFAMD_OK;
}

int lfamd_vendor_gemm_available(void) {
    return lfamd_blaslt_ok() ? 1 : 0;
}

int lfamd_mul_mat_id_multi(int type, int count, const void *const *d_W, long rows, long cols, int experts, int Btype, const void *d_thought,
                           size_t b_row_bytes, int tasks, long tokens, const int32_t *d_plan, int thinkers, float *const *d_result,
                           void *d_ws, size_t ws_bytes, unsigned flags, void *stream) {
    (void)hipGetLastError();
    if (count <= 0)
        return LFAMD_OK;
    if (!d_W || !d_result || !d_plan || !d_thought)
        return fail(LFAMD_ERR_INVALID, "mul_mat_id_multi: null argument%s", "");
    for (int j = 0; j < count; j++) // (the fused launch below goes straight into the kernel: a null stack or result would be a device fault)
        if (!d_W[j] || !d_result[j])
            return fail(LFAMD_ERR_INVALID, "mul_mat_id_multi: null expert stack or result%s", "");
    if (tasks <= 0 || tasks > thinkers || !type_known(Btype))
        return fail(LFAMD_ERR_INVALID, "mul_mat_id_multi: bad shape or activation type%s", "");
    if (count <= 4 && type_known(type) && rows > 0 && cols > 0 && experts > 0 && thinkers > 0 && thinkers <= experts && tokens > 0 &&
        b_row_bytes >= lfamd_row_size(Btype, cols) && lfamd_moe_decode_multi_ok(type, cols, Btype, tasks, tokens, flags)) {
        HIPCHK(lfamd_launch_moe_decode_multi(type, count, d_W, rows, cols, experts, lfamd_packed_size(type, rows, cols), Btype, d_thought,
                                             b_row_bytes, tokens, d_plan, thinkers, d_result, (hipStream_t)stream),
               "mul_mat_id_multi");
        return LFAMD_OK;
    }
    for (int j = 0; j < count; j++) { // batches, other types, per-thinker activations: one operator at a time
        const int r = lfamd_mul_mat_id(type, d_W[j], rows, cols, experts, Btype, d_thought, b_row_bytes, tasks, tokens, d_plan, thinkers,
                                       d_result[j], d_ws, ws_bytes, flags, stream);
        if (r)
            return r;
    }
    return LFAMD_OK;
}

int lfamd_time_mul_mat(int Atype, const void *d_A, long m, long k, int Btype, const void *d_B, size_t b_row_bytes, long n,
                       float *d_C, long ldc, void *d_ws, size_t ws_bytes, unsigned flags, void *stream, int warmup,
                       int iters, float *avg_us) {
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < warmup; i++) {
        int r = lfamd_mul_mat(Atype, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, d_ws, ws_bytes, flags, stream);
        if (r)
            return r;
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0), "hipEventCreate");
    HIPCHK(hipEventCreate(&e1), "hipEventCreate");
    HIPCHK(hipEventRecord(e0, s), "hipEventRecord");
    for (int i = 0; i < iters; i++) {
        int r = lfamd_mul_mat(Atype, d_A, m, k, Btype, d_B, b_row_bytes, n, d_C, ldc, d_ws, ws_bytes, flags, stream);
        if (r)
            return r;
    }
    HIPCHK(hipEventRecord(e1, s), "hipEventRecord");
    HIPCHK(hipEventSynchronize(e1), "hipEventSynchronize");
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = iters > 0 ? ms * 1000.0f / iters : 0.0f;
    return LFAMD_OK;
}
}
