/*
 * oracle.h — CPU restatement of llamafile's quantized-matmul hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under llamafile_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PARITY STATUS: "parity unpinned" for every quantized type — the reference's
 * own tests hold no golden vector or known-answer test for Q4_0/Q8_0/K-quants
 * (SURVEY.md F6), and the reference cannot be built in this image (its path
 * includes the un-vendored llama.cpp/ggml-*.h and cosmo.h).
 * The F32 path is pinned the way the reference's own sgemm_*_test.cpp pins it:
 * against a double-accumulator GEMM with ULP statistics.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.
 */
#ifndef LFAMD_ORACLE_H_
#define LFAMD_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Which compiled variant of the reference is being restated (sgemm.cpp:26-102 picks one
 * per host CPU).  Only properties that change results are modelled. */
typedef struct {
    int vector_registers; /* 32 = AVX512 builds (*_avx512f, *_zen4), 16 = AVX/AVX2/AVXVNNI builds
                             (tinyblas_cpu.h:63-67) */
    int kn;               /* f32 lanes of the float kernels: 16 (AVX512) or 8 (AVX/AVX2)
                             (tinyblas_cpu_sgemm.inc:52-62) */
    int precise;          /* FLAG_precise (--precise) */
    int kahan_contract;   /* 1: madder's sub(mul(a,b),e) is contracted to fma(a,b,-e) by the
                             compiler (SURVEY.md §8c spec check: g++ 11 default contraction) */
} ora_variant;

ora_variant ora_variant_zen4(void); /* AVX512: 32 vregs, kn=16 */
ora_variant ora_variant_avx2(void); /* AVX2: 16 vregs, kn=8 */

/* ---- scalar conversions (ggml-impl.h GGML_FP16_TO_FP32 etc; IEEE, RNE) ---- */
float ora_fp16_to_fp32(uint16_t h);
uint16_t ora_fp32_to_fp16(float f);
float ora_bf16_to_fp32(uint16_t h);
uint16_t ora_fp32_to_bf16(float f);

/* ---- activation quantisers (upstream ggml-quants.c, un-vendored).
 * Never on a comparison path: tests feed identical quantised bytes to both sides. ---- */
void ora_quantize_row_q8_0(const float *x, void *y, long k);
void ora_quantize_row_q8_1(const float *x, void *y, long k);
void ora_quantize_row_q8_K(const float *x, void *y, long k); /* llamafile field order */

/* ---- whole-row dequantisation (ggml-cuda.cu.patch:3217-3471, 3684-3699; Appendix A) ---- */
int ora_dequantize_row(int type, const void *x, float *y, long k);

/* ---- the hot path ---- */

/* llamafile_sgemm (tinyblas_cpu_sgemm.inc:274-331): returns 1 if serviced, 0 for "not
 * supported" (same conditions as the reference's x86 builds).  k, lda, ldb in BLOCKS for
 * quantised types.  Computes only the part of C that thread `ith` of `nth` owns. */
int ora_llamafile_sgemm(long m, long n, long k, const void *A, long lda, const void *B, long ldb,
                        void *C, long ldc, int ith, int nth, int Atype, int Btype, int Ctype,
                        const ora_variant *v);

/* iqk_mul_mat (iqk_mul_mat.inc:182-202): ne00 in ELEMENTS, rows contiguous. */
int ora_iqk_mul_mat(long Nx, long Ny, long ne00, int typeA, const void *A, const void *B, float *C,
                    long stride_C, int ith, int nth);

/* iqk_mul_mat_moe (iqk_mul_mat.inc:204-221). row mapping = {int32 i1, int32 i2} */
int ora_iqk_mul_mat_moe(long Nx, long Ny, long ne00, int ne11, int typeA, const void *A,
                        const void *B, float *C, long nb1, long nb2, const void *vrow_mapping,
                        int ith, int nth);

/* tinyBLAS_Q0_AVX2 (tinyblas_cpu.h:780-1005), bit-exact restatement.  Atype Q8_0 or Q4_0,
 * B is Q8_0.  Pointer mode (NCB|NCC, tinyblas_cpu.h:94-98) when Bptr/Cptr non-NULL:
 * B row j = Bptr[j], C row j = Cptr[j]. */
int ora_q0_gemm(long m, long n, long k, int Atype, const void *A, long lda, const void *B, long ldb,
                const void *const *Bptr, float *C, long ldc, float *const *Cptr, int ith, int nth,
                const ora_variant *v);

/* mode[j*m+i] = 1 if output (i,j) is computed with Kahan compensation, 0 plain
 * (mnpack geometry, tinyblas_cpu.h:794-931). */
void ora_q0_precise_map(long m, long n, const ora_variant *v, uint8_t *mode);

/* tinyBLAS<> float GEMM (tinyblas_cpu.h:419-613) for F32/F16/BF16 A and B (TA,TB given as
 * ggml type ids), faithful lane/ruler-stack order for `v->kn` lanes. */
int ora_float_gemm(long m, long n, long k, int Atype, const void *A, long lda, int Btype,
                   const void *B, long ldb, float *C, long ldc, int ith, int nth,
                   const ora_variant *v);

/* ansiBLAS::sgemm golden (ansiblas.h:27-121): 8 double lanes + double tail, f32 in/out. */
void ora_ansiblas_sgemm(long m, long n, long k, const float *A, long lda, const float *B, long ldb,
                        float *C, long ldc);

/* f64 ground truth: dequantise A (any type) and B (F32, or Q8_0/Q8_1/Q8_K) and accumulate in
 * double.  Used to bound the error of oracle and candidate alike. */
int ora_f64_gemm(long m, long n, long kelems, int Atype, const void *A, size_t a_row_bytes,
                 int Btype, const void *B, size_t b_row_bytes, double *C, long ldc);

/* llamafile_mixmul (tinyblas_cpu_mixmul.inc:77-398) on plain arrays:
 * weights[experts][rows] rows of `cols` elems (row stride w_nb1 bytes, expert stride w_nb2),
 * thought f32 [tokens][tasks][cols], plan i32 [tokens][thinkers], result f32
 * [tokens][thinkers][rows].  Returns 0 for unsupported weight types (Q4_K etc). */
int ora_mixmul(int wtype, const void *weights, long cols, long rows, int experts, size_t w_nb1,
               size_t w_nb2, const float *thought, int tasks, long tokens, const int32_t *plan,
               int thinkers, float *result, const ora_variant *v);

/* ULP distance helpers (float.h flt::toint semantics, sgemm_matmul_test.cpp:78-86). */
long long ora_ulp_diff(float a, float b);

#ifdef __cplusplus
}
#endif
#endif
