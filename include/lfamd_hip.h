/*
 * lfamd_hip.h — C ABI of the MI355X (gfx950) quantized-matmul module, libllamafile_amd_hip.so.
 *
 * This is the thin device-side boundary the host plug-in (include/llamafile_sgemm.h) dlopen()s,
 * the way the reference's llamafile/cuda.c:701-753 dlopen()s ggml-rocm.so and imports a fixed
 * symbol list.  The reference's module exposes a whole ggml backend (12 symbols, cuda.c:726-737);
 * this tier only accelerates GGML_OP_MUL_MAT / MUL_MAT_ID, so the ABI is the small set below:
 * plain pointers and sizes, no C++/torch types.  Every function returns 0 on success or a
 * negative lfamd_status; lfamd_last_error() gives a message.  Nothing here falls back to the CPU.
 *
 * Reference counterparts (paths relative to /root/reference):
 *   lfamd_pack_weights   <- ggml_backend_cuda_buffer_set_tensor (weights upload; the backend owns
 *                           the device copy and may re-lay it out), ggml-cuda.cu.patch:16971-16977
 *   lfamd_mul_mat        <- ggml_cuda_mul_mat policy + ggml_cuda_op_mul_mat_vec_q / _mul_mat_q,
 *                           ggml-cuda.cu.patch:18377-18443, 14714-14806, 14188-14284;
 *                           numerics follow the CPU path llamafile_sgemm, tinyblas_cpu_sgemm.inc:274-331
 *   lfamd_mul_mat_id     <- ggml_cuda_mul_mat_id, ggml-cuda.cu.patch:18499-18635; numerics follow
 *                           iqk_mul_mat_moe (iqk_mul_mat.inc:204-221) / llamafile_mixmul
 *   lfamd_quantize_rows  <- quantize_q8_1 (ggml-cuda.cu.patch:15259-15293); formats follow the CPU
 *                           vec_dot types Q8_0 / Q8_1 / Q8_K
 */
#ifndef LFAMD_HIP_H_
#define LFAMD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFAMD_ABI_VERSION 1

enum lfamd_status {
    LFAMD_OK = 0,
    LFAMD_ERR_UNSUPPORTED = -1, /* type / shape this module has no kernel for (caller decides) */
    LFAMD_ERR_INVALID = -2,     /* precondition violated (the reference asserts) */
    LFAMD_ERR_HIP = -3,         /* HIP runtime error; see lfamd_last_error() */
    LFAMD_ERR_WORKSPACE = -4,   /* workspace too small */
};

/* numerics flags for lfamd_mul_mat */
#define LFAMD_FLAG_Q0_VREGS32 1u /* restate the reference's 32-vector-register (AVX512) build of
                                    tinyBLAS_Q0: Kahan on 2x1/1x2/1x1 edge tiles (tinyblas_cpu.h:797-830) */
#define LFAMD_FLAG_PRECISE 2u    /* FLAG_precise (--precise): Kahan everywhere in the Q0 kernels */
#define LFAMD_FLAG_FORCE_GENERIC 4u /* debugging: the generic one-wave-per-row kernel — only for tensors kept as GGUF rows
                                       (floats, legacy 32-block rows that are not whole 256-weight groups); packed types
                                       answer LFAMD_ERR_UNSUPPORTED */
#define LFAMD_FLAG_GEMM_NARROW 8u   /* testing: force the 128x64 split-K MFMA body (default: chosen by grid size) */
#define LFAMD_FLAG_GEMM_WIDE 16u    /* testing: force the 128x128 MFMA body */
#define LFAMD_FLAG_GEMM_PLAIN 32u   /* testing: the 128x128 body without loader waves (Q4_K / Q5_K default to them) */
#define LFAMD_FLAG_Q80_EXACT 64u    /* Q8_0 batches (n > 8): the BIT-EXACT restatement of tinyBLAS_Q0's 8-lane chains (VALU, ~12x
                                       slower) instead of the default: this module's f16 MFMA body on the resident image, f16(d * q) x
                                       f16(d8 * code), <= 1e-3 (csrc/gemm_lf.hip; rows that are not whole 128-weight groups run the exact
                                       kernel anyway).  n <= 8 — the Q8_0 vecdot of the north star — is always bit-exact;
                                       LFAMD_FLAG_PRECISE implies this flag. */

int lfamd_abi_version(void);
const char *lfamd_last_error(void);

/* Device management. */
int lfamd_device_count(void);
int lfamd_init(int device); /* hipSetDevice + arch check (gfx950 required) */
int lfamd_device_name(int device, char *buf, size_t len);

/* Device memory (thin wrappers so a C host needs no HIP headers). */
int lfamd_malloc(void **dptr, size_t bytes);
int lfamd_free(void *dptr);
int lfamd_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int lfamd_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int lfamd_memset(void *dst, int value, size_t bytes, void *stream);
/* Pinned host memory mapped into the device's address space (the same pointer is valid on both sides): a decode-sized
 * lfamd_mul_mat may take its activation row from it and write its result to it directly (no float atomics on it: n = 1 or
 * n = 2 only, see lfamd_mul_mat).  What llamafile_sgemm's host-pointer path uses for single-column calls. */
int lfamd_host_alloc(void **p, size_t bytes);
int lfamd_host_free(void *p);
int lfamd_stream_sync(void *stream);

/* ---- weights ----------------------------------------------------------------------------
 * Weights are kept on the device in a PACKED layout chosen per type so that every kernel reads
 * them with coalesced 16-byte-per-lane loads (DESIGN.md "Data layout in HBM").  `raw` is the
 * tensor exactly as GGUF/ggml stores it: `rows` rows, `raw_row_bytes` apart, each a sequence of
 * blocks (include/lfamd_blocks.h).  Packing is a device kernel: raw and packed are device
 * pointers.  lfamd_packed_size is the ONLY source of the packed byte count: Q4_K / Q5_K / Q6_K / Q4_0 images are
 * the GGUF size (+ tile round-up); Q2_K / Q3_K / IQ4_XS are compact images of 84 / 116 / 144 bytes per 256 weights (1.00x /
 * 1.055x / 1.06x the file: DESIGN.md section 10.10; batches expand them per call into the canonical image in the workspace); Q8_0 is
 * ONE image of the file's size (1.0625 bytes per weight) that the bit-exact vecdot, the exact batch kernel and the f16 MFMA batch
 * body all read — only a process that opted into the vendor GEMM (LFAMD_USE_BLASLT=1) keeps f16(d * q) rows behind it (3.1 bytes
 * per weight in all); Q4_1 / Q5_0 / Q5_1 are kept as the canonical image their MFMA and decode kernels read (192 bytes per 256
 * weights: 1.2 - 1.5x the file); legacy 32-block rows that are not whole
 * 256-weight groups and float tensors stay as GGUF rows. */
size_t lfamd_packed_size(int type, long rows, long cols);
/* 1 when the process opted into the vendor's GEMM (LFAMD_USE_BLASLT=1 in the environment before the first call, and hipBLASLt
 * loads): batches on PLAIN 16-bit float matrices — F16 / BF16 weight tensors and a second, f16(d * q) image of Q8_0 weights — then
 * go through it (a shape or device it declines falls back to this module's kernels).  0 (the default): every batch runs on this
 * module's own MFMA bodies and nothing depends on the library.  Constant for the life of the process. */
int lfamd_vendor_gemm_available(void);
int lfamd_pack_weights(int type, long rows, long cols, const void *d_raw, size_t raw_row_bytes,
                       void *d_packed, void *stream);

/* Batches (n > 8) of Q4_K / Q5_K / Q6_K run the scaled-operand MFMA body by default: weights as f16(d * sc * q), activations as
 * f16(d8 * code), f32 accumulate — one f16 rounding per operand, relative error ~1e-4 (north star: 1e-3), no scaling per
 * super-block.  Its constants need |d| * 63 < 64 and |dmin| * 63 <= 65504 for every block, which every ggml-quantised
 * model satisfies (d = max|w| / (15 * 63)).  lfamd_scaled_gemm_ok checks a packed matrix once after the upload
 * (synchronises the stream): 1 = in range, 0 = out of range -> pass LFAMD_FLAG_PRECISE with this matrix (exact integer
 * codes, f32 scales; out-of-range scales would otherwise surface as inf / NaN outputs), < 0 = error.  Types other than
 * the K-quants with a resident layout: always 1.  (Q6_K: |d| * 127 * 32 <= 65504.)
 * Activations are normalised per token by a power of two before the f16 staging and the output column is scaled back
 * (exact), so their magnitude is not limited by f16; only a spread of more than ~2^24 INSIDE one token underflows its
 * smallest super-blocks (they contribute < 1e-7 of the result). */
int lfamd_scaled_gemm_ok(int type, long rows, long cols, const void *d_packed, void *stream);

/* Which arithmetic a lfamd_mul_mat call with these arguments runs: 1 = exact integer block dot products with f32 scales (the
 * reference's CPU arithmetic, iqk_mul_mat.inc:601-643; within 2e-6 of it: only the order of the f32 sums differs) — every call
 * of up to 32 columns, LFAMD_FLAG_PRECISE, and since round 4 the Q4_K batches that run on the int8 matrix cores
 * (llamafile_amd/csrc/gemm_i8.hip: grids of at most 256 tiles of 128 x 128 that still fill half the chip — attn_q, attn_output,
 * ffn_down of an 8B model at 512 tokens); 0 = scaled operands on the f16 matrix cores (one f16 rounding per operand, <= 1e-3,
 * measured ~3e-4; Q6_K's exact body also rounds sc * (q - 32) above 2048).  No device call is made. */
int lfamd_mul_mat_is_exact(int Atype, long m, long k, long n, unsigned flags);

/* ---- activations --------------------------------------------------------------------------
 * f32 rows -> the reference's activation block format (vec_dot_type: Q8_0, Q8_1 or Q8_K in
 * llamafile's field order).  Same rounding as the scalar reference quantisers. */
int lfamd_quantize_rows(int vec_dot_type, const float *d_x, long nrows, long cols, size_t x_row_bytes,
                        void *d_y, size_t y_row_bytes, void *stream);

/* ---- GGML_OP_MUL_MAT -----------------------------------------------------------------------
 * C[j*ldc + i] = sum_l A[i][l] * B[j][l]   (C = A^T B, column-major C like llamafile_sgemm)
 *   A: packed weights (lfamd_pack_weights), m rows x k elements, type Atype
 *   B: n rows, either in the reference's activation format Btype (= vec_dot_type of Atype; the
 *      llamafile_sgemm boundary) or F32 (the GGML_OP_MUL_MAT boundary: quantised on the device,
 *      bit-identically to quantize_row_q8_K / q8_0 / q8_1, fused into the kernels); row stride
 *      b_row_bytes
 *   C: f32, ldc >= m; ORDINARY device memory (hipMalloc / lfamd_malloc): the small-batch kernels add their two K halves
 *      with hardware float atomics, which fine-grained or host-mapped memory does not honour
 * Policy (cf. ggml_cuda_mul_mat): one token -> wave-reduction GEMV kernels; 2 .. 8 tokens -> multi-column GEMVs or, where
 * faster, the small-batch MFMA kernel (also 9 .. 32 tokens on deep rows and matrices of at most 8192 rows: csrc/gemm_sb.hip);
 * larger batches -> dequant-to-MFMA GEMM on 64- / 128-token tiles (plain F16 / BF16 weights and Q8_0: the vendor GEMM instead when
 * lfamd_vendor_gemm_available()).  `workspace` must hold lfamd_mul_mat_workspace() bytes (may be NULL
 * if that is 0). */
size_t lfamd_mul_mat_workspace(int Atype, long m, long k, long n);
int lfamd_mul_mat(int Atype, const void *d_A_packed, long m, long k, int Btype, const void *d_B,
                  size_t b_row_bytes, long n, float *d_C, long ldc, void *d_workspace,
                  size_t workspace_bytes, unsigned flags, void *stream);

/* Several GGML_OP_MUL_MAT nodes that read the SAME activations (attn_q/k/v; ffn_gate/up) with weights
 * of one type and row length: what a backend's graph_compute may fuse (ggml_backend_cuda_graph_compute,
 * ggml-cuda.cu.patch:18945, walks the node list and is free to).  For n <= 8 and Q4_K / Q6_K weights
 * this is ONE kernel launch over the concatenated rows; otherwise it runs lfamd_mul_mat per matrix
 * (workspace: the largest lfamd_mul_mat_workspace of the set).  Results are identical either way. */
int lfamd_mul_mat_multi(int Atype, int count, const void *const *d_A_packed, const long *m, long k, int Btype,
                        const void *d_B, size_t b_row_bytes, long n, float *const *d_C, const long *ldc,
                        void *d_workspace, size_t workspace_bytes, unsigned flags, void *stream);

/* The same for sibling nodes whose weight TYPES may differ (attn_q/k = Q4_K or Q5_K with attn_v = Q6_K in the *_K_M
 * files): at decode (n = 1) those two groups run as ONE launch (bit-identical to the separate calls); batches of
 * Q4_K / Q5_K / Q6_K nodes share one staged copy of the activations (scaled-operand body, see above) and, when the two types'
 * tiles fill one round of the chip, one launch; any other mix falls
 * back to one lfamd_mul_mat_multi per run of equal types. */
int lfamd_mul_mat_multi_types(int count, const int *Atype, const void *const *d_A_packed, const long *m, long k, int Btype,
                              const void *d_B, size_t b_row_bytes, long n, float *const *d_C, const long *ldc, void *d_ws,
                              size_t ws_bytes, unsigned flags, void *stream);

/* ---- GGML_OP_MUL_MAT_ID (mixture of experts) -------------------------------------------------
 * For every (token, thinker): result[token][thinker][:] = W[plan[token][thinker]] x
 * thought[token][thinker % tasks][:]   (tinyblas_cpu_mixmul.inc:39-50).
 *   d_W_packed: `experts` packed matrices, each lfamd_packed_size(type, rows, cols) bytes apart
 *   d_thought : activation rows in vec_dot format, index (token*tasks + task), stride b_row_bytes
 *   d_plan    : int32 [tokens][thinkers]
 *   d_result  : f32 [tokens][thinkers][rows]
 * Decode (tokens <= 4, Q4_K / Q6_K experts): the GEMV kernels read the expert index from d_plan themselves; thinkers
 * sharing their activations (tasks == 1) are one launch.  Batches of Q4_K / Q5_K / Q6_K experts (up to 60 Ki rows): a
 * one-work-group routing kernel groups the rows by expert on the device and ONE launch of the 128x128 MFMA body covers
 * (expert, row block, token tile) — Q4_K / Q5_K on scaled operands like lfamd_mul_mat's batches (exact integer codes with
 * LFAMD_FLAG_PRECISE; lfamd_scaled_gemm_ok over the whole stack: rows = experts * roundup(rows, 32)).  Both are
 * asynchronous and graph-capturable — no host read-back (the reference
 * synchronises, ggml-cuda.cu.patch:18528-18531) — and accept Btype F32 (quantised on the device).  Rows whose expert id
 * is out of range are left untouched.  Other expert types gather rows per expert after a routing read-back and run one
 * mat-mul per expert. */
size_t lfamd_mul_mat_id_workspace(int type, long rows, long cols, int experts, long tokens, int thinkers);
int lfamd_mul_mat_id(int type, const void *d_W_packed, long rows, long cols, int experts, int Btype,
                     const void *d_thought, size_t b_row_bytes, int tasks, long tokens,
                     const int32_t *d_plan, int thinkers, float *d_result, void *d_workspace,
                     size_t workspace_bytes, unsigned flags, void *stream);

/* Several GGML_OP_MUL_MAT_ID nodes of one weight type and shape over the SAME activations and routing table (ffn_gate_exps and
 * ffn_up_exps of a layer, ggml-cuda.cu.patch:18945 graph_compute sees them back to back): at decode (tokens <= 4, tasks == 1, Q4_K / Q5_K
 * / Q6_K experts) all (tensor, thinker) GEMVs of a token share launches of up to four — one launch for Mixtral's two tensors x two
 * thinkers; otherwise one lfamd_mul_mat_id per tensor.  d_result[j]: f32 [tokens][thinkers][rows] of tensor j. */
int lfamd_mul_mat_id_multi(int type, int count, const void *const *d_W_packed, long rows, long cols, int experts, int Btype,
                           const void *d_thought, size_t b_row_bytes, int tasks, long tokens, const int32_t *d_plan, int thinkers,
                           float *const *d_result, void *d_ws, size_t ws_bytes, unsigned flags, void *stream);

/* The staged activation image of the int8 batch body as an OUTPUT format of the two fused producers below (vec_dot_type =
 * LFAMD_TYPE_STAGED_Q8K, d_yq = a 16-byte aligned buffer of lfamd_staged_q8k_size(k, nrows) bytes, yq_row_bytes ignored) and as an
 * INPUT format of lfamd_mul_mat / lfamd_mul_mat_multi (Btype = LFAMD_TYPE_STAGED_Q8K, d_B = that buffer, b_row_bytes ignored):
 * the Q8_K codes, scales and block sums of quantize_row_q8_K, laid out the way the Q4_K batch body on the int8 matrix cores reads
 * them — a graph that owns the producer runs NO staging launch in front of the mat-mul (reference: quantize_q8_1 fused in front
 * of MMQ, ggml-cuda.cu.patch:15259-15292, 17896).  The result has the bits of the same call on the producer's f32 output.
 * lfamd_mul_mat_takes_staged() says whether a call accepts the image (Q4_K batches that run the int8 body: grids of at most 256
 * tiles of 128 x 128 whose 128 x 64 tiles fill half the chip — attn_q / attn_output / ffn_down of the Llama shapes at 512 tokens;
 * sibling matrices of one lfamd_mul_mat_multi call count together); other calls answer LFAMD_ERR_UNSUPPORTED for it and want Q8_K
 * blocks or f32 rows. */
#define LFAMD_TYPE_STAGED_Q8K 0x1000
size_t lfamd_staged_q8k_size(long k, long nrows);
int lfamd_mul_mat_takes_staged(int Atype, long m, long k, long n, unsigned flags);
/* The same for the scaled-operand f16 batch bodies (Q4_K / Q5_K / Q6_K batches that do not run the int8 body: attn_q/k/v and
 * ffn_gate/up behind a norm, Q6_K ffn_down behind the SwiGLU): LFAMD_TYPE_STAGED_SCALED, a buffer of lfamd_staged_scaled_size(k,
 * nrows) bytes — f16(q8 * d8 * 2^-e(token)) operands with a per-token power-of-two normalisation, the mins operand, 2^e per token;
 * one image serves sibling matrices of different K-quant types (lfamd_mul_mat_multi / _multi_types).  The bits of the same call on
 * the producer's f32 output.  lfamd_mul_mat_takes_staged_scaled() says whether a call accepts it. */
#define LFAMD_TYPE_STAGED_SCALED 0x1001
size_t lfamd_staged_scaled_size(long k, long nrows);
int lfamd_mul_mat_takes_staged_scaled(int Atype, long m, long k, long n, unsigned flags);

/* ---- the step in front of the path, fused: RMS-norm x weight -> Q8_K -----------------------------
 * y[i] = (x[i] * 1/sqrtf(mean(x^2) + eps)) * weight[i] per row (ggml_compute_forward_rms_norm_f32 + the MUL node; GPU
 * reference rms_norm_f32, ggml-cuda.cu.patch:14926-14960), written as the reference's Q8_K activation blocks
 * (quantize_row_q8_K, llamafile field order) to d_yq and / or as f32 to d_yf (either may be NULL; d_weight may be NULL = 1).
 * The mat-muls behind the norm then take Btype = Q8_K: no quantisation left in their prologue.  k % 256 == 0;
 * vec_dot_type must be Q8_K (the K-quant and IQ4_XS weights' activation format). */
int lfamd_rms_norm_quantize(const float *d_x, size_t x_row_bytes, const float *d_weight, float eps, long nrows, long k,
                            int vec_dot_type, void *d_yq, size_t yq_row_bytes, float *d_yf, size_t yf_row_bytes, void *stream);

/* The step in front of ffn_down, fused the same way: y = silu(gate) * up, silu(x) = x / (1 + expf(-x)) (silu_f32,
 * ggml-cuda.cu.patch:16172-16179, + the MUL node), written as Q8_K blocks to d_yq and / or f32 to d_yf.  k % 256 == 0. */
int lfamd_swiglu_quantize(const float *d_gate, size_t gate_row_bytes, const float *d_up, size_t up_row_bytes, long nrows, long k,
                          int vec_dot_type, void *d_yq, size_t yq_row_bytes, float *d_yf, size_t yf_row_bytes, void *stream);

/* ---- F16 batched GEMM (attention KQ / KQV) ------------------------------------------------------
 * The interface of tinyblasGemmStridedBatchedEx / tinyblasGemmBatchedEx (llamafile/tinyblas.h:59-71, tinyblas.cu:652-857)
 * for the operand arrangement ggml calls them with (ggml_cuda_mul_mat_batched_cublas, ggml-cuda.cu.patch:18231-18376):
 * transa = T, transb = N, f16 operands, result type Ctype = F16 or F32, f32 accumulation on MFMA:
 *     C_b[j * ldc + i] = alpha * sum_l A_b[i * lda + l] * B_b[j * ldb + l] + beta * C_b[j * ldc + i],   b < batch
 * Leading dimensions and strides in ELEMENTS (lda, ldb >= k; ldc >= m); beta is only applied when nonzero.  The
 * pointer-array form takes DEVICE arrays of `batch` device pointers. */
int lfamd_gemm_strided_batched_f16(long m, long n, long k, float alpha, const void *d_A, long lda, long long strideA,
                                   const void *d_B, long ldb, long long strideB, float beta, void *d_C, int Ctype, long ldc,
                                   long long strideC, int batch, void *stream);
int lfamd_gemm_batched_f16(long m, long n, long k, float alpha, const void *const *d_Aarray, long lda, const void *const *d_Barray,
                           long ldb, float beta, void *const *d_Carray, int Ctype, long ldc, int batch, void *stream);

/* ---- collectives (tensor parallel, one process per GPU) ---------------------------------------
 * The exchange step of the sharded path: attn_output / ffn_down are split by input columns and
 * the f32 partial sums of the residual stream are all-reduced; output.weight is split by vocabulary rows and the logits
 * all-gathered.  The reference has no collective — its row split gathers every result on a main GPU with peer copies
 * (ggml_cuda_op_mul_mat, ggml-cuda.cu.patch:17853-18153; :17781-17851, 18077-18121); these calls replace that step.
 * All of them are asynchronous on `stream` and hipGraph-capturable.
 *
 * Bootstrap: rank 0 calls lfamd_comm_unique_id and hands the 128 bytes to every rank through whatever transport the
 * host has (torch.distributed, MPI, a pipe); every rank calls lfamd_comm_init.  id128 may be NULL for world 1 or for a
 * communicator that only ever uses the one-shot path.  RCCL (librccl.so.1) is dlopen()ed at that point, not before.
 *
 * One-shot all-reduce for decode-sized messages (n_embd * 4 bytes = 16-32 KB): every rank allocates an exchange block of
 * lfamd_oneshot_bytes(max_message_bytes) bytes with lfamd_oneshot_alloc — FINE-GRAINED (uncached) device memory: ordinary
 * hipMalloc memory is coherent between GPUs only at kernel boundaries, and attach refuses it —, exports it (64-byte IPC
 * handle), the host gathers
 * the world's handles in rank order and every rank attaches them (the host must barrier between attach and the first
 * all-reduce).  lfamd_comm_allreduce_add_f32 then runs ONE kernel per call for messages that fit: publish (write-through),
 * flag every peer, wait (bounded), sum the world's partials in rank order — bit-identical on every rank — and add the
 * residual in the same pass.  Larger messages (the 8-16 MB prefill tensors) go through ncclAllReduce.
 * A peer is waited for up to LFAMD_ONESHOT_TIMEOUT_S (default 4) seconds of wall time; then the error is latched (later
 * calls do not wait again, their results are void) until lfamd_comm_clear_error.  lfamd_comm_check() != 0 reports it
 * (1 + the rank that never arrived): a host must call it before trusting results — bench.py does after every timed region. */
typedef struct lfamd_comm lfamd_comm;
int lfamd_comm_unique_id(void *id128);
int lfamd_comm_init(lfamd_comm **comm, int rank, int world, const void *id128);
int lfamd_comm_destroy(lfamd_comm *comm);
/* The single-process form (the shape of ncclCommInitAll): one host thread drives `ndev` devices,
 * comms[i] is rank i on HIP device devices[i].  The exchange blocks are allocated here (fine-grained) and shared as plain
 * pointers — nothing to export or attach —, every collective call below makes the rank's device current for its launch
 * (and restores the caller's), and `stream` must be a stream of that device (or NULL).  Issue every rank's call before
 * synchronising any of them.  RCCL communicators are added when the devices are distinct and the library has
 * ncclCommInitAll; otherwise only messages up to max_message_bytes are served.  1 <= ndev <= 8; a device may repeat (how one
 * GPU rehearses two).  Errors: LFAMD_ERR_INVALID (arguments), LFAMD_ERR_HIP (no device / no peer access / out of memory). */
int lfamd_comm_init_all(lfamd_comm **comms, int ndev, const int *devices, size_t max_message_bytes);
size_t lfamd_oneshot_bytes(size_t max_message_bytes);
int lfamd_oneshot_alloc(void **d_block, size_t bytes);
int lfamd_oneshot_free(void *d_block);
int lfamd_oneshot_export(void *d_block, void *handle64);
int lfamd_oneshot_attach(lfamd_comm *comm, void *d_local_block, size_t block_bytes, const void *handles_world_x_64,
                         size_t max_message_bytes);
/* d_out = (d_residual ? d_residual : 0) + sum over ranks of d_partial   (d_out may alias d_partial) */
int lfamd_comm_allreduce_add_f32(lfamd_comm *comm, const float *d_partial, const float *d_residual, float *d_out, long count,
                                 void *stream);
int lfamd_comm_allreduce_sum_f32(lfamd_comm *comm, float *d_inout, long count, void *stream);
int lfamd_comm_allgather(lfamd_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);

int lfamd_comm_check(lfamd_comm *comm);
int lfamd_comm_clear_error(lfamd_comm *comm);

/* ---- instrumentation ------------------------------------------------------------------------
 * Average device time (microseconds, HIP events on `stream`) of `iters` back-to-back launches of
 * the same lfamd_mul_mat call, after `warmup` untimed ones. */
int lfamd_time_mul_mat(int Atype, const void *d_A_packed, long m, long k, int Btype, const void *d_B,
                       size_t b_row_bytes, long n, float *d_C, long ldc, void *d_workspace,
                       size_t workspace_bytes, unsigned flags, void *stream, int warmup, int iters,
                       float *avg_us);

#ifdef __cplusplus
}
#endif
#endif /* LFAMD_HIP_H_ */
