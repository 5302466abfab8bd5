/*
 * lfamd_blocks.h — GGUF/ggml block formats consumed by the quantized-matmul hot path.
 *
 * These are the on-disk / in-memory layouts the reference multiplies
 * (SURVEY.md §8 a-0).  The structs live in the reference's un-vendored
 * ggml-common.h (llama.cpp @ 8b3befc); the in-tree evidence for each layout is
 * cited per struct (paths relative to /root/reference):
 *
 *   Q4_0  llamafile/tinyblas_cpu.h:977-983, llamafile/iqk_mul_mat.inc:1219-1239
 *   Q4_1/Q5_0/Q5_1  llamafile/iqk_mul_mat.inc:1241-1283
 *   Q8_0  llamafile/tinyblas_cpu.h:973-975
 *   Q8_1  llama.cpp.patches/patches/ggml-cuda.cu.patch:15282-15292
 *   Q2_K..Q6_K  llama.cpp.patches/patches/ggml-cuda.cu.patch:3217-3471,
 *               llamafile/iqk_mul_mat.inc:417-599
 *   IQ4_XS llamafile/iqk_mul_mat.inc:432-470
 *   Q8_K  llama.cpp.patches/patches/ggml-common.h.patch:25-35
 *         (llamafile order: d, bsums[16], qs[256] — NOT upstream's d, qs, bsums)
 *
 * Plain C, no dependencies: included by the HIP kernels, the host library and
 * (as data definitions only) by the test oracle.
 */
#ifndef LFAMD_BLOCKS_H_
#define LFAMD_BLOCKS_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum ggml_type numeric ids (also the GGUF on-disk ids). */
enum lfamd_ggml_type {
    LFAMD_TYPE_F32 = 0,
    LFAMD_TYPE_F16 = 1,
    LFAMD_TYPE_Q4_0 = 2,
    LFAMD_TYPE_Q4_1 = 3,
    LFAMD_TYPE_Q5_0 = 6,
    LFAMD_TYPE_Q5_1 = 7,
    LFAMD_TYPE_Q8_0 = 8,
    LFAMD_TYPE_Q8_1 = 9,
    LFAMD_TYPE_Q2_K = 10,
    LFAMD_TYPE_Q3_K = 11,
    LFAMD_TYPE_Q4_K = 12,
    LFAMD_TYPE_Q5_K = 13,
    LFAMD_TYPE_Q6_K = 14,
    LFAMD_TYPE_Q8_K = 15,
    LFAMD_TYPE_IQ4_XS = 23,
    LFAMD_TYPE_I32 = 26,
    LFAMD_TYPE_BF16 = 30,
};

#define LFAMD_QK 32   /* QK4_0 == QK4_1 == QK5_0 == QK5_1 == QK8_0 == QK8_1 */
#define LFAMD_QK_K 256

typedef uint16_t lfamd_half; /* IEEE binary16 bit pattern */
typedef uint16_t lfamd_bf16; /* top 16 bits of binary32 */

#pragma pack(push, 1)

typedef struct {
    lfamd_half d;
    uint8_t qs[16];
} lfamd_block_q4_0; /* 18 B / 32 w */

typedef struct {
    lfamd_half d;
    lfamd_half m;
    uint8_t qs[16];
} lfamd_block_q4_1; /* 20 B */

typedef struct {
    lfamd_half d;
    uint8_t qh[4];
    uint8_t qs[16];
} lfamd_block_q5_0; /* 22 B */

typedef struct {
    lfamd_half d;
    lfamd_half m;
    uint8_t qh[4];
    uint8_t qs[16];
} lfamd_block_q5_1; /* 24 B */

typedef struct {
    lfamd_half d;
    int8_t qs[32];
} lfamd_block_q8_0; /* 34 B */

typedef struct {
    lfamd_half d;
    lfamd_half s; /* d * sum(qs) */
    int8_t qs[32];
} lfamd_block_q8_1; /* 36 B */

typedef struct {
    uint8_t scales[16]; /* low nibble scale, high nibble min */
    uint8_t qs[64];
    lfamd_half d;
    lfamd_half dmin;
} lfamd_block_q2_K; /* 84 B / 256 w */

typedef struct {
    uint8_t hmask[32];
    uint8_t qs[64];
    uint8_t scales[12];
    lfamd_half d;
} lfamd_block_q3_K; /* 110 B */

typedef struct {
    lfamd_half d;
    lfamd_half dmin;
    uint8_t scales[12];
    uint8_t qs[128];
} lfamd_block_q4_K; /* 144 B */

typedef struct {
    lfamd_half d;
    lfamd_half dmin;
    uint8_t scales[12];
    uint8_t qh[32];
    uint8_t qs[128];
} lfamd_block_q5_K; /* 176 B */

typedef struct {
    uint8_t ql[128];
    uint8_t qh[64];
    int8_t scales[16];
    lfamd_half d;
} lfamd_block_q6_K; /* 210 B */

typedef struct {
    lfamd_half d;
    uint16_t scales_h;
    uint8_t scales_l[4];
    uint8_t qs[128];
} lfamd_block_iq4_xs; /* 136 B */

/* llamafile's re-ordered Q8_K (ggml-common.h.patch:25-35) */
typedef struct {
    float d;
    int16_t bsums[16];
    int8_t qs[256];
} lfamd_block_q8_K; /* 292 B */

#pragma pack(pop)

/* elements per block / bytes per block; 0 for unknown types */
static inline int lfamd_blck_size(int type) {
    switch (type) {
    case LFAMD_TYPE_F32:
    case LFAMD_TYPE_F16:
    case LFAMD_TYPE_BF16:
    case LFAMD_TYPE_I32:
        return 1;
    case LFAMD_TYPE_Q4_0:
    case LFAMD_TYPE_Q4_1:
    case LFAMD_TYPE_Q5_0:
    case LFAMD_TYPE_Q5_1:
    case LFAMD_TYPE_Q8_0:
    case LFAMD_TYPE_Q8_1:
        return 32;
    case LFAMD_TYPE_Q2_K:
    case LFAMD_TYPE_Q3_K:
    case LFAMD_TYPE_Q4_K:
    case LFAMD_TYPE_Q5_K:
    case LFAMD_TYPE_Q6_K:
    case LFAMD_TYPE_Q8_K:
    case LFAMD_TYPE_IQ4_XS:
        return 256;
    default:
        return 0;
    }
}

static inline size_t lfamd_type_size(int type) {
    switch (type) {
    case LFAMD_TYPE_F32:
    case LFAMD_TYPE_I32:
        return 4;
    case LFAMD_TYPE_F16:
    case LFAMD_TYPE_BF16:
        return 2;
    case LFAMD_TYPE_Q4_0:
        return sizeof(lfamd_block_q4_0);
    case LFAMD_TYPE_Q4_1:
        return sizeof(lfamd_block_q4_1);
    case LFAMD_TYPE_Q5_0:
        return sizeof(lfamd_block_q5_0);
    case LFAMD_TYPE_Q5_1:
        return sizeof(lfamd_block_q5_1);
    case LFAMD_TYPE_Q8_0:
        return sizeof(lfamd_block_q8_0);
    case LFAMD_TYPE_Q8_1:
        return sizeof(lfamd_block_q8_1);
    case LFAMD_TYPE_Q2_K:
        return sizeof(lfamd_block_q2_K);
    case LFAMD_TYPE_Q3_K:
        return sizeof(lfamd_block_q3_K);
    case LFAMD_TYPE_Q4_K:
        return sizeof(lfamd_block_q4_K);
    case LFAMD_TYPE_Q5_K:
        return sizeof(lfamd_block_q5_K);
    case LFAMD_TYPE_Q6_K:
        return sizeof(lfamd_block_q6_K);
    case LFAMD_TYPE_Q8_K:
        return sizeof(lfamd_block_q8_K);
    case LFAMD_TYPE_IQ4_XS:
        return sizeof(lfamd_block_iq4_xs);
    default:
        return 0;
    }
}

/* ggml_row_size(type, ne): bytes of one row of ne elements */
static inline size_t lfamd_row_size(int type, long ne) {
    int bs = lfamd_blck_size(type);
    return bs ? lfamd_type_size(type) * (size_t)(ne / bs) : 0;
}

/* type_traits[].vec_dot_type: the activation format a weight type is multiplied with
 * (SURVEY.md §8 a-0 footnote; consistent with iqk_mul_mat.inc:1410-1456). */
static inline int lfamd_vec_dot_type(int type) {
    switch (type) {
    case LFAMD_TYPE_Q4_0:
    case LFAMD_TYPE_Q5_0:
    case LFAMD_TYPE_Q8_0:
        return LFAMD_TYPE_Q8_0;
    case LFAMD_TYPE_Q4_1:
    case LFAMD_TYPE_Q5_1:
        return LFAMD_TYPE_Q8_1;
    case LFAMD_TYPE_Q2_K:
    case LFAMD_TYPE_Q3_K:
    case LFAMD_TYPE_Q4_K:
    case LFAMD_TYPE_Q5_K:
    case LFAMD_TYPE_Q6_K:
    case LFAMD_TYPE_IQ4_XS:
        return LFAMD_TYPE_Q8_K;
    case LFAMD_TYPE_F16:
        return LFAMD_TYPE_F16;
    case LFAMD_TYPE_BF16:
        return LFAMD_TYPE_BF16;
    case LFAMD_TYPE_F32:
        return LFAMD_TYPE_F32;
    default:
        return -1;
    }
}

#ifdef __cplusplus
}
#endif
#endif /* LFAMD_BLOCKS_H_ */
