/*
 * oracle.c — CPU restatement of llamafile's quantized-matmul hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  "parity unpinned" for quantized types:
 * the reference holds no golden vectors for them and cannot be built in this image.
 *
 * Build with -ffp-contract=off: every fused operation below is an explicit fmaf().
 */
#include "oracle.h"
#include "../include/lfamd_blocks.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* conversions                                                                                 */

float ora_fp16_to_fp32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ff;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) {
            out = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while (!(man & 0x400));
            man &= 0x3ff;
            out = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        out = sign | 0x7f800000u | (man << 13);
    } else {
        out = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &out, 4);
    return f;
}

uint16_t ora_fp32_to_fp16(float f) { /* round to nearest even, like F16C vcvtps2ph */
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        if (ax > 0x7f800000u)
            return (uint16_t)(sign | 0x7e00 | ((ax >> 13) & 0x3ff));
        return (uint16_t)(sign | 0x7c00);
    }
    if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00);
    }
    if (ax < 0x33000001u) { /* < 2^-25 (or == 2^-25 ties to even 0) */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    if (e < -14) { /* subnormal half */
        int shift = -14 - e + 13; /* total right shift of the 24-bit significand */
        uint32_t r = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1)))
            r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)(e + 15) << 10) | ((man >> 13) & 0x3ff);
    uint32_t rem = man & 0x1fff;
    if (rem > 0x1000 || (rem == 0x1000 && (r & 1)))
        r++;
    return (uint16_t)(sign | r);
}

float ora_bf16_to_fp32(uint16_t h) {
    uint32_t x = (uint32_t)h << 16;
    float f;
    memcpy(&f, &x, 4);
    return f;
}

uint16_t ora_fp32_to_bf16(float f) { /* ggml_compute_fp32_to_bf16: RNE, NaN quieted */
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u)
        return (uint16_t)((x >> 16) | 64);
    return (uint16_t)((x + (0x7fffu + ((x >> 16) & 1))) >> 16);
}

ora_variant ora_variant_zen4(void) {
    ora_variant v = {32, 16, 0, 1};
    return v;
}
ora_variant ora_variant_avx2(void) {
    ora_variant v = {16, 8, 0, 1};
    return v;
}

long long ora_ulp_diff(float a, float b) {
    uint32_t ia, ib;
    memcpy(&ia, &a, 4);
    memcpy(&ib, &b, 4);
    long long d = (long long)ia - (long long)ib; /* sgemm_matmul_test.cpp:78-82 */
    return d < 0 ? -d : d;
}

/* ------------------------------------------------------------------------------------------ */
/* activation quantisers                                                                       */

static inline int nearest_int(float fval) { /* upstream ggml-quants.c nearest_int */
    float val = fval + 12582912.f;
    int i;
    memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

void ora_quantize_row_q8_0(const float *x, void *vy, long k) {
    lfamd_block_q8_0 *y = (lfamd_block_q8_0 *)vy;
    long nb = k / 32;
    for (long i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) {
            float v = fabsf(x[i * 32 + j]);
            if (v > amax)
                amax = v;
        }
        float d = amax / 127.0f; /* ((1 << 7) - 1) */
        float id = d ? 1.0f / d : 0.0f;
        y[i].d = ora_fp32_to_fp16(d);
        for (int j = 0; j < 32; j++)
            y[i].qs[j] = (int8_t)roundf(x[i * 32 + j] * id);
    }
}

void ora_quantize_row_q8_1(const float *x, void *vy, long k) {
    lfamd_block_q8_1 *y = (lfamd_block_q8_1 *)vy;
    long nb = k / 32;
    for (long i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) {
            float v = fabsf(x[i * 32 + j]);
            if (v > amax)
                amax = v;
        }
        float d = amax / 127.0f;
        float id = d ? 1.0f / d : 0.0f;
        y[i].d = ora_fp32_to_fp16(d);
        int sum = 0;
        for (int j = 0; j < 32; j++) {
            int8_t q = (int8_t)roundf(x[i * 32 + j] * id);
            y[i].qs[j] = q;
            sum += q;
        }
        y[i].s = ora_fp32_to_fp16((float)sum * d);
    }
}

void ora_quantize_row_q8_K(const float *x, void *vy, long k) {
    lfamd_block_q8_K *y = (lfamd_block_q8_K *)vy;
    long nb = k / 256;
    for (long i = 0; i < nb; i++) {
        float max = 0, amax = 0;
        for (int j = 0; j < 256; j++) {
            float ax = fabsf(x[j]);
            if (ax > amax) {
                amax = ax;
                max = x[j];
            }
        }
        if (!amax) {
            y[i].d = 0;
            memset(y[i].qs, 0, 256);
            memset(y[i].bsums, 0, sizeof(y[i].bsums));
            x += 256;
            continue;
        }
        const float iscale = -128.f / max;
        for (int j = 0; j < 256; j++) {
            int v = nearest_int(iscale * x[j]);
            y[i].qs[j] = (int8_t)(v > 127 ? 127 : v);
        }
        for (int j = 0; j < 16; j++) {
            int sum = 0;
            for (int ii = 0; ii < 16; ii++)
                sum += y[i].qs[j * 16 + ii];
            y[i].bsums[j] = (int16_t)sum;
        }
        y[i].d = 1 / iscale;
        x += 256;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* integer codes of one block: q[] such that w = scale_sub * q - min_sub (exact ints)         */

/* get_scale_min_k4 (ggml-cuda.cu.patch:3311-3318) */
static inline void scale_min_k4(int j, const uint8_t *q, int *d, int *m) {
    if (j < 4) {
        *d = q[j] & 63;
        *m = q[j + 4] & 63;
    } else {
        *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4);
        *m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4);
    }
}

static const int8_t kvalues_iq4nl[16] = {-127, -104, -83, -65, -49, -35, -22, -10,
                                         1,    13,   25,  38,  53,  69,  89,  113};

/* A K-quant super-block reduced to: 256 integer codes q, 16 integer sub-block scales sc (one
 * per 16 weights; types with 32-wide sub-blocks repeat each scale twice), 16 integer mins mn,
 * and float d, dmin:  w[l] = d*sc[l/16]*q[l] - dmin*mn[l/16]. */
typedef struct {
    int q[256];
    int sc[16];
    int mn[16];
    float d, dmin;
} kblock;

static void unpack_q4_K(const lfamd_block_q4_K *x, kblock *o) {
    o->d = ora_fp16_to_fp32(x->d);
    o->dmin = ora_fp16_to_fp32(x->dmin);
    for (int j = 0; j < 8; j++) {
        int sc, m;
        scale_min_k4(j, x->scales, &sc, &m);
        o->sc[2 * j] = o->sc[2 * j + 1] = sc;
        o->mn[2 * j] = o->mn[2 * j + 1] = m;
    }
    for (int c = 0; c < 4; c++)
        for (int l = 0; l < 32; l++) { /* ggml-cuda.cu.patch:3322-3363 */
            o->q[64 * c + l] = x->qs[32 * c + l] & 0xF;
            o->q[64 * c + 32 + l] = x->qs[32 * c + l] >> 4;
        }
}

static void unpack_q5_K(const lfamd_block_q5_K *x, kblock *o) {
    o->d = ora_fp16_to_fp32(x->d);
    o->dmin = ora_fp16_to_fp32(x->dmin);
    for (int j = 0; j < 8; j++) {
        int sc, m;
        scale_min_k4(j, x->scales, &sc, &m);
        o->sc[2 * j] = o->sc[2 * j + 1] = sc;
        o->mn[2 * j] = o->mn[2 * j + 1] = m;
    }
    for (int c = 0; c < 4; c++)
        for (int l = 0; l < 32; l++) { /* ggml-cuda.cu.patch:3366-3417 */
            o->q[64 * c + l] = (x->qs[32 * c + l] & 0xF) + ((x->qh[l] >> (2 * c)) & 1) * 16;
            o->q[64 * c + 32 + l] = (x->qs[32 * c + l] >> 4) + ((x->qh[l] >> (2 * c + 1)) & 1) * 16;
        }
}

static void unpack_q6_K(const lfamd_block_q6_K *x, kblock *o) {
    o->d = ora_fp16_to_fp32(x->d);
    o->dmin = 0;
    for (int j = 0; j < 16; j++) {
        o->sc[j] = x->scales[j];
        o->mn[j] = 0;
    }
    for (int p = 0; p < 2; p++)
        for (int l = 0; l < 32; l++) { /* ggml-cuda.cu.patch:3422-3471 */
            const uint8_t *ql = x->ql + 64 * p;
            uint8_t qh = x->qh[32 * p + l];
            o->q[128 * p + l] = (int)((ql[l] & 0xF) | (((qh >> 0) & 3) << 4)) - 32;
            o->q[128 * p + 32 + l] = (int)((ql[32 + l] & 0xF) | (((qh >> 2) & 3) << 4)) - 32;
            o->q[128 * p + 64 + l] = (int)((ql[l] >> 4) | (((qh >> 4) & 3) << 4)) - 32;
            o->q[128 * p + 96 + l] = (int)((ql[32 + l] >> 4) | (((qh >> 6) & 3) << 4)) - 32;
        }
}

static void unpack_q2_K(const lfamd_block_q2_K *x, kblock *o) {
    o->d = ora_fp16_to_fp32(x->d);
    o->dmin = ora_fp16_to_fp32(x->dmin);
    for (int j = 0; j < 16; j++) {
        o->sc[j] = x->scales[j] & 0xF;
        o->mn[j] = x->scales[j] >> 4;
    }
    for (int n = 0; n < 2; n++)
        for (int l = 0; l < 32; l++) { /* ggml-cuda.cu.patch:3217-3240; is = 8n + l/16 (+0,2,4,6) */
            uint8_t q = x->qs[32 * n + l];
            o->q[128 * n + l] = (q >> 0) & 3;
            o->q[128 * n + 32 + l] = (q >> 2) & 3;
            o->q[128 * n + 64 + l] = (q >> 4) & 3;
            o->q[128 * n + 96 + l] = (q >> 6) & 3;
        }
}

static void unpack_q3_K(const lfamd_block_q3_K *x, kblock *o) {
    o->d = ora_fp16_to_fp32(x->d);
    o->dmin = 0;
    for (int is = 0; is < 16; is++) { /* ggml-cuda.cu.patch:3262-3308 */
        int us = is < 4    ? (x->scales[is] & 0xF) | (((x->scales[is + 8] >> 0) & 3) << 4)
                 : is < 8  ? (x->scales[is] & 0xF) | (((x->scales[is + 4] >> 2) & 3) << 4)
                 : is < 12 ? (x->scales[is - 8] >> 4) | (((x->scales[is] >> 4) & 3) << 4)
                           : (x->scales[is - 8] >> 4) | (((x->scales[is - 4] >> 6) & 3) << 4);
        o->sc[is] = us - 32;
        o->mn[is] = 0;
    }
    for (int n = 0; n < 2; n++)
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 32; l++) {
                uint8_t m = (uint8_t)(1 << (4 * n + j));
                int q = (x->qs[32 * n + l] >> (2 * j)) & 3;
                o->q[128 * n + 32 * j + l] = q - ((x->hmask[l] & m) ? 0 : 4);
            }
}

static void unpack_iq4_xs(const lfamd_block_iq4_xs *x, kblock *o) {
    o->d = ora_fp16_to_fp32(x->d);
    o->dmin = 0;
    for (int ib = 0; ib < 8; ib++) { /* ggml-cuda.cu.patch:3684-3699 */
        int ls = ((x->scales_l[ib / 2] >> (4 * (ib % 2))) & 0xf) | (((x->scales_h >> (2 * ib)) & 3) << 4);
        o->sc[2 * ib] = o->sc[2 * ib + 1] = ls - 32;
        o->mn[2 * ib] = o->mn[2 * ib + 1] = 0;
        for (int j = 0; j < 16; j++) {
            o->q[32 * ib + j] = kvalues_iq4nl[x->qs[16 * ib + j] & 0xf];
            o->q[32 * ib + 16 + j] = kvalues_iq4nl[x->qs[16 * ib + j] >> 4];
        }
    }
}

static int unpack_kblock(int type, const void *blk, kblock *o) {
    switch (type) {
    case LFAMD_TYPE_Q2_K:
        unpack_q2_K((const lfamd_block_q2_K *)blk, o);
        return 1;
    case LFAMD_TYPE_Q3_K:
        unpack_q3_K((const lfamd_block_q3_K *)blk, o);
        return 1;
    case LFAMD_TYPE_Q4_K:
        unpack_q4_K((const lfamd_block_q4_K *)blk, o);
        return 1;
    case LFAMD_TYPE_Q5_K:
        unpack_q5_K((const lfamd_block_q5_K *)blk, o);
        return 1;
    case LFAMD_TYPE_Q6_K:
        unpack_q6_K((const lfamd_block_q6_K *)blk, o);
        return 1;
    case LFAMD_TYPE_IQ4_XS:
        unpack_iq4_xs((const lfamd_block_iq4_xs *)blk, o);
        return 1;
    default:
        return 0;
    }
}

/* legacy 32-wide blocks: w[l] = d*q[l] + m  (m = 0 for *_0 types) */
typedef struct {
    int q[32];
    float d, m;
} lblock;

static int unpack_lblock(int type, const void *blk, lblock *o) {
    switch (type) {
    case LFAMD_TYPE_Q4_0: { /* tinyblas_cpu.h:977-983 */
        const lfamd_block_q4_0 *x = (const lfamd_block_q4_0 *)blk;
        o->d = ora_fp16_to_fp32(x->d);
        o->m = 0;
        for (int j = 0; j < 16; j++) {
            o->q[j] = (x->qs[j] & 15) - 8;
            o->q[j + 16] = (x->qs[j] >> 4) - 8;
        }
        return 1;
    }
    case LFAMD_TYPE_Q4_1: { /* iqk_mul_mat.inc:1241-1246 */
        const lfamd_block_q4_1 *x = (const lfamd_block_q4_1 *)blk;
        o->d = ora_fp16_to_fp32(x->d);
        o->m = ora_fp16_to_fp32(x->m);
        for (int j = 0; j < 16; j++) {
            o->q[j] = x->qs[j] & 15;
            o->q[j + 16] = x->qs[j] >> 4;
        }
        return 1;
    }
    case LFAMD_TYPE_Q5_0: { /* iqk_mul_mat.inc:1248-1283; ggml-cuda.cu.patch:2846-2877 */
        const lfamd_block_q5_0 *x = (const lfamd_block_q5_0 *)blk;
        uint32_t qh;
        memcpy(&qh, x->qh, 4);
        o->d = ora_fp16_to_fp32(x->d);
        o->m = 0;
        for (int j = 0; j < 16; j++) {
            o->q[j] = ((x->qs[j] & 15) | (((qh >> j) & 1) << 4)) - 16;
            o->q[j + 16] = ((x->qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) - 16;
        }
        return 1;
    }
    case LFAMD_TYPE_Q5_1: {
        const lfamd_block_q5_1 *x = (const lfamd_block_q5_1 *)blk;
        uint32_t qh;
        memcpy(&qh, x->qh, 4);
        o->d = ora_fp16_to_fp32(x->d);
        o->m = ora_fp16_to_fp32(x->m);
        for (int j = 0; j < 16; j++) {
            o->q[j] = (x->qs[j] & 15) | (((qh >> j) & 1) << 4);
            o->q[j + 16] = (x->qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
        }
        return 1;
    }
    case LFAMD_TYPE_Q8_0: { /* tinyblas_cpu.h:973-975 */
        const lfamd_block_q8_0 *x = (const lfamd_block_q8_0 *)blk;
        o->d = ora_fp16_to_fp32(x->d);
        o->m = 0;
        for (int j = 0; j < 32; j++)
            o->q[j] = x->qs[j];
        return 1;
    }
    default:
        return 0;
    }
}

int ora_dequantize_row(int type, const void *vx, float *y, long k) {
    const char *x = (const char *)vx;
    size_t ts = lfamd_type_size(type);
    int bs = lfamd_blck_size(type);
    if (!bs || k % bs)
        return 0;
    switch (type) {
    case LFAMD_TYPE_F32:
        memcpy(y, vx, (size_t)k * 4);
        return 1;
    case LFAMD_TYPE_F16:
        for (long i = 0; i < k; i++)
            y[i] = ora_fp16_to_fp32(((const uint16_t *)vx)[i]);
        return 1;
    case LFAMD_TYPE_BF16:
        for (long i = 0; i < k; i++)
            y[i] = ora_bf16_to_fp32(((const uint16_t *)vx)[i]);
        return 1;
    default:
        break;
    }
    if (bs == 256) {
        kblock kb;
        for (long b = 0; b < k / 256; b++) {
            if (!unpack_kblock(type, x + b * ts, &kb))
                return 0;
            for (int l = 0; l < 256; l++)
                y[b * 256 + l] = kb.d * (float)kb.sc[l / 16] * (float)kb.q[l] - kb.dmin * (float)kb.mn[l / 16];
        }
        return 1;
    }
    lblock lb;
    for (long b = 0; b < k / 32; b++) {
        if (!unpack_lblock(type, x + b * ts, &lb))
            return 0;
        for (int l = 0; l < 32; l++)
            y[b * 32 + l] = lb.d * (float)lb.q[l] + lb.m;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* iqk_mul_mat: exact integer block dots, f32 scales (iqk_mul_mat.inc:601-643, 949-995,       */
/* 1131-1166).  The reference's 8/16-lane f32 partial-sum order is SIMD specific; this        */
/* restatement keeps the integer part exact and accumulates one f32 term per block, which     */
/* SURVEY.md §8c measured to agree with the compiled reference to ~1e-7 relative.             */

/* One A row unpacked once (so the columns loop below only does integer dots). */
typedef struct {
    long nb;      /* blocks in the row */
    int kq;       /* 1: K-quant super-blocks, 0: legacy 32-blocks */
    int8_t *q;    /* [nb*bs] integer codes; IQ4_XS codes are in [-127,113], all others fit too */
    int16_t *sc;  /* K: [nb*16] sub-block scales */
    int16_t *mn;  /* K: [nb*16] sub-block mins */
    float *d;     /* [nb] */
    float *dm;    /* K: dmin, legacy: m */
} iqk_row;

static void iqk_row_alloc(iqk_row *r, int typeA, long ne00) {
    r->kq = lfamd_blck_size(typeA) == 256;
    r->nb = ne00 / (r->kq ? 256 : 32);
    r->q = (int8_t *)malloc((size_t)ne00);
    r->sc = (int16_t *)malloc(sizeof(int16_t) * (size_t)(r->nb * 16));
    r->mn = (int16_t *)malloc(sizeof(int16_t) * (size_t)(r->nb * 16));
    r->d = (float *)malloc(sizeof(float) * (size_t)r->nb);
    r->dm = (float *)malloc(sizeof(float) * (size_t)r->nb);
}

static void iqk_row_free(iqk_row *r) {
    free(r->q), free(r->sc), free(r->mn), free(r->d), free(r->dm);
}

static void iqk_row_unpack(iqk_row *r, int typeA, const char *arow) {
    size_t ts = lfamd_type_size(typeA);
    if (r->kq) {
        kblock kb;
        for (long i = 0; i < r->nb; i++) {
            unpack_kblock(typeA, arow + i * ts, &kb);
            for (int l = 0; l < 256; l++)
                r->q[i * 256 + l] = (int8_t)kb.q[l];
            for (int s = 0; s < 16; s++) {
                r->sc[i * 16 + s] = (int16_t)kb.sc[s];
                r->mn[i * 16 + s] = (int16_t)kb.mn[s];
            }
            r->d[i] = kb.d;
            r->dm[i] = kb.dmin;
        }
    } else {
        lblock lb;
        for (long i = 0; i < r->nb; i++) {
            unpack_lblock(typeA, arow + i * ts, &lb);
            for (int l = 0; l < 32; l++)
                r->q[i * 32 + l] = (int8_t)lb.q[l];
            r->d[i] = lb.d;
            r->dm[i] = lb.m;
        }
    }
}

static float iqk_dot_row(const iqk_row *r, int typeA, const char *brow) {
    if (r->kq) {
        const lfamd_block_q8_K *y = (const lfamd_block_q8_K *)brow;
        float accd = 0.0f, accm = 0.0f;
        for (long i = 0; i < r->nb; i++) {
            const int8_t *q = r->q + i * 256;
            int32_t sumi = 0, summ = 0;
            for (int s = 0; s < 16; s++) {
                int32_t dot = 0;
                for (int l = 0; l < 16; l++)
                    dot += (int32_t)q[16 * s + l] * (int32_t)y[i].qs[16 * s + l];
                sumi += r->sc[i * 16 + s] * dot;
                summ += r->mn[i * 16 + s] * (int32_t)y[i].bsums[s];
            }
            /* accd += (d*d8)*sumi ; accm += (-dmin*d8)*summ  (iqk_mul_mat.inc:284-291, 632) */
            accd = fmaf(r->d[i] * y[i].d, (float)sumi, accd);
            accm = fmaf(-r->dm[i] * y[i].d, (float)summ, accm);
        }
        return accd + accm;
    }
    float acc = 0.0f, accm = 0.0f;
    int bt = lfamd_vec_dot_type(typeA);
    for (long i = 0; i < r->nb; i++) {
        const int8_t *q8;
        float dy, sy = 0.0f;
        if (bt == LFAMD_TYPE_Q8_1) {
            const lfamd_block_q8_1 *y = (const lfamd_block_q8_1 *)brow + i;
            q8 = y->qs;
            dy = ora_fp16_to_fp32(y->d);
            sy = ora_fp16_to_fp32(y->s);
        } else {
            const lfamd_block_q8_0 *y = (const lfamd_block_q8_0 *)brow + i;
            q8 = y->qs;
            dy = ora_fp16_to_fp32(y->d);
        }
        int32_t dot = 0;
        for (int l = 0; l < 32; l++)
            dot += (int32_t)r->q[i * 32 + l] * (int32_t)q8[l];
        acc = fmaf(r->d[i] * dy, (float)dot, acc); /* iqk_mul_mat.inc:1143-1146 */
        if (bt == LFAMD_TYPE_Q8_1)
            accm += r->dm[i] * sy; /* MinusType1, iqk_mul_mat.inc:1110-1127 */
    }
    return acc + accm;
}

static int iqk_supported(int typeA) { /* x86 set_mul_mat, iqk_mul_mat.inc:1408-1463 */
    switch (typeA) {
    case LFAMD_TYPE_Q2_K:
    case LFAMD_TYPE_Q3_K:
    case LFAMD_TYPE_Q4_K:
    case LFAMD_TYPE_Q5_K:
    case LFAMD_TYPE_Q6_K:
    case LFAMD_TYPE_IQ4_XS:
    case LFAMD_TYPE_Q4_0:
    case LFAMD_TYPE_Q4_1:
    case LFAMD_TYPE_Q5_0:
    case LFAMD_TYPE_Q5_1:
        return 1;
    default:
        return 0;
    }
}

int ora_iqk_mul_mat(long Nx, long Ny, long ne00, int typeA, const void *A, const void *B, float *C,
                    long stride_C, int ith, int nth) {
    if (!iqk_supported(typeA))
        return 0;
    size_t row_size_qx = lfamd_row_size(typeA, ne00);
    size_t row_size_q8 = lfamd_row_size(lfamd_vec_dot_type(typeA), ne00);
    long nrc_x = (Nx + nth - 1) / nth; /* iqk_mul_mat.inc:193-195 */
    long first_x = ith * nrc_x;
    if (first_x + nrc_x > Nx)
        nrc_x = Nx - first_x;
    iqk_row r;
    iqk_row_alloc(&r, typeA, ne00);
    for (long ix = first_x; ix < first_x + nrc_x; ix++) {
        iqk_row_unpack(&r, typeA, (const char *)A + ix * row_size_qx);
        for (long iy = 0; iy < Ny; iy++)
            C[iy * stride_C + ix] = iqk_dot_row(&r, typeA, (const char *)B + iy * row_size_q8);
    }
    iqk_row_free(&r);
    return 1;
}

int ora_iqk_mul_mat_moe(long Nx, long Ny, long ne00, int ne11, int typeA, const void *A,
                        const void *B, float *C, long nb1, long nb2, const void *vrow_mapping,
                        int ith, int nth) {
    const int32_t *map = (const int32_t *)vrow_mapping; /* {i1, i2} pairs, iqk_mul_mat.inc:69-72 */
    if (!iqk_supported(typeA))
        return 0;
    size_t row_size_qx = lfamd_row_size(typeA, ne00);
    size_t row_size_q8 = lfamd_row_size(lfamd_vec_dot_type(typeA), ne00);
    long nrc_x = (Nx + nth - 1) / nth;
    long first_x = ith * nrc_x;
    if (first_x + nrc_x > Nx)
        nrc_x = Nx - first_x;
    iqk_row r;
    iqk_row_alloc(&r, typeA, ne00);
    for (long ix = first_x; ix < first_x + nrc_x; ix++) {
        iqk_row_unpack(&r, typeA, (const char *)A + ix * row_size_qx);
        for (long iy = 0; iy < Ny; iy++) {
            int i1 = map[2 * iy], i2 = map[2 * iy + 1];
            /* DataInfo::src1_row / dst_row, iqk_mul_mat.inc:84-101 */
            const char *brow = (const char *)B + ((size_t)(i1 % ne11) + (size_t)i2 * ne11) * row_size_q8;
            float *crow = C + (size_t)i1 * (nb1 / sizeof(float)) + (size_t)i2 * (nb2 / sizeof(float));
            crow[ix] = iqk_dot_row(&r, typeA, brow);
        }
    }
    iqk_row_free(&r);
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* tinyBLAS_Q0_AVX2: bit-exact restatement (tinyblas_cpu.h:780-1005)                          */

typedef struct {
    long m, n;
    const ora_variant *v;
    uint8_t *mode;    /* optional: precise map */
    /* compute state (NULL mode-only) */
    int Atype;
    const char *A;
    long lda;
    const char *B;
    long ldb;
    const void *const *Bptr;
    float *C;
    long ldc;
    float *const *Cptr;
    long k;
    int ith, nth;
    int compute;
} q0ctx;

/* 32 signed codes of an A block as the reference's load() produces them */
static void q0_load_a(int Atype, const void *blk, int8_t q[32], float *d) {
    if (Atype == LFAMD_TYPE_Q8_0) {
        const lfamd_block_q8_0 *b = (const lfamd_block_q8_0 *)blk;
        memcpy(q, b->qs, 32);
        *d = ora_fp16_to_fp32(b->d);
    } else {
        const lfamd_block_q4_0 *b = (const lfamd_block_q4_0 *)blk;
        for (int j = 0; j < 16; j++) {
            q[j] = (int8_t)((b->qs[j] & 15) - 8);
            q[j + 16] = (int8_t)((b->qs[j] >> 4) - 8);
        }
        *d = ora_fp16_to_fp32(b->d);
    }
}

/* updot(sign(a,a), sign(b,a)) → 8 f32 lanes, lane j = bytes 4j..4j+3 (tinyblas_cpu.h:949-953,
 * 985-993).  _mm256_sign_epi8 wraps -(-128) to -128; the u8 operand reads |−128| as 128. */
static void q0_updot(const int8_t qa[32], const int8_t qb[32], float out[8]) {
    for (int j = 0; j < 8; j++) {
        int32_t s = 0;
        for (int t = 0; t < 4; t++) {
            int8_t a = qa[4 * j + t], b = qb[4 * j + t];
            uint8_t ua = (uint8_t)(a < 0 ? -a : a);
            int8_t sb = a < 0 ? (int8_t)(-b) : (a == 0 ? 0 : b);
            s += (int32_t)ua * (int32_t)sb;
        }
        out[j] = (float)s;
    }
}

static float q0_hsum8(const float x[8]) { /* tinyblas_cpu.h:277-296 */
    float t0 = x[0] + x[4], t1 = x[1] + x[5], t2 = x[2] + x[6], t3 = x[3] + x[7];
    return (t0 + t2) + (t1 + t3);
}

static float q0_output(const q0ctx *c, long i, long j, int precise) {
    size_t tsa = lfamd_type_size(c->Atype);
    const char *arow = c->A + (size_t)i * c->lda * tsa;
    const lfamd_block_q8_0 *brow =
        c->Bptr ? (const lfamd_block_q8_0 *)c->Bptr[j] : (const lfamd_block_q8_0 *)c->B + (size_t)j * c->ldb;
    float Cv[8] = {0}, Ce[8] = {0};
    int8_t qa[32];
    float b[8];
    for (long l = 0; l < c->k; l++) {
        float da;
        q0_load_a(c->Atype, arow + l * tsa, qa, &da);
        float a = da * ora_fp16_to_fp32(brow[l].d);
        q0_updot(qa, brow[l].qs, b);
        if (precise) { /* madder, tinyblas_cpu.h:203-209 */
            for (int t = 0; t < 8; t++) {
                float y = c->v->kahan_contract ? fmaf(a, b[t], -Ce[t]) : (a * b[t]) - Ce[t];
                float s = Cv[t] + y;
                Ce[t] = (s - Cv[t]) - y;
                Cv[t] = s;
            }
        } else {
            for (int t = 0; t < 8; t++)
                Cv[t] = fmaf(a, b[t], Cv[t]);
        }
    }
    return q0_hsum8(Cv);
}

static void q0_gemm_region(q0ctx *c, long m0, long m, long n0, long n, int RM, int RN, int precise) {
    long ytiles = RM > 1 ? (m - m0) / RM : 1; /* tinyblas_cpu.h:934-946 */
    long xtiles = RN > 1 ? (n - n0) / RN : 1;
    long tiles = xtiles * ytiles;
    long start = 0, end = tiles;
    if (c->compute) {
        long duty = (tiles + c->nth - 1) / c->nth;
        start = duty * c->ith;
        end = start + duty;
        if (end > tiles)
            end = tiles;
    }
    for (long job = start; job < end; ++job) {
        long ii = m0 + job / xtiles * RM;
        long jj = n0 + job % xtiles * RN;
        for (int j = 0; j < RN; j++)
            for (int i = 0; i < RM; i++) {
                if (c->mode)
                    c->mode[(jj + j) * c->m + (ii + i)] = (uint8_t)precise;
                if (c->compute) {
                    float r = q0_output(c, ii + i, jj + j, precise);
                    if (c->Cptr)
                        c->Cptr[jj + j][ii + i] = r;
                    else
                        c->C[(jj + j) * c->ldc + (ii + i)] = r;
                }
            }
    }
}

#define MIN_(a, b) ((a) < (b) ? (a) : (b))

static void q0_mnpack(q0ctx *c, long m0, long m, long n0, long n) {
    long mc, nc;
    int pr;
    int P = c->v->precise;
    if (c->v->vector_registers == 32) { /* tinyblas_cpu.h:797-862 */
        switch ((MIN_(m - m0, 3) << 4) | MIN_(n - n0, 3)) {
        case 0x33:
            mc = 3, nc = 3, pr = P;
            break;
        case 0x32:
        case 0x23:
        case 0x22:
            mc = 2, nc = 2, pr = P;
            break;
        case 0x31:
        case 0x21:
            mc = 2, nc = 1, pr = 1;
            break;
        case 0x13:
        case 0x12:
            mc = 1, nc = 2, pr = 1;
            break;
        case 0x11:
            mc = 1, nc = 1, pr = 1;
            break;
        default:
            return;
        }
    } else if (!P) { /* tinyblas_cpu.h:866-902 */
        switch ((MIN_(m - m0, 3) << 4) | MIN_(n - n0, 2)) {
        case 0x32:
            mc = 3, nc = 2, pr = 0;
            break;
        case 0x23: /* unreachable with MIN(n,2); kept for fidelity */
            mc = 2, nc = 3, pr = 0;
            break;
        case 0x22:
            mc = 2, nc = 2, pr = 0;
            break;
        case 0x31:
        case 0x21:
            mc = 2, nc = 1, pr = 0;
            break;
        case 0x12:
            mc = 1, nc = 2, pr = 0;
            break;
        case 0x11:
            mc = 1, nc = 1, pr = 0;
            break;
        default:
            return;
        }
    } else { /* tinyblas_cpu.h:903-925 */
        switch ((MIN_(m - m0, 2) << 4) | MIN_(n - n0, 1)) {
        case 0x21:
            mc = 2, nc = 1, pr = 1;
            break;
        case 0x12: /* unreachable with MIN(n,1) */
            mc = 1, nc = 2, pr = 1;
            break;
        case 0x11:
            mc = 1, nc = 1, pr = 1;
            break;
        default:
            return;
        }
    }
    q0_gemm_region(c, m0, m, n0, n, (int)mc, (int)nc, pr);
    long mp = m0 + (m - m0) / mc * mc; /* tinyblas_cpu.h:928-931 */
    long np = n0 + (n - n0) / nc * nc;
    q0_mnpack(c, mp, m, n0, np);
    q0_mnpack(c, m0, m, np, n);
}

void ora_q0_precise_map(long m, long n, const ora_variant *v, uint8_t *mode) {
    q0ctx c;
    memset(&c, 0, sizeof(c));
    c.m = m, c.n = n, c.v = v, c.mode = mode, c.compute = 0;
    memset(mode, 0xff, (size_t)(m * n));
    q0_mnpack(&c, 0, m, 0, n);
}

int ora_q0_gemm(long m, long n, long k, int Atype, const void *A, long lda, const void *B, long ldb,
                const void *const *Bptr, float *C, long ldc, float *const *Cptr, int ith, int nth,
                const ora_variant *v) {
    if (Atype != LFAMD_TYPE_Q8_0 && Atype != LFAMD_TYPE_Q4_0)
        return 0;
    q0ctx c;
    memset(&c, 0, sizeof(c));
    c.m = m, c.n = n, c.v = v, c.Atype = Atype, c.A = (const char *)A, c.lda = lda;
    c.B = (const char *)B, c.ldb = ldb, c.Bptr = Bptr, c.C = C, c.ldc = ldc, c.Cptr = Cptr;
    c.k = k, c.ith = ith, c.nth = nth, c.compute = 1;
    q0_mnpack(&c, 0, m, 0, n);
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* tinyBLAS<> float GEMM (tinyblas_cpu.h:419-613)                                             */

static inline float load_elem(int type, const void *base, long idx) {
    switch (type) {
    case LFAMD_TYPE_F32:
        return ((const float *)base)[idx];
    case LFAMD_TYPE_F16:
        return ora_fp16_to_fp32(((const uint16_t *)base)[idx]);
    default:
        return ora_bf16_to_fp32(((const uint16_t *)base)[idx]);
    }
}

static float hsum_lanes(const float *x, int kn) {
    float t[16];
    memcpy(t, x, sizeof(float) * kn);
    if (kn == 16) { /* _mm512_reduce_add_ps tree */
        for (int i = 0; i < 8; i++)
            t[i] = t[i] + t[i + 8]; /* hi256 + lo256 (order of operands irrelevant for +) */
        for (int i = 0; i < 4; i++)
            t[i] = t[i] + t[i + 4];
        for (int i = 0; i < 2; i++)
            t[i] = t[i] + t[i + 2];
        return t[0] + t[1];
    }
    return q0_hsum8(t); /* kn == 8: tinyblas_cpu.h:277-296 */
}

static int bsr_(unsigned long x) {
    int r = -1;
    while (x) {
        r++;
        x >>= 1;
    }
    return r;
}

static float float_dot(int Atype, const void *arow, int Btype, const void *brow, long k, int KN) {
    const int CHUNK = 8; /* tinyblas_cpu.h:52 */
    float stack[64][16];
    float Cv[16];
    long chunk;
    size_t sp = 0;
    int rule, step = 2;
    for (chunk = 0; chunk + KN * CHUNK * 4 <= k; chunk += KN * CHUNK * 4, step += 2, ++sp) {
        for (int t = 0; t < KN; t++)
            Cv[t] = 0;
        for (long l = 0; l < KN * CHUNK * 4; l += KN)
            for (int t = 0; t < KN; t++)
                Cv[t] = fmaf(load_elem(Atype, arow, chunk + l + t), load_elem(Btype, brow, chunk + l + t), Cv[t]);
        for (rule = bsr_((unsigned long)(step & -step)); --rule;) {
            --sp;
            for (int t = 0; t < KN; t++)
                Cv[t] += stack[sp][t];
        }
        memcpy(stack[sp], Cv, sizeof(float) * KN);
    }
    for (int t = 0; t < KN; t++)
        Cv[t] = 0;
    for (; chunk + KN <= k; chunk += KN)
        for (int t = 0; t < KN; t++)
            Cv[t] = fmaf(load_elem(Atype, arow, chunk + t), load_elem(Btype, brow, chunk + t), Cv[t]);
    while (sp--)
        for (int t = 0; t < KN; t++)
            Cv[t] += stack[sp][t];
    float Cf = hsum_lanes(Cv, KN);
    for (; chunk < k; ++chunk)
        Cf = fmaf(load_elem(Atype, arow, chunk), load_elem(Btype, brow, chunk), Cf);
    return Cf;
}

typedef struct {
    long m, n, k;
    int Atype, Btype;
    const char *A;
    long lda;
    const char *B;
    long ldb;
    float *C;
    long ldc;
    int ith, nth, kn, vregs;
} fctx;

static void f_gemm_region(fctx *c, long m0, long m, long n0, long n, int RM, int RN) {
    long ytiles = RM > 1 ? (m - m0) / RM : 1;
    long xtiles = RN > 1 ? (n - n0) / RN : 1;
    long tiles = xtiles * ytiles;
    long duty = (tiles + c->nth - 1) / c->nth;
    long start = duty * c->ith;
    long end = start + duty;
    if (end > tiles)
        end = tiles;
    size_t sa = lfamd_type_size(c->Atype), sb = lfamd_type_size(c->Btype);
    for (long job = start; job < end; ++job) {
        long ii = m0 + job / xtiles * RM;
        long jj = n0 + job % xtiles * RN;
        for (int j = 0; j < RN; j++)
            for (int i = 0; i < RM; i++)
                c->C[(jj + j) * c->ldc + ii + i] =
                    float_dot(c->Atype, c->A + (size_t)(ii + i) * c->lda * sa, c->Btype,
                              c->B + (size_t)(jj + j) * c->ldb * sb, c->k, c->kn);
    }
}

static void f_mnpack(fctx *c, long m0, long m, long n0, long n) {
    long mc, nc;
    if (c->vregs == 32) { /* tinyblas_cpu.h:438-483 */
        long a = MIN_(m - m0, 5), b = MIN_(n - n0, 5);
        if (a <= 0 || b <= 0)
            return;
        if (a == 5 && b == 5)
            mc = 5, nc = 5;
        else if (a >= 2 && b >= 2)
            mc = 2, nc = 2;
        else if (a >= 2 && b == 1)
            mc = 2, nc = 1;
        else if (a == 1 && b >= 2)
            mc = 1, nc = 2;
        else
            mc = 1, nc = 1;
    } else { /* tinyblas_cpu.h:485-524 */
        long a = MIN_(m - m0, 4), b = MIN_(n - n0, 3);
        if (a <= 0 || b <= 0)
            return;
        if (a == 4 && b == 3)
            mc = 4, nc = 3;
        else if (a >= 2 && b >= 2)
            mc = 2, nc = 2;
        else if (a >= 2 && b == 1)
            mc = 2, nc = 1;
        else if (a == 1 && b >= 2)
            mc = 1, nc = 2;
        else
            mc = 1, nc = 1;
    }
    f_gemm_region(c, m0, m, n0, n, (int)mc, (int)nc);
    long mp = m0 + (m - m0) / mc * mc;
    long np = n0 + (n - n0) / nc * nc;
    f_mnpack(c, mp, m, n0, np);
    f_mnpack(c, m0, m, np, n);
}

int ora_float_gemm(long m, long n, long k, int Atype, const void *A, long lda, int Btype,
                   const void *B, long ldb, float *C, long ldc, int ith, int nth,
                   const ora_variant *v) {
    fctx c = {m, n, k, Atype, Btype, (const char *)A, lda, (const char *)B, ldb, C, ldc, ith, nth, v->kn,
              v->vector_registers};
    f_mnpack(&c, 0, m, 0, n);
    return 1;
}

void ora_ansiblas_sgemm(long m, long n, long k, const float *A, long lda, const float *B, long ldb,
                        float *C, long ldc) {
    for (long j = 0; j < n; j++)
        for (long i = 0; i < m; i++) { /* ansiblas.h:27-121: 8 double lanes, then double tail */
            double v[8] = {0};
            long l = 0;
            for (; l + 8 <= k; l += 8)
                for (int t = 0; t < 8; t++)
                    v[t] = fma((double)A[lda * i + l + t], (double)B[ldb * j + l + t], v[t]);
            double s = 0;
            for (int t = 0; t < 8; t++)
                s += v[t];
            for (; l < k; l++)
                s = fma((double)A[lda * i + l], (double)B[ldb * j + l], s);
            C[ldc * j + i] = (float)s;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* f64 ground truth                                                                            */

static int dequant_b_row(int Btype, const void *brow, float *y, long k) {
    switch (Btype) {
    case LFAMD_TYPE_Q8_0: {
        const lfamd_block_q8_0 *b = (const lfamd_block_q8_0 *)brow;
        for (long i = 0; i < k; i++)
            y[i] = ora_fp16_to_fp32(b[i / 32].d) * (float)b[i / 32].qs[i % 32];
        return 1;
    }
    case LFAMD_TYPE_Q8_1: {
        const lfamd_block_q8_1 *b = (const lfamd_block_q8_1 *)brow;
        for (long i = 0; i < k; i++)
            y[i] = ora_fp16_to_fp32(b[i / 32].d) * (float)b[i / 32].qs[i % 32];
        return 1;
    }
    case LFAMD_TYPE_Q8_K: {
        const lfamd_block_q8_K *b = (const lfamd_block_q8_K *)brow;
        for (long i = 0; i < k; i++)
            y[i] = b[i / 256].d * (float)b[i / 256].qs[i % 256];
        return 1;
    }
    default:
        return ora_dequantize_row(Btype, brow, y, k);
    }
}

int ora_f64_gemm(long m, long n, long kelems, int Atype, const void *A, size_t a_row_bytes,
                 int Btype, const void *B, size_t b_row_bytes, double *C, long ldc) {
    float *a = (float *)malloc(sizeof(float) * (size_t)kelems);
    float *b = (float *)malloc(sizeof(float) * (size_t)kelems * (size_t)n);
    if (!a || !b) {
        free(a);
        free(b);
        return 0;
    }
    int ok = 1;
    for (long j = 0; j < n && ok; j++)
        ok = dequant_b_row(Btype, (const char *)B + j * b_row_bytes, b + j * kelems, kelems);
    for (long i = 0; i < m && ok; i++) {
        ok = ora_dequantize_row(Atype, (const char *)A + i * a_row_bytes, a, kelems);
        if (!ok)
            break;
        /* exact per-element products in double; note a[] is d*sc*q - dmin*mn rounded to f32 for
         * K-quants, so recompute those in double from the integer form for a true ground truth */
        if (lfamd_blck_size(Atype) == 256) {
            kblock kb;
            size_t ts = lfamd_type_size(Atype);
            for (long j = 0; j < n; j++) {
                double s = 0;
                for (long bi = 0; bi < kelems / 256; bi++) {
                    unpack_kblock(Atype, (const char *)A + i * a_row_bytes + bi * ts, &kb);
                    for (int l = 0; l < 256; l++) {
                        double w = (double)kb.d * kb.sc[l / 16] * kb.q[l] - (double)kb.dmin * kb.mn[l / 16];
                        s += w * (double)b[j * kelems + bi * 256 + l];
                    }
                }
                C[j * ldc + i] = s;
            }
        } else {
            for (long j = 0; j < n; j++) {
                double s = 0;
                for (long l = 0; l < kelems; l++)
                    s += (double)a[l] * (double)b[j * kelems + l];
                C[j * ldc + i] = s;
            }
        }
    }
    free(a);
    free(b);
    return ok;
}

/* ------------------------------------------------------------------------------------------ */
/* llamafile_sgemm dispatch (tinyblas_cpu_sgemm.inc:45-331), x86-64 AVX2+FMA builds           */

int ora_llamafile_sgemm(long m, long n, long k, const void *A, long lda, const void *B, long ldb,
                        void *C, long ldc, int ith, int nth, int Atype, int Btype, int Ctype,
                        const ora_variant *v) {
    if (m < 0 || n < 0 || k < 0 || lda < k || ldb < k || ldc < m || nth <= 0 || ith >= nth)
        return -1; /* the reference asserts (tinyblas_cpu_sgemm.inc:277-284) */
    /* iqk pre-dispatch, tinyblas_cpu_sgemm.inc:286-304 */
    if (Btype == LFAMD_TYPE_Q8_K && Ctype == LFAMD_TYPE_F32) {
        if (ora_iqk_mul_mat(m, n, k * 256, Atype, A, B, (float *)C, ldc, ith, nth))
            return 1;
    }
    if ((Btype == LFAMD_TYPE_Q8_0 || Btype == LFAMD_TYPE_Q8_1) && Ctype == LFAMD_TYPE_F32) {
        /* NB: the reference does not check that Btype matches typeA's vec_dot_type here */
        if (lfamd_vec_dot_type(Atype) == Btype &&
            ora_iqk_mul_mat(m, n, k * 32, Atype, A, B, (float *)C, ldc, ith, nth))
            return 1;
    }
    if (Ctype != LFAMD_TYPE_F32)
        return 0;
    switch (Atype) {
    case LFAMD_TYPE_F32:
        if (Btype != LFAMD_TYPE_F32)
            return 0;
        return ora_float_gemm(m, n, k, Atype, A, lda, Btype, B, ldb, (float *)C, ldc, ith, nth, v);
    case LFAMD_TYPE_BF16:
        if (v->kn == 16) { /* zen4: AVX512BF16 branch, tinyblas_cpu_sgemm.inc:66-88 */
            if (Btype == LFAMD_TYPE_F32 && n <= 2)
                return ora_float_gemm(m, n, k, Atype, A, lda, Btype, B, ldb, (float *)C, ldc, ith, nth, v);
            if (Btype == LFAMD_TYPE_F32)
                return 0;
            if (Btype != LFAMD_TYPE_BF16)
                return 0;
            /* n>1 uses _mm512_dpbf16_ps (pairwise bf16 dot, its own rounding): restated with
             * the f32 lane order; compare with tolerance, not bitwise */
            return ora_float_gemm(m, n, k, Atype, A, lda, Btype, B, ldb, (float *)C, ldc, ith, nth, v);
        }
        if (Btype != LFAMD_TYPE_F32)
            return 0;
        return ora_float_gemm(m, n, k, Atype, A, lda, Btype, B, ldb, (float *)C, ldc, ith, nth, v);
    case LFAMD_TYPE_F16:
        if (Btype == LFAMD_TYPE_F32 && n <= 2)
            return ora_float_gemm(m, n, k, Atype, A, lda, Btype, B, ldb, (float *)C, ldc, ith, nth, v);
        if (Btype == LFAMD_TYPE_F32)
            return 0; /* WANT_QUANTIZATION */
        if (Btype != LFAMD_TYPE_F16)
            return 0;
        return ora_float_gemm(m, n, k, Atype, A, lda, Btype, B, ldb, (float *)C, ldc, ith, nth, v);
    case LFAMD_TYPE_Q8_0:
    case LFAMD_TYPE_Q4_0:
        if (Btype != LFAMD_TYPE_Q8_0)
            return 0;
        return ora_q0_gemm(m, n, k, Atype, A, lda, B, ldb, NULL, (float *)C, ldc, NULL, ith, nth, v);
    default:
        return 0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* llamafile_mixmul (tinyblas_cpu_mixmul.inc:77-398)                                          */

int ora_mixmul(int wtype, const void *weights, long cols, long rows, int experts, size_t w_nb1,
               size_t w_nb2, const float *thought, int tasks, long tokens, const int32_t *plan,
               int thinkers, float *result, const ora_variant *v) {
    int vdt;
    int kn_needed;
    switch (wtype) { /* mixmuler, tinyblas_cpu_mixmul.inc:160-268 */
    case LFAMD_TYPE_F32:
        vdt = LFAMD_TYPE_F32, kn_needed = v->kn;
        break;
    case LFAMD_TYPE_F16:
        vdt = LFAMD_TYPE_F16, kn_needed = v->kn;
        break;
    case LFAMD_TYPE_BF16:
        vdt = LFAMD_TYPE_BF16, kn_needed = v->kn == 16 && !v->precise ? 32 : v->kn;
        break;
    case LFAMD_TYPE_Q4_0:
    case LFAMD_TYPE_Q8_0:
        vdt = LFAMD_TYPE_Q8_0, kn_needed = 32;
        break;
    default:
        return 0;
    }
    if (cols % kn_needed)
        return 0; /* tinyblas_cpu_mixmul.inc:272-273 */
    if (w_nb1 % lfamd_type_size(wtype))
        return 0; /* :137-138 */
    /* quantize_thought (:322-343) */
    size_t qrow = lfamd_row_size(vdt, cols);
    char *q = (char *)malloc(qrow * (size_t)tokens * (size_t)tasks);
    if (!q)
        return -1;
    for (long t = 0; t < tokens * tasks; t++) {
        const float *src = thought + t * cols;
        char *dst = q + t * qrow;
        switch (vdt) {
        case LFAMD_TYPE_F32:
            memcpy(dst, src, qrow);
            break;
        case LFAMD_TYPE_F16:
            for (long c = 0; c < cols; c++)
                ((uint16_t *)dst)[c] = ora_fp32_to_fp16(src[c]);
            break;
        case LFAMD_TYPE_BF16:
            for (long c = 0; c < cols; c++)
                ((uint16_t *)dst)[c] = ora_fp32_to_bf16(src[c]);
            break;
        default:
            ora_quantize_row_q8_0(src, dst, cols);
        }
    }
    /* build_row_pointers (:297-320) + per-expert matmul in pointer mode (:281-293) */
    const void **bp = (const void **)malloc(sizeof(void *) * (size_t)(tokens * thinkers));
    float **cp = (float **)malloc(sizeof(float *) * (size_t)(tokens * thinkers));
    for (int e = 0; e < experts; e++) {
        long count = 0;
        for (long token = 0; token < tokens; token++)
            for (int th = 0; th < thinkers; th++)
                if (plan[token * thinkers + th] == e) {
                    cp[count] = result + (token * thinkers + th) * rows;
                    bp[count] = q + (token * tasks + th % tasks) * qrow;
                    count++;
                }
        const char *We = (const char *)weights + (size_t)e * w_nb2;
        long lda = (long)(w_nb1 / lfamd_type_size(wtype));
        if (vdt == LFAMD_TYPE_Q8_0) {
            ora_q0_gemm(rows, count, cols / 32, wtype, We, lda, NULL, 0, bp, NULL, 0, cp, 0, 1, v);
        } else {
            for (long j = 0; j < count; j++)
                for (long i = 0; i < rows; i++)
                    cp[j][i] = float_dot(wtype, We + (size_t)i * w_nb1, vdt, bp[j], cols,
                                         kn_needed == 32 ? 16 : v->kn);
        }
    }
    free(bp);
    free(cp);
    free(q);
    return 1;
}
