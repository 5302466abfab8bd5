// lfamd_device.h — shared device-side definitions for the gfx950 kernels.
//
// PACKED WEIGHT LAYOUTS (DESIGN.md "Data layout in HBM").  The module owns the device copy of the
// weights (like ggml_backend_cuda_buffer_set_tensor, ggml-cuda.cu.patch:16971-16977) and re-lays
// it out at upload so that every kernel reads 16 contiguous bytes per lane, 1 KiB per wave
// instruction, already in MFMA-fragment order.  No padding bytes are added inside a tile: a
// packed tensor is exactly as large as the GGUF one (plus row/block round-up at the edges).
//
// P4K (Q4_K)   tile = 32 rows x 256 weights (one super-block per row) = 4608 B, stored
//              [row-tile rt][super-block b]:
//                qs  : 4 groups g x 64 lanes x 16 B.  lane = (i = lane&31, h = lane>>5); dword dd of
//                      group g is K-step t = 4g+dd and holds the 8 nibbles of weights
//                      k = 16t + 8h + j (j = 0..7) of row i, nibble j at bit 4*NIBPOS(j).
//                hdr : 32 rows x 16 B = the block's original {d, dmin, scales[12]}.
// P6K (Q6_K)   tile = 6720 B: ql (same as P4K qs, low 4 bits of each 6-bit code), qh (2 x 64 lanes x
//              16 B: upper 2 bits, see QHBIT), sc (32 rows x 16 int8), d (32 x f16).
// P80 (Q8_0)   tile = 8 rows x 4 blocks = 1088 B: qs[r][j][dd] dword = bytes 4j..4j+3 of block 4L+dd
//              of row r (lane = r*8+j reads 16 B), then d[r][dd] f16.  Chosen for the bit-exact
//              8-lane accumulation order of tinyBLAS_Q0_AVX2 (tinyblas_cpu.h:949-964).
// RAW          every other type: rows of raw GGUF blocks, row stride = ggml_row_size().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lfamd_blocks.h"

#define P4K_TILE 4608
#define P4K_HDR 4096
#define P5K_TILE 5632 // P4K (qs 4096 + hdr 512) + 1024 B of fifth bits
#define P5K_HDR 4096
#define P5K_QH 4608
// PCK: canonical per-call image the GEMM builds in its workspace for K-quants without a resident packed layout
// (Q2_K, Q3_K): P4K-form nibble image of (q - qmin), then per row 16 int8 sub-block scales, 16 uint8 mins, {d, dmin}
// PK2 / PK3: the RESIDENT images of Q2_K / Q3_K (file size: 84 / 116 bytes per 256 weights; the PCK image above is built from
// them per batch call).  qs: the PCK nibble lattice of TWO K-steps folded into one dword — codes are 2 bits wide, so K-step 2u keeps
// bits 0-1 of every nibble and K-step 2u+1 takes bits 2-3: dword u of [gsel][lane] = A | (B << 2), 2 x 64 lanes x 16 B.  Q3_K's
// third bit: dword x of [gsel][lane] holds K-steps 8 gsel + 4 x + s at bit 4 NIBPOS(j) + s (cf. Q5_K's fifth bits), 2 x 64 x 8 B.
// Then per row 16 scale bytes (Q2_K: the block's sc | mn << 4 bytes as they are; Q3_K: 16 int8 = 6-bit scale - 32) and {d, dmin}.
#define PK2_TILE 2688
#define PK2_SC 2048
#define PK2_D 2560
#define PK3_TILE 3712
#define PK3_HB 2048
#define PK3_SC 3072
#define PK3_D 3584
#define PCK_TILE 5248
#define PCK_SC 4096
#define PCK_MN 4608
#define PCK_D 5120
// PCL: per-call image for the legacy 32-block types without a resident packed layout (Q4_1, Q5_0, Q5_1): nibble image,
// eight f16 d and eight f16 m per row, fifth bits on the P5K lattice
#define PCL_TILE 6144
#define PCL_D 4096
#define PCL_M 4608
#define PCL_QH 5120
// PC8: per-call byte image for IQ4_XS (non-linear 4-bit codebook): value + 128 as one byte per weight, K-steps 2g', 2g'+1
// of lane (i, h) in the 16 bytes at g' * 1024 + lane * 16; then per row {8 int8 sub-block scales, f16 d, pad}
#define PC8_TILE 8704
#define PC8_HDR 8192
#define P6K_TILE 6720
#define P6K_QH 4096
#define P6K_SC 6144
#define P6K_D 6656
#define P80_TILE 1088
#define P80_D 1024

// nibble slot of element j (0..7) inside a K-step dword: pairs (j0,j1),(j2,j3),.. sit 16 bits apart
// so that (x & 0x000F000F) | 0x64006400 is the f16 pair (1024+q0, 1024+q1) with no shuffling.
#define NIBPOS(j) (((j) >> 1) + 4 * ((j)&1))

// Q5_K fifth bits: lane (i, h) owns one dword per group g of four K-steps; the bit of element j of K-step dd sits
// at 4*Q5HPOS(j) + dd, so (H >> dd) has the K-step's eight bits on a 4-bit lattice where
//   GEMM: (Hd & 0x00100010) joins pair (j0,j1), (Hd & 0x01000100) pair (j2,j3) [value bits 7:4],
//         ((Hd >> 8) & 0x00100010) pair (j4,j5), ((Hd << 8) & 0x01000100) pair (j6,j7)
//   GEMV: (Hd & 0x10101010) joins the bytes (j0,j4,j1,j5) of x & 0x0F0F0F0F, and
//         ((Hd >> 4) & 0x00100010) | ((Hd << 12) & 0x10001000) the bytes (j2,j6,j3,j7) of (x >> 4) & 0x0F0F0F0F
__host__ __device__ static inline int q5hpos(int j) {
    const int p[8] = {1, 5, 2, 6, 3, 7, 0, 4};
    return p[j];
}

// bit of the 2-bit high field of element j of K-step (dd&1) inside the qh dword of K-step pair dd>>1
//   half (j&1)*16, field {2,4,6,0}[j>>1] + (dd&1), two bits each
__host__ __device__ static inline int qhbit(int dd, int j) {
    const int f[4] = {2, 4, 6, 0};
    return 16 * (j & 1) + 2 * (f[j >> 1] + (dd & 1));
}

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float16_t_ __attribute__((ext_vector_type(16)));
typedef float float4_t_ __attribute__((ext_vector_type(4)));

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// streamed-once weights: non-temporal 16-byte load (MI355X_MICROARCH.md "nt-weights": -18% issue->landed)
__device__ static inline uint4 ld_nt16(const void *p) {
    u32x4_t v = __builtin_nontemporal_load((const u32x4_t *)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Bounds-checked buffer loads (T8): out-of-range lanes return 0 and make NO memory request, but the
// instruction still counts in vmcnt — so a software pipeline can issue its prefetch unconditionally
// (a descriptor with 0 records for "nothing left to fetch") and keep exact counted waits.
typedef __amdgpu_buffer_rsrc_t lfamd_rsrc;

__device__ static inline lfamd_rsrc make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
}

__device__ static inline uint4 buf_ld16_nt(lfamd_rsrc r, uint32_t off) {
    u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2 /* nt */);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ static inline uint4 buf_ld16(lfamd_rsrc r, uint32_t off) {
    u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ static inline uint2 buf_ld8(lfamd_rsrc r, uint32_t off) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return make_uint2(v.x, v.y);
}

__device__ static inline uint32_t buf_ld2(lfamd_rsrc r, uint32_t off) {
    return (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
}

__device__ static inline float h2f(uint16_t h) {
    return (float)__builtin_bit_cast(_Float16, h);
}

__device__ static inline uint16_t f2h_bits(float f) { // RNE, like F16C
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

__device__ static inline int sdot4(uint32_t a, uint32_t b, int c) { // signed i8 x signed i8
    return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
}

// get_scale_min_k4 (ggml-cuda.cu.patch:3311-3318)
__device__ static inline void scale_min_k4(int j, const uint8_t *q, int &d, int &m) {
    if (j < 4) {
        d = q[j] & 63;
        m = q[j + 4] & 63;
    } else {
        d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4);
        m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4);
    }
}

// vector form: 12 packed bytes (3 dwords) -> sc[0..3], sc[4..7], mn[0..3], mn[4..7] as byte lanes
// (same arithmetic as make_q4_scales, iqk_mul_mat.inc:134-143)
__device__ static inline void q4k_scales_bytes(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t &sc03,
                                               uint32_t &sc47, uint32_t &mn03, uint32_t &mn47) {
    sc03 = a0 & 0x3f3f3f3f;
    mn03 = a1 & 0x3f3f3f3f;
    sc47 = (a2 & 0x0f0f0f0f) | ((a0 >> 2) & 0x30303030);
    mn47 = ((a2 >> 4) & 0x0f0f0f0f) | ((a1 >> 2) & 0x30303030);
}

// ---- DPP lane exchanges (wave64 = 4 rows of 16 lanes): one VALU each instead of a ds_bpermute round trip
template <int CTRL>
__device__ static inline uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ static inline float dpp_f32(float v) {
    return __builtin_bit_cast(float, dpp_u32<CTRL>(__builtin_bit_cast(uint32_t, v)));
}
#define DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
#define DPP_ROW_SHL4 0x104   // lane i reads lane i + 4 of its row of 16

__device__ static inline float readlane_f32(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// maximum over the wave's 64 lanes, in every lane (uniform)
__device__ static inline float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<DPP_MIRROR>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}

__device__ static inline float wave_sum_xor(float v, int mask) {
    return v + __shfl_xor(v, mask, 64);
}

// mnpack geometry of tinyBLAS_Q0_AVX2 (tinyblas_cpu.h:794-931): is output (i, j) of an m x n
// problem computed by a PRECISE (Kahan) tile?  vregs32: AVX512 build; precise: FLAG_precise.
__host__ __device__ static inline bool q0_is_kahan(long i, long j, long m, long n, bool vregs32, bool precise) {
    long m0 = 0, n0 = 0;
    for (int depth = 0; depth < 64; ++depth) {
        long dm = m - m0, dn = n - n0;
        if (dm <= 0 || dn <= 0)
            return precise;
        long a, b, mc, nc;
        bool pr;
        if (vregs32) {
            a = dm < 3 ? dm : 3;
            b = dn < 3 ? dn : 3;
            if (a == 3 && b == 3) {
                mc = 3, nc = 3, pr = precise;
            } else if (a >= 2 && b >= 2) {
                mc = 2, nc = 2, pr = precise;
            } else if (a >= 2) {
                mc = 2, nc = 1, pr = true;
            } else if (b >= 2) {
                mc = 1, nc = 2, pr = true;
            } else {
                mc = 1, nc = 1, pr = true;
            }
        } else if (!precise) {
            a = dm < 3 ? dm : 3;
            b = dn < 2 ? dn : 2;
            if (a == 3 && b == 2) {
                mc = 3, nc = 2;
            } else if (a == 2 && b == 2) {
                mc = 2, nc = 2;
            } else if (a >= 2) {
                mc = 2, nc = 1;
            } else if (b == 2) {
                mc = 1, nc = 2;
            } else {
                mc = 1, nc = 1;
            }
            pr = false;
        } else {
            a = dm < 2 ? dm : 2;
            mc = a == 2 ? 2 : 1, nc = 1, pr = true;
        }
        long mp = m0 + dm / mc * mc;
        long np = n0 + dn / nc * nc;
        if (i < mp && j < np)
            return pr; // inside this level's tiled region
        if (j < np) {  // rows [mp, m) x cols [n0, np): first recursive call
            m0 = mp;
            n = np;
        } else {       // rows [m0, m) x cols [np, n): second recursive call
            n0 = np;
        }
    }
    return precise;
}
