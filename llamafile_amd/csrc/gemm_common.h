// gemm_common.h — pieces shared by the MFMA GEMM kernels (gemm_mfma.hip, gemm_wide.hip): exact-integer
// dequantisation of packed K-quant dwords into f16 MFMA fragments, and the XCD-aware tile order.
#pragma once
#include "lfamd_device.h"

#define XT_ROW_BYTES 512 // 256 f16 codes of one token for one super-block

__device__ static inline half2_t as_half2(uint32_t u) {
    return __builtin_bit_cast(half2_t, u);
}

__device__ static inline half2_t pk_fma(half2_t a, half2_t b, half2_t c) {
    return __builtin_elementwise_fma(a, b, c);
}

__device__ static inline half2_t bcast_h2(float v) {
    _Float16 h = (_Float16)v;
    half2_t r = {h, h};
    return r;
}

union frag_u {
    half8_t v;
    half2_t p[4];
    uint4 u;
};

// One K-step (8 nibbles of this lane) of a Q4_K-family dword -> f16x8 of sc*q.
// S = (sc,sc), O = (-1024 sc), S16 = sc/16, O16 = -64 sc.
// `magic` = 0x64006400 held in a VGPR: gfx9 VOP3 encodes one literal/SGPR only, so with both constants as
// literals hipcc splits (x & m) | magic into v_and + v_or; with the magic in a register it is one v_and_or_b32.
__device__ static inline half8_t dequant_q4(uint32_t x, half2_t S, half2_t O, half2_t S16, half2_t O16, uint32_t magic) {
    frag_u f;
    const uint32_t y = x >> 8;
    f.p[0] = pk_fma(as_half2((x & 0x000F000Fu) | magic), S, O);
    f.p[1] = pk_fma(as_half2((x & 0x00F000F0u) | magic), S16, O16);
    f.p[2] = pk_fma(as_half2((y & 0x000F000Fu) | magic), S, O);
    f.p[3] = pk_fma(as_half2((y & 0x00F000F0u) | magic), S16, O16);
    return f.v;
}

// Q5_K: the fifth bits of the K-step, already shifted onto their lattice (Hd = H >> dd, see q5hpos), join the
// nibbles before the same fused multiply-add: sc*q <= 63*31 = 1953 stays exact in f16.
__device__ static inline half8_t dequant_q5(uint32_t x, uint32_t Hd, half2_t S, half2_t O, half2_t S16, half2_t O16,
                                            uint32_t magic) {
    frag_u f;
    const uint32_t y = x >> 8;
    f.p[0] = pk_fma(as_half2((Hd & 0x00100010u) | ((x & 0x000F000Fu) | magic)), S, O);
    f.p[1] = pk_fma(as_half2((Hd & 0x01000100u) | ((x & 0x00F000F0u) | magic)), S16, O16);
    f.p[2] = pk_fma(as_half2(((Hd >> 8) & 0x00100010u) | ((y & 0x000F000Fu) | magic)), S, O);
    f.p[3] = pk_fma(as_half2(((Hd << 8) & 0x01000100u) | ((y & 0x00F000F0u) | magic)), S16, O16);
    return f.v;
}

// codes with an offset (Q3_K: q = code - 4): -(1024 + off) * sc is not an f16 number in general (1028 * 31 needs 13
// mantissa bits), so the offset is removed first (exact) and the scale applied by a second packed op.
__device__ static inline half8_t dequant_q4_off(uint32_t x, half2_t S, float off, uint32_t magic) {
    frag_u f;
    const uint32_t y = x >> 8;
    const half2_t o1 = bcast_h2(-(1024.0f + off)), o16 = bcast_h2(-(64.0f + off));
    const half2_t r16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    f.p[0] = (as_half2((x & 0x000F000Fu) | magic) + o1) * S;
    f.p[1] = pk_fma(as_half2((x & 0x00F000F0u) | magic), r16, o16) * S;
    f.p[2] = (as_half2((y & 0x000F000Fu) | magic) + o1) * S;
    f.p[3] = pk_fma(as_half2((y & 0x00F000F0u) | magic), r16, o16) * S;
    return f.v;
}

// Q4_0: the code minus 8, no integer scale (the f16 block scale is applied in f32 per 32-block).
__device__ static inline half8_t dequant_q40(uint32_t x, uint32_t magic) {
    frag_u f;
    const uint32_t y = x >> 8;
    const half2_t m1032 = {(_Float16)-1032.0f, (_Float16)-1032.0f};
    const half2_t m72 = {(_Float16)-72.0f, (_Float16)-72.0f};
    const half2_t r16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    f.p[0] = as_half2((x & 0x000F000Fu) | magic) + m1032;
    f.p[1] = pk_fma(as_half2((x & 0x00F000F0u) | magic), r16, m72);
    f.p[2] = as_half2((y & 0x000F000Fu) | magic) + m1032;
    f.p[3] = pk_fma(as_half2((y & 0x00F000F0u) | magic), r16, m72);
    return f.v;
}

// legacy 32-block types on a per-call image: code (4 or 5 bits, fifth bits Hd on the P5K lattice) minus `off`, no scale
template <bool H5>
__device__ static inline half8_t dequant_legacy(uint32_t x, uint32_t Hd, float off, uint32_t magic) {
    frag_u f;
    const uint32_t y = x >> 8;
    const half2_t o1 = bcast_h2(-(1024.0f + off)), o16 = bcast_h2(-(64.0f + off));
    const half2_t r16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    uint32_t c0 = (x & 0x000F000Fu) | magic, c1 = (x & 0x00F000F0u) | magic, c2 = (y & 0x000F000Fu) | magic,
             c3 = (y & 0x00F000F0u) | magic;
    if constexpr (H5) {
        c0 |= Hd & 0x00100010u;
        c1 |= Hd & 0x01000100u;
        c2 |= (Hd >> 8) & 0x00100010u;
        c3 |= (Hd << 8) & 0x01000100u;
    }
    f.p[0] = as_half2(c0) + o1;
    f.p[1] = pk_fma(as_half2(c1), r16, o16);
    f.p[2] = as_half2(c2) + o1;
    f.p[3] = pk_fma(as_half2(c3), r16, o16);
    return f.v;
}

// IQ4_XS byte image: two dwords = the K-step's eight values + 128; v_perm builds the f16 pairs 1024 + u directly
// (bytes [u0, 0x64, u1, 0x64]), then sc * value = (1024 + u) * sc - 1152 * sc (1152 * sc = 9 * sc * 2^7 is exact in f16;
// |sc * value| reaches 4064, so products above 2048 round like Q6_K's).
__device__ static inline half8_t dequant_bytes(uint32_t d0, uint32_t d1, half2_t S, half2_t O) {
    frag_u f;
    f.p[0] = pk_fma(as_half2(__builtin_amdgcn_perm(0x64646464u, d0, 0x04010400u)), S, O);
    f.p[1] = pk_fma(as_half2(__builtin_amdgcn_perm(0x64646464u, d0, 0x04030402u)), S, O);
    f.p[2] = pk_fma(as_half2(__builtin_amdgcn_perm(0x64646464u, d1, 0x04010400u)), S, O);
    f.p[3] = pk_fma(as_half2(__builtin_amdgcn_perm(0x64646464u, d1, 0x04030402u)), S, O);
    return f.v;
}

// The four dequantisation constants (S, O = -1024 S, S16 = S/16, O16 = -64 S) of TWO sub-blocks at once, lane-packed:
// v_perm puts the two scale bytes into f16 slots as 1024 + sc, one packed add removes the 1024, three packed multiplies
// give the rest: 5 VALU per sub-block pair instead of 12 (the K loop is VALU-issue bound: PMC shows VALU and MFMA
// cycles adding up rather than overlapping).  Users pick a half with a broadcast shuffle (folded into op_sel).
struct q4_consts2 {
    half2_t S, O, S16, O16;
};
__device__ static inline q4_consts2 q4_consts_pair(uint32_t word, int byte_lo) { // scales = bytes byte_lo, byte_lo + 1 of word
    q4_consts2 c;
    const uint32_t sel = byte_lo == 0 ? 0x04010400u : 0x04030402u; // [b, 0x64, b', 0x64]
    const half2_t m1024 = {(_Float16)-1024.0f, (_Float16)-1024.0f}, m64 = {(_Float16)-64.0f, (_Float16)-64.0f};
    const half2_t r16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    c.S = as_half2(__builtin_amdgcn_perm(0x64646464u, word, sel)) + m1024;
    c.O = c.S * m1024;
    c.S16 = c.S * r16;
    c.O16 = c.S * m64;
    return c;
}

// the same with the row's f16 super-block scale folded in (scaled-operand GEMM: S = d * sc rounded to f16; the three
// derived constants are exact multiples of it)
__device__ static inline q4_consts2 q4_consts_pair_scaled(uint32_t word, int byte_lo, half2_t d2) {
    q4_consts2 c;
    const uint32_t sel = byte_lo == 0 ? 0x04010400u : 0x04030402u;
    const half2_t m1024 = {(_Float16)-1024.0f, (_Float16)-1024.0f}, m64 = {(_Float16)-64.0f, (_Float16)-64.0f};
    const half2_t r16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    c.S = (as_half2(__builtin_amdgcn_perm(0x64646464u, word, sel)) + m1024) * d2;
    c.O = c.S * m1024;
    c.S16 = c.S * r16;
    c.O16 = c.S * m64;
    return c;
}

__device__ static inline uint32_t opaque_magic() {
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic)); // keep it a register value (see dequant_q4)
    return magic;
}

// Q6_K: codes are 6 bit (ql nibble | qh field), value sc*(code-32).  (code-32) is formed exactly,
// the product with the int8 scale is rounded to f16 (exact up to 2048; RNE to even above).
__device__ static inline half8_t dequant_q6(uint32_t x, uint32_t H, half2_t S) {
    frag_u f;
    const uint32_t y = x >> 8;
    const half2_t m1056 = {(_Float16)-1056.0f, (_Float16)-1056.0f};
    const half2_t m96 = {(_Float16)-96.0f, (_Float16)-96.0f};
    const half2_t r16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    half2_t c0 = as_half2((x & 0x000F000Fu) | (H & 0x00300030u) | 0x64006400u) + m1056;
    half2_t c1 = pk_fma(as_half2((x & 0x00F000F0u) | (H & 0x03000300u) | 0x64006400u), r16, m96);
    half2_t c2 = as_half2((y & 0x000F000Fu) | ((H >> 8) & 0x00300030u) | 0x64006400u) + m1056;
    half2_t c3 = pk_fma(as_half2((y & 0x00F000F0u) | ((H << 8) & 0x03000300u) | 0x64006400u), r16, m96);
    f.p[0] = c0 * S;
    f.p[1] = c1 * S;
    f.p[2] = c2 * S;
    f.p[3] = c3 * S;
    return f.v;
}

// Linear tile order -> (row-block, token tile).  Tiles are walked in SUPER-TILES of 8 row-blocks x 4 token tiles
// (ragged at the edges): an XCD's 32 resident work-groups then share 8 x 128 weight rows (8 x 72 B/row/256k) and
// 4 x 64 activation rows in its L2, instead of each XCD streaming every weight row (measured: 6x the algorithmic
// HBM bytes with the row-blocks-fastest order).  72a + 128b bytes per K element is minimal at a x b = 8 x 4.
__device__ static inline void tile_of(int L, int n_rb, int n_tt, int &rb, int &tt) {
    constexpr int SA = 8, SB = 4;
    const int grp = n_rb * SB;                    // tiles in one group of SB token tiles
    const int g = L / grp;
    const int idx = L - g * grp;
    const int w = min(SB, n_tt - g * SB);         // token tiles in this group (last group may be narrower)
    const int run = idx / (SA * w);
    const int rem = idx - run * SA * w;
    const int hgt = min(SA, n_rb - run * SA);     // row-blocks in this run (last run may be shorter)
    const int tl = rem / hgt;
    rb = run * SA + (rem - tl * hgt);
    tt = g * SB + tl;
}

