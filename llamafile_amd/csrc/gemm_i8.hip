// gemm_i8.hip — prefill GEMM for the resident Q4_K layout on the INT8 matrix cores: the reference's own arithmetic
// (mul_mat_qX_K_q8_K_T, iqk_mul_mat.inc:601-643: int4 x int8 -> int32 sub-block sums, 6-bit sub-block scales as integers,
// f32 super-block scales), exact integer dots instead of the <= 1e-3 of the scaled-operand f16 bodies.
//
// Why: the 128 x 64 tile (the one that fills 256 CUs on 4096 x 4096 x 512) is bound by what a CU can TAKE IN from L2
// (~75 GB/s per CU, tools/ingest_probe.hip), and 64 % of the f16 bodies' bytes are the activations, staged as f16.  Here the
// activations stay the Q8_K codes: one byte each (551 instead of 807 KB per work-group at k = 4096).
//
// `v_mfma_i32_32x32x32_i8` takes one 32-weight sub-block per instruction.  The 6-bit sub-block scale cannot ride in an int8
// operand with the nibble (63 * 15 = 945), so it is split sc = 8 a + b (a, b <= 7: a * q, b * q <= 105 fit): two MFMAs per
// sub-block into two int32 accumulators that live for a whole super-block, I = 8 * hi + lo — at twice the f16 rate that is the
// f16 body's MFMA time, and the multiply is ONE v_pk_mul_lo_u16 per four weights (byte products cannot carry into a
// neighbour).  The mins term is one f16 MFMA per super-block on the Q8_K block's own bsums (16-sums <= 2032: exact in f16)
// against the 6-bit mins.  Per super-block and output element: I -> f32, acc += d8 * (d * I - dmin * M): the f32 arithmetic
// of gemm_sb16i_kernel (gemm_sb.hip), <= 2e-6 of the oracle.
//
// Shape of the work: work-group = 128 weight rows x 64 tokens, 8 waves.  Waves 0..3 COMPUTE, one per SIMD, a 32-row x 64-token
// tile each over the whole K: a dequantised fragment pair feeds four MFMAs, and nothing is dequantised twice (two computing
// waves per SIMD on 32 x 32 tiles were tried first: their VALU streams — 240 instructions per wave and super-block, half of them
// the same dequantisation — do not overlap, 2340 cycles per super-block for 1088 of MFMA; tools/valu_rate_probe.hip: two waves
// of a SIMD issue VOP3 instructions at 3.3 cycles each together, one alone at 4.6).  Waves 4..7 only LOAD: everything arrives by
// LDS-DMA (weights 18 KB + codes 16 KB + bsums 2 KB + d8 per super-block, ten 1 KiB pieces per loader and stage — an LDS-DMA
// instruction costs its issuer 60-180 cycles, which a computing wave does not have), three stages ahead in a ring of four; ONE
// barrier per super-block, placed so that the NEXT stage has landed before the current one is computed — its first operands
// are read before the barrier they would otherwise wait behind.
#include "gemm_wide_impl.h"

#include <stdlib.h>

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define I8_COLS 64
#define I8_WQS 0         // 4 row tiles x 4096 B of packed nibbles
#define I8_WHD 16384     // 4 x 512 B of row headers {d, dmin, scales[12]}
#define I8_X 18432       // 64 tokens x 256 codes, 16-byte chunks XOR-swizzled by token
#define I8_S 34816       // 64 tokens x 16 f16 bsums
#define I8_D 36864       // 64 f32 d8 (+ 768 B the piece's other lanes write)
#define I8_SLOT 37888
#define I8_STAGES 4

#if GEMM_DIAG == 7 // development: s_memtime stamps of one work-group's waves 0 and 4 (tools/i8_stamps.py)
#define I8_STAMP_WG 100
__device__ unsigned long long g_i8_stamps[2 * 256];
extern "C" int lfamd_debug_i8_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_i8_stamps), sizeof(g_i8_stamps));
}
#define ISTAMP()                                                                                                     \
    do {                                                                                                             \
        if (blockIdx.x == I8_STAMP_WG && lane == 0 && rw == 0 && stamp_n < 256)                                      \
            g_i8_stamps[tw * 256 + stamp_n++] = __builtin_amdgcn_s_memtime();                                        \
    } while (0)
#else
#define ISTAMP()
#endif

// LDS-DMA piece: 64 lanes x 16 B from base + voff to LDS at lds_dst + 16 * lane.  SGPR base kept by scalar adds; M0 written and
// not restored (nothing else in this kernel reads it).  No offset field: it would move the LDS address too.
__device__ static inline void i8_dma16(const void *base, uint32_t lds_dst, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// LDS reads hipcc neither counts nor moves (left alone it fetches a fragment right in front of its MFMA and waits lgkmcnt(0))
template <int IMM>
__device__ static inline void i8_dsr(u32x4 &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

__device__ static inline uint32_t pk_mul_u16(uint32_t bytes, uint32_t pair) { // four byte products (<= 105 each: no carries)
    const u16x2_t r = __builtin_bit_cast(u16x2_t, bytes) * __builtin_bit_cast(u16x2_t, pair);
    return __builtin_bit_cast(uint32_t, r);
}

__global__ __launch_bounds__(512) void gemm_i8_kernel(const gemm_mats mats, int nb, const int8_t *__restrict__ Xq,
                                                      const float *__restrict__ d8T, const _Float16 *__restrict__ Xs, long n,
                                                      long n_pad, int n_rb, int n_ct) {
#ifdef I8_CHECK_NB // tools/isa_hazards.py: a fixed trip count
    nb = I8_CHECK_NB;
#endif
    __shared__ __attribute__((aligned(16))) uint8_t lds[I8_STAGES * I8_SLOT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;

    // ---- tile of this work-group (the order of gemm_ks: XCD-aware super-tiles)
    int rb, ct;
    {
        const int n_wg = n_rb * n_ct;
        const int id = (int)blockIdx.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = id & 7;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
        tile_of(L, n_rb, n_ct, rb, ct);
    }
    int mj = 0;
#pragma unroll
    for (int jj = 1; jj < GEMM_MAX_MATS; jj++)
        if (jj < mats.count && rb >= mats.rb_end[jj - 1])
            mj = jj;
    if (mj > 0)
        rb -= mats.rb_end[mj - 1];
    const uint8_t *__restrict__ A = mats.A[mj];
    float *__restrict__ C = mats.C[mj];
    const long m = mats.m[mj], ldc = mats.ldc[mj];
    const long n0 = (long)ct * I8_COLS;
    const int n_row_tiles = (int)((m + 31) / 32);
    const uint32_t lds0 = lds_addr(lds);

    if (wave >= 4) {
        // ================= loader waves: ten 1 KiB pieces per stage each, three stages ahead, counted waits =================
        const int lw = wave - 4;
        auto tile_off = [&](int r) { // byte offset of row tile 4 rb + r's first super-block (tiles past the matrix: its last one)
            const int t = rb * 4 + r;
            return (uint32_t)(t < n_row_tiles ? t : n_row_tiles - 1) * (uint32_t)nb * (uint32_t)P4K_TILE;
        };
        // 0..3: the four nibble groups of row tile lw (4 KiB in a row at both ends)
        const uint32_t voffW = tile_off(lw) + (uint32_t)(lane * 16);
        const uint32_t dstW = (uint32_t)(I8_WQS + lw * 4096);
        // 4..7: code pieces 4 lw .. 4 lw + 3: tokens 4 p .. 4 p + 3, lane = (token 4 p + (lane >> 4), slot lane & 15) <- chunk slot ^ (token & 15)
        uint32_t voffX[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int tk = 4 * (4 * lw + e) + (lane >> 4);
            voffX[e] = (uint32_t)(tk * 256 + (((lane & 15) ^ (tk & 15)) << 4));
        }
        const uint32_t dstX = (uint32_t)(I8_X + 4 * lw * 1024);
        // 8: loaders 0, 1 the headers of row tiles 2 lw, 2 lw + 1; loaders 2, 3 the bsums.  9: the d8 row (the same 256 bytes from all
        // four, lanes past 15 repeat them into the padding behind): ten pieces per loader and stage, one counted wait for all
        const uint32_t voff8 = lw < 2 ? tile_off(2 * lw + h) + (uint32_t)(P4K_HDR + i * 16) : (uint32_t)((lw & 1) * 1024 + lane * 16);
        const uint32_t dst8 = lw < 2 ? (uint32_t)(I8_WHD + lw * 1024) : (uint32_t)(I8_S + (lw & 1) * 1024);
        const uint32_t voffD = (uint32_t)((lane & 15) * 16);
        const size_t xstride = (size_t)n_pad * 256, dstride = (size_t)n_pad * 4;
        const size_t stride8 = lw < 2 ? (size_t)P4K_TILE : (size_t)n_pad * 32;
        const uint8_t *wt_n = uniform_ptr(A);
        const uint8_t *xq_n = uniform_ptr((const uint8_t *)Xq + (size_t)n0 * 256);
        const uint8_t *p8_n = uniform_ptr(lw < 2 ? A : (const uint8_t *)Xs + (size_t)n0 * 32);
        const uint8_t *d8_n = uniform_ptr((const uint8_t *)d8T + (size_t)n0 * 4);
        // five wait states between the VALU writes of those SGPRs (v_readfirstlane) and the first vector-memory instruction that
        // reads them: hipcc pads such hazards itself, but not in front of an asm statement (tools/isa_hazards.py checks the ISA)
        asm volatile("s_nop 4" ::"s"(wt_n), "s"(xq_n), "s"(p8_n), "s"(d8_n));
        auto dma_stage = [&](uint32_t slot_base) {
            i8_dma16(wt_n, slot_base + dstW, voffW);
            i8_dma16(wt_n, slot_base + dstW + 1024, voffW + 1024);
            i8_dma16(wt_n, slot_base + dstW + 2048, voffW + 2048);
            i8_dma16(wt_n, slot_base + dstW + 3072, voffW + 3072);
#pragma unroll
            for (int e = 0; e < 4; e++)
                i8_dma16(xq_n, slot_base + dstX + (uint32_t)(e * 1024), voffX[e]);
            i8_dma16(p8_n, slot_base + dst8, voff8);
            i8_dma16(d8_n, slot_base + (uint32_t)I8_D, voffD);
            wt_n += P4K_TILE, xq_n += xstride, p8_n += stride8, d8_n += dstride; // (past the last super-block nothing more is issued)
        };
#pragma unroll
        for (int st = 0; st < 3; st++)
            if (st < nb) // (uniform)
                dma_stage(lds0 + (uint32_t)(st * I8_SLOT));
        if (nb >= 3)
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (nb == 2)
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory"); // stage 0 has landed for everybody
        for (int b = 0; b < nb; b++) {
            // this wave's pieces of stage b + 1 (all but the ten of stage b + 2 that are younger); behind the barrier everybody's
            // have landed and everybody is done with stage b - 1, whose slot takes stage b + 3
            if (b + 2 < nb)
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (b + 3 < nb)
                dma_stage(lds0 + (uint32_t)(((b + 3) & 3) * I8_SLOT));
        }
        return;
    }

    // ================= compute waves: row tile rw, both token tiles =================
    const int rw = wave;
    const int rt = rb * 4 + rw;
#if GEMM_DIAG == 7
    const int tw = 0;
    int stamp_n = 0;
#endif
    ISTAMP();
    // ---- this lane's LDS read addresses in slot 0 (token tile 1: + 8192 in the code image, + 1024 in the bsums, + 128 in d8)
    const uint32_t adQ = lds0 + (uint32_t)(I8_WQS + rw * 4096 + lane * 16), adH = lds0 + (uint32_t)(I8_WHD + rw * 512 + i * 16);
    uint32_t adX[8];
#pragma unroll
    for (int jb = 0; jb < 8; jb++)
        adX[jb] = lds0 + (uint32_t)(I8_X + i * 256 + (((2 * jb + h) ^ (i & 15)) << 4));
    const uint32_t adS = lds0 + (uint32_t)(I8_S + i * 32 + h * 16), adD = lds0 + (uint32_t)(I8_D + 4 * h * 4);

    float acc[2][16];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[t][r] = 0.0f;
    const v16i_t zero16i = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float16_t_ zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    asm volatile("s_barrier" ::: "memory"); // stage 0 has landed for everybody
    ISTAMP();

    // Registers of the LDS pipeline (all reads are asm, every wait is counted: LDS reads of a wave return in issue order):
    //   HD the row header, Q[2] the nibble groups (one serves two sub-blocks), XF[4][2] the code fragments of both token tiles two
    //   sub-blocks ahead, SF[2] the bsums fragments, DF[4] a token tile's d8.  A stage ENDS by fetching HD, Q[0], XF[0], XF[1] of
    //   the next one (landed: the barrier at the top of THIS stage said so) and waiting for them, so nothing in flight crosses the
    //   loop's back edge (where hipcc may copy registers).
    u32x4 HD, Q[2], XF[4][2], SF[2], DF[8];
    {
        i8_dsr<0>(HD, adH);
        i8_dsr<0>(Q[0], adQ);
        i8_dsr<0>(XF[0][0], adX[0]);
        i8_dsr<8192>(XF[0][1], adX[0]);
        i8_dsr<0>(XF[1][0], adX[1]);
        i8_dsr<8192>(XF[1][1], adX[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(HD), "+v"(Q[0]), "+v"(XF[0][0]), "+v"(XF[0][1]), "+v"(XF[1][0]), "+v"(XF[1][1]));
    }

    for (int b = 0; b < nb; b++) {
        const uint32_t so = (uint32_t)((b & 3) * I8_SLOT), so_n = (uint32_t)(((b + 1) & 3) * I8_SLOT);
        ISTAMP();
        asm volatile("s_barrier" ::: "memory"); // stage b + 1 has landed for everybody (the loaders waited for their pieces)
        ISTAMP();

        // row header: d, dmin, the eight 6-bit scales as a = sc >> 3, b = sc & 7 byte lanes, the eight mins
        uint32_t sc03, sc47, mn03, mn47;
        q4k_scales_bytes(HD.y, HD.z, HD.w, sc03, sc47, mn03, mn47);
        const uint32_t a03 = (sc03 >> 3) & 0x07070707u, a47 = (sc47 >> 3) & 0x07070707u;
        const uint32_t b03 = sc03 & 0x07070707u, b47 = sc47 & 0x07070707u;
        float d = h2f((uint16_t)(HD.x & 0xffff)), dmin = h2f((uint16_t)(HD.x >> 16));
        uint32_t mw = h ? mn47 : mn03;
        // everything the stage needs from HD is in registers of its own from here on (left alone hipcc reads d and dmin as f16
        // halves of HD.x in the epilogue, keeps HD alive, and copies the NEXT header over it — before the wait for that read)
        asm volatile("" : "+v"(d), "+v"(dmin), "+v"(mw));

        v16i_t hi[2] = {zero16i, zero16i}, lo[2] = {zero16i, zero16i};
#pragma unroll
        for (int jb = 0; jb < 8; jb++) {
            // look-ahead: the fragments of sub-block jb + 2; at even jb the nibble group of sub-blocks jb + 2, jb + 3
            if (jb + 2 < 8) {
                i8_dsr<0>(XF[(jb + 2) & 3][0], adX[jb + 2] + so);
                i8_dsr<8192>(XF[(jb + 2) & 3][1], adX[jb + 2] + so);
                if ((jb & 1) == 0)
                    i8_dsr<0>(Q[((jb >> 1) + 1) & 1], adQ + so + (uint32_t)(((jb >> 1) + 1) * 1024));
            }
            if (jb == 4) { // the bsums fragments (their MFMAs go right behind the K loop's last)
                i8_dsr<0>(SF[0], adS + so);
                i8_dsr<1024>(SF[1], adS + so);
            } else if (jb == 6) { // both token tiles' d8, early enough to have landed when the epilogue starts
                i8_dsr<0>(DF[0], adD + so);
                i8_dsr<32>(DF[1], adD + so);
                i8_dsr<64>(DF[2], adD + so);
                i8_dsr<96>(DF[3], adD + so);
                i8_dsr<128>(DF[4], adD + so);
                i8_dsr<160>(DF[5], adD + so);
                i8_dsr<192>(DF[6], adD + so);
                i8_dsr<224>(DF[7], adD + so);
            }
            // counted wait for XF[jb & 3] (and, at even jb, Q[(jb >> 1) & 1]): what was issued after them may stay in flight
            u32x4 &xa = XF[jb & 3][0], &xb = XF[jb & 3][1];
            u32x4 &qg = Q[(jb >> 1) & 1];
            if (jb == 0)
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xa), "+v"(xb), "+v"(qg));
            else if (jb == 1)
                asm volatile("" : "+v"(xa), "+v"(xb)); // (older than what the wait at jb = 0 left in flight)
            else if (jb < 4)
                asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(xa), "+v"(xb), "+v"(qg));
            else if (jb < 6)
                asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(xa), "+v"(xb), "+v"(qg));
            else if (jb == 6)
                asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(xa), "+v"(xb), "+v"(qg));
            else
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xa), "+v"(xb), "+v"(SF[0]), "+v"(SF[1]));
            const int e = jb & 1;
            const uint32_t x0 = e ? qg.z : qg.x, x1 = e ? qg.w : qg.y;
            const uint32_t r0 = x0 & 0x0F0F0F0Fu, r1 = (x0 >> 4) & 0x0F0F0F0Fu, r2 = x1 & 0x0F0F0F0Fu, r3 = (x1 >> 4) & 0x0F0F0F0Fu;
            // the scales of sub-blocks jb, jb ^ 1 as the two 16-bit halves of one register ([a_even, 0, a_odd, 0]); the multiply takes
            // its half for both products through op_sel (a perm per sub-block PAIR and factor instead of one per sub-block)
            const uint32_t sel2 = 0x0c000c00u | (uint32_t)(jb & 2) | ((uint32_t)((jb & 2) + 1) << 16);
            const u16x2_t pa2 = __builtin_bit_cast(u16x2_t, __builtin_amdgcn_perm(0u, jb < 4 ? a03 : a47, sel2));
            const u16x2_t pb2 = __builtin_bit_cast(u16x2_t, __builtin_amdgcn_perm(0u, jb < 4 ? b03 : b47, sel2));
            const uint32_t pa = __builtin_bit_cast(uint32_t, (jb & 1) ? __builtin_shufflevector(pa2, pa2, 1, 1) : __builtin_shufflevector(pa2, pa2, 0, 0));
            const uint32_t pb = __builtin_bit_cast(uint32_t, (jb & 1) ? __builtin_shufflevector(pb2, pb2, 1, 1) : __builtin_shufflevector(pb2, pb2, 0, 0));
            const v4i_t whi = {(int)pk_mul_u16(r0, pa), (int)pk_mul_u16(r1, pa), (int)pk_mul_u16(r2, pa), (int)pk_mul_u16(r3, pa)};
            const v4i_t wlo = {(int)pk_mul_u16(r0, pb), (int)pk_mul_u16(r1, pb), (int)pk_mul_u16(r2, pb), (int)pk_mul_u16(r3, pb)};
            const v4i_t av0 = __builtin_bit_cast(v4i_t, xa), av1 = __builtin_bit_cast(v4i_t, xb);
            hi[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av0, whi, hi[0], 0, 0, 0);
            hi[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av1, whi, hi[1], 0, 0, 0);
            lo[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av0, wlo, lo[0], 0, 0, 0);
            lo[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av1, wlo, lo[1], 0, 0, 0);
        }
        // mins: M = sum_j m_j * bsum_j as one f16 MFMA per token tile over the block's sixteen 16-sums; operand k = 8 h + e <-> m_{4 h + e / 2}
        frag_u wm;
        {
            const half2_t m1024 = {(_Float16)-1024.0f, (_Float16)-1024.0f};
#pragma unroll
            for (int p = 0; p < 4; p++)
                wm.p[p] = as_half2(__builtin_amdgcn_perm(0x64646464u, mw, 0x04000400u | (uint32_t)p | ((uint32_t)p << 16))) + m1024;
        }
        const float16_t_ tm0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, SF[0]), wm.v, zero16, 0, 0, 0);
        const float16_t_ tm1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, SF[1]), wm.v, zero16, 0, 0, 0);
        // acc += d8 * (d * I - dmin * M)  (register r = token 8 (r >> 2) + 4 h + (r & 3) of the token tile)
        auto epilogue = [&](int t, const float16_t_ &tm) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                // (whole-vector cast: __builtin_bit_cast(float, DF[g].y) on a vector ELEMENT reads element 0 with this hipcc)
                const float4_t_ d8f = __builtin_bit_cast(float4_t_, DF[4 * t + g]);
                const float d8v[4] = {d8f.x, d8f.y, d8f.z, d8f.w};
#ifndef I8_SCALAR_EPILOGUE // two outputs per v_pk_mul_f32 / v_pk_fma_f32 (the same roundings: every operation is the scalar one, twice)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int r = 4 * g + e;
                    const f32x2_t If = {(float)((hi[t][r] << 3) + lo[t][r]), (float)((hi[t][r + 1] << 3) + lo[t][r + 1])};
                    const f32x2_t tm2 = {tm[r], tm[r + 1]}, d2 = {d, d}, nd2 = {-dmin, -dmin}, d82 = {d8v[e], d8v[e + 1]};
                    f32x2_t a2 = {acc[t][r], acc[t][r + 1]};
                    const f32x2_t u = __builtin_elementwise_fma(nd2, tm2, d2 * If);
                    a2 = __builtin_elementwise_fma(u, d82, a2);
                    acc[t][r] = a2.x, acc[t][r + 1] = a2.y;
                }
#else
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int r = 4 * g + e;
                    const float u = fmaf(-dmin, tm[r], d * (float)((hi[t][r] << 3) + lo[t][r]));
                    acc[t][r] = fmaf(u, d8v[e], acc[t][r]);
                }
#endif
            }
        };
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(DF[0]), "+v"(DF[1]), "+v"(DF[2]), "+v"(DF[3]));
        // the next stage's first operands (this stage's once more at the very end: never used)
        i8_dsr<0>(HD, adH + so_n);
        i8_dsr<0>(Q[0], adQ + so_n);
        i8_dsr<0>(XF[0][0], adX[0] + so_n);
        i8_dsr<8192>(XF[0][1], adX[0] + so_n);
        i8_dsr<0>(XF[1][0], adX[1] + so_n);
        i8_dsr<8192>(XF[1][1], adX[1] + so_n);
        epilogue(0, tm0);
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(DF[4]), "+v"(DF[5]), "+v"(DF[6]), "+v"(DF[7]));
        epilogue(1, tm1);
        // nothing in flight crosses the back edge
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(HD), "+v"(Q[0]), "+v"(XF[0][0]), "+v"(XF[0][1]), "+v"(XF[1][0]), "+v"(XF[1][1]));
    }
    ISTAMP();

    // ---- store: lane (i, h) holds weight row 32 rt + i, register r = token n0 + 32 t + 8 (r >> 2) + 4 h + (r & 3)
    if (rt < n_row_tiles) {
        const long row = (long)rt * 32 + i;
        if (row < m) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const long tk = n0 + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
                    if (tk < n)
                        C[tk * ldc + row] = acc[t][r];
                }
        }
    }
#if GEMM_DIAG == 7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ISTAMP();
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Activation staging for the int8 body: the quantize_row_q8_K arithmetic (f32 rows) or the caller's Q8_K blocks, written as
//   Xq [nb][n_pad][256] int8 — sub-block jb, K half h -> 16 bytes: K-step 2 jb: elements (0,4,1,5,2,6,3,7) of k = 32 jb + 8 h + j,
//                              then K-step 2 jb + 1 the same (the byte order of the unpacked P4K nibbles)
//   d8T [nb][n_pad] f32, Xs [nb][n_pad][16] f16 = the block's bsums (sums of 16 codes: |S| <= 2032, exact)
// Half a wave per (token, super-block): lane l holds the EIGHT values 8 (l & 31) + e of block (l >> 5) — one whole 8-code group —
// so the fixed part of the quantiser (DPP maximum, the two IEEE divisions) is paid once per two blocks, and the group's two
// dwords (j0, j4, j1, j5), (j2, j6, j3, j7) are built in the lane (the one-wave-per-block form needed ~220 instructions per block
// and ran 6.2 us at 512 x 4096; this one ~75 per block).  Eight waves per work-group = sixteen super-blocks of one token;
// tokens n .. n_pad are zero.
template <bool F32IN>
__global__ __launch_bounds__(512) void prep_i8_kernel(const uint8_t *__restrict__ X, size_t x_row_bytes, long n, long n_pad, int nb,
                                                       int8_t *__restrict__ Xq, float *__restrict__ d8T, _Float16 *__restrict__ Xs,
                                                       const int32_t *__restrict__ src_idx) {
    // grid = (token, group of sixteen super-blocks): no division in the index arithmetic
    const long tok = (long)blockIdx.x;
    const int lane = threadIdx.x & 63, grp = lane & 31;
    const bool hi = lane >= 32;
    const int b = (int)blockIdx.y * 16 + 2 * (int)(threadIdx.x >> 6) + (hi ? 1 : 0);
    const bool live = b < nb; // (a block past the row: its half-wave computes on zeros and stores nothing)
    const long src = src_idx ? (long)src_idx[tok] : (tok < n ? tok : -1);
    const size_t o = (size_t)(live ? b : 0) * n_pad + tok;
    uint32_t y0 = 0, y1 = 0;
    float d = 0.0f;
    int S = 0; // sum of this lane's eight codes
    if constexpr (F32IN) {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (src >= 0 && live) {
            const float4 *p = (const float4 *)((const float *)(X + src * x_row_bytes) + (size_t)b * 256 + 8 * grp);
            const float4 va = p[0], vb = p[1];
            v[0] = va.x, v[1] = va.y, v[2] = va.z, v[3] = va.w, v[4] = vb.x, v[5] = vb.y, v[6] = vb.z, v[7] = vb.w;
        }
        // quantize_row_q8_K: the FIRST element of largest magnitude gives the sign of iscale = -128 / max; nearest-even codes
        // clamped at 127; d = 1 / iscale  (the arithmetic of gemv_impl.h: stage_f32_q8k_wave2)
        float a[8];
#pragma unroll
        for (int e = 0; e < 8; e++)
            a[e] = fabsf(v[e]);
        float am = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7])));
        am = fmaxf(am, dpp_f32<DPP_XOR1>(am));
        am = fmaxf(am, dpp_f32<DPP_XOR2>(am));
        am = fmaxf(am, dpp_f32<DPP_HALF_MIRROR>(am));
        am = fmaxf(am, dpp_f32<DPP_MIRROR>(am));
        const float amax_lo = fmaxf(readlane_f32(am, 0), readlane_f32(am, 16)), amax_hi = fmaxf(readlane_f32(am, 32), readlane_f32(am, 48));
        const float amax = hi ? amax_hi : amax_lo;
        bool any = false;
        float cand = v[7];
#pragma unroll
        for (int e = 7; e >= 0; e--) { // the FIRST element that reaches the maximum wins
            const bool me = a[e] == amax;
            cand = me ? v[e] : cand;
            any = any || me;
        }
        const unsigned long long ball = __builtin_amdgcn_ballot_w64(any);
        const uint32_t blo = (uint32_t)ball, bhi = (uint32_t)(ball >> 32);
        const int first_lo = blo ? __builtin_ctz(blo) : 0, first_hi = 32 + (bhi ? __builtin_ctz(bhi) : 0);
        const float val_lo = readlane_f32(cand, first_lo), val_hi = readlane_f32(cand, first_hi);
        const bool nz = amax != 0.0f;
        const float val = nz ? (hi ? val_hi : val_lo) : 1.0f;
        const float iscale = -128.0f / val;
        // bytes (j0, j4, j1, j5) | (j2, j6, j3, j7)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            int q = (int)rintf(iscale * v[e]);
            q = q > 127 ? 127 : q;
            q = nz ? q : 0;
            S += q;
            const int pos = (e >> 2) + 2 * (e & 1); // byte of element e inside its dword: e = 0, 4 -> 0, 1; 1, 5 -> 2, 3; ...
            if (((e >> 1) & 1) == 0)
                y0 |= (uint32_t)(q & 0xff) << (8 * pos);
            else
                y1 |= (uint32_t)(q & 0xff) << (8 * pos);
        }
        d = nz ? 1.0f / iscale : 0.0f;
    } else {
        if (src >= 0 && live) {
            const lfamd_block_q8_K *y = (const lfamd_block_q8_K *)(X + src * x_row_bytes) + b;
            const uint32_t w0 = *(const uint32_t *)((const uint8_t *)y->qs + 8 * grp), w1 = *(const uint32_t *)((const uint8_t *)y->qs + 8 * grp + 4);
            y0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u); // (j0, j4, j1, j5)
            y1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u); // (j2, j6, j3, j7)
#pragma unroll
            for (int e = 0; e < 4; e++)
                S += (int)(int8_t)(w0 >> (8 * e)) + (int)(int8_t)(w1 >> (8 * e));
            d = y->d;
        }
    }
    // group grp = codes 8 grp .. 8 grp + 7: sub-block grp >> 2, K-step (grp >> 1) & 1, K half grp & 1
    S += (int)dpp_u32<DPP_XOR1>((uint32_t)S); // bsums[grp / 2]: codes 16 (grp / 2) .. + 15
    if (live) {
        *(uint2 *)(Xq + o * 256 + (grp >> 2) * 32 + (grp & 1) * 16 + ((grp >> 1) & 1) * 8) = make_uint2(y0, y1);
        if ((grp & 1) == 0)
            Xs[o * 16 + (grp >> 1)] = (_Float16)(float)S;
        if (grp == 0)
            d8T[o] = d;
    }
}

// Which launches take the int8 body (LFAMD_GEMM_NO_I8: the f16 bodies instead — A/B runs).  Q4_K, at most TWO rounds of its 128 x 64
// tiles (<= 256 tiles of 128 x 128) and at least half a round (smaller grids keep the K-split launches of gemm_lw.hip).  Measured,
// prep + GEMM per call at 512 tokens against the scaled-operand f16 bodies (profiles/r04_i8_grid_limit.txt): 4096 x 4096 30.5 us
// (32.9), 4096 x 14336 83 (88), 6144 x 4096 47.5 (56.8), 8192 x 4096 56.6 (61.9), 8192 x 8192 92.8 (106.2), 4096 x 4096 x 1024 51.3
// (60.4), 6144 x 4096 x 300 29.0 (54.9) — faster AND exact; beyond two rounds the 256 x 128 f16 tile wins (9216 x 4096: 70.9 against
// 65.0, 14336 x 4096: 97 against 75-80) and stays.  row_blocks128: of all the matrices of the launch together.
extern "C" int lfamd_gemm_i8_ok(int Atype, long row_blocks128, long n) {
    static const bool off = getenv("LFAMD_GEMM_NO_I8") != nullptr;
    if (off || Atype != LFAMD_TYPE_Q4_K || n < 1)
        return 0;
    static const long max128 = getenv("LFAMD_I8_MAX_TILES128") ? atol(getenv("LFAMD_I8_MAX_TILES128")) : 256; // A/B runs
    // (below half a round of its tiles the K-split launches of gemm_lw.hip are as fast or faster: 1024 x 4096 x 512 22.8 against 24.1 us,
    // 4096 x 14336 x 128 40.1 against 63.0)
    return row_blocks128 * ((n + I8_COLS - 1) / I8_COLS) >= 128 && row_blocks128 * ((n + 127) / 128) <= max128;
}

extern "C" size_t lfamd_gemm_i8_workspace(long k, long n) { // Xq, d8T, Xs
    const size_t n_pad = ((size_t)n + 127) / 128 * 128, nb = (size_t)(k / 256);
    return n_pad * nb * 256 + n_pad * nb * 4 + n_pad * nb * 32;
}

// The staged activation image the body reads (what prep_i8_kernel writes, and what the fused producers of norm_quant.hip write
// directly — lfamd_rms_norm_quantize / lfamd_swiglu_quantize with LFAMD_TYPE_STAGED_Q8K): Xq [nb][n_pad][256], d8T [nb][n_pad] f32,
// Xs [nb][n_pad][16] f16, n_pad = n rounded up to 128, lfamd_gemm_i8_workspace(k, n) bytes in all.
static hipError_t gemm_i8_go(int count, const void *const *A, const long *m, long k, const void *image, long n, float *const *C, const long *ldc,
                             hipStream_t s) {
    const int nb = (int)(k / 256);
    const long n_pad = (n + 127) / 128 * 128;
    const int8_t *Xq = (const int8_t *)image;
    const float *d8T = (const float *)((const uint8_t *)image + (size_t)n_pad * nb * 256);
    const _Float16 *Xs = (const _Float16 *)((const uint8_t *)d8T + (size_t)n_pad * nb * 4);
    gemm_mats mats;
    int n_rb = 0;
    mats.count = 0;
    mats.moe_cnt = mats.moe_poff = mats.moe_slot_row = nullptr, mats.expert_bytes = 0, mats.moe_ct_max = 0;
    for (int j = 0; j < count; j++) {
        if (m[j] <= 0)
            continue;
        const int q = mats.count++;
        mats.A[q] = (const uint8_t *)A[j], mats.C[q] = C[j], mats.m[q] = m[j], mats.ldc[q] = ldc[j];
        n_rb += (int)((m[j] + 127) / 128);
        mats.rb_end[q] = n_rb;
    }
    if (mats.count == 0)
        return hipGetLastError();
    for (int q = mats.count; q < GEMM_MAX_MATS; q++)
        mats.A[q] = mats.A[0], mats.C[q] = mats.C[0], mats.m[q] = 0, mats.ldc[q] = 0, mats.rb_end[q] = n_rb;
    const int n_ct = (int)((n + I8_COLS - 1) / I8_COLS);
    gemm_i8_kernel<<<(unsigned)(n_rb * n_ct), 512, 0, s>>>(mats, nb, Xq, d8T, Xs, n, n_pad, n_rb, n_ct);
    return hipGetLastError();
}

// B: f32 rows or Q8_K blocks; ws: lfamd_gemm_i8_workspace(k, n) bytes.
extern "C" hipError_t lfamd_launch_gemm_i8(int count, const void *const *A, const long *m, long k, int Btype, const void *B, size_t b_row_bytes,
                                           long n, float *const *C, const long *ldc, void *ws, const int32_t *src_idx, hipStream_t s) {
    if (n <= 0 || count <= 0)
        return hipSuccess;
    if (count > GEMM_MAX_MATS || k % 256 || (Btype != LFAMD_TYPE_F32 && Btype != LFAMD_TYPE_Q8_K))
        return hipErrorInvalidValue;
    const int nb = (int)(k / 256);
    const long n_pad = (n + 127) / 128 * 128;
    int8_t *Xq = (int8_t *)ws;
    float *d8T = (float *)((uint8_t *)ws + (size_t)n_pad * nb * 256);
    _Float16 *Xs = (_Float16 *)((uint8_t *)d8T + (size_t)n_pad * nb * 4);
    const dim3 pg((unsigned)n_pad, (unsigned)((nb + 15) / 16));
    if (Btype == LFAMD_TYPE_F32)
        prep_i8_kernel<true><<<pg, 512, 0, s>>>((const uint8_t *)B, b_row_bytes, n, n_pad, nb, Xq, d8T, Xs, src_idx);
    else
        prep_i8_kernel<false><<<pg, 512, 0, s>>>((const uint8_t *)B, b_row_bytes, n, n_pad, nb, Xq, d8T, Xs, src_idx);
    return gemm_i8_go(count, A, m, k, ws, n, C, ldc, s);
}

// The activations arrive as the staged image already (a fused producer wrote it): the GEMM alone, no staging launch.
extern "C" hipError_t lfamd_launch_gemm_i8_staged(int count, const void *const *A, const long *m, long k, const void *image, long n,
                                                  float *const *C, const long *ldc, hipStream_t s) {
    if (n <= 0 || count <= 0)
        return hipSuccess;
    if (count > GEMM_MAX_MATS || k % 256 || !image || ((uintptr_t)image & 15))
        return hipErrorInvalidValue;
    return gemm_i8_go(count, A, m, k, image, n, C, ldc, s);
}
