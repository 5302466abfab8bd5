// gemm_mfma.hip — prefill GEMM: packed K-quant weights dequantised per wave straight into MFMA
// fragments, activations as exact-integer f16 tiles in LDS, v_mfma_f32_32x32x16_f16.
//
// Replaces the reference's mul_mat_q (ggml-cuda.cu.patch:13829-14130; dp4a, 32-wide warps, no MFMA
// on gfx950 — SURVEY.md F5) and follows the CPU path's arithmetic, mul_mat_qX_K_q8_K_T
// (iqk_mul_mat.inc:601-643):
//     C[n][i] = sum_b d8[n][b] * ( d[i][b] * sum_j sc[i][b][j] * <q[i][b][j], q8[n][b][j]>
//                                  - dmin[i][b] * sum_j mn[i][b][j] * bsum[n][b][j] )
// The integer parts are computed EXACTLY on the matrix cores: the weight operand is sc*q (<= 63*15,
// exact in f16), the activation operand is the int8 code (exact in f16), products and their sums over
// one 256-wide super-block stay below 2^24 in the f32 accumulator.  The two f32 scales are applied
// once per super-block on the VALU.  The mins term is one extra MFMA per super-block with the pair
// sums of bsums split into two exact f16 parts (pack.hip, prep_q8k_kernel).
//
// Orientation: MFMA A operand = activations (rows = tokens), B operand = weights (cols = weight
// rows), so a lane owns one weight row (its d/dmin are per-lane scalars) and the C store is 32
// consecutive floats per half-wave.
//
// Structure: work-group = 8 waves = 2 K-groups x 4 row-tiles on a 128-row x 64-token output tile (intra-
// work-group split-K: two waves per SIMD, so one wave's dequant VALU work runs under the other's MFMAs, and
// the K loop is half as long); weights go HBM -> VGPR -> fragment (no LDS, one super-block prefetched ahead),
// activations go global -> LDS directly (global_load_lds) into two XOR-swizzled tiles per K-group, one
// barrier per super-block; the two partial accumulators meet in LDS at the end.  Tiles are assigned to
// work-groups XCD-aware so an XCD's work-groups share a token tile.
#include "gemm_common.h"
#ifndef GEMM_DIAG
#define GEMM_DIAG 0
#endif

#include <stdlib.h>

#define TOK_TILE 64

// per-super-block weight operands a wave keeps in registers
template <int TYPE>
struct wregs {
    uint4 qs[4];
    uint4 hd;    // Q4_K: {d, dmin, scales[12]};  Q6_K: 16 int8 scales
    uint4 qh[2]; // Q6_K only
    uint32_t dw; // Q6_K only (f16 bits)
};

struct sregs { // per-super-block activation-side scales
    float4_t_ d8[2][4];
    half8_t xm[2]; // Q4_K only
};

#define GEMM_KG 2 // K-groups per work-group (intra-work-group split-K)

template <int TYPE>
__global__ __launch_bounds__(512) void gemm_kq_kernel(const uint8_t *__restrict__ A, long m, int nb,
                                                      const _Float16 *__restrict__ Xh, const float *__restrict__ d8T,
                                                      const _Float16 *__restrict__ Xm, long n, long n_pad,
                                                      float *__restrict__ C, long ldc, int n_rb, int n_wg) {
    // per K-group two activation tiles (64 tokens x 256 codes, f16): the global->LDS copy of the next
    // super-block runs under the MFMAs of the current one.  2 x 2 x 32 KiB = 128 KiB.
    __shared__ __attribute__((aligned(16))) uint8_t xt[GEMM_KG][2][TOK_TILE * XT_ROW_BYTES];
    constexpr int TILE = TYPE == LFAMD_TYPE_Q4_K ? P4K_TILE : TYPE == LFAMD_TYPE_Q5_K ? P5K_TILE : P6K_TILE;
    constexpr bool MINS = TYPE == LFAMD_TYPE_Q4_K || TYPE == LFAMD_TYPE_Q5_K; // Q4_K family: {d, dmin, 6-bit scales/mins}
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = wave >> 2, rw = wave & 3; // K-group, row-tile inside the work-group
    const int i = lane & 31, h = lane >> 5;

    // XCD-aware tile assignment (T1): consecutive block ids go round-robin over the 8 XCDs, so give every
    // XCD one contiguous run of the linear tile order (row-blocks fastest): the work-groups of an XCD then
    // share a token tile (512 KiB of activation codes stays in that XCD's L2) and stream distinct weights.
    const int id = blockIdx.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = id & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    int tt, rb;
    tile_of(L, n_rb, n_wg / n_rb, rb, tt);

    const long n_row_tiles = (m + 31) / 32;
    const long rt = (long)rb * 4 + rw;
    const bool active = rt < n_row_tiles;
    const long n0 = (long)tt * TOK_TILE;
    const long k = (long)nb * 256;
    const uint8_t *tile0 = A + (size_t)(active ? rt : 0) * nb * TILE;
    const int nit = (nb - kg + GEMM_KG - 1) / GEMM_KG; // this K-group's super-blocks: kg, kg+2, ...
    const int nit_max = (nb + GEMM_KG - 1) / GEMM_KG;

    // ---- issue the operands of super-block it of this K-group: activations straight into LDS
    // (global_load_lds, the XOR swizzle applied on the SOURCE address since the LDS side is lane-linear),
    // weights to VGPRs
    auto prefetch = [&](int it, wregs<TYPE> &w) {
#if GEMM_DIAG == 2 // development: compute chain only (operands loaded once)
        if (it > 1)
            return;
#endif
        const int b = it * GEMM_KG + kg;
        uint8_t *dst = xt[kg][it & 1];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int wi = rw * 8 + e;        // wave-instruction: rows 2wi, 2wi+1 of the tile (1 KiB)
            const int nn = 2 * wi + h, p = i; // this lane fills slot p of row nn with logical chunk p ^ (nn & 15)
            const uint8_t *src = (const uint8_t *)Xh + ((size_t)b * n_pad + n0 + nn) * 512 + ((p ^ (nn & 15)) * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + wi * 1024), 16, 0, 0);
        }
        const uint8_t *tile = tile0 + (size_t)b * TILE;
#pragma unroll
        for (int g = 0; g < 4; g++)
            w.qs[g] = *(const uint4 *)(tile + g * 1024 + lane * 16);
        if constexpr (MINS) {
            w.hd = *(const uint4 *)(tile + P4K_HDR + i * 16); // P5K_HDR == P4K_HDR
            if constexpr (TYPE == LFAMD_TYPE_Q5_K)
                w.qh[0] = *(const uint4 *)(tile + P5K_QH + lane * 16);
        } else {
            w.qh[0] = *(const uint4 *)(tile + P6K_QH + 0 * 1024 + lane * 16);
            w.qh[1] = *(const uint4 *)(tile + P6K_QH + 1 * 1024 + lane * 16);
            w.hd = *(const uint4 *)(tile + P6K_SC + i * 16);
            w.dw = *(const uint16_t *)(tile + P6K_D + i * 2);
        }
    };
    // hipcc's waitcnt pass does not count LDS-DMA: beside it, the first use of ANY register load waits
    // vmcnt(0) (cdna_hip_programming.md §5 trap (b)).  The weights of the current super-block landed before
    // the previous barrier, so "use" every register of the set here — an empty asm, no instruction — and
    // only THEN issue the next super-block's loads: the wait is paid while nothing is outstanding, and the
    // MFMAs below run with the prefetch genuinely in flight.
    auto touch = [&](wregs<TYPE> &w) {
#pragma unroll
        for (int g = 0; g < 4; g++)
            asm volatile("" : "+v"(w.qs[g].x), "+v"(w.qs[g].y), "+v"(w.qs[g].z), "+v"(w.qs[g].w));
        asm volatile("" : "+v"(w.hd.x), "+v"(w.hd.y), "+v"(w.hd.z), "+v"(w.hd.w));
        if constexpr (TYPE == LFAMD_TYPE_Q5_K)
            asm volatile("" : "+v"(w.qh[0].x), "+v"(w.qh[0].y), "+v"(w.qh[0].z), "+v"(w.qh[0].w));
        if constexpr (TYPE == LFAMD_TYPE_Q6_K) {
            asm volatile("" : "+v"(w.qh[0].x), "+v"(w.qh[0].y), "+v"(w.qh[0].z), "+v"(w.qh[0].w));
            asm volatile("" : "+v"(w.qh[1].x), "+v"(w.qh[1].y), "+v"(w.qh[1].z), "+v"(w.qh[1].w));
            asm volatile("" : "+v"(w.dw));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_scales = [&](int it, sregs &sr) {
        const int b = it * GEMM_KG + kg;
        if constexpr (MINS) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
                sr.xm[nt] = *(const half8_t *)(Xm + ((size_t)b * n_pad + n0 + nt * 32 + i) * 16 + 8 * h);
        }
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++)
                sr.d8[nt][r4] = *(const float4_t_ *)(d8T + (size_t)b * n_pad + n0 + nt * 32 + 8 * r4 + 4 * h);
    };

    float16_t_ acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[nt][r] = 0.0f;

    // byte offset of this lane's fragment chunk inside the tile: row i, chunk (c ^ (i & 15)) with c = 2t + h.
    // c & 15 takes 8 values per lane (t & 7); bit 4 of c and the token tile are immediates.
    uint32_t xoff[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
        xoff[u] = (uint32_t)(i * XT_ROW_BYTES + ((((2 * u + h) & 15) ^ (i & 15)) * 16));
    const float16_t_ zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t magic = opaque_magic();

    auto compute = [&](int it, const wregs<TYPE> &w, const sregs &sr) {
#if GEMM_DIAG == 1 // development: memory + barrier chain only
        acc[0][0] += (float)(w.qs[0].x ^ w.qs[1].y ^ w.qs[2].z ^ w.qs[3].w ^ w.hd.x) + sr.d8[0][0][0] + sr.d8[1][3][3];
        return;
#endif
        const uint8_t *xb = xt[kg][it & 1];
        float16_t_ tmp[2]; // first K-step accumulates onto the constant 0 (an inline operand, no register init)
        const uint32_t qw[16] = {w.qs[0].x, w.qs[0].y, w.qs[0].z, w.qs[0].w, w.qs[1].x, w.qs[1].y, w.qs[1].z, w.qs[1].w,
                                 w.qs[2].x, w.qs[2].y, w.qs[2].z, w.qs[2].w, w.qs[3].x, w.qs[3].y, w.qs[3].z, w.qs[3].w};
        if constexpr (MINS) {
            const uint4 hd = w.hd;
            const uint32_t hq5[4] = {w.qh[0].x, w.qh[0].y, w.qh[0].z, w.qh[0].w}; // Q5_K only
            (void)hq5;
            const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
            uint32_t sc03, sc47, mn03, mn47;
            q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
#pragma unroll
            for (int j = 0; j < 8; j++) { // 32-wide sub-block = K-steps 2j, 2j+1
                const float scf = (float)(((j < 4 ? sc03 : sc47) >> (8 * (j & 3))) & 0xff);
                const half2_t S = bcast_h2(scf), O = bcast_h2(-1024.0f * scf);
                const half2_t S16 = bcast_h2(scf * 0.0625f), O16 = bcast_h2(-64.0f * scf);
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int t = 2 * j + e;
                    half8_t wf;
                    if constexpr (TYPE == LFAMD_TYPE_Q5_K)
                        wf = dequant_q5(qw[t], hq5[t >> 2] >> (t & 3), S, O, S16, O16, magic);
                    else
                        wf = dequant_q4(qw[t], S, O, S16, O16, magic);
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) {
                        const half8_t xf = *(const half8_t *)(xb + xoff[t & 7] + nt * 32 * XT_ROW_BYTES + ((2 * t) & 16) * 16);
                        tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, wf, t == 0 ? zero16 : tmp[nt], 0, 0, 0);
                    }
                }
            }
            // ---- mins: one MFMA per token tile, K = 16 = {lo parts | hi parts} of the 8 pair sums
            frag_u wm;
            const float mscale = h ? 64.0f : 1.0f;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t mw = p < 2 ? mn03 : mn47;
                const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
                half2_t v = {(_Float16)(m0 * mscale), (_Float16)(m1 * mscale)};
                wm.p[p] = v;
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const float16_t_ tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(sr.xm[nt], wm.v, zero16, 0, 0, 0);
                // ---- per-super-block scaling: acc += d8[n] * (d * tmp - dmin * tm)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        const float u = fmaf(-dmin, tm[r], d * tmp[nt][r]);
                        acc[nt][r] = fmaf(u, sr.d8[nt][r4][e], acc[nt][r]);
                    }
            }
        } else { // Q6_K: 16-wide sub-blocks, one per K-step; no mins
            const uint32_t hw[8] = {w.qh[0].x, w.qh[0].y, w.qh[0].z, w.qh[0].w, w.qh[1].x, w.qh[1].y, w.qh[1].z, w.qh[1].w};
            const uint32_t scw[4] = {w.hd.x, w.hd.y, w.hd.z, w.hd.w};
            const float dw = h2f((uint16_t)w.dw);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const float scf = (float)(int)(int8_t)((scw[t >> 2] >> (8 * (t & 3))) & 0xff);
                const half2_t S = bcast_h2(scf);
                uint32_t H = hw[t >> 1];
                if (t & 1)
                    H >>= 2;
                const half8_t wf = dequant_q6(qw[t], H, S);
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    const half8_t xf = *(const half8_t *)(xb + xoff[t & 7] + nt * 32 * XT_ROW_BYTES + ((2 * t) & 16) * 16);
                    tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, wf, t == 0 ? zero16 : tmp[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * r4 + e;
                        acc[nt][r] = fmaf(dw * tmp[nt][r], sr.d8[nt][r4][e], acc[nt][r]);
                    }
        }
    };

    // ---- software pipeline, two weight register sets (no copies): super-block it+1 is in flight while it
    // computes.  __syncthreads() drains the LDS-DMA (vmcnt(0)) before the tile is read one iteration later.
    // Both K-groups run the same number of barriers (nit_max); a group without a super-block left idles.
    wregs<TYPE> wa, wb;
    sregs sr;
    if (0 < nit)
        prefetch(0, wa);
    __syncthreads();
    for (int it = 0; it < nit_max; it += 2) {
        if (it < nit) {
            touch(wa);
            // unconditional (index clamped): a branch around the loads ends in register copies at the join,
            // i.e. a use, i.e. a vmcnt(0) right after the issue
            prefetch(it + 1 < nit ? it + 1 : nit - 1, wb);
            load_scales(it, sr); // consumed after the MFMAs
            compute(it, wa, sr);
        }
        __syncthreads();
        if (it + 1 < nit_max) {
            if (it + 1 < nit) {
                touch(wb);
                prefetch(it + 2 < nit ? it + 2 : nit - 1, wa);
                load_scales(it + 1, sr);
                compute(it + 1, wb, sr);
            }
            __syncthreads();
        }
    }

    // ---- combine the two K-groups through LDS (the activation tiles are dead now), then store:
    // reg r of token tile nt is token n0 + 32nt + (r&3) + 8(r>>2) + 4h, weight row 32rt + i
    float *red = (float *)&xt[0][0][0]; // [rw][nt][r][lane] : 4 x 2 x 16 x 64 floats = 32 KiB
    if (kg == 1) {
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                red[((rw * 2 + nt) * 16 + r) * 64 + lane] = acc[nt][r];
    }
    __syncthreads();
    if (kg == 0 && active) {
        const long row = rt * 32 + i;
        if (row < m) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const long tok = n0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (tok < n)
                        C[tok * ldc + row] = acc[nt][r] + red[((rw * 2 + nt) * 16 + r) * 64 + lane];
                }
        }
    }
}

// =====================================================================================================
// EXPERIMENTAL (LFAMD_GEMM_LDS3=1; default is gemm_kq_kernel above): measured 43.8 us vs 41.0 us at
// 4096x4096x512 — with one wave per SIMD the ~500 VALU instructions per super-block (dequant, scale constants,
// AGPR moves, epilogue) are issue-bound at 4 cycles each, so removing the load stalls alone does not pay; kept
// because the hand-counted LDS-DMA pipeline is the base for the next step (two waves per SIMD).
// Q4_K, three-stage LDS-DMA pipeline.  EVERY global read of the K loop is a global_load_lds (activation
// tile, this wave's weight tile, scales), so no register-destination load exists for hipcc to wait on and the
// waits are hand-counted: s_waitcnt vmcnt(G3_PER_STAGE) leaves the youngest stage in flight across a raw
// s_barrier (cdna_hip_programming.md §5 "Pipelining across barriers", T3+T4).  Two super-blocks stream while
// one computes.  4 waves, 128 rows x 64 tokens per work-group, 156.75 KiB of LDS (one work-group per CU).
//
//   stage (53 504 B):  X 64 tokens x 512 B (XOR-swizzled chunks) | W 4 waves x 4608 B (packed tile of the
//                      wave's 32 rows) | d8 64 x f32 | xm 64 tokens x 32 B | 256 B spare
//   iteration b:  wait stage b (vmcnt) ; barrier ; issue stage b+2 ; compute stage b
// The barrier both publishes stage b (every wave waited for its own part first) and retires stage b-1,
// whose slot is the one refilled right after it.

#define G3_X 0
#define G3_W 32768
#define G3_D8 (G3_W + 4 * P4K_TILE)
#define G3_XM (G3_D8 + 256)
#define G3_SPARE (G3_XM + 2048)
#define G3_STAGE (G3_SPARE + 256)
#define G3_PER_STAGE 14 // global_load_lds instructions every wave issues per stage

// LDS-DMA in inline asm (cdna_hip_programming.md §5.7, glds16_asm): hipcc's waitcnt pass makes every ds_read
// wait vmcnt(0) for a pending __builtin_amdgcn_global_load_lds it cannot prove disjoint (here: always, even
// with one LDS object per stage), which would drain the pipeline at the top of every compute phase.  Hidden in
// asm, the DMA is invisible to that pass and ordered only by the hand-counted s_waitcnt + s_barrier below.
// lds_wave_base must be wave-uniform (an SGPR); M0 is saved/restored inside the same statement.
__device__ static inline void glds16(const void *gsrc, void *lds_wave_base) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(uint8_t *)lds_wave_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

__global__ __launch_bounds__(256) void gemm_q4k_lds3_kernel(const uint8_t *__restrict__ A, long m, int nb,
                                                            const _Float16 *__restrict__ Xh, const float *__restrict__ d8T,
                                                            const _Float16 *__restrict__ Xm, long n, long n_pad,
                                                            float *__restrict__ C, long ldc, int n_rb, int n_wg) {
    // three distinct LDS objects, addressed with compile-time stage numbers (loop unrolled x3): hipcc's waitcnt
    // pass makes a ds_read wait for every pending LDS-DMA it cannot prove disjoint, i.e. vmcnt(0), if the stage
    // is picked by a run-time index into one array
    __shared__ __attribute__((aligned(16))) uint8_t lds0[G3_STAGE];
    __shared__ __attribute__((aligned(16))) uint8_t lds1[G3_STAGE];
    __shared__ __attribute__((aligned(16))) uint8_t lds2[G3_STAGE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;

    // XCD-aware tile assignment (see gemm_kq_kernel)
    const int id = blockIdx.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = id & 7;
    const int Lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    int tt, rb;
    tile_of(Lin, n_rb, n_wg / n_rb, rb, tt);

    const long n_row_tiles = (m + 31) / 32;
    const long rt = (long)rb * 4 + wave;
    const bool active = rt < n_row_tiles;
    const long n0 = (long)tt * TOK_TILE;
    const long k = (long)nb * 256;
    const uint8_t *wtile0 = A + (size_t)(active ? rt : 0) * nb * P4K_TILE; // inactive waves stream tile 0, unused

    auto issue = [&](int b, uint8_t *st) {
        // activations: wave w fills rows 16w .. 16w+15 (8 instructions x 2 rows); lane p of a row fetches the
        // logical chunk p ^ (row & 15) so the image is swizzled although the LDS side is lane-linear
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int wi = wave * 8 + e;
            const int nn = 2 * wi + h;
            glds16((const uint8_t *)Xh + ((size_t)b * n_pad + n0 + nn) * 512 + ((i ^ (nn & 15)) * 16),
                   st + G3_X + wi * 1024);
        }
        // this wave's packed weight tile: 4 x 1 KiB of nibbles + 512 B of headers (upper half-wave idle)
        const uint8_t *wt = wtile0 + (size_t)b * P4K_TILE;
        uint8_t *wl = st + G3_W + wave * P4K_TILE;
#pragma unroll
        for (int g = 0; g < 4; g++)
            glds16(wt + g * 1024 + lane * 16, wl + g * 1024);
        if (lane < 32)
            glds16(wt + P4K_HDR + lane * 16, wl + P4K_HDR);
        // scales: one more instruction per wave (wave 0: d8, waves 1-2: the two halves of xm, wave 3: spare)
        if (wave == 0) {
            if (lane < 16)
                glds16((const uint8_t *)(d8T + (size_t)b * n_pad + n0) + lane * 16, st + G3_D8);
        } else if (wave == 3) {
            if (lane < 16)
                glds16((const uint8_t *)(d8T + (size_t)b * n_pad + n0) + lane * 16, st + G3_SPARE);
        } else {
            // xm rows are 32 B per (token, super-block), token stride nb*32 B: lane = (token half*32.., 16-B half)
            const int tok = (wave - 1) * 32 + (lane >> 1);
            glds16((const uint8_t *)Xm + ((size_t)b * n_pad + n0 + tok) * 32 + (lane & 1) * 16,
                   st + G3_XM + (wave - 1) * 1024);
        }
    };

    float16_t_ acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            acc[nt][r] = 0.0f;

    // byte offset of this lane's fragment chunk inside a token row, per K-step parity: (c ^ (i & 15)) * 16 with
    // c = 2t + h; c & 15 takes 16 values -> precomputed once, the rest is an immediate
    uint32_t xoff[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
        xoff[u] = (uint32_t)(i * XT_ROW_BYTES + ((((2 * u + h) & 15) ^ (i & 15)) * 16));
    const uint32_t magic = opaque_magic();

    auto compute = [&](const uint8_t *st) {
        const uint8_t *xb = st + G3_X;
        const uint8_t *wl = st + G3_W + wave * P4K_TILE;
        float16_t_ tmp[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                tmp[nt][r] = 0.0f;
        const uint4 hd = *(const uint4 *)(wl + P4K_HDR + i * 16);
        const float d = h2f((uint16_t)(hd.x & 0xffff)), dmin = h2f((uint16_t)(hd.x >> 16));
        uint32_t sc03, sc47, mn03, mn47;
        q4k_scales_bytes(hd.y, hd.z, hd.w, sc03, sc47, mn03, mn47);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint4 q4 = *(const uint4 *)(wl + g * 1024 + lane * 16);
            const uint32_t qw[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (int e2 = 0; e2 < 2; e2++) { // sub-block j = 2g + e2 = K-steps 4g + 2e2, +1
                const int j = 2 * g + e2;
                const float scf = (float)(((j < 4 ? sc03 : sc47) >> (8 * (j & 3))) & 0xff);
                const half2_t S = bcast_h2(scf), O = bcast_h2(-1024.0f * scf);
                const half2_t S16 = bcast_h2(scf * 0.0625f), O16 = bcast_h2(-64.0f * scf);
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int t = 4 * g + 2 * e2 + e;
                    const half8_t wf = dequant_q4(qw[2 * e2 + e], S, O, S16, O16, magic);
                    // chunk c = 2t + h: bit 4 of c is an immediate (256 B), the low 4 bits come from xoff
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) {
                        const half8_t xf = *(const half8_t *)(xb + xoff[t & 7] + nt * 32 * XT_ROW_BYTES + ((2 * t) & 16) * 16);
                        tmp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, wf, tmp[nt], 0, 0, 0);
                    }
                }
            }
        }
        // ---- mins: one MFMA per token tile, K = 16 = {lo parts | hi parts} of the 8 pair sums
        frag_u wm;
        const float mscale = h ? 64.0f : 1.0f;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t mw = p < 2 ? mn03 : mn47;
            const float m0 = (float)((mw >> (16 * (p & 1))) & 0xff), m1 = (float)((mw >> (16 * (p & 1) + 8)) & 0xff);
            half2_t v = {(_Float16)(m0 * mscale), (_Float16)(m1 * mscale)};
            wm.p[p] = v;
        }
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const half8_t xm = *(const half8_t *)(st + G3_XM + (nt * 32 + i) * 32 + h * 16);
            float16_t_ tm;
#pragma unroll
            for (int r = 0; r < 16; r++)
                tm[r] = 0.0f;
            tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wm.v, tm, 0, 0, 0);
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const float4_t_ d8 = *(const float4_t_ *)(st + G3_D8 + (nt * 32 + 8 * r4 + 4 * h) * 4);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int r = 4 * r4 + e;
                    const float u = fmaf(-dmin, tm[r], d * tmp[nt][r]);
                    acc[nt][r] = fmaf(u, d8[e], acc[nt][r]);
                }
            }
        }
    };

    // ---- pipeline (unrolled x3: stage objects are compile-time)
    auto wait_stage = [&](int b) {
        // stage b is complete once at most the younger stage's G3_PER_STAGE loads remain (none at the end)
        if (b + 1 < nb)
            asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    issue(0, lds0);
    if (nb > 1)
        issue(1, lds1);
    for (int b = 0; b < nb; b += 3) {
        wait_stage(b);
        if (b + 2 < nb)
            issue(b + 2, lds2);
        compute(lds0);
        if (b + 1 < nb) {
            wait_stage(b + 1);
            if (b + 3 < nb)
                issue(b + 3, lds0);
            compute(lds1);
        }
        if (b + 2 < nb) {
            wait_stage(b + 2);
            if (b + 4 < nb)
                issue(b + 4, lds1);
            compute(lds2);
        }
    }

    // ---- store: reg r of token tile nt is token n0 + 32nt + (r&3) + 8(r>>2) + 4h, weight row 32rt + i
    if (active) {
        const long row = rt * 32 + i;
        if (row < m) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const long tok = n0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (tok < n)
                        C[tok * ldc + row] = acc[nt][r];
                }
        }
    }
}

extern "C" hipError_t lfamd_launch_gemm_kq(int Atype, const void *A, long m, long k, const void *Xh, const void *d8T,
                                           const void *Xm, long n, long n_pad, float *C, long ldc, hipStream_t s) {
    if (m <= 0 || n <= 0)
        return hipSuccess;
    int nb = (int)(k / 256);
    long n_row_tiles = (m + 31) / 32;
    const int n_rb = (int)((n_row_tiles + 3) / 4), n_tt = (int)(n_pad / TOK_TILE);
    const int n_wg = n_rb * n_tt;
    if (Atype == LFAMD_TYPE_Q4_K && getenv("LFAMD_GEMM_LDS3")) { // experimental variant, see the kernel's header
        gemm_q4k_lds3_kernel<<<n_wg, 256, 0, s>>>((const uint8_t *)A, m, nb, (const _Float16 *)Xh,
                                                              (const float *)d8T, (const _Float16 *)Xm, n, n_pad, C, ldc, n_rb,
                                                              n_wg);
    } else if (Atype == LFAMD_TYPE_Q4_K)
        gemm_kq_kernel<LFAMD_TYPE_Q4_K><<<n_wg, 512, 0, s>>>((const uint8_t *)A, m, nb, (const _Float16 *)Xh,
                                                               (const float *)d8T, (const _Float16 *)Xm, n, n_pad, C, ldc,
                                                               n_rb, n_wg);
    else if (Atype == LFAMD_TYPE_Q5_K)
        gemm_kq_kernel<LFAMD_TYPE_Q5_K><<<n_wg, 512, 0, s>>>((const uint8_t *)A, m, nb, (const _Float16 *)Xh,
                                                               (const float *)d8T, (const _Float16 *)Xm, n, n_pad, C, ldc,
                                                               n_rb, n_wg);
    else if (Atype == LFAMD_TYPE_Q6_K)
        gemm_kq_kernel<LFAMD_TYPE_Q6_K><<<n_wg, 512, 0, s>>>((const uint8_t *)A, m, nb, (const _Float16 *)Xh,
                                                               (const float *)d8T, (const _Float16 *)Xm, n, n_pad, C, ldc,
                                                               n_rb, n_wg);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
