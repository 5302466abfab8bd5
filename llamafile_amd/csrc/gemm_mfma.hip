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

extern "C" hipError_t lfamd_launch_gemm_kq(int Atype, const void *A, long m, long k, const void *Xh, const void *d8T,
                                           const void *Xm, long n, long n_pad, float *C, long ldc, hipStream_t s) {
    if (m <= 0 || n <= 0)
        return hipSuccess;
    int nb = (int)(k / 256);
    long n_row_tiles = (m + 31) / 32;
    const int n_rb = (int)((n_row_tiles + 3) / 4), n_tt = (int)(n_pad / TOK_TILE);
    const int n_wg = n_rb * n_tt;
    if (Atype == LFAMD_TYPE_Q4_K)
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
