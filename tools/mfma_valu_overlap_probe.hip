// development probe: do the matrix pipe and the vector ALU of a SIMD run side by side?  (MI355X)
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap_probe.hip -o /tmp/mvo.bin && /tmp/mvo.bin
// One work-group per CU, 4 / 8 waves (one / two per SIMD).  An iteration is 4 x v_mfma_i32_32x32x32_i8 (independent accumulators)
// and / or 16 x v_pk_mul_lo_u16 (independent chains); printed: s_memtime cycles per iteration as a wave sees them.
//   mfma        : the MFMAs alone                           valu        : the VALU instructions alone
//   both        : 16 VALU then 4 MFMA, unrelated registers   dependent   : the MFMAs' B operand is what the VALU just wrote
//   interleaved : 4 x (4 VALU, 1 MFMA), unrelated registers
//   split roles : 8 waves, waves 0..3 run `mfma`, waves 4..7 run `valu` (both times printed)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REPS 256

#define MFMA(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VALU4(r0, r1, r2, r3, s)                                                                                     \
    asm volatile("v_pk_mul_lo_u16 %0, %4, %0\n\tv_pk_mul_lo_u16 %1, %4, %1\n\tv_pk_mul_lo_u16 %2, %4, %2\n\tv_pk_mul_lo_u16 %3, %4, %3" \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)                                                            \
                 : "v"(s))

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long *out, uint32_t *sink) {
    const int wave = threadIdx.x >> 6;
    v16i acc[4];
    for (int k = 0; k < 4; k++)
        for (int e = 0; e < 16; e++)
            acc[k][e] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3};
    v4i b[4];
    uint32_t r[16];
    for (int k = 0; k < 16; k++)
        r[k] = threadIdx.x * 3 + k;
    for (int k = 0; k < 4; k++)
        b[k] = v4i{(int)r[4 * k], (int)r[4 * k + 1], (int)r[4 * k + 2], (int)r[4 * k + 3]};
    uint32_t s = 0x00010001u;
    int mode = MODE;
    if (MODE == 5)
        mode = wave < 4 ? 0 : 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
        for (int i = 0; i < REPS; i++) {
            MFMA(acc[0], a, b[0]);
            MFMA(acc[1], a, b[1]);
            MFMA(acc[2], a, b[2]);
            MFMA(acc[3], a, b[3]);
        }
    } else if (mode == 1) {
        for (int i = 0; i < REPS; i++) {
            VALU4(r[0], r[1], r[2], r[3], s);
            VALU4(r[4], r[5], r[6], r[7], s);
            VALU4(r[8], r[9], r[10], r[11], s);
            VALU4(r[12], r[13], r[14], r[15], s);
        }
    } else if (mode == 2) {
        for (int i = 0; i < REPS; i++) {
            VALU4(r[0], r[1], r[2], r[3], s);
            VALU4(r[4], r[5], r[6], r[7], s);
            VALU4(r[8], r[9], r[10], r[11], s);
            VALU4(r[12], r[13], r[14], r[15], s);
            MFMA(acc[0], a, b[0]);
            MFMA(acc[1], a, b[1]);
            MFMA(acc[2], a, b[2]);
            MFMA(acc[3], a, b[3]);
        }
    } else if (mode == 3) {
        for (int i = 0; i < REPS; i++) {
            VALU4(b[0].x, b[0].y, b[0].z, b[0].w, s);
            VALU4(b[1].x, b[1].y, b[1].z, b[1].w, s);
            VALU4(b[2].x, b[2].y, b[2].z, b[2].w, s);
            VALU4(b[3].x, b[3].y, b[3].z, b[3].w, s);
            MFMA(acc[0], a, b[0]);
            MFMA(acc[1], a, b[1]);
            MFMA(acc[2], a, b[2]);
            MFMA(acc[3], a, b[3]);
        }
    } else if (mode == 4) {
        for (int i = 0; i < REPS; i++) {
            VALU4(r[0], r[1], r[2], r[3], s);
            MFMA(acc[0], a, b[0]);
            VALU4(r[4], r[5], r[6], r[7], s);
            MFMA(acc[1], a, b[1]);
            VALU4(r[8], r[9], r[10], r[11], s);
            MFMA(acc[2], a, b[2]);
            VALU4(r[12], r[13], r[14], r[15], s);
            MFMA(acc[3], a, b[3]);
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 16 + wave] = t1 - t0;
    uint32_t x = 0;
    for (int k = 0; k < 16; k++)
        x ^= r[k];
    for (int k = 0; k < 4; k++)
        for (int e = 0; e < 16; e++)
            x ^= (uint32_t)acc[k][e];
    for (int k = 0; k < 4; k++)
        x ^= (uint32_t)(b[k].x ^ b[k].y ^ b[k].z ^ b[k].w);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    unsigned long long *out, h[256 * 16];
    uint32_t *sink;
    hipMalloc(&out, sizeof(h));
    hipMalloc(&sink, 256 * 512 * 4);
    struct {
        const char *name;
        void (*fn)(unsigned long long *, uint32_t *);
        int split;
    } ks[] = {{"mfma (4 MFMA)", probe<0>, 0},        {"valu (16 VALU)", probe<1>, 0},           {"both (16 VALU, 4 MFMA)", probe<2>, 0},
              {"dependent (16 VALU -> 4 MFMA)", probe<3>, 0}, {"interleaved 4 x (4 VALU, 1 MFMA)", probe<4>, 0}, {"split roles", probe<5>, 1}};
    for (auto &k : ks)
        for (int waves : {4, 8}) {
            if (k.split && waves != 8)
                continue;
            hipMemset(out, 0, sizeof(h));
            for (int rep = 0; rep < 3; rep++)
                hipLaunchKernelGGL(k.fn, dim3(256), dim3(waves * 64), 0, 0, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double sa = 0, sb = 0;
            for (int bI = 0; bI < 256; bI++)
                for (int w = 0; w < waves; w++)
                    (w < 4 ? sa : sb) += (double)h[bI * 16 + w];
            if (k.split)
                printf("%-36s %d waves/WG: MFMA waves %7.1f, VALU waves %7.1f cycles per iteration\n", k.name, waves, sa / (256 * 4) / REPS,
                       sb / (256 * 4) / REPS);
            else
                printf("%-36s %d waves/WG: %7.1f cycles per iteration and wave\n", k.name, waves, (sa + sb) / (256 * waves) / REPS);
        }
    return 0;
}
