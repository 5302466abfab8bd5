// development probe: issue cost of the VALU instructions the int8 GEMM body is made of, one wave per SIMD and two (MI355X).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/valu_rate_probe.bin && tools/valu_rate_probe.bin
// Each kernel runs REPS x 64 independent instructions of one kind (8 register chains) between two s_memtime stamps; printed:
// cycles per instruction as one wave sees it (a wave64 VALU instruction occupies the SIMD-32 for 2 cycles; a lone wave issues one
// per 4), for 4 waves per work-group (one per SIMD) and 8 (two per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REPS 64
#define BODY8(INS)                                                                                                            \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])               \
                 : "v"(s))

#define I_AND(k) "v_and_b32 %" #k ", %8, %" #k "\n\t"
#define I_PKMUL(k) "v_pk_mul_lo_u16 %" #k ", %8, %" #k "\n\t"
#define I_PERM(k) "v_perm_b32 %" #k ", %8, %" #k ", %8\n\t"
#define I_CVT(k) "v_cvt_f32_i32 %" #k ", %" #k "\n\t"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 3, %8\n\t"
#define I_FMA(k) "v_fma_f32 %" #k ", %8, %" #k ", %" #k "\n\t"
#define I_LSHR(k) "v_lshrrev_b32 %" #k ", 4, %" #k "\n\t"
#define I_MUL(k) "v_mul_f32 %" #k ", %8, %" #k "\n\t"
#define I_PKADDH(k) "v_pk_add_f16 %" #k ", %8, %" #k "\n\t"
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %8\n\t"
#define I_MULU24(k) "v_mul_u32_u24 %" #k ", %8, %" #k "\n\t"
#define I_MADI24(k) "v_mad_i32_i24 %" #k ", %8, %" #k ", %" #k "\n\t"
#define I_PKMAD(k) "v_pk_mad_u16 %" #k ", %8, %" #k ", %" #k "\n\t"
#define I_PKFMA32(k) "v_pk_fma_f32 %" #k ", %9, %" #k ", %" #k "\n\t"
#define I_PKMUL32(k) "v_pk_mul_f32 %" #k ", %9, %" #k "\n\t"

#define BODY8_64(INS)                                                                                                         \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])               \
                 : "v"(s), "v"(s2))
#define KERNEL64(NAME, INS)                                                                                                   \
    __global__ void NAME(unsigned long long *out, uint32_t *sink) {                                                           \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                                \
        f2 q[8];                                                                                                              \
        for (int k = 0; k < 8; k++)                                                                                           \
            q[k] = f2{(float)(threadIdx.x + k), 1.0f};                                                                        \
        uint32_t s = threadIdx.x | 0x01010101u;                                                                               \
        f2 s2 = {1.0001f, 0.9999f};                                                                                           \
        __syncthreads();                                                                                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                           \
        for (int i = 0; i < REPS; i++)                                                                                        \
            BODY8_64(INS);                                                                                                    \
        asm volatile("s_nop 0" ::: "memory");                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                           \
        if ((threadIdx.x & 63) == 0)                                                                                          \
            out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                                              \
        float acc = 0;                                                                                                        \
        for (int k = 0; k < 8; k++)                                                                                           \
            acc += q[k].x + q[k].y;                                                                                           \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = __builtin_bit_cast(uint32_t, acc);                                      \
    }

#define KERNEL(NAME, INS)                                                                                                     \
    __global__ void NAME(unsigned long long *out, uint32_t *sink) {                                                           \
        uint32_t r[8];                                                                                                        \
        for (int k = 0; k < 8; k++)                                                                                           \
            r[k] = threadIdx.x * 7 + k;                                                                                       \
        uint32_t s = threadIdx.x | 0x01010101u;                                                                               \
        __syncthreads();                                                                                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                           \
        for (int i = 0; i < REPS; i++)                                                                                        \
            BODY8(INS);                                                                                                       \
        asm volatile("s_nop 0" ::: "memory");                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                           \
        if ((threadIdx.x & 63) == 0)                                                                                          \
            out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                                              \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = r[0] ^ r[1] ^ r[2] ^ r[3] ^ r[4] ^ r[5] ^ r[6] ^ r[7];                    \
    }

KERNEL(k_and, I_AND)
KERNEL(k_pkmul, I_PKMUL)
KERNEL(k_perm, I_PERM)
KERNEL(k_cvt, I_CVT)
KERNEL(k_lshladd, I_LSHLADD)
KERNEL(k_fma, I_FMA)
KERNEL(k_lshr, I_LSHR)
KERNEL(k_mul, I_MUL)
KERNEL(k_pkaddh, I_PKADDH)
KERNEL(k_andor, I_ANDOR)
KERNEL(k_mulu24, I_MULU24)
KERNEL(k_madi24, I_MADI24)
KERNEL(k_pkmad, I_PKMAD)
KERNEL64(k_pkfma32, I_PKFMA32)
KERNEL64(k_pkmul32, I_PKMUL32)

int main() {
    unsigned long long *out, h[256 * 16];
    uint32_t *sink;
    hipMalloc(&out, sizeof(h));
    hipMalloc(&sink, 256 * 1024 * 4);
    struct {
        const char *name;
        void (*fn)(unsigned long long *, uint32_t *);
    } ks[] = {{"v_and_b32", k_and},         {"v_lshrrev_b32", k_lshr},   {"v_and_or_b32", k_andor}, {"v_perm_b32", k_perm},
              {"v_pk_mul_lo_u16", k_pkmul}, {"v_pk_mad_u16", k_pkmad},   {"v_mul_u32_u24", k_mulu24}, {"v_mad_i32_i24", k_madi24},
              {"v_lshl_add_u32", k_lshladd}, {"v_cvt_f32_i32", k_cvt},   {"v_mul_f32", k_mul},      {"v_fma_f32", k_fma},
              {"v_pk_add_f16", k_pkaddh}, {"v_pk_fma_f32", k_pkfma32}, {"v_pk_mul_f32", k_pkmul32}};
    for (auto &k : ks)
        for (int waves : {4, 8, 16}) {
            hipMemset(out, 0, sizeof(h));
            for (int rep = 0; rep < 3; rep++)
                hipLaunchKernelGGL(k.fn, dim3(256), dim3(waves * 64), 0, 0, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double sum = 0;
            int cnt = 0;
            for (int b = 0; b < 256; b++)
                for (int w = 0; w < waves; w++)
                    sum += (double)h[b * 16 + w], cnt++;
            printf("%-18s %2d waves/WG: %6.2f cycles per instruction per wave (%.2f per SIMD)\n", k.name, waves, sum / cnt / (REPS * 64.0),
                   sum / cnt / (REPS * 64.0) / (waves / 4));
        }
    return 0;
}
