// development probe: lane semantics of v_permlane16_swap / v_permlane32_swap (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o) {
    unsigned u = threadIdx.x;
    auto a = __builtin_amdgcn_permlane16_swap(u, u + 100, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(u, u + 100, false, false);
    o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"p16 [0]", "p16 [1]", "p32 [0]", "p32 [1]"};
    for (int r = 0; r < 4; r++) { printf("%s:", names[r]); for (int i = 0; i < 64; i += 8) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
