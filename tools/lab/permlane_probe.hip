// Prints what v_permlane32_swap / v_permlane16_swap do on gfx950 (lane contents before: first = lane, second = 100 + lane).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
    int* d; hipMalloc(&d, 256 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 first ", "swap32 second", "swap16 first ", "swap16 second"};
    for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int l = 0; l < 64; l += 8) printf(" %3d", h[v * 64 + l]); printf("\n"); }
    return 0;
}
