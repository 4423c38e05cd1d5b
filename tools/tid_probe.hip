// Probe: buffer_load ... lds with ADD_TID_ENABLE (stride 16, no VGPR address).  The source buffer is
// 64 MB so that a mis-decoded stride cannot leave the allocation; prints the byte stride observed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* X, int flags, double* out) {
    __shared__ double lds[256];
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)16, 64, flags);
    if (threadIdx.x < 64) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, 0, 4096, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __syncthreads();
    if (threadIdx.x < 128) out[threadIdx.x] = lds[threadIdx.x];
}
int main() {
    const size_t n = 8u << 20;                       // 8M doubles = 64 MB
    std::vector<double> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (double)i;
    double *d, *o; hipMalloc(&d, n * 8); hipMalloc(&o, 128 * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    for (int flags : {1 << 23}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, flags, o);
        double r[128]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("flags 0x%x: first values %.0f %.0f %.0f %.0f ... lane 63: %.0f %.0f (expect 512 513 514 515 ... 638 639 for stride 16 at byte offset 4096)\n",
               flags, r[0], r[1], r[2], r[3], r[126], r[127]);
    }
    return 0;
}
