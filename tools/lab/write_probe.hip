// write_probe.hip -- how fast can 4.2 GB be WRITTEN, by access pattern (the mixture rows kernel's A_n rows: N x 528 doubles)
//   mode 0: linear, 8 B per lane (512 B contiguous per wave instruction)
//   mode 1: linear, 16 B per lane (1 KB contiguous per wave instruction)
//   mode 2: one row of 528 doubles per wave, 16 B per lane, rows taken in grid-stride order
//   mode 3: the MFMA-layout pattern: per row 12 instructions, each four 128 B segments at packed-triangle offsets
//   mode 4: as 2, but non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k0(double* A, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) A[e] = 1.0;
}
__global__ __launch_bounds__(256) void k1(d2* A, long n2) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n2; e += (long)gridDim.x * 256) A[e] = d2{1.0, 2.0};
}
template <int NT>
__global__ __launch_bounds__(256) void k2(double* A, long N, long lda) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long n = (long)blockIdx.x * 4 + wave; n < N; n += (long)gridDim.x * 4) {
        d2* row = reinterpret_cast<d2*>(A + n * lda);
        for (int e = lane; e < lda / 2; e += 64) {
            if (NT) __builtin_nontemporal_store(d2{1.0, 2.0}, row + e); else row[e] = d2{1.0, 2.0};
        }
    }
}
__global__ __launch_bounds__(256) void k3(double* A, long N, long lda) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    for (long n = (long)blockIdx.x * 4 + wave; n < N; n += (long)gridDim.x * 4) {
        double* arow = A + n * lda;
        for (int bi = 0; bi < 2; ++bi) for (int bj = 0; bj <= bi; ++bj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = 16 * bi + g + 4 * v, c = 16 * bj + j;
                if (c <= r) arow[r * (r + 1) / 2 + c] = 1.0;
            }
    }
}
int main(int argc, char** argv) {
    const long N = 1000000, lda = 528;
    double* A; hipMalloc(&A, N * lda * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode) for (int grid : {2048, 4096, 16384}) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k0, dim3(grid), dim3(256), 0, 0, A, N * lda);
            if (mode == 1) hipLaunchKernelGGL(k1, dim3(grid), dim3(256), 0, 0, (d2*)A, N * lda / 2);
            if (mode == 2) hipLaunchKernelGGL(k2<0>, dim3(grid), dim3(256), 0, 0, A, N, lda);
            if (mode == 3) hipLaunchKernelGGL(k3, dim3(grid), dim3(256), 0, 0, A, N, lda);
            if (mode == 4) hipLaunchKernelGGL(k2<1>, dim3(grid), dim3(256), 0, 0, A, N, lda);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
        }
        printf("mode %d grid %5d: %.3f ms  %.2f TB/s\n", mode, grid, best, N * lda * 8 / best / 1e9);
    }
    return 0;
}
