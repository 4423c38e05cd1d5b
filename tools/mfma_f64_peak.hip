// Measures the sustained v_mfma_f64_16x16x4_f64 rate of the chip it runs on (no memory traffic):
// the practical ceiling the weighted-SYRK kernel is priced against next to the 78.6 TFLOP/s
// vendor figure.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, const double* in, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(const char* label, int blocks, int threads, int iters, double* dout, double* din) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<NACC>), dim3(blocks), dim3(threads), 0, 0, dout, din, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mfma_loop<NACC>), dim3(blocks), dim3(threads), 0, 0, dout, din, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64.0;
    const double flops = waves * (double)iters * NACC * 2048.0;
    printf("%-44s blocks=%5d thr=%4d  %8.3f ms  %7.2f TFLOP/s\n", label, blocks, threads, ms, flops / ms / 1e9);
}

int main() {
    double *dout, *din; hipMalloc(&dout, sizeof(double) * 4096 * 1024); hipMalloc(&din, sizeof(double) * 128);
    double h[128];
    srand(1);
    for (int i = 0; i < 128; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    const int it = 20000;
    run<16>("random data, 1 wave/SIMD, 16 acc", 256, 256, it, dout, din);
    run<16>("random data, 2 waves/SIMD, 16 acc", 512, 256, it, dout, din);
    run<16>("random data, 4 waves/SIMD, 16 acc", 1024, 256, it, dout, din);
    run<4>("random data, 1 wave/SIMD, 4 acc", 256, 256, it * 4, dout, din);
    run<1>("random data, 1 wave/SIMD, 1 acc (dependent)", 256, 256, it * 8, dout, din);
    for (int i = 0; i < 128; ++i) h[i] = 0.0;
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    run<16>("zero data, 2 waves/SIMD, 16 acc", 512, 256, it, dout, din);
    return 0;
}
