// mfma_vmem_probe.hip -- what does a vector-memory (or LDS) instruction cost a wave that is otherwise issuing fp64
// MFMAs back to back?  (Timing lab, not part of the product.)
// Each wave loops: 16 x v_mfma_f64_16x16x4_f64 (independent accumulators) + NV memory instructions of one kind, placed
// either as one group after the MFMA block (PLACE = 0) or spread one per 16/NV MFMAs (PLACE = 1).
//   KIND 0: global_load_dwordx4 from a small L1/L2-resident buffer into registers nobody reads
//   KIND 1: global_load_lds_dwordx4 (LDS-DMA) into a scratch LDS area
//   KIND 2: ds_read_b128 from LDS
//   KIND 3: global_load_dwordx4 with a scalar base (saddr) + 32-bit VGPR offset
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_vmem_probe.hip -o tools/mfma_vmem_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__device__ __forceinline__ void mem_op(const double* gp, double* lp, d2& sink, int off) {
    if (KIND == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(gp + off) : "memory");
    if (KIND == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + off),
                                                    (__attribute__((address_space(3))) void*)(lp), 16, 0, 0);
    if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "+v"(sink) : "v"((unsigned)(uintptr_t)lp + (unsigned)(threadIdx.x & 63) * 16u) : "memory");
    if (KIND == 4) asm volatile("v_mul_f64 %0, %0, 1.0" : "+v"(sink[0]));
    if (KIND == 5) asm volatile("v_mov_b64 %0, %1" : "+v"(sink[0]) : "v"(sink[1]));
    if (KIND == 6) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(sink[0]) : "v"(sink[1]));
    if (KIND == 7) { unsigned t = (unsigned)off; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(t)); sink[0] += (double)0; (void)t; }
    if (KIND == 8) { float t = (float)off; asm volatile("v_mul_f32 %0, %0, %0" : "+v"(t)); (void)t; }
    if (KIND == 3) {
        const unsigned voff = (unsigned)(threadIdx.x & 63) * 16u + (unsigned)off * 8u;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(sink) : "v"(voff), "s"(gp) : "memory");
    }
}

template <int KIND, int NV, int PLACE>
__global__ __launch_bounds__(256, 2)
void probe_kernel(const double* __restrict__ buf, int iters, double* __restrict__ out)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    d4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + lane, b = 0.5 - lane;
    d2 sink[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) sink[t] = (d2){0.0, 0.0};
    const double* gp = (KIND == 3) ? buf : buf + lane * 2;
    double* lp = lds + wave * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            if (PLACE == 1 && NV > 0 && (t % (16 / (NV > 16 ? 16 : NV))) == 0 && t / (16 / (NV > 16 ? 16 : NV)) < NV) {
                __builtin_amdgcn_sched_barrier(0);
                mem_op<KIND>(gp, lp, sink[(t / (16 / NV)) & 7], 128 * ((t / (16 / NV)) & 7));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PLACE == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) mem_op<KIND>(gp, lp, sink[v & 7], 128 * (v & 7));
        }
        __builtin_amdgcn_sched_barrier(0);
        if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // bound the number in flight
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
#pragma unroll
    for (int t = 0; t < 8; ++t) s += sink[t][0] * 1e-300;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV, int PLACE>
static void run(const double* buf, double* out, int wgs_per_cu, const char* name) {
    const int iters = 4000;
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe_kernel<KIND, NV, PLACE>), dim3(grid), dim3(256), 32768, 0, buf, 100, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_kernel<KIND, NV, PLACE>), dim3(grid), dim3(256), 32768, 0, buf, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // cycles per loop iteration per SIMD at 2.4 GHz; each SIMD runs wgs_per_cu waves (one wave of every workgroup)
    const double cyc = ms * 1e-3 * 2.4e9 / iters;
    const double ideal = 16.0 * 64.0 * wgs_per_cu;
    printf("%-10s NV %2d place %d waves/SIMD %d: %8.1f cycles per iteration (MFMA floor %6.0f)  +%6.1f per memory instruction per wave\n",
           name, NV, PLACE, wgs_per_cu, cyc, ideal, NV ? (cyc - ideal) / (NV * wgs_per_cu) : 0.0);
}

int main() {
    double *buf, *out;
    CHECK(hipMalloc(&buf, 1 << 20));
    CHECK(hipMemset(buf, 0, 1 << 20));
    CHECK(hipMalloc(&out, (size_t)256 * 2 * 256 * 8));
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, 0>(buf, out, w, "none");
        run<0, 2, 0>(buf, out, w, "gload"); run<0, 4, 0>(buf, out, w, "gload"); run<0, 8, 0>(buf, out, w, "gload");
        run<0, 4, 1>(buf, out, w, "gload"); run<0, 8, 1>(buf, out, w, "gload");
        run<3, 4, 0>(buf, out, w, "gload-s"); run<3, 8, 0>(buf, out, w, "gload-s"); run<3, 8, 1>(buf, out, w, "gload-s");
        run<1, 2, 0>(buf, out, w, "lds-dma"); run<1, 4, 0>(buf, out, w, "lds-dma"); run<1, 8, 0>(buf, out, w, "lds-dma");
        run<1, 4, 1>(buf, out, w, "lds-dma"); run<1, 8, 1>(buf, out, w, "lds-dma");
        run<2, 4, 0>(buf, out, w, "ds_read"); run<2, 8, 0>(buf, out, w, "ds_read"); run<2, 16, 0>(buf, out, w, "ds_read");
        run<2, 8, 1>(buf, out, w, "ds_read"); run<2, 16, 1>(buf, out, w, "ds_read");
        run<4, 4, 0>(buf, out, w, "v_mul_f64"); run<4, 8, 0>(buf, out, w, "v_mul_f64"); run<4, 16, 0>(buf, out, w, "v_mul_f64"); run<4, 16, 1>(buf, out, w, "v_mul_f64");
        run<5, 8, 0>(buf, out, w, "v_mov_b64"); run<5, 16, 0>(buf, out, w, "v_mov_b64"); run<5, 16, 1>(buf, out, w, "v_mov_b64");
        run<6, 8, 0>(buf, out, w, "lshl_add64"); run<6, 16, 0>(buf, out, w, "lshl_add64");
        run<7, 16, 0>(buf, out, w, "v_add_u32"); run<8, 16, 0>(buf, out, w, "v_mul_f32");
    }
    return 0;
}
