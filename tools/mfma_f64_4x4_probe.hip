// Probes v_mfma_f64_4x4x4_4b_f64 on the GPU it runs on: which (A lane, B lane) pairs contribute to which
// output lane, and its issue rate next to v_mfma_f64_16x16x4_f64 (development aid).
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_4x4_probe.hip -o tools/mfma_f64_4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned long long* mask /* [64 la][64 lb] */) {
    const int lane = threadIdx.x, la = blockIdx.x >> 6, lb = blockIdx.x & 63;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    const unsigned long long m = __ballot(d != 0.0);
    if (lane == 0) mask[blockIdx.x] = m;
}

template <int KIND>
__global__ void rate(double* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    if (KIND == 0) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[u], 0, 0, 0);
        double s = 0; for (int u = 0; u < 8; ++u) s += acc[u];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        d4 acc[8];
        for (int u = 0; u < 8; ++u) acc[u] = (d4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        double s = 0; for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

int main() {
    unsigned long long* dm; hipMalloc(&dm, 4096 * 8);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, dm);
    static unsigned long long hm[4096];
    if (hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost) != hipSuccess) { printf("probe failed\n"); return 2; }
    // hypothesis: A lane = i + 4 blk + 16 k, B lane = j + 4 blk + 16 k, D lane = j + 4 blk + 16 i  (and variants)
    int printed = 0;
    for (int la = 0; la < 64 && printed < 40; ++la) for (int lb = 0; lb < 64 && printed < 40; ++lb)
        if (hm[la * 64 + lb]) {
            if (la < 6 || (la % 17) == 0) { printf("A lane %2d x B lane %2d -> D lanes:", la, lb); for (int l = 0; l < 64; ++l) if (hm[la * 64 + lb] >> l & 1) printf(" %d", l); printf("\n"); ++printed; }
        }
    int bad = 0;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        const int ia = la & 3, ba = (la >> 2) & 3, ka = la >> 4, jb = lb & 3, bb = (lb >> 2) & 3, kb = lb >> 4;
        unsigned long long want = 0;
        if (ba == bb && ka == kb) want = 1ull << (jb + 4 * ba + 16 * ia);
        if (hm[la * 64 + lb] != want) ++bad;
    }
    printf("PROBE 4x4x4_4b: mismatches with map A(i + 4 blk + 16 k), B(j + 4 blk + 16 k), D(j + 4 blk + 16 i) = %d\n", bad);
    double* out; hipMalloc(&out, 1024 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, out, iters);
            else           hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)1024 * 4 * iters * 8 * (kind == 0 ? 512.0 : 2048.0);
        printf("RATE %s: %.2f ms, %.1f TFLOP/s\n", kind == 0 ? "4x4x4_4b" : "16x16x4", ms, flops / ms / 1e9);
    }
    return 0;
}
