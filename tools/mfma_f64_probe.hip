// Probes the lane -> (row, col) maps of v_mfma_f64_16x16x4_f64 on the GPU it runs on, with
// exact small-integer data, and reports whether they match what csrc/k_wsyrk.hip assumes:
//   A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
//   D reg r -> (row = (lane >> 4) + 4 r, col = lane & 15).
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A /*16x4 row-major*/, const double* B /*4x16*/, double* out /*64x4*/) {
    const int lane = threadIdx.x;
    const double a = A[(lane & 15) * 4 + (lane >> 4)];
    const double b = B[(lane >> 4) * 16 + (lane & 15)];
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

int main() {
    double hA[64], hB[64], hC[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = (double)(1 + i * 5 + k * 3);      // asymmetric
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (double)(2 + k * 7 + j * j);     // asymmetric
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    if (hipDeviceSynchronize() != hipSuccess) { printf("PROBE: kernel failed\n"); return 2; }
    hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    int bad_assumed = 0, bad_f32style = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const double v = hC[lane * 4 + r];
        if (v != ref[((lane >> 4) + 4 * r) * 16 + (lane & 15)]) ++bad_assumed;
        if (v != ref[(4 * (lane >> 4) + r) * 16 + (lane & 15)]) ++bad_f32style;
    }
    printf("PROBE mfma_f64_16x16x4: mismatches with assumed map (row=(lane>>4)+4r) = %d; with f32-style map (row=4(lane>>4)+r) = %d\n", bad_assumed, bad_f32style);
    if (bad_assumed != 0) {
        // print where each lane/reg value is found in ref, to derive the true map
        for (int lane = 0; lane < 64; lane += 7) for (int r = 0; r < 4; ++r) {
            const double v = hC[lane * 4 + r];
            for (int e = 0; e < 256; ++e) if (ref[e] == v) printf("  lane %d reg %d -> row %d col %d\n", lane, r, e / 16, e % 16);
        }
    }
    return bad_assumed == 0 ? 0 : 1;
}
