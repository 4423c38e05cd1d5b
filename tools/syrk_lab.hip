// syrk_lab.hip -- standalone timing lab for the weighted SYRK S = X^T diag(c) X (not part of the product library).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/syrk_lab.hip -o tools/syrk_lab
// Run:   tools/syrk_lab [N] [variant] [n_splits] [reps]
//
// Variant 1: wave-private operands -- every wavefront owns a 64 x 64 block of S and loads its MFMA fragments
// straight from global memory into registers (no LDS stage, no workgroup barrier), DEPTH k-steps in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <type_traits>

typedef int64_t i64;
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(double* X, i64 n, uint64_t seed) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
        z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        X[i] = (double)(int64_t)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    }
}

// ---- variant 1 ------------------------------------------------------------------------------------------------
// Work item = (split, 64 x 64 block (I, J), I >= J, of the 16 x 16 half-panel grid).  One wave per item.
// Lane (i = lane & 15, k = lane >> 4) of k-step ks holds, for h = 0, 1:
//   a[h] = X[n0 + 4 ks + k][64 I + 32 h + 2 i .. +1],   b[h] likewise with J   (one 16-byte load each)
// so MFMA tile m = 2 h + p covers columns 32 h + 2 i + p of the block (a fixed permutation, undone at the store).
template <int DEPTH>
__global__ __launch_bounds__(256, 2)
void syrk_wave_kernel(const double* __restrict__ X, i64 ldx, i64 N, const double* __restrict__ cpad,
                      int n_blocks /* 136 */, i64 rows_per_split, double* __restrict__ partial)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // item order inside the grid: blockIdx -> (split, group of 4 blocks); off-diagonal blocks first
    const int groups = n_blocks / 4;                    // 34
    const int g = blockIdx.x % groups;
    const int split = blockIdx.x / groups;
    const int item = g * 4 + wave;                      // 0 .. 135: 0..119 off-diagonal, 120..135 diagonal
    int I, J;
    if (item < 120) {
        I = (int)((1.f + sqrtf(1.f + 8.f * (float)item)) * 0.5f);
        while (I * (I - 1) / 2 > item) --I;
        while ((I + 1) * I / 2 <= item) ++I;
        J = item - I * (I - 1) / 2;
    } else { I = J = item - 120; }
    const bool diag = (I == J);

    i64 r0 = (i64)split * rows_per_split, r1 = r0 + rows_per_split;
    if (r1 > N) r1 = N;
    if (r0 > N) r0 = N;
    const int nks = (int)((r1 - r0 + 3) / 4);

    d4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};

    const int l15 = lane & 15, l4 = lane >> 4;
    const double* pa = X + (i64)64 * I + 2 * l15;
    const double* pb = X + (i64)64 * J + 2 * l15;

    auto body = [&](auto diag_tag) {
        constexpr bool DIAG = decltype(diag_tag)::value;
        d2 fa[DEPTH][2], fb[DEPTH][2];
        double fc[DEPTH];
        auto issue = [&](int ks, int slot) {
            i64 n = r0 + (i64)4 * ks + l4;
            if (n > N - 1) n = N - 1;                       // clamped: readable; c is zero past N
            const double* ra = pa + n * ldx;
            fa[slot][0] = *reinterpret_cast<const d2*>(ra);
            fa[slot][1] = *reinterpret_cast<const d2*>(ra + 32);
            if (!DIAG) {
                const double* rb = pb + n * ldx;
                fb[slot][0] = *reinterpret_cast<const d2*>(rb);
                fb[slot][1] = *reinterpret_cast<const d2*>(rb + 32);
            }
            fc[slot] = cpad[r0 + (i64)4 * ks + l4];
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d, d);
        for (int ks = 0; ks < nks; ks += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const double cv = (ks + d < nks) ? fc[d] : 0.0;
                double as[4], bs[4];
                bs[0] = fa[d][0][0]; bs[1] = fa[d][0][1]; bs[2] = fa[d][1][0]; bs[3] = fa[d][1][1];
                as[0] = bs[0] * cv; as[1] = bs[1] * cv; as[2] = bs[2] * cv; as[3] = bs[3] * cv;
                if (!DIAG) { bs[0] = fb[d][0][0]; bs[1] = fb[d][0][1]; bs[2] = fb[d][1][0]; bs[3] = fb[d][1][1]; }
                __builtin_amdgcn_sched_barrier(0);
                issue(ks + d + DEPTH, d);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (!DIAG || (m >> 1) >= (n >> 1))
                            acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[m], bs[n], acc[m * 4 + n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (diag) body(std::true_type{}); else body(std::false_type{});

    double* out = partial + ((i64)split * n_blocks + item) * 4096;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(32 * (m >> 1) + 2 * (l4 + 4 * r) + (m & 1)) * 64 + 32 * (n >> 1) + 2 * l15 + (n & 1)] = acc[m * 4 + n][r];
}

// ---- variant 2: the same work split, loads written as inline asm with exact vmcnt waits ------------------------------
// "+v": the destination is updated in place, so a slot keeps its physical registers around the loop (an "=v" output lets
// the compiler rename the slot and copy it at the back edge -- before the data has landed)
#define GLOAD4(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "+v"(dst) : "v"(ptr) : "memory")
#define GLOAD2(dst, ptr) asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(dst) : "v"(ptr) : "memory")
template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); }

template <int DEPTH, bool DIAG>
__device__ __forceinline__ void wave_body(const double* pa, const double* pb, const double* pc, i64 ldx, i64 rows_left_clamp,
                                          int nks, d4 (&acc)[16])
{
    constexpr int LPS = DIAG ? 3 : 5;                  // loads per slot
    d2 fa[DEPTH][2], fb[DEPTH][2];
    double fc[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { fa[d][0] = fa[d][1] = fb[d][0] = fb[d][1] = (d2){0.0, 0.0}; fc[d] = 0.0; }
    const i64 step = 4 * ldx;
    i64 adv = 0;                                        // element offset of the next k-step to load (clamped)
    auto issue = [&](int slot) {
        const double* ra = pa + adv;
        GLOAD4(fa[slot][0], ra, 0);
        GLOAD4(fa[slot][1], ra, 256);
        if (!DIAG) {
            const double* rb = pb + adv;
            GLOAD4(fb[slot][0], rb, 0);
            GLOAD4(fb[slot][1], rb, 256);
        }
        GLOAD2(fc[slot], pc);
        pc += 4;
        if (adv + step <= rows_left_clamp) adv += step;  // stay on a readable row at the very end (c is zero there)
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d);
    for (int ks = 0; ks < nks; ks += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            wait_vm<LPS * (DEPTH - 1)>();
            const double cv = (ks + d < nks) ? fc[d] : 0.0;
            double as[4], bs[4];
            bs[0] = fa[d][0][0]; bs[1] = fa[d][0][1]; bs[2] = fa[d][1][0]; bs[3] = fa[d][1][1];
            as[0] = bs[0] * cv; as[1] = bs[1] * cv; as[2] = bs[2] * cv; as[3] = bs[3] * cv;
            if (!DIAG) { bs[0] = fb[d][0][0]; bs[1] = fb[d][0][1]; bs[2] = fb[d][1][0]; bs[3] = fb[d][1][1]; }
            // private copies: the loads issued next overwrite the slot while the MFMAs below still read operands
            asm volatile("" : "+v"(as[0]), "+v"(as[1]), "+v"(as[2]), "+v"(as[3]));
            asm volatile("" : "+v"(bs[0]), "+v"(bs[1]), "+v"(bs[2]), "+v"(bs[3]));
            __builtin_amdgcn_sched_barrier(0);
            issue(d);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (!DIAG || (m >> 1) >= (n >> 1))
                        acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[m], bs[n], acc[m * 4 + n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vm<0>();
}

template <int DEPTH>
__global__ __launch_bounds__(256, 2)
void syrk_wave2_kernel(const double* __restrict__ X, i64 ldx, i64 N, const double* __restrict__ cpad,
                       int n_blocks /* 136 */, i64 rows_per_split, double* __restrict__ partial)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int groups = n_blocks / 4;
    const int g = blockIdx.x % groups;
    const int split = blockIdx.x / groups;
    const int item = g * 4 + wave;
    int I, J;
    if (item < 120) {
        I = (int)((1.f + sqrtf(1.f + 8.f * (float)item)) * 0.5f);
        while (I * (I - 1) / 2 > item) --I;
        while ((I + 1) * I / 2 <= item) ++I;
        J = item - I * (I - 1) / 2;
    } else { I = J = item - 120; }
    i64 r0 = (i64)split * rows_per_split, r1 = r0 + rows_per_split;
    if (r1 > N) r1 = N;
    if (r0 > N) r0 = N;
    const int nks = (int)((r1 - r0 + 3) / 4);
    d4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int l15 = lane & 15, l4 = lane >> 4;
    if (nks > 0) {
        i64 nfirst = r0 + l4; if (nfirst > N - 1) nfirst = N - 1;
        const double* pa = X + nfirst * ldx + 64 * I + 2 * l15;
        const double* pb = X + nfirst * ldx + 64 * J + 2 * l15;
        const double* pc = cpad + r0 + l4;
        // largest element advance that keeps this lane's row <= N - 1
        const i64 clamp = ((N - 1 - nfirst) / 4) * 4 * ldx;
        if (I == J) wave_body<DEPTH, true>(pa, pb, pc, ldx, clamp, nks, acc);
        else        wave_body<DEPTH, false>(pa, pb, pc, ldx, clamp, nks, acc);
    }
    double* out = partial + ((i64)split * n_blocks + item) * 4096;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(32 * (m >> 1) + 2 * (l4 + 4 * r) + (m & 1)) * 64 + 32 * (n >> 1) + 2 * l15 + (n & 1)] = acc[m * 4 + n][r];
}

__global__ void reduce_kernel(const double* partial, int n_splits, i64 elems, double* out) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    double s = 0.0;
    for (int k = 0; k < n_splits; ++k) s += partial[(i64)k * elems + e];
    out[e] = s;
}

int main(int argc, char** argv) {
    const i64 N = argc > 1 ? atoll(argv[1]) : 1000000;
    const int variant = argc > 2 ? atoi(argv[2]) : 1;
    int S = argc > 3 ? atoi(argv[3]) : 128;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const int depth = argc > 5 ? atoi(argv[5]) : 3;
    const int P = 1024;
    const i64 LD = argc > 6 ? atoll(argv[6]) : P;      // row stride in doubles (lab: padding against cache-set aliasing)
    const int hot = argc > 7 ? atoi(argv[7]) : 0;      // 1: every k-step re-reads the first rows (L1/L2 hits only)
    double *X, *c, *partial, *out;
    CHECK(hipMalloc(&X, (size_t)N * LD * 8));
    CHECK(hipMalloc(&c, (size_t)(N + 4096) * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, N * LD, 12345ull);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, c, N, 777ull);
    CHECK(hipMemset(c + N, 0, 4096 * 8));
    CHECK(hipDeviceSynchronize());
    const int n_blocks = 136;
    i64 rps = (N + S - 1) / S; rps = ((rps + 3) / 4) * 4;
    const i64 elems = (i64)n_blocks * 4096;
    CHECK(hipMalloc(&partial, (size_t)S * elems * 8));
    CHECK(hipMalloc(&out, (size_t)elems * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = S * (n_blocks / 4);
    auto launch = [&]() {
        if (variant == 2) {
            if (depth == 2) hipLaunchKernelGGL(syrk_wave2_kernel<2>, dim3(grid), dim3(256), 0, 0, X, hot ? (i64)0 : LD, N, c, n_blocks, rps, partial);
            else if (depth == 4) hipLaunchKernelGGL(syrk_wave2_kernel<4>, dim3(grid), dim3(256), 0, 0, X, hot ? (i64)0 : LD, N, c, n_blocks, rps, partial);
            else if (depth == 5) hipLaunchKernelGGL(syrk_wave2_kernel<5>, dim3(grid), dim3(256), 0, 0, X, hot ? (i64)0 : LD, N, c, n_blocks, rps, partial);
            else hipLaunchKernelGGL(syrk_wave2_kernel<3>, dim3(grid), dim3(256), 0, 0, X, hot ? (i64)0 : LD, N, c, n_blocks, rps, partial);
            return;
        }
        if (depth == 2) hipLaunchKernelGGL(syrk_wave_kernel<2>, dim3(grid), dim3(256), 0, 0, X, hot ? (i64)0 : LD, N, c, n_blocks, rps, partial);
        else if (depth == 4) hipLaunchKernelGGL(syrk_wave_kernel<4>, dim3(grid), dim3(256), 0, 0, X, hot ? (i64)0 : LD, N, c, n_blocks, rps, partial);
        else hipLaunchKernelGGL(syrk_wave_kernel<3>, dim3(grid), dim3(256), 0, 0, X, hot ? (i64)0 : LD, N, c, n_blocks, rps, partial);
    };
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops = (double)N * P * (P + 1);
    printf("variant %d depth %d ld %lld hot %d N %lld splits %d: %.3f ms  %.2f TFLOP/s  (%.3f of 78.6)\n", variant, depth, (long long)LD, hot, (long long)N, S, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 78.6);
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, 0, partial, S, elems, out);
    CHECK(hipDeviceSynchronize());
    // spot check against a host sum over the first rows only when N is small, else over sampled entries
    std::vector<double> hout(elems);
    CHECK(hipMemcpy(hout.data(), out, elems * 8, hipMemcpyDeviceToHost));
    const i64 NC = N < 20000 ? N : 20000;          // compare a second run restricted to NC rows
    if (NC == N && !hot) {
        std::vector<double> hx((size_t)N * LD), hc(N);
        CHECK(hipMemcpy(hx.data(), X, (size_t)N * LD * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hc.data(), c, (size_t)N * 8, hipMemcpyDeviceToHost));
        double worst = 0.0, scale = 0.0;
        for (int t = 0; t < 400; ++t) {
            int a = (t * 7919 + 13) % P, b = (t * 104729 + 7) % P;
            if (a < b) { int q = a; a = b; b = q; }
            double s = 0.0;
            for (i64 n = 0; n < N; ++n) s += hc[n] * hx[n * LD + a] * hx[n * LD + b];
            const int I = a / 64, J = b / 64;
            const int item = (I == J) ? 120 + I : I * (I - 1) / 2 + J;
            double got = hout[(i64)item * 4096 + (a % 64) * 64 + (b % 64)];
            if (I == J && (a % 64) / 32 == (b % 64) / 32 && (a % 64) < (b % 64)) continue;
            worst = fmax(worst, fabs(got - s)); scale = fmax(scale, fabs(s));
        }
        printf("check over %lld rows: max abs err %.3e (scale %.3e)\n", (long long)N, worst, scale);
    }
    return 0;
}
