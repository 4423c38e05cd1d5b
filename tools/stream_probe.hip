// stream_probe.hip -- how fast can ONE pass over an N x 1024 fp64 matrix be read on this chip, per load mechanism?
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB row pieces into a double-buffered 2 x 64 KiB stage, vmcnt(0) + barrier per
//           8-row chunk) -- the staging of hvp_multi_kernel with the arithmetic removed
//   mode 1: register loads (global_load_dwordx4 nt, the same 1 KiB row pieces, two chunks in flight per wave, no barrier)
//   mode 2: register loads, then ds_write_b128 into a wave-private LDS slice and one ds_read back (the register-staged
//           transposition a rewrite of hvp_multi_kernel would use)
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe ; run: tools/stream_probe [N]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define GLDS16_S(sbase, voff, ldsaddr) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")
constexpr int P = 1024, ROWS = 8, NW = 8;

__global__ __launch_bounds__(512, 1)
void k_dma(const double* __restrict__ X, long N, double* __restrict__ out)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
    const long nch = N / ROWS;
    unsigned voff[8];
    for (int j = 0; j < 8; ++j) voff[j] = (unsigned)((128 * j + 2 * lane) * 8);
    auto issue = [&](long ch, int buf) {
        const unsigned base = lds0 + (unsigned)(buf * ROWS * P) * 8u;
        const char* rowp = reinterpret_cast<const char*>(X + (ch * ROWS + wave) * (long)P);
#pragma unroll
        for (int j = 0; j < 8; ++j) GLDS16_S(rowp, voff[j], base + (unsigned)(wave * P + 128 * j) * 8u);
    };
    double acc = 0.0;
    long ch = blockIdx.x; int buf = 0;
    if (ch < nch) issue(ch, 0);
    for (; ch < nch; ch += gridDim.x) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        const long nxt = ch + gridDim.x;
        if (nxt < nch) issue(nxt, buf ^ 1);
        acc += lds[buf * ROWS * P + tid];                    // touch the stage
        buf ^= 1;
    }
    if (acc == 12345.678) out[tid] = acc;
}

template <int MODE>
__global__ __launch_bounds__(512, 1)
void k_reg(const double* __restrict__ X, long N, double* __restrict__ out)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const long nch = N / ROWS;
    // wave w owns columns [128 w, 128 w + 128) of every row of the chunk; lane (l15, l4): rows {l4, l4 + 4}, column pairs 32 h + 2 l15
    auto load = [&](long ch, d2 (&x)[8]) {
        const double* base = X + (ch * ROWS) * (long)P + 128 * wave + 2 * l15;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            x[h] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(base + (long)l4 * P + 32 * h));
            x[4 + h] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(base + (long)(l4 + 4) * P + 32 * h));
        }
    };
    double acc = 0.0;
    d2 xa[8], xb[8];
    long ch = blockIdx.x;
    if (ch < nch) load(ch, xa);
    double* mine = lds + wave * (ROWS * 128);                // wave-private slice: 8 rows x 128 columns
    for (;;) {
        long nxt = ch + gridDim.x;
        if (nxt < nch) load(nxt, xb);
        if (ch >= nch) break;
#pragma unroll
        for (int h = 0; h < 8; ++h) acc += xa[h][0] + xa[h][1];
        if (MODE == 2) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                *reinterpret_cast<d2*>(mine + l4 * 128 + 32 * h + 2 * l15) = xa[h];
                *reinterpret_cast<d2*>(mine + (l4 + 4) * 128 + 32 * h + 2 * l15) = xa[4 + h];
            }
            acc += mine[(lane & 3) * 128 + 8 * l4];
        }
        ch = nxt; nxt = ch + gridDim.x;
        if (nxt < nch) load(nxt, xa);
        if (ch >= nch) break;
#pragma unroll
        for (int h = 0; h < 8; ++h) acc += xb[h][0] + xb[h][1];
        if (MODE == 2) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                *reinterpret_cast<d2*>(mine + l4 * 128 + 32 * h + 2 * l15) = xb[h];
                *reinterpret_cast<d2*>(mine + (l4 + 4) * 128 + 32 * h + 2 * l15) = xb[4 + h];
            }
            acc += mine[(lane & 3) * 128 + 8 * l4];
        }
        ch = nxt;
    }
    if (acc == 12345.678) out[tid] = acc;
}


typedef double d4 __attribute__((ext_vector_type(4)));
// mode 3: the LDS-DMA staging with NM fp64 MFMAs (16x16x4) per wave and chunk between the issue of the next stage and the
// wait for it (how well do loads and matrix work overlap with ONE stage in flight?); NBUF stages of RB rows, vmcnt-counted
template <int NM, int NBUF, int RB>
__global__ __launch_bounds__(512, 1)
void k_dma_mfma(const double* __restrict__ X, long N, double* __restrict__ out)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
    const long nch = N / RB;
    constexpr int PER = RB * 8 / NW;              // DMA instructions per wave and stage (RB rows x 8 pieces / 8 waves)
    unsigned voff[8];
    for (int j = 0; j < 8; ++j) voff[j] = (unsigned)((128 * j + 2 * lane) * 8);
    auto issue = [&](long ch, int buf) {
        const unsigned base = lds0 + (unsigned)(buf * RB * P) * 8u;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int piece = wave * PER + i;      // piece = row * 8 + j
            const int row = piece >> 3, j = piece & 7;
            const char* rowp = reinterpret_cast<const char*>(X + (ch * RB + row) * (long)P);
            GLDS16_S(rowp, voff[j], base + (unsigned)(row * P + 128 * j) * 8u);
        }
    };
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + lane, b = 0.5;
    long ch = blockIdx.x;
    // prologue: NBUF - 1 stages in flight
    for (int k = 0; k < NBUF - 1; ++k) if (ch + (long)k * gridDim.x < nch) issue(ch + (long)k * gridDim.x, k);
    int buf = 0;
    for (; ch < nch; ch += gridDim.x) {
        // wait until only the NBUF - 2 younger stages may still be in flight
        if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
        if (NBUF == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
        __builtin_amdgcn_s_barrier();
        const long nxt = ch + (long)(NBUF - 1) * gridDim.x;
        int nb = buf + NBUF - 1; if (nb >= NBUF) nb -= NBUF;
        if (nxt < nch) issue(nxt, nb);
        else {                                     // keep the instruction count per iteration constant for the vmcnt arithmetic
#pragma unroll
            for (int i = 0; i < PER; ++i) GLDS16_S(reinterpret_cast<const char*>(X), voff[0], lds0 + (unsigned)(NBUF * RB * P) * 8u);
        }
        b += lds[buf * RB * P + tid];
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 3], 0, 0, 0);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678) out[tid] = acc[0][0];
}

__global__ void fill_kernel(double* X, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        X[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;      // pseudo-random in [-0.5, 0.5): all mantissa bits busy
    }
}

int main(int argc, char** argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 1000000;
    double *X, *out;
    hipMalloc(&X, (size_t)N * P * 8); hipMalloc(&out, 4096 * 8);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, N * P);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds_dma = 2 * ROWS * P * 8, lds_reg = NW * ROWS * 128 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {256, 512}) {
            if (mode == 0 && grid == 512) continue;           // 2 x 64 KiB of LDS: one workgroup per CU
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_dma, dim3(grid), dim3(512), lds_dma, 0, X, N, out);
                if (mode == 1) hipLaunchKernelGGL(k_reg<1>, dim3(grid), dim3(512), lds_reg, 0, X, N, out);
                if (mode == 2) hipLaunchKernelGGL(k_reg<2>, dim3(grid), dim3(512), lds_reg, 0, X, N, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("mode %d grid %d: %.3f ms  %.2f TB/s\n", mode, grid, best, (double)N * P * 8 / (best * 1e-3) / 1e12);
        }

#define RUN3(NM, NBUF, RB) do { \
        const size_t l = (size_t)(NBUF * RB * P + 1024) * 8; \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma_mfma<NM, NBUF, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); \
        float best = 1e9f; \
        for (int rep = 0; rep < 5; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL((k_dma_mfma<NM, NBUF, RB>), dim3(256), dim3(512), l, 0, X, N, out); \
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; } \
        printf("dma+mfma: %2d MFMAs per wave and %d-row stage, %d stages: %.3f ms  %.2f TB/s  (MFMA-only time %.3f ms)\n", NM, RB, NBUF, best, \
               (double)N * P * 8 / (best * 1e-3) / 1e12, (double)(N / RB) / 256.0 * NM * 2 * 64 / 2.4e9 * 1e3); } while (0)
    RUN3(0, 2, 8); RUN3(16, 2, 8); RUN3(24, 2, 8); RUN3(32, 2, 8);
    RUN3(8, 4, 4); RUN3(12, 4, 4); RUN3(16, 4, 4);
    RUN3(12, 3, 4); RUN3(16, 3, 4);
    return 0;
}
