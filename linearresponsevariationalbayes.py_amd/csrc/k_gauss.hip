// k_gauss.hip -- the Gaussian-loss shortcut of the Hessian build: curvature and c o y per observation, S beta from the tile-packed
// lower triangle, value and gradient from the sums the weighted SYRK already formed (X is read once per build).
#include "lrvb_internal.h"
#include "k_kernels.h"
#include <math.h>

// l = 1/2 tau (z - y)^2: the curvature c_n = w_n tau does not depend on theta, and with S = X^T diag(c) X,
// r = X^T (c o y):   d f / d beta = S beta - r,   sum_n w_n l_n = 1/2 beta^T S beta - beta^T r + 1/2 sum c y^2.
// r rides on the SYRK's diagonal tiles (k_wsyrk.hip), so a Hessian build reads X exactly once.
// c = w tau, c y, and the block's share of sum c y^2 (2048 observations per block; shares summed in a fixed order later)
__global__ __launch_bounds__(256)
void gauss_coef_kernel(i64 n, double tau, const double* __restrict__ w, const double* __restrict__ y,
                       double* __restrict__ cw, double* __restrict__ cy, double* __restrict__ cyy_part) {
    __shared__ double sh[256];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const i64 i = (i64)blockIdx.x * 2048 + k * 256 + threadIdx.x;
        if (i < n) { const double cv = w[i] * tau, yv = y[i]; cw[i] = cv; cy[i] = cv * yv; s += cv * yv * yv; }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) cyy_part[blockIdx.x] = sh[0];
}

// S beta from the tile-packed lower triangle, one workgroup per tile (bi >= bj), every tile row read once, coalesced:
// wave w takes rows w, w + 4, ...; the row dot products are the tile's share of (S beta) in block row bi, and -- off
// the diagonal, or below it inside a diagonal tile -- the same loaded values accumulate the share of block row bj
// (the transposed tile).  part: [2 nb][nb * 128], zeroed by the caller; slot [bj] holds row shares, [nb + bi] column shares.
__global__ __launch_bounds__(1024)
void tiles_symv_kernel(const double* __restrict__ tiles, int nb, i64 P, const double* __restrict__ beta, double* __restrict__ part)
{
    __shared__ double colsh[16][128];         // sixteen waves: eight rows of the tile each (four waves were a 32-step latency chain)
    const int t = blockIdx.x;
    int bi = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const bool diag = bi == bj;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* tile = tiles + (i64)t * (WS_TILE * WS_TILE);
    const i64 width = (i64)nb * WS_TILE;
    const i64 j0 = (i64)bj * WS_TILE + lane, j1 = j0 + 64;
    const double b0 = j0 < P ? beta[j0] : 0.0, b1 = j1 < P ? beta[j1] : 0.0;
    double c0 = 0.0, c1 = 0.0;
    for (int ii = wave; ii < WS_TILE; ii += 16) {
        const i64 i = (i64)bi * WS_TILE + ii;
        double s0 = tile[ii * WS_TILE + lane], s1 = tile[ii * WS_TILE + lane + 64];
        if (i >= P) { s0 = 0.0; s1 = 0.0; }
        if (diag) { if (lane > ii) s0 = 0.0; if (lane + 64 > ii) s1 = 0.0; }        // lower triangle incl. the diagonal
        double d = s0 * b0 + s1 * b1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (lane == 0 && i < P) part[(i64)bj * width + i] = d;
        const double bi_val = i < P ? beta[i] : 0.0;
        // transposed share: strictly below the diagonal inside a diagonal tile (the diagonal itself is in the row share)
        c0 += ((diag && lane == ii) ? 0.0 : s0) * bi_val;
        c1 += ((diag && lane + 64 == ii) ? 0.0 : s1) * bi_val;
    }
    colsh[wave][lane] = c0; colsh[wave][lane + 64] = c1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const i64 j = (i64)bj * WS_TILE + threadIdx.x;
        if (j < P) {
            double cs = colsh[0][threadIdx.x];
#pragma unroll
            for (int g = 1; g < 16; ++g) cs += colsh[g][threadIdx.x];
            part[(i64)(nb + bi) * width + j] = cs;
        }
    }
}

// one block: g = S beta - r from the shares, value = 1/2 beta^T S beta - beta^T r + 1/2 sum c y^2
__global__ __launch_bounds__(1024)
void gauss_finish_kernel(const double* __restrict__ part, int nb, i64 P, const double* __restrict__ beta, const double* __restrict__ r,
                         const double* __restrict__ cyy_part, int n_cyy, double* __restrict__ value_out, double* __restrict__ g_out)
{
    __shared__ double sh[1024];
    const i64 width = (i64)nb * WS_TILE;
    double acc_v = 0.0;
    for (int k = threadIdx.x; k < n_cyy; k += 1024) acc_v += 0.5 * cyy_part[k];
    for (i64 i = threadIdx.x; i < P; i += 1024) {
        double acc = 0.0;
        for (int k = 0; k < 2 * nb; ++k) acc += part[(i64)k * width + i];
        g_out[i] = acc - r[i];
        acc_v += beta[i] * (0.5 * acc - r[i]);
    }
    sh[threadIdx.x] = acc_v;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) *value_out = sh[0];
}
