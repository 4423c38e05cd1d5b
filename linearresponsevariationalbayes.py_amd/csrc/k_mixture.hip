// k_mixture.hip -- launchers of the mixture-model pipeline (BASELINE.json config 3); the per-row kernels live in
// k_mixture_rows.h and are instantiated for K = 2 .. 32 in k_mixture_inst0..3.hip.
#include "k_mixture_rows.h"
#include <stdlib.h>

int mixture_rows_launch_0(lrvb_ctx* c, int K, unsigned grid, unsigned dgrid, const double* theta_z_dev, int V,
                           const double* lam_dev, double* Amat_dev, i64 lda, double* U_dev, double* gfree_dev,
                           int* bad_dev, int* todo, int* todo_count);
int mixture_rows_launch_1(lrvb_ctx* c, int K, unsigned grid, unsigned dgrid, const double* theta_z_dev, int V,
                           const double* lam_dev, double* Amat_dev, i64 lda, double* U_dev, double* gfree_dev,
                           int* bad_dev, int* todo, int* todo_count);
int mixture_rows_launch_2(lrvb_ctx* c, int K, unsigned grid, unsigned dgrid, const double* theta_z_dev, int V,
                           const double* lam_dev, double* Amat_dev, i64 lda, double* U_dev, double* gfree_dev,
                           int* bad_dev, int* todo, int* todo_count);
int mixture_rows_launch_3(lrvb_ctx* c, int K, unsigned grid, unsigned dgrid, const double* theta_z_dev, int V,
                           const double* lam_dev, double* Amat_dev, i64 lda, double* U_dev, double* gfree_dev,
                           int* bad_dev, int* todo, int* todo_count);

// Xk[n, :] = packed lower triangle of x~_n x~_n^T, row length q(q+1)/2 (+1 zero column when odd)
__global__ __launch_bounds__(256)
void kron_rows_kernel(const double* __restrict__ X, int V, i64 N, double* __restrict__ Xk, i64 ldk)
{
    __shared__ double xs[4][32];
    const int q = V + 1, qp = q * (q + 1) / 2;
    const int sub = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (i64 n0 = (i64)blockIdx.x * 4; n0 < N; n0 += (i64)gridDim.x * 4) {     // one wavefront per row
        const i64 n = n0 + sub;
        __syncthreads();
        if (n < N && lane < q) xs[sub][lane] = lane == 0 ? 1.0 : X[n * V + lane - 1];
        __syncthreads();
        if (n < N)
            for (int e = lane; e < (int)ldk; e += 64) {
                // packed pair index e = a (a + 1) / 2 + b, b <= a
                int a = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
                while (a * (a + 1) / 2 > e) --a;
                while ((a + 1) * (a + 2) / 2 <= e) ++a;
                const int b = e - a * (a + 1) / 2;
                Xk[n * ldk + e] = (e < qp) ? xs[sub][a] * xs[sub][b] : 0.0;
            }
    }
}

// R (q^2 x K^2, row-major) from the packed result Rs (q(q+1)/2 x lda): both index pairs are symmetric
__global__ void mixture_expand_kernel(const double* __restrict__ Rs, i64 lda, int q, int K, double* __restrict__ Rfull)
{
    const i64 total = (i64)q * q * K * K;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (i64)gridDim.x * blockDim.x) {
        const int kk = (int)(e % ((i64)K * K)), jj = (int)(e / ((i64)K * K));
        int k = kk / K, kp = kk - k * K, j = jj / q, jp = jj - j * q;
        if (kp > k) { const int t = k; k = kp; kp = t; }
        if (jp > j) { const int t = j; j = jp; jp = t; }
        Rfull[e] = Rs[(i64)(j * (j + 1) / 2 + jp) * lda + k * (k + 1) / 2 + kp];
    }
}

int launch_mixture_expand(lrvb_ctx* c, const double* Rs, i64 lda, int q, int K, double* Rfull)
{
    const i64 total = (i64)q * q * K * K;
    i64 grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(mixture_expand_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, Rs, lda, q, K, Rfull);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

__global__ __launch_bounds__(256)
void mixture_val_reduce_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out2) {
    __shared__ double sh[4][2];                            // one workgroup, fixed order: deterministic (one wavefront walking
    double a = 0.0, b = 0.0;                               // 4096 pairs took 26 us)
    for (int i = threadIdx.x; i < nblk; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = mx_wave_sum(a); b = mx_wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = a; sh[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) { out2[0] = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]); out2[1] = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]); }
}

int launch_mixture_rows(lrvb_ctx* c, int K, const double* theta_z_dev, const double* lam_dev,
                        double* Amat_dev, i64 lda, double* U_dev, double* gfree_dev, double* val2_dev, int* bad_dev)
{
    const int V = (int)c->P;
    if (V + 1 > 32 || K > 32 || K < 2) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "mixture kernel supports V + 1 <= 32 and 2 <= K <= 32");
    if (c->N > 2147483647LL) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "mixture kernel indexes rows with 32 bits");
    if (lda != (i64)mixture_rows_lda(K)) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "mixture kernel writes rows of K (K + 1) / 2 doubles padded to an even length (got lda = %lld)", (long long)lda);
    i64 grid = ((c->N + 1) / 2 + 3) / 4;                  // two rows per wavefront, four wavefronts per workgroup
    if (grid > 4096) grid = 4096;
    LRVB_TRY(buf_reserve(c, c->part_val, (size_t)(2 * grid)));
    // todo list of rows for the dense pass: N ints + the counter, in the observation scratch buffer
    LRVB_TRY(buf_reserve(c, c->lp, (size_t)(c->N / 2 + 2)));
    int* todo = reinterpret_cast<int*>(c->lp.p) + 2;
    int* todo_count = reinterpret_cast<int*>(c->lp.p);
    HIP_TRY(hipMemsetAsync(bad_dev, 0, sizeof(int), c->stream));
    HIP_TRY(hipMemsetAsync(todo_count, 0, sizeof(int), c->stream));
    i64 dgrid = (c->N + 3) / 4;
    if (dgrid > 2048) dgrid = 2048;
    {
        int done = 0;
        done = done || mixture_rows_launch_0(c, K, (unsigned)grid, (unsigned)dgrid, theta_z_dev, V, lam_dev, Amat_dev, lda, U_dev, gfree_dev, bad_dev, todo, todo_count);
        done = done || mixture_rows_launch_1(c, K, (unsigned)grid, (unsigned)dgrid, theta_z_dev, V, lam_dev, Amat_dev, lda, U_dev, gfree_dev, bad_dev, todo, todo_count);
        done = done || mixture_rows_launch_2(c, K, (unsigned)grid, (unsigned)dgrid, theta_z_dev, V, lam_dev, Amat_dev, lda, U_dev, gfree_dev, bad_dev, todo, todo_count);
        done = done || mixture_rows_launch_3(c, K, (unsigned)grid, (unsigned)dgrid, theta_z_dev, V, lam_dev, Amat_dev, lda, U_dev, gfree_dev, bad_dev, todo, todo_count);
        if (!done) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "mixture kernel supports 2 <= K <= 32 (got %d)", K);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(mixture_val_reduce_kernel, dim3(1), dim3(256), 0, c->stream, c->part_val.p, (int)grid, val2_dev);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

int launch_kron_rows(lrvb_ctx* c, double* Xk_dev, i64 ldk)
{
    i64 grid = (c->N + 3) / 4;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(kron_rows_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, c->X.p, (int)c->P, c->N, Xk_dev, ldk);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
