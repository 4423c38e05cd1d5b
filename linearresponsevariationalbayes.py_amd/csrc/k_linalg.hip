// k_linalg.hip -- dense fp64 linear algebra behind the linear-response solve:
//   GEMM on v_mfma_f64_16x16x4_f64 (J^T H J, M H^-1 M^T, Cholesky trailing updates),
//   blocked Cholesky factor / solve (scipy.linalg.cho_factor / cho_solve at
//   LRVB/ModelSensitivity.py:594-602, LRVB/SparseObjectives.py:539-552),
//   and the vector kernels of the conjugate-gradient loop (LRVB/ConjugateGradient.py:81-85).
#include "lrvb_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int GM_TILE = 64;
constexpr int GM_KC = 16;
constexpr int GM_STRIDE = GM_TILE + 16;   // (stride mod 32) == 16 -> conflict-free ds_read_b64

// C = alpha * op(A) op(B) + beta * C, row-major; op(A) is M x K, op(B) is K x N.
__global__ __launch_bounds__(256)
void gemm_f64_kernel(int transA, int transB, i64 M, i64 N, i64 K, double alpha,
                     const double* __restrict__ A, i64 lda, const double* __restrict__ B, i64 ldb,
                     double beta, double* __restrict__ C, i64 ldc)
{
    __shared__ double As[GM_KC][GM_STRIDE];   // [k][m]
    __shared__ double Bs[GM_KC][GM_STRIDE];   // [k][n]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const i64 m0 = (i64)blockIdx.y * GM_TILE, n0 = (i64)blockIdx.x * GM_TILE;

    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

    for (i64 k0 = 0; k0 < K; k0 += GM_KC) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid;
            int kk, mm;
            if (transA) { mm = idx & 63; kk = idx >> 6; } else { kk = idx & 15; mm = idx >> 4; }
            const i64 gm = m0 + mm, gk = k0 + kk;
            double v = 0.0;
            if (gm < M && gk < K) v = transA ? A[gk * lda + gm] : A[gm * lda + gk];
            As[kk][mm] = v;
            int kb, nn;
            if (transB) { kb = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kb = idx >> 6; }
            const i64 gn = n0 + nn, gkb = k0 + kb;
            double u = 0.0;
            if (gn < N && gkb < K) u = transB ? B[gn * ldb + gkb] : B[gkb * ldb + gn];
            Bs[kb][nn] = u;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GM_KC / 4; ++ks) {
            const int krow = ks * 4 + (lane >> 4);
            double af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = As[krow][wr * 32 + a * 16 + (lane & 15)];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = Bs[krow][wc * 32 + b * 16 + (lane & 15)];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const i64 gm = m0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r;
                const i64 gn = n0 + wc * 32 + b * 16 + (lane & 15);
                if (gm < M && gn < N) {
                    double* dst = C + gm * ldc + gn;
                    const double prev = (beta == 0.0) ? 0.0 : beta * (*dst);
                    *dst = alpha * acc[a][b][r] + prev;
                }
            }
}

int launch_gemm(lrvb_ctx* c, bool transA, bool transB, i64 M, i64 Nn, i64 K, double alpha,
                const double* A, i64 lda, const double* B, i64 ldb, double beta, double* C, i64 ldc) {
    if (M <= 0 || Nn <= 0) return LRVB_OK;
    dim3 grid((unsigned)((Nn + GM_TILE - 1) / GM_TILE), (unsigned)((M + GM_TILE - 1) / GM_TILE));
    hipLaunchKernelGGL(gemm_f64_kernel, grid, dim3(256), 0, c->stream, transA ? 1 : 0, transB ? 1 : 0,
                       M, Nn, K, alpha, A, lda, B, ldb, beta, C, ldc);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// ---- Cholesky ------------------------------------------------------------------------
constexpr int CH_NB = 64;

// factor the nb x nb diagonal block at A (lower, in place); one workgroup of 64 threads
__global__ __launch_bounds__(64)
void potrf_diag_kernel(double* __restrict__ A, i64 lda, int nb, int* __restrict__ info, int col0)
{
    __shared__ double L[CH_NB][CH_NB + 1];
    const int t = threadIdx.x;
    for (int i = 0; i < nb; ++i) if (t < nb) L[i][t] = (t <= i) ? A[(i64)i * lda + t] : 0.0;
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        if (t == 0) {
            double d = L[j][j];
            for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
            if (!(d > 0.0)) { if (*info == 0) *info = col0 + j + 1; d = NAN; }
            L[j][j] = sqrt(d);
        }
        __syncthreads();
        if (t > j && t < nb) {
            double s = L[t][j];
            for (int k = 0; k < j; ++k) s -= L[t][k] * L[j][k];
            L[t][j] = s / L[j][j];
        }
        __syncthreads();
    }
    for (int i = 0; i < nb; ++i) if (t < nb && t <= i) A[(i64)i * lda + t] = L[i][t];
}

// rows below the diagonal block: P <- P * Ljj^{-T}  (one thread per row)
__global__ __launch_bounds__(256)
void trsm_panel_kernel(const double* __restrict__ Ljj, double* __restrict__ P, i64 lda, int nb, i64 rows)
{
    __shared__ double L[CH_NB][CH_NB + 1];
    for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
        const int i = e / nb, j = e % nb;
        L[i][j] = (j <= i) ? Ljj[(i64)i * lda + j] : 0.0;
    }
    __syncthreads();
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double* row = P + r * lda;
    double x[CH_NB];
#pragma unroll 4
    for (int j = 0; j < nb; ++j) {
        double s = row[j];
        for (int k = 0; k < j; ++k) s -= x[k] * L[j][k];
        x[j] = s / L[j][j];
    }
    for (int j = 0; j < nb; ++j) row[j] = x[j];
}

int launch_potrf_lower(lrvb_ctx* c, double* A, i64 n, i64 lda, int* info_dev) {
    HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), c->stream));
    for (i64 j0 = 0; j0 < n; j0 += CH_NB) {
        const int nb = (int)((n - j0 < CH_NB) ? (n - j0) : CH_NB);
        double* Ajj = A + j0 * lda + j0;
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(64), 0, c->stream, Ajj, lda, nb, info_dev, (int)j0);
        HIP_TRY(hipGetLastError());
        const i64 rows = n - j0 - nb;
        if (rows > 0) {
            double* Pn = A + (j0 + nb) * lda + j0;
            hipLaunchKernelGGL(trsm_panel_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream,
                               Ajj, Pn, lda, nb, rows);
            HIP_TRY(hipGetLastError());
            // trailing update A22 -= P P^T (full square; only the lower part is read later)
            double* A22 = A + (j0 + nb) * lda + (j0 + nb);
            LRVB_TRY(launch_gemm(c, false, true, rows, rows, nb, -1.0, Pn, lda, Pn, lda, 1.0, A22, lda));
        }
    }
    return LRVB_OK;
}

// X_j <- Ljj^{-1} B_j (forward) or Ljj^{-T} B_j (backward); one thread per right-hand side
__global__ __launch_bounds__(256)
void trsv_block_kernel(const double* __restrict__ Ljj, i64 ldl, int nb, double* __restrict__ Bj, i64 ldb,
                       i64 nrhs, int backward)
{
    __shared__ double L[CH_NB][CH_NB + 1];
    for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
        const int i = e / nb, j = e % nb;
        L[i][j] = (j <= i) ? Ljj[(i64)i * ldl + j] : 0.0;
    }
    __syncthreads();
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nrhs) return;
    double x[CH_NB];
    if (!backward) {
        for (int i = 0; i < nb; ++i) {
            double s = Bj[(i64)i * ldb + q];
            for (int k = 0; k < i; ++k) s -= L[i][k] * x[k];
            x[i] = s / L[i][i];
        }
    } else {
        for (int i = nb - 1; i >= 0; --i) {
            double s = Bj[(i64)i * ldb + q];
            for (int k = i + 1; k < nb; ++k) s -= L[k][i] * x[k];
            x[i] = s / L[i][i];
        }
    }
    for (int i = 0; i < nb; ++i) Bj[(i64)i * ldb + q] = x[i];
}

int launch_potrs_lower(lrvb_ctx* c, const double* L, i64 n, i64 ldl, double* B, i64 nrhs, i64 ldb) {
    const unsigned gq = (unsigned)((nrhs + 255) / 256);
    // forward: L Y = B
    for (i64 j0 = 0; j0 < n; j0 += CH_NB) {
        const int nb = (int)((n - j0 < CH_NB) ? (n - j0) : CH_NB);
        hipLaunchKernelGGL(trsv_block_kernel, dim3(gq), dim3(256), 0, c->stream,
                           L + j0 * ldl + j0, ldl, nb, B + j0 * ldb, ldb, nrhs, 0);
        HIP_TRY(hipGetLastError());
        const i64 rows = n - j0 - nb;
        if (rows > 0)
            LRVB_TRY(launch_gemm(c, false, false, rows, nrhs, nb, -1.0, L + (j0 + nb) * ldl + j0, ldl,
                                 B + j0 * ldb, ldb, 1.0, B + (j0 + nb) * ldb, ldb));
    }
    // backward: L^T X = Y
    i64 last = ((n - 1) / CH_NB) * CH_NB;
    for (i64 j0 = last; j0 >= 0; j0 -= CH_NB) {
        const int nb = (int)((n - j0 < CH_NB) ? (n - j0) : CH_NB);
        hipLaunchKernelGGL(trsv_block_kernel, dim3(gq), dim3(256), 0, c->stream,
                           L + j0 * ldl + j0, ldl, nb, B + j0 * ldb, ldb, nrhs, 1);
        HIP_TRY(hipGetLastError());
        if (j0 > 0)
            LRVB_TRY(launch_gemm(c, true, false, j0, nrhs, nb, -1.0, L + j0 * ldl, ldl,
                                 B + j0 * ldb, ldb, 1.0, B, ldb));
    }
    return LRVB_OK;
}

// ---- vector kernels --------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void dot_kernel(const double* __restrict__ a, const double* __restrict__ b, i64 n, double* __restrict__ out)
{
    __shared__ double sh[1024];
    double s = 0.0;
    for (i64 i = threadIdx.x; i < n; i += 1024) s += a[i] * b[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}
int launch_dot(lrvb_ctx* c, const double* a, const double* b, i64 n, double* out_dev) {
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, c->stream, a, b, n, out_dev);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

__global__ void axpby_kernel(i64 n, double alpha, const double* __restrict__ x, double beta, double* __restrict__ y)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = alpha * x[i] + (beta == 0.0 ? 0.0 : beta * y[i]);
}
int launch_axpby(lrvb_ctx* c, i64 n, double alpha, const double* x, double beta, double* y) {
    if (n <= 0) return LRVB_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, alpha, x, beta, y);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// y = alpha * op(A) x + beta * y; A is M x N row-major. One wave per output element.
__global__ __launch_bounds__(256)
void gemv_kernel(int trans, i64 M, i64 N, double alpha, const double* __restrict__ A, i64 lda,
                 const double* __restrict__ x, double beta, double* __restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const i64 o = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const i64 nout = trans ? N : M;
    if (o >= nout) return;
    double s = 0.0;
    if (!trans) { for (i64 k = lane; k < N; k += 64) s += A[o * lda + k] * x[k]; }
    else        { for (i64 k = lane; k < M; k += 64) s += A[k * lda + o] * x[k]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) y[o] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[o]);
}
int launch_gemv(lrvb_ctx* c, bool trans, i64 M, i64 Nn, double alpha, const double* A, i64 lda,
                const double* x, double beta, double* y) {
    const i64 nout = trans ? Nn : M;
    if (nout <= 0) return LRVB_OK;
    hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)((nout + 3) / 4)), dim3(256), 0, c->stream,
                       trans ? 1 : 0, M, Nn, alpha, A, lda, x, beta, y);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
