// k_elementwise.hip -- small element-wise / row-wise kernels of the orchestration (copies, scalings, row dots, fixed-order
// partial sums).  Launched from lrvb_api.hip; prototypes in k_kernels.h.
#include "lrvb_internal.h"
#include "k_kernels.h"
#include <math.h>

// small upload: the device reads the pinned slot itself (a kernel in stream order; a copy-engine transfer would put a
// cross-queue dependency in front of the next kernel -- measured: configuration 2's step 0.66 -> 1.36 ms)
__global__ void upload_kernel(double* __restrict__ dst, const double* __restrict__ slot, i64 n) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] = slot[e];
}

__global__ void mul_kernel(i64 n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] * b[i];
}

__global__ void fma3_kernel(i64 n, const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ v,
                            const double* __restrict__ j1, const double* __restrict__ he, double* __restrict__ o) {
    // o = j1 * he + a * b * v     (box HVP epilogue: j1 (H_eta u) + g_eta eta'' v)
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = j1[i] * he[i] + a[i] * b[i] * v[i];
}

__global__ void square_kernel(i64 n, const double* __restrict__ a, double* __restrict__ o) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] * a[i];
}

__global__ void transpose_kernel(i64 rows, i64 cols, const double* __restrict__ a, double* __restrict__ o) {
    // o (cols x rows) = a^T (a is rows x cols)
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = blockIdx.y;
    if (j < cols && i < rows) o[j * rows + i] = a[i * cols + j];
}

__global__ void diag_scale_kernel(i64 D, i64 V, double scale, const double* __restrict__ j1, double* __restrict__ C) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = blockIdx.y;
    if (j < V && i < D) C[i * V + j] = (i == j) ? scale * j1[i] : 0.0;
}

__global__ void symmetrize_lower_kernel(i64 n, double* __restrict__ A, i64 ld) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = blockIdx.y;
    if (j < n && i < n && j > i) A[i * ld + j] = A[j * ld + i];
}

// Every free-coordinate build leaves a copy of its result with the library (8 MB at D = 1024: a ~4 us device copy beside a
// 15 ms build).  A product H v asked for at the SAME point afterwards -- lrvb_hvp, every iteration of lrvb_cg_solve and
// lrvb_cg_solve_multi, i.e. ConjugateGradientSolver (LRVB/ConjugateGradient.py:63-105) after fun_free_hessian at the optimum --
// is then a D x D matrix product instead of a pass over the N x P design (1.3-1.5 ms at the headline shape).  The copy is
// dropped whenever data, weights, a hyper-parameter, the reduce hook or the tuning change (the setters), and is only used
// for the exact point it was built at.  Adopted device buffers fall under the contract of lrvb_set_data_dev /
// lrvb_set_weights_dev: install them again after their contents change.
__global__ void vec_differs_kernel(i64 n, const double* __restrict__ a, const double* __restrict__ b, int* __restrict__ flag) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(a[i] == b[i])) atomicOr(flag, 1);
}

__global__ void fill_kernel(i64 n, double v, double* __restrict__ o) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = v;
}

// part[b] = sum of v over block b's contiguous slice (fixed tree); the partials are summed by sum_partials_kernel
__global__ __launch_bounds__(256)
void vec_block_sums_kernel(i64 n, const double* __restrict__ v, double* __restrict__ part) {
    __shared__ double sh[256];
    const i64 per = (n + gridDim.x - 1) / gridDim.x;
    const i64 a = (i64)blockIdx.x * per, b = a + per < n ? a + per : n;
    double s = 0.0;
    for (i64 i = a + threadIdx.x; i < b; i += 256) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// out[n - n0, q] = -(G H^-1 M^T)[n, q] = d (moment q) / d w_n  by linear response, for rows n0..n1 of G.
// G = diag(l') X J_glm is never formed: W = H^-1 M^T (D x Q) from the resident Cholesky factor,
// Z = J_glm W (P x Q), and the rows of X are multiplied by Z in one pass, scaled by -l'_n.
__global__ void row_scale_rows_kernel(i64 total, i64 Q, const double* __restrict__ rowscale, double alpha, double* __restrict__ C) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;      // total = rows * Q (the EW launcher passes the element count first)
    if (e < total) C[e] *= alpha * rowscale[e / Q];
}

__global__ void scale_slice_rows_kernel(i64 total, i64 Q, const double* __restrict__ j1, const double* __restrict__ W, double* __restrict__ Z) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;      // total = P * Q
    if (e < total) Z[e] = (j1 ? j1[e / Q] : 1.0) * W[e];
}

// out[0] = sum of part[0 .. n) in a fixed order (one workgroup: strided partial sums, then a tree)
__global__ __launch_bounds__(256)
void sum_partials_kernel(const double* __restrict__ part, i64 n, double* __restrict__ out) {
    __shared__ double sh[256];
    double a = 0.0;
    for (i64 i = threadIdx.x; i < n; i += 256) a += part[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

__global__ void rowscale_kernel(i64 n, i64 P, const double* __restrict__ cvec, const double* __restrict__ B, double* __restrict__ o) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = cvec[i / P] * B[i];
}

// Q independent CG recurrences (the ones `ConjugateGradientSolver.get_hinv_vec_subsets` runs one after the
// other, LRVB/ConjugateGradient.py:87-105) advance in lockstep, so that the Hessian-vector products of an
// iteration become ONE pair of skinny MFMA GEMMs over X -- T = X U_glm^T (N x Q), then X^T diag(c) T (P x Q)
// -- instead of Q fused passes.  Block vectors are Q x D row-major (one right-hand side per row).
__global__ void mul_rows_kernel(i64 n, i64 D, const double* __restrict__ a, const double* __restrict__ v, double* __restrict__ o) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) o[e] = a[e % D] * v[e];
}

__global__ void fma3_rows_kernel(i64 n, i64 D, const double* __restrict__ g, const double* __restrict__ j2, const double* __restrict__ v,
                                 const double* __restrict__ j1, const double* __restrict__ he, double* __restrict__ o) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) { const i64 d = e % D; o[e] = j1[d] * he[e] + g[d] * j2[d] * v[e]; }
}

__global__ void diag_mul_add_rows_kernel(i64 n, i64 V, double scale, const double* __restrict__ a, const double* __restrict__ u, double* __restrict__ out) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] += scale * a[e % V] * u[e];
}

__global__ void scatter_rows_T_kernel(i64 n, i64 Q, i64 V, i64 P, i64 off, const double* __restrict__ Rt, i64 ldr, double* __restrict__ out) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;        // e over Q x P
    if (e < n) { const i64 q = e / P, p = e - q * P; out[q * V + off + p] = Rt[p * ldr + q]; }
}

// per-row scalars of the block recurrences
__global__ void rows_dot_kernel(i64 Q, i64 D, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
    __shared__ double sh[256];
    const i64 q = blockIdx.x;
    double s = 0.0;
    for (i64 d = threadIdx.x; d < D; d += 256) s += a[q * D + d] * b[q * D + d];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) out[q] = sh[0];
}

__global__ void rows_axpby_kernel(i64 n, i64 D, const double* __restrict__ alpha, const double* __restrict__ x,
                                  const double* __restrict__ beta, double* __restrict__ y) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;        // y = alpha[q] x + beta[q] y
    if (e < n) { const i64 q = e / D; y[e] = alpha[q] * x[e] + beta[q] * y[e]; }
}
