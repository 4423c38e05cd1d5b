// k_hyper.hip -- closed forms of the declared objective's derivatives with respect to its HYPER-parameters
//
//     f(eta; w, b, m, A, s, tau) = sum_n w_n l(y_n, x_n . beta; tau) + s (1/2 (eta - m)^T A (eta - m) + b^T eta)
//
// The reference forms d2 f / d theta d eps^T by `jacobian(grad_1, argnum = 2nd)` of an autograd closure
// (LRVB/SparseObjectives.py:333-339, 389-449) and d f / d eps by `grad(..., argnum = 1)` (:381-387), for ANY hyper_par
// (LRVB/ModelSensitivity.py:555-612).  For the declared objective every one of them is a closed form in r = eta - m,
// A r, b and the data gradient, written here in VECTOR coordinates of theta (V x Ph) with the diagonal packing
// Jacobian of all-box layouts fused in; general layouts multiply by the dense J afterwards (lrvb_api.hip).
#include "lrvb_internal.h"

// vech index c -> (i, j), j <= i, row-major lower triangle (SymIndex of LRVB/MatrixParameters.py:16-23)
__device__ __forceinline__ void vech_ij(i64 cidx, i64& i, i64& j) {
    i64 a = (i64)((sqrt(8.0 * (double)cidx + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > cidx) --a;
    while ((a + 1) * (a + 2) / 2 <= cidx) ++a;
    i = a; j = cidx - a * (a + 1) / 2;
}

// Cv[k, c] = d2 f / d eta_k d eps_c, scaled by j1[k] when the layout is all-box and the input is free
__global__ __launch_bounds__(256)
void hyper_cross_kernel(i64 total, i64 V, i64 Ph, int kind, int quad_kind, double scale, const double* __restrict__ A,
                        const double* __restrict__ r, const double* __restrict__ col, const double* __restrict__ j1,
                        double* __restrict__ Cv)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const i64 k = e / Ph, cidx = e - k * Ph;
    double v = 0.0;
    switch (kind) {
    case LRVB_HYPER_TILT:                                   // d/db of s b^T eta
        v = (k == cidx) ? scale : 0.0; break;
    case LRVB_HYPER_QUAD_M:                                 // d/dm of s A (eta - m)
        v = (quad_kind == LRVB_QUAD_DENSE) ? -scale * A[k * V + cidx] : ((k == cidx) ? -scale * A[k] : 0.0); break;
    case LRVB_HYPER_QUAD_A:
        if (quad_kind == LRVB_QUAD_DIAG) { v = (k == cidx) ? scale * r[k] : 0.0; }
        else {                                              // hyper = vech(A): f = 1/2 sum_i A_ii r_i^2 + sum_{i>j} A_ij r_i r_j
            i64 i, j; vech_ij(cidx, i, j);
            if (k == i) v += scale * r[j];
            if (k == j && i != j) v += scale * r[i];
        }
        break;
    default:                                                // one column: A r + b (scale), data gradient / tau (lik_info)
        v = col[k]; break;
    }
    if (j1) v *= j1[k];
    Cv[e] = v;
}

// g[c] = d f / d eps_c for the vector-valued hyper-parameters
__global__ __launch_bounds__(256)
void hyper_grad_kernel(i64 Ph, i64 V, int kind, int quad_kind, double scale, const double* __restrict__ eta,
                       const double* __restrict__ r, const double* __restrict__ Ar, double* __restrict__ g)
{
    const i64 cidx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (cidx >= Ph) return;
    double v = 0.0;
    if (kind == LRVB_HYPER_TILT) v = scale * eta[cidx];
    else if (kind == LRVB_HYPER_QUAD_M) v = -scale * Ar[cidx];
    else if (quad_kind == LRVB_QUAD_DIAG) v = 0.5 * scale * r[cidx] * r[cidx];
    else { i64 i, j; vech_ij(cidx, i, j); v = (i == j) ? 0.5 * scale * r[i] * r[i] : scale * r[i] * r[j]; }
    g[cidx] = v;
}

__global__ void hyper_col_kernel(i64 V, double a, const double* __restrict__ x, double bcoef, const double* __restrict__ y,
                                 double* __restrict__ out)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) out[i] = a * x[i] + (y ? bcoef * y[i] : 0.0);
}

int launch_hyper_cross(lrvb_ctx* c, int kind, i64 Ph, const double* r, const double* col, const double* j1, double* Cv) {
    const i64 total = c->V * Ph;
    if (total <= 0) return LRVB_OK;
    hipLaunchKernelGGL(hyper_cross_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, total, c->V, Ph, kind,
                       c->quad_kind, c->quad_scale, (const double*)c->quadA.p, r, col, j1, Cv);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

int launch_hyper_grad(lrvb_ctx* c, int kind, i64 Ph, const double* eta, const double* r, const double* Ar, double* g) {
    hipLaunchKernelGGL(hyper_grad_kernel, dim3((unsigned)((Ph + 255) / 256)), dim3(256), 0, c->stream, Ph, c->V, kind,
                       c->quad_kind, c->quad_scale, eta, r, Ar, g);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// out = a x + bcoef y  (y nullable)
int launch_hyper_col(lrvb_ctx* c, double a, const double* x, double bcoef, const double* y, double* out) {
    hipLaunchKernelGGL(hyper_col_kernel, dim3((unsigned)((c->V + 255) / 256)), dim3(256), 0, c->stream, c->V, a, x, bcoef, y, out);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
