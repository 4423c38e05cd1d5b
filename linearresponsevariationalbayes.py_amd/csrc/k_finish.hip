// k_finish.hip -- N-independent assembly after the observation sums:
//   quadratic term  quad_scale * (1/2 (eta-m)^T A (eta-m) + b^T eta)   (value, gradient, HVP)
//   H_free = J^T H_eta J + sum_k g_k d2eta_k   -- convert_vector_to_free_hessian,
//   LRVB/Parameters.py:397-424.  For all-box layouts J is diagonal and the whole conversion is
//   one elementwise kernel reading the tile-packed weighted-SYRK output.
#include "lrvb_internal.h"

// d = eta - m ; Ad = A d (diag case) -- dense case uses gemv for Ad
__global__ void quad_diff_kernel(i64 V, const double* __restrict__ eta, const double* __restrict__ m,
                                 const double* __restrict__ a_diag, double* __restrict__ d, double* __restrict__ Ad)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const double di = eta[i] - (m ? m[i] : 0.0);
    d[i] = di;
    if (a_diag) Ad[i] = a_diag[i] * di;
}

// g_eta += scale (Ad + b);  value += scale (1/2 d.Ad + b.eta)   (single workgroup)
__global__ __launch_bounds__(1024)
void quad_finish_kernel(i64 V, double scale, const double* __restrict__ eta, const double* __restrict__ d,
                        const double* __restrict__ Ad, const double* __restrict__ b,
                        double* __restrict__ g_eta, double* __restrict__ value)
{
    __shared__ double sh[1024];
    double s = 0.0;
    for (i64 i = threadIdx.x; i < V; i += 1024) {
        const double bi = b ? b[i] : 0.0;
        if (g_eta) g_eta[i] += scale * (Ad[i] + bi);
        s += 0.5 * d[i] * Ad[i] + bi * eta[i];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && value) *value += scale * sh[0];
}

// vtmp = r = eta - m, vtmp2 = A r
int launch_quad_diff(lrvb_ctx* c, const double* eta_dev) {
    if (c->quad_kind == LRVB_QUAD_NONE) return LRVB_OK;
    const i64 V = c->V;
    LRVB_TRY(buf_reserve(c, c->vtmp, (size_t)V));
    LRVB_TRY(buf_reserve(c, c->vtmp2, (size_t)V));
    const double* a_diag = (c->quad_kind == LRVB_QUAD_DIAG) ? c->quadA.p : nullptr;
    hipLaunchKernelGGL(quad_diff_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, c->stream,
                       V, eta_dev, c->quadM.p, a_diag, c->vtmp.p, c->vtmp2.p);
    HIP_TRY(hipGetLastError());
    if (c->quad_kind == LRVB_QUAD_DENSE)
        LRVB_TRY(launch_gemv(c, false, V, V, 1.0, c->quadA.p, V, c->vtmp.p, 0.0, c->vtmp2.p));
    return LRVB_OK;
}

int launch_quad_grad_value(lrvb_ctx* c, const double* eta_dev, double* g_eta_dev, double* value_dev) {
    if (c->quad_kind == LRVB_QUAD_NONE) return LRVB_OK;
    const i64 V = c->V;
    LRVB_TRY(launch_quad_diff(c, eta_dev));
    hipLaunchKernelGGL(quad_finish_kernel, dim3(1), dim3(1024), 0, c->stream, V, c->quad_scale, eta_dev,
                       c->vtmp.p, c->vtmp2.p, c->quadB.p, g_eta_dev, value_dev);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

__global__ void diag_mul_add_kernel(i64 V, double scale, const double* __restrict__ a, const double* __restrict__ u,
                                    double* __restrict__ out)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) out[i] += scale * a[i] * u[i];
}

int launch_quad_hvp(lrvb_ctx* c, const double* u_vec_V, double* out_vec_V) {
    if (c->quad_kind == LRVB_QUAD_NONE) return LRVB_OK;
    const i64 V = c->V;
    if (c->quad_kind == LRVB_QUAD_DIAG) {
        hipLaunchKernelGGL(diag_mul_add_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, c->stream,
                           V, c->quad_scale, c->quadA.p, u_vec_V, out_vec_V);
        HIP_TRY(hipGetLastError());
        return LRVB_OK;
    }
    return launch_gemv(c, false, V, V, c->quad_scale, c->quadA.p, V, u_vec_V, 1.0, out_vec_V);
}

__device__ __forceinline__ double tile_lookup(const double* __restrict__ tiles, i64 i, i64 j) {
    const i64 a = i > j ? i : j, b = i > j ? j : i;
    const i64 ba = a / WS_TILE, bb = b / WS_TILE;
    const i64 t = ba * (ba + 1) / 2 + bb;
    return tiles[t * (WS_TILE * WS_TILE) + (a % WS_TILE) * WS_TILE + (b % WS_TILE)];
}

// all-box layouts (D == V, free index == vector index):
// H[i][j] = j1_i j1_j ( S[i-off][j-off] + scale A_ij ) + d_ij g_i j2_i
__global__ __launch_bounds__(256)
void finish_box_kernel(i64 D, const double* __restrict__ tiles, i64 P, i64 glm_off,
                       int quad_kind, double scale, const double* __restrict__ A,
                       const double* __restrict__ g_eta, const double* __restrict__ j1,
                       const double* __restrict__ j2, int with_third, double* __restrict__ H, i64 ld)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = blockIdx.y;
    if (j >= D) return;
    double h = 0.0;
    const i64 ic = i - glm_off, jc = j - glm_off;
    if (tiles != nullptr && ic >= 0 && ic < P && jc >= 0 && jc < P) h = tile_lookup(tiles, ic, jc);
    if (quad_kind == LRVB_QUAD_DENSE) h += scale * A[i * D + j];
    else if (quad_kind == LRVB_QUAD_DIAG && i == j) h += scale * A[i];
    if (j1) h *= j1[i] * j1[j];
    if (with_third && i == j) h += g_eta[i] * j2[i];
    H[i * ld + j] = h;
}

int launch_finish_box(lrvb_ctx* c, const double* tiles_dev, const double* g_eta_dev,
                      const double* j1, const double* j2, bool with_third, double* H_dev, i64 ld) {
    dim3 grid((unsigned)((c->D + 255) / 256), (unsigned)c->D);
    hipLaunchKernelGGL(finish_box_kernel, grid, dim3(256), 0, c->stream, c->D, tiles_dev, c->P, c->glm_off,
                       c->quad_kind, c->quad_scale, c->quadA.p, g_eta_dev, j1, j2, with_third ? 1 : 0, H_dev, ld);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// general layouts: dense H_eta (V x V) = scatter(S) + scale A
__global__ __launch_bounds__(256)
void build_heta_kernel(i64 V, const double* __restrict__ tiles, i64 P, i64 glm_off, int quad_kind,
                       double scale, const double* __restrict__ A, double* __restrict__ Heta)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = blockIdx.y;
    if (j >= V) return;
    double h = 0.0;
    const i64 ic = i - glm_off, jc = j - glm_off;
    if (tiles != nullptr && ic >= 0 && ic < P && jc >= 0 && jc < P) h = tile_lookup(tiles, ic, jc);
    if (quad_kind == LRVB_QUAD_DENSE) h += scale * A[i * V + j];
    else if (quad_kind == LRVB_QUAD_DIAG && i == j) h += scale * A[i];
    Heta[i * V + j] = h;
}

int launch_build_Heta(lrvb_ctx* c, const double* tiles_dev, double* Heta_dev) {
    dim3 grid((unsigned)((c->V + 255) / 256), (unsigned)c->V);
    hipLaunchKernelGGL(build_heta_kernel, grid, dim3(256), 0, c->stream, c->V, tiles_dev, c->P, c->glm_off,
                       c->quad_kind, c->quad_scale, c->quadA.p, Heta_dev);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

__global__ void scatter_glm_kernel(i64 V, i64 P, i64 glm_off, const double* __restrict__ g_glm,
                                   double* __restrict__ g_eta)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const i64 ic = i - glm_off;
    g_eta[i] = (g_glm != nullptr && ic >= 0 && ic < P) ? g_glm[ic] : 0.0;
}

int launch_scatter_glm(lrvb_ctx* c, const double* g_glm_P, double* g_eta_V) {
    hipLaunchKernelGGL(scatter_glm_kernel, dim3((unsigned)((c->V + 255) / 256)), dim3(256), 0, c->stream,
                       c->V, c->P, c->glm_off, g_glm_P, g_eta_V);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
