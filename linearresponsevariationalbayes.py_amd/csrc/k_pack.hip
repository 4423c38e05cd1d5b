// k_pack.hip -- the free <-> vector packing maps and their first/second derivatives, on device.
//
// Reference semantics (formulas restated, not code):
//   box      LRVB/Parameters.py:31-61       eta = f | exp(f)+lb | ub-exp(-f) | (ub-lb) s(f)+lb
//   psd      LRVB/MatrixParameters.py:16-23, 101-112
//            f (row-major lower triangle, idx(a,b) = b + a(a+1)/2) -> L (exp on the diagonal)
//            -> A = L L^T + diag_lb I; eta = lower triangle of A in the same order
//   simplex  LRVB/SimplexParams.py:11-23    p = softmax([0, f_1..f_{K-1}]) per row
// The reference gets the Jacobians/Hessians of box and psd blocks from autograd
// (Parameters.py:63-64, MatrixParameters.py:132-135) and of simplex rows from closed forms
// (SimplexParams.py:33-63); here all three are closed forms (derivations in DESIGN.md).
#include "lrvb_internal.h"
#include <cstring>
#include <math.h>

__device__ __forceinline__ i64 ld_idx(i64 a, i64 b) { return b + a * (a + 1) / 2; }   // a >= b

// ---- box ----------------------------------------------------------------------------
__device__ __forceinline__ void box_eval(double f, double lb, double ub, double& e, double& d1, double& d2) {
    const bool has_lb = lb > -INFINITY, has_ub = ub < INFINITY;
    if (!has_lb && !has_ub) { e = f; d1 = 1.0; d2 = 0.0; }
    else if (has_lb && !has_ub) { const double x = exp(f); e = x + lb; d1 = x; d2 = x; }
    else if (!has_lb && has_ub) { const double x = exp(-f); e = ub - x; d1 = x; d2 = -x; }
    else {
        // stable logistic; the reference's exp(f)/(1+exp(f)) overflows for f > 709
        const double ef = exp(-fabs(f));
        const double s = f >= 0.0 ? 1.0 / (1.0 + ef) : ef / (1.0 + ef);
        const double r = ub - lb, sp = s * (1.0 - s);
        e = r * s + lb; d1 = r * sp; d2 = r * sp * (1.0 - 2.0 * s);
    }
}

__global__ void box_constrain_kernel(const double* __restrict__ theta, i64 free_off, i64 vec_off, i64 n,
                                     double lb, double ub, double* __restrict__ eta,
                                     double* __restrict__ j1, double* __restrict__ j2)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double e, d1, d2;
    box_eval(theta[free_off + i], lb, ub, e, d1, d2);
    eta[vec_off + i] = e;
    if (j1) j1[free_off + i] = d1;
    if (j2) j2[free_off + i] = d2;
}

__global__ void box_unconstrain_kernel(const double* __restrict__ eta, i64 free_off, i64 vec_off, i64 n,
                                       double lb, double ub, double* __restrict__ theta, int* __restrict__ bad)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = eta[vec_off + i];
    if (!(v <= ub) || !(v >= lb)) { atomicOr(bad, 1); }
    const bool has_lb = lb > -INFINITY, has_ub = ub < INFINITY;
    double f;
    if (!has_lb && !has_ub) f = v;
    else if (has_lb && !has_ub) f = log(v - lb);
    else if (!has_lb && has_ub) f = -log(ub - v);
    else f = log(v - lb) - log(ub - v);
    theta[free_off + i] = f;
}

// ---- psd ----------------------------------------------------------------------------
__device__ __forceinline__ double psd_L(const double* __restrict__ f, i64 a, i64 b) {
    if (b > a) return 0.0;
    const double v = f[ld_idx(a, b)];
    return a == b ? exp(v) : v;
}

__global__ void psd_constrain_kernel(const double* __restrict__ theta, i64 free_off, i64 vec_off, i64 k,
                                     double diag_lb, double* __restrict__ eta)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 m = k * (k + 1) / 2;
    if (e >= m) return;
    i64 i = (i64)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    while (i * (i + 1) / 2 > e) --i;
    const i64 j = e - i * (i + 1) / 2;
    const double* f = theta + free_off;
    double s = 0.0;
    for (i64 c = 0; c <= j; ++c) s += psd_L(f, i, c) * psd_L(f, j, c);
    if (i == j) s += diag_lb;
    eta[vec_off + e] = s;
}

// one workgroup: Cholesky of (A - diag_lb I) in LDS-free global scratch (k is small), log-diagonal
__global__ void psd_unconstrain_kernel(const double* __restrict__ eta, i64 free_off, i64 vec_off, i64 k,
                                       double diag_lb, double* __restrict__ theta, int* __restrict__ bad)
{
    // serial left-looking Cholesky by one thread per column element, synchronised per column
    double* f = theta + free_off;
    const double* a = eta + vec_off;
    const int tid = threadIdx.x;
    for (i64 j = 0; j < k; ++j) {
        // diagonal
        if (tid == 0) {
            double d = a[ld_idx(j, j)] - diag_lb;
            for (i64 c = 0; c < j; ++c) { const double l = f[ld_idx(j, c)]; d -= l * l; }
            if (!(d > 0.0)) { atomicOr(bad, 2); d = NAN; }
            f[ld_idx(j, j)] = sqrt(d);
        }
        __syncthreads();
        const double ljj = f[ld_idx(j, j)];
        for (i64 i = j + 1 + tid; i < k; i += blockDim.x) {
            double s = a[ld_idx(i, j)];
            for (i64 c = 0; c < j; ++c) s -= f[ld_idx(i, c)] * f[ld_idx(j, c)];
            f[ld_idx(i, j)] = s / ljj;
        }
        __syncthreads();
    }
    for (i64 j = tid; j < k; j += blockDim.x) f[ld_idx(j, j)] = log(f[ld_idx(j, j)]);
}

// dense Jacobian block: J[(i,j),(a,b)] = dL_ab (d_ia L_jb + d_ja L_ib)
__global__ void psd_jac_kernel(const double* __restrict__ theta, i64 free_off, i64 vec_off, i64 k,
                               double* __restrict__ J, i64 ldj)
{
    const i64 m = k * (k + 1) / 2;
    const i64 col = (i64)blockIdx.x * blockDim.x + threadIdx.x;   // (a,b)
    const i64 row = blockIdx.y;                                   // (i,j)
    if (col >= m || row >= m) return;
    i64 i = (i64)((sqrt(8.0 * (double)row + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= row) ++i;
    while (i * (i + 1) / 2 > row) --i;
    const i64 j = row - i * (i + 1) / 2;
    i64 a = (i64)((sqrt(8.0 * (double)col + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= col) ++a;
    while (a * (a + 1) / 2 > col) --a;
    const i64 b = col - a * (a + 1) / 2;
    const double* f = theta + free_off;
    const double dL = (a == b) ? exp(f[ld_idx(a, a)]) : 1.0;
    double v = 0.0;
    if (i == a) v += psd_L(f, j, b);
    if (j == a) v += psd_L(f, i, b);
    J[(vec_off + row) * ldj + free_off + col] = dL * v;
}

// third-order block: T[(a,b),(c,d)] = d_bd dL_ab dL_cd Gs_ac (1 + d_ac)
//                                   + d_(ab)(cd) d_ab L_aa (2 g_aa L_aa + sum_{i>a} g_ia L_ia)
__global__ void psd_third_kernel(const double* __restrict__ theta, const double* __restrict__ g_eta,
                                 i64 free_off, i64 vec_off, i64 k, double* __restrict__ T, i64 ldt)
{
    const i64 m = k * (k + 1) / 2;
    const i64 col = (i64)blockIdx.x * blockDim.x + threadIdx.x;   // (c,d)
    const i64 row = blockIdx.y;                                   // (a,b)
    if (col >= m || row >= m) return;
    i64 a = (i64)((sqrt(8.0 * (double)row + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= row) ++a;
    while (a * (a + 1) / 2 > row) --a;
    const i64 b = row - a * (a + 1) / 2;
    i64 cc = (i64)((sqrt(8.0 * (double)col + 1.0) - 1.0) * 0.5);
    while ((cc + 1) * (cc + 2) / 2 <= col) ++cc;
    while (cc * (cc + 1) / 2 > col) --cc;
    const i64 d = col - cc * (cc + 1) / 2;
    const double* f = theta + free_off;
    const double* g = g_eta + vec_off;
    double v = 0.0;
    if (b == d) {
        const double dLab = (a == b) ? exp(f[ld_idx(a, a)]) : 1.0;
        const double dLcd = (cc == d) ? exp(f[ld_idx(cc, cc)]) : 1.0;
        const double gs = (a >= cc) ? g[ld_idx(a, cc)] : g[ld_idx(cc, a)];
        v += dLab * dLcd * gs * ((a == cc) ? 2.0 : 1.0);
    }
    if (row == col && a == b) {
        const double laa = exp(f[ld_idx(a, a)]);
        double s = 2.0 * g[ld_idx(a, a)] * laa;
        for (i64 i = a + 1; i < k; ++i) s += g[ld_idx(i, a)] * f[ld_idx(i, a)];
        v += laa * s;
    }
    T[(free_off + row) * ldt + free_off + col] += v;
}

// ---- simplex ------------------------------------------------------------------------
__device__ __forceinline__ void simplex_row(const double* __restrict__ f, i64 K, double& mx, double& lse) {
    mx = 0.0;                                     // the reference category has logit 0
    for (i64 j = 0; j < K - 1; ++j) mx = fmax(mx, f[j]);
    double s = exp(0.0 - mx);
    for (i64 j = 0; j < K - 1; ++j) s += exp(f[j] - mx);
    lse = mx + log(s);
}
__device__ __forceinline__ double simplex_p(const double* __restrict__ f, i64 kk, double lse) {
    return exp((kk == 0 ? 0.0 : f[kk - 1]) - lse);
}

__global__ void simplex_constrain_kernel(const double* __restrict__ theta, i64 free_off, i64 vec_off,
                                         i64 rows, i64 K, double* __restrict__ eta)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * K) return;
    const i64 r = e / K, kk = e % K;
    const double* f = theta + free_off + r * (K - 1);
    double mx, lse; simplex_row(f, K, mx, lse);
    eta[vec_off + e] = simplex_p(f, kk, lse);
}

__global__ void simplex_unconstrain_kernel(const double* __restrict__ eta, i64 free_off, i64 vec_off,
                                           i64 rows, i64 K, double* __restrict__ theta)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * (K - 1)) return;
    const i64 r = e / (K - 1), j = e % (K - 1);
    const double* p = eta + vec_off + r * K;
    theta[free_off + e] = log(p[j + 1]) - log(p[0]);
}

// J_row[k, j] = p_k (d_{k,j+1} - p_{j+1})
__global__ void simplex_jac_kernel(const double* __restrict__ theta, i64 free_off, i64 vec_off,
                                   i64 rows, i64 K, double* __restrict__ J, i64 ldj)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 per = K * (K - 1);
    if (e >= rows * per) return;
    const i64 r = e / per, rem = e % per, kk = rem / (K - 1), j = rem % (K - 1);
    const double* f = theta + free_off + r * (K - 1);
    double mx, lse; simplex_row(f, K, mx, lse);
    const double pk = simplex_p(f, kk, lse), pj = simplex_p(f, j + 1, lse);
    J[(vec_off + r * K + kk) * ldj + free_off + r * (K - 1) + j] = pk * ((kk == j + 1 ? 1.0 : 0.0) - pj);
}

// T_row[i, j] = sum_k g_k p_k [ (d_{k,i+1} - p_{i+1})(d_{k,j+1} - p_{j+1}) - p_{i+1}(d_ij - p_{j+1}) ]
__global__ void simplex_third_kernel(const double* __restrict__ theta, const double* __restrict__ g_eta,
                                     i64 free_off, i64 vec_off, i64 rows, i64 K,
                                     double* __restrict__ T, i64 ldt)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 per = (K - 1) * (K - 1);
    if (e >= rows * per) return;
    const i64 r = e / per, rem = e % per, i = rem / (K - 1), j = rem % (K - 1);
    const double* f = theta + free_off + r * (K - 1);
    const double* g = g_eta + vec_off + r * K;
    double mx, lse; simplex_row(f, K, mx, lse);
    const double pi = simplex_p(f, i + 1, lse), pj = simplex_p(f, j + 1, lse);
    double s = 0.0;
    for (i64 kk = 0; kk < K; ++kk) {
        const double pk = simplex_p(f, kk, lse);
        const double di = (kk == i + 1 ? 1.0 : 0.0) - pi;
        const double dj = (kk == j + 1 ? 1.0 : 0.0) - pj;
        s += g[kk] * pk * (di * dj - pi * ((i == j ? 1.0 : 0.0) - pj));
    }
    T[(free_off + r * (K - 1) + i) * ldt + free_off + r * (K - 1) + j] += s;
}

// ---- diagonal helpers for box blocks in the dense path ----------------------------------
__global__ void box_jac_dense_kernel(const double* __restrict__ theta, i64 free_off, i64 vec_off, i64 n,
                                     double lb, double ub, double* __restrict__ J, i64 ldj)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double e, d1, d2; box_eval(theta[free_off + i], lb, ub, e, d1, d2);
    J[(vec_off + i) * ldj + free_off + i] = d1;
}
__global__ void box_third_dense_kernel(const double* __restrict__ theta, const double* __restrict__ g_eta,
                                       i64 free_off, i64 vec_off, i64 n, double lb, double ub,
                                       double* __restrict__ T, i64 ldt)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double e, d1, d2; box_eval(theta[free_off + i], lb, ub, e, d1, d2);
    T[(free_off + i) * ldt + free_off + i] += g_eta[vec_off + i] * d2;
}

static inline unsigned nblk(i64 n, int t = 256) { return (unsigned)((n + t - 1) / t); }

// ---- all box blocks of a layout in ONE launch ---------------------------------------------------------------------------
// A layout with many small box blocks (the hierarchical model's global block: mean, five scalars, ...) paid one 4-5 us
// launch per block and map (constrain, Jacobian, second-order term: ~21 launches in the config-4 step).  The per-element
// description of every box entry -- free index, vector index, bounds -- is uploaded once with the context (boxmap), and each
// map is one kernel over all box entries.  MODE 0: eta, eta', eta''; 1: dense Jacobian entries; 2: second-order term
// T[f, f] += g[v] eta''; 3: unconstrain.
template <int MODE>
__global__ void box_all_kernel(i64 nbe, const i64* __restrict__ foff, const i64* __restrict__ voff, const double* __restrict__ lbs,
                               const double* __restrict__ ubs, const double* __restrict__ in, const double* __restrict__ g,
                               double* __restrict__ o0, double* __restrict__ o1, double* __restrict__ o2, i64 ld, int* __restrict__ bad)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbe) return;
    const i64 f = foff[i], v = voff[i];
    const double lb = lbs[i], ub = ubs[i];
    if (MODE == 3) {
        const double x = in[v];
        if (!(x <= ub) || !(x >= lb)) { atomicOr(bad, 1); }
        const bool has_lb = lb > -INFINITY, has_ub = ub < INFINITY;
        double t;
        if (!has_lb && !has_ub) t = x;
        else if (has_lb && !has_ub) t = log(x - lb);
        else if (!has_lb && has_ub) t = -log(ub - x);
        else t = log(x - lb) - log(ub - x);
        o0[f] = t;
        return;
    }
    double e, d1, d2;
    box_eval(in[f], lb, ub, e, d1, d2);
    if (MODE == 0) { o0[v] = e; if (o1) o1[f] = d1; if (o2) o2[f] = d2; }
    if (MODE == 1) o0[v * ld + f] = d1;
    if (MODE == 2) o0[f * ld + f] += g[v] * d2;
}
static bool box_fused(const lrvb_ctx* c) { return c->n_box_blocks >= 2 && c->boxmap.p != nullptr; }
template <int MODE>
static int launch_box_all(lrvb_ctx* c, const double* in, const double* g, double* o0, double* o1, double* o2, i64 ld, int* bad) {
    const i64 nbe = c->n_box_entries;
    if (nbe <= 0) return LRVB_OK;
    const i64* foff = reinterpret_cast<const i64*>(c->boxmap.p);
    const i64* voff = foff + nbe;
    const double* lbs = c->boxmap.p + 2 * nbe;
    const double* ubs = lbs + nbe;
    hipLaunchKernelGGL(box_all_kernel<MODE>, dim3(nblk(nbe)), dim3(256), 0, c->stream, nbe, foff, voff, lbs, ubs, in, g, o0, o1, o2, ld, bad);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
// called once by lrvb_ctx_create (after the blocks are known): the per-entry description of the box blocks
int upload_boxmap(lrvb_ctx* c) {
    i64 nbe = 0; int nb = 0;
    for (const auto& b : c->blocks) if (b.kind == LRVB_BLOCK_BOX && b.free_size > 0) { nbe += b.free_size; ++nb; }
    c->n_box_blocks = nb; c->n_box_entries = nbe;
    if (nb < 2) return LRVB_OK;
    std::vector<double> host((size_t)(4 * nbe));
    i64* foff = reinterpret_cast<i64*>(host.data());
    i64* voff = foff + nbe;
    double* lbs = host.data() + 2 * nbe; double* ubs = lbs + nbe;
    i64 e = 0;
    for (const auto& b : c->blocks)
        if (b.kind == LRVB_BLOCK_BOX)
            for (i64 i = 0; i < b.free_size; ++i, ++e) { foff[e] = b.free_off + i; voff[e] = b.vec_off + i; lbs[e] = b.lb; ubs[e] = b.ub; }
    LRVB_TRY(buf_reserve(c, c->boxmap, (size_t)(4 * nbe)));
    HIP_TRY(hipMemcpyAsync(c->boxmap.p, host.data(), (size_t)(4 * nbe) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRVB_OK;
}

int launch_constrain(lrvb_ctx* c, const double* theta_dev, double* eta_dev, double* j1_dev, double* j2_dev) {
    const bool fused = box_fused(c);
    if (fused) LRVB_TRY(launch_box_all<0>(c, theta_dev, nullptr, eta_dev, j1_dev, j2_dev, 0, nullptr));
    for (const auto& b : c->blocks) {
        if (b.kind == LRVB_BLOCK_BOX) {
            if (b.free_size > 0 && !fused)
                hipLaunchKernelGGL(box_constrain_kernel, dim3(nblk(b.free_size)), dim3(256), 0, c->stream,
                                   theta_dev, b.free_off, b.vec_off, b.free_size, b.lb, b.ub, eta_dev, j1_dev, j2_dev);
        } else if (b.kind == LRVB_BLOCK_PSD) {
            hipLaunchKernelGGL(psd_constrain_kernel, dim3(nblk(b.vec_size)), dim3(256), 0, c->stream,
                               theta_dev, b.free_off, b.vec_off, b.dim0, b.lb, eta_dev);
        } else {
            hipLaunchKernelGGL(simplex_constrain_kernel, dim3(nblk(b.vec_size)), dim3(256), 0, c->stream,
                               theta_dev, b.free_off, b.vec_off, b.dim0, b.dim1, eta_dev);
        }
        HIP_TRY(hipGetLastError());
    }
    return LRVB_OK;
}

int launch_unconstrain(lrvb_ctx* c, const double* eta_dev, double* theta_dev, int* bad_flag_dev) {
    const bool fused = box_fused(c);
    if (fused) LRVB_TRY(launch_box_all<3>(c, eta_dev, nullptr, theta_dev, nullptr, nullptr, 0, bad_flag_dev));
    for (const auto& b : c->blocks) {
        if (b.kind == LRVB_BLOCK_BOX) {
            if (b.free_size > 0 && !fused)
                hipLaunchKernelGGL(box_unconstrain_kernel, dim3(nblk(b.free_size)), dim3(256), 0, c->stream,
                                   eta_dev, b.free_off, b.vec_off, b.free_size, b.lb, b.ub, theta_dev, bad_flag_dev);
        } else if (b.kind == LRVB_BLOCK_PSD) {
            hipLaunchKernelGGL(psd_unconstrain_kernel, dim3(1), dim3(64), 0, c->stream,
                               eta_dev, b.free_off, b.vec_off, b.dim0, b.lb, theta_dev, bad_flag_dev);
        } else {
            hipLaunchKernelGGL(simplex_unconstrain_kernel, dim3(nblk(b.free_size)), dim3(256), 0, c->stream,
                               eta_dev, b.free_off, b.vec_off, b.dim0, b.dim1, theta_dev);
        }
        HIP_TRY(hipGetLastError());
    }
    return LRVB_OK;
}

// ---- J^T A without the dense Jacobian (layouts of box and log-Cholesky blocks) ------------------------------------------------
// The packing Jacobian is diagonal on box blocks and, on a k x k log-Cholesky block, has at most k + 1 entries per column:
//   J[(i, j), (a, b)] = dL_ab (d_ia L_jb + d_ja L_ib)            (psd_jac_kernel above; (i, j), (a, b) packed lower triangles)
// so row (a, b) of J^T A is dL_ab sum_{t >= b} L_tb G_a[t, :], where G_a gathers k rows of A: A[(a, t)] for t < a, 2 A[(a, a)],
// A[(t, a)] for t > a.  One workgroup stages G_a for 64 columns of A in LDS once and serves all a + 1 rows (a, 0..a) from it:
// k^2 rows of A are read per block instead of ~k^3 / 3 by a row-at-a-time product, and nothing like the two dense 2 V^3
// products of convert_vector_to_free_hessian (LRVB/Parameters.py:397-424): at the 995 parameters of configuration 4 (one
// 43 x 43 block) 2 x 59 us of MFMA products, reductions and unpacking become two launches of ~10 us.
// TRANS_IN: A is read transposed (A[c][v]: the second product, J^T (J^T H)^T).  Table rows (5 x i64 per blockIdx.y):
// [1, a, free_off, vec_off, k] or [0, first box entry, one past the last, -, -].
typedef double jt_d4 __attribute__((ext_vector_type(4)));
template <bool TRANS_IN, int NB>                      // NB >= ceil(k / 4): staging rows per wavefront
__global__ __launch_bounds__(256)
void jt_apply_kernel(const i64* __restrict__ rows, const i64* __restrict__ bfoff, const i64* __restrict__ bvoff,
                     const double* __restrict__ blb, const double* __restrict__ bub, const double* __restrict__ theta,
                     const double* __restrict__ A, i64 lda, i64 n, double* __restrict__ out, i64 ldo)
{
    extern __shared__ double jt_sm[];                 // G[kr][65] | L[kr][KP]   (kr = k rounded up to 4, KP to 16; zero padding)
    constexpr int GS = 65;                            // row stride of G: the transposed staging writes down a column
    const int tid = threadIdx.x, cx = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const i64 c0 = (i64)blockIdx.x * 64, c = c0 + cx;
    const i64* r = rows + 5 * (i64)blockIdx.y;
    if (r[0] == 0) {                                  // a chunk of box coordinates: a row scaling
        for (i64 e = r[1] + wave; e < r[2]; e += 4) {
            const i64 f = bfoff[e], v = bvoff[e];
            double ev, d1, d2;
            box_eval(theta[f], blb[e], bub[e], ev, d1, d2);
            if (c < n) out[f * ldo + c] = d1 * (TRANS_IN ? A[c * lda + v] : A[v * lda + c]);
        }
        return;
    }
    const int a = (int)r[1], k = (int)r[4];
    const int kr = (k + 3) & ~3, KP = (k + 15) & ~15;
    const i64 fo = r[2], vo = r[3];
    double* G = jt_sm;
    double* Ls = jt_sm + (size_t)kr * GS;
    const double* f = theta + fo;
    // Staging: every global load of the block is issued before the first one is used (fully unrolled, values parked in
    // registers).
    {
        double raw[NB], gv[16];
        const int lc = cx < k ? cx : 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {                                          // row t = wave + 4 j of L, entry lc
            const int t = wave + 4 * j;
            const int tt = t < k ? t : k - 1;
            raw[j] = f[ld_idx(tt, lc <= tt ? lc : 0)];
        }
        if (!TRANS_IN) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int t = wave + 4 * j;
                const int tt = t < k ? t : k - 1;
                const i64 v = vo + (tt <= a ? ld_idx(a, tt) : ld_idx(tt, a));
                gv[j] = A[v * lda + (c < n ? c : n - 1)];
            }
        } else {                                      // lane <-> t: the entries (a, 0..a) of a row of A are contiguous
            const i64 v = vo + (lc <= a ? ld_idx(a, lc) : ld_idx(lc, a));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const i64 col = c0 + wave + 4 * j;
                gv[j] = A[(col < n ? col : n - 1) * lda + v];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = wave + 4 * j;
            if (t < kr && cx < KP) Ls[t * KP + cx] = (t >= k || cx > t) ? 0.0 : (cx == t ? exp(raw[j]) : raw[j]);   // zero above the diagonal and in the padding
        }
        if (!TRANS_IN) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int t = wave + 4 * j;
                if (t < kr) G[t * GS + cx] = (t < k && c < n) ? gv[j] * (t == a ? 2.0 : 1.0) : 0.0;
            }
        } else if (cx < kr) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int cc = wave + 4 * j;
                G[cx * GS + cc] = (cx < k && c0 + cc < n) ? gv[j] * (cx == a ? 2.0 : 1.0) : 0.0;
            }
        }
    }
    __syncthreads();
    // rows (a, 0..a) of the result for this block's 64 columns = L[:, 0..a]^T G: 16 x 16 MFMA tiles, wave w <-> columns
    // [16 w, 16 w + 16), one tile of 16 rows b at a time; the k-steps below the tile's first row are zeros of L and skipped.
    // (Two VALU versions -- L from LDS per (b, t), and L through v_readlane -- took 24-29 us per launch at k = 43, n = 996:
    // ~40 times the useful multiply-adds in overhead instructions.)
    const int l15 = cx & 15, l4 = cx >> 4;
    const i64 col = c0 + 16 * wave + l15;
    for (int mt = 0; 16 * mt <= a; ++mt) {
        jt_d4 acc = (jt_d4){0.0, 0.0, 0.0, 0.0};
        for (int kk = 4 * mt; 4 * kk < kr; ++kk) {
            const int t = 4 * kk + l4;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[t * KP + 16 * mt + l15], G[t * GS + 16 * wave + l15], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = 16 * mt + l4 + 4 * q;
            if (b <= a && col < n) out[(fo + ld_idx(a, b)) * ldo + col] = (b == a) ? acc[q] * Ls[a * KP + a] : acc[q];      // dL_aa = exp(f_aa) = L_aa
        }
    }
}

// called once by lrvb_ctx_create: the row table of jt_apply_kernel (none when the layout has a simplex block or a
// log-Cholesky block too large for LDS: the dense route stays)
int upload_jtmap(lrvb_ctx* c) {
    c->jt_rows = 0; c->jt_lds = 0;
    i64 nbe = 0; int kmax = 0;
    for (const auto& b : c->blocks) {
        if (b.kind == LRVB_BLOCK_BOX) nbe += b.free_size;
        else if (b.kind == LRVB_BLOCK_PSD) { if (b.dim0 > 63) return LRVB_OK; if ((int)b.dim0 > kmax) kmax = (int)b.dim0; }
        else return LRVB_OK;
    }
    std::vector<i64> rows;
    for (const auto& b : c->blocks)
        if (b.kind == LRVB_BLOCK_PSD)
            for (i64 a = 0; a < b.dim0; ++a) { rows.push_back(1); rows.push_back(a); rows.push_back(b.free_off); rows.push_back(b.vec_off); rows.push_back(b.dim0); }
    for (i64 e = 0; e < nbe; e += 16) { rows.push_back(0); rows.push_back(e); rows.push_back(e + 16 < nbe ? e + 16 : nbe); rows.push_back(0); rows.push_back(0); }
    const size_t nrows = rows.size() / 5;
    if (nrows == 0 || nrows > 65535) return LRVB_OK;
    // [rows (5 i64 each) | box free index | box vector index | lb | ub]
    std::vector<double> host(rows.size() + (size_t)(4 * nbe));
    memcpy(host.data(), rows.data(), rows.size() * sizeof(i64));
    i64* foff = reinterpret_cast<i64*>(host.data() + rows.size());
    i64* voff = foff + nbe;
    double* lbs = host.data() + rows.size() + 2 * nbe; double* ubs = lbs + nbe;
    i64 e = 0;
    for (const auto& b : c->blocks)
        if (b.kind == LRVB_BLOCK_BOX)
            for (i64 i = 0; i < b.free_size; ++i, ++e) { foff[e] = b.free_off + i; voff[e] = b.vec_off + i; lbs[e] = b.lb; ubs[e] = b.ub; }
    LRVB_TRY(buf_reserve(c, c->jtmap, host.size()));
    HIP_TRY(hipMemcpyAsync(c->jtmap.p, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->jt_rows = (i64)nrows; c->jt_box = nbe; {
        const int kr = (kmax + 3) & ~3, KP = (kmax + 15) & ~15;
        c->jt_lds = (size_t)(kr * 65 + kr * KP) * sizeof(double); c->jt_kmax = kmax;
        if (c->jt_lds > 65536) {                       // 61 <= k <= 63: past the default dynamic-LDS limit
            HIP_TRY(hipFuncSetAttribute((const void*)jt_apply_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->jt_lds));
            HIP_TRY(hipFuncSetAttribute((const void*)jt_apply_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->jt_lds));
        }
    }
    return LRVB_OK;
}

// out (D x n, leading dimension ldo) = J(theta)^T A  (A: V x n, leading dimension lda; trans_in: A given as n x V)
int launch_jt_apply(lrvb_ctx* c, const double* theta_dev, const double* A, i64 lda, i64 n, double* out, i64 ldo, bool trans_in) {
    if (c->jt_rows <= 0) LRVB_FAIL(LRVB_ERR_STATE, "no structured Jacobian for this layout");
    const i64* rows = reinterpret_cast<const i64*>(c->jtmap.p);
    const i64* foff = rows + 5 * c->jt_rows;
    const i64* voff = foff + c->jt_box;
    const double* lbs = c->jtmap.p + 5 * c->jt_rows + 2 * c->jt_box;
    const double* ubs = lbs + c->jt_box;
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)c->jt_rows);
#define JT_LAUNCH(TR, NB) hipLaunchKernelGGL((jt_apply_kernel<TR, NB>), grid, dim3(256), c->jt_lds, c->stream, rows, foff, voff, lbs, ubs, theta_dev, A, lda, n, out, ldo)
    if (c->jt_kmax <= 16)      { if (trans_in) JT_LAUNCH(true, 4);  else JT_LAUNCH(false, 4); }
    else if (c->jt_kmax <= 32) { if (trans_in) JT_LAUNCH(true, 8);  else JT_LAUNCH(false, 8); }
    else                       { if (trans_in) JT_LAUNCH(true, 16); else JT_LAUNCH(false, 16); }
#undef JT_LAUNCH
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// Two buffers cleared by ONE launch (the one-call steps of configurations 2 and 4 are chains of short launches: every node
// of the chain costs 3-5 us whatever it does).
__global__ __launch_bounds__(256)
void zero2_kernel(double* __restrict__ a, i64 na, double* __restrict__ b, i64 nb) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) a[i] = 0.0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) b[i] = 0.0;
}
int launch_zero2(lrvb_ctx* c, double* a, size_t na, double* b, size_t nb) {
    const size_t n = na > nb ? na : nb;
    if (n == 0) return LRVB_OK;
    size_t grid = (n + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(zero2_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, a, (i64)na, b, (i64)nb);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

int launch_dense_jac(lrvb_ctx* c, const double* theta_dev, double* J_dev, i64 ld, i64 rows_alloc, bool zeroed) {
    // ld / rows_alloc: the caller's (zero-padded, even-width) allocation; 0 = the plain V x D matrix
    const i64 ldj = ld > 0 ? ld : c->D;
    if (!zeroed) HIP_TRY(hipMemsetAsync(J_dev, 0, sizeof(double) * (size_t)(rows_alloc > 0 ? rows_alloc : c->V) * (size_t)ldj, c->stream));
    const bool fused = box_fused(c);
    if (fused) LRVB_TRY(launch_box_all<1>(c, theta_dev, nullptr, J_dev, nullptr, nullptr, ldj, nullptr));
    for (const auto& b : c->blocks) {
        if (b.kind == LRVB_BLOCK_BOX) {
            if (b.free_size > 0 && !fused)
                hipLaunchKernelGGL(box_jac_dense_kernel, dim3(nblk(b.free_size)), dim3(256), 0, c->stream,
                                   theta_dev, b.free_off, b.vec_off, b.free_size, b.lb, b.ub, J_dev, ldj);
        } else if (b.kind == LRVB_BLOCK_PSD) {
            dim3 grid(nblk(b.free_size), (unsigned)b.vec_size);
            hipLaunchKernelGGL(psd_jac_kernel, grid, dim3(256), 0, c->stream,
                               theta_dev, b.free_off, b.vec_off, b.dim0, J_dev, ldj);
        } else {
            hipLaunchKernelGGL(simplex_jac_kernel, dim3(nblk(b.dim0 * b.dim1 * (b.dim1 - 1))), dim3(256), 0, c->stream,
                               theta_dev, b.free_off, b.vec_off, b.dim0, b.dim1, J_dev, ldj);
        }
        HIP_TRY(hipGetLastError());
    }
    return LRVB_OK;
}

// T (D x D) must be zero-initialised or hold a matrix to accumulate into.
int launch_third_order(lrvb_ctx* c, const double* theta_dev, const double* g_eta_dev, double* T_dev) {
    const i64 ldt = c->D;
    const bool fused = box_fused(c);
    if (fused) LRVB_TRY(launch_box_all<2>(c, theta_dev, g_eta_dev, T_dev, nullptr, nullptr, ldt, nullptr));
    for (const auto& b : c->blocks) {
        if (b.kind == LRVB_BLOCK_BOX) {
            if (b.free_size > 0 && !fused)
                hipLaunchKernelGGL(box_third_dense_kernel, dim3(nblk(b.free_size)), dim3(256), 0, c->stream,
                                   theta_dev, g_eta_dev, b.free_off, b.vec_off, b.free_size, b.lb, b.ub, T_dev, ldt);
        } else if (b.kind == LRVB_BLOCK_PSD) {
            dim3 grid(nblk(b.free_size), (unsigned)b.free_size);
            hipLaunchKernelGGL(psd_third_kernel, grid, dim3(256), 0, c->stream,
                               theta_dev, g_eta_dev, b.free_off, b.vec_off, b.dim0, T_dev, ldt);
        } else {
            const i64 k1 = b.dim1 - 1;
            hipLaunchKernelGGL(simplex_third_kernel, dim3(nblk(b.dim0 * k1 * k1)), dim3(256), 0, c->stream,
                               theta_dev, g_eta_dev, b.free_off, b.vec_off, b.dim0, b.dim1, T_dev, ldt);
        }
        HIP_TRY(hipGetLastError());
    }
    return LRVB_OK;
}
