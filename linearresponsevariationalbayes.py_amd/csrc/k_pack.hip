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
