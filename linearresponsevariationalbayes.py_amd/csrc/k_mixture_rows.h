#pragma once
// k_mixture_rows.h (included by k_mixture.hip and k_mixture_inst*.hip) -- the per-observation ("local") block of a mixture model with simplex-constrained
// responsibilities (BASELINE.json config 3: Dirichlet-multinomial mixture, K = 32, N = 1e6).
//
// Row n has a SimplexParam row z_n = softmax([0, f_n]) (LRVB/SimplexParams.py:11-18) and the local
// objective  l_n = -w_n sum_k z_nk s_nk + w_n sum_k z_nk log z_nk,  s_nk = sum_j x~_nj Lam_jk
// (x~ = (1, x_n), Lam = [E log pi; E log phi]).  One WAVEFRONT per row does everything the
// reference's Python triple loop over COO triplets does for that row (SimplexParams.py:106-155)
// and the elimination of the row's local block from the global Hessian:
//
//   p, s, g = d l / d z                                   (lane k <-> category k)
//   H_nn  = J^T diag(w / p) J + sum_k g_k d2 p_k          ((K-1) x (K-1), closed forms of
//                                                          SimplexParams.py:33-63; lane i <-> row i)
//   M = D^-1 H_nn D^-1,  D = diag(sqrt(p_2..p_K))         (scaling: M stays O(w) when p saturates)
//   L L^T = M   in registers, broadcasts by v_readlane (as the 64 x 64 Cholesky block)
//   Y = L^-1 (J D^-1)^T  (lane k <-> column k),   A_n = Y^T Y = J H_nn^-1 J^T   (K x K)
//
// and writes w_n^2 vec(A_n) (the row of the operand of the Schur-complement GEMM), the row
// [x~_n | z_n] of the sufficient-statistics matrix, the free local gradient, and value partials.
#include "lrvb_internal.h"
#include <math.h>

__device__ __forceinline__ double mx_bcast(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mx_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double mx_wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Shared prelude of one row on one wavefront: lane k <-> category k (lane 0 = reference, logit 0).
struct MixRow {
    double wn, p, logp, xt, s, g, gdotp;    // per lane (p, g: category `lane`)
    double ps, gs;                          // categories 0 and m swapped (m = arg-max category)
    int m;
    bool cat;
};

// the three global inputs of a row, loaded one row ahead of their use
struct MixIn { double wn, f, x; };
__device__ __forceinline__ MixIn mixture_row_load(int K, const double* __restrict__ theta_z, const double* __restrict__ X,
                                                  int V, const double* __restrict__ w, i64 n, int lane)
{
    const int KM = K - 1;
    MixIn in;
    in.wn = w[n];
    in.f = (lane < KM) ? theta_z[n * KM + lane] : 0.0;
    in.x = (lane >= 1 && lane <= V) ? X[n * V + lane - 1] : 0.0;
    return in;
}

__device__ __forceinline__ MixRow mixture_row_prelude(int K, const MixIn& in, int V, const double* lam_s, int lane)
{
    const int KM = K - 1;
    MixRow r;
    r.wn = in.wn;
    const double f = in.f;
    double logit = __shfl_up(f, 1, 64);
    if (lane == 0) logit = 0.0;
    r.cat = lane < K;
    const double mxl = mx_wave_max(r.cat ? logit : -INFINITY);
    const double ex = r.cat ? exp(logit - mxl) : 0.0;
    const double den = mx_wave_sum(ex);
    r.p = ex / den;
    r.logp = r.cat ? (logit - mxl - log(den)) : 0.0;
    r.xt = (lane == 0) ? 1.0 : in.x;
    double s = 0.0;
    for (int j = 0; j <= V; ++j) s += mx_bcast(r.xt, j) * (r.cat ? lam_s[j * K + lane] : 0.0);
    r.s = s;
    r.g = r.cat ? -r.wn * (s - r.logp - 1.0) : 0.0;
    r.gdotp = mx_wave_sum(r.g * r.p);
    // Two changes of variables keep the local block well conditioned when responsibilities saturate,
    // neither of which changes A_n = J H_nn^-1 J^T:
    //  (a) the REFERENCE category of the row is its arg-max category m instead of category 0 (a linear
    //      change of the free coordinates): H_nn has an eigenvalue ~ w p_ref, which underflows when
    //      p_0 -> 0.  Categories 0 and m are swapped for the elimination only.
    //  (b) scaling by D = diag(sqrt(p_{i+1})):  M = D^-1 H_nn D^-1,
    //      M_ij = r_i r_j (2 g.p - w - g_{i+1} - g_{j+1}) + d_ij (w + g_{i+1} - g.p)
    r.m = __builtin_amdgcn_readfirstlane((int)__ffsll((unsigned long long)__ballot(r.cat && logit == mxl)) - 1);
    const double p_m = mx_bcast(r.p, r.m), g_m = mx_bcast(r.g, r.m), p_0 = mx_bcast(r.p, 0), g_0 = mx_bcast(r.g, 0);
    r.ps = (lane == 0) ? p_m : ((lane == r.m) ? p_0 : r.p);
    r.gs = (lane == 0) ? g_m : ((lane == r.m) ? g_0 : r.g);
    return r;
}

// Pass 1: values, local gradient, statistics row, and A_n by the O(K) diagonal-plus-rank-two formula.
// Rows that need the dense factorisation (a small or negative d_k, or force_dense) are appended to `todo`.
template <int K>
__global__ __launch_bounds__(256)
void mixture_rows_kernel(const double* __restrict__ theta_z, const double* __restrict__ X, int V,
                         const double* __restrict__ w, const double* __restrict__ Lam, i64 N,
                         double* __restrict__ Amat, i64 lda, double* __restrict__ U,
                         double* __restrict__ gfree, double* __restrict__ part_val, int* __restrict__ bad,
                         int force_dense, int* __restrict__ todo, int* __restrict__ todo_count)
{
    constexpr int KM = K - 1;
    __shared__ double lam_s[32 * 32];
    __shared__ double vsum[4][2];
    __shared__ double fstage[4][32 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < (V + 1) * K; e += 256) lam_s[e] = Lam[e];
    __syncthreads();

    double v_lin = 0.0, v_ent = 0.0;        // -w sum z s  and  w sum z log z of this wave's rows
    int flag = 0;
    const i64 nstep = (i64)gridDim.x * 4;
    i64 n = (i64)blockIdx.x * 4 + wave;
    MixIn nxt_in = mixture_row_load(K, theta_z, X, V, w, n < N ? n : N - 1, lane);
    for (; n < N; n += nstep) {
        const MixIn in = nxt_in;
        nxt_in = mixture_row_load(K, theta_z, X, V, w, (n + nstep < N) ? n + nstep : N - 1, lane);   // next row in flight
        const MixRow r = mixture_row_prelude(K, in, V, lam_s, lane);
        const double wn = r.wn, ps = r.ps;
        const bool cat = r.cat;
        const int m = r.m;
        v_lin += mx_wave_sum(cat ? -wn * r.p * r.s : 0.0);
        v_ent += mx_wave_sum(cat ? wn * r.p * r.logp : 0.0);
        // free local gradient: J^T g, lane j <-> free index j:  p_{j+1} (g_{j+1} - g.p)
        {
            const double pn = __shfl_down(r.p, 1, 64), gn = __shfl_down(r.g, 1, 64);
            if (lane < KM) gfree[n * KM + lane] = pn * (gn - r.gdotp);
        }
        // sufficient-statistics row [x~ (32) | z (32)]
        {
            const double zsh = __shfl(r.p, lane - 32, 64);
            U[n * 64 + lane] = (lane < 32) ? r.xt : ((lane - 32 < K) ? zsh : 0.0);
        }
        if (lda > (i64)K * (K + 1) / 2 && lane == 0) Amat[n * lda + (i64)K * (K + 1) / 2] = 0.0;     // even-width padding column
        // M = Dg - r s^T - s r^T with Dg = diag(d), d_k = w + g_k - g.p, s = r o (d - w/2): a DIAGONAL plus a
        // rank-two term, so M^-1 follows from the Woodbury identity in O(K) and
        //   A = diag(t) - t p^T - p t^T + alpha p p^T - [a1 a2] T^-1 [a1 a2]^T,   t_k = p_k / d_k (k >= 1),
        //   T = [[alpha, beta - 1], [beta - 1, gamma]],  alpha = sum t,  beta = sum r s / d,  gamma = sum s^2 / d,
        //   a1 = t - alpha p,  a2 = (p - w t / 2)[k >= 1] - beta p.
        // With Dg > 0, M is positive definite iff det T < 0 (inertia additivity); det T = -p_ref at the
        // optimum of the row.
        const bool loc = cat && lane >= 1;
        const double dk = wn + r.gs - r.gdotp;
        const double dmin = -mx_wave_max(loc ? -dk : -INFINITY);
        const bool fastp = __builtin_amdgcn_readfirstlane((int)(!force_dense && wn > 0.0 && dmin > 0.05 * wn)) != 0;
        if (!fastp) {
            if (lane == 0) todo[atomicAdd(todo_count, 1)] = (int)n;
            continue;
        }
        const double t = loc ? ps / dk : 0.0;
        const double alpha = mx_wave_sum(t);
        const double sumq = mx_wave_sum(loc ? ps : 0.0);
        const double beta = sumq - 0.5 * wn * alpha;
        const double gamma = mx_wave_sum(loc ? ps * dk : 0.0) - wn * sumq + 0.25 * wn * wn * alpha;
        const double det = alpha * gamma - (beta - 1.0) * (beta - 1.0);
        if (!(det < 0.0)) flag = 1;
        const double idet = 1.0 / det;
        const double t11 = gamma * idet, t12 = (1.0 - beta) * idet, t22 = alpha * idet;
        const double a1 = cat ? t - alpha * ps : 0.0;
        const double a2 = cat ? (loc ? ps - 0.5 * wn * t : 0.0) - beta * ps : 0.0;
        const double w2 = wn * wn;
        const double b1 = w2 * (a1 * t11 + a2 * t12), b2 = w2 * (a1 * t12 + a2 * t22);
        const double c1 = w2 * (alpha * ps - t), c2 = w2 * ps, c3 = w2 * t;
        double* fs = fstage[wave];
        if (lane < 32) { fs[4 * lane] = ps; fs[4 * lane + 1] = t; fs[4 * lane + 2] = a1; fs[4 * lane + 3] = a2; }
        __builtin_amdgcn_wave_barrier();
        // A is symmetric: only its lower triangle is stored, packed (row r, column c <= r at r(r+1)/2 + c);
        // lane <-> column, so a row is one contiguous store
        const int colp = (lane == 0) ? m : ((lane == m) ? 0 : lane);
        double* arow = Amat + n * lda + colp;
#pragma unroll 8
        for (int kp = 0; kp < K; ++kp) {
            const double pk = fs[4 * kp], tk = fs[4 * kp + 1], a1k = fs[4 * kp + 2], a2k = fs[4 * kp + 3];
            double v = c1 * pk - c2 * tk - b1 * a1k - b2 * a2k;
            if (kp == lane) v += c3;
            const int rowp = (kp == 0) ? m : ((kp == m) ? 0 : kp);
            if (cat && rowp >= colp) arow[rowp * (rowp + 1) / 2] = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { vsum[wave][0] = v_lin; vsum[wave][1] = v_ent; }
    if (flag && lane == 0) atomicOr(bad, 1);
    __syncthreads();
    if (tid == 0) {
        part_val[2 * blockIdx.x] = ((vsum[0][0] + vsum[1][0]) + vsum[2][0]) + vsum[3][0];
        part_val[2 * blockIdx.x + 1] = ((vsum[0][1] + vsum[1][1]) + vsum[2][1]) + vsum[3][1];
    }
}

// Pass 2 (rows listed in `todo` only): the dense route.  M = L L^T in registers with v_readlane
// broadcasts (as the 64 x 64 Cholesky block of k_linalg.hip), Y = L^-1 (J D^-1)^T (lane k <-> column k),
// A_n = Y^T Y.
template <int KT>                        // size class: K <= KT; rows/columns past K - 1 are identity padding
__global__ __launch_bounds__(256)
void mixture_rows_dense_kernel(int K, const double* __restrict__ theta_z, const double* __restrict__ X, int V,
                               const double* __restrict__ w, const double* __restrict__ Lam,
                               double* __restrict__ Amat, i64 lda, int* __restrict__ bad,
                               const int* __restrict__ todo, const int* __restrict__ todo_count)
{
    constexpr int KMT = KT - 1;
    const int KM = K - 1;
    __shared__ double lam_s[32 * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < (V + 1) * K; e += 256) lam_s[e] = Lam[e];
    __syncthreads();
    const int count = *todo_count;
    int flag = 0;
    for (int q = blockIdx.x * 4 + wave; q < count; q += gridDim.x * 4) {
        const i64 n = todo[q];
        const MixRow r = mixture_row_prelude(K, mixture_row_load(K, theta_z, X, V, w, n, lane), V, lam_s, lane);
        const double wn = r.wn, ps = r.ps, gdotp = r.gdotp;
        const bool cat = r.cat;
        const int m = r.m;
        const double p1 = __shfl_down(ps, 1, 64), g1 = __shfl_down(r.gs, 1, 64);
        const double r1 = sqrt(p1);
        double a[KMT];                                  // lane i <-> row i
#pragma unroll
        for (int j = 0; j < KMT; ++j) {
            const double rj = mx_bcast(r1, j), gj = mx_bcast(g1, j);
            double h = r1 * rj * (2.0 * gdotp - wn - g1 - gj);
            if (j == lane) h += wn + g1 - gdotp;
            a[j] = (lane < KM && j < KM) ? h : ((j == lane) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int j = 0; j < KMT; ++j) {
            const double d = mx_bcast(a[j], j);
            if (!(d > 0.0)) flag = 1;
            const double rs = 1.0 / sqrt(d);
            a[j] = a[j] * rs;
#pragma unroll
            for (int k = j + 1; k < KMT; ++k) a[k] -= a[j] * mx_bcast(a[j], k);
        }
        // lane k solves L y = Jhat[k, :]^T,  Jhat[k][i] = r_i (d_{k,i+1} - p_k)   (r_i = 0 past K - 1)
        double y[KMT];
#pragma unroll
        for (int i = 0; i < KMT; ++i) {
            double rhs = (i < KM) ? mx_bcast(r1, i) * ((lane == i + 1 ? 1.0 : 0.0) - ps) : 0.0;
#pragma unroll
            for (int c = 0; c < i; ++c) rhs -= mx_bcast(a[c], i) * y[c];
            y[i] = rhs / mx_bcast(a[i], i);
        }
        const double w2 = wn * wn;
        const int colp = (lane == 0) ? m : ((lane == m) ? 0 : lane);
        double* arow = Amat + n * lda + colp;
        for (int kp = 0; kp < K; ++kp) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < KMT; ++i) acc += y[i] * mx_bcast(y[i], kp);
            const int rowp = (kp == 0) ? m : ((kp == m) ? 0 : kp);
            if (cat && rowp >= colp) arow[rowp * (rowp + 1) / 2] = w2 * acc;      // packed lower triangle
        }
    }
    if (flag && lane == 0) atomicOr(bad, 1);
}
