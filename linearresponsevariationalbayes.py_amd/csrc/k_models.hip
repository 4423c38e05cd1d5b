// k_models.hip -- per-model kernels of the BASELINE.json configurations: per-observation quadratic forms, group sums and the
// elimination of the 2G local parameters of the hierarchical model, the Dirichlet blocks and the Schur assembly of the mixture,
// the operands of the Wishart model's G^T G, higher-order loss derivatives, the Gauss-Hermite logistic term.
#include "lrvb_internal.h"
#include "k_kernels.h"
#include <math.h>

__global__ __launch_bounds__(256)
void obs_quadform_kernel(const double* __restrict__ Z, i64 ldz, int q, const double* __restrict__ M,
                         const double* __restrict__ cvec, i64 K, i64 n0, i64 n1, double* __restrict__ out)
{
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 n = n0 + blockIdx.y;
    if (k >= K || n >= n1) return;
    const double* z = Z + n * ldz;
    const double* Mk = M + k * (i64)q * q;
    double s = 0.0;
    for (int a = 0; a < q; ++a) {
        double t = 0.0;
        for (int b = 0; b < q; ++b) t += Mk[a * q + b] * z[b];
        s += z[a] * t;
    }
    out[(n - n0) * K + k] = 0.5 * s + (cvec ? cvec[k] : 0.0);
}

// out[g, 0] = sum_{n in g} w_n,  out[g, 1 + j] = sum_{n in g} w_n z_nj.  One wavefront per group walks
// the group's rows in a fixed order (counting-sort permutation built once on the host when the group
// ids are set), lane = column: deterministic, no atomics.  These are the per-group Sigma w, Sigma w y,
// Sigma w x of doc/lmm.lyx:105-160.
__global__ __launch_bounds__(256)
void group_sums_kernel(const double* __restrict__ Z, i64 ldz, int q, const double* __restrict__ w,
                       const i64* __restrict__ perm, const i64* __restrict__ offs, i64 n_groups,
                       double* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const i64 g = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const i64 b = offs[g], e = offs[g + 1];
    double sw = 0.0, s0 = 0.0;                  // lane j < q accumulates column j; every lane tracks sum w
    i64 k = b;
    for (; k + 1 < e; k += 2) {                 // two rows in flight
        const i64 n0 = perm[k], n1 = perm[k + 1];
        const double w0 = w[n0], w1 = w[n1];
        const double z0 = lane < q ? Z[n0 * ldz + lane] : 0.0;
        const double z1 = lane < q ? Z[n1 * ldz + lane] : 0.0;
        sw += w0; s0 += w0 * z0;
        sw += w1; s0 += w1 * z1;
    }
    if (k < e) {
        const i64 n0 = perm[k];
        const double w0 = w[n0];
        sw += w0; s0 += w0 * (lane < q ? Z[n0 * ldz + lane] : 0.0);
    }
    double* dst = out + g * (i64)(q + 1);
    if (lane == 0) dst[0] = sw;
    if (lane < q) dst[1 + lane] = s0;
}

// One wavefront per batch of groups, lane = column of the group's row [W_g | sum w x (p) | sum w y] of the resident
// statistics.  For group g (doc/lmm.lyx:105-160; e_g, i_g the mean and information of q(u_g)):
//   r_g = sum w y - (sum w x) . m,   a_g = W_g e_g - r_g,   d_g = e_g - e_mu,   D_g = ty W_g + tm,
// the two columns of the arrow Hessian's cross block that belong to (e_g, i_g), in vector coordinates of the p + 5
// coupled global rows [mean of q(beta) (p) | e_mu | a_y | b_y | a_mu | b_mu], times d local / d free,
//   c_e = [ty sum w x | -tm | a_g tay | a_g tby | d_g tam | d_g tbm],
//   c_i = [0 | 0 | -W_g tay / (2 i_g^2) | -W_g tby / (2 i_g^2) | -tam / (2 i_g^2) | -tbm / (2 i_g^2)] * (i_g - lb),
// go to rows 2g and 2g + 1 of C (width ldc), the reciprocals of the free local diagonal to the weights; the sums over
// groups that the global gradient and the scalars of the ELBO need are accumulated per wave (fixed order) in `part`.
//   sums[0 .. p) = sum_g e_g sum w x;  sums[64 + k]: 0 sum e_g r_g, 1 sum W_g (e_g^2 + 1 / i_g), 2 sum d_g^2 + 1 / i_g,
//   3 sum d_g, 4 sum log i_g, 5 sum W_g, 6 sum (local free gradient)^2 (a stationarity diagnostic)
__global__ __launch_bounds__(256)
void lmm_group_kernel(const double* __restrict__ gs, i64 G, int p, const double* __restrict__ par, const double* __restrict__ floc,
                      double* __restrict__ C, int ldc, double* __restrict__ wts, double* __restrict__ part)
{
    const int lane = threadIdx.x & 63;
    const i64 gw = (i64)blockIdx.x * 4 + (threadIdx.x >> 6), GW = (i64)gridDim.x * 4;
    const double ty = par[0], tm = par[1], e_mu = par[2], tay = par[3], tby = par[4], tam = par[5], tbm = par[6], lb = par[7];
    const double mj = (lane >= 1 && lane <= p) ? par[8 + lane - 1] : 0.0;       // m aligned with the sum w x lanes
    const int q1 = p + 2;                                                     // entries of a statistics row
    double v1 = 0.0;                                                          // lane 1 + j: sum_g e_g (sum w x)_j
    double sc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (i64 g = gw; g < G; g += GW) {
        const double val = lane < q1 ? gs[g * q1 + lane] : 0.0;
        double dotv = val * mj;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dotv += __shfl_xor(dotv, off);
        const double W = __shfl(val, 0), sy = __shfl(val, p + 1);
        const double eg = floc[g], ig = lb + exp(floc[G + g]), jl = ig - lb;
        const double rg = sy - dotv, a = W * eg - rg, d = eg - e_mu, Dg = ty * W + tm;
        const double i2 = 1.0 / (ig * ig);
        const double dl_i = Dg * i2 / ig - 0.5 * i2, g_i = -0.5 * Dg * i2 + 0.5 / ig, g_e = ty * a + tm * d;
        const double dfe = Dg, dfi = dl_i * jl * jl + g_i * jl;
        const double sx = __shfl(val, lane + 1 < 64 ? lane + 1 : 63);           // lane r < p: (sum w x)_r
        double ce, ci;
        if (lane < p) { ce = ty * sx; ci = 0.0; }
        else if (lane == p) { ce = -tm; ci = 0.0; }
        else if (lane == p + 1) { ce = a * tay; ci = -0.5 * W * i2 * tay * jl; }
        else if (lane == p + 2) { ce = a * tby; ci = -0.5 * W * i2 * tby * jl; }
        else if (lane == p + 3) { ce = d * tam; ci = -0.5 * i2 * tam * jl; }
        else if (lane == p + 4) { ce = d * tbm; ci = -0.5 * i2 * tbm * jl; }
        else { ce = 0.0; ci = 0.0; }
        if (lane < ldc) { C[(2 * g) * (i64)ldc + lane] = ce; C[(2 * g + 1) * (i64)ldc + lane] = ci; }
        if (lane == 0) { wts[2 * g] = 1.0 / dfe; wts[2 * g + 1] = 1.0 / dfi; }
        v1 += val * eg;
        sc[0] += eg * rg; sc[1] += W * (eg * eg + 1.0 / ig); sc[2] += d * d + 1.0 / ig; sc[3] += d;
        sc[4] += log(ig); sc[5] += W; sc[6] += g_e * g_e + (g_i * jl) * (g_i * jl);
    }
    // one partial row per WORKGROUP (the four waves added in wave order): few groups per wave keep the dependent chain of a
    // wave short (G = 1e4: four groups per wave, 625 workgroups; 20 per wave on 128 workgroups took 25 us), and the partial
    // count stays what lmm_sums_kernel adds in one workgroup
    __shared__ double red[4][128];
    const int wave = threadIdx.x >> 6;
    red[wave][lane] = (lane >= 1 && lane <= p) ? v1 : 0.0;    // shifted by one: slot 1 + j
    red[wave][64 + lane] = lane < 7 ? sc[lane] : 0.0;
    __syncthreads();
    if (threadIdx.x < 128)
        part[(i64)blockIdx.x * 128 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// sums[k] = sum over the wave partials in a fixed order (eight interleaved slices, then the slices in order); the vector
// part is moved down by one slot
__global__ __launch_bounds__(1024)
void lmm_sums_kernel(const double* __restrict__ part, int n_waves, double* __restrict__ sums) {
    __shared__ double sh[8][128];
    const int k = threadIdx.x & 127, sl = threadIdx.x >> 7;
    double a0 = 0.0, a1 = 0.0;
    int wv = sl;
    for (; wv + 8 < n_waves; wv += 16) { a0 += part[(i64)wv * 128 + k]; a1 += part[(i64)(wv + 8) * 128 + k]; }
    if (wv < n_waves) a0 += part[(i64)wv * 128 + k];
    sh[sl][k] = a0 + a1;
    __syncthreads();
    if (sl == 0) {
        double a = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) a += sh[s][k];
        if (k < 64) { if (k >= 1) sums[k - 1] = a; if (k == 63) sums[63] = 0.0; }
        else sums[k] = a;
    }
}

// tail[0..1] = val2, tail[2] = number of rows whose local block was not positive definite, tail[3] = 0
__global__ void mixture_tail_kernel(i64 n, const double* __restrict__ val2, const int* __restrict__ bad, double* __restrict__ tail) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tail[i] = (i < 2) ? val2[i] : (i == 2 ? (double)(*bad) : 0.0);
}

// Rm[(j K + k), (j' K + k')] = R[(j q + j'), (k K + k')]: the (q^2 x K^2) operand re-indexed as the square
// matrix that sits between d vec(Lam) / d free and its transpose
__global__ void mixture_permute_kernel(i64 total, int q, int K, const double* __restrict__ R, double* __restrict__ Rm)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const i64 n = (i64)q * K;
    const i64 r = e / n, cc = e - r * n;
    const int j = (int)(r / K), k = (int)(r - (i64)j * K), jp = (int)(cc / K), kp = (int)(cc - (i64)jp * K);
    Rm[e] = R[((i64)j * q + jp) * ((i64)K * K) + (i64)k * K + kp];
}

// H = diag(s) Hgg diag(s) + diag(d) - 1/2 (S + S^T)
__global__ void mixture_schur_finish_kernel(i64 total, i64 n, const double* __restrict__ Hgg, const double* __restrict__ sc,
                                            const double* __restrict__ dg, const double* __restrict__ S, double* __restrict__ H)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const i64 r = e / n, cc = e - r * n;
    double v = Hgg[e];
    if (sc) v *= sc[r] * sc[cc];
    if (dg && r == cc) v += dg[r];
    H[e] = v - 0.5 * (S[e] + S[cc * n + r]);
}

// ---- J^T R J for J = "diagonal + constant per Dirichlet" without forming J (configuration 3's Schur assembly) ----------------
// d E log p / d alpha of a Dirichlet and the Dirichlet entropy / expectation Hessians are diagonal plus a constant on the
// blocks of a partition (diag(psi1(alpha)) - psi1(alpha_0): LRVB/ExponentialFamilies.py:118-120, DirichletParams.py:19-26):
//   J[r, c] = ( [r == c] d[r] + [g(r) == g(c)] gam[g(r)] ) * s[c],
// g(r) = 0 for r < K (the Dirichlet over the K mixture weights), 1 + (r mod K) otherwise (the K Dirichlets over the vocabulary,
// parameter (v, k) at index K + v K + k).  Rounds 2-3 wrote such matrices out (dirichlet_blocks_kernel) and multiplied.
// With J[k, i] = ([k == i] d_k + [g(k) == g(i)] gam_g(k)) s_i,
//   (J^T R J)[i, j] = s_i s_j ( d_i d_j R[i, j] + d_i gam_g(j) RS[i, g(j)] + gam_g(i) SR[g(i), j] d_j + gam_g(i) gam_g(j) SRS[g(i), g(j)] ),
// RS[i, b] = sum_{l in b} R[i, l], SR[a, j] = sum_{k in a} R[k, j], SRS[a, b] = sum_{k in a} RS[k, b]: O(n^2) work instead of the two
// dense n^3 products (2 x (43 + 33 + 6) us of MFMA product, split reduction and unpacking at n = 1024).  Groups: 0 = indices
// [0, K), 1 + k = indices K + v K + k.
__global__ __launch_bounds__(256)
void dirichlet_rowsums_kernel(i64 n, int K, const double* __restrict__ R, double* __restrict__ RS /* n x (K + 1) */) {
    const i64 i = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* row = R + i * n;
    for (int b = threadIdx.x & 63; b <= K; b += 64) {
        double a = 0.0;
        if (b == 0) { for (int l = 0; l < K; ++l) a += row[l]; }
        else        { for (i64 l = K + (b - 1); l < n; l += K) a += row[l]; }
        RS[i * (K + 1) + b] = a;
    }
}
__global__ __launch_bounds__(256)
void dirichlet_colsums_kernel(i64 n, int K, const double* __restrict__ R, double* __restrict__ SR /* (K + 1) x n */) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (j >= n) return;
    double s = 0.0;
    if (a == 0) { for (int k = 0; k < K; ++k) s += R[(i64)k * n + j]; }
    else        { for (i64 k = K + (a - 1); k < n; k += K) s += R[k * n + j]; }
    SR[(i64)a * n + j] = s;
}
__global__ void dirichlet_blocksums_kernel(i64 n, int K, const double* __restrict__ RS, double* __restrict__ SRS /* (K + 1)^2 */) {
    const int a = blockIdx.x;
    for (int b = threadIdx.x; b <= K; b += blockDim.x) {
        double s = 0.0;
        if (a == 0) { for (int k = 0; k < K; ++k) s += RS[(i64)k * (K + 1) + b]; }
        else        { for (i64 k = K + (a - 1); k < n; k += K) s += RS[k * (K + 1) + b]; }
        SRS[a * (K + 1) + b] = s;
    }
}
// H = Hgg o (sc sc^T) + diag(dg) - 1/2 (S + S^T), S = J^T R J as above, Hgg = "diagonal h_diag + constant h_const per Dirichlet"
__global__ void dirichlet_schur_finish_kernel(i64 total, i64 n, int K, const double* __restrict__ R, const double* __restrict__ RS,
                                              const double* __restrict__ SR, const double* __restrict__ SRS,
                                              const double* __restrict__ d, const double* __restrict__ gam,
                                              const double* __restrict__ h_diag, const double* __restrict__ h_const,
                                              const double* __restrict__ sc, const double* __restrict__ dg, double* __restrict__ H)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const i64 i = e / n, j = e - i * n;
    const int gi = i < K ? 0 : 1 + (int)(i % K), gj = j < K ? 0 : 1 + (int)(j % K);
    const int nb = K + 1;
    const double di = d[i], dj = d[j], ci = gam[gi], cj = gam[gj];
    const double blk = ci * cj * SRS[gi * nb + gj];
    const double sij = di * dj * R[i * n + j] + di * cj * RS[i * nb + gj] + ci * SR[(i64)gi * n + j] * dj + blk;
    const double sji = dj * di * R[j * n + i] + dj * ci * RS[j * nb + gi] + cj * SR[(i64)gj * n + i] * di + cj * ci * SRS[gj * nb + gi];
    double v = (gi == gj) ? h_const[gi] : 0.0;
    if (i == j) v += h_diag[i];
    const double ss = sc[i] * sc[j];
    v *= ss;
    if (i == j) v += dg[i];
    H[e] = v - 0.5 * ss * (sij + sji);
}

// M~ (Pv x V, Pv = q (q + 1) / 2) holds M_k FOLDED onto the packed lower triangle v = a (a + 1) / 2 + b, b <= a:
// z^T M_k z = sum_{a >= b} (M_k[a][b] + M_k[b][a]) z_a z_b (the diagonal once) -- the Kronecker kernel forms z_a z_b for b <= a
// only (wsyrk_kron_kernel).  Everything stays on the device.
__device__ __forceinline__ void tri_pair(i64 v, int& a, int& b) {
    a = (int)((sqrt(8.0 * (double)v + 1.0) - 1.0) * 0.5);
    while ((i64)a * (a + 1) / 2 > v) --a;
    while ((i64)(a + 1) * (a + 2) / 2 <= v) ++a;
    b = (int)(v - (i64)a * (a + 1) / 2);
}
__global__ void mtilde_kernel(const double* __restrict__ M, i64 V, int q, double* __restrict__ Mt /* Pv x V */) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 v = blockIdx.y;                       // packed row index
    if (k >= V) return;
    int a, b;
    tri_pair(v, a, b);
    const double* Mk = M + k * (i64)q * q;
    Mt[v * V + k] = (b == a) ? Mk[a * q + a] : Mk[a * q + b] + Mk[b * q + a];
}

// s in the same packed order (the entries of the symmetric q x q matrix of second moments)
__global__ void svec_kernel(const double* __restrict__ S1 /* q x q */, int q, double* __restrict__ sv /* Pv */) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= q * (q + 1) / 2) return;
    int a, b;
    tri_pair(v, a, b);
    sv[v] = S1[a * q + b];
}

// ---- the per-coordinate matrices of the Wishart + MVN model as a SPARSE operand (configuration 5, lrvb_wishart_gram) ----------
// The per-observation term of the model, l_n = 1/2 nu ((y_n - m)^T V (y_n - m) + tr(V Sigma_mu)) - 1/2 E log|Lambda|
// (LRVB/NormalParams.py:6-23, WishartParams.py:6-35), is quadratic in z = [y; 1]: d l_n / d eta_k = 1/2 z^T M_k z + c_k with
//   d/d m_i     : M = -nu (e_q (V e_i)^T + (V e_i) e_q^T) + 2 nu (V m)_i e_q e_q^T            (e_q: the slot of the constant 1)
//   d/d nu      : M = [[V, -V m], [-(V m)^T, m^T V m]]
//   d/d V_(r c) : M = nu (E_rc + E_cr [r != c]) - nu (em e_q^T + e_q em^T) + nu m_r m_c (2 [r != c] + [r == c]) e_q e_q^T,
//                 em = m_c e_r + [r != c] m_r e_c
//   d/d Lambda_mu: M = 0 (it enters through c only).
// Folded onto the packed lower triangle of z z^T (mtilde_kernel's rule), column k of M~ has 64 nonzeros for a mean coordinate
// (the last triangle row (d, 0..d)), at most four for a coordinate of V ((r, c), (d, r), (d, c), (d, d)), none for the information
// block of q(mu), and is dense for nu only: 14 K of 8.5 M entries at d = 63.  The two products around K4 -- 2176^2 x 4096 and
// 2176 x 4096^2 on the matrix cores, 1.9 ms, until the end of round 4 -- are gathers of a few rows each; the 134 MB of matrices
// (8 V q^2 bytes) and the 71 MB of M~ that rounds 3-4 wrote on the device are never formed.  Pinned against the generic entry
// point fed with host-built matrices (tests/test_gpu_wishart.py::test_gram_with_device_generated_matrices, 1e-13) and exact AD.
__device__ __forceinline__ int wish_nnz(const WishartGen& g, i64 k, i64 pv) {
    const i64 d = g.d, mm = d * (d + 1) / 2;
    if (k >= g.ms && k < g.ms + d) return (int)d + 1;
    if (k == g.inu) return 0;                             // the dense column: a matrix-vector product of its own (wishart_nu_coef_kernel)
    if (k >= g.vs && k < g.vs + mm) return 4;             // (the third entry of a diagonal coordinate has coefficient zero)
    return 0;
}
__device__ __forceinline__ void wish_entry(const WishartGen& g, i64 k, int t, i64& row, double& coef) {
    const i64 d = g.d;
    const i64 rd = d * (d + 1) / 2;                       // first packed index of triangle row d
    if (k >= g.ms && k < g.ms + d) {
        const i64 i = k - g.ms;
        row = rd + t;
        coef = t < d ? -2.0 * g.nu * g.v[(i64)t * d + i] : 2.0 * g.nu * g.vm[i];
    } else {
        int r, cc; tri_pair(k - g.vs, r, cc);
        const bool off = r != cc;
        if (t == 0)      { row = (i64)r * (r + 1) / 2 + cc; coef = off ? 2.0 * g.nu : g.nu; }
        else if (t == 1) { row = rd + r;  coef = -2.0 * g.nu * g.m[cc]; }
        else if (t == 2) { row = rd + cc; coef = off ? -2.0 * g.nu * g.m[r] : 0.0; }
        else             { row = rd + d;  coef = g.nu * g.m[r] * g.m[cc] * (off ? 2.0 : 1.0); }
    }
}
// the dense column of M~ (the coordinate nu), written out: pv coefficients
__global__ void wishart_nu_coef_kernel(WishartGen g, i64 pv, double* __restrict__ cnu) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pv) return;
    const i64 d = g.d;
    int a, b; tri_pair(t, a, b);
    double coef;
    if (a < d) coef = a == b ? g.v[(i64)a * d + a] : g.v[(i64)a * d + b] + g.v[(i64)b * d + a];
    else if (b < d) coef = -2.0 * g.vm[b];
    else coef = g.mvm;
    cnu[t] = coef;
}
__global__ void scatter_column_kernel(i64 n, const double* __restrict__ x, double* __restrict__ M, i64 ld, i64 col) {
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) M[r * ld + col] = x[r];
}
// out (rows x V, leading dimension ldo) = A M~ for a row-major A (rows x Pv-or-more, leading dimension lda): thread <-> column k
__global__ __launch_bounds__(256)
void wishart_sparse_right_kernel(WishartGen g, i64 pv, i64 V, const double* __restrict__ A, i64 lda, double* __restrict__ out, i64 ldo) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 r = blockIdx.y;
    if (k >= V) return;
    if (k == g.inu) return;                               // written by the matrix-vector product with the nu coefficients
    const double* arow = A + r * lda;
    const int nz = wish_nnz(g, k, pv);
    double s = 0.0;
    for (int t = 0; t < nz; ++t) { i64 row; double coef; wish_entry(g, k, t, row, coef); s += coef * arow[row]; }
    out[r * ldo + k] = s;
}
// out (V x n, leading dimension ldo) = M~^T B for a row-major B (Pv-or-more x n, leading dimension ldb): thread <-> column j of B
__global__ __launch_bounds__(256)
void wishart_sparse_left_kernel(WishartGen g, i64 pv, i64 n, const double* __restrict__ B, i64 ldb, double* __restrict__ out, i64 ldo) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 k = blockIdx.y;
    if (j >= n || k == g.inu) return;                     // (row nu: a matrix-vector product of its own)
    const int nz = wish_nnz(g, k, pv);
    double s = 0.0;
    for (int t = 0; t < nz; ++t) { i64 row; double coef; wish_entry(g, k, t, row, coef); s += coef * B[row * ldb + j]; }
    out[k * ldo + j] = s;
}

__global__ void rank_terms_kernel(i64 V, const double* __restrict__ n_obs_dev, const double* __restrict__ t, const double* __restrict__ cvec,
                                  double* __restrict__ A /* V x V, holds M~^T K4 M~ */) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = blockIdx.y;
    if (j >= V) return;
    const double n_obs = *n_obs_dev;                 // the number of observations of ALL shards (summed with the statistics)
    A[i * V + j] = 0.25 * A[i * V + j] + 0.5 * (t[i] * cvec[j] + cvec[i] * t[j]) + n_obs * cvec[i] * cvec[j];
}


// D^j g_eta [u_1 .. u_j] for the declared objective: the building block of the reference's higher-order
// sensitivity (`ParametricSensitivityTaylorExpansion`, LRVB/ModelSensitivity.py:382-515, which obtains the
// same quantity from j nested autograd JVPs of the gradient closure, :38-62, 221-234).  In vector coordinates
// the linear predictor is linear in eta, so the mixed derivative has the closed form
//   X^T ( w o loss^(j+1)(z) o (X u_2) o ... o (X u_j) o (X u_1) )   [+ s A u_1 when j = 1],
// i.e. the cached-curvature Hessian-vector pass with a different per-observation coefficient: one skinny
// product for z and the X u_k, one elementwise kernel, one fused pass.  j = 0 is the gradient itself.
__device__ __forceinline__ double loss_derivative(int loss, double lik, int m, double y, double z) {
    if (loss == LRVB_LOSS_GAUSSIAN) return m == 1 ? lik * (z - y) : (m == 2 ? lik : 0.0);
    if (loss == LRVB_LOSS_POISSON) { const double e = exp(z); return m == 1 ? e - y : e; }
    // logistic: loss' = sigma - y, loss^(m) = sigma^(m-1), polynomials in s = sigma(z) from
    // P_1 = s - s^2, P_(k+1) = P_k' (s - s^2)
    const double s = 1.0 / (1.0 + exp(-z));
    switch (m) {
    case 1: return s - y;
    case 2: return s * (1.0 - s);
    case 3: return s * (1.0 + s * (-3.0 + s * 2.0));
    case 4: return s * (1.0 + s * (-7.0 + s * (12.0 - s * 6.0)));
    case 5: return s * (1.0 + s * (-15.0 + s * (50.0 + s * (-60.0 + s * 24.0))));
    case 6: return s * (1.0 + s * (-31.0 + s * (180.0 + s * (-390.0 + s * (360.0 - s * 120.0)))));
    case 7: return s * (1.0 + s * (-63.0 + s * (602.0 + s * (-2100.0 + s * (3360.0 + s * (-2520.0 + s * 720.0))))));
    default: return 0.0;
    }
}

__global__ void dk_coef_kernel(i64 n, int loss, double lik, int m, const double* __restrict__ w, const double* __restrict__ y,
                               const double* __restrict__ T, int Q, double* __restrict__ coef) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p = w[i] * loss_derivative(loss, lik, m, y[i], T[i * Q]);
    for (int k = 1; k < Q; ++k) p *= T[i * Q + k];
    coef[i] = p;
}

__global__ void obs_loss_kernel(i64 n, int loss, double lik, const double* __restrict__ y, const double* __restrict__ z,
                                double* __restrict__ out) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double zi = z[i], yi = y[i];
    double v;
    if (loss == LRVB_LOSS_GAUSSIAN) { const double d = zi - yi; v = 0.5 * lik * d * d; }
    else if (loss == LRVB_LOSS_POISSON) v = exp(zi) - yi * zi;
    else v = (zi > 0.0 ? zi + log1p(exp(-zi)) : log1p(exp(zi))) - yi * zi;       // log(1 + e^z) - y z, overflow-free
    out[i] = v;
}

// phi(m, s) = sum_k w_k log(1 + exp(m + sqrt(2) s x_k)) / sqrt(pi) and the derivatives of THIS SUM with respect to (m, s)
// (what autograd forms from the reference's expression), up to second order.  log(1 + e^t) in the overflow-free form.
__device__ __forceinline__ void gh_logistic_point(double m, double sd, const double* __restrict__ gx, const double* __restrict__ gw, int K,
                                                  double& v, double& dm, double& ds, double& dmm, double& dms, double& dss) {
    const double r2 = 1.4142135623730951, ispi = 0.5641895835477563;     // sqrt(2), 1 / sqrt(pi)
    v = dm = ds = dmm = dms = dss = 0.0;
    for (int k = 0; k < K; ++k) {
        const double xk = r2 * gx[k], wk = gw[k] * ispi;
        const double t = m + sd * xk;
        const double e = exp(-fabs(t));
        const double sp = (t > 0.0 ? t : 0.0) + log1p(e);                // log(1 + e^t)
        const double sg = t >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);     // sigmoid(t)
        const double s2 = sg * (1.0 - sg);
        v += wk * sp; dm += wk * sg; ds += wk * sg * xk;
        dmm += wk * s2; dms += wk * s2 * xk; dss += wk * s2 * xk * xk;
    }
}

__global__ __launch_bounds__(256)
void gh_logistic_kernel(i64 n, const double* __restrict__ zm, const double* __restrict__ zs, const double* __restrict__ gx,
                        const double* __restrict__ gw, int K, int order, double* __restrict__ val, double* __restrict__ d1, double* __restrict__ d2)
{
    __shared__ double sx[128], sw[128];
    if ((int)threadIdx.x < K) { sx[threadIdx.x] = gx[threadIdx.x]; sw[threadIdx.x] = gw[threadIdx.x]; }
    __syncthreads();
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v, dm, ds, dmm, dms, dss;
    gh_logistic_point(zm[i], zs[i], sx, sw, K, v, dm, ds, dmm, dms, dss);
    val[i] = v;
    if (order >= 1) { d1[2 * i] = dm; d1[2 * i + 1] = ds; }
    if (order >= 2) { d2[3 * i] = dmm; d2[3 * i + 1] = dms; d2[3 * i + 2] = dss; }
}

// z_n ~ N(mu_n, v_n), mu = X mean, v = (X o X) var.  Per observation psi(mu, v) = phi(mu, sqrt(v)) - y mu; this kernel turns
// (mu, v) into the weighted coefficient vectors of the gradient and of the three Hessian products, and block sums of w psi.
__global__ __launch_bounds__(256)
void logitnormal_coef_kernel(i64 n, const double* __restrict__ mu, const double* __restrict__ vv, const double* __restrict__ y,
                             const double* __restrict__ w, const double* __restrict__ gx, const double* __restrict__ gw, int K,
                             double* __restrict__ a1, double* __restrict__ a2, double* __restrict__ c11, double* __restrict__ c12,
                             double* __restrict__ c22, double* __restrict__ vpart)
{
    __shared__ double sx[128], sw[128], red[4];
    if ((int)threadIdx.x < K) { sx[threadIdx.x] = gx[threadIdx.x]; sw[threadIdx.x] = gw[threadIdx.x]; }
    __syncthreads();
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    double contrib = 0.0;
    if (i < n) {
        const double m = mu[i], var = vv[i], sd = sqrt(var), wi = w[i];
        double v, dm, ds, dmm, dms, dss;
        gh_logistic_point(m, sd, sx, sw, K, v, dm, ds, dmm, dms, dss);
        // chain sd = sqrt(var): sd' = 1 / (2 sd), sd'' = -1 / (4 sd^3)
        const double s1 = 0.5 / sd, s2 = -0.25 / (sd * var);
        contrib = wi * (v - y[i] * m);
        a1[i] = wi * (dm - y[i]);
        a2[i] = wi * ds * s1;
        c11[i] = wi * dmm;
        c12[i] = wi * dms * s1;
        c22[i] = wi * (dss * s1 * s1 + ds * s2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) contrib += __shfl_xor(contrib, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) vpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
