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

typedef double d4 __attribute__((ext_vector_type(4)));

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

// All-reduce over the 32 lanes of a HALF wavefront without the LDS crossbar: four DPP steps inside the 16-lane rows
// (xor 1, xor 2, half-row mirror, row mirror) and one v_permlane16_swap (gfx950) across the two rows of the half --
// 17 VALU instructions for a double, against 6 x (2 ds_bpermute + wait) for the full-wave butterfly above.
template <int CTRL> __device__ __forceinline__ double mx_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// v of rows (0, 1, 2, 3) -> a = (0, 0, 2, 2) and b = (1, 1, 3, 3) (or the other way round: both uses are symmetric)
#define MX_ROW_PAIR(v, a, b) \
    const unsigned v##_lo = (unsigned)__double2loint(v), v##_hi = (unsigned)__double2hiint(v); \
    const auto v##_l = __builtin_amdgcn_permlane16_swap(v##_lo, v##_lo, false, false); \
    const auto v##_h = __builtin_amdgcn_permlane16_swap(v##_hi, v##_hi, false, false); \
    const double a = __hiloint2double((int)v##_h[0], (int)v##_l[0]), b = __hiloint2double((int)v##_h[1], (int)v##_l[1])
__device__ __forceinline__ double mx_half_sum(double v) {
    v += mx_dpp<0xB1>(v);           // quad_perm [1, 0, 3, 2]
    v += mx_dpp<0x4E>(v);           // quad_perm [2, 3, 0, 1]
    v += mx_dpp<0x141>(v);          // row_half_mirror
    v += mx_dpp<0x140>(v);          // row_mirror
    MX_ROW_PAIR(v, a, b);
    return a + b;
}
__device__ __forceinline__ double mx_half_max(double v) {
    v = fmax(v, mx_dpp<0xB1>(v));
    v = fmax(v, mx_dpp<0x4E>(v));
    v = fmax(v, mx_dpp<0x141>(v));
    v = fmax(v, mx_dpp<0x140>(v));
    MX_ROW_PAIR(v, a, b);
    return fmax(a, b);
}

// Pass 1: values, local gradient, statistics row, and A_n by the O(K) diagonal-plus-rank-two formula.
// Rows that need the dense factorisation (a small or negative d_k, or force_dense) are appended to `todo`.
//
// Round 2 ran one row per wavefront with full-wave ds_bpermute butterflies and K broadcast reads + 4 K fused multiply-adds
// per lane for A_n: 2.55 ms at N = 1e6, K = 32 (2.1 TB/s of writes).  What changed (measurements: DESIGN.md section 12):
//  * TWO rows per wavefront: lanes 0..31 <-> the categories of row 2q, lanes 32..63 <-> those of row 2q + 1 (K <= 32): no
//    lane idles at K = 32 and every reduction stays inside a half (mx_half_sum, no LDS crossbar).
//  * A_n = diag(c3) + sum_kk R[kk] C[kk]^T is a rank-FOUR update of a diagonal: exactly one v_mfma_f64_16x16x4 per 16 x 16
//    block (three blocks of the lower triangle at K = 32).
//  * The packed row of A_n is assembled in LDS and written out LINEARLY, 16 bytes per lane.  The kernel is bound by the
//    write requests the vector L1 (TCP) can have in flight towards L2 (TA_DATA_STALLED_BY_TC and TCP_PENDING_STALL ~ 100 % of
//    the kernel's cycles, ~565 cycles per request): written straight from the MFMA layout, a row was 12 instructions of
//    four unaligned 128-byte segments = ~90 requests of 47 bytes on average; linearly it is 66 full 64-byte requests.
//  * Every memory operation of the loop is written by hand so that no wait drains the stores: the inputs of the NEXT pair
//    arrive by LDS-DMA, every store is ALWAYS issued (rows or lanes that must not be written are masked through EXEC,
//    never branched around), exactly MX_STORES stores follow the DMAs of a pair and the wait for them is vmcnt(MX_STORES).
//    (hipcc cannot count stores behind branches and waits vmcnt(0) for any load -- draining the stores in every iteration.)
// stores and LDS-DMA loads under an explicit EXEC mask (call sites are wave-uniform control flow: EXEC is all ones there)
#define MX_ST(sbase, voff, val, mask) asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_mov_b64 exec, -1" \
    :: "v"(voff), "v"(val), "s"(sbase), "s"(mask) : "memory")
// (cache-policy bits on these stores -- nt, sc0, sc1, both, nt sc1 -- were measured in round 3: 1.148-1.18 ms, inside the noise)
#define MX_ST16(sbase, voff, val, mask) asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_mov_b64 exec, -1" \
    :: "v"(voff), "v"(val), "s"(sbase), "s"(mask) : "memory")
#define MX_DMA4(sbase, voff, ldsaddr, mask) asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dword %0, %1\n\ts_mov_b64 exec, -1" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr), "s"(mask) : "memory")
template <int N_> struct MxWait { static __device__ __forceinline__ void vm() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); } };
typedef double mx_d2 __attribute__((ext_vector_type(2)));

// row length of the operand matrix the kernel writes: the packed lower triangle, padded to an even number of doubles
__host__ __device__ constexpr int mixture_rows_lda(int K) { return K * (K + 1) / 2 + ((K * (K + 1) / 2) & 1); }

template <int K>
__global__ __launch_bounds__(256)
void mixture_rows_kernel(const double* __restrict__ theta_z, const double* __restrict__ X, int V,
                         const double* __restrict__ w, const double* __restrict__ Lam, i64 N,
                         double* __restrict__ Amat, double* __restrict__ U,
                         double* __restrict__ gfree, double* __restrict__ part_val, int* __restrict__ bad,
                         int force_dense, int* __restrict__ todo, int* __restrict__ todo_count)
{
    constexpr int KM = K - 1;
    constexpr int NB = (K > 16) ? 2 : 1;      // 16 x 16 blocks per side of A_n
    constexpr int TRI = K * (K + 1) / 2, LDA = mixture_rows_lda(K);
    constexpr int NSEG = (LDA * 8 + 1023) / 1024;             // 1 KiB store instructions per row
    constexpr int MX_STORES = 3 + 2 * NSEG;                   // per iteration: U (2), gradient, the two rows of A
    // factors of a row staged for the MFMA by OUTPUT category c (the swap of categories 0 and m undone):
    //   R[kk][c] at (c >> 4) * 64 + kk * 16 + (c & 15),  C[kk][c] at 128 + the same,  c3[c] at 256 + c
    // so that the operand of block b (lane <-> (kk = lane >> 4, c & 15 = lane & 15)) is 64 consecutive doubles: conflict-free
    constexpr int FROW = 288;
    constexpr int INB = 2 * 64 + 2;           // inputs of a pair: [logits of rows 2q, 2q + 1 (2 KM) | x rows (2 V) | w (2)], doubles
    constexpr int RB = (LDA > 128 ? LDA : 128) + 2;           // packed row of A_n; before that, x~ and the (p, g) exchange of both halves
    __shared__ double lam_s[32 * 32];
    __shared__ double vsum[4][2];
    __shared__ double fstage[4][2][FROW];
    __shared__ __attribute__((aligned(16))) double rowbuf[4][RB];
    __shared__ double inbuf[4][2][INB];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, hl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < (V + 1) * K; e += 256) lam_s[e] = Lam[e];
    __syncthreads();

    const bool cat = hl < K, loc = cat && hl >= 1;
    const i64 npair = (N + 1) / 2, nstep = (i64)gridDim.x * 4;
    const unsigned lds_in = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)inbuf[wave][0];
    const unsigned voff4 = (unsigned)lane * 4u;
    // five LDS-DMA instructions, always: the dwords of the pair's two logit rows (contiguous: 4 KM), of its two x rows (4 V),
    // of its two weights; dwords past the last row of the problem are masked
    auto issue_inputs = [&](i64 q, int buf) {
        i64 rows = N - 2 * q;
        if (rows > 2) rows = 2;
        if (rows < 0) rows = 0;
        const i64 q0 = rows > 0 ? q : 0;                                 // keep the (unused) addresses in range
        const int nz = (int)rows * 2 * KM, nx = (int)rows * 2 * V, nw = (int)rows * 2;
        const char* zb = reinterpret_cast<const char*>(theta_z + 2 * q0 * KM);
        const char* xb = reinterpret_cast<const char*>(X + 2 * q0 * V);
        const char* wb = reinterpret_cast<const char*>(w + 2 * q0);
        const unsigned dst = lds_in + (unsigned)(buf * INB) * 8u;
        MX_DMA4(zb, voff4, dst, __ballot(lane < nz));
        MX_DMA4(zb + 256, voff4, dst + 256u, __ballot(lane + 64 < nz));
        MX_DMA4(xb, voff4, dst + 512u, __ballot(lane < nx));
        MX_DMA4(xb + 256, voff4, dst + 768u, __ballot(lane + 64 < nx));
        MX_DMA4(wb, voff4, dst + 1024u, __ballot(lane < nw));
    };

    double v_lin = 0.0, v_ent = 0.0;        // -w sum z s  and  w sum z log z, per LANE until the end of the kernel
    int flag = 0;
    double* fs = fstage[wave][h];
    double* rbw = rowbuf[wave];
    double* es = rbw + 64 * h;              // x~ of the half, then its (p, g) exchange: both dead before the row is assembled
    const double* lam_col = lam_s + (cat ? hl : 0);
    const int pg = lane >> 4, pj = lane & 15;

    i64 q = (i64)blockIdx.x * 4 + wave;
    int buf = 0;
    issue_inputs(q, 0);
    MxWait<0>::vm();
    for (; q < npair; q += nstep, buf ^= 1) {
        const i64 n = 2 * q + h;
        const bool valid = n < N;
        const double* ib = inbuf[wave][buf];
        const double logit = loc ? ib[h * KM + hl - 1] : 0.0;            // lane (h, hl) <-> row 2 q + h, category hl; logit 0 for category 0
        const double xt = (hl >= 1 && hl <= V) ? ib[64 + h * V + hl - 1] : ((hl == 0) ? 1.0 : 0.0);
        const double wn = ib[128 + h];
        es[hl] = xt;                                                      // x~ of the row, broadcast source of the s loop
        issue_inputs(q + nstep, buf ^ 1);                                 // 5 DMAs, then exactly MX_STORES stores until the wait below
        const unsigned long long vm_ = __ballot(valid);
        const char* ub = reinterpret_cast<const char*>(U + 2 * q * 64);
        const char* gb = reinterpret_cast<const char*>(gfree + 2 * q * KM);
        MX_ST(ub, (unsigned)(h * 64 + hl) * 8u, xt, vm_);                 // sufficient-statistics row [x~ (32) | z (32)]
        const double mxl = mx_half_max(cat ? logit : -INFINITY);
        const double ex = cat ? exp(logit - mxl) : 0.0;
        const double den = mx_half_sum(ex);
        const double p = ex / den;
        const double logp = cat ? (logit - mxl - log(den)) : 0.0;
        MX_ST(ub, (unsigned)(h * 64 + 32 + hl) * 8u, cat ? p : 0.0, vm_);
        double s = 0.0;
        for (int j = 0; j <= V; ++j) s += es[j] * lam_col[j * K];
        if (!cat) s = 0.0;
        const double g = cat ? -wn * (s - logp - 1.0) : 0.0;
        const double gdotp = mx_half_sum(g * p);
        if (valid) {
            v_lin += cat ? -wn * p * s : 0.0;
            v_ent += cat ? wn * p * logp : 0.0;
        }
        MX_ST(gb, (unsigned)(h * KM + hl - 1) * 8u, p * (g - gdotp), __ballot(valid && loc && gfree != nullptr));     // J^T g:  p_{j+1} (g_{j+1} - g.p); issued (masked) also when nobody asked for it
        // reference category of the row = its arg-max category m (see mixture_row_prelude): categories 0 and m change places
        const unsigned long long bal = __ballot(cat && logit == mxl);
        int m = __ffs((unsigned)(h ? (bal >> 32) : bal)) - 1;
        if (m < 0) m = 0;                                                 // a row of NaN logits: keep the addresses in range
        __builtin_amdgcn_wave_barrier();
        es[2 * hl] = p; es[2 * hl + 1] = g;
        __builtin_amdgcn_wave_barrier();
        const double p_m = es[2 * m], g_m = es[2 * m + 1], p_0 = es[0], g_0 = es[1];
        __builtin_amdgcn_wave_barrier();
        const double ps = (hl == 0) ? p_m : ((hl == m) ? p_0 : p);
        const double gs = (hl == 0) ? g_m : ((hl == m) ? g_0 : g);
        // M = Dg - r s^T - s r^T with Dg = diag(d), d_k = w + g_k - g.p, s = r o (d - w/2): a DIAGONAL plus a
        // rank-two term, so M^-1 follows from the Woodbury identity in O(K) and
        //   A = diag(t) - t p^T - p t^T + alpha p p^T - [a1 a2] T^-1 [a1 a2]^T,   t_k = p_k / d_k (k >= 1),
        //   T = [[alpha, beta - 1], [beta - 1, gamma]],  alpha = sum t,  beta = sum r s / d,  gamma = sum s^2 / d,
        //   a1 = t - alpha p,  a2 = (p - w t / 2)[k >= 1] - beta p.
        // With Dg > 0, M is positive definite iff det T < 0 (inertia additivity); det T = -p_ref at the
        // optimum of the row.
        const double dk = wn + gs - gdotp;
        const double dmin = -mx_half_max(loc ? -dk : -INFINITY);
        const bool fastp = !force_dense && wn > 0.0 && dmin > 0.05 * wn;         // uniform over the half
        if (valid && !fastp && hl == 0) todo[atomicAdd(todo_count, 1)] = (int)n;
        const bool emit = valid && fastp;
        const double t = loc ? ps / dk : 0.0;
        const double alpha = mx_half_sum(t);
        const double sumq = mx_half_sum(loc ? ps : 0.0);
        const double beta = sumq - 0.5 * wn * alpha;
        const double gamma = mx_half_sum(loc ? ps * dk : 0.0) - wn * sumq + 0.25 * wn * wn * alpha;
        const double det = alpha * gamma - (beta - 1.0) * (beta - 1.0);
        if (emit && !(det < 0.0)) flag = 1;
        const double idet = 1.0 / det;
        const double t11 = gamma * idet, t12 = (1.0 - beta) * idet, t22 = alpha * idet;
        const double a1 = cat ? t - alpha * ps : 0.0;
        const double a2 = cat ? (loc ? ps - 0.5 * wn * t : 0.0) - beta * ps : 0.0;
        const double w2 = wn * wn;
        const int colp = (hl == 0) ? m : ((hl == m) ? 0 : hl);            // lane <-> OUTPUT column (the swap undone)
        double* fc = fs + (colp >> 4) * 64 + (colp & 15);
        fc[0] = ps;                          fc[16] = t;              fc[32] = a1;            fc[48] = a2;
        fc[128] = w2 * (alpha * ps - t);     fc[144] = -w2 * ps;
        fc[160] = -w2 * (a1 * t11 + a2 * t12);                        fc[176] = -w2 * (a1 * t12 + a2 * t22);
        fs[256 + colp] = w2 * t;
        __builtin_amdgcn_wave_barrier();
        const unsigned long long em = __ballot(emit);
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const bool on = (em >> (32 * row)) & 1ull;                    // wave-uniform; a row that is off still issues its stores
            const double* rb = fstage[wave][row];
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                const double aop = rb[64 * bi + lane];
#pragma unroll
                for (int bj = 0; bj <= bi; ++bj) {
                    const double bop = rb[128 + 64 * bj + lane];
                    d4 acc = {0.0, 0.0, 0.0, 0.0};
                    if (bi == bj) {
                        const double dg = rb[256 + 16 * bj + pj];
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[v] = (pg + 4 * v == pj) ? dg : 0.0;
                    }
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {                          // D: reg v <-> row g + 4 v, column j of the block
                        const int r = 16 * bi + pg + 4 * v, c = 16 * bj + pj;
                        if (r < K && c <= r) rbw[r * (r + 1) / 2 + c] = acc[v];       // packed lower triangle
                    }
                }
            }
            if (LDA > TRI && lane == 0) rbw[TRI] = 0.0;                   // even-width padding column
            __builtin_amdgcn_wave_barrier();
            const char* arow = reinterpret_cast<const char*>(Amat + (2 * q + row) * (i64)LDA);
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                const int e2 = sg * 64 + lane;                            // pair of doubles
                const mx_d2 val = *reinterpret_cast<const mx_d2*>(rbw + 2 * (e2 < LDA / 2 ? e2 : 0));
                MX_ST16(arow, (unsigned)e2 * 16u, val, __ballot(on && e2 < LDA / 2));
            }
            __builtin_amdgcn_wave_barrier();
        }
        MxWait<MX_STORES>::vm();                                          // the DMAs of the next pair have landed; the stores fly on
    }
    v_lin = mx_wave_sum(v_lin); v_ent = mx_wave_sum(v_ent);
    if (lane == 0) { vsum[wave][0] = v_lin; vsum[wave][1] = v_ent; }
    if (flag && hl == 0) atomicOr(bad, 1);
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
