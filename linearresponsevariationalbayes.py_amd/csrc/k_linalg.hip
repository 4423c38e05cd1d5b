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
    if ((transB & 2) && blockIdx.x > blockIdx.y) return;      // lower tiles only
    transB &= 1;

    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

    // the next K-chunk is fetched into registers while the current one is on the MFMA pipe
    double ar[4], br[4];
    auto fetch = [&](i64 k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid;
            int kk, mm;
            if (transA) { mm = idx & 63; kk = idx >> 6; } else { kk = idx & 15; mm = idx >> 4; }
            const i64 gm = m0 + mm, gk = k0 + kk;
            ar[it] = (gm < M && gk < K) ? (transA ? A[gk * lda + gm] : A[gm * lda + gk]) : 0.0;
            int kb, nn;
            if (transB) { kb = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kb = idx >> 6; }
            const i64 gn = n0 + nn, gkb = k0 + kb;
            br[it] = (gn < N && gkb < K) ? (transB ? B[gn * ldb + gkb] : B[gkb * ldb + gn]) : 0.0;
        }
    };
    fetch(0);
    for (i64 k0 = 0; k0 < K; k0 += GM_KC) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid;
            int kk, mm;
            if (transA) { mm = idx & 63; kk = idx >> 6; } else { kk = idx & 15; mm = idx >> 4; }
            As[kk][mm] = ar[it];
            int kb, nn;
            if (transB) { kb = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kb = idx >> 6; }
            Bs[kb][nn] = br[it];
        }
        __syncthreads();
        if (k0 + GM_KC < K) fetch(k0 + GM_KC);
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

// C (M x M, lower 64 x 64 tiles only) = alpha * A A^T + beta * C, A is M x K row-major
int launch_gemm_lower(lrvb_ctx* c, i64 M, i64 K, double alpha, const double* A, i64 lda,
                      double beta, double* C, i64 ldc) {
    if (M <= 0) return LRVB_OK;
    const unsigned t = (unsigned)((M + GM_TILE - 1) / GM_TILE);
    hipLaunchKernelGGL(gemm_f64_kernel, dim3(t, t), dim3(256), 0, c->stream, 0, 1 | 2,
                       M, M, K, alpha, A, lda, A, lda, beta, C, ldc);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
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
// Right-looking blocked factorisation, block size 64:
//   (1) one wavefront factors the 64 x 64 diagonal block held in registers (lane i = row i,
//       broadcasts by v_readlane) and also forms W = L_jj^-1 (four waves, blocked 16 -> 32 -> 64 on MFMA);
//   (2) panel  P <- P W^T           (MFMA GEMM, in place: one 64-column tile per row tile);
//   (3) trailing A22 -= P P^T       (MFMA GEMM, lower tiles only).
// The solves use the stored W blocks: X_j = W_j B_j / W_j^T B_j are GEMMs too.
constexpr int CH_NB = 64;

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

constexpr int CH_LS = CH_NB + 2;          // LDS row stride of the diagonal block (even: 16-byte pair reads)

constexpr int CH_WS = CH_NB + 1;          // LDS row stride of W

// 16 x 16 x 16 tile product on one wave, acc += X Y: X(i, k) = xs[i * ldx + k], Y(k, j) = ys[k * ldy + j]
__device__ __forceinline__ d4 tile16_mm(const double* xs, int ldx, const double* ys, int ldy, d4 acc, int lane) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        const double av = xs[(lane & 15) * ldx + 4 * kc + (lane >> 4)];
        const double bv = ys[(4 * kc + (lane >> 4)) * ldy + (lane & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    return acc;
}
// acc += X T with T held in MFMA result layout (register r = rows 4 r + (lane >> 4), which is exactly
// the B operand of K-chunk r)
__device__ __forceinline__ d4 tile16_mm_reg(const double* xs, int ldx, d4 t, d4 acc, int lane) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        const double av = xs[(lane & 15) * ldx + 4 * kc + (lane >> 4)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, t[kc], acc, 0, 0, 0);
    }
    return acc;
}

// The body of the diagonal-block step, for a block that is already in Ls (lower triangle, identity padding past nb; Ws zeroed;
// a workgroup barrier behind both): factor, write L back to A, form W = L^-1.
__device__ __forceinline__ void potrf_inv_diag_body(double* __restrict__ Ls, double* __restrict__ Ws, double* __restrict__ colbuf4,
                                                    double* __restrict__ A, i64 lda, int nb, double* __restrict__ W /* 64 x 64 */,
                                                    int* __restrict__ info, int col0)
{
    // Wave 0 factors the block, lane i <-> row i.  The cross-lane traffic of the factorisation (a step
    // needs the eliminated columns of every later row k in every lane) goes through LDS: the lanes deposit
    // their entries once per step and read the others' back with wave-uniform 16-byte reads instead of
    // v_readlane broadcasts.  LDS operations of one wave execute in order, so the loop needs no barrier.
    // W = L^-1 is then formed by all four waves, blocked 16 -> 32 -> 64 with MFMA products:
    //   inv [[P, 0], [Q, R]] = [[P^-1, 0], [-R^-1 Q P^-1, R^-1]].
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {
        double a[CH_NB];
#pragma unroll
        for (int k = 0; k < CH_NB; ++k) a[k] = Ls[lane * CH_LS + k];
        int bad = 0;
        // FOUR columns per step (round 3; one column per step: 19 of the kernel's 29 us were this chain, ~700 cycles per
        // column of which the LDS round trip and its fences were the larger part).  The lanes deposit their entries of
        // columns j..j+3 once; every lane factors the 4 x 4 pivot block P = Lp Lp^T redundantly (uniform reads), takes its own
        // row of L by forward substitution, y_i Lp^T = x_i, and updates its trailing row with
        //   A[i][k] -= y_i . y_k = (Lp^-T y_i) . x_k = u_i . x_k
        // where x_k is the RAW deposit of row k: no second exchange.  Same arithmetic as the column-by-column
        // right-looking factorisation up to the order of the four subtractions.
#pragma unroll
        for (int j = 0; j < CH_NB; j += 4) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) colbuf4[lane * 4 + cq] = a[j + cq];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const double p00 = colbuf4[j * 4];
            const double p10 = colbuf4[(j + 1) * 4], p11 = colbuf4[(j + 1) * 4 + 1];
            const double p20 = colbuf4[(j + 2) * 4], p21 = colbuf4[(j + 2) * 4 + 1], p22 = colbuf4[(j + 2) * 4 + 2];
            const double p30 = colbuf4[(j + 3) * 4], p31 = colbuf4[(j + 3) * 4 + 1], p32 = colbuf4[(j + 3) * 4 + 2],
                         p33 = colbuf4[(j + 3) * 4 + 3];
            const double d0 = p00;
            const double rs0 = rsqrt(d0);            // reciprocal square roots instead of a division and a square root each
            const double l10 = p10 * rs0, l20 = p20 * rs0, l30 = p30 * rs0;
            const double d1 = fma(-l10, l10, p11);
            const double rs1 = rsqrt(d1);
            const double l21 = fma(-l20, l10, p21) * rs1, l31 = fma(-l30, l10, p31) * rs1;
            const double d2 = fma(-l21, l21, fma(-l20, l20, p22));
            const double rs2 = rsqrt(d2);
            const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * rs2;
            const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33)));
            const double rs3 = rsqrt(d3);
            if (bad == 0) {
                if (!(d0 > 0.0)) bad = j + 1;
                else if (!(d1 > 0.0)) bad = j + 2;
                else if (!(d2 > 0.0)) bad = j + 3;
                else if (!(d3 > 0.0)) bad = j + 4;
            }
            // this lane's row of L in columns j..j+3
            const double y0 = a[j] * rs0;
            const double y1 = fma(-y0, l10, a[j + 1]) * rs1;
            const double y2 = fma(-y1, l21, fma(-y0, l20, a[j + 2])) * rs2;
            const double y3 = fma(-y2, l32, fma(-y1, l31, fma(-y0, l30, a[j + 3]))) * rs3;
            a[j] = y0; a[j + 1] = y1; a[j + 2] = y2; a[j + 3] = y3;
            // u = Lp^-T y
            const double u3 = y3 * rs3;
            const double u2 = fma(-l32, u3, y2) * rs2;
            const double u1 = fma(-l31, u3, fma(-l21, u2, y1)) * rs1;
            const double u0 = fma(-l30, u3, fma(-l20, u2, fma(-l10, u1, y0))) * rs0;
#pragma unroll
            for (int k = j + 4; k < CH_NB; ++k)
                a[k] = fma(-u3, colbuf4[k * 4 + 3], fma(-u2, colbuf4[k * 4 + 2], fma(-u1, colbuf4[k * 4 + 1], fma(-u0, colbuf4[k * 4], a[k]))));
        }
        if (lane == 0 && bad != 0 && *info == 0) *info = col0 + bad;
#pragma unroll
        for (int k = 0; k < CH_NB; ++k) Ls[lane * CH_LS + k] = (k <= lane) ? a[k] : 0.0;
    }
    __syncthreads();
    for (int r = wave; r < nb; r += 4)
        if (lane <= r && lane < nb) A[(i64)r * lda + lane] = Ls[r * CH_LS + lane];

    // level 1: wave w inverts the 16 x 16 diagonal block w; lane c solves L_ww x = e_c
    {
        const double* Lw = Ls + (16 * wave) * CH_LS + 16 * wave;
        const int cc = lane & 15;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double s = (i == cc) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) s = fma(-Lw[i * CH_LS + k], x[k], s);
            x[i] = s / Lw[i * CH_LS + i];
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Ws[(16 * wave + i) * CH_WS + 16 * wave + cc] = x[i];
        }
    }
    __syncthreads();
    const d4 zero4 = (d4){0.0, 0.0, 0.0, 0.0};
    // level 2: waves 0 and 1 complete the two 32 x 32 diagonal inverses, W[d1][d0] = -W[d1][d1] L[d1][d0] W[d0][d0]
    if (wave < 2) {
        const int d0 = 2 * wave, d1 = d0 + 1;
        const d4 t = tile16_mm(Ls + (16 * d1) * CH_LS + 16 * d0, CH_LS, Ws + (16 * d0) * CH_WS + 16 * d0, CH_WS, zero4, lane);
        const d4 o = tile16_mm_reg(Ws + (16 * d1) * CH_WS + 16 * d1, CH_WS, t, zero4, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Ws[(16 * d1 + (lane >> 4) + 4 * r) * CH_WS + 16 * d0 + (lane & 15)] = -o[r];
    }
    __syncthreads();
    // level 3: wave (a, b) forms tile (2 + a, b) of -W[2:4][2:4] L[2:4][0:2] W[0:2][0:2]
    {
        const int ta = wave >> 1, tb = wave & 1;
        d4 t[2];
#pragma unroll
        for (int cq = 0; cq < 2; ++cq) {
            d4 acc = zero4;
#pragma unroll
            for (int e = 0; e < 2; ++e)
                acc = tile16_mm(Ls + (16 * (2 + cq)) * CH_LS + 16 * e, CH_LS, Ws + (16 * e) * CH_WS + 16 * tb, CH_WS, acc, lane);
            t[cq] = acc;
        }
        d4 o = zero4;
#pragma unroll
        for (int cq = 0; cq < 2; ++cq)
            o = tile16_mm_reg(Ws + (16 * (2 + ta)) * CH_WS + 16 * (2 + cq), CH_WS, t[cq], o, lane);
        __syncthreads();       // every wave has read the 32 x 32 diagonal inverses before the corner is written
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Ws[(16 * (2 + ta) + (lane >> 4) + 4 * r) * CH_WS + 16 * tb + (lane & 15)] = -o[r];
    }
    __syncthreads();
    for (int r = wave; r < CH_NB; r += 4) W[r * CH_NB + lane] = Ws[r * CH_WS + lane];
}

__global__ __launch_bounds__(256)
void potrf_inv_diag_kernel(double* __restrict__ A, i64 lda, int nb, double* __restrict__ W /* 64 x 64 */,
                           int* __restrict__ info, int col0)
{
    __shared__ double Ls[CH_NB * CH_LS];
    __shared__ double Ws[CH_NB * CH_WS];
    __shared__ __attribute__((aligned(16))) double colbuf4[CH_NB * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // coalesced load of the block (row r, column lane), identity padding past nb
    for (int r = wave; r < CH_NB; r += 4) {
        Ls[r * CH_LS + lane] = (r < nb && lane < nb) ? ((lane <= r) ? A[(i64)r * lda + lane] : 0.0) : ((r == lane) ? 1.0 : 0.0);
        Ws[r * CH_WS + lane] = 0.0;
    }
    __syncthreads();
    potrf_inv_diag_body(Ls, Ws, colbuf4, A, lda, nb, W, info, col0);
}

// Trailing update of a Cholesky step, A22 -= P P^T on the lower 64 x 64 tiles, with the NEXT step's diagonal block fused in:
// the workgroup of tile (0, 0) keeps its updated tile in LDS and goes straight on to factor it and form its inverse while
// the other workgroups are still updating theirs -- one dependent-kernel boundary and the 16 us of the trailing update leave
// the critical path of every step (a step was diag 27 us -> panel 11 us -> trailing 16 us, each waiting for the last).
// Both operands of a Cholesky step's products are 64 columns wide: they are staged WHOLE (one barrier, sixteen k-steps
// without another) instead of in 16-column chunks with two barriers each -- at this size the generic kernel's time was its
// barriers and dependent loads (10.6 us for a 64-deep product).
constexpr int CH_KS = CH_NB + 1;           // [k][m] stage rows; an ODD stride: the staging writes run along k (64 lanes, one per row of the stage) and would all fall on one bank with an even one
__device__ __forceinline__ void chol_tile_product(const double (*As)[CH_KS], const double (*Bs)[CH_KS], d4 (&acc)[2][2],
                                                  int wr, int wc, int lane)
{
#pragma unroll
    for (int ks = 0; ks < CH_NB / 4; ++ks) {
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
}

// Panel of a Cholesky step, in place: P (M x 64, row-major) <- P W^T with W the inverse of the diagonal block's factor.
// One workgroup per 64 rows reads and writes only its own rows.
__global__ __launch_bounds__(256)
void chol_panel_kernel(i64 M, double* __restrict__ P, i64 ldp, const double* __restrict__ W /* 64 x 64 */)
{
    __shared__ double As[CH_NB][CH_KS];   // [k][m] = P[m][k]
    __shared__ double Bs[CH_NB][CH_KS];   // [k][n] = W[n][k]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const i64 m0 = (i64)blockIdx.x * CH_NB;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 256 + tid;
        const int kk = idx & 63, mm = idx >> 6;
        As[kk][mm] = (m0 + mm < M) ? P[(m0 + mm) * ldp + kk] : 0.0;
        Bs[kk][mm] = W[mm * CH_NB + kk];
    }
    __syncthreads();
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    chol_tile_product(As, Bs, acc, wr, wc, lane);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const i64 gm = m0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r;
                if (gm < M) P[gm * ldp + wc * 32 + b * 16 + (lane & 15)] = acc[a][b][r];
            }
}

// Trailing update of a Cholesky step, A22 -= P P^T on the lower 64 x 64 tiles, with the NEXT step's diagonal block fused in:
// the workgroup of tile (0, 0) keeps its updated tile in LDS and goes straight on to factor it and form its inverse while
// the other workgroups are still updating theirs -- one dependent-kernel boundary and the 16 us of the trailing update leave
// the critical path of every step (a step was diag 27 us -> panel 11 us -> trailing 16 us, each waiting for the last).
__global__ __launch_bounds__(256)
void chol_trailing_diag_kernel(i64 M, const double* __restrict__ P, i64 ldp, double* __restrict__ C, i64 ldc,
                               int nb_next, double* __restrict__ W_next, int* __restrict__ info, int col0_next)
{
    // the two 64 x 64 operand stages are dead when the head workgroup starts on the diagonal block: same LDS
    __shared__ __attribute__((aligned(16))) double lds[2 * CH_NB * CH_KS > CH_NB * CH_LS + CH_NB * CH_WS + CH_NB * 4
                                                       ? 2 * CH_NB * CH_KS : CH_NB * CH_LS + CH_NB * CH_WS + CH_NB * 4];
    double (*As)[CH_KS] = reinterpret_cast<double (*)[CH_KS]>(lds);
    double (*Bs)[CH_KS] = reinterpret_cast<double (*)[CH_KS]>(lds + CH_NB * CH_KS);
    if (blockIdx.x > blockIdx.y) return;                      // lower tiles only
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const i64 m0 = (i64)blockIdx.y * CH_NB, n0 = (i64)blockIdx.x * CH_NB;
#pragma unroll
    for (int it = 0; it < 16; ++it) {                         // both operands are row pieces of the panel: [row][k], k fastest
        const int idx = it * 256 + tid;
        const int kk = idx & 63, mm = idx >> 6;
        As[kk][mm] = (m0 + mm < M) ? P[(m0 + mm) * ldp + kk] : 0.0;
        Bs[kk][mm] = (n0 + mm < M) ? P[(n0 + mm) * ldp + kk] : 0.0;
    }
    double cpre[2][2][4];                                     // the tile itself, requested before the product
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const i64 gm = m0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r, gn = n0 + wc * 32 + b * 16 + (lane & 15);
                cpre[a][b][r] = (gm < M && gn < M) ? C[gm * ldc + gn] : 0.0;
            }
    __syncthreads();
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    chol_tile_product(As, Bs, acc, wr, wc, lane);
    const bool head = (blockIdx.x == 0 && blockIdx.y == 0);   // the next diagonal block
    double* Ls = lds;
    double* Ws = lds + CH_NB * CH_LS;
    double* colbuf4 = Ws + CH_NB * CH_WS;
    if (head) {
        __syncthreads();                                      // every wave is done with the operand stages
        for (int r = wave; r < CH_NB; r += 4) {
            Ls[r * CH_LS + lane] = (r == lane) ? 1.0 : 0.0;   // identity padding; the tile's entries are written below
            Ws[r * CH_WS + lane] = 0.0;
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lm = wr * 32 + a * 16 + (lane >> 4) + 4 * r, ln = wc * 32 + b * 16 + (lane & 15);
                const i64 gm = m0 + lm, gn = n0 + ln;
                if (gm < M && gn < M) {
                    double* dst = C + gm * ldc + gn;
                    const double v = cpre[a][b][r] - acc[a][b][r];
                    if (!head) *dst = v;                       // the head tile is written by the factorisation (its lower part)
                    else if (lm < nb_next && ln < nb_next) Ls[lm * CH_LS + ln] = (ln <= lm) ? v : 0.0;
                }
            }
    if (!head) return;
    __syncthreads();
    potrf_inv_diag_body(Ls, Ws, colbuf4, C, ldc, nb_next, W_next, info, col0_next);
}

int launch_potrf_lower(lrvb_ctx* c, double* A, i64 n, i64 lda, int* info_dev) {
    HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), c->stream));
    const i64 nblk = (n + CH_NB - 1) / CH_NB;
    LRVB_TRY(buf_reserve(c, c->cholW, (size_t)(nblk * CH_NB * CH_NB)));
    for (i64 j0 = 0, jb = 0; j0 < n; j0 += CH_NB, ++jb) {
        const int nb = (int)((n - j0 < CH_NB) ? (n - j0) : CH_NB);
        double* Ajj = A + j0 * lda + j0;
        double* Wj = c->cholW.p + jb * CH_NB * CH_NB;
        if (j0 == 0) {                                       // every later diagonal block is factored by the step before it
            hipLaunchKernelGGL(potrf_inv_diag_kernel, dim3(1), dim3(256), 0, c->stream, Ajj, lda, nb, Wj, info_dev, (int)j0);
            HIP_TRY(hipGetLastError());
        }
        const i64 rows = n - j0 - nb;
        if (rows > 0) {
            double* Pn = A + (j0 + nb) * lda + j0;
            // P <- P W^T (in place: each workgroup reads and writes only its own 64 rows)
            hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)((rows + CH_NB - 1) / CH_NB)), dim3(256), 0, c->stream, rows, Pn, lda, (const double*)Wj);
            HIP_TRY(hipGetLastError());
            // trailing update A22 -= P P^T on the lower tiles + the next diagonal block's factorisation and inverse
            double* A22 = A + (j0 + nb) * lda + (j0 + nb);
            const int nb_next = (int)(rows < CH_NB ? rows : CH_NB);
            const unsigned t = (unsigned)((rows + GM_TILE - 1) / GM_TILE);
            hipLaunchKernelGGL(chol_trailing_diag_kernel, dim3(t, t), dim3(256), 0, c->stream, rows, (const double*)Pn, lda,
                               A22, lda, nb_next, Wj + CH_NB * CH_NB, info_dev, (int)(j0 + nb));
            HIP_TRY(hipGetLastError());
        }
    }
    return LRVB_OK;
}

// One step of a blocked triangular solve with the 64 x 64 inverse blocks W of the factor's diagonal (round 3: a step was
// two dependent launches of the generic GEMM, ~11 us each for 64-deep products; 32 launches for a forward solve at D = 1024):
//   forward  (BACK = false), step jb:  B_r -= L[r, jb] Y_jb for the row blocks r > jb, and the workgroups of row block jb + 1
//                                      go straight on to Y_{jb+1} = W_{jb+1} B_{jb+1} -- their tile is complete at that point;
//   backward (BACK = true),  step jb:  B_r -= L[jb, r]^T X_jb for r < jb, and row block jb - 1 goes on to
//                                      X_{jb-1} = W_{jb-1}^T B_{jb-1}.
// head_only: just the multiplication of block jb by W (the first step of a chain).  Same products in the same order as the
// two-launch form; the operands are staged whole (chol_tile_product).
template <bool BACK>
__global__ __launch_bounds__(256)
void trsm_step_kernel(const double* __restrict__ L, i64 ldl, const double* __restrict__ Wall, i64 n, int jb, int head_only,
                      double* __restrict__ B, i64 nrhs, i64 ldb)
{
    __shared__ double As[CH_NB][CH_KS];   // [k][m]
    __shared__ double Bs[CH_NB][CH_KS];   // [k][c]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const bool head = head_only || blockIdx.y == 0;
    const int rblk = head_only ? jb : (BACK ? jb - 1 - (int)blockIdx.y : jb + 1 + (int)blockIdx.y);
    const i64 r0 = (i64)rblk * CH_NB, j0 = (i64)jb * CH_NB, c0 = (i64)blockIdx.x * CH_NB;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    // everything the workgroup will need from global memory is requested up front: its own tile of B and (head) the inverse
    // block ride under the first product instead of costing a round trip each on the critical path of the chain
    double cpre[2][2][4], wpre[16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lm = wr * 32 + a * 16 + (lane >> 4) + 4 * r, lc = wc * 32 + b * 16 + (lane & 15);
                cpre[a][b][r] = ((r0 + lm < n) && (c0 + lc < nrhs)) ? B[(r0 + lm) * ldb + c0 + lc] : 0.0;
            }
    if (head) {
        const double* Wp = Wall + (i64)rblk * CH_NB * CH_NB;
#pragma unroll
        for (int it = 0; it < 16; ++it) wpre[it] = Wp[it * 256 + tid];
    }
    if (!head_only) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = it * 256 + tid;
            const int lo = idx & 63, hi = idx >> 6;
            if (BACK) {                   // As[k][m] = L[jb block row k][r block column m]: m runs along the row of L
                As[hi][lo] = (j0 + hi < n && r0 + lo < n) ? L[(j0 + hi) * ldl + r0 + lo] : 0.0;
            } else {                      // As[k][m] = L[r block row m][jb block column k]: k runs along the row of L
                As[lo][hi] = (r0 + hi < n && j0 + lo < n) ? L[(r0 + hi) * ldl + j0 + lo] : 0.0;
            }
            Bs[hi][lo] = (j0 + hi < n && c0 + lo < nrhs) ? B[(j0 + hi) * ldb + c0 + lo] : 0.0;
        }
        __syncthreads();
        chol_tile_product(As, Bs, acc, wr, wc, lane);
    }
    if (head) __syncthreads();                                // every wave is done with the stages before they are refilled
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lm = wr * 32 + a * 16 + (lane >> 4) + 4 * r, lc = wc * 32 + b * 16 + (lane & 15);
                const bool in = (r0 + lm < n) && (c0 + lc < nrhs);
                const double v = cpre[a][b][r] - acc[a][b][r];   // (0 outside the matrix: masked loads, zero stage rows)
                if (!head) { if (in) B[(r0 + lm) * ldb + c0 + lc] = v; }
                else Bs[lm][lc] = v;                           // operand of the product with W: [k = row of the block][c]
            }
    if (!head) return;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 256 + tid;
        const int lo = idx & 63, hi = idx >> 6;
        if (BACK) As[hi][lo] = wpre[it];                      // As[k][m] = W^T[m][k] = W[k][m]
        else      As[lo][hi] = wpre[it];                      // As[k][m] = W[m][k]
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    chol_tile_product(As, Bs, acc, wr, wc, lane);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lm = wr * 32 + a * 16 + (lane >> 4) + 4 * r, lc = wc * 32 + b * 16 + (lane & 15);
                if (r0 + lm < n && c0 + lc < nrhs) B[(r0 + lm) * ldb + c0 + lc] = acc[a][b][r];
            }
}

// The fused step is for the latency-bound regime (a step of a few hundred tiles: one launch instead of two dependent ones).
// A step with more tiles than the chip holds at once is throughput-bound: there the pipelined generic GEMM does the update
// (several workgroups per CU, loads under the MFMAs) and a head-only launch the multiplication by W.
constexpr i64 TRSM_FUSED_MAX_TILES = 512;
static int trsm_chain(lrvb_ctx* c, const double* L, i64 n, i64 ldl, double* B, i64 nrhs, i64 ldb, bool backward) {
    const int nblk = (int)((n + CH_NB - 1) / CH_NB);
    const unsigned ct = (unsigned)((nrhs + CH_NB - 1) / CH_NB);
    const double* W = c->cholW.p;
    if (!backward) {
        hipLaunchKernelGGL(trsm_step_kernel<false>, dim3(ct, 1), dim3(256), 0, c->stream, L, ldl, W, n, 0, 1, B, nrhs, ldb);
        for (int jb = 0; jb + 1 < nblk; ++jb) {
            const i64 rblocks = nblk - 1 - jb;
            if ((i64)ct * rblocks <= TRSM_FUSED_MAX_TILES) {
                hipLaunchKernelGGL(trsm_step_kernel<false>, dim3(ct, (unsigned)rblocks), dim3(256), 0, c->stream, L, ldl, W, n, jb, 0, B, nrhs, ldb);
            } else {
                const i64 j0 = (i64)jb * CH_NB, rows = n - j0 - CH_NB;
                LRVB_TRY(launch_gemm(c, false, false, rows, nrhs, CH_NB, -1.0, L + (j0 + CH_NB) * ldl + j0, ldl,
                                     B + j0 * ldb, ldb, 1.0, B + (j0 + CH_NB) * ldb, ldb));
                hipLaunchKernelGGL(trsm_step_kernel<false>, dim3(ct, 1), dim3(256), 0, c->stream, L, ldl, W, n, jb + 1, 1, B, nrhs, ldb);
            }
        }
    } else {
        hipLaunchKernelGGL(trsm_step_kernel<true>, dim3(ct, 1), dim3(256), 0, c->stream, L, ldl, W, n, nblk - 1, 1, B, nrhs, ldb);
        for (int jb = nblk - 1; jb >= 1; --jb) {
            if ((i64)ct * jb <= TRSM_FUSED_MAX_TILES) {
                hipLaunchKernelGGL(trsm_step_kernel<true>, dim3(ct, (unsigned)jb), dim3(256), 0, c->stream, L, ldl, W, n, jb, 0, B, nrhs, ldb);
            } else {
                const i64 j0 = (i64)jb * CH_NB;
                const i64 nb = (n - j0 < CH_NB) ? (n - j0) : CH_NB;
                LRVB_TRY(launch_gemm(c, true, false, j0, nrhs, nb, -1.0, L + j0 * ldl, ldl, B + j0 * ldb, ldb, 1.0, B, ldb));
                hipLaunchKernelGGL(trsm_step_kernel<true>, dim3(ct, 1), dim3(256), 0, c->stream, L, ldl, W, n, jb - 1, 1, B, nrhs, ldb);
            }
        }
    }
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// forward substitution only: B <- L^-1 B
int launch_trsm_lower_forward(lrvb_ctx* c, const double* L, i64 n, i64 ldl, double* B, i64 nrhs, i64 ldb) {
    if (n <= 0 || nrhs <= 0) return LRVB_OK;
    return trsm_chain(c, L, n, ldl, B, nrhs, ldb, false);
}

int launch_potrs_lower(lrvb_ctx* c, const double* L, i64 n, i64 ldl, double* B, i64 nrhs, i64 ldb) {
    if (n <= 0 || nrhs <= 0) return LRVB_OK;
    LRVB_TRY(trsm_chain(c, L, n, ldl, B, nrhs, ldb, false));      // forward: L Y = B
    return trsm_chain(c, L, n, ldl, B, nrhs, ldb, true);         // backward: L^T X = Y
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

// three dot products in one launch (one workgroup, fixed reduction tree: deterministic): out[k] = a_k . b_k
__global__ __launch_bounds__(1024)
void dot3_kernel(const double* __restrict__ a0, const double* __restrict__ b0, const double* __restrict__ a1,
                 const double* __restrict__ b1, const double* __restrict__ a2, const double* __restrict__ b2,
                 i64 n, double* __restrict__ out)
{
    __shared__ double sh[3][1024];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (i64 i = threadIdx.x; i < n; i += 1024) { s0 += a0[i] * b0[i]; s1 += a1[i] * b1[i]; s2 += a2[i] * b2[i]; }
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + off];
            sh[2][threadIdx.x] += sh[2][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) out[threadIdx.x] = sh[threadIdx.x][0];
}
int launch_dot3(lrvb_ctx* c, const double* a0, const double* b0, const double* a1, const double* b1,
                const double* a2, const double* b2, i64 n, double* out3_dev) {
    hipLaunchKernelGGL(dot3_kernel, dim3(1), dim3(1024), 0, c->stream, a0, b0, a1, b1, a2, b2, n, out3_dev);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// conjugate-gradient update in one launch: z += alpha d, r += alpha q, out = [r . r, z . z] of the updated vectors
__global__ __launch_bounds__(1024)
void cg_update_kernel(i64 n, double alpha, const double* __restrict__ d, const double* __restrict__ q,
                      double* __restrict__ z, double* __restrict__ r, double* __restrict__ out)
{
    __shared__ double sh[2][1024];
    double s0 = 0.0, s1 = 0.0;
    for (i64 i = threadIdx.x; i < n; i += 1024) {
        const double zi = z[i] + alpha * d[i], ri = r[i] + alpha * q[i];
        z[i] = zi; r[i] = ri;
        s0 += ri * ri; s1 += zi * zi;
    }
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x < 2) out[threadIdx.x] = sh[threadIdx.x][0];
}
int launch_cg_update(lrvb_ctx* c, i64 n, double alpha, const double* d, const double* q, double* z, double* r, double* out2_dev) {
    hipLaunchKernelGGL(cg_update_kernel, dim3(1), dim3(1024), 0, c->stream, n, alpha, d, q, z, r, out2_dev);
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
    if (!trans) { for (i64 k = lane; k < N; k += 64) s += __builtin_nontemporal_load(A + o * lda + k) * x[k]; }
    else        { for (i64 k = lane; k < M; k += 64) s += A[k * lda + o] * x[k]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) y[o] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[o]);
}
// Transposed product of a TALL matrix (M rows >> N columns): y[j] = alpha sum_k A[k][j] x[k] + beta y[j].  One wave per
// output walking a column touches a new cache line with every element (8.9 ms for 1e6 x 256); here a workgroup owns a
// range of rows, its four waves read whole 512-byte row pieces, and the workgroup partials are added in a fixed order.
__global__ __launch_bounds__(256)
void gemv_t_tall_kernel(i64 M, i64 N, const double* __restrict__ A, i64 lda, const double* __restrict__ x,
                        i64 rows_per_block, double* __restrict__ part /* [gridDim.x][N] */)
{
    // thread t owns column t (+ 256, + 512, ...) for the workgroup's rows: the four waves read one whole 2 KiB piece of a
    // row together, eight rows in flight per thread, and no cross-thread reduction is needed inside the workgroup
    const i64 r0 = (i64)blockIdx.x * rows_per_block;
    i64 r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    for (i64 col = threadIdx.x; col < N; col += 256) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        i64 k = r0;
        for (; k + 7 < r1; k += 8) {
            const double a0 = __builtin_nontemporal_load(A + k * lda + col), a1 = __builtin_nontemporal_load(A + (k + 1) * lda + col);
            const double a2 = __builtin_nontemporal_load(A + (k + 2) * lda + col), a3 = __builtin_nontemporal_load(A + (k + 3) * lda + col);
            const double a4 = __builtin_nontemporal_load(A + (k + 4) * lda + col), a5 = __builtin_nontemporal_load(A + (k + 5) * lda + col);
            const double a6 = __builtin_nontemporal_load(A + (k + 6) * lda + col), a7 = __builtin_nontemporal_load(A + (k + 7) * lda + col);
            s0 += a0 * x[k];     s1 += a1 * x[k + 1]; s2 += a2 * x[k + 2]; s3 += a3 * x[k + 3];
            s0 += a4 * x[k + 4]; s1 += a5 * x[k + 5]; s2 += a6 * x[k + 6]; s3 += a7 * x[k + 7];
        }
        for (; k < r1; ++k) s0 += __builtin_nontemporal_load(A + k * lda + col) * x[k];
        part[(i64)blockIdx.x * N + col] = (s0 + s1) + (s2 + s3);
    }
}
__global__ __launch_bounds__(512)
void gemv_t_tall_reduce_kernel(const double* __restrict__ part, int nblk, i64 N, double alpha, double beta, double* __restrict__ y)
{
    __shared__ double sh[8][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const i64 col = (i64)blockIdx.x * 64 + cx;
    double s0 = 0.0, s1 = 0.0;
    if (col < N) {
        int b = ry;
        for (; b + 8 < nblk; b += 16) { s0 += part[(i64)b * N + col]; s1 += part[(i64)(b + 8) * N + col]; }
        for (; b < nblk; b += 8) s0 += part[(i64)b * N + col];
    }
    sh[ry][cx] = s0 + s1;
    __syncthreads();
    if (ry == 0 && col < N) {
        double t = sh[0][cx];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += sh[g][cx];
        y[col] = alpha * t + (beta == 0.0 ? 0.0 : beta * y[col]);
    }
}

// Product of a TALL matrix with a vector (M rows >> N columns, N even, 16-byte aligned rows): y[r] = alpha A[r,:].x + beta y[r].
// A wave takes four rows at a time and keeps all their 16-byte loads in flight before it reduces (one wave per row
// with one load outstanding ran at 4.5 TB/s at 1e6 x 256); x sits in registers.
template <int NV>       // 16-byte loads per lane and row: N <= 128 NV
__global__ __launch_bounds__(256)
void gemv_n_tall_kernel(i64 M, i64 N, const double* __restrict__ A, i64 lda, const double* __restrict__ x,
                        double alpha, double beta, double* __restrict__ y)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const i64 wave = (i64)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (i64)gridDim.x * 4;
    v2d xv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const i64 col = 2 * (lane + 64 * k);
        xv[k] = (col < N) ? *reinterpret_cast<const v2d*>(x + col) : (v2d){0.0, 0.0};
    }
    for (i64 r0 = wave * 4; r0 < M; r0 += nwaves * 4) {
        v2d a[4][NV];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            i64 r = r0 + rr; if (r > M - 1) r = M - 1;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const i64 col = 2 * (lane + 64 * k);
                a[rr][k] = (col < N) ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(A + r * lda + col)) : (v2d){0.0, 0.0};
            }
        }
        double s[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < NV; ++k) t += a[rr][k][0] * xv[k][0] + a[rr][k][1] * xv[k][1];
            s[rr] = t;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) s[rr] += __shfl_xor(s[rr], off, 64);
        }
        if (lane < 4 && r0 + lane < M) {
            const double v = lane == 0 ? s[0] : lane == 1 ? s[1] : lane == 2 ? s[2] : s[3];
            y[r0 + lane] = alpha * v + (beta == 0.0 ? 0.0 : beta * y[r0 + lane]);
        }
    }
}

int launch_gemv(lrvb_ctx* c, bool trans, i64 M, i64 Nn, double alpha, const double* A, i64 lda,
                const double* x, double beta, double* y) {
    const i64 nout = trans ? Nn : M;
    if (nout <= 0) return LRVB_OK;
    // (both streaming kernels were written for tall matrices and are the right ones for the square D x D and V x D products of
    // the solvers too -- dense preconditioner, packing Jacobian of a PSD / simplex layout, CG on a resident matrix: the
    // one-wave-per-output kernel below took 25-160 us for a 1024 x 1024 product, its transposed form walking columns)
    if (!trans && M >= 256 && Nn <= 1024 && !(Nn & 1) && !(lda & 1) && !(((uintptr_t)A) & 15) && !(((uintptr_t)x) & 15)) {
        const unsigned grid = (unsigned)(M >= 32768 ? 2048 : (M + 15) / 16);
        const int nv = (int)((Nn + 127) / 128);
#define LRVB_GEMV_N(NVV) hipLaunchKernelGGL(gemv_n_tall_kernel<NVV>, dim3(grid), dim3(256), 0, c->stream, M, Nn, A, lda, x, alpha, beta, y)
        if (nv <= 1) LRVB_GEMV_N(1); else if (nv <= 2) LRVB_GEMV_N(2); else if (nv <= 4) LRVB_GEMV_N(4); else LRVB_GEMV_N(8);
#undef LRVB_GEMV_N
        HIP_TRY(hipGetLastError());
        return LRVB_OK;
    }
    if (trans && M >= 64 && Nn <= 8192) {
        i64 nblk = M >= 8192 ? (M + 255) / 256 : (M + 7) / 8;        // a short matrix still gets a few hundred workgroups
        if (nblk > 1024) nblk = 1024;
        i64 rpb = (M + nblk - 1) / nblk;
        rpb = (rpb + 7) & ~(i64)7;
        nblk = (M + rpb - 1) / rpb;
        LRVB_TRY(buf_reserve(c, c->red_scratch, (size_t)(nblk * Nn)));
        hipLaunchKernelGGL(gemv_t_tall_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, M, Nn, A, lda, x, rpb, c->red_scratch.p);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(gemv_t_tall_reduce_kernel, dim3((unsigned)((Nn + 63) / 64)), dim3(512), 0, c->stream,
                           (const double*)c->red_scratch.p, (int)nblk, Nn, alpha, beta, y);
        HIP_TRY(hipGetLastError());
        return LRVB_OK;
    }
    hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)((nout + 3) / 4)), dim3(256), 0, c->stream,
                       trans ? 1 : 0, M, Nn, alpha, A, lda, x, beta, y);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// C (PA x PB) = A^T B for operands of a few hundred rows and columns (A: K x PA, B: K x PB, row-major): 16 x 16 output tiles, the
// contraction staged through LDS 16 rows at a time.  The generic 64 x 64-tile GEMM puts a 254 x 254 product (configuration 2's
// free conversion) on 16 workgroups (22 us); this one fills the chip.
__global__ __launch_bounds__(256)
void gemm_tn_small_kernel(i64 K, i64 PA, i64 PB, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C)
{
    constexpr int KC = 64;                                 // 64 rows of the contraction per stage: four global loads per operand and
    __shared__ double As[KC][17];                          // thread in flight at once (16-row stages were a chain of 16 load latencies)
    __shared__ double Bs[KC][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const i64 i0 = (i64)blockIdx.y * 16, j0 = (i64)blockIdx.x * 16;
    double acc = 0.0;
    for (i64 k0 = 0; k0 < K; k0 += KC) {
        double a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const i64 k = k0 + ty + 16 * r;
            a[r] = (k < K && i0 + tx < PA) ? A[k * PA + i0 + tx] : 0.0;
            b[r] = (k < K && j0 + tx < PB) ? B[k * PB + j0 + tx] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { As[ty + 16 * r][tx] = a[r]; Bs[ty + 16 * r][tx] = b[r]; }
        __syncthreads();
#pragma unroll 16
        for (int kk = 0; kk < KC; ++kk) acc += As[kk][ty] * Bs[kk][tx];
        __syncthreads();
    }
    if (i0 + ty < PA && j0 + tx < PB) C[(i0 + ty) * PB + j0 + tx] = acc;
}
int launch_gemm_tn_small(lrvb_ctx* c, i64 K, i64 PA, i64 PB, const double* A, const double* B, double* C) {
    dim3 grid((unsigned)((PB + 15) / 16), (unsigned)((PA + 15) / 16));
    hipLaunchKernelGGL(gemm_tn_small_kernel, grid, dim3(256), 0, c->stream, K, PA, PB, A, B, C);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
