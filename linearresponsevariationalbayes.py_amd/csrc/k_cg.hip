// k_cg.hip -- the fused head / tail kernels of the blocked conjugate-gradient loop (lrvb_cg_solve_multi).
#include "lrvb_internal.h"
#include "k_kernels.h"
#include <math.h>

// step lengths of the blocked CG on the device: alpha_q = (r.z)_q / (p.Hp)_q for the systems still running (the ones whose
// direction was updated in this iteration: z coefficient 1), 0 for the frozen ones; the coefficient rows of the two updates
// x += alpha p, r -= alpha q are written in place.  Saves the second host round trip of every iteration.
__global__ void cg_multi_alpha_kernel(int Q, double* __restrict__ s) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const bool act = s[4 * Q + q] != 0.0;
    const double alpha = act ? s[2 * Q + q] / s[3 * Q + q] : 0.0;
    s[4 * Q + q] = alpha; s[5 * Q + q] = 1.0; s[6 * Q + q] = -alpha; s[7 * Q + q] = 1.0;
}

// s: [ |b|^2 (Q) | rho_prev (Q) | r.r (Q) | live (Q) | iterations (Q) | live after even iterations (Q) | after odd ones (Q) ].  head: r.r, the convergence test, beta, p = r + beta p
// and the product's operand u = eta' o p in one kernel (one workgroup per system); tail: q = eta' (W + s A u) + g eta'' p
// formed on the fly, p.q, alpha, x += alpha p, r -= alpha q.  The host runs ONE ITERATION AHEAD of its convergence test
// (the status of iteration k is read on a side stream while iteration k + 1 is already queued); when every system has
// stopped, the queued product is skipped on the device by the `live` flags.  Round 2's loop had twelve launches, a
// device-to-host and a host-to-device copy per iteration: ~135 us beside the 1.52 ms product.
__global__ __launch_bounds__(256)
void cg_multi_head_kernel(i64 D, i64 it, double tol, const double* __restrict__ j1, const double* __restrict__ R,
                          double* __restrict__ Pm, double* __restrict__ U, double* __restrict__ s, i64 Q)
{
    __shared__ double sh[256];
    __shared__ double bc[2];
    const i64 q = blockIdx.x;
    const double* r = R + q * D;
    double a = 0.0;
    for (i64 d = threadIdx.x; d < D; d += 256) a += r[d] * r[d];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const double rr = sh[0];
        double live = s[3 * Q + q], beta = 0.0;
        if (live != 0.0) {
            if (sqrt(rr) < tol * sqrt(s[q])) { live = 0.0; s[4 * Q + q] = (double)it; }
            else { const double rp = s[Q + q]; beta = (it > 0 && rp != 0.0) ? rr / rp : 0.0; s[Q + q] = rr; s[4 * Q + q] = (double)(it + 1); }
        }
        s[2 * Q + q] = rr; s[3 * Q + q] = live;
        // the flag of THIS iteration's test, in the slot of its parity: what the host reads back (the live flags themselves are
        // rewritten by the head kernel of iteration it + 1, which may already be queued -- a copy of s[3Q..] could hold either state)
        s[(5 + (it & 1)) * Q + q] = live;
        bc[0] = live; bc[1] = beta;
    }
    __syncthreads();
    const bool live = bc[0] != 0.0;
    const double beta = bc[1];
    double* p = Pm + q * D; double* u = U + q * D;
    for (i64 d = threadIdx.x; d < D; d += 256) {
        const double pv = live ? r[d] + beta * p[d] : p[d];           // a stopped system keeps its direction
        p[d] = pv; u[d] = j1 ? j1[d] * pv : pv;                       // j1 null: the product runs in free coordinates (resident Hessian)
    }
}

__global__ __launch_bounds__(256)
void cg_multi_tail_kernel(i64 D, double sq, const double* __restrict__ quadA /* nullable */, const double* __restrict__ j1,
                          const double* __restrict__ j2, const double* __restrict__ g, const double* __restrict__ W,
                          const double* __restrict__ U, const double* __restrict__ Pm, double* __restrict__ X,
                          double* __restrict__ R, const double* __restrict__ s, i64 Q)
{
    __shared__ double sh[256];
    const i64 q = blockIdx.x;
    if (s[3 * Q + q] == 0.0) return;                                  // stopped: nothing moves
    const double* p = Pm + q * D; const double* w = W + q * D; const double* u = U + q * D;
    auto qv = [&](i64 d) { return j1 ? j1[d] * (w[d] + (quadA ? sq * quadA[d] * u[d] : 0.0)) + g[d] * j2[d] * p[d] : w[d]; };
    double a = 0.0;
    for (i64 d = threadIdx.x; d < D; d += 256) a += p[d] * qv(d);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
    const double alpha = s[2 * Q + q] / sh[0];
    double* x = X + q * D; double* r = R + q * D;
    for (i64 d = threadIdx.x; d < D; d += 256) { x[d] += alpha * p[d]; r[d] -= alpha * qv(d); }
}

// W (Q x D) = U (Q x D) H for a SYMMETRIC resident matrix H (D x D, row-major), Q <= 16 per launch group: the block product of
// the blocked CG when the Hessian of the point is resident.  H is read ONCE (8 MB at D = 1024), in 128-byte row segments: a
// workgroup owns 16 columns, wave w the rows k = 4 w + (lane >> 4) (mod 16), lane & 15 the column; the 16 x 256-row chunk of U the
// rows need is staged in LDS and read as broadcasts.  Per-lane partial sums meet in LDS in a fixed order (no atomics: bitwise
// reproducible).  The generic 64 x 64-tile GEMM put this skinny product on 16 workgroups (~100 us per CG iteration).
__global__ __launch_bounds__(256)
void symm_block_kernel(i64 D, int Q, const double* __restrict__ U, const double* __restrict__ H, double* __restrict__ W)
{
    constexpr int QB = 16, KC = 256;
    __shared__ double Us[QB][KC + 1];
    __shared__ double red[16][QB][17];                    // [row slot of the workgroup][q][column]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, rslot = wave * 4 + (lane >> 4);           // 16 row slots per workgroup
    const i64 j = (i64)blockIdx.x * 16 + col;
    double acc[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) acc[q] = 0.0;
    for (i64 k0 = 0; k0 < D; k0 += KC) {
        __syncthreads();
        for (int e = threadIdx.x; e < QB * KC; e += 256) {
            const int q = e / KC, kk = e - q * KC;
            Us[q][kk] = (q < Q && k0 + kk < D) ? U[(i64)q * D + k0 + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = rslot; kk < KC; kk += 16) {
            const i64 k = k0 + kk;
            const double h = (k < D && j < D) ? H[k * D + j] : 0.0;
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[q] += Us[q][kk] * h;
        }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) red[rslot][q][col] = acc[q];
    __syncthreads();
    {   // 256 threads = 16 q x 16 columns: each sums the 16 row slots in order
        const int q = threadIdx.x >> 4, cc = threadIdx.x & 15;
        const i64 jj = (i64)blockIdx.x * 16 + cc;
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][q][cc];
        if (q < Q && jj < D) W[(i64)q * D + jj] = s;
    }
}
int launch_symm_block(lrvb_ctx* c, i64 Q, i64 D, const double* U, const double* H, double* W) {
    for (i64 q0 = 0; q0 < Q; q0 += 16) {
        const int qn = (int)((Q - q0 < 16) ? Q - q0 : 16);
        hipLaunchKernelGGL(symm_block_kernel, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, c->stream, D, qn, U + q0 * D, H, W + q0 * D);
        HIP_TRY(hipGetLastError());
    }
    return LRVB_OK;
}
