// k_glm.hip -- one streaming pass over the observations: value, gradient, per-observation
// curvature, and the Hessian-vector product of the data term
//
//     f_data(beta) = sum_n w_n loss(y_n, x_n . beta).
//
// Replaces what autograd executes for Objective.fun_free / fun_free_grad / fun_free_hvp
// (LRVB/SparseObjectives.py:120-125, 152-154, 183-187: forward + reverse (+ reverse-over-
// reverse) tape walks over N-sized numpy arrays).  HBM-bound: each row of X is read exactly
// once per pass and kept in registers between the dot product and the rank-1 update:
//
//   wavefront <- 2 rows at a time; lane l holds columns {128 it + 2l, 128 it + 2l + 1};
//   z = x.beta (and t = x.u in HVP mode) by __shfl_xor butterflies;
//   coef = w loss'(y, z)             (PASS_GRAD; also stores loss' and w loss'')
//        = w loss''(y, z) t          (PASS_HVP)
//        = cw_n t                    (PASS_HVP_C, curvature cached by an earlier PASS_GRAD)
//   acc[columns of this lane] += coef * x
//
// Block partials (4 waves combined through LDS) go to part_vec[block][P]; a second kernel
// sums them in a fixed order (deterministic, no atomics).
#include "lrvb_internal.h"

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// DPP moves of a double inside a row of 16 lanes (quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140): the four last steps of a butterfly sum without the LDS crossbar
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// gfx950's lane swaps (checked on the chip by tools/lab/permlane_probe.hip): v_permlane32_swap(a, b) leaves [a.lo, b.lo] and
// [a.hi, b.hi] (halves of 32 lanes), v_permlane16_swap(a, b) leaves [a.r0, b.r0, a.r2, b.r2] and [a.r1, b.r1, a.r3, b.r3] (rows of
// 16 lanes).  The sums of the two results: one exchange step of a butterfly for TWO values at once, without the LDS crossbar.
__device__ __forceinline__ double swap_add32(double a, double b) {       // [a.lo + a.hi | b.lo + b.hi]
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {       // [a.r0 + a.r1 | b.r0 + b.r1 | a.r2 + a.r3 | b.r2 + b.r3]
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// Sums of R <= 4 per-lane values over the wavefront in ONE butterfly: the swap steps exchange different rows in the two
// directions, so the number of live values halves with the lane distance; lane L ends with the wave total of row
// L / (64 / G), G = 1, 2, 4 groups for R = 1, 2, 3..4 (row 3 of R = 3 is zero).  Fixed order of additions.
template <int R>
__device__ __forceinline__ double group_sums(const double (&v)[R], int lane) {
    double s;
    if (R == 1) {
        s = swap_add32(v[0], v[0]);
        s = swap_add16(s, s);
    } else if (R == 2) {
        s = swap_add32(v[0], v[1]);
        s = swap_add16(s, s);
    } else {
        const double p = swap_add16(v[0], v[1]);                           // [a01 | b01 | a23 | b23]
        const double q = swap_add16(v[2], R > 3 ? v[R - 1] : 0.0);         // [c01 | d01 | c23 | d23]
        s = swap_add32(p, q);                                              // [a | b | c | d]
    }
    s += dpp_f64<0x140>(s);          // lane i <-> 15 - i
    s += dpp_f64<0x141>(s);          // i <-> 7 - i of each half row
    s += dpp_f64<0x4E>(s);
    s += dpp_f64<0xB1>(s);
    return s;
}

__device__ __forceinline__ void loss_eval(int loss, double lik_info, double y, double z,
                                          double& l0, double& l1, double& l2) {
    if (loss == LRVB_LOSS_GAUSSIAN) {
        const double d = z - y;
        l0 = 0.5 * lik_info * d * d; l1 = lik_info * d; l2 = lik_info;
    } else if (loss == LRVB_LOSS_LOGISTIC) {
        // softplus(z) - y z, evaluated without overflow
        const double az = fabs(z);
        const double e = exp(-az);
        const double sp = (z > 0.0 ? z : 0.0) + log1p(e);
        const double sig = z >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
        l0 = sp - y * z; l1 = sig - y; l2 = sig * (1.0 - sig);
    } else {   // LRVB_LOSS_POISSON
        const double ez = exp(z);
        l0 = ez - y * z; l1 = ez - y; l2 = ez;
    }
}

template <int NIT, int MODE>
__global__ __launch_bounds__(PASS_THREADS)      // (PASS_THREADS, 2) spills at NIT = 8: 3.1 ms instead of 1.6
void glm_pass_kernel(const double* __restrict__ X, i64 ldx, i64 N, int P,
                     const double* __restrict__ y, const double* __restrict__ w,
                     const double* __restrict__ beta, const double* __restrict__ u,
                     int loss, double lik_info,
                     double* __restrict__ lp_out, double* __restrict__ cw_io,
                     double* __restrict__ part_vec, double* __restrict__ part_val,
                     int vec_ok_i, int store_obs)
{
    // rows per stage: two, except in the mode that needs both coefficient vectors in registers
    constexpr int R = (MODE == PASS_HVP) ? 1 : 2;
    __shared__ double red[3][NIT * 128];
    __shared__ double redv[4];
    const bool vec_ok = vec_ok_i != 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    double bt[NIT][2], ut[NIT][2], acc[NIT][2];
    int colc[NIT];                                  // clamped column of this lane's pair (always readable)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = it * 128 + 2 * lane;
        bt[it][0] = (MODE != PASS_HVP_C && col < P) ? beta[col] : 0.0;
        bt[it][1] = (MODE != PASS_HVP_C && col + 1 < P) ? beta[col + 1] : 0.0;
        ut[it][0] = (MODE != PASS_GRAD && col < P) ? u[col] : 0.0;
        ut[it][1] = (MODE != PASS_GRAD && col + 1 < P) ? u[col + 1] : 0.0;
        acc[it][0] = 0.0; acc[it][1] = 0.0;
        colc[it] = vec_ok ? (col < P ? col : 0) : col;
    }
    double val = 0.0;

    // Branch-free raw loads from clamped addresses (rows past N re-read row N-1 and get a zero
    // coefficient; columns past P meet zero entries of beta / u and are never written back), so
    // every load of a stage is in flight before anything waits.
    // the per-row scalars (y, w or the cached curvature) travel with the stage, so that their latency
    // is hidden behind the previous stage like that of the row itself
    auto load_stage = [&](double (&x)[R][NIT][2], double (&sc)[R][2], i64 base) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            i64 n = base + rr; if (n > N - 1) n = N - 1;
            const double* rowp = X + n * ldx;
            if (MODE == PASS_HVP_C) { sc[rr][0] = cw_io[n]; sc[rr][1] = 0.0; }
            else { sc[rr][0] = y[n]; sc[rr][1] = w[n]; }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (vec_ok) {
                    // X is read once per pass: the streaming (non-temporal) policy keeps it from displacing the block partials
                    // and vectors in L2 -- 1.43 -> 1.34 ms per pass (6.1 TB/s) at N = 1e6, P = 1024
                    typedef double v2d __attribute__((ext_vector_type(2)));
                    const v2d tv = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(rowp + colc[it]));
                    x[rr][it][0] = tv[0]; x[rr][it][1] = tv[1];
                } else {
                    const int c0 = colc[it] < P ? colc[it] : 0, c1 = colc[it] + 1 < P ? colc[it] + 1 : 0;
                    x[rr][it][0] = rowp[c0]; x[rr][it][1] = rowp[c1];
                }
            }
        }
    };
    // The R rows of a stage share ONE butterfly (group_sums: lane L ends with the dot product of row L / GS) and ONE evaluation
    // of the loss terms (row L / GS in lane L: exp / log1p of the logistic and Poisson losses cost more than the row's
    // multiply-adds); the rank-one coefficients come back through scalar registers.
    constexpr int GS = R == 1 ? 64 : 32;
    const int gr = lane / GS;
    const bool glead = (lane & (GS - 1)) == 0;
    auto consume = [&](double (&x)[R][NIT][2], double (&sc)[R][2], i64 base) {
        double zr[R], tr[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            double z = 0.0, tt = 0.0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (MODE != PASS_HVP_C) z += x[rr][it][0] * bt[it][0] + x[rr][it][1] * bt[it][1];
                if (MODE != PASS_GRAD)  tt += x[rr][it][0] * ut[it][0] + x[rr][it][1] * ut[it][1];
            }
            zr[rr] = z; tr[rr] = tt;
        }
        double z = 0.0, tt = 0.0;
        if (MODE != PASS_HVP_C) z = group_sums<R>(zr, lane);
        if (MODE != PASS_GRAD)  tt = group_sums<R>(tr, lane);
        double s0 = sc[0][0], s1 = sc[0][1];
#pragma unroll
        for (int rr = 1; rr < R; ++rr) if (gr == rr) { s0 = sc[rr][0]; s1 = sc[rr][1]; }
        const i64 n = base + gr;
        const bool live = n < N;
        double coef_l;
        if (MODE == PASS_HVP_C) {
            coef_l = s0 * tt;
        } else {
            double l0, l1, l2;
            loss_eval(loss, lik_info, s0, z, l0, l1, l2);
            if (MODE == PASS_GRAD) {
                coef_l = s1 * l1;
                if (live && glead) val += s1 * l0;
                if (store_obs && live && glead) { lp_out[n] = l1; cw_io[n] = s1 * l2; }
            } else {
                coef_l = s1 * l2 * tt;
            }
        }
        if (!live) coef_l = 0.0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const double coef = readlane_f64(coef_l, rr * GS);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                acc[it][0] += coef * x[rr][it][0];
                acc[it][1] += coef * x[rr][it][1];
            }
        }
    };

    // software pipeline: the loads of stage k+1 are issued before stage k is consumed (explicit
    // ping-pong between two statically named register sets)
    const i64 step = (i64)gridDim.x * 4 * R;
    i64 base = ((i64)blockIdx.x * 4 + wave) * R;
    if (base < N) {
        double xa[R][NIT][2], xb[R][NIT][2], sa[R][2], sb[R][2];
        load_stage(xa, sa, base);
        for (;;) {
            i64 nxt = base + step;
            if (nxt < N) load_stage(xb, sb, nxt);
            consume(xa, sa, base);
            if (nxt >= N) break;
            base = nxt; nxt = base + step;
            if (nxt < N) load_stage(xa, sa, nxt);
            consume(xb, sb, base);
            if (nxt >= N) break;
            base = nxt;
        }
    }

    // combine the 4 waves of the block (fixed order: wave 0 + 1 + 2 + 3)
    if (wave > 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            red[wave - 1][it * 128 + 2 * lane] = acc[it][0];
            red[wave - 1][it * 128 + 2 * lane + 1] = acc[it][1];
        }
    }
    {                                    // the group leaders' sums, in row order
        double vt = 0.0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) vt += readlane_f64(val, rr * GS);
        if (lane == 0) redv[wave] = vt;
    }
    __syncthreads();
    if (wave == 0) {
        double* dst = part_vec + (i64)blockIdx.x * P;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = it * 128 + 2 * lane;
            const double s0 = ((acc[it][0] + red[0][col]) + red[1][col]) + red[2][col];
            const double s1 = ((acc[it][1] + red[0][col + 1]) + red[1][col + 1]) + red[2][col + 1];
            if (col < P) dst[col] = s0;
            if (col + 1 < P) dst[col + 1] = s1;
        }
        if (lane == 0 && MODE == PASS_GRAD) part_val[blockIdx.x] = ((redv[0] + redv[1]) + redv[2]) + redv[3];
    }
}

// out[col] = sum_b part_vec[b][col]; value = sum_b part_val[b].  Fixed summation order
// (deterministic): 64 columns x 8 row slices per block, 4 loads in flight per thread, slices
// combined through LDS in slice order.
__global__ __launch_bounds__(512)
void pass_reduce_kernel(const double* __restrict__ part_vec, const double* __restrict__ part_val,
                        int nblk, int P, double* __restrict__ out_vec, double* __restrict__ out_val)
{
    __shared__ double sh[8][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (col < P) {
        int b = ry;
        for (; b + 24 < nblk; b += 32) {
            s0 += part_vec[(i64)b * P + col];
            s1 += part_vec[(i64)(b + 8) * P + col];
            s2 += part_vec[(i64)(b + 16) * P + col];
            s3 += part_vec[(i64)(b + 24) * P + col];
        }
        for (; b < nblk; b += 8) s0 += part_vec[(i64)b * P + col];
    }
    sh[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && col < P) {
        double s = sh[0][cx];
#pragma unroll
        for (int r = 1; r < 8; ++r) s += sh[r][cx];
        out_vec[col] = s;
    }
    if (out_val != nullptr && blockIdx.x == 0) {
        __shared__ double shv[512];
        double s = 0.0;
        for (int b = threadIdx.x; b < nblk; b += 512) s += part_val[b];
        shv[threadIdx.x] = s;
        __syncthreads();
        for (int off = 256; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off) shv[threadIdx.x] += shv[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) *out_val = shv[0];
    }
}

// First level for many block partials: workgroup g adds the ROWS [16 g, 16 g + 16) of part_vec, all P columns -- every
// load a contiguous run of the row (the column-strided walk of pass_reduce_kernel over 2048 rows of 8 KiB touched a new
// page with every load: 32 us for 16 MB, whatever the number of workgroups).  scratch[g][P]; fixed order.
__global__ __launch_bounds__(512)
void pass_reduce_rows_kernel(const double* __restrict__ part_vec, int nblk, int P, double* __restrict__ scratch)
{
    const int b0 = blockIdx.x * 16;
    for (int col = threadIdx.x; col < P; col += 512) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (b0 + k < nblk) ? part_vec[(i64)(b0 + k) * P + col] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
        scratch[(i64)blockIdx.x * P + col] = s;
    }
}

// the fixed-order reduction of nblk block partials (vector part: P columns; value part: out_val != NULL)
static int launch_pass_reduce(lrvb_ctx* c, int nblk, int P, double* out_vec, double* out_val) {
    const int nbx = P > 0 ? (P + 63) / 64 : 1;
    if (P > 0 && nblk >= 256) {
        const int ng = (nblk + 15) / 16;
        LRVB_TRY(buf_reserve(c, c->red_scratch, (size_t)ng * (size_t)P));
        hipLaunchKernelGGL(pass_reduce_rows_kernel, dim3((unsigned)ng), dim3(512), 0, c->stream,
                           (const double*)c->part_vec.p, nblk, P, c->red_scratch.p);
        HIP_TRY(hipGetLastError());
        // second level over the ng group sums; the value partials (one per ORIGINAL block) are summed by its first workgroup
        hipLaunchKernelGGL(pass_reduce_kernel, dim3((unsigned)nbx), dim3(512), 0, c->stream,
                           (const double*)c->red_scratch.p, (const double*)c->part_val.p, ng, P, out_vec, (double*)nullptr);
        HIP_TRY(hipGetLastError());
        if (out_val) {
            hipLaunchKernelGGL(pass_reduce_kernel, dim3(1), dim3(512), 0, c->stream,
                               (const double*)c->part_vec.p, (const double*)c->part_val.p, nblk, 0, out_vec, out_val);
            HIP_TRY(hipGetLastError());
        }
        return LRVB_OK;
    }
    hipLaunchKernelGGL(pass_reduce_kernel, dim3((unsigned)nbx), dim3(512), 0, c->stream,
                       (const double*)c->part_vec.p, (const double*)c->part_val.p, nblk, P, out_vec, out_val);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

template <int NIT>
static int launch_pass_nit(lrvb_ctx* c, PassMode mode, const double* beta, const double* u,
                           int grid, int vec_ok, int store_obs) {
    dim3 g(grid), b(PASS_THREADS);
    switch (mode) {
    case PASS_GRAD:
        hipLaunchKernelGGL((glm_pass_kernel<NIT, PASS_GRAD>), g, b, 0, c->stream, c->X.p, c->P, c->N, (int)c->P,
                           c->y.p, c->w.p, beta, u, c->loss, c->lik_info, c->lp.p, c->cw.p,
                           c->part_vec.p, c->part_val.p, vec_ok, store_obs);
        break;
    case PASS_HVP:
        hipLaunchKernelGGL((glm_pass_kernel<NIT, PASS_HVP>), g, b, 0, c->stream, c->X.p, c->P, c->N, (int)c->P,
                           c->y.p, c->w.p, beta, u, c->loss, c->lik_info, c->lp.p, c->cw.p,
                           c->part_vec.p, c->part_val.p, vec_ok, store_obs);
        break;
    default:
        hipLaunchKernelGGL((glm_pass_kernel<NIT, PASS_HVP_C>), g, b, 0, c->stream, c->X.p, c->P, c->N, (int)c->P,
                           c->y.p, c->w.p, beta, u, c->loss, c->lik_info, c->lp.p, c->cw.p,
                           c->part_vec.p, c->part_val.p, vec_ok, store_obs);
        break;
    }
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}



// ---- wide designs in ONE pass (round 4): 1024 < n_cols <= 4096 --------------------------------------------------------------
// A row of up to 4096 columns (32 KB) does not fit the registers of one wavefront, but it fits those of a WORKGROUP: wave w of
// its NW (two up to 2048 columns, four beyond: the fewer waves share a row, the fewer wait at its barrier) holds the columns
// [w NIT 128, (w + 1) NIT 128) of the stage's R rows, exactly as the narrow kernel holds a whole row.  The dot products are the
// only thing the waves share: each reduces its part for all R rows in one merged butterfly (group_sums), the NW partial sums
// meet in LDS (two slots, alternating by stage: ONE barrier per stage) and are added in wave order, so every wave sees the same
// z; the loss terms are evaluated once per stage (row L / (64 / R) in lane L), the rank-one update runs from the registers again
// and the waves write disjoint columns of the block partial.  X is read once, as for narrow designs (the two-pass route below
// reads it twice).  The grid is the number of workgroups the chip holds at once (wide1_run).
template <int NW, int NIT, int R, int MODE>
__global__ __launch_bounds__(NW * 64)
void glm_pass_wide1_kernel(const double* __restrict__ X, i64 ldx, i64 N, int P,
                           const double* __restrict__ y, const double* __restrict__ w,
                           const double* __restrict__ beta, const double* __restrict__ u,
                           int loss, double lik_info,
                           double* __restrict__ lp_out, double* __restrict__ cw_io,
                           double* __restrict__ part_vec, double* __restrict__ part_val,
                           int vec_ok_i, int store_obs)
{
    __shared__ double zpart[2][R][2][NW];            // [stage parity][row][z | t][wave]
    const bool vec_ok = vec_ok_i != 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int cbase = wave * NIT * 128;

    double bt[NIT][2], ut[NIT][2], acc[NIT][2];
    int colc[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = cbase + it * 128 + 2 * lane;
        bt[it][0] = (MODE != PASS_HVP_C && col < P) ? beta[col] : 0.0;
        bt[it][1] = (MODE != PASS_HVP_C && col + 1 < P) ? beta[col + 1] : 0.0;
        ut[it][0] = (MODE != PASS_GRAD && col < P) ? u[col] : 0.0;
        ut[it][1] = (MODE != PASS_GRAD && col + 1 < P) ? u[col + 1] : 0.0;
        acc[it][0] = 0.0; acc[it][1] = 0.0;
        colc[it] = col < P ? col : 0;                // clamped: always readable (the product with a zero of beta / u discards it)
    }
    double val = 0.0;

    auto load_stage = [&](double (&x)[R][NIT][2], double (&sc)[R][2], i64 base) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            i64 n = base + rr; if (n > N - 1) n = N - 1;
            const double* rowp = X + n * ldx;
            if (MODE == PASS_HVP_C) { sc[rr][0] = cw_io[n]; sc[rr][1] = 0.0; }
            else { sc[rr][0] = y[n]; sc[rr][1] = w[n]; }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (vec_ok) {
                    typedef double v2d __attribute__((ext_vector_type(2)));
                    const v2d tv = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(rowp + colc[it]));
                    x[rr][it][0] = tv[0]; x[rr][it][1] = tv[1];
                } else {
                    const int c1 = colc[it] + 1 < P ? colc[it] + 1 : 0;
                    x[rr][it][0] = rowp[colc[it]]; x[rr][it][1] = rowp[c1];
                }
            }
        }
    };
    // lane groups of the merged butterfly: lane L works for row L / GS of the stage after it
    constexpr int GS = R == 1 ? 64 : (R == 2 ? 32 : 16);
    const int grp = lane / GS;
    const int gr = grp < R ? grp : 0;
    const bool glead = (lane & (GS - 1)) == 0 && grp < R;
    auto consume = [&](double (&x)[R][NIT][2], double (&sc)[R][2], i64 base, int parity) {
        // this wave's quarter of the dot products, all R rows in one butterfly
        double zr[R], tr[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            double z = 0.0, tt = 0.0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (MODE != PASS_HVP_C) z += x[rr][it][0] * bt[it][0] + x[rr][it][1] * bt[it][1];
                if (MODE != PASS_GRAD)  tt += x[rr][it][0] * ut[it][0] + x[rr][it][1] * ut[it][1];
            }
            zr[rr] = z; tr[rr] = tt;
        }
        double zs = 0.0, ts = 0.0;
        if (MODE != PASS_HVP_C) zs = group_sums<R>(zr, lane);
        if (MODE != PASS_GRAD)  ts = group_sums<R>(tr, lane);
        if (glead) { zpart[parity][gr][0][wave] = zs; zpart[parity][gr][1][wave] = ts; }
        __syncthreads();                              // the only barrier of the stage (the other parity's slots are free until the next one)
        // the loss terms of the R rows are evaluated ONCE per wave, row L / GS in lane L (exp / log1p of the logistic and
        // Poisson losses cost more than the row's multiply-adds), then broadcast through scalar registers
        double z = zpart[parity][gr][0][0], tt = zpart[parity][gr][1][0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) { z += zpart[parity][gr][0][ww]; tt += zpart[parity][gr][1][ww]; }     // in wave order
        double s0 = sc[0][0], s1 = sc[0][1];
#pragma unroll
        for (int rr = 1; rr < R; ++rr) if (gr == rr) { s0 = sc[rr][0]; s1 = sc[rr][1]; }
        const i64 n = base + gr;
        const bool live = n < N;
        double coef_l;
        if (MODE == PASS_HVP_C) {
            coef_l = s0 * tt;
        } else {
            double l0, l1, l2;
            loss_eval(loss, lik_info, s0, z, l0, l1, l2);
            if (MODE == PASS_GRAD) {
                coef_l = s1 * l1;
                if (live && glead) val += s1 * l0;
                if (store_obs && live && glead && wave == 0) { lp_out[n] = l1; cw_io[n] = s1 * l2; }
            } else {
                coef_l = s1 * l2 * tt;
            }
        }
        if (!live) coef_l = 0.0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const double coef = readlane_f64(coef_l, rr * GS);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                acc[it][0] += coef * x[rr][it][0];
                acc[it][1] += coef * x[rr][it][1];
            }
        }
    };

    // every workgroup runs the SAME number of stages (the barrier is workgroup-wide): rows past N are dead slots
    const i64 step = (i64)gridDim.x * R;
    const i64 n_stages = (N + step - 1) / step;
    i64 base = (i64)blockIdx.x * R;
    {
        double xa[R][NIT][2], xb[R][NIT][2], sa[R][2], sb[R][2];
        load_stage(xa, sa, base);
        for (i64 s = 0; s < n_stages; s += 2) {
            if (s + 1 < n_stages) load_stage(xb, sb, base + step);
            consume(xa, sa, base, 0);
            if (s + 1 >= n_stages) break;
            base += step;
            if (s + 2 < n_stages) load_stage(xa, sa, base + step);
            consume(xb, sb, base, 1);
            base += step;
        }
    }
    double* dst = part_vec + (i64)blockIdx.x * P;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = cbase + it * 128 + 2 * lane;
        if (col < P) dst[col] = acc[it][0];
        if (col + 1 < P) dst[col + 1] = acc[it][1];
    }
    if (MODE == PASS_GRAD) {                            // the group leaders' sums, in row order (every wave holds the same ones)
        double vt = 0.0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) vt += readlane_f64(val, rr * GS);
        if (tid == 0) part_val[blockIdx.x] = vt;
    }
}

// One launch of the one-pass kernel with R rows per stage.  The grid is the number of workgroups the chip holds at once
// (occupancy x CUs, asked of the runtime once per instantiation): every workgroup runs the same number of stages, so a grid
// above the resident count would run its surplus as a second, partly filled round (NIT = 3 at 161 registers: 768 of 1024
// workgroups resident, 4.2 TB/s).
template <int NW, int NIT, int R>
static int wide1_run(lrvb_ctx* c, PassMode mode, const double* beta, const double* u,
                     double* out_vec_P, double* value_out_dev, bool store_obs) {
    static int slots[3] = {0, 0, 0};
    const int mi = (int)mode;
    if (!slots[mi]) {
        const void* k = mode == PASS_GRAD ? (const void*)glm_pass_wide1_kernel<NW, NIT, R, PASS_GRAD>
                      : mode == PASS_HVP  ? (const void*)glm_pass_wide1_kernel<NW, NIT, R, PASS_HVP>
                                          : (const void*)glm_pass_wide1_kernel<NW, NIT, R, PASS_HVP_C>;
        int per_cu = 0, n_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, NW * 64, 0));
        HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device));
        if (per_cu < 1) per_cu = 1;
        if (n_cu < 1) n_cu = 256;
        slots[mi] = per_cu * n_cu;
    }
    i64 grid = (c->N + R - 1) / R;
    if (grid > slots[mi]) grid = slots[mi];
    if (grid < 1) grid = 1;
    LRVB_TRY(buf_reserve(c, c->part_vec, (size_t)(grid * c->P)));
    LRVB_TRY(buf_reserve(c, c->part_val, (size_t)grid));
    LRVB_TRY(buf_reserve(c, c->lp, (size_t)c->N));
    LRVB_TRY(reserve_obs_vec(c, c->cw));
    const int vec_ok = ((c->P % 2) == 0) && ((((uintptr_t)c->X.p) & 15) == 0);
    const int so = store_obs ? 1 : 0;
    if (c->prof_on && mode == PASS_GRAD) LRVB_TRY(prof_mark(c, PROF_PASS));
    dim3 g((unsigned)grid), b(NW * 64);
#define WIDE1_LAUNCH(M) hipLaunchKernelGGL((glm_pass_wide1_kernel<NW, NIT, R, M>), g, b, 0, c->stream, c->X.p, c->P, c->N, (int)c->P, \
        c->y.p, c->w.p, beta, u, c->loss, c->lik_info, c->lp.p, c->cw.p, c->part_vec.p, c->part_val.p, vec_ok, so)
    if (mode == PASS_GRAD) WIDE1_LAUNCH(PASS_GRAD); else if (mode == PASS_HVP) WIDE1_LAUNCH(PASS_HVP); else WIDE1_LAUNCH(PASS_HVP_C);
#undef WIDE1_LAUNCH
    HIP_TRY(hipGetLastError());
    if (c->prof_on && mode == PASS_GRAD) LRVB_TRY(prof_mark(c, PROF_PASS));
    LRVB_TRY(launch_pass_reduce(c, (int)grid, (int)c->P, out_vec_P, (mode == PASS_GRAD) ? value_out_dev : nullptr));
    if (c->prof_on && mode == PASS_GRAD) c->prof.pass_bytes = 8.0 * (double)c->N * (double)(c->P + 3);
    return LRVB_OK;
}
static int launch_glm_pass_wide1(lrvb_ctx* c, PassMode mode, const double* beta, const double* u,
                                 double* out_vec_P, double* value_out_dev, bool store_obs) {
    // NW waves x NIT x 128 columns: the smallest shape that covers the row (idle lanes still issue their clamped loads); the
    // fewer waves share a row, the fewer wait at its barrier and the more doubles of X each lane has in flight
    if (c->P <= 1536) return wide1_run<2, 6, 2>(c, mode, beta, u, out_vec_P, value_out_dev, store_obs);
    if (c->P <= 2048) return wide1_run<2, 8, 2>(c, mode, beta, u, out_vec_P, value_out_dev, store_obs);
    if (c->P <= 3072) return wide1_run<4, 6, 2>(c, mode, beta, u, out_vec_P, value_out_dev, store_obs);
    return wide1_run<4, 8, 2>(c, mode, beta, u, out_vec_P, value_out_dev, store_obs);
}

// ---- designs wider than 4096 columns: the row no longer fits in the registers of one workgroup ----
// (and, under tuning bit 0, every design wider than 1024 columns: the route round 3 had for them)
// Two passes over X instead of one: (1) one wavefront per row streams the row in 128-column chunks and
// reduces the dot product(s), evaluates the loss terms and leaves the rank-one coefficient of the row in
// coef[n]; (2) one workgroup per (row block, 128-column tile) accumulates sum_n coef_n x_n over its rows.
// Same outputs, same fixed-order reductions as the fused kernel; twice its traffic.
constexpr int WIDE_ROWS = 2048;          // rows per block of the accumulation pass

template <int MODE>
__global__ __launch_bounds__(256)
void glm_wide_rows_kernel(const double* __restrict__ X, i64 ldx, i64 N, int P,
                          const double* __restrict__ y, const double* __restrict__ w,
                          const double* __restrict__ beta, const double* __restrict__ u,
                          int loss, double lik_info, double* __restrict__ lp_out, double* __restrict__ cw_io,
                          double* __restrict__ coef_out, double* __restrict__ part_val, int store_obs)
{
    __shared__ double redv[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double val = 0.0;
    for (i64 n = (i64)blockIdx.x * 4 + wave; n < N; n += (i64)gridDim.x * 4) {
        const double* rowp = X + n * ldx;
        double z = 0.0, tt = 0.0;
        for (int c0 = 0; c0 < P; c0 += 128) {
            const int col = c0 + 2 * lane;
            const double x0 = (col < P) ? rowp[col] : 0.0, x1 = (col + 1 < P) ? rowp[col + 1] : 0.0;
            if (MODE != PASS_HVP_C) z += x0 * ((col < P) ? beta[col] : 0.0) + x1 * ((col + 1 < P) ? beta[col + 1] : 0.0);
            if (MODE != PASS_GRAD)  tt += x0 * ((col < P) ? u[col] : 0.0) + x1 * ((col + 1 < P) ? u[col + 1] : 0.0);
        }
        if (MODE != PASS_HVP_C) z = wave_sum(z);
        if (MODE != PASS_GRAD)  tt = wave_sum(tt);
        double coef;
        if (MODE == PASS_HVP_C) {
            coef = cw_io[n] * tt;
        } else {
            double l0, l1, l2;
            loss_eval(loss, lik_info, y[n], z, l0, l1, l2);
            const double wn = w[n];
            if (MODE == PASS_GRAD) {
                coef = wn * l1;
                val += wn * l0;
                if (store_obs && lane == 0) { lp_out[n] = l1; cw_io[n] = wn * l2; }
            } else {
                coef = wn * l2 * tt;
            }
        }
        if (lane == 0) coef_out[n] = coef;
    }
    if (lane == 0) redv[wave] = val;
    __syncthreads();
    if (tid == 0 && MODE == PASS_GRAD) part_val[blockIdx.x] = ((redv[0] + redv[1]) + redv[2]) + redv[3];
}

// part_vec[row block][P] (+)= sum over the block's rows of coef_n x_n, one 128-column tile per workgroup
__global__ __launch_bounds__(256)
void glm_wide_accum_kernel(const double* __restrict__ X, i64 ldx, i64 N, int P, const double* __restrict__ coef,
                           double* __restrict__ part_vec)
{
    __shared__ double red[3][128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = blockIdx.x * 128 + 2 * lane;
    const i64 r0 = (i64)blockIdx.y * WIDE_ROWS;
    i64 r1 = r0 + WIDE_ROWS; if (r1 > N) r1 = N;
    double a0 = 0.0, a1 = 0.0;
    for (i64 n = r0 + wave; n < r1; n += 4) {
        const double cf = coef[n];
        const double* rowp = X + n * ldx;
        a0 += cf * ((col < P) ? rowp[col] : 0.0);
        a1 += cf * ((col + 1 < P) ? rowp[col + 1] : 0.0);
    }
    if (wave > 0) { red[wave - 1][2 * lane] = a0; red[wave - 1][2 * lane + 1] = a1; }
    __syncthreads();
    if (wave == 0) {
        double* dst = part_vec + (i64)blockIdx.y * P;
        const double s0 = ((a0 + red[0][2 * lane]) + red[1][2 * lane]) + red[2][2 * lane];
        const double s1 = ((a1 + red[0][2 * lane + 1]) + red[1][2 * lane + 1]) + red[2][2 * lane + 1];
        if (col < P) dst[col] = s0;
        if (col + 1 < P) dst[col + 1] = s1;
    }
}

static int launch_glm_pass_wide(lrvb_ctx* c, PassMode mode, const double* beta, const double* u,
                                double* out_vec_P, double* value_out_dev, bool store_obs) {
    i64 grid1 = (c->N + 3) / 4;
    if (grid1 > 4096) grid1 = 4096;
    const i64 nblk = (c->N + WIDE_ROWS - 1) / WIDE_ROWS;
    LRVB_TRY(buf_reserve(c, c->part_vec, (size_t)(nblk * c->P)));
    LRVB_TRY(buf_reserve(c, c->part_val, (size_t)grid1));
    LRVB_TRY(buf_reserve(c, c->lp, (size_t)c->N));
    LRVB_TRY(reserve_obs_vec(c, c->cw));
    LRVB_TRY(reserve_obs_vec(c, c->zbuf));                  // coef
    if (c->prof_on && mode == PASS_GRAD) LRVB_TRY(prof_mark(c, PROF_PASS));
    const int so = store_obs ? 1 : 0;
    dim3 g1((unsigned)grid1), b(256);
#define WIDE_ROWS_LAUNCH(M) hipLaunchKernelGGL((glm_wide_rows_kernel<M>), g1, b, 0, c->stream, c->X.p, c->P, c->N, (int)c->P, \
        c->y.p, c->w.p, beta, u, c->loss, c->lik_info, c->lp.p, c->cw.p, c->zbuf.p, c->part_val.p, so)
    if (mode == PASS_GRAD) WIDE_ROWS_LAUNCH(PASS_GRAD); else if (mode == PASS_HVP) WIDE_ROWS_LAUNCH(PASS_HVP); else WIDE_ROWS_LAUNCH(PASS_HVP_C);
#undef WIDE_ROWS_LAUNCH
    HIP_TRY(hipGetLastError());
    dim3 g2((unsigned)((c->P + 127) / 128), (unsigned)nblk);
    hipLaunchKernelGGL(glm_wide_accum_kernel, g2, b, 0, c->stream, c->X.p, c->P, c->N, (int)c->P, c->zbuf.p, c->part_vec.p);
    HIP_TRY(hipGetLastError());
    if (c->prof_on && mode == PASS_GRAD) LRVB_TRY(prof_mark(c, PROF_PASS));
    // the value partials are per block of the FIRST kernel, the vector partials per row block of the second:
    // two calls of the fixed-order reduction
    LRVB_TRY(launch_pass_reduce(c, (int)nblk, (int)c->P, out_vec_P, nullptr));
    if (mode == PASS_GRAD && value_out_dev)
        LRVB_TRY(launch_pass_reduce(c, (int)grid1, 0, out_vec_P, value_out_dev));
    if (c->prof_on && mode == PASS_GRAD) c->prof.pass_bytes = 2.0 * 8.0 * (double)c->N * (double)(c->P + 3);
    return LRVB_OK;
}

int launch_glm_pass(lrvb_ctx* c, PassMode mode, const double* beta_dev, const double* u_dev,
                    double* out_vec_P, double* value_out_dev, bool store_obs) {
    if (c->P > 4 * PASS_MAX_COLS) return launch_glm_pass_wide(c, mode, beta_dev, u_dev, out_vec_P, value_out_dev, store_obs);
    if (c->P > PASS_MAX_COLS && !c->force_generic_wsyrk)      // (the tuning bit that forces the generic kernels keeps the two-pass route testable)
        return launch_glm_pass_wide1(c, mode, beta_dev, u_dev, out_vec_P, value_out_dev, store_obs);
    if (c->P > PASS_MAX_COLS) return launch_glm_pass_wide(c, mode, beta_dev, u_dev, out_vec_P, value_out_dev, store_obs);
    // 8 blocks per CU worth of row pairs, capped by the work available
    const i64 rows_per_stage = (mode == PASS_HVP) ? 1 : 2;
    i64 pairs = (c->N + rows_per_stage - 1) / rows_per_stage;
    i64 grid = (pairs + 3) / 4;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    LRVB_TRY(buf_reserve(c, c->part_vec, (size_t)(grid * c->P)));
    LRVB_TRY(buf_reserve(c, c->part_val, (size_t)grid));
    LRVB_TRY(buf_reserve(c, c->lp, (size_t)c->N));
    LRVB_TRY(reserve_obs_vec(c, c->cw));
    const int vec_ok = ((c->P % 2) == 0) && ((((uintptr_t)c->X.p) & 15) == 0);
    if (c->prof_on && mode == PASS_GRAD) LRVB_TRY(prof_mark(c, PROF_PASS));
    int st;
    const int so = store_obs ? 1 : 0;
    if (c->P <= 128)      st = launch_pass_nit<1>(c, mode, beta_dev, u_dev, (int)grid, vec_ok, so);
    else if (c->P <= 256) st = launch_pass_nit<2>(c, mode, beta_dev, u_dev, (int)grid, vec_ok, so);
    else if (c->P <= 512) st = launch_pass_nit<4>(c, mode, beta_dev, u_dev, (int)grid, vec_ok, so);
    else                  st = launch_pass_nit<8>(c, mode, beta_dev, u_dev, (int)grid, vec_ok, so);
    LRVB_TRY(st);
    if (c->prof_on && mode == PASS_GRAD) LRVB_TRY(prof_mark(c, PROF_PASS));
    LRVB_TRY(launch_pass_reduce(c, (int)grid, (int)c->P, out_vec_P, (mode == PASS_GRAD) ? value_out_dev : nullptr));
    if (c->prof_on && mode == PASS_GRAD) {
        c->prof.pass_bytes = 8.0 * (double)c->N * (double)(c->P + 3);
    }
    return LRVB_OK;
}

// Rows n0..n1 of the per-observation gradient matrix in VECTOR coordinates of the
// coefficient slice:  Gv[n, :] = loss'(y_n, z_n) * x_n   (the caller applies J).
// For an all-box layout J is diagonal and is fused here: G[n, j] = lp_n x_nj j1[glm_off + j],
// written into a (n1-n0) x D matrix whose other columns are zero.
__global__ __launch_bounds__(256)
void obs_grad_box_kernel(const double* __restrict__ X, i64 ldx, int P, const double* __restrict__ lp,
                         const double* __restrict__ j1, i64 glm_off, i64 D, i64 n0, i64 n1,
                         double* __restrict__ G)
{
    const i64 n = n0 + blockIdx.y;
    if (n >= n1) return;
    const double lpn = lp[n];
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < D; j += (i64)gridDim.x * blockDim.x) {
        const i64 jc = j - glm_off;
        double v = 0.0;
        if (jc >= 0 && jc < P) v = lpn * X[n * ldx + jc] * j1[j];
        G[(n - n0) * D + j] = v;
    }
}

__global__ __launch_bounds__(256)
void obs_grad_vec_kernel(const double* __restrict__ X, i64 ldx, int P, const double* __restrict__ lp,
                         i64 n0, i64 n1, double* __restrict__ Gv)
{
    const i64 n = n0 + blockIdx.y;
    if (n >= n1) return;
    const double lpn = lp[n];
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < P; j += (i64)gridDim.x * blockDim.x)
        Gv[(n - n0) * P + j] = lpn * X[n * ldx + j];
}

int launch_obs_grad(lrvb_ctx* c, i64 n0, i64 n1, double* G_dev, int mode, const double* scale_vec) {
    // mode 0: (n1-n0) x D, columns scaled by scale_vec (= j1 of an all-box layout, or ones with D := V)
    // mode 1: (n1-n0) x P raw coefficient-slice columns
    if (n1 <= n0) return LRVB_OK;
    const i64 rows = n1 - n0;
    if (rows > 65535) LRVB_FAIL(LRVB_ERR_INVALID, "obs_grad: at most 65535 rows per call (got %lld)", (long long)rows);
    if (mode == 0) {
        const i64 width = c->all_box && scale_vec == c->j1.p ? c->D : c->V;
        dim3 grid((unsigned)((width + 255) / 256), (unsigned)rows);
        hipLaunchKernelGGL(obs_grad_box_kernel, grid, dim3(256), 0, c->stream, c->X.p, c->P, (int)c->P,
                           c->lp.p, scale_vec, c->glm_off, width, n0, n1, G_dev);
    } else {
        dim3 grid((unsigned)((c->P + 255) / 256), (unsigned)rows);
        hipLaunchKernelGGL(obs_grad_vec_kernel, grid, dim3(256), 0, c->stream, c->X.p, c->P, (int)c->P,
                           c->lp.p, n0, n1, G_dev);
    }
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
