// k_hvp_multi.hip -- R = X^T diag(c) X U for up to 16 vectors at once, X read ONCE.
//
// The blocked conjugate-gradient solver (lrvb_cg_solve_multi; the loop over masks of
// LRVB/ConjugateGradient.py:87-105) needs Q Hessian-vector products per iteration.  One after the other
// they cost Q fused passes over X; as two skinny GEMMs (T = X U, then X^T diag(c) T) two passes on
// generic kernels.  Here both contractions happen on a row chunk while it sits in LDS:
//
//   chunk = 8 observations x P columns, staged by LDS-DMA (double buffered, 2 x 64 KiB at P = 1024)
//   step A:  T (8 x 16)   = X_chunk U          contraction over COLUMNS:  A[i = row][k = col], B = U slice
//   step B:  R (P x 16)  += X_chunk^T (c o T)  contraction over ROWS:     A[i = col][k = row], B = c o T
//
// The two steps want the chunk in transposed register layouts -- LDS is the transposer: both read the same
// staged rows with different address patterns (16-byte reads, conflict-free with a row stride of P + 2).
// Wave w owns the column quarter [w P/4, (w+1) P/4): its slice of U stays in registers for the whole kernel
// (the B operand of step A), and so do its 16-row tiles of R (the accumulators of step B); the four partial
// T tiles meet in LDS once per chunk, and the sum comes back in exactly the register layout step B needs
// as its B operand (D register r <-> row (lane >> 4) + 4 r).  Even P <= 1024 (columns are padded to whole
// 128-column DMA instructions inside LDS; the padding meets zeros of U), Q <= 16.
#include "lrvb_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define HM_GLDS16(gp, lp) __builtin_amdgcn_global_load_lds( \
    (const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

// scalar-base form: global address = sbase (SGPR pair) + voff (32-bit per-lane byte offset), LDS destination base in M0
#define HM_GLDS16_S(sbase, voff, ldsaddr) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")

#define HM_GLDS4_S(sbase, voff, ldsaddr) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")

constexpr int HM_ROWS = 8;               // observations per chunk
constexpr int HM_REGIONS = 1;           // chunk order: one chip-wide window (regions of 8 / 32 / 64 workgroups measured: no effect)

// TONLY = true stops after step A and writes the scaled rows of T instead (out[n][q] = c_n (X U)[n][q]):
// the streamed weight-sensitivity product of lrvb_obs_influence.
// NW waves per workgroup (4, or 8 = two per SIMD where the column count allows it: while one wave sits in a barrier or
// waits for LDS the other keeps the MFMA pipe busy).
template <int NB, bool TONLY, int NW>    // NB = P / 128
__global__ __launch_bounds__(64 * NW, 1)
void hvp_multi_kernel(const double* __restrict__ X, int Preal, i64 N, const double* __restrict__ cw,
                      const double* __restrict__ U, i64 ldu, int Q, double* __restrict__ Rpart,
                      double* __restrict__ Tout, i64 ldt, const double* __restrict__ live /* nullable: Q flags */)
{
    if (live) {                            // the blocked CG queues one iteration ahead of its convergence test: when every
        bool any = false;                  // system of this block has stopped, the product is skipped on the device
        for (int q = 0; q < Q; ++q) any = any || (live[q] != 0.0);
        if (!any) return;
    }
    constexpr int P = NB * 128;           // columns rounded up to whole 128-column DMA instructions (Preal is even)
    constexpr int PW = P / NW;            // columns per wave (a multiple of 32)
    constexpr int NBW = PW / 32;          // 32-column blocks per wave
    constexpr int RPW = HM_ROWS / NW;     // rows of a chunk staged by one wave (2 or 1)
    constexpr int STRIDE = P + 2;         // doubles; (STRIDE / 2) odd -> rows land on distinct 16-byte bank groups
    constexpr int NT = PW / 16;           // 16-column tiles of R per wave
    extern __shared__ double lds[];       // [2][HM_ROWS][STRIDE] chunk buffers | [2][NW][2][64] partial T tiles (two slots)
    double* Tpart = lds + 2 * HM_ROWS * STRIDE;
    double* Cst = Tpart + 2 * NW * 2 * 64;           // [2][HM_ROWS] weights of the staged chunks (they ride with the stage's DMA)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int pc0 = wave * PW;

    // this wave's slice of U as MFMA B operands: block b (32 columns), sub-step t: k = pc0 + 32 b + 8 l4 + t, q = l15
    double uf[NBW][8];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int k = pc0 + 32 * b + 8 * l4 + t;
            uf[b][t] = (l15 < Q && k < Preal) ? U[(i64)l15 * ldu + k] : 0.0;        // padding columns carry a zero vector
        }

    d4 acc[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) acc[m] = (d4){0.0, 0.0, 0.0, 0.0};

    const i64 nchunks = (N + HM_ROWS - 1) / HM_ROWS;
    // LDS-DMA addresses: wave-uniform row base in SGPRs + a per-lane byte offset that never changes, so staging a
    // chunk issues no vector-ALU instruction (beside fp64 MFMAs every VALU instruction costs the SIMD ~8 cycles of
    // matrix time: tools/mfma_vmem_probe.hip)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
    unsigned voff[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int col = 128 * j + 2 * lane; if (col > Preal - 2) col = Preal - 2;     // stay inside the row; the duplicates meet zeros of U
        voff[j] = (unsigned)col * 8u;
    }
    auto issue = [&](i64 ch, int buf) {
        // wave w stages RPW rows of the chunk: P / 128 instructions of 1 KiB per row
        const unsigned base = lds0 + (unsigned)(buf * (HM_ROWS * STRIDE)) * 8u;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = RPW * wave + rr;
            i64 n = ch * HM_ROWS + row; if (n > N - 1) n = N - 1;       // rows past N carry weight zero (wave-uniform clamp)
            const char* rowp = reinterpret_cast<const char*>(X + n * (i64)Preal);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                HM_GLDS16_S(rowp, voff[j], base + (unsigned)(row * STRIDE + 128 * j) * 8u);
        }
        // the chunk's 8 weights (64 bytes) travel with it: loading them at the top of the iteration that uses them put a
        // fresh global load in front of the vmcnt(0) wait -- its whole latency, every chunk
        if (wave == 0 && lane < 16)
            HM_GLDS4_S(reinterpret_cast<const char*>(cw + ch * HM_ROWS), (unsigned)lane * 4u,
                       lds0 + (unsigned)(2 * HM_ROWS * STRIDE + 2 * NW * 2 * 64 + buf * HM_ROWS) * 8u);
    };

    // Chunk order: the workgroups form HM_REGIONS groups, each streaming its own contiguous part of the rows, the workgroups
    // of a group taking its chunks round-robin (HM_REGIONS = 1: one chip-wide window).
    const int nreg = HM_REGIONS <= (int)gridDim.x ? HM_REGIONS : 1;
    const int region = (int)(blockIdx.x % nreg), gstride = (int)(gridDim.x / nreg);
    const i64 per = (nchunks + nreg - 1) / nreg;
    const i64 cbeg = region * per, cend = (cbeg + per < nchunks) ? cbeg + per : nchunks;
    i64 ch = cbeg + (blockIdx.x / nreg);
    if ((int)(blockIdx.x / nreg) >= gstride) ch = cend;          // leftover workgroups of an uneven split: nothing to do
    const i64 step = gstride;
    int buf = 0;
    // step A of one staged chunk: partial T over this wave's columns.
    // v_mfma_f64_4x4x4_4b: four independent 4 x 4 x 4 blocks per instruction, A lane = i + 4 blk + 16 k,
    // B lane = j + 4 blk + 16 k, D lane = j + 4 blk + 16 i (tools/mfma_f64_4x4_probe.hip).  Block blk takes the
    // vectors q = 4 blk + j, all four blocks the same four observations: one instruction per row group of four,
    // two per k-step -- a 16 x 16 x 4 tile would spend half of its rows on padding (8-row chunks).
    // The result lands as T[row = 4 rg + l4][q = l15], the layout step B wants.
    auto step_a = [&](const double* Xs, double (&tp)[2]) {
        const double* arow = Xs + (lane & 3) * STRIDE + pc0 + 8 * l4;
        double tpa[2] = {0.0, 0.0}, tpb[2] = {0.0, 0.0};       // [row group]; even / odd k-steps on separate chains
        // fragments of block b + 1 are read while the 16 MFMAs of block b run (explicit register double buffer)
        d2 fr[2][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fr[0][i] = *reinterpret_cast<const d2*>(arow + 2 * i);
            fr[0][4 + i] = *reinterpret_cast<const d2*>(arow + 4 * STRIDE + 2 * i);
        }
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int cur = b & 1;
            if (b + 1 < NBW) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fr[cur ^ 1][i] = *reinterpret_cast<const d2*>(arow + 32 * (b + 1) + 2 * i);
                    fr[cur ^ 1][4 + i] = *reinterpret_cast<const d2*>(arow + 4 * STRIDE + 32 * (b + 1) + 2 * i);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                tpa[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fr[cur][t >> 1][0], uf[b][t], tpa[0], 0, 0, 0);
                tpa[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(fr[cur][4 + (t >> 1)][0], uf[b][t], tpa[1], 0, 0, 0);
                tpb[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fr[cur][t >> 1][1], uf[b][t + 1], tpb[0], 0, 0, 0);
                tpb[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(fr[cur][4 + (t >> 1)][1], uf[b][t + 1], tpb[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        tp[0] = tpa[0] + tpb[0]; tp[1] = tpa[1] + tpb[1];
    };
    auto load_weights = [&](int b, double& c0, double& c1) {
        // weights of the staged chunk's rows in the D-register layout of T: reg s <-> row l4 + 4 s (after the barrier)
        c0 = Cst[b * HM_ROWS + l4]; c1 = Cst[b * HM_ROWS + 4 + l4];
    };
    // the full T of a chunk from the eight partial tiles in LDS slot `sl`, scaled by the chunk's weights
    auto gather_t = [&](const double* Tp, double c0, double c1, double& t0, double& t1) {
        double pa[NW], pb[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) { pa[w] = Tp[(w * 2 + 0) * 64 + lane]; pb[w] = Tp[(w * 2 + 1) * 64 + lane]; }
        __builtin_amdgcn_sched_barrier(0);                    // all reads in flight before the first add
        t0 = pa[0]; t1 = pb[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) { t0 += pa[w]; t1 += pb[w]; }
        t0 *= c0; t1 *= c1;
    };

    if (TONLY) {
        if (ch < cend) issue(ch, 0);
        for (; ch < cend; ch += step) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): this wave's part of the stage has landed
            __syncthreads();                                      // ... and everybody's; the other buffer is free again
            double c0, c1;
            load_weights(buf, c0, c1);
            const i64 nxt = ch + step;
            if (nxt < cend) issue(nxt, buf ^ 1);
            double tp[2];
            step_a(lds + buf * (HM_ROWS * STRIDE), tp);
            Tpart[(wave * 2 + 0) * 64 + lane] = tp[0];             // rows l4       (register 0)
            Tpart[(wave * 2 + 1) * 64 + lane] = tp[1];             // rows l4 + 4   (register 1)
            // LDS traffic only: a __syncthreads() here makes hipcc drain vmcnt(0) as well, i.e. wait for the NEXT
            // chunk's LDS-DMA in the middle of this one
            __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            double t0, t1;
            gather_t(Tpart, c0, c1, t0, t1);
            if (wave == 0 && l15 < Q) {
                const i64 n = ch * HM_ROWS + l4;
                if (n < N) Tout[n * ldt + l15] = t0;
                if (n + 4 < N) Tout[(n + 4) * ldt + l15] = t1;
            }
            buf ^= 1;
        }
        return;
    }

    // ---- both contractions, ONE barrier per chunk ------------------------------------------------------------------
    // Round 2's loop had two: "the stage has landed" and "the partial T tiles have met".  Alone (its loads switched off)
    // that loop took 1.30 ms for the 8.2 GB headline matrix against 0.83 ms of matrix-pipe time, the loads alone 1.13 ms
    // (tools/stream_probe.hip: LDS-DMA of this very staging streams at 7.3 TB/s), together 1.52 ms: the kernel is bound by
    // what sits between its MFMAs, and all eight waves -- both waves of every SIMD -- stop at each barrier together.
    // Now step B runs one chunk BEHIND step A: iteration k does  barrier | issue DMA(k+1) | gather T(k-1), step B(k-1) from
    // REGISTERS | step A(k), write partial T(k) | copy this wave's step-B fragments of chunk k to registers.  The partial
    // tiles are written before the barrier and read after it (two slots, alternating), the stage buffer of chunk k-1 is
    // free at barrier k because its fragments were copied out, and the MFMAs of B(k-1) and A(k) run back to back.
    const double* brow0_0 = lds + l4 * STRIDE + pc0 + 2 * l15;               // step-B fragment rows inside buffer 0
    d2 xb[NT / 2][2];                                                        // fragments of the previous chunk
    double c0p = 0.0, c1p = 0.0;
    bool have_prev = false;
    int slot = 0;
    auto step_b = [&](double t0, double t1) {
        // one 16-byte fragment feeds two tiles: tile m takes the columns pc0 + 32 (m >> 1) + 2 i + (m & 1)
#pragma unroll
        for (int h = 0; h < NT / 2; ++h) {
            acc[2 * h]     = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[h][0][0], t0, acc[2 * h], 0, 0, 0);
            acc[2 * h + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[h][0][1], t0, acc[2 * h + 1], 0, 0, 0);
            acc[2 * h]     = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[h][1][0], t1, acc[2 * h], 0, 0, 0);
            acc[2 * h + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[h][1][1], t1, acc[2 * h + 1], 0, 0, 0);
        }
    };
    if (ch < cend) issue(ch, 0);
    for (; ch < cend; ch += step) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): this wave's part of the stage has landed
        __syncthreads();                                      // everybody's has; T(k-1) is complete; the other buffer is free
        double c0, c1;
        load_weights(buf, c0, c1);
        const i64 nxt = ch + step;
        if (nxt < cend) issue(nxt, buf ^ 1);
        if (have_prev) {
            double t0, t1;
            gather_t(Tpart + (slot ^ 1) * (NW * 2 * 64), c0p, c1p, t0, t1);
            step_b(t0, t1);
        }
        const double* Xs = lds + buf * (HM_ROWS * STRIDE);
        double tp[2];
        step_a(Xs, tp);
        double* Tp = Tpart + slot * (NW * 2 * 64);
        Tp[(wave * 2 + 0) * 64 + lane] = tp[0];                // rows l4       (register 0)
        Tp[(wave * 2 + 1) * 64 + lane] = tp[1];                // rows l4 + 4   (register 1)
        const double* b0 = brow0_0 + buf * (HM_ROWS * STRIDE);
#pragma unroll
        for (int h = 0; h < NT / 2; ++h) {
            xb[h][0] = *reinterpret_cast<const d2*>(b0 + 32 * h);
            xb[h][1] = *reinterpret_cast<const d2*>(b0 + 4 * STRIDE + 32 * h);
        }
        c0p = c0; c1p = c1; have_prev = true;
        slot ^= 1; buf ^= 1;
    }
    if (have_prev) {                                          // the last chunk's step B
        __syncthreads();
        double t0, t1;
        gather_t(Tpart + (slot ^ 1) * (NW * 2 * 64), c0p, c1p, t0, t1);
        step_b(t0, t1);
    }
    if (TONLY) return;
    // partial R of this workgroup: [P][16]
    double* out = Rpart + (i64)blockIdx.x * P * 16;
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = pc0 + 32 * (m >> 1) + 2 * (l4 + 4 * r) + (m & 1);
            out[p * 16 + l15] = acc[m][r];
        }
}

// Out[q][off + p] = sum over workgroups of Rpart[g][p][q], fixed order (deterministic): a block owns 32 consecutive
// elements e = p * 16 + q; its 8 thread rows each sum every 8th partial (coalesced 256-byte reads), then the eight row
// sums are added in a fixed order through LDS -- G = 256 partials of 128 KiB are read at streaming rate
// (the one-thread-per-element form took 64 us per call, 4 % of a blocked-CG iteration)
__global__ __launch_bounds__(256)
void hvp_multi_reduce_kernel(const double* __restrict__ Rpart, int G, int P, int Ppad, int Q, i64 ldo, i64 off,
                             double* __restrict__ Out, const double* __restrict__ live /* nullable: Q flags */)
{
    if (live) {
        bool any = false;
        for (int q = 0; q < Q; ++q) any = any || (live[q] != 0.0);
        if (!any) return;
    }
    __shared__ double sh[8][32];
    const int col = threadIdx.x & 31, row = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + col;                         // e = p * 16 + q
    double s = 0.0;
    if (e < P * 16)
        for (int g = row; g < G; g += 8) s += Rpart[(i64)g * Ppad * 16 + e];
    sh[row][col] = s;
    __syncthreads();
    if (row == 0 && e < P * 16) {
        double t = sh[0][col];
#pragma unroll
        for (int r = 1; r < 8; ++r) t += sh[r][col];
        const int p = e >> 4, q = e & 15;
        if (q < Q) Out[(i64)q * ldo + off + p] = t;
    }
}

bool hvp_multi_supported(const lrvb_ctx* c, i64 Q) {
    return Q >= 1 && Q <= 16 && c->P % 2 == 0 && c->P >= 2 && c->P <= 1024 && c->N >= 1 &&
           ((((uintptr_t)c->X.p) & 15) == 0);
}

// Out (Q x ldo, row q) [off .. off + P) = X^T diag(cw) X U[q, off .. off + P);  U is Q x ldu row-major.
int launch_hvp_multi(lrvb_ctx* c, i64 Q, const double* U_dev, i64 ldu, double* Out_dev, i64 ldo)
{
    if (!hvp_multi_supported(c, Q)) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "fused multi-vector pass: even n_cols <= 1024, 16-byte aligned rows, at most 16 vectors");
    const int Preal = (int)c->P;
    const int P = ((Preal + 127) / 128) * 128;
    const i64 nchunks = (c->N + HM_ROWS - 1) / HM_ROWS;
    int grid = 256;                                           // one workgroup per CU (64+ KiB of LDS each)
    if (grid > nchunks) grid = (int)nchunks;
    LRVB_TRY(buf_reserve(c, c->part_vec, (size_t)grid * (size_t)P * 16));
    const double* Uoff = U_dev + c->glm_off;
    const double* live = c->hm_live;           // set by the blocked CG around its products, null otherwise
    const bool eight = ((P / 128) % 2 == 0) && !c->hm_four_waves;      // two waves per SIMD where the columns split evenly
    const size_t lds_bytes = (size_t)(2 * HM_ROWS * (P + 2) + 2 * (eight ? 8 : 4) * 2 * 64 + 2 * HM_ROWS) * sizeof(double);
#define HM_LAUNCH_W(NB, NW) do { \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&hvp_multi_kernel<NB, false, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        hipLaunchKernelGGL((hvp_multi_kernel<NB, false, NW>), dim3((unsigned)grid), dim3(64 * NW), lds_bytes, c->stream, \
                           c->X.p, Preal, c->N, c->cw.p, Uoff, ldu, (int)Q, c->part_vec.p, (double*)nullptr, (i64)0, live); } while (0)
#define HM_LAUNCH(NB) HM_LAUNCH_W(NB, 4)
#define HM_LAUNCH_E(NB) do { if (eight) HM_LAUNCH_W(NB, 8); else HM_LAUNCH_W(NB, 4); } while (0)
    switch (P / 128) {
    case 1: HM_LAUNCH(1); break; case 2: HM_LAUNCH_E(2); break; case 3: HM_LAUNCH(3); break; case 4: HM_LAUNCH_E(4); break;
    case 5: HM_LAUNCH(5); break; case 6: HM_LAUNCH_E(6); break; case 7: HM_LAUNCH(7); break; default: HM_LAUNCH_E(8); break;
    }
#undef HM_LAUNCH_E
#undef HM_LAUNCH_W
#undef HM_LAUNCH
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(hvp_multi_reduce_kernel, dim3((unsigned)((Preal * 16 + 31) / 32)), dim3(256), 0, c->stream,
                       c->part_vec.p, grid, Preal, P, (int)Q, ldo, c->glm_off, Out_dev, live);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// Tout[n - n0][q] = rowscale[n] * sum_p X[n][p] Zt[q][p]  for rows n0 <= n < n1 and q < Q <= 16 (Zt is Q x ldz row-major).
// rowscale must be followed by at least 8 zeros past n1 - 1 when n1 == N (reserve_obs_vec provides 64).
int launch_rows_times_matrix(lrvb_ctx* c, i64 n0, i64 n1, i64 Q, const double* Zt_dev, i64 ldz,
                             const double* rowscale_dev, double* Tout_dev, i64 ldt)
{
    if (!hvp_multi_supported(c, Q)) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "fused multi-vector pass: even n_cols <= 1024, 16-byte aligned rows, at most 16 vectors");
    const int Preal = (int)c->P;
    const int P = ((Preal + 127) / 128) * 128;
    const i64 rows = n1 - n0;
    if (rows <= 0) return LRVB_OK;
    const i64 nchunks = (rows + HM_ROWS - 1) / HM_ROWS;
    int grid = 256;
    if (grid > nchunks) grid = (int)nchunks;
    const bool eight = ((P / 128) % 2 == 0) && !c->hm_four_waves;
    const size_t lds_bytes = (size_t)(2 * HM_ROWS * (P + 2) + 2 * (eight ? 8 : 4) * 2 * 64 + 2 * HM_ROWS) * sizeof(double);
#define HM_LAUNCH_TW(NB, NW) do { \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&hvp_multi_kernel<NB, true, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        hipLaunchKernelGGL((hvp_multi_kernel<NB, true, NW>), dim3((unsigned)grid), dim3(64 * NW), lds_bytes, c->stream, \
                           c->X.p + n0 * (i64)Preal, Preal, rows, rowscale_dev + n0, Zt_dev, ldz, (int)Q, (double*)nullptr, Tout_dev, ldt, (const double*)nullptr); } while (0)
#define HM_LAUNCH_T(NB) HM_LAUNCH_TW(NB, 4)
#define HM_LAUNCH_TE(NB) do { if (eight) HM_LAUNCH_TW(NB, 8); else HM_LAUNCH_TW(NB, 4); } while (0)
    switch (P / 128) {
    case 1: HM_LAUNCH_T(1); break; case 2: HM_LAUNCH_TE(2); break; case 3: HM_LAUNCH_T(3); break; case 4: HM_LAUNCH_TE(4); break;
    case 5: HM_LAUNCH_T(5); break; case 6: HM_LAUNCH_TE(6); break; case 7: HM_LAUNCH_T(7); break; default: HM_LAUNCH_TE(8); break;
    }
#undef HM_LAUNCH_TE
#undef HM_LAUNCH_TW
#undef HM_LAUNCH_T
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
