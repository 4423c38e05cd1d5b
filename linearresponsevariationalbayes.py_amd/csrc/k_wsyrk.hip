// k_wsyrk.hip -- weighted SYRK  S = X^T diag(c) X  in fp64 on the gfx950 matrix cores.
//
// This is the O(N D^2) part of Objective.fun_free_hessian (reference: autograd.hessian at
// LRVB/SparseObjectives.py:103, 156-158 -- D reverse passes over the N observations) and of
// the Gram matrix G^T G (c = loss'^2).  The reference has no kernel to translate; the layout
// below is designed for CDNA4:
//
//   * output-stationary 128x128 tiles of the LOWER triangle of S, one tile per workgroup
//     (4 wavefronts, each a 64x64 sub-tile = 4x4 v_mfma_f64_16x16x4_f64 accumulators);
//   * the observation axis is split n_splits ways; a workgroup streams its row range through
//     a double-buffered LDS stage of 16 observations x (128 + 128) columns, register-staged
//     (global_load_dwordx4 issued one stage ahead of the MFMAs that consume it);
//   * blockIdx -> (split, tile) is XCD-aware: workgroups that share an XCD (blockIdx % 8)
//     walk the same row range, so every X line is pulled from HBM into one L2 only and the
//     35 other tiles of that split hit in L2;
//   * split partials are reduced by a second, deterministic kernel (no atomics -> results do
//     not depend on dispatch order).
//
// MFMA operand maps (v_mfma_f64_16x16x4_f64, one f64 per lane for A and B):
//   A[i = lane & 15][k = lane >> 4],  B[k = lane >> 4][j = lane & 15],
//   D: 4 f64 per lane, reg r -> row i = (lane >> 4) + 4 r, col j = lane & 15.
// With S_ij = sum_n c_n X_ni X_nj the contraction index k is the observation, so both
// operands are read from the staged rows with the SAME (row = k, col = i or j) pattern.
#include "lrvb_internal.h"
#include <type_traits>
#include <math.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// (stride mod 32 doubles) == 16: the two 16-lane halves of a ds_read_b64 lane group read
// consecutive observations and land on disjoint banks.
constexpr int WS_LDS_STRIDE = WS_TILE + 16;      // (strides 130 .. 152 measured in round 3: the kernel time does not move)

int wsyrk_num_tiles(i64 P) {
    i64 nb = (P + WS_TILE - 1) / WS_TILE;
    return (int)(nb * (nb + 1) / 2);
}

int wsyrk_auto_splits(const lrvb_ctx* c) {
    if (c->n_splits_user > 0) return ((c->n_splits_user + 7) / 8) * 8;
    int T = wsyrk_num_tiles(c->P);
    // aim at ~18 work items per CU (256 CUs): shorter workgroups drift less against the other
    // tiles of their split, which is what keeps shared X lines in the XCD's L2 (measured at
    // N = 1e6, P = 1024: 29.6 GB fetched with 64 splits, 23.6 GB with 128, same kernel time);
    // keep >= 256 observations per split
    i64 s = (4608 + T / 2) / T;
    s = ((s + 7) / 8) * 8;
    // every split writes a full set of partial tiles (128 KiB each): keep >= ~2000 observations per split so that
    // the partial traffic stays small beside the staged rows (a 125k-row shard of the 8-GPU run: 64 splits, 2.40 ms
    // per build against 2.48 ms with 128)
    i64 max_by_rows = c->N / 1900;
    max_by_rows = (max_by_rows / 8) * 8;
    if (s > max_by_rows) s = max_by_rows;
    if (s > 128) s = 128;
    if (s < 8) s = 8;
    return (int)s;
}

// Raw pair load from a clamped (always readable) address.  Out-of-range values are zeroed later,
// in store_stage: anything that CONSUMES a loaded value here would make hipcc wait for the load
// before the MFMA block instead of behind it.
template <bool VEC>
__device__ __forceinline__ void ws_load_pair_raw(const double* __restrict__ rowp, int col, int P,
                                                 double& v0, double& v1) {
    if (VEC) {                      // P even, 16-byte aligned rows: a pair never straddles column P
        const double2 t = *reinterpret_cast<const double2*>(rowp + (col < P ? col : 0));
        v0 = t.x; v1 = t.y;
    } else {
        v0 = rowp[col < P ? col : 0];
        v1 = rowp[col + 1 < P ? col + 1 : 0];
    }
}

template <bool VEC>
__global__ __launch_bounds__(WS_THREADS, 2)
void wsyrk_kernel(const double* __restrict__ X, i64 ldx, i64 N, int P,
                  const double* __restrict__ cvec, int n_splits, int T, i64 rows_per_split,
                  double* __restrict__ partial)
{
    __shared__ double lds[2][2][WS_KC][WS_LDS_STRIDE];   // [stage][A|B][obs][col]

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int xcd = b & 7;                  // label of the workgroups that share an XCD
    const int q = b >> 3;
    const int split_local = q / T;
    const int t = q - split_local * T;
    const int split = split_local * 8 + xcd;

    int bi = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const bool diag = (bi == bj);

    i64 r0 = (i64)split * rows_per_split;
    i64 r1 = r0 + rows_per_split;
    if (r1 > N) r1 = N;
    if (r0 > N) r0 = N;
    const int nch = (int)((r1 - r0 + WS_KC - 1) / WS_KC);

    const int wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const bool skip = diag && (wr == 0) && (wc == 1);   // strictly-upper 64x64 of a diagonal tile

    d4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};

    // staging registers: 4 pairs per panel per thread
    double sa[4][2], sb[4][2], sc[4];
    const int colA0 = bi * WS_TILE, colB0 = bj * WS_TILE;

    auto load_stage = [&](int ch) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * WS_THREADS + tid;
            const int row = idx >> 6;
            const int c2 = (idx & 63) * 2;
            const i64 n = r0 + (i64)ch * WS_KC + row;
            const i64 ne = n < r1 ? n : r1 - 1;            // clamped: always a readable row
            const double* rowp = X + ne * ldx;
            sc[it] = cvec[ne];
            ws_load_pair_raw<VEC>(rowp, colA0 + c2, P, sa[it][0], sa[it][1]);
            ws_load_pair_raw<VEC>(rowp, colB0 + c2, P, sb[it][0], sb[it][1]);   // diagonal tiles: same lines, L1 hits
        }
    };
    auto store_stage = [&](int ch, int buf) {
        // pin the staged values here: every consumer of a loaded value (masking, c_n scaling) sits
        // behind this point, so the one vmcnt wait of the stage lands after the MFMA block
#pragma unroll
        for (int it = 0; it < 4; ++it)
            asm volatile("" : "+v"(sa[it][0]), "+v"(sa[it][1]), "+v"(sb[it][0]), "+v"(sb[it][1]), "+v"(sc[it]));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * WS_THREADS + tid;
            const int row = idx >> 6;
            const int c2 = (idx & 63) * 2;
            const i64 n = r0 + (i64)ch * WS_KC + row;
            const double cv = n < r1 ? sc[it] : 0.0;       // dead rows contribute 0 * finite
            const int ca = colA0 + c2, cb = colB0 + c2;
            const double a0 = ca < P ? sa[it][0] * cv : 0.0, a1 = ca + 1 < P ? sa[it][1] * cv : 0.0;
            const double b0 = cb < P ? sb[it][0] : 0.0, b1 = cb + 1 < P ? sb[it][1] : 0.0;
            *reinterpret_cast<double2*>(&lds[buf][0][row][c2]) = make_double2(a0, a1);
            *reinterpret_cast<double2*>(&lds[buf][1][row][c2]) = make_double2(b0, b1);
        }
    };

    if (nch > 0) {
        load_stage(0);
        store_stage(0, 0);
    }
    __syncthreads();

    int buf = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = (ch + 1 < nch);
        if (more) load_stage(ch + 1);
        if (!skip) {
#pragma unroll
            for (int kk = 0; kk < WS_KC / 4; ++kk) {
                const int krow = kk * 4 + (lane >> 4);
                double af[4], bf[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = lds[buf][0][krow][wr * 64 + m * 16 + (lane & 15)];
#pragma unroll
                for (int n = 0; n < 4; ++n) bf[n] = lds[buf][1][krow][wc * 64 + n * 16 + (lane & 15)];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
            }
        }
        if (more) store_stage(ch + 1, buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    double* out = partial + ((i64)split * T + t) * (i64)(WS_TILE * WS_TILE);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = wr * 64 + m * 16 + (lane >> 4) + 4 * r;
                const int j = wc * 64 + n * 16 + (lane & 15);
                out[i * WS_TILE + j] = acc[m][n][r];
            }
}

// ---- LDS-DMA variant (the fast path: even P, 16-byte aligned rows) ------------------------------
// Same tiling, but the stage is filled by global_load_lds_dwordx4 (one 1 KiB panel row per wave
// instruction, no staging registers, no ds_write, no masking VALU) issued one stage ahead; the
// c_n scaling moves to the A-fragment read (4 v_mul_f64 per 16 MFMAs, issued in their shadow).
// Out-of-range rows are clamped to a readable row and neutralised by c = 0 (the c vector carries
// >= 32 zeros of padding past N); out-of-range columns produce garbage only in tile entries that
// nobody reads.  ALL LDS lives in one array (a second __shared__ object next to an LDS-DMA target
// makes hipcc drain vmcnt(0) before every ds_read).
constexpr int WS_PANEL = WS_KC * WS_LDS_STRIDE;          // doubles per panel
constexpr int WS_BUF = 2 * WS_PANEL + 64;                // A panel, B panel, 32 c values, 32 (c y) values (Gaussian shortcut)

#define WS_GLDS16(gp, lp) __builtin_amdgcn_global_load_lds( \
    (const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)
#define WS_GLDS4(gp, lp) __builtin_amdgcn_global_load_lds( \
    (const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 4, 0, 0)
// scalar-base form: global address = sbase (SGPR pair, wave-uniform) + voff (32-bit per-lane byte offset); the LDS
// destination base travels in M0 (the hardware adds lane * size)
#define WS_GLDS16_S(sbase, voff, ldsaddr) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")
// the same with the streaming cache policy (sc1 nt): the diagonal tiles are the only readers of their panel, so their lines
// need not displace what the off-diagonal readers share in L2 (14.97 against 15.07 ms over six alternating runs)
#define WS_GLDS16_S_NT(sbase, voff, ldsaddr) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 sc1 nt" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")
#define WS_GLDS4_S(sbase, voff, ldsaddr) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")

__global__ __launch_bounds__(WS_THREADS, 2)
void wsyrk_glds_kernel(const double* __restrict__ X, i64 ldx, i64 N, int P,
                       const double* __restrict__ cpad, int n_splits, int nb, i64 rows_per_split,
                       double* __restrict__ partial,
                       const double* __restrict__ cypad /* nullable */, double* __restrict__ rpart /* n_splits x nb*128 */)
{
    // cypad != NULL: the diagonal-tile workgroups also accumulate r = X^T (c o y) for their 128 columns over their
    // rows (the panel and c y sit in LDS anyway: 8 FMAs + 12 LDS reads per thread and stage).  With it a Gaussian
    // loss -- whose curvature w tau does not depend on theta -- needs NO separate pass over X for a Hessian build:
    // d f / d beta = S beta - r  (lrvb_api.hip: hessian_partial).
    __shared__ double lds[2 * WS_BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: wave-uniform branches below
    // Work order inside one XCD's queue (blockIdx % 8 = XCD label, blockIdx / 8 = position):
    // first every off-diagonal tile of every split this XCD owns (16 MFMA tiles per wave and
    // k-step), then the diagonal tiles (9 per wave and k-step, i.e. 9/16 of the duration):
    // longest-first keeps the 2-blocks-per-CU schedule free of a ragged tail.
    const int xcd = blockIdx.x & 7;
    const int q = blockIdx.x >> 3;
    const int S8 = n_splits >> 3;
    const int n_off = nb * (nb - 1) / 2;
    int bi, bj, split_local;
    if (q < n_off * S8) {
        split_local = q / n_off;
        const int u = q - split_local * n_off;
        bi = (int)((1.f + sqrtf(1.f + 8.f * (float)u)) * 0.5f);
        while (bi * (bi - 1) / 2 > u) --bi;
        while ((bi + 1) * bi / 2 <= u) ++bi;
        bj = u - bi * (bi - 1) / 2;
    } else {
        const int qd = q - n_off * S8;
        split_local = qd / nb;
        bi = bj = qd - split_local * nb;
    }
    const bool diag = (bi == bj);
    const int t = bi * (bi + 1) / 2 + bj;
    const int T = nb * (nb + 1) / 2;
    const int split = split_local * 8 + xcd;

    i64 r0 = (i64)split * rows_per_split;
    i64 r1 = r0 + rows_per_split;
    if (r1 > N) r1 = N;
    if (r0 > N) r0 = N;
    const int nch = (int)((r1 - r0 + WS_KC - 1) / WS_KC);

    // accumulators: full tile = 4 x 4 MFMA tiles (64 x 64 per wave); diagonal tile = the 16-row
    // blocks {wave, 7 - wave} of the lower triangle: (wave + 1) + (8 - wave) = 9 MFMA tiles per wave
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};

    // this lane's column inside each panel (clamped so the 16-byte load stays inside the row)
    int ca = bi * WS_TILE + 2 * lane; if (ca > P - 2) ca = P - 2;
    int cb = bj * WS_TILE + 2 * lane; if (cb > P - 2) cb = P - 2;

    // LDS-DMA of one stage.  Every address is (wave-uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset that
    // never changes): the instruction takes both directly, so a stage costs NO vector-ALU work -- beside a stream of
    // fp64 MFMAs every VALU instruction costs the SIMD ~8 cycles of matrix time (tools/mfma_vmem_probe.hip), and the
    // 64-bit per-lane address arithmetic of the pointer form was 13 of them per stage.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
    unsigned voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        voffA[i] = (unsigned)(((i64)(wave + 4 * i) * ldx + ca) * 8);
        voffB[i] = (unsigned)(((i64)(wave + 4 * i) * ldx + cb) * 8);
    }
    const unsigned voffC = (unsigned)lane * 4u;
    i64 full = (N - r0) / WS_KC;                                  // stages of this split with all 16 rows inside X
    const int nch_full = (int)(full < nch ? full : nch);

    auto issue_stage = [&](int ch, int buf) {
        const unsigned base = lds0 + (unsigned)(buf * WS_BUF) * 8u;
        const i64 n0 = r0 + (i64)ch * WS_KC;
        if (ch < nch_full) {                                     // wave-uniform: all 16 rows exist
            const char* sb = reinterpret_cast<const char*>(X) + n0 * ldx * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned la = base + (unsigned)((wave + 4 * i) * WS_LDS_STRIDE) * 8u;
                if (diag) WS_GLDS16_S_NT(sb, voffA[i], la); else WS_GLDS16_S(sb, voffA[i], la);
                if (!diag) WS_GLDS16_S(sb, voffB[i], la + (unsigned)WS_PANEL * 8u);
            }
        } else {                                                 // last rows of the matrix: clamp to a readable row
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave + 4 * i;
                i64 n = n0 + row; if (n > N - 1) n = N - 1;
                const char* sb = reinterpret_cast<const char*>(X) + (n - row) * ldx * 8;
                const unsigned la = base + (unsigned)(row * WS_LDS_STRIDE) * 8u;
                WS_GLDS16_S(sb, voffA[i], la);
                if (!diag) WS_GLDS16_S(sb, voffB[i], la + (unsigned)WS_PANEL * 8u);
            }
        }
        if (wave == 0)       // 64 dwords = c[n0 .. n0+31]; reads past N hit the zero padding
            WS_GLDS4_S(reinterpret_cast<const char*>(cpad + n0), voffC, base + (unsigned)(2 * WS_PANEL) * 8u);
        if (wave == 1 && diag && cypad)
            WS_GLDS4_S(reinterpret_cast<const char*>(cypad + n0), voffC, base + (unsigned)(2 * WS_PANEL + 32) * 8u);
    };

    if (nch > 0) issue_stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): LDS-DMA of the stage has landed
    __syncthreads();

    const int l15 = lane & 15, l4 = lane >> 4;
    if (!diag) {
        // Wave w owns rows [32 w, 32 w + 32) of the tile and all 128 columns: 2 A fragments x 8 B fragments = 16 MFMAs
        // per k-step, and only TWO c_n multiplies (the scaling sits on the A side).  With 16-byte LDS reads MFMA tile
        // (m, n = 2 h + p) holds rows {32 w + 2 i + m}, columns {32 h + 2 j + p}; the store below undoes it.
        // Every LDS address of the loop is a loop-invariant register + an immediate: the stage loop is unrolled over
        // the two buffers so that no address is recomputed (each VALU instruction costs ~8 cycles of matrix time).
        const double* a_base = lds + l4 * WS_LDS_STRIDE + 32 * wave + 2 * l15;
        const double* b_base = lds + WS_PANEL + l4 * WS_LDS_STRIDE + 2 * l15;
        const double* c_base = lds + 2 * WS_PANEL + l4;
        auto stage = [&](auto buf_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
            double af[2][2], bf[2][8], cv[2];
            auto read_frags = [&](int kk, int set) {
                constexpr int dummy = 0; (void)dummy;
                const int o = BUF * WS_BUF + kk * 4 * WS_LDS_STRIDE;
                cv[set] = c_base[BUF * WS_BUF + kk * 4];
                const d2 va = *reinterpret_cast<const d2*>(a_base + o);
                af[set][0] = va[0]; af[set][1] = va[1];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const d2 vb = *reinterpret_cast<const d2*>(b_base + o + 32 * h);
                    bf[set][2 * h] = vb[0]; bf[set][2 * h + 1] = vb[1];
                }
            };
            read_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < WS_KC / 4; ++kk) {
                const int set = kk & 1;
                // order: (wait for set kk) -> scale -> issue reads of set kk+1 -> 16 MFMAs: the only lgkmcnt wait of
                // the k-step sits a full MFMA block after the reads it covers
                const double as0 = af[set][0] * cv[set], as1 = af[set][1] * cv[set];
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < WS_KC / 4) read_frags(kk + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(as0, bf[set][n], acc[n], 0, 0, 0);
                    acc[8 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(as1, bf[set][n], acc[8 + n], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads();      // explicit vmcnt(0): the DMA of the next stage must have landed in every wave
        };
        for (int ch = 0; ch < nch; ch += 2) {
            if (ch + 1 < nch) issue_stage(ch + 1, 1);
            stage(std::integral_constant<int, 0>{});
            if (ch + 1 < nch) {
                if (ch + 2 < nch) issue_stage(ch + 2, 0);
                stage(std::integral_constant<int, 1>{});
            }
        }
        double* out = partial + ((i64)split * T + t) * (i64)(WS_TILE * WS_TILE);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(32 * wave + 2 * (l4 + 4 * r) + m) * WS_TILE + 32 * (n >> 1) + 2 * l15 + (n & 1)] = acc[m * 8 + n][r];
    } else {
        const int rb0 = wave, rb1 = 7 - wave;        // this wave's two 16-row blocks
        d2 racc = (d2){0.0, 0.0};
        const double* row_base = lds + l4 * WS_LDS_STRIDE + l15;       // loop-invariant LDS addresses, as above
        const double* c_base = lds + 2 * WS_PANEL + l4;
        auto stage = [&](auto buf_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
            double a0[2], a1[2], bf[2][8], cv[2];
            auto read_frags = [&](int kk, int set) {
                const double* rowp = row_base + BUF * WS_BUF + kk * 4 * WS_LDS_STRIDE;
                cv[set] = c_base[BUF * WS_BUF + kk * 4];
                a0[set] = rowp[rb0 * 16];
                a1[set] = rowp[rb1 * 16];
#pragma unroll
                for (int n = 0; n < 8; ++n) bf[set][n] = rowp[n * 16];
            };
            read_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < WS_KC / 4; ++kk) {
                const int set = kk & 1;
                const double s0 = a0[set] * cv[set], s1 = a1[set] * cv[set];
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < WS_KC / 4) read_frags(kk + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (n <= rb0) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(s0, bf[set][n], acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    if (n <= rb1) acc[4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(s1, bf[set][n], acc[4 + n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (cypad) {                 // r[2 cp .. 2 cp + 1] += sum over this wave's 4 rows of (c y)_row x[row][.]: 6 LDS reads, 8 FMAs
                const double* xs = lds + BUF * WS_BUF + wave * 4 * WS_LDS_STRIDE + 2 * lane;
                const d2* cys = reinterpret_cast<const d2*>(lds + BUF * WS_BUF + 2 * WS_PANEL + 32 + wave * 4);
                const d2 cy01 = cys[0], cy23 = cys[1];
                const d2 x0 = *reinterpret_cast<const d2*>(xs), x1 = *reinterpret_cast<const d2*>(xs + WS_LDS_STRIDE);
                const d2 x2 = *reinterpret_cast<const d2*>(xs + 2 * WS_LDS_STRIDE), x3 = *reinterpret_cast<const d2*>(xs + 3 * WS_LDS_STRIDE);
                racc[0] += cy01[0] * x0[0] + cy01[1] * x1[0] + cy23[0] * x2[0] + cy23[1] * x3[0];
                racc[1] += cy01[0] * x0[1] + cy01[1] * x1[1] + cy23[0] * x2[1] + cy23[1] * x3[1];
            }
            __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads();
        };
        for (int ch = 0; ch < nch; ch += 2) {
            if (ch + 1 < nch) issue_stage(ch + 1, 1);
            stage(std::integral_constant<int, 0>{});
            if (ch + 1 < nch) {
                if (ch + 2 < nch) issue_stage(ch + 2, 0);
                stage(std::integral_constant<int, 1>{});
            }
        }
        if (cypad) {                                           // the two row halves meet in LDS (the stage buffers are idle now)
            lds[wave * WS_TILE + 2 * lane] = racc[0];
            lds[wave * WS_TILE + 2 * lane + 1] = racc[1];
            __syncthreads();
            if (tid < 128)
                rpart[((i64)split * nb + bi) * WS_TILE + tid] = (lds[tid] + lds[WS_TILE + tid]) + (lds[2 * WS_TILE + tid] + lds[3 * WS_TILE + tid]);
        }
        // lower-triangle blocks get the sums, the rest of the two block rows is zeroed (never read,
        // but kept finite for the split reduction / all-reduce)
        double* out = partial + ((i64)split * T + t) * (i64)(WS_TILE * WS_TILE);
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v0 = (n < 4 && n <= rb0) ? acc[n < 4 ? n : 0][r] : 0.0;
                const double v1 = (n <= rb1) ? acc[4 + n][r] : 0.0;
                out[(rb0 * 16 + l4 + 4 * r) * WS_TILE + n * 16 + l15] = v0;
                out[(rb1 * 16 + l4 + 4 * r) * WS_TILE + n * 16 + l15] = v1;
            }
    }
}

// ---- narrow matrices (P <= 64): the sufficient-statistics shape of configs 2, 4, 5 --------------
// S = Z^T diag(c) Z with a handful of columns is a streaming problem (HBM-bound up to P ~ 48, balanced
// at P = 64): no tiling over columns, no LDS staging.  Every wavefront streams its own rows straight
// into MFMA operand registers -- a k-step is 4 consecutive rows, lane (i = lane & 15, k = lane >> 4)
// loads the column PAIR (32 m + 2 i, 32 m + 2 i + 1) of row k with one 16-byte load, so one wave
// instruction reads 4 x 256 contiguous bytes.  The same registers serve as the A operand (after the
// c_n scaling) and as the B operand.  MFMA tile t = 2 m + p therefore holds columns 32 m + 2 i + p:
// a fixed column permutation that is undone when the block partial is written.
template <int NPAIR, bool ALIGNED16>      // NPAIR = ceil(P / 32): 1 -> 2 tiles (3 MFMAs per k-step), 2 -> 4 tiles (10)
__global__ __launch_bounds__(256)
void gram_small_kernel(const double* __restrict__ Z, i64 ldz, i64 N, int P,
                       const double* __restrict__ cpad, double* __restrict__ partial /* [grid][64*64] */, int ones_col)
{
    constexpr int NT = 2 * NPAIR;
    constexpr int NACC = NT * (NT + 1) / 2;
    constexpr int KS = 4;                       // k-steps (of 4 rows) per stage
    __shared__ double red[64 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;

    d4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = (d4){0.0, 0.0, 0.0, 0.0};

    int colp[NPAIR];                            // first column of this lane's pair, clamped for the load
    bool in0[NPAIR], in1[NPAIR];
    // ones_col = P (a spare column of the 32 NPAIR the tiles hold; -1 = none): a virtual column of ones, so that entry (P, P) of
    // the block partial is sum_n c_n and row P the weighted column sums -- no separate pass for the sum of the weights
    double fill0[NPAIR], fill1[NPAIR];          // what a lane past the last column contributes: 0, or 1 in the column of ones (the selects of the loop below stay the same)
#pragma unroll
    for (int m = 0; m < NPAIR; ++m) {
        const int c0 = 32 * m + 2 * li;
        in0[m] = c0 < P; in1[m] = c0 + 1 < P;
        fill0[m] = c0 == ones_col ? 1.0 : 0.0; fill1[m] = c0 + 1 == ones_col ? 1.0 : 0.0;
        colp[m] = in1[m] ? c0 : (P >= 2 ? P - 2 : 0);
    }

    // Every stage's loads are issued unconditionally (a stage past the end re-reads the last one with zero weights): with
    // the prefetch behind `if (next < nstages)` hipcc's wait-count bookkeeping assumed the path that issued nothing and
    // every consume waited for ALL outstanding loads -- the next stage never overlapped the MFMAs of the current one
    // (round 3, the same finding as in k_lmm.hip).  Here it bought little (0.175 -> 0.167 ms for 1e6 x 64 = 3.1 TB/s): at 260
    // registers the kernel runs ONE wave per SIMD with one 16-row stage in flight, ~1 us of MFMA work against a longer load
    // latency; the LDS-DMA ring of k_lmm.hip (three waves per SIMD, six slots) is the design that streams at 4.5 TB/s.
    auto load_stage = [&](double (&x)[KS][NPAIR][2], double (&cv)[KS], i64 row0, double live) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            i64 n = row0 + ks * 4 + lk;
            const double lv = n < N ? live : 0.0;               // rows past N weigh nothing (the vector needs no padding here)
            if (n > N - 1) n = N - 1;
            cv[ks] = cpad[n] * lv;
            const double* rowp = Z + n * ldz;
#pragma unroll
            for (int m = 0; m < NPAIR; ++m) {
                if (ALIGNED16) {
                    typedef double v2d __attribute__((ext_vector_type(2)));       // streamed once: non-temporal
                    const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(rowp + colp[m]));
                    x[ks][m][0] = t[0]; x[ks][m][1] = t[1];
                } else {
                    x[ks][m][0] = rowp[colp[m]]; x[ks][m][1] = rowp[colp[m] + (P >= 2 ? 1 : 0)];
                }
            }
        }
    };
    auto consume = [&](double (&x)[KS][NPAIR][2], double (&cv)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            double b[NT], a[NT];
#pragma unroll
            for (int m = 0; m < NPAIR; ++m) {
                // a clamped pair (P odd or lane past the last column) is re-mapped to zeros
                // (a select of two registers: written as x[ks][m][P >= 2 ? 1 : 0] it became a dynamic register index, i.e. a
                // readfirstlane loop in every k-step)
                const double xl = (P >= 2) ? x[ks][m][1] : x[ks][m][0];
                const double v0 = in1[m] ? x[ks][m][0] : (in0[m] ? xl : fill0[m]);
                const double v1 = in1[m] ? x[ks][m][1] : fill1[m];
                b[2 * m] = v0; b[2 * m + 1] = v1;
                a[2 * m] = v0 * cv[ks]; a[2 * m + 1] = v1 * cv[ks];
            }
            int idx = 0;
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = 0; tb <= ta; ++tb) {
                    acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[idx], 0, 0, 0);
                    ++idx;
                }
        }
    };

    // stages of 16 rows, dealt round-robin to the waves of the whole grid; double-buffered in registers
    const i64 stage_rows = KS * 4;
    const i64 nstages = (N + stage_rows - 1) / stage_rows;
    const i64 gw = (i64)blockIdx.x * 4 + wave, GW = (i64)gridDim.x * 4;
    if (gw < nstages) {
        double xa[KS][NPAIR][2], xb[KS][NPAIR][2], ca[KS], cb[KS];
        const i64 mine = (nstages - gw + GW - 1) / GW;          // stages of this wave
        auto row0_of = [&](i64 k) { const i64 st = gw + k * GW; return (st < nstages ? st : nstages - 1) * stage_rows; };
        load_stage(xa, ca, row0_of(0), 1.0);
        for (i64 k = 0; k < mine; k += 2) {
            load_stage(xb, cb, row0_of(k + 1), (k + 1 < mine) ? 1.0 : 0.0);
            consume(xa, ca);
            load_stage(xa, ca, row0_of(k + 2), (k + 2 < mine) ? 1.0 : 0.0);
            consume(xb, cb);                                    // a dead stage contributes zeros
        }
    }

    // wave partial -> 64 x 64 block in TRUE column order (tile t, lane index i <-> column 32 (t>>1) + 2 i + (t&1)),
    // waves combined through LDS in a fixed order, both triangles filled
    auto scatter = [&](double* dst, bool add) {
        int idx = 0;
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int tb = 0; tb <= ta; ++tb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ia = lk + 4 * r;                                 // row index inside tile ta
                    const int row = 32 * (ta >> 1) + 2 * ia + (ta & 1);
                    const int col = 32 * (tb >> 1) + 2 * li + (tb & 1);
                    const double v = acc[idx][r];
                    if (add) { dst[row * 64 + col] += v; if (ta != tb) dst[col * 64 + row] += v; }
                    else     { dst[row * 64 + col] = v;  if (ta != tb) dst[col * 64 + row] = v; }
                }
                ++idx;
            }
    };
    // the four wave partials are summed in LDS in wave order (deterministic), then the used
    // 32 NPAIR x 32 NPAIR corner goes to this block's slot
    if (wave == 0) scatter(red, false);
    __syncthreads();
    if (wave == 1) scatter(red, true);
    __syncthreads();
    if (wave == 2) scatter(red, true);
    __syncthreads();
    if (wave == 3) scatter(red, true);
    __syncthreads();
    double* out = partial + (i64)blockIdx.x * (64 * 64);
    for (int e = tid; e < 64 * 64; e += 256) {
        const int row = e >> 6, col = e & 63;
        if (row < 32 * NPAIR && col < 32 * NPAIR) out[e] = red[e];
    }
}

// tiles[0] (128 x 128) <- sum_b partial[b] (64 x 64), fixed order, ONE launch over the used C x C corner only (C = 32 or 64):
// a workgroup owns 16 consecutive corner entries, slice s of its 16 sums the blocks s, s + 16, ... on four independent chains,
// the slices meet in LDS and are added in order.  (Round 3: two launches over all 4096 entries of up to 1024 blocks, 11 + 6 us
// around a 14 us kernel at N = 1e5 x 22.)
__global__ __launch_bounds__(256)
void gram_small_reduce_kernel(const double* __restrict__ partial, int nblk, int C, int P, double* __restrict__ tile0,
                              double* __restrict__ dense_out /* nullable: the leading pd x pd entries, leading dimension ldd */, i64 ldd, int pd,
                              double* __restrict__ csum_out /* nullable: entry (P, P) = sum of the weights (ones column) */)
{
    __shared__ double sh[16][17];
    {   // the rest of the 128 x 128 tile is zeroed here (it is never read, but it travels in the sum over ranks and must stay
        // finite): every workgroup clears its share, skipping the P x P entries that are written below -- no separate memset
        const int per = (WS_TILE * WS_TILE) / (int)gridDim.x;
        for (int i = (int)blockIdx.x * per + (int)threadIdx.x; i < ((int)blockIdx.x + 1) * per; i += 256) {
            const int r = i / WS_TILE, cc = i - r * WS_TILE;
            if (r >= P || cc >= P) tile0[i] = 0.0;
        }
    }
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int id = blockIdx.x * 16 + el;                       // entry of the C x C corner (C * C is a multiple of 16)
    const int row = id / C, col = id - row * C;
    const double* src = partial + row * 64 + col;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = sl;
    for (; b + 48 < nblk; b += 64) {
        s0 += src[(i64)b * 4096]; s1 += src[(i64)(b + 16) * 4096];
        s2 += src[(i64)(b + 32) * 4096]; s3 += src[(i64)(b + 48) * 4096];
    }
    for (; b < nblk; b += 16) s0 += src[(i64)b * 4096];
    sh[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    const bool inner = row < P && col < P, corner = csum_out && row == P && col == P;
    if (sl == 0 && (inner || corner)) {
        double a = sh[0][el];
#pragma unroll
        for (int q = 1; q < 16; ++q) a += sh[q][el];
        if (inner) { tile0[row * WS_TILE + col] = a; if (dense_out && row < pd && col < pd) dense_out[row * ldd + col] = a; }
        else *csum_out = a;
    }
}

// dense_out (nullable): the leading pd x pd entries (pd = 0: P) of the result also as a dense matrix of leading dimension
// ldd (saves the unpacking launch); csum_out (nullable; needs a spare column, P not 32 or 64): sum_n c_n through a virtual
// column of ones.  Rows past N are never read (clamped, weight zero): Z and cvec_dev need no padding.
int launch_gram_small_on(lrvb_ctx* c, const double* Z, i64 N, i64 P, const double* cvec_dev, double* tiles_out_dev,
                         double* dense_out, i64 ldd, double* csum_out, i64 pd) {
    if (P > 64) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "narrow Gram kernel supports at most 64 columns");
    if (csum_out && (P == 32 || P == 64)) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "no spare column for the sum of the weights");
    const int ones_col = csum_out ? (int)P : -1;
    // stages of 16 rows.  P <= 32 (four workgroups per CU resident): at least four stages per wave where the matrix is small --
    // fewer block partials to add afterwards (N = 1e5 x 22: 391 workgroups instead of 1024).  Wider (one wave per SIMD, one stage
    // in flight: latency-bound): two stages per wave (2e4 x 48, tools/lab/time_gram_small.py: 157 workgroups 20.7 us, 313 -- one
    // stage per wave -- 35.6 us, 79: 23.7 us), up to 1024 workgroups (1e6 x 64 inside a configuration-3 step: 157 us; with the
    // 256 resident ones only: 236 us).
    const i64 stages = (N + 15) / 16;
    i64 grid = P <= 32 ? (stages + 15) / 16 : (stages + 7) / 8;
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    LRVB_TRY(buf_reserve(c, c->tile_part, (size_t)grid * 4096));
    const bool aligned16 = ((P % 2) == 0) && ((((uintptr_t)Z) & 15) == 0);
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
#define GS_LAUNCH(NP, AL) hipLaunchKernelGGL((gram_small_kernel<NP, AL>), dim3((unsigned)grid), dim3(256), 0, c->stream, \
                                             Z, P, N, (int)P, cvec_dev, c->tile_part.p, ones_col)
    if (P <= 32) { if (aligned16) GS_LAUNCH(1, true); else GS_LAUNCH(1, false); }
    else         { if (aligned16) GS_LAUNCH(2, true); else GS_LAUNCH(2, false); }
#undef GS_LAUNCH
    HIP_TRY(hipGetLastError());
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    const int C = P <= 32 ? 32 : 64;                          // C * C / 16 = 64 or 256 workgroups: both divide the 16384 tile entries
    hipLaunchKernelGGL(gram_small_reduce_kernel, dim3((unsigned)(C * C / 16)), dim3(256), 0, c->stream,
                       (const double*)c->tile_part.p, (int)grid, C, (int)P, tiles_out_dev, dense_out, ldd, (int)(pd > 0 ? pd : P), csum_out);
    HIP_TRY(hipGetLastError());
    if (c->prof_on) {
        c->prof.wsyrk_flops = (double)N * (double)P * (double)(P + 1);
        c->prof.wsyrk_bytes = 8.0 * ((double)N * (double)(P + 1)) + 8.0 * 0.5 * (double)P * (double)(P + 1);
    }
    return LRVB_OK;
}

static int launch_gram_small(lrvb_ctx* c, const double* cvec_dev, double* tiles_out_dev) {
    return launch_gram_small_on(c, c->X.p, c->N, c->P, cvec_dev, tiles_out_dev);
}

__global__ void wsyrk_reduce_kernel(const double* __restrict__ partial, int n_splits, i64 tile_elems_total,
                                    double* __restrict__ tiles);
// ---- two-operand variant: C = A^T diag(c) B over the observation axis -----------------------------
// A is N x PA, B is N x PB (both row-major, even widths, 16-byte aligned): every 128 x 128 tile of the
// PA x PB result is an "off-diagonal" tile of the scheme above.  Used for the Schur complement of the
// mixture model (config 3): sum_n (x~_n (x) x~_n) vec(A_n)^T.  Output: row-major tiles [ta][tb].
// SLIVER = 1 (PA = PB = 4 x 128 + 16: the 528 x 528 Schur operand of the K = 32, V = 31 mixture): only the 16 interior
// tiles are workgroups.  The 16-column slivers of A and B are never staged: every workgroup keeps ONE k-step of each in
// registers (A sliver: the k-step bi of a stage, B sliver: k-step bj; plain global loads one stage ahead) and adds
//   its rows of A_panel(bi)^T c B_sliver   over the k-steps bj   (2 MFMAs per wave and stage, A operand re-read from LDS),
//   its columns of A_sliver^T c B_panel(bj) over the k-steps bi  (2 MFMAs),   and, if bi == bj, the corner (1 MFMA, wave 0):
// the four workgroups of a tile row / column each cover a quarter of the k axis, their partial sums go to separate
// 16-wide groups of the edge tile's slot and are added when the tiles are unpacked.  With the edge tiles as workgroups
// of their own (SLIVER = 0) nine of 25 workgroups per split paid a full stage of loads and barriers for 1/8 of the MFMAs:
// 10.65 ms for 5.6e11 flops.  The caller guarantees 16 readable, finite rows past N in A and B.
// SLIVER = 2: the same with the A operand GENERATED ON CHIP.  A is then the N x 31 data matrix X of the mixture (lda = 31) and
// the virtual operand row is the packed lower triangle of x~ x~^T, x~ = [1, x] (528 products of 32 numbers): a stage DMAs
// the 16 rows of X (4 KB instead of the 16 KB panel), and a fragment element is the product of two LDS reads whose
// addresses (the pair (a, b) of its packed column) are computed once per lane.  kron_rows_kernel and its 4.2 GB operand
// (0.9 ms to write, as much to read back) disappear.
template <int SLIVER>
__global__ __launch_bounds__(WS_THREADS, 2)
void atb_glds_kernel(const double* __restrict__ A, i64 lda, int PA, const double* __restrict__ B, i64 ldb, int PB,
                     i64 N, const double* __restrict__ cpad, int n_splits, int nba, int nbb, i64 rows_per_split,
                     double* __restrict__ partial)
{
    __shared__ double lds[2 * WS_BUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7;
    const int qpos = blockIdx.x >> 3;
    const int T = nba * nbb;                      // tile slots of the partial buffer (5 x 5 with SLIVER)
    const int TW = SLIVER ? 16 : T;               // workgroups per split
    const int split_local = qpos / TW;
    const int tw = qpos - split_local * TW;
    const int bi = SLIVER ? (tw >> 2) : tw / nbb, bj = SLIVER ? (tw & 3) : tw - bi * nbb;
    const int t = bi * nbb + bj;
    const int split = split_local * 8 + xcd;
    i64 r0 = (i64)split * rows_per_split;
    i64 r1 = r0 + rows_per_split;
    if (r1 > N) r1 = N;
    if (r0 > N) r0 = N;
    const int nch = (int)((r1 - r0 + WS_KC - 1) / WS_KC);

    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    int ca = bi * WS_TILE + 2 * lane; if (ca > PA - 2) ca = PA - 2;
    int cb = bj * WS_TILE + 2 * lane; if (cb > PB - 2) cb = PB - 2;

    // scalar row bases + constant 32-bit lane offsets, as in wsyrk_glds_kernel: a stage issues no VALU instruction
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
    const unsigned voffA = (unsigned)ca * 8u, voffB = (unsigned)cb * 8u, voffC = (unsigned)lane * 4u;
    const unsigned voffX = (unsigned)(lane < 62 ? lane : 61) * 4u;
    auto issue_stage = [&](int ch, int buf) {
        const unsigned base = lds0 + (unsigned)(buf * WS_BUF) * 8u;
        const i64 n0 = r0 + (i64)ch * WS_KC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave + 4 * i;
            i64 n = n0 + row; if (n > N - 1) n = N - 1;                 // wave-uniform clamp (scalar)
            const unsigned la = base + (unsigned)(row * WS_LDS_STRIDE) * 8u;
            if (SLIVER == 2)      // row n of X: 31 doubles = 62 dwords behind the constant slot 0 (lanes 62, 63 repeat lane 61's dword into the pad)
                WS_GLDS4_S(reinterpret_cast<const char*>(A + n * lda), voffX, la + 8u);
            else
                WS_GLDS16_S(reinterpret_cast<const char*>(A + n * lda), voffA, la);
            WS_GLDS16_S(reinterpret_cast<const char*>(B + n * ldb), voffB, la + (unsigned)WS_PANEL * 8u);
        }
        if (wave == 0)
            WS_GLDS4_S(reinterpret_cast<const char*>(cpad + n0), voffC, base + (unsigned)(2 * WS_PANEL) * 8u);
    };
    if (SLIVER == 2) {            // x~_0 = 1 in slot 0 of the 16 rows of both buffers; the DMA never writes there
        if (tid < 32) lds[(tid >> 4) * WS_BUF + (tid & 15) * WS_LDS_STRIDE] = 1.0;
        __syncthreads();
    }
    if (nch > 0) issue_stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): LDS-DMA of the stage has landed
    __syncthreads();

    const int l15 = lane & 15, l4 = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    // 16-column MFMA tiles of this wave's 64 x 64 block that hold at least one real column (wave-uniform)
    int mt_a = (PA - bi * WS_TILE - wr * 64 + 15) / 16; mt_a = mt_a < 0 ? 0 : (mt_a > 4 ? 4 : mt_a);
    int mt_b = (PB - bj * WS_TILE - wc * 64 + 15) / 16; mt_b = mt_b < 0 ? 0 : (mt_b > 4 ? 4 : mt_b);
    mt_a = __builtin_amdgcn_readfirstlane(mt_a); mt_b = __builtin_amdgcn_readfirstlane(mt_b);
    // interior = the WHOLE 128 x 128 tile holds real columns (the same answer in every wave: the two loops below tile the
    // workgroup differently and both contain barriers)
    const bool full = (PA - bi * WS_TILE >= WS_TILE) && (PB - bj * WS_TILE >= WS_TILE);
    double* out = partial + ((i64)split * T + t) * (i64)(WS_TILE * WS_TILE);
    if (full) {
        // interior tile: the loop of wsyrk_glds_kernel's off-diagonal tiles -- wave w owns rows [32 w, 32 w + 32) x all 128
        // columns (2 scaling multiplies per k-step), 16-byte LDS reads, stage loop unrolled over the two buffers with
        // loop-invariant addresses
        const double* a_base = lds + l4 * WS_LDS_STRIDE + 32 * wave + 2 * l15;
        const double* b_base = lds + WS_PANEL + l4 * WS_LDS_STRIDE + 2 * l15;
        const double* c_base = lds + 2 * WS_PANEL + l4;
        // sliver mode: the k-step of the stage is part of the ADDRESS (LDS pointers below, row offsets of the global loads)
        const double* a_base_s = a_base + bj * 4 * WS_LDS_STRIDE;                   // A panel rows of k-step bj (pairs with the B sliver)
        const double* b_base_s = b_base + bi * 4 * WS_LDS_STRIDE + 32 * wave;       // B panel blocks (2 w, 2 w + 1) of k-step bi (A sliver)
        const double* c_base_i = c_base + bj * 4;
        const double* c_base_ii = c_base + bi * 4;
        const unsigned voffSA = (unsigned)(((i64)(4 * bi + l4) * lda + 4 * WS_TILE + l15) * 8);
        const unsigned voffSB = (unsigned)(((i64)(4 * bj + l4) * ldb + 4 * WS_TILE + l15) * 8);
        double asl[2] = {0.0, 0.0}, bsl[2] = {0.0, 0.0};
        d4 acc_i0 = (d4){0.0, 0.0, 0.0, 0.0}, acc_i1 = acc_i0, acc_ii0 = acc_i0, acc_ii1 = acc_i0, acc_c = acc_i0;
        // SLIVER = 2: packed column v = a (a + 1) / 2 + b (b <= a) of the virtual operand is x~_a x~_b; this lane's two columns
        // of the panel and its column of the sliver, as LDS addresses (row l4 of a k-step; slot s of a row holds x~_s)
        auto pair_of = [](int v, int& a, int& b) {
            a = (int)((sqrtf(8.f * (float)v + 1.f) - 1.f) * 0.5f);
            while (a * (a + 1) / 2 > v) --a;
            while ((a + 1) * (a + 2) / 2 <= v) ++a;
            b = v - a * (a + 1) / 2;
        };
        const double *ga0 = lds, *gb0 = lds, *ga1 = lds, *gb1 = lds, *gas = lds, *gbs = lds;
        if (SLIVER == 2) {
            int a, b;
            pair_of(bi * WS_TILE + 32 * wave + 2 * l15, a, b);     ga0 = lds + l4 * WS_LDS_STRIDE + a; gb0 = lds + l4 * WS_LDS_STRIDE + b;
            pair_of(bi * WS_TILE + 32 * wave + 2 * l15 + 1, a, b); ga1 = lds + l4 * WS_LDS_STRIDE + a; gb1 = lds + l4 * WS_LDS_STRIDE + b;
            pair_of(4 * WS_TILE + l15, a, b);
            gas = lds + (4 * bi + l4) * WS_LDS_STRIDE + a; gbs = lds + (4 * bi + l4) * WS_LDS_STRIDE + b;     // k-step bi, as the register sliver
        }
        const bool corner = SLIVER && (bi == bj) && (wave == 0);                    // wave-uniform
        auto load_sliver = [&](int ch, double& a, double& b) {
            const i64 n0 = r0 + (i64)ch * WS_KC;
            if (SLIVER == 1) asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(a) : "v"(voffSA), "s"(A + n0 * lda) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(b) : "v"(voffSB), "s"(B + n0 * ldb) : "memory");
        };
        if (SLIVER) {
            load_sliver(0, asl[0], bsl[0]);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" : "+v"(asl[0]), "+v"(bsl[0]));
        }
        auto stage = [&](auto buf_tag, int ch) {
            constexpr int BUF = decltype(buf_tag)::value;
            double af[2][2], gf[2][2], bf[2][8], cv[2];
            if (SLIVER) load_sliver(ch + 1 < nch ? ch + 1 : ch, asl[BUF ^ 1], bsl[BUF ^ 1]);     // no branch around the asm loads
            d2 ax = (d2){0.0, 0.0}, bx = (d2){0.0, 0.0};
            double cxi = 0.0, cxii = 0.0;
            double asg = 0.0;
            if (SLIVER) {
                if (SLIVER == 2) {
                    const int os = BUF * WS_BUF + bj * 4 * WS_LDS_STRIDE;
                    ax = (d2){ga0[os] * gb0[os], ga1[os] * gb1[os]};
                    asg = gas[BUF * WS_BUF] * gbs[BUF * WS_BUF];
                } else {
                    ax = *reinterpret_cast<const d2*>(a_base_s + BUF * WS_BUF);
                }
                bx = *reinterpret_cast<const d2*>(b_base_s + BUF * WS_BUF);
                cxi = c_base_i[BUF * WS_BUF]; cxii = c_base_ii[BUF * WS_BUF];
            }
            auto read_frags = [&](int kk, int set) {
                const int o = BUF * WS_BUF + kk * 4 * WS_LDS_STRIDE;
                cv[set] = c_base[BUF * WS_BUF + kk * 4];
                if (SLIVER == 2) {
                    af[set][0] = ga0[o]; af[set][1] = ga1[o]; gf[set][0] = gb0[o]; gf[set][1] = gb1[o];
                } else {
                    const d2 va = *reinterpret_cast<const d2*>(a_base + o);
                    af[set][0] = va[0]; af[set][1] = va[1];
                }
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const d2 vb = *reinterpret_cast<const d2*>(b_base + o + 32 * h);
                    bf[set][2 * h] = vb[0]; bf[set][2 * h + 1] = vb[1];
                }
            };
            read_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < WS_KC / 4; ++kk) {
                const int set = kk & 1;
                const double as0 = (SLIVER == 2 ? af[set][0] * gf[set][0] : af[set][0]) * cv[set];
                const double as1 = (SLIVER == 2 ? af[set][1] * gf[set][1] : af[set][1]) * cv[set];
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < WS_KC / 4) read_frags(kk + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(as0, bf[set][n], acc[n], 0, 0, 0);
                    acc[8 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(as1, bf[set][n], acc[8 + n], 0, 0, 0);
                }
                if (SLIVER && kk == 1) {          // the sliver products of this stage, behind the second MFMA block
                    const double sa = (SLIVER == 2 ? asg : asl[BUF]) * cxii;
                    acc_i0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[0] * cxi, bsl[BUF], acc_i0, 0, 0, 0);
                    acc_i1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[1] * cxi, bsl[BUF], acc_i1, 0, 0, 0);
                    acc_ii0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, bx[0], acc_ii0, 0, 0, 0);
                    acc_ii1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, bx[1], acc_ii1, 0, 0, 0);
                    if (corner) acc_c = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, bsl[BUF], acc_c, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (SLIVER) asm volatile("" : "+v"(asl[BUF ^ 1]), "+v"(bsl[BUF ^ 1]));      // consumers stay behind the wait
            __syncthreads();
        };
        for (int ch = 0; ch < nch; ch += 2) {
            if (ch + 1 < nch) issue_stage(ch + 1, 1);
            stage(std::integral_constant<int, 0>{}, ch);
            if (ch + 1 < nch) {
                if (ch + 2 < nch) issue_stage(ch + 2, 0);
                stage(std::integral_constant<int, 1>{}, ch + 1);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(32 * wave + 2 * (l4 + 4 * r) + m) * WS_TILE + 32 * (n >> 1) + 2 * l15 + (n & 1)] = acc[m * 8 + n][r];
        if (SLIVER) {
            // edge tile slots: (bi, 4) takes this workgroup's quarter of A_panel(bi)^T c B_sliver in the column group bj,
            // (4, bj) its quarter of A_sliver^T c B_panel(bj) in the row group bi, (4, 4) the corner quarter in row group bi
            double* oe = partial + ((i64)split * T + (bi * nbb + 4)) * (i64)(WS_TILE * WS_TILE);
            double* of = partial + ((i64)split * T + (4 * nbb + bj)) * (i64)(WS_TILE * WS_TILE);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                oe[(32 * wave + 2 * (l4 + 4 * r)) * WS_TILE + 16 * bj + l15] = acc_i0[r];
                oe[(32 * wave + 2 * (l4 + 4 * r) + 1) * WS_TILE + 16 * bj + l15] = acc_i1[r];
                of[(16 * bi + l4 + 4 * r) * WS_TILE + 32 * wave + 2 * l15] = acc_ii0[r];
                of[(16 * bi + l4 + 4 * r) * WS_TILE + 32 * wave + 2 * l15 + 1] = acc_ii1[r];
            }
            if (corner) {
                double* oc = partial + ((i64)split * T + (4 * nbb + 4)) * (i64)(WS_TILE * WS_TILE);
#pragma unroll
                for (int r = 0; r < 4; ++r) oc[(16 * bi + l4 + 4 * r) * WS_TILE + l15] = acc_c[r];
            }
        }
        return;
    }
    // ragged edge of the operand widths (e.g. 528 = 4 x 128 + 16 columns): 64 x 64 wave tiles in plain 16-column blocks,
    // MFMA tiles that lie wholly past the last column are skipped (wave-uniform), so an edge workgroup costs its loads
    // and little else (the paired-column mapping of the interior loop would keep twice as many tiles alive here)
    int buf = 0;
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) issue_stage(ch + 1, buf ^ 1);
        const double* As = lds + buf * WS_BUF;
        const double* Bs = As + WS_PANEL;
        const double* Cs = As + 2 * WS_PANEL;
        double af[2][4], bf[2][4], cv[2];
        auto read_frags = [&](int kk, int set) {
            const int krow = kk * 4 + l4;
            cv[set] = Cs[krow];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[set][m] = As[krow * WS_LDS_STRIDE + wr * 64 + m * 16 + l15];
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[set][n] = Bs[krow * WS_LDS_STRIDE + wc * 64 + n * 16 + l15];
        };
        read_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < WS_KC / 4; ++kk) {
            const int set = kk & 1;
            double as[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) as[m] = af[set][m] * cv[set];
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < WS_KC / 4) read_frags(kk + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (m < mt_a && n < mt_b)
                        acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[m], bf[set][n], acc[m * 4 + n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): LDS-DMA of the stage has landed
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(wr * 64 + m * 16 + l4 + 4 * r) * WS_TILE + wc * 64 + n * 16 + l15] = acc[m * 4 + n][r];
}

int launch_atb_kron32(lrvb_ctx* c, const double* X31, const double* B, i64 N, const double* cvec_dev, double* C_dev);

__global__ __launch_bounds__(256)
void atb_tiles_to_dense_kernel(const double* __restrict__ tiles, int nbb, i64 PA, i64 PB, double* __restrict__ C, i64 ldc, int sliver)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = blockIdx.y;
    if (j >= PB || i >= PA) return;
    const i64 ti = i / WS_TILE, tj = j / WS_TILE, li = i % WS_TILE, lj = j % WS_TILE;
    const double* tile = tiles + (ti * nbb + tj) * (WS_TILE * WS_TILE);
    if (!sliver || (ti < 4 && tj < 4)) { C[i * ldc + j] = tile[li * WS_TILE + lj]; return; }
    // sliver mode: the edge slots hold four k-quarters side by side (atb_glds_kernel<1>); added in a fixed order
    double s = 0.0;
    if (ti < 4)      for (int g = 0; g < 4; ++g) s += tile[li * WS_TILE + 16 * g + lj];            // (bi, 4): column groups
    else             for (int g = 0; g < 4; ++g) s += tile[(16 * g + li) * WS_TILE + lj];          // (4, bj) and the corner: row groups
    C[i * ldc + j] = s;
}

// C (PA x PB, row-major, ld = PB) = A^T diag(c) B;  cvec_dev carries zero padding past N.
int launch_atb(lrvb_ctx* c, const double* A, i64 PA, const double* B, i64 PB, i64 N,
               const double* cvec_dev, double* C_dev, bool rows_padded) {
    if ((PA % 2) || (PB % 2) || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15))
        LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "atb: operands must have even widths and 16-byte aligned rows");
    const int nba = (int)((PA + WS_TILE - 1) / WS_TILE), nbb = (int)((PB + WS_TILE - 1) / WS_TILE);
    const int T = nba * nbb;
    // 4 x 128 + 16 columns on both sides, 16 finite rows past N: the slivers ride on the 16 interior workgroups
    const bool sliver = rows_padded && PA == 4 * WS_TILE + 16 && PB == 4 * WS_TILE + 16;
    const int TW = sliver ? 16 : T;
    i64 S = (4608 + TW / 2) / TW;
    S = ((S + 7) / 8) * 8;
    i64 max_by_rows = (N / 256 / 8) * 8;
    if (S > max_by_rows) S = max_by_rows;
    if (S > 128) S = 128;
    if (S < 8) S = 8;
    i64 rps = (N + S - 1) / S;
    rps = ((rps + WS_KC - 1) / WS_KC) * WS_KC;
    const i64 tile_elems = (i64)T * WS_TILE * WS_TILE;
    LRVB_TRY(buf_reserve(c, c->tile_part, (size_t)(tile_elems * (S + 1))));
    double* part = c->tile_part.p;
    double* tiles = c->tile_part.p + tile_elems * S;
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    if (sliver)
        hipLaunchKernelGGL(atb_glds_kernel<1>, dim3((unsigned)(S * 16)), dim3(WS_THREADS), 0, c->stream,
                           A, PA, (int)PA, B, PB, (int)PB, N, cvec_dev, (int)S, nba, nbb, rps, part);
    else
        hipLaunchKernelGGL(atb_glds_kernel<0>, dim3((unsigned)(S * T)), dim3(WS_THREADS), 0, c->stream,
                           A, PA, (int)PA, B, PB, (int)PB, N, cvec_dev, (int)S, nba, nbb, rps, part);
    HIP_TRY(hipGetLastError());
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    hipLaunchKernelGGL(wsyrk_reduce_kernel, dim3((unsigned)((tile_elems / 2 + 255) / 256)), dim3(256), 0, c->stream,
                       part, (int)S, tile_elems, tiles);
    HIP_TRY(hipGetLastError());
    dim3 grid((unsigned)((PB + 255) / 256), (unsigned)PA);
    hipLaunchKernelGGL(atb_tiles_to_dense_kernel, grid, dim3(256), 0, c->stream, tiles, nbb, PA, PB, C_dev, PB, sliver ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}


// C (528 x 528) = Xk^T diag(c) B with Xk[n, :] = packed lower triangle of [1, x_n][1, x_n]^T generated on chip from the
// N x 31 matrix X31 (atb_glds_kernel<2>); B is N x 528 with 16 zero rows past N, its rows 16-byte aligned.
int launch_atb_kron32(lrvb_ctx* c, const double* X31, const double* B, i64 N, const double* cvec_dev, double* C_dev) {
    const i64 PA = 4 * WS_TILE + 16, PB = PA;
    if ((((uintptr_t)B) & 15) || (((uintptr_t)X31) & 7)) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "atb: operand alignment");
    const int nba = 5, nbb = 5, T = 25;
    i64 S = 128;
    i64 max_by_rows = (N / 256 / 8) * 8;
    if (S > max_by_rows) S = max_by_rows;
    if (S < 8) S = 8;
    i64 rps = (N + S - 1) / S;
    rps = ((rps + WS_KC - 1) / WS_KC) * WS_KC;
    const i64 tile_elems = (i64)T * WS_TILE * WS_TILE;
    LRVB_TRY(buf_reserve(c, c->tile_part, (size_t)(tile_elems * (S + 1))));
    double* part = c->tile_part.p;
    double* tiles = c->tile_part.p + tile_elems * S;
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    hipLaunchKernelGGL(atb_glds_kernel<2>, dim3((unsigned)(S * 16)), dim3(WS_THREADS), 0, c->stream,
                       X31, (i64)31, (int)PA, B, PB, (int)PB, N, cvec_dev, (int)S, nba, nbb, rps, part);
    HIP_TRY(hipGetLastError());
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    hipLaunchKernelGGL(wsyrk_reduce_kernel, dim3((unsigned)((tile_elems / 2 + 255) / 256)), dim3(256), 0, c->stream,
                       part, (int)S, tile_elems, tiles);
    HIP_TRY(hipGetLastError());
    dim3 grid((unsigned)((PB + 255) / 256), (unsigned)PA);
    hipLaunchKernelGGL(atb_tiles_to_dense_kernel, grid, dim3(256), 0, c->stream, tiles, nbb, PA, PB, C_dev, PB, 1);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// ---- Kronecker-row variant: K4 = sum_n c_n u_n u_n^T,  u_n = packed lower triangle of z_n z_n^T ------------------
// The Gram matrix G^T G of per-observation gradients of an objective that is QUADRATIC IN THE DATA
// (g_n[k] = 1/2 z_n^T M_k z_n + c_k) is M~^T K4 M~ (+ rank-one terms).  z z^T is symmetric, so the virtual operand row
// is its packed lower triangle (column v = a (a + 1) / 2 + b <-> z_na z_nb, b <= a: q (q + 1) / 2 = 2080 columns at q = 64)
// and the caller folds M_k onto the triangle (mtilde_kernel): N Pv (Pv + 1) = 4.3e12 flops for BASELINE.json config 5
// instead of the 1.7e13 of the full 64 q-column Kronecker product rounds 1-3 formed (the same symmetry configuration 3's
// operand uses).  The virtual rows are generated on chip and never touch HBM.  Same tiling and queue order as
// wsyrk_glds_kernel; the stage is the 16 x 64 block of z (zero-padded past q) plus c, and an MFMA operand element is the
// product of two LDS reads at per-lane offsets (the pair (a, b) of its column, computed once per kernel); columns past Pv
// read a constant zero slot.
constexpr int KR_STRIDE = 176;           // [0, 64): z, [64]: c, [65]: 0.0, [80, 144): c z, [145]: 0.0; (stride mod 32) == 16 -> conflict-free
constexpr int KR_ZERO = 65;              // (the row-side operand reads c z_a from the second copy: one multiply per element on either side)
constexpr int KR_CZ = 80;

__device__ __forceinline__ void kr_pair(int v, int pv, int& a, int& b) {
    if (v >= pv) { a = KR_ZERO; b = KR_ZERO; return; }
    a = (int)((sqrtf(8.f * (float)v + 1.f) - 1.f) * 0.5f);
    while (a * (a + 1) / 2 > v) --a;
    while ((a + 1) * (a + 2) / 2 <= v) ++a;
    b = v - a * (a + 1) / 2;
}

__global__ __launch_bounds__(WS_THREADS, 2)
void wsyrk_kron_kernel(const double* __restrict__ Z, i64 ldz, i64 N, int q, int pv,
                       const double* __restrict__ cpad, int n_splits, int nb, i64 rows_per_split,
                       double* __restrict__ partial)
{
    __shared__ double lds[2 * WS_KC * KR_STRIDE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7;
    const int qpos = blockIdx.x >> 3;
    const int S8 = n_splits >> 3;
    const int n_off = nb * (nb - 1) / 2;
    int bi, bj, split_local;
    if (qpos < n_off * S8) {
        split_local = qpos / n_off;
        const int u = qpos - split_local * n_off;
        bi = (int)((1.f + sqrtf(1.f + 8.f * (float)u)) * 0.5f);
        while (bi * (bi - 1) / 2 > u) --bi;
        while ((bi + 1) * bi / 2 <= u) ++bi;
        bj = u - bi * (bi - 1) / 2;
    } else {
        const int qd = qpos - n_off * S8;
        split_local = qd / nb;
        bi = bj = qd - split_local * nb;
    }
    const bool diag = (bi == bj);
    const int t = bi * (bi + 1) / 2 + bj;
    const int T = nb * (nb + 1) / 2;
    const int split = split_local * 8 + xcd;
    i64 r0 = (i64)split * rows_per_split;
    i64 r1 = r0 + rows_per_split;
    if (r1 > N) r1 = N;
    if (r0 > N) r0 = N;
    const int nch = (int)((r1 - r0 + WS_KC - 1) / WS_KC);

    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};

    // stage: thread -> (row = tid >> 4, columns 4 seg .. 4 seg + 3), branch-free clamped loads
    const int srow = tid >> 4, seg = tid & 15;
    double sv[4], sc;
    auto load_stage = [&](int ch) {
        i64 n = r0 + (i64)ch * WS_KC + srow;
        sc = cpad[n];                                     // zero padding past N
        if (n > N - 1) n = N - 1;
        const double* rowp = Z + n * ldz;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int col = 4 * seg + j; sv[j] = rowp[col < q ? col : 0]; }
    };
    auto store_stage = [&](int buf) {
        asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sc));
        double* dst = lds + buf * (WS_KC * KR_STRIDE) + srow * KR_STRIDE;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int col = 4 * seg + j; const double zv = col < q ? sv[j] : 0.0; dst[col] = zv; dst[KR_CZ + col] = zv * sc; }
    };
    if (tid < 2 * WS_KC) { lds[tid * KR_STRIDE + KR_ZERO] = 0.0; lds[tid * KR_STRIDE + KR_CZ + KR_ZERO] = 0.0; }      // the zero slots of every row of both buffers (never rewritten)

    if (nch > 0) { load_stage(0); store_stage(0); }
    __syncthreads();

    const int l15 = lane & 15, l4 = lane >> 4;
    int buf = 0;
    if (!diag) {
        const int wr = wave >> 1, wc = wave & 1;
        // this lane's four columns on the row side (A operand) and on the column side (B operand) of the wave's 64 x 64 block
        int ia[4], ib[4], ja[4], jb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            kr_pair(bi * WS_TILE + wr * 64 + 16 * m + l15, pv, ia[m], ib[m]);
            kr_pair(bj * WS_TILE + wc * 64 + 16 * m + l15, pv, ja[m], jb[m]);
        }
        // 16-row blocks of this wave that hold real columns: fewer than four only in the last tile row (Pv = 2080 = 16 x 128 + 32),
        // whose other blocks are padding and are skipped (wave-uniform); the column side of an off-diagonal tile is always full
        int mt_a = (pv - (bi * WS_TILE + wr * 64) + 15) / 16;
        mt_a = __builtin_amdgcn_readfirstlane(mt_a < 0 ? 0 : (mt_a > 4 ? 4 : mt_a));
        for (int ch = 0; ch < nch; ++ch) {
            const bool more = ch + 1 < nch;
            if (more) load_stage(ch + 1);
            const double* zs = lds + buf * (WS_KC * KR_STRIDE);
            double xa[2][4], xb[2][4], ya[2][4], yb[2][4];
            auto read_frags = [&](int kk, int set) {
                const double* rowp = zs + (kk * 4 + l4) * KR_STRIDE;
#pragma unroll
                for (int m = 0; m < 4; ++m) { xa[set][m] = rowp[KR_CZ + ia[m]]; xb[set][m] = rowp[ib[m]]; ya[set][m] = rowp[ja[m]]; yb[set][m] = rowp[jb[m]]; }
            };
            read_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < WS_KC / 4; ++kk) {
                const int set = kk & 1;
                double af[4], bf[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) { af[m] = xa[set][m] * xb[set][m]; bf[m] = ya[set][m] * yb[set][m]; }
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < WS_KC / 4) read_frags(kk + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (m < mt_a)
                            acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m * 4 + n], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) store_stage(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        double* out = partial + ((i64)split * T + t) * (i64)(WS_TILE * WS_TILE);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(wr * 64 + m * 16 + l4 + 4 * r) * WS_TILE + wc * 64 + n * 16 + l15] = acc[m * 4 + n][r];
    } else {
        const int rb0 = wave, rb1 = 7 - wave;
        const int nreal = (pv - bi * WS_TILE + 15) / 16;             // 16-column blocks of this tile that hold real columns (8 except in the last tile)
        // row blocks rb0 / rb1 (A operand) and all eight column blocks of the tile (B operand; block n is used iff n <= rb)
        int i0a, i0b, i1a, i1b, ja[8], jb[8];
        kr_pair(bi * WS_TILE + 16 * rb0 + l15, pv, i0a, i0b);
        kr_pair(bi * WS_TILE + 16 * rb1 + l15, pv, i1a, i1b);
#pragma unroll
        for (int n = 0; n < 8; ++n) kr_pair(bi * WS_TILE + 16 * n + l15, pv, ja[n], jb[n]);
        for (int ch = 0; ch < nch; ++ch) {
            const bool more = ch + 1 < nch;
            if (more) load_stage(ch + 1);
            const double* zs = lds + buf * (WS_KC * KR_STRIDE);
            double ya[2][8], yb[2][8], a0[2], a1[2];
            auto read_frags = [&](int kk, int set) {
                const double* rowp = zs + (kk * 4 + l4) * KR_STRIDE;
#pragma unroll
                for (int n = 0; n < 8; ++n) { ya[set][n] = rowp[ja[n]]; yb[set][n] = rowp[jb[n]]; }
                a0[set] = rowp[KR_CZ + i0a] * rowp[i0b];
                a1[set] = rowp[KR_CZ + i1a] * rowp[i1b];
            };
            read_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < WS_KC / 4; ++kk) {
                const int set = kk & 1;
                double bf[8];
#pragma unroll
                for (int n = 0; n < 8; ++n) bf[n] = ya[set][n] * yb[set][n];
                const double af0 = a0[set], af1 = a1[set];
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < WS_KC / 4) read_frags(kk + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (n <= rb0 && rb0 < nreal) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af0, bf[n], acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    if (n <= rb1 && rb1 < nreal) acc[4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1, bf[n], acc[4 + n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) store_stage(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        double* out = partial + ((i64)split * T + t) * (i64)(WS_TILE * WS_TILE);
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v0 = (n < 4 && n <= rb0) ? acc[n < 4 ? n : 0][r] : 0.0;
                const double v1 = (n <= rb1) ? acc[4 + n][r] : 0.0;
                out[(rb0 * 16 + l4 + 4 * r) * WS_TILE + n * 16 + l15] = v0;
                out[(rb1 * 16 + l4 + 4 * r) * WS_TILE + n * 16 + l15] = v1;
            }
    }
}

// K4 tiles (virtual dimension Pv = q (q + 1) / 2, nb = ceil(Pv / 128) tile rows) from the context's data matrix.
int launch_wsyrk_kron(lrvb_ctx* c, const double* cvec_dev, double* tiles_out_dev) {
    const int q = (int)c->P;
    if (q > 64) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "Kronecker Gram kernel supports n_cols <= 64 (got %d)", q);
    const int pv = q * (q + 1) / 2;
    const int nb = (pv + WS_TILE - 1) / WS_TILE;
    const int T = nb * (nb + 1) / 2;
    // ~9200 workgroups (18 rounds of the 512 resident ones): the tiles of the packed triangle differ in weight (diagonal tiles, the
    // ragged last tile row), and more, shorter workgroups even that out -- 64 splits 80.4 ms, 32 splits 82.2-82.8 ms at q = 64
    i64 S = (9216 + T / 2) / T;
    S = ((S + 7) / 8) * 8;
    i64 max_by_rows = (c->N / 256 / 8) * 8;
    if (S > max_by_rows) S = max_by_rows;
    if (S > 128) S = 128;
    if (S < 8) S = 8;
    i64 rps = (c->N + S - 1) / S;
    rps = ((rps + WS_KC - 1) / WS_KC) * WS_KC;
    const i64 tile_elems = (i64)T * WS_TILE * WS_TILE;
    LRVB_TRY(buf_reserve(c, c->tile_part, (size_t)(tile_elems * S)));
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    hipLaunchKernelGGL(wsyrk_kron_kernel, dim3((unsigned)(S * T)), dim3(WS_THREADS), 0, c->stream,
                       c->X.p, c->P, c->N, q, pv, cvec_dev, (int)S, nb, rps, c->tile_part.p);
    HIP_TRY(hipGetLastError());
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    const i64 nthreads = tile_elems / 2;
    hipLaunchKernelGGL(wsyrk_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream,
                       c->tile_part.p, (int)S, tile_elems, tiles_out_dev);
    HIP_TRY(hipGetLastError());
    if (c->prof_on) {
        const double pvd = (double)pv;
        c->prof.wsyrk_flops = (double)c->N * pvd * (pvd + 1.0);
        c->prof.wsyrk_bytes = 8.0 * (double)c->N * (q + 1.0) + 8.0 * 0.5 * pvd * (pvd + 1.0);
    }
    return LRVB_OK;
}

// Deterministic second stage: tiles[t][e] = sum_s partial[s][t][e]  (fixed order).
__global__ __launch_bounds__(256)
void wsyrk_reduce_kernel(const double* __restrict__ partial, int n_splits, i64 tile_elems_total,
                         double* __restrict__ tiles)
{
    const i64 e2 = ((i64)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (e2 >= tile_elems_total) return;
    double s0 = 0.0, s1 = 0.0;
    typedef double v2d __attribute__((ext_vector_type(2)));
    for (int s = 0; s < n_splits; ++s) {                       // every partial is read exactly once: streaming loads
        const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(partial + (i64)s * tile_elems_total + e2));
        s0 += v[0]; s1 += v[1];
    }
    *reinterpret_cast<double2*>(tiles + e2) = make_double2(s0, s1);
}

// r[p] = sum_s rpart[s][p]: eight row groups of the block sum every eighth split, then the groups are added in group
// order (a fixed order; one thread per column walking all splits was a 27 us latency chain at 64 splits)
__global__ __launch_bounds__(512)
void rpart_reduce_kernel(const double* __restrict__ rpart, int n_splits, i64 width, i64 P, double* __restrict__ r)
{
    __shared__ double sh[8][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const i64 p = (i64)blockIdx.x * 64 + cx;
    double s0 = 0.0, s1 = 0.0;
    if (p < P) {
        int k = ry;
        for (; k + 8 < n_splits; k += 16) { s0 += rpart[(i64)k * width + p]; s1 += rpart[(i64)(k + 8) * width + p]; }
        for (; k < n_splits; k += 8) s0 += rpart[(i64)k * width + p];
    }
    sh[ry][cx] = s0 + s1;
    __syncthreads();
    if (ry == 0 && p < P) {
        double t = sh[0][cx];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += sh[g][cx];
        r[p] = t;
    }
}

bool wsyrk_fast_path(const lrvb_ctx* c) {
    return c->P > 64 && (c->P % 2) == 0 && ((((uintptr_t)c->X.p) & 15) == 0) && !c->force_generic_wsyrk;
}

int launch_wsyrk(lrvb_ctx* c, const double* cvec_dev, double* tiles_out_dev) {
    return launch_wsyrk_r(c, cvec_dev, tiles_out_dev, nullptr, nullptr);
}

// cy_dev / r_out_dev non-null (fast path only): additionally r = X^T cy (P doubles), accumulated by the diagonal tiles
int launch_wsyrk_r(lrvb_ctx* c, const double* cvec_dev, double* tiles_out_dev, const double* cy_dev, double* r_out_dev) {
    if (cy_dev && !wsyrk_fast_path(c)) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "the column sums ride on the LDS-DMA SYRK kernel only");
    if (c->P <= 64 && !c->force_generic_wsyrk)       // cvec_dev carries zero padding past N (reserve_obs_vec)
        return launch_gram_small(c, cvec_dev, tiles_out_dev);
    const int T = wsyrk_num_tiles(c->P);
    const int S = wsyrk_auto_splits(c);
    i64 rps = (c->N + S - 1) / S;
    rps = ((rps + WS_KC - 1) / WS_KC) * WS_KC;
    if (rps < WS_KC) rps = WS_KC;
    const i64 tile_elems = (i64)T * WS_TILE * WS_TILE;
    const int nbt = (int)((c->P + WS_TILE - 1) / WS_TILE);
    const i64 rwidth = (i64)nbt * WS_TILE;
    LRVB_TRY(buf_reserve(c, c->tile_part, (size_t)(tile_elems * S) + (size_t)(rwidth * S)));
    double* rpart = c->tile_part.p + tile_elems * S;
    const i64 ldx = c->P;
    const int vec_ok = ((ldx % 2) == 0) && ((((uintptr_t)c->X.p) & 15) == 0);
    const int grid = S * T;
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    if (vec_ok && c->P >= 2 && !c->force_generic_wsyrk)     // cvec_dev carries >= 32 zeros past N (reserve_obs_vec)
        hipLaunchKernelGGL(wsyrk_glds_kernel, dim3(grid), dim3(WS_THREADS), 0, c->stream,
                           c->X.p, ldx, c->N, (int)c->P, cvec_dev, S, nbt, rps, c->tile_part.p, cy_dev, rpart);
    else if (vec_ok)
        hipLaunchKernelGGL(wsyrk_kernel<true>, dim3(grid), dim3(WS_THREADS), 0, c->stream,
                           c->X.p, ldx, c->N, (int)c->P, cvec_dev, S, T, rps, c->tile_part.p);
    else
        hipLaunchKernelGGL(wsyrk_kernel<false>, dim3(grid), dim3(WS_THREADS), 0, c->stream,
                           c->X.p, ldx, c->N, (int)c->P, cvec_dev, S, T, rps, c->tile_part.p);
    HIP_TRY(hipGetLastError());
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    const i64 nthreads = tile_elems / 2;
    hipLaunchKernelGGL(wsyrk_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream,
                       c->tile_part.p, S, tile_elems, tiles_out_dev);
    HIP_TRY(hipGetLastError());
    if (cy_dev && r_out_dev) {
        hipLaunchKernelGGL(rpart_reduce_kernel, dim3((unsigned)((c->P + 63) / 64)), dim3(512), 0, c->stream,
                           rpart, S, rwidth, c->P, r_out_dev);
        HIP_TRY(hipGetLastError());
    }
    if (c->prof_on) {
        c->prof.wsyrk_flops = (double)c->N * (double)c->P * (double)(c->P + 1);
        c->prof.wsyrk_bytes = 8.0 * ((double)c->N * (double)(c->P + 1)) + 8.0 * 0.5 * (double)c->P * (double)(c->P + 1);
    }
    return LRVB_OK;
}

// Expand tile-packed lower triangle into a dense symmetric block of `dense` at
// (row_off, col_off); element (a, b), a >= b, lives in tile (a/128, b/128).
__global__ __launch_bounds__(256)
void tiles_to_dense_kernel(const double* __restrict__ tiles, i64 P, double* __restrict__ dense,
                           i64 ld, i64 row_off, i64 col_off, int accumulate)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = blockIdx.y;
    if (j >= P || i >= P) return;
    const i64 a = i > j ? i : j, b = i > j ? j : i;
    const i64 ba = a / WS_TILE, bb = b / WS_TILE;
    const i64 t = ba * (ba + 1) / 2 + bb;
    const double v = tiles[t * (WS_TILE * WS_TILE) + (a % WS_TILE) * WS_TILE + (b % WS_TILE)];
    double* dst = dense + (row_off + i) * ld + col_off + j;
    if (accumulate) *dst += v; else *dst = v;
}

int launch_tiles_to_dense(lrvb_ctx* c, const double* tiles_dev, i64 P, double* dense_dev, i64 ld,
                          i64 row_off, i64 col_off, bool accumulate) {
    if (P <= 0) return LRVB_OK;
    dim3 grid((unsigned)((P + 255) / 256), (unsigned)P);
    hipLaunchKernelGGL(tiles_to_dense_kernel, grid, dim3(256), 0, c->stream,
                       tiles_dev, P, dense_dev, ld, row_off, col_off, accumulate ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
