// k_lmm.hip -- grouped sufficient statistics of hierarchical models (BASELINE.json config 4) in ONE pass over the
// observations: S = Z^T diag(w) Z on the fp64 matrix cores AND the per-group sums [sum_g w | sum_g w z]
// (doc/lmm.lyx:105-160: sum w x x^T, sum w x y, sum w y^2, per group sum w, sum w y, sum w x), from the same registers.
//
// The rows are held in a group-sorted copy Zs (built once, when the group ids are set: a stable counting sort, so the
// rows of a group keep their order), which turns the per-group sums into a segmented reduction over rows in natural
// order: no gather, no atomics, fixed summation order (bitwise reproducible), every row read once at streaming rate.
// Round 2 read X twice (narrow Gram kernel + one wavefront per group walking a permutation with two rows in flight):
// 0.21 + 0.18 + 0.07 ms for the 1.25e6 x 44 shard.
//
// Work split: wave w owns rows [w R, (w + 1) R) of the sorted order.  It walks them in k-steps of 4 rows that never
// straddle a group boundary (rows past the group's end get weight zero and are simply loaded again by the next step),
// so every k-step belongs to one group: lane (i = lane & 15, k = lane >> 4) loads columns (2 i, 2 i + 1) of row k with
// one 16-byte load per 32-column pair group plus one 8-byte load for a trailing block of <= 16 columns -- 44 columns are
// 3 MFMA blocks = 6 tiles per k-step (the 16-byte-only layout of gram_small_kernel needs 4 blocks = 10 tiles).  The
// scaled operand a = w z feeds the MFMA A side and, by one add per block, the lane's group-sum accumulator; at the end
// of a group the four row slots are folded with two cross-lane adds and 16 lanes store the row.  A group cut by a
// wave boundary leaves a piece per wave in `bpart`, summed in wave order by group_fixup_kernel.
#include "lrvb_internal.h"
#include <stdlib.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int LRVB_GS_DEPTH = 6; // k-steps of 4 rows in flight per wave (8 waves per CU: ~68 KB of loads in flight per CU at 44 columns)

// ---- one-time: group-sorted copy of the rows; per call: weights in the same order ------------------------------------
__global__ __launch_bounds__(256)
void permute_rows_kernel(const double* __restrict__ Z, int q, i64 N, const i64* __restrict__ perm, double* __restrict__ Zs)
{
    const int lane = threadIdx.x & 63;
    const i64 row = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N || lane >= q) return;
    Zs[row * q + lane] = Z[perm[row] * q + lane];
}
__global__ void gather_weights_kernel(i64 N, const double* __restrict__ w, const i64* __restrict__ perm, double* __restrict__ ws)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) ws[i] = w[perm[i]];
}

// ---- the fused pass -----------------------------------------------------------------------------------------------------
// What bounds it (measured, round 3: tools/lab ablations of this kernel at N = 1.25e6, q = 44).  The kernel is bound by
// its LOADS: with the MFMAs and the group sums compiled out it ran exactly as long (110 us = 4.0 TB/s) as with them.
// The register-direct layout of gram_small_kernel -- lane (i, k) loads columns (2 i, 2 i + 1) of row k -- splits a
// 4-row k-step of 1408 contiguous bytes into 4 x 256-byte pieces plus 4 x 96-byte pieces that all start in the middle
// of 128-byte lines: 19 line requests for 11 lines of data, and the per-CU budget of outstanding requests, not HBM, set
// the rate (prefetch depth 4 / 6 / 8, two or three waves per SIMD, cached or non-temporal loads: no difference).
// So the rows now travel by LDS-DMA as what they are in memory -- one flat, 128-byte-aligned run per k-step, 16 bytes
// per lane, every line requested once (global_load_lds_dwordx4; the four weights of the step ride behind it) -- into a
// wave-private ring of DEPTH slots, and the MFMA fragments are ds_read from there in the (i, k) layout.  No barrier: a
// slot is written and read by the same wave, ordered by hand-counted s_waitcnt vmcnt.  The k-steps sit at fixed 4-row
// offsets (the Gram does not care about groups); only the ~1 step in 30 that contains a group boundary takes a slow
// path that hands its rows out group by group.  Lanes past the last column read column 0 and leave finite garbage in
// tile entries that are never written out.  A plain step is 3 multiplies + 4 accumulations + 6 MFMAs.
#define GS_DMA16(sbase, voff, ldsaddr) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")
#define GS_DMA4(sbase, voff, ldsaddr) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" \
    :: "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory")
template <int N_> struct GsWait { static __device__ __forceinline__ void vm() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); } };

template <int NPG, int SINGLE>      // NPG pair groups of 32 columns, SINGLE trailing block of <= 16 columns
__global__ __launch_bounds__(256)
void grouped_stats_kernel(const double* __restrict__ Zs, int q, const double* __restrict__ ws,
                          const i64* __restrict__ offs, const i64* __restrict__ wg0, i64 N, i64 R, i64 NW,
                          double* __restrict__ gs, double* __restrict__ bpart /* NW x 2 x 128 */, double* __restrict__ partial)
{
    constexpr int NB = 2 * NPG + SINGLE;
    constexpr int NACC = NB * (NB + 1) / 2;
    constexpr int DEPTH = LRVB_GS_DEPTH;
    constexpr int NDMA = NPG + SINGLE + 1;       // DMA instructions per k-step: ceil(32 q / 1024) for the rows + 1 for the weights
    extern __shared__ double dyn[];              // [4 waves][DEPTH][slot] staging ring; the first 64 x 64 doubles again for the final sum
    double* red = dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const i64 w = (i64)blockIdx.x * 4 + wave;
    const int slot = 4 * q + 4;                  // doubles: the 4 rows of a k-step, then their 4 weights (16-byte multiple: q is even)
    double* ring = dyn + (size_t)wave * DEPTH * slot;

    d4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = (d4){0.0, 0.0, 0.0, 0.0};

    // lane-constant element offsets of the MFMA fragments inside a slot
    bool okp[NPG > 0 ? NPG : 1]; int offp[NPG > 0 ? NPG : 1];
#pragma unroll
    for (int m = 0; m < NPG; ++m) { const int c0 = 32 * m + 2 * li; okp[m] = c0 + 1 < q; offp[m] = lk * q + (okp[m] ? c0 : 0); }
    const int cs = 32 * NPG + li;
    const bool oks = SINGLE && cs < q;
    const int offs_s = lk * q + (oks ? cs : 0);
    const int offw = 4 * q + lk;

    if (w < NW) {
        // Row indices are 32-bit here (the launcher checks N < 2^31): the scalar unit has no ordered 64-bit compare.
        const int r0 = (int)(w * R);                                   // a multiple of 4
        const int r1 = (int)((w + 1) * R < N ? (w + 1) * R : N);
        const int nsteps = (r1 - r0 + 3) >> 2;
        const int step_bytes = 32 * q;
        const int n_full = step_bytes >> 10, rem_lanes = (step_bytes & 1023) >> 4;
        const unsigned lds_ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)ring;
        const unsigned voff16 = (unsigned)lane * 16u, voff4 = (unsigned)lane * 4u;

        // Every slot is (re)loaded whether or not steps remain, so the count of DMA instructions between a slot's loads
        // and its consumption is the same everywhere -- that is what the hand-written vmcnt waits rely on.  A step past
        // the end re-reads the last step's rows with weights taken from the zero padding behind ws.
        auto issue = [&](int k, int step) {
            const bool live = step < nsteps;
            const int row0 = r0 + 4 * (live ? step : nsteps - 1);
            const char* zb = reinterpret_cast<const char*>(Zs + (i64)row0 * q);       // Zs and ws carry 4 zero rows past N
            const char* wb = reinterpret_cast<const char*>(live ? ws + row0 : ws + N);
            const unsigned dst = lds_ring + (unsigned)(k * slot) * 8u;
            if (NPG + SINGLE >= 2) {                         // q > 32: one or two whole 1 KiB instructions
                GS_DMA16(zb, voff16, dst);
                if (n_full == 2) GS_DMA16(zb + 1024, voff16, dst + 1024u);
                else if (lane < rem_lanes) GS_DMA16(zb + 1024, voff16, dst + 1024u);
            } else {                                          // q <= 32: one instruction, whole (q = 32) or partial
                if (n_full == 1) GS_DMA16(zb, voff16, dst);
                else if (lane < rem_lanes) GS_DMA16(zb, voff16, dst);
            }
            if (lane < 8) GS_DMA4(wb, voff4, dst + (unsigned)step_bytes);
        };

        double gp[NPG > 0 ? NPG : 1][2], gsg = 0.0, gw = 0.0;      // this lane's share of the current group's sums
#pragma unroll
        for (int m = 0; m < NPG; ++m) { gp[m][0] = 0.0; gp[m][1] = 0.0; }
        int g = (int)wg0[w], gb = (int)offs[g], gE = (int)offs[g + 1];   // the group that holds the next unassigned row

        auto flush = [&]() {
            // fold the four row slots (lanes li, li + 16, li + 32, li + 48), then lanes 0..15 store the group's row
            gw += __shfl_xor(gw, 16); gw += __shfl_xor(gw, 32);
            if (SINGLE) { gsg += __shfl_xor(gsg, 16); gsg += __shfl_xor(gsg, 32); }
#pragma unroll
            for (int m = 0; m < NPG; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h) { gp[m][h] += __shfl_xor(gp[m][h], 16); gp[m][h] += __shfl_xor(gp[m][h], 32); }
            const bool whole = gb >= r0 && gE <= r1;
            double* dst = whole ? gs + (i64)g * (q + 1) : bpart + (w * 2 + (gb < r0 ? 0 : 1)) * 128;       // a row has q + 1 <= 65 entries
            if (lk == 0) {
                if (li == 0) dst[0] = gw;
#pragma unroll
                for (int m = 0; m < NPG; ++m) if (okp[m]) { dst[1 + 32 * m + 2 * li] = gp[m][0]; dst[2 + 32 * m + 2 * li] = gp[m][1]; }
                if (oks) dst[1 + cs] = gsg;
            }
            gw = 0.0; gsg = 0.0;
#pragma unroll
            for (int m = 0; m < NPG; ++m) { gp[m][0] = 0.0; gp[m][1] = 0.0; }
        };

        auto consume = [&](int k, int step) {
            const int row0 = r0 + 4 * step;
            // the DMA instructions of the DEPTH - 1 younger slots may still be in flight; this slot's have landed
            GsWait<(DEPTH - 1) * NDMA>::vm();
            const double* sl = ring + k * slot;
            const double wk = sl[offw];
            double a[NB > 0 ? NB : 1], b[NB > 0 ? NB : 1];
#pragma unroll
            for (int m = 0; m < NPG; ++m) {
                const v2d t = *reinterpret_cast<const v2d*>(sl + offp[m]);
                b[2 * m] = t[0]; b[2 * m + 1] = t[1];
                a[2 * m] = t[0] * wk; a[2 * m + 1] = t[1] * wk;
            }
            if (SINGLE) { b[2 * NPG] = sl[offs_s]; a[2 * NPG] = b[2 * NPG] * wk; }
            int idx = 0;
#pragma unroll
            for (int ta = 0; ta < NB; ++ta)
#pragma unroll
                for (int tb = 0; tb <= ta; ++tb) {
                    acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[idx], 0, 0, 0);
                    ++idx;
                }
            const int ge = gE < r1 ? gE : r1;                  // this wave's share of the current group ends here
            if (row0 + 4 <= ge - 1 || step >= nsteps) {        // the whole step lies strictly inside the group (or is a dead slot: weights zero)
#pragma unroll
                for (int m = 0; m < NPG; ++m) { gp[m][0] += a[2 * m]; gp[m][1] += a[2 * m + 1]; }
                if (SINGLE) gsg += a[2 * NPG];
                gw += wk;
            } else {
                // a step that contains the end of the group (and possibly whole small groups): hand its rows out group by group
                int pos = row0;
                const int stop = row0 + 4 < r1 ? row0 + 4 : r1;
                while (pos < stop) {
                    const int gend = gE < r1 ? gE : r1;
                    const int seg = gend < stop ? gend : stop;
                    const bool mine = row0 + lk >= pos && row0 + lk < seg;
#pragma unroll
                    for (int m = 0; m < NPG; ++m) { gp[m][0] += mine ? a[2 * m] : 0.0; gp[m][1] += mine ? a[2 * m + 1] : 0.0; }
                    if (SINGLE) gsg += mine ? a[2 * NPG] : 0.0;
                    gw += mine ? wk : 0.0;
                    pos = seg;
                    if (seg == gend) {                         // the group (or this wave's piece of it) is complete
                        flush();
                        if (seg < r1) {                        // next non-empty group (rows remain, so there is one)
                            int g2 = g + 1, b2 = (int)offs[g2], e2 = (int)offs[g2 + 1];
                            while (e2 == b2) { ++g2; b2 = e2; e2 = (int)offs[g2 + 1]; }
                            g = g2; gb = b2; gE = e2;
                        }
                    }
                }
            }
        };

#pragma unroll
        for (int k = 0; k < DEPTH; ++k) issue(k, k);
        for (int s0 = 0; s0 < nsteps; s0 += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) { consume(k, s0 + k); issue(k, s0 + k + DEPTH); }
        }
    }
    GsWait<0>::vm();                 // the DMAs of the dead slots must have landed before the ring is reused below
    __syncthreads();

    // The four wave partials meet in LDS in the MFMA register layout (tile t, register r, lane: conflict-free stores, no
    // index arithmetic), are added in wave order by all 256 threads and leave as ONE block partial of NACC x 256 doubles in
    // that layout -- 12 KB at 44 columns.  The first version scattered each wave into a 64 x 64 block in true column
    // order, one wave after the other, and wrote 32 KB per block: several us of serial epilogue on every workgroup, all
    // of which end at the same time.  The column permutation is undone once, by the last reduction kernel.
    double* stile = dyn;
    double* out = partial + (i64)blockIdx.x * (NACC * 256);
    constexpr int TCH = NACC <= 6 ? NACC : (NACC + 1) / 2;       // tiles per round: 4 waves x TCH x 2 KB <= 48 KB of LDS
#pragma unroll
    for (int t0 = 0; t0 < NACC; t0 += TCH) {
        if (t0) __syncthreads();
#pragma unroll
        for (int t = 0; t < TCH; ++t)
            if (t0 + t < NACC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) stile[((wave * TCH + t) * 4 + r) * 64 + lane] = acc[t0 + t][r];
            }
        __syncthreads();
        for (int e = tid; e < TCH * 256; e += 256)
            if (t0 * 256 + e < NACC * 256)
                out[t0 * 256 + e] = (stile[e] + stile[TCH * 256 + e]) + (stile[2 * TCH * 256 + e] + stile[3 * TCH * 256 + e]);
    }
}

// After the pass, ONE launch: blocks [0, nl1) sum the block partials (slice s of S sums the blocks s, s + S, ...: fixed
// order), the remaining blocks sum the pieces of the groups cut by wave boundaries, in wave order.
__global__ __launch_bounds__(256)
void grouped_post_kernel(const double* __restrict__ partial, int nblk, int elems, int S, double* __restrict__ lvl, int nl1,
                         const i64* __restrict__ offs, i64 G, i64 R, int q, const double* __restrict__ bpart, double* __restrict__ gs)
{
    if ((int)blockIdx.x < nl1) {
        const int per = (elems + 255) / 256;                   // column chunks of 256 elements
        const int sl = blockIdx.x / per;
        const int e = (blockIdx.x - sl * per) * 256 + threadIdx.x;
        if (e >= elems) return;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int b = sl;
        for (; b + 3 * S < nblk; b += 4 * S) {
            a0 += partial[(i64)b * elems + e];
            a1 += partial[(i64)(b + S) * elems + e];
            a2 += partial[(i64)(b + 2 * S) * elems + e];
            a3 += partial[(i64)(b + 3 * S) * elems + e];
        }
        for (; b < nblk; b += S) a0 += partial[(i64)b * elems + e];
        lvl[(i64)sl * elems + e] = (a0 + a1) + (a2 + a3);
        return;
    }
    const int lane = threadIdx.x & 63;
    const i64 g = (i64)(blockIdx.x - nl1) * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const i64 b = offs[g], e = offs[g + 1];
    if (e == b) {                                  // an empty group: nobody else writes its row
        if (lane < q + 1) gs[g * (i64)(q + 1) + lane] = 0.0;
        if (64 + lane < q + 1) gs[g * (i64)(q + 1) + 64 + lane] = 0.0;
        return;
    }
    const i64 wa = b / R, wb = (e - 1) / R;
    if (wa == wb) return;                          // written whole by its wave
    double a = 0.0, a2 = 0.0;                      // entries lane and 64 + lane of the row (q + 1 <= 65)
    for (i64 w = wa; w <= wb; ++w) {
        const double* src = bpart + (w * 2 + (w == wa ? 1 : 0)) * 128;
        a += src[lane]; a2 += src[64 + lane];
    }
    if (lane < q + 1) gs[g * (i64)(q + 1) + lane] = a;
    if (64 + lane < q + 1) gs[g * (i64)(q + 1) + 64 + lane] = a2;
}

// The S slice sums in order, scattered from the MFMA tile layout into the dense q x q matrix (both triangles):
// element (t, r, lane) of tile (ta, tb) is row col_of(ta, (lane >> 4) + 4 r), column col_of(tb, lane & 15).
__global__ __launch_bounds__(256)
void grouped_dense_kernel(const double* __restrict__ lvl, int S, int elems, int npg, int q, double* __restrict__ dense)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int s = 0;
    for (; s + 3 < S; s += 4) {
        a0 += lvl[(i64)s * elems + e]; a1 += lvl[(i64)(s + 1) * elems + e];
        a2 += lvl[(i64)(s + 2) * elems + e]; a3 += lvl[(i64)(s + 3) * elems + e];
    }
    for (; s < S; ++s) a0 += lvl[(i64)s * elems + e];
    const double v = (a0 + a1) + (a2 + a3);
    const int t = e >> 8, r = (e >> 6) & 3, lane = e & 63;
    int ta = 0; while ((ta + 1) * (ta + 2) / 2 <= t) ++ta;          // t = ta (ta + 1) / 2 + tb, tb <= ta
    const int tb = t - ta * (ta + 1) / 2;
    const int ia = (lane >> 4) + 4 * r, ib = lane & 15;
    const int row = ta < 2 * npg ? 32 * (ta >> 1) + 2 * ia + (ta & 1) : 32 * npg + ia;
    const int col = tb < 2 * npg ? 32 * (tb >> 1) + 2 * ib + (tb & 1) : 32 * npg + ib;
    if (row < q && col < q) {                      // entries of lanes past the last column are garbage: never stored
        dense[row * q + col] = v;
        if (ta != tb) dense[col * q + row] = v;
    }
}

// rows per wave of the fused pass: ~3072 waves (12 per CU), a multiple of 4 rows, at least 64
i64 grouped_rows_per_wave(i64 N) {
    i64 R = (N + 3071) / 3072;                    // three waves per SIMD (the kernel's register budget), all resident at once
    R = (R + 3) & ~(i64)3;
    if (R < 64) R = 64;
    return R;
}

bool grouped_fused_supported(const lrvb_ctx* c) {
    return c->P >= 2 && c->P <= 64 && (c->P % 2) == 0 && c->n_groups > 0;
}

// S (q x q dense) and the group sums (G x (q + 1)) of the context's rows and weights; requires grouped_fused_supported.
int launch_grouped_stats_fused(lrvb_ctx* c, double* S_dense_dev, double* gs_dev) {
    const i64 N = c->N, G = c->n_groups;
    const int q = (int)c->P;
    if (N >= ((i64)1 << 31) - 8 || G >= ((i64)1 << 31) - 8) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "grouped statistics: row and group counts must fit in 31 bits");
    const i64 R = grouped_rows_per_wave(N), NW = (N + R - 1) / R;
    const i64* perm = reinterpret_cast<const i64*>(c->groups.p);
    const i64* offs = perm + N;
    const i64* wg0 = offs + (G + 1);
    if (!c->zs_valid) {                                                   // one-time: the group-sorted copy of the rows
        LRVB_TRY(buf_reserve(c, c->Zs, (size_t)(N + 4) * (size_t)q));
        HIP_TRY(hipMemsetAsync(c->Zs.p + (size_t)N * q, 0, (size_t)4 * q * sizeof(double), c->stream));
        hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, c->stream, (const double*)c->X.p, q, N, perm, c->Zs.p);
        HIP_TRY(hipGetLastError());
        c->zs_valid = true;
    }
    const int npg = q / 32, rem = q % 32;
    const int layout = rem == 0 ? npg * 2 : (rem <= 16 ? npg * 2 + 1 : (npg + 1) * 2);     // 2 NPG + SINGLE
    const int NPGk = layout / 2, NBk = layout, NACCk = NBk * (NBk + 1) / 2, elems = NACCk * 256;
    const i64 grid = (NW + 3) / 4;
    const int S = grid >= 64 ? 16 : 1;
    LRVB_TRY(buf_reserve(c, c->tile_part, (size_t)(grid + S) * (size_t)elems));
    LRVB_TRY(buf_reserve(c, c->bpart, (size_t)NW * 256));
    double* lvl = c->tile_part.p + (size_t)grid * elems;
    if (!c->ws_valid) {
        // the weights in group-sorted order.  Weights that the library holds as its own copy (lrvb_set_weights) are sorted
        // once per upload; weights adopted from a caller's buffer (lrvb_set_weights_dev) may change at any time: every call.
        LRVB_TRY(buf_reserve(c, c->ws, (size_t)(N + 4)));
        HIP_TRY(hipMemsetAsync(c->ws.p + N, 0, 4 * sizeof(double), c->stream));
        hipLaunchKernelGGL(gather_weights_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, N, (const double*)c->w.p, perm, c->ws.p);
        HIP_TRY(hipGetLastError());
        c->ws_valid = c->w.owned;
    }
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    size_t lds_bytes = (size_t)4 * LRVB_GS_DEPTH * (size_t)(4 * q + 4) * sizeof(double);
    const int tch = NACCk <= 6 ? NACCk : (NACCk + 1) / 2;
    if (lds_bytes < (size_t)4 * tch * 256 * sizeof(double)) lds_bytes = (size_t)4 * tch * 256 * sizeof(double);
#define GS_LAUNCH(NPG_, SINGLE_) hipLaunchKernelGGL((grouped_stats_kernel<NPG_, SINGLE_>), dim3((unsigned)grid), dim3(256), lds_bytes, c->stream, \
        (const double*)c->Zs.p, q, (const double*)c->ws.p, offs, wg0, N, R, NW, gs_dev, c->bpart.p, c->tile_part.p)
    switch (layout) {
    case 1: GS_LAUNCH(0, 1); break;
    case 2: GS_LAUNCH(1, 0); break;
    case 3: GS_LAUNCH(1, 1); break;
    case 4: GS_LAUNCH(2, 0); break;
    default: LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "grouped statistics: %d columns", q);
    }
#undef GS_LAUNCH
    HIP_TRY(hipGetLastError());
    const int nl1 = S * ((elems + 255) / 256);
    hipLaunchKernelGGL(grouped_post_kernel, dim3((unsigned)(nl1 + (G + 3) / 4)), dim3(256), 0, c->stream, (const double*)c->tile_part.p, (int)grid, elems, S, lvl, nl1,
                       offs, G, R, q, (const double*)c->bpart.p, gs_dev);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(grouped_dense_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream, (const double*)lvl, S, elems, NPGk, q, S_dense_dev);
    HIP_TRY(hipGetLastError());
    if (c->prof_on) {
        LRVB_TRY(prof_mark(c, PROF_WSYRK));
        c->prof.wsyrk_flops = (double)N * (double)q * (double)(q + 1);
        c->prof.wsyrk_bytes = 8.0 * (double)N * (double)(q + 1) + 4.0 * (double)N;
    }
    return LRVB_OK;
}
