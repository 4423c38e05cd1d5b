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

// ---- N-independent closed forms ON THE DEVICE (round 4) -----------------------------------------------------------------------
// Round 3 evaluated the closed forms of configurations 2 and 4 on the host: every step copied the reduced statistics back
// (three synchronous round trips in configuration 4), did 43 x 43 algebra in numpy and sent the blocks up again -- 0.55 ms of
// host time around 0.2 ms of kernels.  The vector-coordinate Hessian and gradient are AFFINE in the statistics with
// coefficients that depend on theta only (P = Lambda^-1, polygamma values, the prior), so the host now sends those
// coefficients ONCE per step, before anything is computed, and one workgroup combines them with the statistics where they
// lie: no device-to-host copy inside a step.
//
// hp (host pack), doubles: [0] ty [1] tm [2] e_mu [3] i_mu [4] a_y [5] b_y [6] a_mu [7] b_mu [8] d ty/d a_y [9] d ty/d b_y
// [10] d tm/d a_mu [11] d tm/d b_mu [12] kappa0 [13] mu0 [14] a0y [15] b0y [16] a0m [17] b0m [18] G [19] psi1(a_y)
// [20] psi2(a_y) [21] psi1(a_mu) [22] psi2(a_mu); from [32]: m (p), beta0 (p), P (p x p), Lambda0 (p x p), P Lambda0 P (p x p).
// sums / Md: the output of the group elimination (lmm_group_kernel + lmm_sums_kernel + Gram), S the q x q weighted Gram.
// Formulas: LMMObjective._arrow / _global_hessian_device of hierarchical.py (doc/lmm.lyx:105-160), pinned against exact AD in
// tests/test_lmm_host_math.py.
// sum over the 1024 threads of the workgroup, the same value in every thread: a butterfly inside each wavefront, the 16 wave
// sums through LDS, added in wave order (two barriers; the ten-round tree through LDS this replaces cost ~2.5 us per sum, and
// the closed-forms kernels take two and four of them)
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sh[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) r += sh[w];
    __syncthreads();
    return r;
}
__device__ __forceinline__ void vech_rc(int k, int& r, int& c) {
    int a = (int)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > k) --a;
    while ((a + 1) * (a + 2) / 2 <= k) ++a;
    r = a; c = k - a * (a + 1) / 2;
}
__global__ __launch_bounds__(1024)
void lmm_closed_forms_kernel(LmmIdx ix, const double* __restrict__ Sg, const double* __restrict__ sums_g, const double* __restrict__ Md,
                             const double* __restrict__ hp, double* __restrict__ g, double* __restrict__ H, double* __restrict__ Gc,
                             const double* __restrict__ part, int n_part, double* __restrict__ sums_out)
{
    extern __shared__ double lds[];                       // S (q^2) | P (p^2) | T (p^2) | PSP (p^2): one global read each, then LDS
    __shared__ double sh[1024];
    __shared__ double um[64];
    __shared__ double sc[16];
    __shared__ double sums[128];
    const int p = ix.p, q = p + 1, tid = threadIdx.x;
    double* S = lds; double* P = S + q * q; double* T = P + p * p; double* PSP = T + p * p;
    for (int e = tid; e < q * q; e += 1024) S[e] = Sg[e];
    for (int e = tid; e < p * p; e += 1024) P[e] = hp[32 + 2 * p + e];
    // the sums over groups: given (sums_g), or formed here from the partial rows of lmm_group_kernel in lmm_sums_kernel's order
    // (eight interleaved slices, then the slices in order; the vector part moved down by one slot) -- one launch less per step
    if (part) {
        const int k = tid & 127, sl = tid >> 7;
        double a0 = 0.0, a1 = 0.0;
        int wv = sl;
        for (; wv + 8 < n_part; wv += 16) { a0 += part[(i64)wv * 128 + k]; a1 += part[(i64)(wv + 8) * 128 + k]; }
        if (wv < n_part) a0 += part[(i64)wv * 128 + k];
        sh[sl * 128 + k] = a0 + a1;
        __syncthreads();
        if (sl == 0) {
            double a = 0.0;
#pragma unroll
            for (int t = 0; t < 8; ++t) a += sh[t * 128 + k];
            if (k < 64) { if (k >= 1) sums[k - 1] = a; if (k == 63) sums[63] = 0.0; }
            else sums[k] = a;
        }
        __syncthreads();
        if (sums_out && tid < 128) sums_out[tid] = sums[tid];
    } else {
        if (tid < 128) sums[tid] = sums_g[tid];
    }
    __syncthreads();
    const i64 ld = ix.ld;
    const double ty = hp[0], tm = hp[1], e_mu = hp[2], i_mu = hp[3], ay = hp[4], by = hp[5], am = hp[6], bm = hp[7];
    const double tay = hp[8], tby = hp[9], tam = hp[10], tbm = hp[11], kappa0 = hp[12], mu0 = hp[13];
    const double a0y = hp[14], b0y = hp[15], a0m = hp[16], b0m = hp[17], Gn = hp[18];
    const double* m = hp + 32; const double* beta0 = m + p; const double* lam0 = beta0 + p + p * p; const double* PL0P = lam0 + p * p;
    // T = Sxx P; um = Sxx m - Sxy + v1; rss; trace(T).  The two p x p x p products of this kernel run on the matrix cores: wave w
    // owns the 16 x 16 tile w of the result (ceil(p / 16)^2 <= 16 tiles), operands straight from LDS (as scalar loops over LDS
    // the two products were about half of the kernel's 27 us at p = 43: one CU's LDS bandwidth)
    const int lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nt = (p + 15) / 16;
    auto tile_product = [&](const double* A_, int lda_, const double* B_, int ldb_, double* C_) {
        if (wv < nt * nt) {
            const int ti = wv / nt, tj = wv - ti * nt;
            const int ai = 16 * ti + l15, bj = 16 * tj + l15;
            typedef double cf_d4 __attribute__((ext_vector_type(4)));
            cf_d4 acc = (cf_d4){0.0, 0.0, 0.0, 0.0};
            for (int kk = 0; 4 * kk < p; ++kk) {
                const int k = 4 * kk + l4;
                const double av = (ai < p && k < p) ? A_[ai * lda_ + k] : 0.0;
                const double bv = (bj < p && k < p) ? B_[k * ldb_ + bj] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + l4 + 4 * r;
                if (i < p && bj < p) C_[i * p + bj] = acc[r];
            }
        }
    };
    tile_product(S, q, P, p, T);
    __syncthreads();
    double tr = 0.0;
    if (tid < p) tr = T[tid * p + tid];
    double rs = 0.0;
    if (tid < p) {
        double a = 0.0;
        for (int k = 0; k < p; ++k) a += S[tid * q + k] * m[k];
        rs = m[tid] * (a - 2.0 * S[tid * q + p]);
        um[tid] = a - S[tid * q + p] + sums[tid];
    }
    const double trT = block_sum_1024(tr, sh);
    const double rss = block_sum_1024(rs, sh) + S[p * q + p];
    // PSP = P T; Gc = ty PSP + P Lambda0 P
    tile_product(P, p, T, p, PSP);
    __syncthreads();
    for (int e = tid; e < p * p; e += 1024) Gc[e] = ty * PSP[e] + PL0P[e];
    if (tid == 0) {
        const double s_eg_rg = sums[64], s_W_e2 = sums[65], s_d2 = sums[66], W = sums[69];
        const double Ay = rss + trT - 2.0 * s_eg_rg + s_W_e2, Am = s_d2 + Gn / i_mu;
        sc[0] = 0.5 * Ay + b0y; sc[1] = 0.5 * Am + b0m;                 // f_ty, f_tm
        sc[2] = -0.5 * W - (a0y - 1.0); sc[3] = -0.5 * Gn - (a0m - 1.0);  // f_Ly, f_Lm
    }
    __syncthreads();
    const double f_ty = sc[0], f_tm = sc[1], f_Ly = sc[2], f_Lm = sc[3], dsum = sums[67];
    const double p1y = hp[19], p2y = hp[20], p1m = hp[21], p2m = hp[22];
    // gamma blocks: gradient (2) and Hessian (2 x 2) in (shape, rate) of f_t a / b + f_L (psi(a) - log b) - entropy
    const double gy0 = f_ty / by + f_Ly * p1y - (1.0 + (1.0 - ay) * p1y), gy1 = -f_ty * ay / (by * by) - f_Ly / by + 1.0 / by;
    const double Hy00 = f_Ly * p2y + p1y - (1.0 - ay) * p2y, Hy01 = -f_ty / (by * by), Hy11 = 2.0 * f_ty * ay / (by * by * by) + f_Ly / (by * by) - 1.0 / (by * by);
    const double gm0 = f_tm / bm + f_Lm * p1m - (1.0 + (1.0 - am) * p1m), gm1 = -f_tm * am / (bm * bm) - f_Lm / bm + 1.0 / bm;
    const double Hm00 = f_Lm * p2m + p1m - (1.0 - am) * p2m, Hm01 = -f_tm / (bm * bm), Hm11 = 2.0 * f_tm * am / (bm * bm * bm) + f_Lm / (bm * bm) - 1.0 / (bm * bm);
    // gradient of the global parameters (vector coordinates): feeds the second-order packing term
    if (tid < p) {
        double a = ty * um[tid];
        for (int k = 0; k < p; ++k) a += lam0[tid * p + k] * (m[k] - beta0[k]);
        g[ix.ms + tid] = a;
    }
    const int mm = p * (p + 1) / 2;
    for (int k = tid; k < mm; k += 1024) {
        int r, c; vech_rc(k, r, c);
        const double fac = r == c ? 1.0 : 2.0;
        g[ix.ls + k] = (-0.5 * Gc[r * p + c] + 0.5 * P[r * p + c]) * fac;
        const double gl = -0.5 * PSP[r * p + c] * fac;                   // d g_vech(Lambda) / d E tau
        H[(i64)(ix.ls + k) * ld + ix.iay] = gl * tay; H[(i64)ix.iay * ld + ix.ls + k] = gl * tay;
        H[(i64)(ix.ls + k) * ld + ix.iby] = gl * tby; H[(i64)ix.iby * ld + ix.ls + k] = gl * tby;
    }
    if (tid == 0) {
        g[ix.iem] = -tm * dsum + kappa0 * (e_mu - mu0);
        g[ix.iim] = -0.5 * (tm * Gn + kappa0) / (i_mu * i_mu) + 0.5 / i_mu;
        g[ix.iay] = gy0; g[ix.iby] = gy1; g[ix.iam] = gm0; g[ix.ibm] = gm1;
    }
    // the dense part of the global block on the p + 6 rows [mean of q(beta) | e_mu, i_mu, a_y, b_y, a_mu, b_mu], the Schur
    // term of the eliminated group parameters (Md, on the p + 5 coupled rows) already subtracted
    const int n6 = p + 6, R = p + 5;
    for (int e = tid; e < n6 * n6; e += 1024) {
        const int a = e / n6, b = e - a * n6;
        auto row_of = [&](int t) { return t < p ? ix.ms + t : (t == p ? ix.iem : (t == p + 1 ? ix.iim : (t == p + 2 ? ix.iay : (t == p + 3 ? ix.iby : (t == p + 4 ? ix.iam : ix.ibm))))); };
        auto m_of = [&](int t) { return t < p ? t : (t == p ? p : (t == p + 1 ? -1 : t - 1)); };
        double v = 0.0;
        const int jem = p, jim = p + 1, jay = p + 2, jby = p + 3, jam = p + 4, jbm = p + 5;
        if (a < p && b < p) v = ty * S[a * q + b] + lam0[a * p + b];
        else if (a < p && b == jay) v = um[a] * tay;
        else if (a < p && b == jby) v = um[a] * tby;
        else if (b < p && a == jay) v = um[b] * tay;
        else if (b < p && a == jby) v = um[b] * tby;
        else if (a == jem && b == jem) v = tm * Gn + kappa0;
        else if ((a == jem && b == jam) || (a == jam && b == jem)) v = -dsum * tam;
        else if ((a == jem && b == jbm) || (a == jbm && b == jem)) v = -dsum * tbm;
        else if (a == jim && b == jim) v = (tm * Gn + kappa0) / (i_mu * i_mu * i_mu) - 0.5 / (i_mu * i_mu);
        else if ((a == jim && b == jam) || (a == jam && b == jim)) v = -0.5 * Gn / (i_mu * i_mu) * tam;
        else if ((a == jim && b == jbm) || (a == jbm && b == jim)) v = -0.5 * Gn / (i_mu * i_mu) * tbm;
        else if (a == jay && b == jay) v = Hy00;
        else if ((a == jay && b == jby) || (a == jby && b == jay)) v = Hy01;
        else if (a == jby && b == jby) v = Hy11;
        else if (a == jam && b == jam) v = Hm00;
        else if ((a == jam && b == jbm) || (a == jbm && b == jam)) v = Hm01;
        else if (a == jbm && b == jbm) v = Hm11;
        const int ma = m_of(a), mb = m_of(b);
        if (ma >= 0 && mb >= 0) v -= Md[ma * R + mb];
        H[(i64)row_of(a) * ld + row_of(b)] = v;
    }
}
static int closed_forms_lds(const void* kernel, size_t bytes) {
    if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return LRVB_OK;
}
// sums: the 128 sums over groups on the device; or part / n_part: the partial rows of lmm_group_kernel, summed inside the kernel
// and written to sums (for the caller's diagnostics)
int launch_lmm_closed_forms(lrvb_ctx* c, const LmmIdx& ix, const double* S, double* sums, const double* Md, const double* hp,
                            double* scratch, double* g, double* H, double* Gc, const double* part, int n_part) {
    (void)scratch;
    const size_t bytes = ((size_t)(ix.p + 1) * (ix.p + 1) + 3 * (size_t)ix.p * ix.p) * sizeof(double);
    LRVB_TRY(closed_forms_lds(reinterpret_cast<const void*>(lmm_closed_forms_kernel), bytes));
    hipLaunchKernelGGL(lmm_closed_forms_kernel, dim3(1), dim3(1024), bytes, c->stream, ix, S, (const double*)sums, Md, hp, g, H, Gc, part, n_part, sums);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// The (k (k + 1) / 2)^2 block of the information matrix of a MVNParam in one launch:
//   D^T ( 1/2 (G (x) P + P (x) G) - 1/2 P (x) P ) D     (three lrvb_hvec_add_symkron calls before)
__device__ __forceinline__ double symkron_entry(const double* __restrict__ A, const double* __restrict__ B, int k, int i, int j, int p, int q) {
    double v = A[i * k + p] * B[j * k + q];
    if (i != j) v += A[j * k + p] * B[i * k + q];
    if (p != q) v += A[i * k + q] * B[j * k + p];
    if (i != j && p != q) v += A[j * k + q] * B[i * k + p];
    return v;
}
__global__ __launch_bounds__(256)
void symkron3_kernel(i64 total, int m, int k, const double* __restrict__ G, const double* __restrict__ P, double* __restrict__ H, i64 ld, i64 off)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int r = (int)(e / m), cidx = (int)(e - (i64)r * m);
    int i, j, p, q;
    vech_rc(r, i, j); vech_rc(cidx, p, q);
    H[(off + r) * ld + off + cidx] = 0.5 * (symkron_entry(G, P, k, i, j, p, q) + symkron_entry(P, G, k, i, j, p, q)) - 0.5 * symkron_entry(P, P, k, i, j, p, q);
}
int launch_symkron3(lrvb_ctx* c, int k, const double* G, const double* P, double* H, i64 ld, i64 off) {
    const int m = k * (k + 1) / 2;
    const i64 total = (i64)m * m;
    hipLaunchKernelGGL(symkron3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, total, m, k, G, P, H, ld, off);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
__global__ void add_padded_kernel(i64 total, i64 n, const double* __restrict__ src, i64 lds, double* __restrict__ dst, i64 ldd) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const i64 i = e / n, j = e - i * n;
    dst[i * ldd + j] = src[i * lds + j];
}
int launch_add_padded(lrvb_ctx* c, i64 n, const double* src, i64 lds, double* dst, i64 ldd) {
    const i64 total = n * n;
    hipLaunchKernelGGL(add_padded_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, total, n, src, lds, dst, ldd);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}

// Configuration 2 (MVNRegressionObjective of quadform.py; LRVB/NormalParams.py:6-23, GammaParams.py:4-16): value, gradient and the
// vector-coordinate Hessian from [S ((k + 1)^2) | W] where they lie.  hp: [0] a [1] b [2] a0 [3] b0 [4] psi(a) [5] psi1(a)
// [6] psi2(a) [7] gammaln(a) [8] log|Lambda|; from [32]: m (k), mu0 (k), P (k x k), Lambda0 (k x k), P Lambda0 P (k x k).
__global__ __launch_bounds__(1024)
void mvnreg_closed_forms_kernel(MvnRegIdx ix, const double* __restrict__ Sg, const double* __restrict__ hp,
                                double* __restrict__ g, double* __restrict__ H, double* __restrict__ Gc, double* __restrict__ value_out)
{
    extern __shared__ double lds[];                       // S (q^2 + 1) | P (k^2) | T (k^2) | PSP (k^2)
    __shared__ double sh[1024];
    __shared__ double u[64];
    const int k = ix.k, q = k + 1, tid = threadIdx.x;
    double* S = lds; double* P = S + q * q + 1; double* T = P + k * k; double* PSP = T + k * k;
    for (int t = tid; t < q * q + 1; t += 1024) S[t] = Sg[t];
    for (int t = tid; t < k * k; t += 1024) P[t] = hp[32 + 2 * k + t];
    __syncthreads();
    const i64 ld = ix.ld;
    const double a = hp[0], b = hp[1], a0 = hp[2], b0 = hp[3], dig = hp[4], p1 = hp[5], p2 = hp[6], lgam = hp[7], logdet = hp[8];
    const double* m = hp + 32; const double* mu0 = m + k; const double* lam0 = mu0 + k + k * k; const double* PL0P = lam0 + k * k;
    const double W = S[q * q];
    const double e = a / b, L = dig - log(b);
    double tr = 0.0, trl = 0.0;
    for (int t = tid; t < k * k; t += 1024) {
        const int i = t / k, j = t - i * k;
        double s = 0.0;
        for (int r = 0; r < k; ++r) s += S[i * q + r] * P[r * k + j];
        T[t] = s;
        if (i == j) tr += s;
        trl += lam0[t] * P[j * k + i];                                  // tr(Lambda0 P)
    }
    double rs = 0.0, pq = 0.0;
    if (tid < k) {
        double s = 0.0, l = 0.0;
        for (int r = 0; r < k; ++r) { s += S[tid * q + r] * m[r]; l += lam0[tid * k + r] * (m[r] - mu0[r]); }
        rs = m[tid] * (s - 2.0 * S[tid * q + k]);
        pq = (m[tid] - mu0[tid]) * l;
        u[tid] = s - S[tid * q + k];
        g[ix.ms + tid] = e * u[tid] + l;
    }
    const double trT = block_sum_1024(tr, sh), trLP = block_sum_1024(trl, sh), prior_q = block_sum_1024(pq, sh);
    const double rss = block_sum_1024(rs, sh) + S[k * q + k];
    for (int t = tid; t < k * k; t += 1024) {
        const int i = t / k, j = t - i * k;
        double s = 0.0;
        for (int r = 0; r < k; ++r) s += P[i * k + r] * T[r * k + j];
        PSP[t] = s;
        Gc[t] = e * s + PL0P[t];
    }
    __syncthreads();
    const double f_e = 0.5 * rss + 0.5 * trT + b0, f_L = -0.5 * W - (a0 - 1.0);
    if (tid == 0) {
        const double entropy_gamma = a - log(b) + lgam + (1.0 - a) * dig;
        if (value_out) *value_out = 0.5 * e * rss + 0.5 * (e * trT + trLP) - 0.5 * W * L + 0.5 * prior_q - (a0 - 1.0) * L + b0 * e + 0.5 * logdet
                                    - entropy_gamma - 0.5 * ((double)k + (double)k * log(2.0 * M_PI));
        g[ix.ia] = f_e / b + f_L * p1 - (1.0 + (1.0 - a) * p1);
        g[ix.ib] = -f_e * a / (b * b) - f_L / b + 1.0 / b;
        H[(i64)ix.ia * ld + ix.ia] = f_L * p2 + p1 - (1.0 - a) * p2;
        H[(i64)ix.ia * ld + ix.ib] = -f_e / (b * b); H[(i64)ix.ib * ld + ix.ia] = -f_e / (b * b);
        H[(i64)ix.ib * ld + ix.ib] = 2.0 * f_e * a / (b * b * b) + f_L / (b * b) - 1.0 / (b * b);
    }
    for (int t = tid; t < k * k; t += 1024) {                            // H[ms, ms] = C = e Sxx + Lambda0
        const int i = t / k, j = t - i * k;
        H[(i64)(ix.ms + i) * ld + ix.ms + j] = e * S[i * q + j] + lam0[t];
    }
    if (tid < k) {
        H[(i64)(ix.ms + tid) * ld + ix.ia] = u[tid] / b; H[(i64)ix.ia * ld + ix.ms + tid] = u[tid] / b;
        H[(i64)(ix.ms + tid) * ld + ix.ib] = -u[tid] * a / (b * b); H[(i64)ix.ib * ld + ix.ms + tid] = -u[tid] * a / (b * b);
    }
    const int mm = k * (k + 1) / 2;
    for (int t = tid; t < mm; t += 1024) {
        int r, c; vech_rc(t, r, c);
        const double fac = r == c ? 1.0 : 2.0;
        g[ix.ls + t] = (-0.5 * Gc[r * k + c] + 0.5 * P[r * k + c]) * fac;
        const double gl = -0.5 * PSP[r * k + c] * fac;
        H[(i64)(ix.ls + t) * ld + ix.ia] = gl / b; H[(i64)ix.ia * ld + ix.ls + t] = gl / b;
        H[(i64)(ix.ls + t) * ld + ix.ib] = -gl * a / (b * b); H[(i64)ix.ib * ld + ix.ls + t] = -gl * a / (b * b);
    }
}
int launch_mvnreg_closed_forms(lrvb_ctx* c, const MvnRegIdx& ix, const double* S, const double* hp, double* scratch,
                               double* g, double* H, double* Gc, double* value_out) {
    (void)scratch;
    const size_t bytes = ((size_t)(ix.k + 1) * (ix.k + 1) + 1 + 3 * (size_t)ix.k * ix.k) * sizeof(double);
    LRVB_TRY(closed_forms_lds(reinterpret_cast<const void*>(mvnreg_closed_forms_kernel), bytes));
    hipLaunchKernelGGL(mvnreg_closed_forms_kernel, dim3(1), dim3(1024), bytes, c->stream, ix, S, hp, g, H, Gc, value_out);
    HIP_TRY(hipGetLastError());
    return LRVB_OK;
}
