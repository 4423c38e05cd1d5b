// k_hvec.hip -- the vector-coordinate Hessian assembled on the device from small blocks: dense blocks, blocks scattered over
// index lists, and  coef * D^T (A (x) B) D  Kronecker blocks of symmetric-matrix parameters (LRVB/MatrixParameters.py:16-41).
#include "lrvb_internal.h"
#include "k_kernels.h"
#include <math.h>

// The N-independent closed forms of the quadratic-in-data objectives are Kronecker products of k x k
// matrices sandwiched between duplication matrices: D^T (A (x) B) D is a k(k+1)/2-square block (4 M entries
// at k = 63) that is cheap to WRITE but expensive to form with dense host algebra.  The host sends A and B.
__global__ void hvec_symkron_kernel(i64 total, i64 m, int k, const double* __restrict__ A, const double* __restrict__ B, double coef,
                                    double* __restrict__ H, i64 ld, i64 row_off, i64 col_off, int mirror)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;                                  // total = m * m (the EW launcher passes the element count first)
    const i64 r = e / m, cidx = e - r * m;
    // vech index -> (i, j), j <= i, row-major lower triangle (SymIndex of LRVB/MatrixParameters.py:16-23)
    int i = (int)((sqrt(8.0 * (double)r + 1.0) - 1.0) * 0.5);
    while ((i64)i * (i + 1) / 2 > r) --i;
    while ((i64)(i + 1) * (i + 2) / 2 <= r) ++i;
    const int j = (int)(r - (i64)i * (i + 1) / 2);
    int p = (int)((sqrt(8.0 * (double)cidx + 1.0) - 1.0) * 0.5);
    while ((i64)p * (p + 1) / 2 > cidx) --p;
    while ((i64)(p + 1) * (p + 2) / 2 <= cidx) ++p;
    const int q = (int)(cidx - (i64)p * (p + 1) / 2);
    double v = A[i * k + p] * B[j * k + q];
    if (i != j) v += A[j * k + p] * B[i * k + q];
    if (p != q) v += A[i * k + q] * B[j * k + p];
    if (i != j && p != q) v += A[j * k + q] * B[i * k + p];
    v *= coef;
    H[(row_off + r) * ld + col_off + cidx] += v;
    if (mirror) H[(col_off + cidx) * ld + row_off + r] += v;
}

__global__ void hvec_add_block_kernel(i64 total, i64 cols, const double* __restrict__ Bk, double* __restrict__ H, i64 ld,
                                      i64 row_off, i64 col_off, int mirror)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;                                  // total = rows * cols
    const i64 r = e / cols, cidx = e - r * cols;
    H[(row_off + r) * ld + col_off + cidx] += Bk[e];
    if (mirror) H[(col_off + cidx) * ld + row_off + r] += Bk[e];
}

// H[rows[a], cols[b]] += block[a, b]: a dense block scattered over index lists (the coupled rows of an arrow Hessian
// are not contiguous).  The index lists travel as doubles behind the block (one upload).
__global__ void hvec_add_indexed_kernel(i64 total, i64 nc, const double* __restrict__ Bk, const double* __restrict__ ridx,
                                        const double* __restrict__ cidx, double* __restrict__ H, i64 ld)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const i64 a = e / nc, b = e - a * nc;
    H[(i64)ridx[a] * ld + (i64)cidx[b]] += Bk[e];
}
