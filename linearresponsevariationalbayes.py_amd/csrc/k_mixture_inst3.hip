// k_mixture_inst3.hip -- instantiations of the per-row mixture kernels for K = 28 .. 32
// (split over four translation units so that they compile in parallel).
#include "k_mixture_rows.h"

int mixture_rows_launch_3(lrvb_ctx* c, int K, unsigned grid, unsigned dgrid, const double* theta_z_dev, int V,
                             const double* lam_dev, double* Amat_dev, i64 lda, double* U_dev, double* gfree_dev,
                             int* bad_dev, int* todo, int* todo_count)
{
#define MX_LAUNCH(KK) do { \
        hipLaunchKernelGGL(mixture_rows_kernel<KK>, dim3(grid), dim3(256), 0, c->stream, \
            theta_z_dev, c->X.p, V, c->w.p, lam_dev, c->N, Amat_dev, U_dev, gfree_dev, c->part_val.p, bad_dev, \
            c->force_dense_rows, todo, todo_count); \
        hipLaunchKernelGGL(mixture_rows_dense_kernel<((KK) <= 8 ? 8 : ((KK) <= 16 ? 16 : 32))>, dim3(dgrid), dim3(256), 0, c->stream, \
            KK, theta_z_dev, c->X.p, V, c->w.p, lam_dev, Amat_dev, lda, bad_dev, todo, todo_count); } while (0)
    switch (K) {
    case 28: MX_LAUNCH(28); break;
    case 29: MX_LAUNCH(29); break;
    case 30: MX_LAUNCH(30); break;
    case 31: MX_LAUNCH(31); break;
    case 32: MX_LAUNCH(32); break;
    default: return 0;
    }
#undef MX_LAUNCH
    return 1;
}
