// k_mixture_inst0.hip -- instantiations of the per-row mixture kernels for K = 2 .. 12
// (split over four translation units so that they compile in parallel).
#include "k_mixture_rows.h"

int mixture_rows_launch_0(lrvb_ctx* c, int K, unsigned grid, unsigned dgrid, const double* theta_z_dev, int V,
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
    case 2: MX_LAUNCH(2); break;
    case 3: MX_LAUNCH(3); break;
    case 4: MX_LAUNCH(4); break;
    case 5: MX_LAUNCH(5); break;
    case 6: MX_LAUNCH(6); break;
    case 7: MX_LAUNCH(7); break;
    case 8: MX_LAUNCH(8); break;
    case 9: MX_LAUNCH(9); break;
    case 10: MX_LAUNCH(10); break;
    case 11: MX_LAUNCH(11); break;
    case 12: MX_LAUNCH(12); break;
    default: return 0;
    }
#undef MX_LAUNCH
    return 1;
}
