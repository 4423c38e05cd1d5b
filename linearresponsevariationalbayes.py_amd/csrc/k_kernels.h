// k_kernels.h -- prototypes of the kernels that lrvb_api.hip launches directly (definitions: k_elementwise.hip, k_gauss.hip,
// k_hvec.hip, k_models.hip, k_cg.hip).  gfx950 only.
#pragma once
#include "lrvb_internal.h"

// the Wishart + MVN model's per-coordinate matrices in O(d^2) numbers (wish_entry, k_models.hip)
struct WishartGen { i64 d, ms, ls, inu, vs; double nu, mvm; const double* m; const double* vm; const double* v; };

// k_elementwise.hip
__global__ void upload_kernel(double* __restrict__ dst, const double* __restrict__ slot, i64 n);
__global__ void mul_kernel(i64 n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o);
__global__ void fma3_kernel(i64 n, const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ v,
                            const double* __restrict__ j1, const double* __restrict__ he, double* __restrict__ o);
__global__ void square_kernel(i64 n, const double* __restrict__ a, double* __restrict__ o);
__global__ void transpose_kernel(i64 rows, i64 cols, const double* __restrict__ a, double* __restrict__ o);
__global__ void diag_scale_kernel(i64 D, i64 V, double scale, const double* __restrict__ j1, double* __restrict__ C);
__global__ void symmetrize_lower_kernel(i64 n, double* __restrict__ A, i64 ld);
__global__ void vec_differs_kernel(i64 n, const double* __restrict__ a, const double* __restrict__ b, int* __restrict__ flag);
__global__ void fill_kernel(i64 n, double v, double* __restrict__ o);
__global__ __launch_bounds__(256)
void vec_block_sums_kernel(i64 n, const double* __restrict__ v, double* __restrict__ part);
__global__ void row_scale_rows_kernel(i64 total, i64 Q, const double* __restrict__ rowscale, double alpha, double* __restrict__ C);
__global__ void scale_slice_rows_kernel(i64 total, i64 Q, const double* __restrict__ j1, const double* __restrict__ W, double* __restrict__ Z);
__global__ __launch_bounds__(256)
void sum_partials_kernel(const double* __restrict__ part, i64 n, double* __restrict__ out);
__global__ void rowscale_kernel(i64 n, i64 P, const double* __restrict__ cvec, const double* __restrict__ B, double* __restrict__ o);
__global__ void mul_rows_kernel(i64 n, i64 D, const double* __restrict__ a, const double* __restrict__ v, double* __restrict__ o);
__global__ void fma3_rows_kernel(i64 n, i64 D, const double* __restrict__ g, const double* __restrict__ j2, const double* __restrict__ v,
                                 const double* __restrict__ j1, const double* __restrict__ he, double* __restrict__ o);
__global__ void diag_mul_add_rows_kernel(i64 n, i64 V, double scale, const double* __restrict__ a, const double* __restrict__ u, double* __restrict__ out);
__global__ void scatter_rows_T_kernel(i64 n, i64 Q, i64 V, i64 P, i64 off, const double* __restrict__ Rt, i64 ldr, double* __restrict__ out);
__global__ void rows_dot_kernel(i64 Q, i64 D, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out);
__global__ void rows_axpby_kernel(i64 n, i64 D, const double* __restrict__ alpha, const double* __restrict__ x,
                                  const double* __restrict__ beta, double* __restrict__ y);

// k_gauss.hip
__global__ __launch_bounds__(256)
void gauss_coef_kernel(i64 n, double tau, const double* __restrict__ w, const double* __restrict__ y,
                       double* __restrict__ cw, double* __restrict__ cy, double* __restrict__ cyy_part);
__global__ __launch_bounds__(1024)
void tiles_symv_kernel(const double* __restrict__ tiles, int nb, i64 P, const double* __restrict__ beta, double* __restrict__ part);
__global__ __launch_bounds__(1024)
void gauss_finish_kernel(const double* __restrict__ part, int nb, i64 P, const double* __restrict__ beta, const double* __restrict__ r,
                         const double* __restrict__ cyy_part, int n_cyy, double* __restrict__ value_out, double* __restrict__ g_out);

// k_hvec.hip
__global__ void hvec_symkron_kernel(i64 total, i64 m, int k, const double* __restrict__ A, const double* __restrict__ B, double coef,
                                    double* __restrict__ H, i64 ld, i64 row_off, i64 col_off, int mirror);
__global__ void hvec_add_block_kernel(i64 total, i64 cols, const double* __restrict__ Bk, double* __restrict__ H, i64 ld,
                                      i64 row_off, i64 col_off, int mirror);
__global__ void hvec_add_indexed_kernel(i64 total, i64 nc, const double* __restrict__ Bk, const double* __restrict__ ridx,
                                        const double* __restrict__ cidx, double* __restrict__ H, i64 ld);

// k_models.hip
__global__ __launch_bounds__(256)
void obs_quadform_kernel(const double* __restrict__ Z, i64 ldz, int q, const double* __restrict__ M,
                         const double* __restrict__ cvec, i64 K, i64 n0, i64 n1, double* __restrict__ out);
__global__ __launch_bounds__(256)
void group_sums_kernel(const double* __restrict__ Z, i64 ldz, int q, const double* __restrict__ w,
                       const i64* __restrict__ perm, const i64* __restrict__ offs, i64 n_groups,
                       double* __restrict__ out);
__global__ __launch_bounds__(256)
void lmm_group_kernel(const double* __restrict__ gs, i64 G, int p, const double* __restrict__ par, const double* __restrict__ floc,
                      double* __restrict__ C, int ldc, double* __restrict__ wts, double* __restrict__ part);
__global__ __launch_bounds__(1024)
void lmm_sums_kernel(const double* __restrict__ part, int n_waves, double* __restrict__ sums);
__global__ void mixture_tail_kernel(i64 n, const double* __restrict__ val2, const int* __restrict__ bad, double* __restrict__ tail);
__global__ void mixture_permute_kernel(i64 total, int q, int K, const double* __restrict__ R, double* __restrict__ Rm);
__global__ void mixture_schur_finish_kernel(i64 total, i64 n, const double* __restrict__ Hgg, const double* __restrict__ sc,
                                            const double* __restrict__ dg, const double* __restrict__ S, double* __restrict__ H);
__global__ void mtilde_kernel(const double* __restrict__ M, i64 V, int q, double* __restrict__ Mt /* (64 q) x V */);
__global__ void svec_kernel(const double* __restrict__ S1 /* q x q */, int q, double* __restrict__ sv /* 64 q */);
__global__ void rank_terms_kernel(i64 V, const double* __restrict__ n_obs_dev, const double* __restrict__ t, const double* __restrict__ cvec,
                                  double* __restrict__ A /* V x V, holds M~^T K4 M~ */);
__global__ void dk_coef_kernel(i64 n, int loss, double lik, int m, const double* __restrict__ w, const double* __restrict__ y,
                               const double* __restrict__ T, int Q, double* __restrict__ coef);
__global__ void obs_loss_kernel(i64 n, int loss, double lik, const double* __restrict__ y, const double* __restrict__ z,
                                double* __restrict__ out);
__global__ __launch_bounds__(256)
void gh_logistic_kernel(i64 n, const double* __restrict__ zm, const double* __restrict__ zs, const double* __restrict__ gx,
                        const double* __restrict__ gw, int K, int order, double* __restrict__ val, double* __restrict__ d1, double* __restrict__ d2);
__global__ __launch_bounds__(256)
void logitnormal_coef_kernel(i64 n, const double* __restrict__ mu, const double* __restrict__ vv, const double* __restrict__ y,
                             const double* __restrict__ w, const double* __restrict__ gx, const double* __restrict__ gw, int K,
                             double* __restrict__ a1, double* __restrict__ a2, double* __restrict__ c11, double* __restrict__ c12,
                             double* __restrict__ c22, double* __restrict__ vpart);

__global__ __launch_bounds__(256)
void dirichlet_rowsums_kernel(i64 n, int K, const double* __restrict__ R, double* __restrict__ RS);
__global__ __launch_bounds__(256)
void dirichlet_colsums_kernel(i64 n, int K, const double* __restrict__ R, double* __restrict__ SR);
__global__ void dirichlet_blocksums_kernel(i64 n, int K, const double* __restrict__ RS, double* __restrict__ SRS);
__global__ void dirichlet_schur_finish_kernel(i64 total, i64 n, int K, const double* __restrict__ R, const double* __restrict__ RS,
                                              const double* __restrict__ SR, const double* __restrict__ SRS,
                                              const double* __restrict__ d, const double* __restrict__ gam,
                                              const double* __restrict__ h_diag, const double* __restrict__ h_const,
                                              const double* __restrict__ sc, const double* __restrict__ dg, double* __restrict__ H);

__global__ void wishart_nu_coef_kernel(WishartGen g, i64 pv, double* __restrict__ cnu);
__global__ void scatter_column_kernel(i64 n, const double* __restrict__ x, double* __restrict__ M, i64 ld, i64 col);
__global__ __launch_bounds__(256)
void wishart_sparse_right_kernel(WishartGen g, i64 pv, i64 V, const double* __restrict__ A, i64 lda, double* __restrict__ out, i64 ldo);
__global__ __launch_bounds__(256)
void wishart_sparse_left_kernel(WishartGen g, i64 pv, i64 n, const double* __restrict__ B, i64 ldb, double* __restrict__ out, i64 ldo);

// k_cg.hip
__global__ void cg_multi_alpha_kernel(int Q, double* __restrict__ s);
__global__ __launch_bounds__(256)
void cg_multi_head_kernel(i64 D, i64 it, double tol, const double* __restrict__ j1, const double* __restrict__ R,
                          double* __restrict__ Pm, double* __restrict__ U, double* __restrict__ s, i64 Q);
__global__ __launch_bounds__(256)
void cg_multi_tail_kernel(i64 D, double sq, const double* __restrict__ quadA /* nullable */, const double* __restrict__ j1,
                          const double* __restrict__ j2, const double* __restrict__ g, const double* __restrict__ W,
                          const double* __restrict__ U, const double* __restrict__ Pm, double* __restrict__ X,
                          double* __restrict__ R, const double* __restrict__ s, i64 Q);
