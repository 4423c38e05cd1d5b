/*
 * lrvb_hip.h -- C ABI of liblrvb_hip.so, the MI355X (gfx950) implementation of the
 * LinearResponseVariationalBayes hot path: dense ELBO-Hessian assembly, Hessian-vector
 * product, per-observation gradient matrix G / Gram G^T G, and the linear-response solve
 * (Cholesky and conjugate gradient).
 *
 * The reference (pure Python on autograd) has no FFI; the boundary this library slots in
 * behind is the callable surface of `Objective` / `TwoParameterObjective` /
 * `ParametricSensitivityLinearApproximation` / `ConjugateGradientSolver`.  Every entry point
 * below cites the reference interface it replaces (paths relative to the reference checkout,
 * LRVB/ = LinearResponseVariationalBayes/).
 *
 * Conventions
 *   - Every function returns an int status: 0 = OK, negative = error; the message is then
 *     available from lrvb_last_error() (thread-local string).
 *   - All matrices are C-contiguous (row-major) IEEE fp64.
 *   - Pointers are caller-owned HOST pointers unless the parameter name ends in `_dev`
 *     (then it is a device pointer valid on the context's device).  The library owns only
 *     what lives inside the opaque context.
 *   - A context is bound to one HIP device and one HIP stream; one in-flight call per
 *     context (the reference's Objective is equally non-re-entrant:
 *     LRVB/SparseObjectives.py:131-150).
 *   - "free" = unconstrained flat vector theta (length D); "vector" = constrained flat
 *     vector eta (length V).  Layout = concatenation of blocks in push order
 *     (LRVB/ParameterDictionary.py:39-46, 55-65, 88-99).
 */
#ifndef LRVB_HIP_H
#define LRVB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared below are exported. */
#pragma GCC visibility push(default)


#define LRVB_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------------------- */
#define LRVB_OK                0
#define LRVB_ERR_INVALID      -1   /* bad argument (maps to ValueError on the Python side)  */
#define LRVB_ERR_SIZE         -2   /* wrong vector length (ValueError,
                                      LRVB/ParameterDictionary.py:56-60, 89-93)            */
#define LRVB_ERR_HIP          -3   /* HIP runtime error                                     */
#define LRVB_ERR_STATE        -4   /* call sequence error (e.g. solve before factor)        */
#define LRVB_ERR_NOT_POSDEF   -5   /* Cholesky breakdown (scipy raises LinAlgError)         */
#define LRVB_ERR_UNSUPPORTED  -6   /* layout/model combination not built                    */

/* ---- packing layout (free <-> vector maps) --------------------------------------------
 * One descriptor per parameter block, in ModelParamsDict push order.                     */
#define LRVB_BLOCK_BOX      0  /* ScalarParam/VectorParam/ArrayParam: elementwise box
                                  constraint, LRVB/Parameters.py:31-61                     */
#define LRVB_BLOCK_PSD      1  /* PosDefMatrixParam: log-Cholesky,
                                  LRVB/MatrixParameters.py:101-112                          */
#define LRVB_BLOCK_SIMPLEX  2  /* SimplexParam: row softmax with reference category 0,
                                  LRVB/SimplexParams.py:11-23                               */

typedef struct lrvb_block_desc {
    int32_t kind;        /* LRVB_BLOCK_*                                                    */
    int32_t reserved;
    int64_t free_off;    /* first index in theta                                            */
    int64_t vec_off;     /* first index in eta                                              */
    int64_t free_size;   /* box: n; psd: k(k+1)/2; simplex: rows*(K-1)                       */
    int64_t vec_size;    /* box: n; psd: k(k+1)/2; simplex: rows*K                           */
    int64_t dim0;        /* psd: k; simplex: rows; box: n                                    */
    int64_t dim1;        /* simplex: K; otherwise 0                                          */
    double  lb;          /* box lower bound (-inf allowed); psd: diag_lb                     */
    double  ub;          /* box upper bound (+inf allowed)                                   */
} lrvb_block_desc;

/* ---- model description -----------------------------------------------------------------
 * The objective the context differentiates, written in VECTOR coordinates:
 *
 *   f(eta) = sum_n w_n * loss(y_n, x_n . eta[glm_off : glm_off+n_cols])      (data term)
 *          + quad_scale * ( 1/2 (eta-m)^T A (eta-m) + b^T eta )             (quadratic term)
 *
 * and f_free(theta) = f(eta(theta)).  This replaces the opaque zero-argument closure `fun`
 * of LRVB/SparseObjectives.py:95-129 with a declared model, because a device kernel cannot
 * trace Python.                                                                          */
#define LRVB_LOSS_NONE      0
#define LRVB_LOSS_GAUSSIAN  1  /* loss = 1/2 * lik_info * (y - z)^2                         */
#define LRVB_LOSS_LOGISTIC  2  /* loss = log(1 + e^z) - y z                                  */
#define LRVB_LOSS_POISSON   3  /* loss = e^z - y z                                           */
#define LRVB_LOSS_DATA_ONLY 4  /* no GLM term: the context only holds the weighted data matrix Z
                                  (slot X, n_obs x n_cols) for objectives that are QUADRATIC IN
                                  THE DATA, f = 1/2 tr(Q(eta) Z^T diag(w) Z) + W c(eta) + R(eta)
                                  (Example.ipynb:247-274 and the conjugate-normal configs): the
                                  O(N) work is lrvb_weighted_gram / lrvb_obs_quadform, the
                                  N-independent closed forms stay with the caller.              */

#define LRVB_QUAD_NONE      0
#define LRVB_QUAD_DIAG      1  /* A = diag(a), a has length V                                */
#define LRVB_QUAD_DENSE     2  /* A is V x V symmetric                                       */

typedef struct lrvb_model_desc {
    int32_t n_blocks;
    int32_t loss;                  /* LRVB_LOSS_*                                           */
    const lrvb_block_desc* blocks; /* n_blocks entries                                       */
    int64_t n_obs;                 /* rows of X held by THIS context (its shard)             */
    int64_t n_cols;                /* columns of X                                           */
    int64_t glm_off;               /* offset of the coefficient slice inside eta             */
    double  lik_info;              /* Gaussian precision                                     */
    int32_t quad_kind;             /* LRVB_QUAD_*                                            */
    int32_t reserved;
} lrvb_model_desc;

typedef struct lrvb_ctx lrvb_ctx;

/* data slots for lrvb_set_data */
#define LRVB_SLOT_X        0   /* n_obs x n_cols design                                     */
#define LRVB_SLOT_Y        1   /* n_obs responses                                           */
#define LRVB_SLOT_QUAD_A   2   /* V (diag) or V x V (dense)                                  */
#define LRVB_SLOT_QUAD_M   3   /* V, centre of the quadratic term (default 0)               */
#define LRVB_SLOT_QUAD_B   4   /* V, linear tilt (default 0)                                 */

/* ---- library ------------------------------------------------------------------------- */
int         lrvb_version(void);
const char* lrvb_last_error(void);
int         lrvb_device_count(int* out);

/* ---- context: replaces Objective.__init__ (LRVB/SparseObjectives.py:96-116) ----------- */
int lrvb_ctx_create (lrvb_ctx** out, int device_id, const lrvb_model_desc* model);
int lrvb_ctx_destroy(lrvb_ctx* ctx);
int lrvb_ctx_sync   (lrvb_ctx* ctx);                 /* hipStreamSynchronize on the ctx stream */
/* ---- STREAM ORDERING (the contract of every entry point that takes or returns a DEVICE pointer: the `_dev`
 * functions, lrvb_set_data_dev / lrvb_set_weights_dev, lrvb_allreduce_hessian, the reduce hook) -------------------
 * All device work of a context is issued on ONE HIP stream, "the context's stream", and `_dev` entry points return
 * as soon as that work is queued.  A device operand is read, and a device result written, IN THE ORDER OF THAT
 * STREAM and in no other order.  The caller is responsible for ordering its own producers and consumers against it:
 *   (1) By default the context's stream is a private BLOCKING stream (hipStreamDefault): HIP orders it after all
 *       work already queued on the legacy default stream (handle NULL -- torch's current stream unless the caller
 *       changed it), and orders later default-stream work after it.  A caller that works on the default stream
 *       therefore needs nothing else: operands written by default-stream kernels are visible, results are complete
 *       before a later default-stream kernel or copy reads them.
 *   (2) A caller that works on ANY OTHER stream (torch side streams, hipStreamNonBlocking streams, per-thread
 *       default streams) must do one of: lrvb_ctx_set_stream(ctx, its stream, 1) -- the context then runs ON that
 *       stream; or bracket the calls with lrvb_ctx_wait_stream (before: the context's stream waits for everything
 *       queued so far on the caller's stream) and lrvb_stream_wait_ctx (after: the caller's stream waits for
 *       everything the context has queued); or synchronise on the host (its own stream before, lrvb_ctx_sync after).
 *   (3) Host-pointer entry points (no `_dev` suffix) synchronise the context's stream before they return: their
 *       outputs are complete on return, and their inputs may be reused at once.
 * Buffers adopted with lrvb_set_data_dev / lrvb_set_weights_dev are read by every later call: a caller that
 * overwrites them must order that write after the context's queued work (rule 1 or 2) like any other consumer.
 *
 * lrvb_ctx_set_stream: use_caller_stream != 0 runs the context on the caller-owned HIP stream `hip_stream` (e.g.
 * torch's current stream -- whose handle is NULL for the legacy default stream -- so that kernels, RCCL collectives
 * and the caller's own work are ordered without host synchronisation).  use_caller_stream == 0: back to a private
 * blocking stream.  The previous stream is drained first.                                                          */
int lrvb_ctx_set_stream(lrvb_ctx* ctx, void* hip_stream, int use_caller_stream);
/* Event hand-offs, no host synchronisation: the context's stream waits for the work queued so far on `hip_stream`
 * (call before handing the context operands produced there) ...                                                    */
int lrvb_ctx_wait_stream(lrvb_ctx* ctx, void* hip_stream);
/* ... and `hip_stream` waits for the work the context has queued so far (call before consuming results there).     */
int lrvb_stream_wait_ctx(lrvb_ctx* ctx, void* hip_stream);
int lrvb_ctx_sizes  (lrvb_ctx* ctx, int64_t* D, int64_t* V, int64_t* n_obs);

/* Observations / constants: uploaded once, resident in HBM afterwards.  rows/cols must
 * match the model description.  The `_dev` form adopts (does not copy, does not free) a
 * device buffer, e.g. one a torch tensor owns.                                             */
int lrvb_set_data    (lrvb_ctx* ctx, int slot, const double* host, int64_t rows, int64_t cols);
int lrvb_set_data_dev(lrvb_ctx* ctx, int slot, const double* data_dev, int64_t rows, int64_t cols);
/* Per-observation weights w (Example.ipynb:254, `self.weights`); default all ones.        */
int lrvb_set_weights    (lrvb_ctx* ctx, const double* w, int64_t n);
int lrvb_set_weights_dev(lrvb_ctx* ctx, const double* w_dev, int64_t n);
/* Multiplier of the quadratic term (the `z*y` keyword pass-through of
 * LRVB/test_objectives.py:161-217).                                                        */
int lrvb_set_quad_scale (lrvb_ctx* ctx, double scale);
/* Precision tau of the Gaussian loss 1/2 tau (y - z)^2 (the `lik_info` of the model description) as a settable
 * hyper-parameter: LRVB/ModelSensitivity.py:555-612 takes ANY hyper_par, and a likelihood precision is one of the
 * reference's own examples of it (regression_utils.py:59-132 carries `lik_info` as an argument).                    */
int lrvb_set_lik_info   (lrvb_ctx* ctx, double lik_info);

/* ---- packing: A15-A18 forward maps ---------------------------------------------------- */
/* eta = constrain(theta): ModelParamsDict.set_free + get_vector
 * (LRVB/ParameterDictionary.py:55-63, 97-99)                                               */
int lrvb_constrain  (lrvb_ctx* ctx, const double* free_in, int64_t D, double* vec_out, int64_t V);
/* theta = unconstrain(eta): set_vector + get_free (LRVB/ParameterDictionary.py:64-65, 88-96);
 * LRVB_ERR_INVALID if a value is out of bounds (LRVB/Parameters.py:15-28)                   */
int lrvb_unconstrain(lrvb_ctx* ctx, const double* vec_in, int64_t V, double* free_out, int64_t D);
/* Dense Jacobian d eta / d theta (V x D): ModelParamsDict.free_to_vector_jac(...).todense()
 * (LRVB/ParameterDictionary.py:70-78)                                                       */
int lrvb_free_to_vector_jac(lrvb_ctx* ctx, const double* free_in, int64_t D, double* jac_out);
/* H_free = J^T H_vec J + sum_k g_k d2 eta_k : convert_vector_to_free_hessian
 * (LRVB/Parameters.py:397-424)                                                              */
int lrvb_free_hessian_from_vector(lrvb_ctx* ctx, const double* free_in, const double* g_vec,
                                  const double* H_vec, double* H_free_out);

/* ---- objective in free coordinates ---------------------------------------------------- */
/* Objective.fun_free (LRVB/SparseObjectives.py:120-125)                                    */
int lrvb_value  (lrvb_ctx* ctx, const double* free_in, int64_t D, double* out);
/* Objective.fun_free_grad (LRVB/SparseObjectives.py:152-154); value_out may be NULL        */
int lrvb_grad   (lrvb_ctx* ctx, const double* free_in, int64_t D, double* value_out, double* g_out);
/* Objective.fun_free_hessian (LRVB/SparseObjectives.py:156-158, autograd.hessian at :103).
 * H_out is D x D with leading dimension ld (>= D).  Both triangles are written.            */
int lrvb_hessian(lrvb_ctx* ctx, const double* free_in, int64_t D, double* H_out, int64_t ld);
/* Objective.fun_free_hvp (LRVB/SparseObjectives.py:183-187): out = H(theta) v.  Host-callback optimisers
 * (scipy's cg and trust-ncg, as the reference drives them) call this many times at one point: the point state
 * (eta, packing Jacobian, per-observation curvature) of the last lrvb_hvp / lrvb_hvp_vec call is kept and reused
 * when the next call names the same point and no other entry point of this context ran in between (lrvb_cg_solve and
 * lrvb_cg_solve_multi keep and reuse the state the same way: the reference's ConjugateGradientSolver is built for ONE
 * point and solves for many right-hand sides there, LRVB/ConjugateGradient.py:63-105).  Data
 * installed zero-copy with lrvb_set_data_dev must be installed again after its contents change.               */
int lrvb_hvp    (lrvb_ctx* ctx, const double* free_in, const double* v, int64_t D, double* out);

/* ---- objective in vector coordinates: Objective.fun_vector* (:127-129, 164-174, 189-193) */
int lrvb_value_vec  (lrvb_ctx* ctx, const double* vec_in, int64_t V, double* out);
int lrvb_grad_vec   (lrvb_ctx* ctx, const double* vec_in, int64_t V, double* value_out, double* g_out);
int lrvb_hessian_vec(lrvb_ctx* ctx, const double* vec_in, int64_t V, double* H_out, int64_t ld);
int lrvb_hvp_vec    (lrvb_ctx* ctx, const double* vec_in, const double* v, int64_t V, double* out);

/* Vector-coordinate Hessian assembled ON THE DEVICE from small host blocks, then converted to free
 * coordinates (convert_vector_to_free_hessian, LRVB/Parameters.py:397-424) without a V x V host matrix:
 * the N-independent closed forms of the normal-family ELBOs (LRVB/NormalParams.py, WishartParams.py,
 * ExponentialFamilies.py) are sums of blocks  coef * D^T (A (x) B) D  with k x k matrices A, B and the
 * duplication matrix D of the symmetric-matrix vector form (MatrixParameters.py:16-41), plus small dense
 * blocks.  begin -> add_* ... -> finish.  `mirror` also adds the transposed block at the mirrored position
 * (off-diagonal blocks of a symmetric matrix).  finish: is_free = 1 returns J^T H J + sum_k g_k d2 eta_k
 * (H_out may be NULL: the result stays resident for lrvb_chol_factor_last), is_free = 0 returns H itself. */
int lrvb_hvec_begin(lrvb_ctx* ctx);
int lrvb_hvec_add_block(lrvb_ctx* ctx, const double* block, int64_t rows, int64_t cols, int64_t row_off,
                        int64_t col_off, int mirror);
/* H[rows[a], cols[b]] += block[a, b] (nr x nc, row-major): a dense block scattered over index lists -- the coupled
 * rows of an arrow Hessian are not contiguous.  No index may appear twice in one list.                               */
int lrvb_hvec_add_indexed(lrvb_ctx* ctx, const double* block, int64_t nr, int64_t nc, const int64_t* rows, const int64_t* cols);
int lrvb_hvec_add_symkron(lrvb_ctx* ctx, const double* A, const double* B, int64_t k, double coef,
                          int64_t row_off, int64_t col_off, int mirror);
int lrvb_hvec_finish(lrvb_ctx* ctx, const double* point, int64_t n_in, int is_free, const double* g_vec,
                     double* H_out);
/* The same assembly as ONE call (round 3: a begin / add / add / ... / finish sequence cost a binding call, an upload and a
 * launch per block -- more than the kernels of the small configurations).  `ops` holds n_ops records of 8 int64:
 *   [kind, off, a, b, c, d, e, f]   with `off` the position of the record's operands in `data` (n_data doubles, uploaded once),
 *   kind 0  add_block:    block at off (a x b, row-major), row_off c, col_off d, mirror e;
 *   kind 1  add_indexed:  block at off (a x b), then a row indices and b column indices stored as doubles;
 *   kind 2  add_symkron:  A at off, B at f (both a x a; two records may share an operand), row_off c, col_off d, mirror e,
 *                         the coefficient at data[b].
 * Then exactly lrvb_hvec_finish(point, n_in, is_free, g_vec, H_out).                                                       */
int lrvb_hvec_program(lrvb_ctx* ctx, const int64_t* ops, int64_t n_ops, const double* data, int64_t n_data,
                      const double* point, int64_t n_in, int is_free, const double* g_vec, double* H_out);

/* ---- cross Hessians: TwoParameterObjective.fun_hessian_free1_vector2
 * (LRVB/SparseObjectives.py:429-438) for the two hyper-parameters a declared model has ---- */
/* Rows n0..n1 of G (shape (n1-n0) x D): G[n,:] = d/dtheta of d f/d w_n, i.e. the transpose
 * of the D x N cross Hessian w.r.t. the observation weights (Example.ipynb:425-441).        */
int lrvb_obs_grad(lrvb_ctx* ctx, const double* free_in, int64_t D, int64_t n0, int64_t n1,
                  double* G_out);
/* l(y_n, z_n) for rows n0..n1: the gradient of the objective with respect to the observation weights,
 * TwoParameterObjective.fun_grad2 (LRVB/SparseObjectives.py:381-387) with par2 = the weights in vector
 * coordinates; `point` is the free vector (is_free != 0) or the vector-coordinate point.             */
int lrvb_obs_loss(lrvb_ctx* ctx, const double* point, int64_t n_in, int is_free, int64_t n0, int64_t n1,
                  double* out);
/* Same in vector coordinates ((n1-n0) x V): TwoParameterObjective.fun_vector_hessian21
 * (LRVB/SparseObjectives.py:418-427) with par2 = the weights.                               */
int lrvb_obs_grad_vec(lrvb_ctx* ctx, const double* vec_in, int64_t V, int64_t n0, int64_t n1,
                      double* G_out);
/* Weight sensitivity of moments by linear response, streamed over the observations:
 *   out[n - n0, q] = d m_q / d w_n = -(M H^-1 G^T)[q, n]       ((n1 - n0) x Q, row-major)
 * -- `moment_jac @ ParametricSensitivityLinearApproximation.get_dinput_dhyper()` with
 * hyper_par = the observation weights (LRVB/ModelSensitivity.py:596-606, Example.ipynb:425-441),
 * transposed, without forming the D x N cross Hessian or the D x N sensitivity.  M is Q x D
 * (free) / Q x V (vector) row-major; H is the factor left by lrvb_chol_factor (same coordinates).
 * With M = I (Q = D) the result is the whole sensitivity matrix, transposed.                   */
int lrvb_obs_influence(lrvb_ctx* ctx, const double* free_in, int64_t D, const double* M, int64_t Q,
                       int64_t n0, int64_t n1, double* out);
int lrvb_obs_influence_vec(lrvb_ctx* ctx, const double* vec_in, int64_t V, const double* M, int64_t Q,
                           int64_t n0, int64_t n1, double* out);
/* D x V cross Hessian w.r.t. the linear tilt b of the quadratic term
 * (the `hyper_param @ theta` term of LRVB/test_model_sensitivity.py:56-66).                 */
int lrvb_cross_hessian_tilt(lrvb_ctx* ctx, const double* free_in, int64_t D, double* C_out);
/* ---- cross Hessians and gradients with respect to the OTHER hyper-parameters of the declared objective:
 * TwoParameterObjective.fun_hessian_free1_vector2 / fun_vector_hessian12 / fun_grad2 (LRVB/SparseObjectives.py:381-449)
 * and the `hyper_par` of ParametricSensitivityLinearApproximation (LRVB/ModelSensitivity.py:555-612, whose defining use
 * is PRIOR sensitivity) for eps in
 *   LRVB_HYPER_TILT        b                 (Ph = V)
 *   LRVB_HYPER_QUAD_M      m, the centre / prior mean of the quadratic term (Ph = V)
 *   LRVB_HYPER_QUAD_A      A: its diagonal (LRVB_QUAD_DIAG, Ph = V) or the row-major lower triangle of the symmetric
 *                          matrix (LRVB_QUAD_DENSE, Ph = V (V + 1) / 2, the vector form of LRVB/MatrixParameters.py:16-41)
 *   LRVB_HYPER_QUAD_SCALE  s                 (Ph = 1)
 *   LRVB_HYPER_LIK_INFO    tau of the Gaussian loss (Ph = 1; one pass over the observations, summed over the ranks)
 * all in VECTOR coordinates of the hyper-parameter (a free hyper-parameter chains through its own packing Jacobian on
 * the caller's side: the objective is differentiated once in eps).  `point` is the free vector (is_free != 0) or the
 * vector-coordinate point of the input parameter; C_out is n x Ph row-major (n = D or V), g_out has Ph entries.
 * Closed forms in r = eta - m, A r, b and the data gradient, evaluated on the device with the packing Jacobian of the
 * input applied there.                                                                                              */
#define LRVB_HYPER_TILT        0
#define LRVB_HYPER_QUAD_M      1
#define LRVB_HYPER_QUAD_A      2
#define LRVB_HYPER_QUAD_SCALE  3
#define LRVB_HYPER_LIK_INFO    4
int lrvb_hyper_size(lrvb_ctx* ctx, int kind, int64_t* n_hyper);
int lrvb_cross_hessian_hyper(lrvb_ctx* ctx, int kind, const double* point, int64_t n_in, int is_free,
                             double* C_out, int64_t n_hyper);
int lrvb_hyper_grad(lrvb_ctx* ctx, int kind, const double* point, int64_t n_in, int is_free,
                    double* g_out, int64_t n_hyper);
/* out (D x Q, row-major) = J(theta)^T B for a host matrix B (V x Q): the chain rule d/d theta = J^T d/d eta for cross
 * Hessians whose vector-coordinate form is an N-independent closed form of the caller's (the priors of the model
 * families: LRVB/ExponentialFamilies.py:186-204 `mvn_prior`, `gamma_prior`, `dirichlet_prior`).                       */
int lrvb_jac_t_matmul(lrvb_ctx* ctx, const double* free_in, int64_t D, const double* B, int64_t Q, double* out);
/* Gram matrix G^T G (D x D) of the per-observation gradient matrix; G is generated on chip
 * and never materialised.                                                                   */
int lrvb_gram(lrvb_ctx* ctx, const double* free_in, int64_t D, double* GtG_out, int64_t ld);

/* ---- the non-conjugate logistic term (LRVB/Modeling.py) -----------------------------------
 * Per element i: phi_i = sum_k gh_w[k] log(1 + exp(z_mean[i] + sqrt(2) z_sd[i] gh_x[k])) / sqrt(pi), the Gauss-Hermite
 * value of E log(1 + e^z), z ~ N(z_mean, z_sd^2): get_e_logistic_term_guass_hermite(..., aggregate_all=False),
 * LRVB/Modeling.py:36-52 (log(1 + e^t) in the overflow-free form; the reference's log1p(exp(t)) is inf past t = 709).
 * order >= 1: d1 (n x 2) = [d/dz_mean, d/dz_sd]; order >= 2: d2 (n x 3) = [mean mean, mean sd, sd sd] -- derivatives of
 * the quadrature SUM, i.e. what autograd returns for the reference's expression.  1 <= n_nodes <= 128.
 * get_e_logistic_term (:16-32) is the same sum with nodes draws / sqrt(2) and weights sqrt(pi) / n_draws.             */
int lrvb_gh_logistic(lrvb_ctx* ctx, int64_t n, const double* z_mean, const double* z_sd, const double* gh_x,
                     const double* gh_w, int32_t n_nodes, int32_t order, double* val, double* d1, double* d2);
/* The model that expectation is written for: logistic regression with q(beta_j) = N(mean_j, var_j).  Data term
 *   sum_n w_n ( phi(x_n . mean, sqrt(x_n^2 . var)) - y_n x_n . mean )
 * of the context's X (n_obs x n_cols = P), y and weights, in the coordinates (mean, var): value, gradient (2 P, nullable)
 * and the three P x P blocks [d2/dmean2 | d2/dmean dvar | d2/dvar2] of the Hessian (3 P^2, row-major, nullable):
 * X^T D11 X, X^T D12 (X o X), (X o X)^T D22 (X o X) on the fp64 matrix cores with the per-observation second
 * derivatives of the quadrature sum as weights.  The context must hold X and y (any GLM loss).                       */
int lrvb_logitnormal_terms(lrvb_ctx* ctx, const double* mean, const double* var, int64_t P, const double* gh_x,
                           const double* gh_w, int32_t n_nodes, double* value_out, double* grad_out, double* H_blocks_out);

/* ---- objectives that are quadratic in the data ------------------------------------------
 * S = Z^T diag(w) Z (n_cols x n_cols, both triangles) with the context's current weights: the
 * weighted sufficient statistics sum_n w_n z_n z_n^T that `np.einsum('ni,ij,nj,n', ...)` at
 * Example.ipynb:262 and LRVB/regression_utils.py:59-88 contract on the host.                  */
int lrvb_weighted_gram(lrvb_ctx* ctx, double* S_out, int64_t ld);
/* The same with the sum of the weights, W = sum_n w_n (the -1/2 W log|Lambda| term of Example.ipynb:247-274), formed on the
 * device and summed over the ranks in the SAME reduction as S.                                                       */
int lrvb_weighted_gram_sum(lrvb_ctx* ctx, double* S_out, int64_t ld, double* wsum_out);
/* out[n - n0, k] = 1/2 z_n^T M_k z_n + c_k for K symmetric matrices M_k (K x n_cols x n_cols)
 * and offsets c (K): rows of the cross Hessian d2 f / d w_n d eta_k of such an objective
 * (TwoParameterObjective.fun_vector_hessian21 with par2 = weights,
 * LRVB/SparseObjectives.py:418-427; Example.ipynb:425-441).                                   */
int lrvb_obs_quadform(lrvb_ctx* ctx, const double* M, const double* c, int64_t K,
                      int64_t n0, int64_t n1, double* out);

/* Grouped sufficient statistics for hierarchical models (doc/lmm.lyx:105-160): group ids are set
 * once (0 <= gid < n_groups); lrvb_group_sums returns, with the current weights, an
 * n_groups x (1 + n_cols) matrix [ sum_g w | sum_g w z ].  n_cols <= 64.                        */
int lrvb_set_groups(lrvb_ctx* ctx, const int32_t* gid, int64_t n, int64_t n_groups);
int lrvb_group_sums(lrvb_ctx* ctx, double* out);
/* Both statistics of a hierarchical model in one call, [S = Z^T diag(w) Z (q x q) | group sums (G x (1 + q))], in ONE
 * device buffer that goes to the sum-over-ranks hook once and stays resident for lrvb_lmm_group_terms.  Either host
 * copy may be NULL.  For an even n_cols the rows are read from a group-sorted copy that is built when the group ids and
 * the data are set (one pass instead of two): data adopted with lrvb_set_data_dev must be installed again after its
 * contents change (as for the point state of lrvb_hvp); weights adopted with lrvb_set_weights_dev are re-read by
 * every call.                                                                                                         */
int lrvb_grouped_stats(lrvb_ctx* ctx, double* S_out, double* gs_out);
/* The hierarchical linear mixed model of doc/lmm.lyx:77-160 (y_i ~ N(x_i.beta + u_g[i], 1/tau_y), u_g ~ N(mu, 1/tau_mu),
 * q(u_g) = N(e_g, 1/i_g)): elimination of the 2 G local parameters on the device, from the resident grouped
 * statistics (Z = [x | y], q = p + 1).  par (8 + p) = [E tau_y, E tau_mu, E mu, d E tau_y / d (a_y, b_y), d E tau_mu / d
 * (a_mu, b_mu), lower bound of the i_g, mean of q(beta) (p)]; f_local (2 G) = FREE local parameters [e_g | log(i_g - lb)].
 * out (128 + (p + 5)^2): out[0 .. p) = sum_g e_g sum_g w x; out[64 ..] = sum e_g r_g, sum W_g (e_g^2 + 1/i_g),
 * sum (e_g - E mu)^2 + 1/i_g, sum (e_g - E mu), sum log i_g, sum W_g, squared norm of the free local gradient
 * (r_g = sum_g w y - m . sum_g w x); then M = H_gl H_ll^-1 H_lg on the p + 5 coupled rows [mean of q(beta) | E mu | a_y |
 * b_y | a_mu | b_mu] in vector coordinates of those rows and free coordinates of the local parameters (H_ll is
 * diagonal for this model) -- what the reference would obtain from the D x D autograd Hessian and a sparse solve
 * (LRVB/SparseObjectives.py:581-657).                                                                                  */
int lrvb_lmm_group_terms(lrvb_ctx* ctx, const double* par, int64_t n_par, const double* f_local, int64_t n_local, double* out);
/* ---- a whole step of configurations 2 and 4 as ONE call (round 4) --------------------------------------------------------
 * The N-independent closed forms of these models are affine in the sufficient statistics, with coefficients that depend on
 * theta only.  The caller sends those coefficients (`hp`: P = Lambda^-1, polygamma values, priors -- see csrc/k_lmm.hip for
 * the layout) together with theta in ONE upload; the library forms the statistics (summed over the ranks), combines them
 * with the coefficients in a single-workgroup kernel where they lie, writes the Kronecker block of the information matrix,
 * and converts to free coordinates (LRVB/Parameters.py:397-424) -- no device-to-host copy inside the call.  The result
 * stays in HBM (lrvb_chol_factor_last factors it); H_out / value_out / sums_out may be NULL.
 *
 * lrvb_mvnreg_hessian: MVNParam regression (configuration 2; LRVB/NormalParams.py:6-23, GammaParams.py:4-16,
 *   regression_utils.py:59-132), the context holds the rows [x | y]; idx = vector positions of [mean, vech(information),
 *   shape, rate].  From the third call of one shape on, the launch chain behind the upload is replayed as a captured
 *   hipGraph (a dozen dependent kernels of ~5 us: 117 -> 92 us per step); plain launches under lrvb_profile_enable, a
 *   sum-over-ranks hook or tuning bit 0, and after any change of stream, shape or buffer addresses (re-captured).
 * lrvb_lmm_global_hessian: hierarchical LMM (configuration 4; doc/lmm.lyx:77-160): `data_ctx` holds the rows [x | y] and the
 *   groups, `global_ctx` the packing of the global parameters; idx = vector positions of [mean, vech(information), e_mu,
 *   i_mu, a_y, b_y, a_mu, b_mu]; free_val = [global free parameters | e_1..e_G | log(i_g - info_lb)].  Result: the Schur
 *   complement of the arrow Hessian onto the global block, in free coordinates, in global_ctx.  Both contexts' launches
 *   are queued on data_ctx's stream for the length of the call; global_ctx's own stream waits for it before and after.  */
int lrvb_mvnreg_hessian(lrvb_ctx* ctx, const double* free_in, int64_t D, const double* hp, int64_t n_hp, const int32_t* idx,
                        double* value_out, double* H_out);
int lrvb_lmm_global_hessian(lrvb_ctx* data_ctx, lrvb_ctx* global_ctx, const double* free_val, int64_t n_free, const double* hp,
                            int64_t n_hp, const int32_t* idx, double info_lb, double* sums_out, double* H_out);

/* Mixture models with a SimplexParam row per observation (LRVB/SimplexParams.py:69-175): for every
 * row, on one wavefront, the simplex map and its closed-form Jacobian / Hessian (:33-63), the local
 * (K-1) x (K-1) free Hessian block, its Cholesky factor and A_n = J_n H_nn^-1 J_n^T; then the
 * Schur-complement operand R = sum_n w_n^2 (x~_n (x) x~_n) vec(A_n)^T ((V+1)^2 x K^2) by an MFMA GEMM,
 * the sufficient statistics S64 = [x~ | z]^T diag(w) [x~ | z] (64 x 64, x~ padded to 32, z to 32),
 * val2 = [-sum w z.s, sum w z log z] and the free local gradient (N x (K-1); may be NULL, as may
 * R_out).  s_n = x~_n Lam, Lam = [E log pi; E log phi] ((V+1) x K), x~_n = (1, x_n).
 * V + 1 <= 32, 2 <= K <= 32.  theta_z == NULL evaluates at the simplex logits of the previous call, which
 * stay resident in HBM (the local part of the evaluation point: N (K-1) doubles).                    */
int lrvb_mixture_rows(lrvb_ctx* ctx, int32_t K, const double* theta_z, const double* Lam,
                      double* val2_out, double* gfree_out, double* S64_out, double* R_out);

/* Schur complement of the mixture's global (Dirichlet) block onto itself, assembled on the device:
 *   H_out = diag(scale) Hgg diag(scale) + diag(diag_add) - sym(Jlam^T Rm Jlam),
 * Rm[(j K + k), (j' K + k')] = R[(j q + j'), (k K + k')], n = q K.  R is the (q^2 x K^2) operand of
 * lrvb_mixture_rows (after the all-reduce over shards when there is one); NULL = the result of the last
 * lrvb_mixture_rows call on this context, still resident.  Jlam (n x n) = d vec(Lam) / d free with rows
 * ordered (j, k); Hgg (n x n) the global block without the Schur term; scale / diag_add (n, nullable)
 * the free-coordinate chain rule of a box block (LRVB/Parameters.py:397-424).  Two n^3 MFMA GEMMs
 * replace the host contraction the reference would do with sparse Jacobian lists.  All host pointers. */
int lrvb_mixture_schur(lrvb_ctx* ctx, int32_t K, int32_t q, const double* R, const double* Jlam,
                       const double* Hgg, const double* scale, const double* diag_add, double* H_out);
/* lrvb_mixture_rows without the per-row gradient and WITHOUT copying the Schur operand to the host: it is summed over
 * the ranks (with the other statistics, in one reduction) and stays resident for the Schur assembly below.          */
int lrvb_mixture_stats(lrvb_ctx* ctx, int32_t K, const double* theta_z, const double* Lam, int32_t want_schur,
                       double* val2_out, double* S64_out);
/* lrvb_mixture_schur with both n x n inputs generated on the device from their O(n) description.  The Dirichlet
 * blocks make d vec(Lam) / d alpha and the global Hessian block "diagonal plus a constant per Dirichlet"
 * (LRVB/ExponentialFamilies.py:114-120, DirichletParams.py:19-26): vecs (4 n) = [diagonal of d Lam | diagonal of Hgg |
 * scale | diag_add], consts (2 (K + 1)) = [constant of d Lam per Dirichlet | constant of Hgg per Dirichlet]
 * (Dirichlet 0 = the K mixture weights at indices 0 .. K-1, Dirichlet 1 + k = column k of the (V, K) array at
 * indices K + v K + k).  Uses the operand the last lrvb_mixture_rows / lrvb_mixture_stats call left on the device;
 * the result stays on the device for lrvb_chol_factor_last, H_out may be NULL.                                      */
int lrvb_mixture_schur_dirichlet(lrvb_ctx* ctx, int32_t K, int32_t q, const double* vecs, const double* consts, double* H_out);

/* Gram matrix G^T G (D x D, free coordinates) of the per-observation gradients
 * g_n[k] = 1/2 z_n^T M_k z_n + c_k (K = V matrices, one per vector coordinate, not necessarily symmetric): the Kronecker
 * rows -- the packed lower triangle of z_n z_n^T, q (q + 1) / 2 virtual columns, the matrices folded onto it -- are generated
 * on chip and contracted on the fp64 matrix cores; G (N x D) is never materialised (BASELINE.json config 5).  n_cols <= 64. */
int lrvb_quadform_gram(lrvb_ctx* ctx, const double* M, const double* c, int64_t K,
                       const double* free_in, double* GtG_out, int64_t ld);
/* The same Gram matrix for the Wishart + MVN model (BASELINE.json configuration 5: y_n ~ N(mu, Lambda^-1), q(mu) = MVNParam(d),
 * q(Lambda) = WishartParam(d); LRVB/NormalParams.py:6-23, WishartParams.py:6-35) with the V matrices M_k described by
 * (nu, m, V) alone: they are 0.2 % dense (64 entries per mean coordinate, four per coordinate of V, one dense matrix for nu) and
 * enter the contraction as gathers -- the 8 V (d + 1)^2-byte operand (134 MB at d = 63) is never formed, on either side of PCIe --
 * and with GtG_out == NULL the result stays in HBM as well (lrvb_chol_factor_last factors it; a sharded step then moves
 * nothing over PCIe but c and theta).
 * offsets = vector-coordinate positions of [mean of q(mu), vech(information of q(mu)), nu, vech(V)]; v is d x d row-major;
 * cvec (V) the constants c_k of the per-observation gradient, as for lrvb_quadform_gram.                                  */
int lrvb_wishart_gram(lrvb_ctx* ctx, int64_t d, const int64_t* offsets, double nu, const double* m, const double* v,
                      const double* cvec, const double* free_in, double* GtG_out, int64_t ld);

/* ---- linear-response solve ------------------------------------------------------------ */
/* scipy.linalg.cho_factor at LRVB/ModelSensitivity.py:594 / SparseObjectives.py:539.
 * The factor stays on the device inside the context.                                        */
int lrvb_chol_factor(lrvb_ctx* ctx, const double* H, int64_t D);
/* Factor the Hessian the context last built on the device (no host round trip).            */
int lrvb_chol_factor_last(lrvb_ctx* ctx);
/* scipy.linalg.cho_solve at LRVB/ModelSensitivity.py:600-602: X = H^{-1} B, B is D x nrhs   */
int lrvb_chol_solve (lrvb_ctx* ctx, const double* B, int64_t D, int64_t nrhs, double* X_out);
/* Device-resident forms (H_dev has leading dimension ld; B_dev is D x nrhs, overwritten
 * with the solution; cov_dev is Q x Q).                                                     */
int lrvb_chol_factor_dev(lrvb_ctx* ctx, const double* H_dev, int64_t D, int64_t ld);
int lrvb_chol_solve_dev (lrvb_ctx* ctx, double* B_dev, int64_t D, int64_t nrhs);
int lrvb_lrvb_cov_dev   (lrvb_ctx* ctx, const double* M_dev, int64_t Q, int64_t D, double* cov_dev);
/* LRVB covariance M H^{-1} M^T (Q x Q) for a moment Jacobian M (Q x D)
 * (Example.ipynb:398-415; LRVB/SparseObjectives.py:541-558)                                 */
int lrvb_lrvb_cov   (lrvb_ctx* ctx, const double* M, int64_t Q, int64_t D, double* cov_out);
/* ConjugateGradientSolver.get_hinv_vec (LRVB/ConjugateGradient.py:81-85): solves
 * H(theta) x = b by CG on device HVPs.  Stopping rule of scipy cg with tol (legacy) =
 * rtol, atol = 0: ||b - Hx|| <= tol*||b||; x0 NULL = zeros; Minv NULL = no preconditioner
 * (else a dense D x D approximate inverse, applied as z = Minv r); maxiter <= 0 = 10*D.
 * info_out: 0 converged, >0 = iterations at which it stopped unconverged (scipy's code).    */
int lrvb_cg_solve(lrvb_ctx* ctx, const double* free_in, const double* b, const double* x0,
                  const double* Minv, double tol, int64_t maxiter, int64_t D,
                  double* x_out, int* info_out, int64_t* iters_out);
/* Blocked variant: Q right-hand sides (rows of B, Q x D row-major) advance in lockstep -- the loop over
 * masks of ConjugateGradientSolver.get_hinv_vec_subsets (LRVB/ConjugateGradient.py:87-105) -- so that the
 * Q Hessian-vector products of an iteration share one pair of passes over the observations.  Each row
 * follows exactly the recurrence and stopping rule of lrvb_cg_solve; info_out / iters_out have Q entries. */
int lrvb_cg_solve_multi(lrvb_ctx* ctx, const double* free_in, const double* B, const double* X0 /*nullable*/,
                        const double* Minv /*nullable D x D*/, double tol, int64_t maxiter, int64_t D, int64_t Q,
                        double* X_out, int* info_out, int64_t* iters_out);

/* The same conjugate-gradient loop on a dense symmetric D x D matrix kept on the device (Hessians
 * assembled from sufficient statistics).  H == NULL reuses the matrix of the previous call.   */
int lrvb_cg_solve_matrix(lrvb_ctx* ctx, const double* H, const double* b, const double* x0,
                         const double* Minv, double tol, int64_t maxiter, int64_t D,
                         double* x_out, int* info_out, int64_t* iters_out);

/* ---- device-resident / multi-GPU entry points -----------------------------------------
 * Observations shard over ranks (one process, one context per GPU).  A build is
 *   lrvb_hessian_partial_dev  on every rank  -> stats buffer (this shard's sums)
 *   sum all-reduce of the stats buffer       (torch.distributed / RCCL, outside this lib)
 *   lrvb_hessian_finish_dev   on every rank  -> full free-coordinate Hessian
 * stats layout: [ value (1) | d f_data / d beta (n_cols) | tile-packed lower triangle of
 * X^T diag(w loss'') X (lrvb_stats_size - 1 - n_cols) ] -- sums over observations only; the
 * N-independent quadratic term is added by `finish` on every rank after the reduction.     */
int lrvb_stats_size(lrvb_ctx* ctx, int64_t* n_doubles);
int lrvb_hessian_partial_dev(lrvb_ctx* ctx, const double* free_dev, double* stats_dev);
int lrvb_hessian_finish_dev (lrvb_ctx* ctx, const double* free_dev, const double* stats_dev,
                             double* H_dev, int64_t ld);
/* Single-GPU convenience = partial + finish with everything resident.                      */
int lrvb_hessian_dev(lrvb_ctx* ctx, const double* free_dev, double* H_dev, int64_t ld);

/* Sum-over-ranks hook.  With a hook installed, every sum over observations that the declared-objective entry
 * points form -- [value | gradient] of lrvb_value / lrvb_grad (and their _vec forms), the product of lrvb_hvp /
 * lrvb_hvp_dev, the block products of lrvb_cg_solve_multi, the statistics buffer inside lrvb_hessian /
 * lrvb_hessian_dev, the tiles of lrvb_gram, every product inside lrvb_cg_solve, lrvb_minimize_trust_ncg and
 * lrvb_dk_grad_vec -- is handed to `fn(user, buf_dev, n, hip_stream)` BEFORE the N-independent terms (quadratic
 * term, packing second-order terms) are added.  `fn` must replace buf_dev[0 .. n) by its sum over all ranks, ordered
 * after the work already queued on `hip_stream` (the context's stream) and before whatever is queued on it next
 * -- e.g. an RCCL all-reduce launched on that stream -- and return 0.  Every rank then holds the global value,
 * gradient, product, Hessian and iterates, bit-identical, so the device CG / trust-region loops stay in lockstep
 * with one D-vector all-reduce per product; no counterpart in the reference, which is
 * single-process.  The statistics calls of the other model families are sums over observations too and go through
 * the hook the same way, EXACTLY ONCE PER CALL, on the device buffer, before anything is copied to the host:
 * lrvb_weighted_gram (S), lrvb_group_sums, lrvb_grouped_stats ([S | group sums]), lrvb_mixture_rows /
 * lrvb_mixture_stats ([S64 | val2 | count of indefinite rows | packed Schur operand]: every rank fails together when
 * any rank has an indefinite row), lrvb_quadform_gram ([K4 tiles | s | observation count]) and
 * lrvb_logitnormal_terms ([Hessian blocks | gradient | value], the part that was asked for).  A call must therefore
 * be made by ALL ranks, with the same arguments apart from the rows they hold.  lrvb_hessian_partial_dev and the
 * per-observation row outputs (lrvb_obs_*, the gradient rows of lrvb_mixture_rows) stay rank-local by contract.
 * fn == NULL removes the hook.                                                                                */
typedef int (*lrvb_reduce_fn)(void* user, double* buf_dev, int64_t n, void* hip_stream);
int lrvb_set_reduce_hook(lrvb_ctx* ctx, lrvb_reduce_fn fn, void* user);

/* In-library collective (lrvb_comm_init / lrvb_allreduce_hessian), one process per GPU:
 * rank 0 obtains an id with lrvb_comm_unique_id and hands the 128 bytes to the other ranks by any channel; every rank
 * calls lrvb_comm_init(ctx, world_size, rank, &id) (collective: returns when all ranks have joined), which creates its
 * rank of ONE RCCL communicator on the context's device and installs it as the sum-over-ranks hook above (in-place
 * ncclAllReduce of doubles on the context's stream, RCCL over xGMI).  lrvb_allreduce_hessian sums a statistics buffer
 * of lrvb_hessian_partial_dev explicitly.  librccl is loaded at run time (dlopen), not linked.                     */
typedef struct lrvb_comm_id { char bytes[128]; } lrvb_comm_id;
int lrvb_comm_unique_id(lrvb_comm_id* id_out);
int lrvb_comm_init(lrvb_ctx* ctx, int world_size, int rank, const lrvb_comm_id* id);
int lrvb_comm_destroy(lrvb_ctx* ctx);
int lrvb_allreduce_hessian(lrvb_ctx* ctx, double* stats_dev, int64_t n_doubles);
int lrvb_hvp_dev    (lrvb_ctx* ctx, const double* free_dev, const double* v_dev, double* out_dev);
int lrvb_gram_dev   (lrvb_ctx* ctx, const double* free_dev, double* GtG_dev, int64_t ld);

/* ---- profiling (bench.py's roofline.achieved) ------------------------------------------ */
/* D^j g [u_1 .. u_j]: the j-th directional derivative of the vector-coordinate gradient g = d f / d eta along the
 * rows of U (j x V, row-major; j = `order` in 0..6, j = 0 returns g itself).  The reference's higher-order
 * sensitivity (`ParametricSensitivityTaylorExpansion`, LRVB/ModelSensitivity.py:382-515) builds this from j nested
 * autograd JVPs of the gradient closure (`append_jvp`, :38-62; `generate_two_term_derivative_array`, :221-234).
 * w_override (N, nullable): evaluate with these observation weights instead of the context's -- the derivative
 * of g along a direction in WEIGHT space, since the objective is linear in the weights; include_quad = 0
 * drops the N-independent quadratic term (it does not depend on the weights).  All host pointers.          */
int lrvb_dk_grad_vec(lrvb_ctx* ctx, const double* vec_in, int64_t V, int32_t order, const double* U,
                     const double* w_override, int32_t include_quad, double* out);

/* Trust-region Newton-CG minimisation of the objective in free coordinates, entirely on the device: the
 * optimiser `minimize_objective_trust_ncg` (LRVB/OptimizationUtils.py:44-75) hands to
 * scipy.optimize.minimize(method='trust-ncg') with fun_free / fun_free_grad / fun_free_hvp -- or, with a
 * preconditioner, the `_cond` family (LRVB/SparseObjectives.py:202-240: f(A y), A^T g, A^T H A v).  Same
 * algorithm and constants (Steihaug-Toint CG subproblem; eta, radius rules), so the iterates agree with the
 * scipy route to rounding, but no host callback per Hessian-vector product and ONE curvature pass per point.
 * y0, y_out: iterate in the optimiser's coordinates (x = A y; A = `precond`, D x D row-major, nullable = identity);
 * x_out (nullable): the minimiser in free coordinates.  maxiter <= 0 -> 200 D (scipy's default).
 * status: 0 = ||gradient|| < gtol, 1 = maxiter reached, 2 = the quadratic model predicted no decrease.       */
typedef struct lrvb_opt_result {
    double  fun;             /* objective at the returned point                      */
    double  jac_mag;         /* 2-norm of the (preconditioned) gradient there        */
    double  trust_radius;    /* final radius                                         */
    int32_t status, nit;     /* see above; outer iterations                          */
    int32_t nfev, njev, nhev;/* value / gradient / Hessian-vector evaluations        */
    int32_t nbuild;          /* Hessians built inside the CG runs (256 <= D <= 8192, models with a data term): after
                                max(8, D / 64) products at one point the point's Hessian is built (about D / 86 passes over
                                the observations) and the remaining products of the run are D x D matrix-vector products;
                                0 under tuning bit 3.  (Occupies what was tail padding: the struct size is unchanged.)   */
} lrvb_opt_result;
int lrvb_minimize_trust_ncg(lrvb_ctx* ctx, const double* y0, int64_t D, const double* precond,
                            double gtol, int64_t maxiter, double initial_trust_radius,
                            double max_trust_radius, double eta,
                            double* y_out, double* x_out, lrvb_opt_result* res);

typedef struct lrvb_prof {
    double  wsyrk_ms;       /* HIP-event time of the weighted-SYRK kernel, summed           */
    int64_t wsyrk_calls;
    double  wsyrk_flops;    /* algorithmic flops per launch: n_obs * n_cols * (n_cols + 1)   */
    double  wsyrk_bytes;    /* algorithmic bytes per launch: 8*(n_obs*(n_cols+1)) + tri      */
    double  pass_ms;        /* HIP-event time of the fused value/grad/curvature pass         */
    int64_t pass_calls;
    double  pass_bytes;     /* 8 * n_obs * (n_cols + 3)                                       */
    double  build_ms;       /* whole lrvb_hessian_dev calls                                  */
    int64_t build_calls;
    double  reduce_ms;      /* HIP-event time around the sum-over-ranks hook (the exchange step of SURVEY 8(e)): the
                               collective itself plus, on its first use in a step, the wait for the slowest rank   */
    int64_t reduce_calls;
} lrvb_prof;
int lrvb_profile_enable(lrvb_ctx* ctx, int on);   /* off by default (event overhead)        */
int lrvb_profile_get   (lrvb_ctx* ctx, lrvb_prof* out);
int lrvb_profile_reset (lrvb_ctx* ctx);
/* Tuning knobs: number of row splits of the weighted-SYRK grid (0 = automatic); `reserved` bit 0 =
 * always use the register-staged SYRK kernel, bit 1 = mixture rows always take the dense per-row
 * factorisation, bit 2 = the fused multi-vector pass (blocked CG, streamed influence) with four
 * waves per workgroup instead of eight, bit 3 = never use the resident Hessian (below): products are always passes over
 * the observations.  Every setting computes the same results by another code path
 * (they exist so that tests can compare the paths); any other bit of `reserved` is LRVB_ERR_INVALID.
 *
 * THE RESIDENT HESSIAN.  Every free-coordinate build (lrvb_hessian, lrvb_hessian_dev, lrvb_hessian_finish_dev) leaves a copy
 * of its result inside the context.  lrvb_hvp, lrvb_cg_solve and lrvb_cg_solve_multi asked for the SAME point afterwards form
 * their products H v from that matrix (a D x D product, identical on every rank, no pass over X and no reduction) -- the
 * reference's ConjugateGradientSolver is used exactly so: fun_free_hessian / fun_free_hvp at one optimum, many right-hand
 * sides (LRVB/ConjugateGradient.py:63-105).  The copy is dropped by lrvb_set_data(_dev), lrvb_set_weights(_dev),
 * lrvb_set_quad_scale (new value), lrvb_set_lik_info, lrvb_set_reduce_hook and lrvb_comm_init / _destroy;
 * buffers adopted with the `_dev` setters must be installed again after their contents change.
 * A long run of products at ONE point builds the matrix by itself: past max(8, D / 64) matrix-free products at the point
 * lrvb_hvp / lrvb_cg_solve last named (256 <= D <= 8192, models with a data term) the point's Hessian is built -- about
 * D / 86 passes over the observations -- and made resident; lrvb_minimize_trust_ncg does the same inside its CG runs
 * (lrvb_opt_result.nbuild).  Tuning bit 3 switches both off.                                                            */
int lrvb_set_tuning(lrvb_ctx* ctx, int n_splits, int reserved);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* LRVB_HIP_H */
