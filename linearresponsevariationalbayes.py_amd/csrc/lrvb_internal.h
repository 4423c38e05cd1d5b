// lrvb_internal.h -- shared declarations of liblrvb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/lrvb_hip.h"

typedef int64_t i64;

// ---- error plumbing -----------------------------------------------------------------
void lrvb_set_error(const char* fmt, ...);
#define LRVB_FAIL(code, ...) do { lrvb_set_error(__VA_ARGS__); return (code); } while (0)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                    \
    lrvb_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    return LRVB_ERR_HIP; } } while (0)
#define LRVB_TRY(expr) do { int s_ = (expr); if (s_ != LRVB_OK) return s_; } while (0)

// ---- geometry of the weighted-SYRK kernel -------------------------------------------
constexpr int WS_TILE = 128;          // output tile edge (block)
constexpr int WS_KC   = 16;           // observations staged per LDS stage
constexpr int WS_THREADS = 256;       // 4 waves, each a 64x64 sub-tile
constexpr int PASS_THREADS = 256;     // fused pass: 4 waves per block
constexpr int PASS_MAX_COLS = 1024;   // register-resident row: 8 x 128 columns

struct DevBuf {
    double* p = nullptr;
    size_t  n = 0;          // capacity in doubles
    bool    owned = true;
};

struct lrvb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool stream_owned = true;
    hipEvent_t ev_order = nullptr;          // hand-off event of lrvb_ctx_wait_stream / lrvb_stream_wait_ctx

    // layout
    std::vector<lrvb_block_desc> blocks;
    lrvb_block_desc* blocks_dev = nullptr;
    DevBuf qg_Mt, qg_T1, qg_Av;    // lrvb_quadform_gram / lrvb_wishart_gram: M~, K4 M~, M~^T K4 M~ (kept between calls: three allocations and a synchronising free per step before)
    DevBuf jtmap; i64 jt_rows = 0, jt_box = 0; size_t jt_lds = 0; int jt_kmax = 0;      // structured J^T product: row table, box entries, dynamic LDS bytes
    DevBuf boxmap; int n_box_blocks = 0; i64 n_box_entries = 0;   // per-entry [free index | vector index | lb | ub] of all box blocks (k_pack.hip: one launch per map)
    i64 D = 0, V = 0;
    bool all_box = true;

    // model
    int loss = LRVB_LOSS_NONE;
    bool data_only = false;
    i64 N = 0, P = 0, glm_off = 0;
    double lik_info = 1.0;
    int quad_kind = LRVB_QUAD_NONE;
    double quad_scale = 1.0;

    // sum-over-ranks hook (lrvb_set_reduce_hook): null = single process
    lrvb_reduce_fn reduce_fn = nullptr; void* reduce_user = nullptr;
    void* comm = nullptr; int comm_world = 1, comm_rank = 0;          // RCCL communicator of lrvb_comm_init (ncclComm_t)

    // resident data
    DevBuf X, y, w, quadA, quadM, quadB;
    bool have_X = false, have_y = false;
    bool x2_ready = false;          // mx_Xk holds X o X of the current design (lrvb_logitnormal_terms)

    // per-evaluation state (device)
    DevBuf theta, eta, j1, j2, vtmp, vtmp2, vtmp3, g_eta, g_free;
    DevBuf lp, cw, zbuf;            // per-observation: loss', w*loss'', linear predictor
    DevBuf cyv, rvec;               // Gaussian shortcut: (c o y) per observation, r = X^T (c o y)
    DevBuf part_vec, part_val;     // fused-pass block partials
    DevBuf red_scratch;            // first level of the block-partial reduction: sums of 16 rows each (pass_reduce_rows_kernel)
    DevBuf stats;                  // [value | g_glm (P) | S tiles]
    DevBuf tile_part;              // weighted-SYRK split partials
    DevBuf Heta, Hfree, Jdense, Tdense, work1;   // dense V x V / D x D scratch
    DevBuf groups; i64 n_groups = 0;   // [perm (N) | offsets (G+1)] as int64
    DevBuf Zs, ws, bpart; bool zs_valid = false, ws_valid = false;   // fused grouped statistics (k_lmm.hip): group-sorted rows (+4 zero rows), weights in that order, pieces of groups cut by wave boundaries
    DevBuf qstats;                 // [S (q x q) | sum w] of lrvb_mvnreg_hessian
    DevBuf gstats; bool gstats_valid = false;   // [S (q x q) | group sums (G x (q+1))] of lrvb_grouped_stats, summed over ranks
    DevBuf mx_theta, mx_lam, mx_A, mx_U, mx_g, mx_Xk, mx_R;   // mixture rows pipeline (kept between calls)
    i64 mx_theta_n = 0;            // simplex logits resident in mx_theta (entries; 0 = none)
    DevBuf cgH; i64 cgH_n = 0;     // dense matrix of lrvb_cg_solve_matrix
    // the free-coordinate Hessian of the last build, kept by the library: products at the SAME point with the SAME data, weights
    // and hyper-parameters (lrvb_hvp, lrvb_cg_solve, lrvb_cg_solve_multi) are D x D matrix products instead of passes over X
    DevBuf Hres, hres_theta; bool hres_valid = false; bool hres_pt_host = false; std::vector<double> hres_pt;
    bool no_resident = false;      // tuning/testing: always take the matrix-free products
    // captured launch chains (hipGraph): the device part of a one-call step is a dozen dependent launches of 3-15 us; replayed as
    // a graph the gaps between them go.  A slot is valid for one shape, one set of buffer addresses (buf_epoch moves whenever a
    // buffer is reallocated or adopted) and one stream.
    struct GraphSlot { hipGraphExec_t exec = nullptr; i64 key[6] = {0, 0, 0, 0, 0, 0}; unsigned long long epoch = 0; hipStream_t stream = nullptr; bool warmed = false, broken = false; };
    GraphSlot mv_graph;            // lrvb_mvnreg_hessian
    unsigned long long buf_epoch = 1;
    i64 pt_products = 0;           // matrix-free products made at the remembered point (lrvb_hvp / lrvb_cg_solve): past
                                   // max(8, D / 64) of them the point's Hessian is built and made resident (lrvb_api.hip)
    DevBuf chol, cholW;            // D x D Cholesky factor (lower); inverses of its 64 x 64 diagonal blocks
    DevBuf hprog;                  // operands of an lrvb_hvec_program call
    bool chol_valid = false;
    bool hvec_open = false;        // between lrvb_hvec_begin and lrvb_hvec_finish
    i64 chol_n = 0;
    DevBuf rhs, cgx, cgr, cgp, cgq, cgz, scal;
    // host-callback optimisers call lrvb_hvp many times at ONE point: the point state (eta, J, g_eta, curvature)
    // of the last lrvb_hvp / lrvb_hvp_vec call is reused when the next call names the same point and no other
    // entry point ran in between (every entry point clears the flag in ctx_bind)
    std::vector<double> hvp_pt; bool hvp_pt_valid = false; bool hvp_pt_free = false;
    bool hvp_pt_prepared = false;   // the dense packing Jacobian / third-order matrix of general layouts are built too
    DevBuf dkw;                    // lrvb_dk_grad_vec: the caller's weight direction (N)
    DevBuf opt;                    // trust-region Newton-CG: 12 D-vectors (+ the D x D preconditioner)
    DevBuf cgm[9];                 // blocked CG: B, X, R, P, Q, Z (Q x D), U, W (Q x V), R^T (P x Q)
    DevBuf cgT;                    // N x Q products X U^T of the blocked HVP
    const double* hm_live = nullptr;   // blocked CG: device flags of the systems still running (launch_hvp_multi passes them on)
    hipStream_t aux_stream = nullptr; hipEvent_t aux_ev[2] = {nullptr, nullptr};   // side stream for the CG status read-back
    DevBuf gpad;                   // even-width zero-padded copies of odd-width TN GEMM operands
    DevBuf ones; i64 ones_n = 0;   // [1 x n | 0 x 64] contraction weights of the plain TN GEMM
    double* host_pinned = nullptr; size_t host_pinned_n = 0;
    // small host -> device uploads: a ring of pinned slots, so that the copy is a real asynchronous copy in stream order
    // and the call does not have to synchronise the stream (a pageable source has to be consumed before the call returns)
    static constexpr int UP_SLOTS = 16; static constexpr size_t UP_SLOT_DOUBLES = 65536;   // 512 KB per slot: the 2 G local parameters of config 4 fit one
    double* up_ring = nullptr; double* up_ring_dev = nullptr; hipEvent_t up_ev[UP_SLOTS] = {}; int up_next = 0;

    int n_splits_user = 0;
    bool force_generic_wsyrk = false;   // tuning/testing: use the register-staged kernel
    int  mx_res_K = 0, mx_res_q = 0;    // shape of the expanded mixture operand R resident in mx_A (0 = none)
    bool hm_four_waves = false;         // tuning/testing: fused multi-vector pass with four waves per workgroup
    int  force_dense_rows = 0;          // tuning/testing: mixture rows always take the dense factorisation
    int pass_grid = 0;

    // profiling: event pairs are recorded without host synchronisation and summed in
    // lrvb_profile_get (so the timed region of bench.py is not perturbed)
    bool prof_on = false;
    std::vector<hipEvent_t> ev_pool[4];     // 0 = wsyrk, 1 = pass, 2 = build, 3 = sum-over-ranks hook
    size_t ev_used[4] = {0, 0, 0, 0};
    lrvb_prof prof{};
};

enum { PROF_WSYRK = 0, PROF_PASS = 1, PROF_BUILD = 2, PROF_REDUCE = 3, PROF_POOLS = 4 };
int prof_mark(lrvb_ctx* c, int which);      // records the next event of pool `which` on the ctx stream

int  buf_reserve(lrvb_ctx* c, DevBuf& b, size_t n);
int  reserve_obs_vec(lrvb_ctx* c, DevBuf& b);   // N doubles + 64 zeros of padding (LDS-DMA over-read)
void buf_free(DevBuf& b);

// ---- kernel launchers (each returns an lrvb status) ----------------------------------
// k_pack.hip
int upload_boxmap(lrvb_ctx* c);
int upload_jtmap(lrvb_ctx* c);            // row table of the structured J^T product (k_pack.hip); c->jt_rows = 0 where the layout has none
int launch_jt_apply(lrvb_ctx* c, const double* theta_dev, const double* A, i64 lda, i64 n, double* out, i64 ldo, bool trans_in);
int launch_constrain(lrvb_ctx* c, const double* theta_dev, double* eta_dev, double* j1_dev, double* j2_dev);
int launch_unconstrain(lrvb_ctx* c, const double* eta_dev, double* theta_dev, int* bad_flag_dev);
int launch_dense_jac(lrvb_ctx* c, const double* theta_dev, double* J_dev /* V x D */, i64 ld = 0, i64 rows_alloc = 0,
                     bool zeroed = false /* the caller has cleared J (launch_zero2) */);
int launch_zero2(lrvb_ctx* c, double* a, size_t na, double* b, size_t nb);       // two buffers cleared by one launch
int launch_third_order(lrvb_ctx* c, const double* theta_dev, const double* g_eta_dev, double* T_dev /* D x D */);

// k_glm.hip
enum PassMode { PASS_GRAD = 0, PASS_HVP = 1, PASS_HVP_C = 2 };
int launch_glm_pass(lrvb_ctx* c, PassMode mode, const double* beta_dev, const double* u_dev,
                    double* out_vec_P /* reduced */, double* value_out_dev /* nullable */,
                    bool store_obs);
bool hvp_multi_supported(const lrvb_ctx* c, i64 Q);
int  launch_hvp_multi(lrvb_ctx* c, i64 Q, const double* U_dev, i64 ldu, double* Out_dev, i64 ldo);
int  launch_rows_times_matrix(lrvb_ctx* c, i64 n0, i64 n1, i64 Q, const double* Zt_dev, i64 ldz,
                              const double* rowscale_dev, double* Tout_dev, i64 ldt);
int launch_obs_grad(lrvb_ctx* c, i64 n0, i64 n1, double* G_dev, int mode, const double* scale_vec);

// k_wsyrk.hip
int  wsyrk_num_tiles(i64 P);
int  wsyrk_auto_splits(const lrvb_ctx* c);
int  launch_wsyrk(lrvb_ctx* c, const double* cvec_dev, double* tiles_out_dev /* T*128*128 */);
bool wsyrk_fast_path(const lrvb_ctx* c);
int  launch_wsyrk_r(lrvb_ctx* c, const double* cvec_dev, double* tiles_out_dev, const double* cy_dev /* nullable */, double* r_out_dev /* P */);
int  launch_gram_small_on(lrvb_ctx* c, const double* Z, i64 N, i64 P, const double* cvec_dev, double* tiles_out_dev,
                          double* dense_out = nullptr, i64 ldd = 0, double* csum_out = nullptr, i64 pd = 0);
int  launch_mixture_rows(lrvb_ctx* c, int K, const double* theta_z_dev, const double* lam_dev,
                         double* Amat_dev, i64 lda, double* U_dev, double* gfree_dev, double* val2_dev, int* bad_dev);
int  launch_kron_rows(lrvb_ctx* c, double* Xk_dev, i64 ldk);
int  launch_mixture_expand(lrvb_ctx* c, const double* Rs, i64 lda, int q, int K, double* Rfull);
int  launch_atb(lrvb_ctx* c, const double* A, i64 PA, const double* B, i64 PB, i64 N,
                const double* cvec_dev, double* C_dev, bool rows_padded = false /* 16 finite rows past N in A and B */);
int  launch_atb_kron32(lrvb_ctx* c, const double* X31, const double* B, i64 N, const double* cvec_dev, double* C_dev);
int  launch_wsyrk_kron(lrvb_ctx* c, const double* cvec_dev, double* tiles_out_dev /* nb = ceil(q (q + 1) / 2 / 128) tile rows */);
int  launch_tiles_to_dense(lrvb_ctx* c, const double* tiles_dev, i64 P, double* dense_dev, i64 ld,
                           i64 row_off, i64 col_off, bool accumulate);

// k_lmm.hip
struct LmmIdx { int p, ms, ls, iem, iim, iay, iby, iam, ibm; i64 ld; };    // vector-coordinate positions of the global parameters
int  launch_lmm_closed_forms(lrvb_ctx* c, const LmmIdx& ix, const double* S, double* sums, const double* Md, const double* hp,
                             double* scratch, double* g, double* H, double* Gc, const double* part = nullptr, int n_part = 0);
int  launch_symkron3(lrvb_ctx* c, int k, const double* G, const double* P, double* H, i64 ld, i64 off);
struct MvnRegIdx { int k, ms, ls, ia, ib; i64 ld; };
int  launch_mvnreg_closed_forms(lrvb_ctx* c, const MvnRegIdx& ix, const double* S /* (k+1)^2 | W */, const double* hp, double* scratch,
                                double* g, double* H, double* Gc, double* value_out);
int  launch_add_padded(lrvb_ctx* c, i64 n, const double* src, i64 lds, double* dst, i64 ldd);
i64  grouped_rows_per_wave(i64 N);
bool grouped_fused_supported(const lrvb_ctx* c);
int  launch_grouped_stats_fused(lrvb_ctx* c, double* S_dense_dev /* q x q */, double* gs_dev /* G x (q + 1) */);

// k_linalg.hip
int launch_gemm(lrvb_ctx* c, bool transA, bool transB, i64 M, i64 Nn, i64 K, double alpha,
                const double* A, i64 lda, const double* B, i64 ldb, double beta, double* C, i64 ldc);
int launch_gemm_tn_small(lrvb_ctx* c, i64 K, i64 PA, i64 PB, const double* A, const double* B, double* C);
int launch_gemm_lower(lrvb_ctx* c, i64 M, i64 K, double alpha, const double* A, i64 lda,
                      double beta, double* C, i64 ldc);
int launch_potrf_lower(lrvb_ctx* c, double* A, i64 n, i64 lda, int* info_dev);
int launch_potrs_lower(lrvb_ctx* c, const double* L, i64 n, i64 ldl, double* B, i64 nrhs, i64 ldb);
int launch_trsm_lower_forward(lrvb_ctx* c, const double* L, i64 n, i64 ldl, double* B, i64 nrhs, i64 ldb);
int launch_dot(lrvb_ctx* c, const double* a, const double* b, i64 n, double* out_dev);
int launch_axpby(lrvb_ctx* c, i64 n, double alpha, const double* x, double beta, double* y);
int launch_dot3(lrvb_ctx* c, const double* a0, const double* b0, const double* a1, const double* b1,
                const double* a2, const double* b2, i64 n, double* out3_dev);
int launch_cg_update(lrvb_ctx* c, i64 n, double alpha, const double* d, const double* q, double* z, double* r, double* out2_dev);
int launch_gemv(lrvb_ctx* c, bool trans, i64 M, i64 Nn, double alpha, const double* A, i64 lda,
                const double* x, double beta, double* y);

// k_hyper.hip
int launch_hyper_cross(lrvb_ctx* c, int kind, i64 Ph, const double* r, const double* col, const double* j1, double* Cv /* V x Ph */);
int launch_hyper_grad(lrvb_ctx* c, int kind, i64 Ph, const double* eta, const double* r, const double* Ar, double* g /* Ph */);
int launch_hyper_col(lrvb_ctx* c, double a, const double* x, double bcoef, const double* y, double* out /* V */);

// k_cg.hip
int launch_symm_block(lrvb_ctx* c, i64 Q, i64 D, const double* U, const double* H /* symmetric, D x D */, double* W);

// k_finish.hip
int launch_quad_diff(lrvb_ctx* c, const double* eta_dev);    // vtmp = eta - m, vtmp2 = A (eta - m)
int launch_quad_grad_value(lrvb_ctx* c, const double* eta_dev, double* g_eta_dev /* += */, double* value_dev /* += */);
int launch_quad_hvp(lrvb_ctx* c, const double* u_vec_V, double* out_vec_V /* += scale*A u */);
int launch_finish_box(lrvb_ctx* c, const double* tiles_dev, const double* g_eta_dev,
                      const double* j1, const double* j2, bool with_third, double* H_dev, i64 ld);
int launch_build_Heta(lrvb_ctx* c, const double* tiles_dev, double* Heta_dev /* V x V */);
int launch_scatter_glm(lrvb_ctx* c, const double* g_glm_P, double* g_eta_V /* zero + scatter */);
