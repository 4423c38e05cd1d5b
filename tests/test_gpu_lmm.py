"""BASELINE.json config 4 on the GPU: hierarchical LMM (doc/lmm.lyx:77-160).  Grouped sufficient
statistics and the weighted Gram come from the device; oracle = exact AD (torch.func fp64) of the
restated -ELBO at small sizes, the dense arrow Hessian's inverse at a middle size, torch scatter
sums at N = 1.25e6 (one GPU's shard of the N = 1e7 configuration), p = 43, G = 1e4."""
import numpy as np
import pytest
import torch

import torch_ref as tr
from oracle import packing as opk
from helpers import rel_err, on_torch_stream
from test_lmm_host_math import make_par, random_eta, host_stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def _layout(p, G):
    return opk.Layout([opk.box_block(p), opk.psd_block(p), opk.box_block(1), opk.box_block(1, lb=0.0),
                       opk.box_block(1, lb=0.0), opk.box_block(1, lb=0.0), opk.box_block(1, lb=0.0), opk.box_block(1, lb=0.0),
                       opk.box_block(G), opk.box_block(G, lb=0.0)])


def _problem(vb, rng, N, p, G):
    x = rng.normal(size=(N, p))
    gid = rng.integers(0, G, size=N).astype(np.int32); gid[:G] = np.arange(G)
    u = rng.normal(size=G) * 0.7 + 0.3
    y = x @ rng.normal(size=p) + u[gid] + rng.normal(size=N) * 0.5
    par = make_par(p, G)
    priors = dict(beta_prior_mean=np.zeros(p), beta_prior_info=0.2 * np.eye(p), mu_prior_mean=0.1, mu_prior_info=0.3,
                  tau_y_prior=(2.0, 1.0), tau_mu_prior=(1.5, 0.5))
    fun = vb.LMMObjective(par, x, y, gid, G, **priors)
    ft = tr.lmm_objective(x, y, gid, G, priors['beta_prior_mean'], priors['beta_prior_info'], 0.1, 0.3, (2.0, 1.0), (1.5, 0.5),
                          layout=_layout(p, G))
    return x, y, gid, par, fun, ft


def _cavi_optimum(x, y, gid, G, beta0, lam0, mu0, kappa0, tau_y_prior, tau_mu_prior, sweeps=300, tol=1e-13):
    """The minimiser of the -ELBO by the model's closed-form coordinate updates (every factor is conjugate given the
    others: doc/lmm.lyx:105-160), returned in vector coordinates [m, tril(Lambda), e_mu, i_mu, a_y, b_y, a_mu, b_mu,
    e_1..e_G, i_1..i_G].  Used to place full-size checks AT the optimum, where the Hessian is positive definite."""
    N, p = x.shape
    a0y, b0y = tau_y_prior; a0m, b0m = tau_mu_prior
    XtX = x.T @ x
    Wg = np.bincount(gid, minlength=G).astype(np.float64)
    ty, tm = 1.0, 1.0
    eu, iu = np.zeros(G), np.ones(G)
    em, im = mu0, kappa0
    a_y = a0y + 0.5 * N; a_m = a0m + 0.5 * G
    for _ in range(sweeps):
        lam = lam0 + ty * XtX
        m = np.linalg.solve(lam, lam0 @ beta0 + ty * (x.T @ (y - eu[gid])))
        r = y - x @ m
        iu = tm + ty * Wg
        eu_new = (tm * em + ty * np.bincount(gid, weights=r, minlength=G)) / iu
        im = kappa0 + G * tm
        em = (kappa0 * mu0 + tm * np.sum(eu_new)) / im
        sig = np.linalg.inv(lam)
        b_y = b0y + 0.5 * (np.sum((r - eu_new[gid]) ** 2) + np.sum(XtX * sig) + np.sum(Wg / iu))
        b_m = b0m + 0.5 * np.sum((eu_new - em) ** 2 + 1.0 / iu + 1.0 / im)
        ty_new, tm_new = a_y / b_y, a_m / b_m
        done = abs(ty_new - ty) < tol * ty and abs(tm_new - tm) < tol * tm and np.max(np.abs(eu_new - eu)) < 1e-13
        ty, tm, eu = ty_new, tm_new, eu_new
        if done:
            break
    # one more consistent pass of the Gaussian factors at the final precisions
    lam = lam0 + ty * XtX
    m = np.linalg.solve(lam, lam0 @ beta0 + ty * (x.T @ (y - eu[gid])))
    iu = tm + ty * Wg
    return np.concatenate([m, lam[np.tril_indices(p)], [em, kappa0 + G * tm, a_y, b_y, a_m, b_m], eu, iu])


def test_cavi_helper_is_stationary_at_small_size(vb):
    rng = np.random.default_rng(9)
    N, p, G = 3000, 3, 12
    x, y, gid, par, fun, ft = _problem(vb, rng, N, p, G)
    eta = _cavi_optimum(x, y, gid, G, np.zeros(p), 0.2 * np.eye(p), 0.1, 0.3, (2.0, 1.0), (1.5, 0.5))
    theta = _layout(p, G).unconstrain(eta)
    g = torch.func.grad(ft)(torch.tensor(theta), torch.ones(N, dtype=torch.float64)).numpy()
    assert np.max(np.abs(g)) < 1e-8 * abs(ft(torch.tensor(theta), torch.ones(N, dtype=torch.float64)).item())
    assert np.linalg.norm(vb.Objective(par, fun).fun_free_grad(theta)) < 1e-6


def test_small_dense_parity(vb):
    rng = np.random.default_rng(3)
    N, p, G = 400, 3, 6
    x, y, gid, par, fun, ft = _problem(vb, rng, N, p, G)
    lay = _layout(p, G)
    assert par.free_size() == lay.D
    objective = vb.Objective(par, fun)
    w = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w)
    # device statistics against numpy
    assert rel_err(fun.local_stats(), host_stats(x, y, gid, G, w)) < 1e-12
    theta = lay.unconstrain(random_eta(rng, p, G))
    tt, tw = torch.tensor(theta), torch.tensor(w)
    H_ad = torch.func.hessian(ft)(tt, tw).numpy()
    assert abs(objective.fun_free(theta) - ft(tt, tw).item()) < 1e-10 * abs(ft(tt, tw).item())
    assert rel_err(objective.fun_free_grad(theta), torch.func.grad(ft)(tt, tw).numpy()) < 1e-9
    assert rel_err(objective.fun_free_hessian(theta), H_ad) < 1e-9
    ng = fun.n_global
    HS = fun.global_hessian(theta)
    assert rel_err(np.linalg.inv(HS), np.linalg.inv(H_ad)[:ng, :ng]) < 1e-7


def test_schur_complement_middle_size(vb):
    rng = np.random.default_rng(4)
    N, p, G = 30000, 6, 300
    x, y, gid, par, fun, ft = _problem(vb, rng, N, p, G)
    lay = _layout(p, G)
    theta = lay.unconstrain(random_eta(rng, p, G))
    H = vb.Objective(par, fun).fun_free_hessian(theta)          # dense 633 x 633 arrow matrix
    ng = fun.n_global
    assert ng == p + p * (p + 1) // 2 + 6
    HS = fun.global_hessian(theta)
    assert rel_err(np.linalg.inv(HS), np.linalg.inv(H)[:ng, :ng]) < 1e-7
    # sparse export of the same arrow matrix (SparseObjectives.get_sparse_sub_hessian placement) and its
    # JSON round trip; a sparse solve agrees with the dense one
    import scipy.sparse
    import scipy.sparse.linalg
    Hs = fun.sparse_hessian(theta)
    assert scipy.sparse.isspmatrix_csr(Hs) and Hs.shape == H.shape
    assert Hs.nnz <= ng * ng + 2 * ng * 2 * G + 2 * G            # dense global block, borders, diagonal
    assert rel_err(Hs.toarray(), H) < 1e-12
    packed = vb.SparseObjectives.json_pack_csr_matrix(Hs)
    import json
    back = vb.SparseObjectives.json_unpack_csr_matrix(json.loads(json.dumps(packed)))
    assert (back != Hs).nnz == 0
    b = rng.normal(size=H.shape[0])
    assert rel_err(scipy.sparse.linalg.spsolve(Hs.tocsc(), b), np.linalg.solve(H, b)) < 1e-8
    # a few AD Hessian-vector products pin the dense matrix itself
    w1 = torch.ones(N, dtype=torch.float64)
    grad_fn = torch.func.grad(ft)
    v = rng.normal(size=lay.D)
    hv = torch.func.jvp(lambda th: grad_fn(th, w1), (torch.tensor(theta),), (torch.tensor(v),))[1].numpy()
    assert rel_err(H @ v, hv) < 1e-8


def test_config4_shard_scale_statistics(vb):
    """One GPU's shard of config 4 (N = 1e7 over 8 GPUs -> 1.25e6 rows), p = 43, G = 1e4: the grouped
    sums and the Gram against torch on the same device; global D = 43 + 946 + 6 = 995."""
    N, p, G = 1250000, 43, 10000
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device=dev); gen.manual_seed(4)
    Z = torch.randn((N, p + 1), dtype=torch.float64, device=dev, generator=gen)
    gid = torch.randint(0, G, (N,), device=dev, generator=gen, dtype=torch.int32)
    w = torch.rand((N,), dtype=torch.float64, device=dev, generator=gen) + 0.5
    par = make_par(p, G)
    assert par.free_size() - 2 * G == 995
    ctx = on_torch_stream(vb.DeviceContext(par.layout_blocks(), loss='data_only', n_obs=N, n_cols=p + 1), dev)
    ctx.set_data_dev(0, Z.data_ptr(), N, p + 1)
    ctx.set_weights_dev(w.data_ptr(), N)
    ctx.set_groups(gid.cpu().numpy(), G)
    gs = ctx.group_sums()
    ref = torch.zeros((G, p + 2), dtype=torch.float64, device=dev)
    ref.index_add_(0, gid.long(), torch.cat([w[:, None], w[:, None] * Z], dim=1))
    assert rel_err(gs, ref.cpu().numpy()) < 1e-11
    S = ctx.weighted_gram()
    assert rel_err(S, ((Z.T * w) @ Z).cpu().numpy()) < 1e-11


def test_config4_full_pipeline_at_shard_scale(vb):
    """The whole config-4 pipeline at one GPU's shard size (N = 1.25e6, p = 43, G = 1e4; global D = 995, 2e4 local
    parameters): fit, device statistics, arrow-Hessian Schur complement, device Cholesky, LRVB covariance.  No oracle
    finishes at this size, so the assertions are the properties that hold at any size: the fit is stationary, the Schur
    complement of the arrow Hessian at the optimum is symmetric positive definite, and cov @ H_S = I."""
    rng = np.random.default_rng(44)
    N, p, G = 1_250_000, 43, 10_000
    x = rng.normal(size=(N, p))
    gid = rng.integers(0, G, size=N).astype(np.int32); gid[:G] = np.arange(G)
    u = rng.normal(size=G) * 0.7 + 0.3
    y = x @ rng.normal(size=p) + u[gid] + rng.normal(size=N) * 0.5
    par = make_par(p, G)
    fun = vb.LMMObjective(par, x, y, gid, G, beta_prior_mean=np.zeros(p), beta_prior_info=0.2 * np.eye(p), mu_prior_mean=0.1,
                          mu_prior_info=0.3, tau_y_prior=(2.0, 1.0), tau_mu_prior=(1.5, 0.5))
    objective = vb.Objective(par, fun)
    assert par.free_size() == 995 + 2 * G
    theta_opt = _layout(p, G).unconstrain(_cavi_optimum(x, y, gid, G, np.zeros(p), 0.2 * np.eye(p), 0.1, 0.3, (2.0, 1.0), (1.5, 0.5)))
    g = objective.fun_free_grad(theta_opt)
    assert np.max(np.abs(g)) < 1e-7 * abs(objective.fun_free(theta_opt)), np.max(np.abs(g))
    HS = fun.global_hessian(theta_opt)
    assert HS.shape == (995, 995)
    assert np.max(np.abs(HS - HS.T)) < 1e-9 * np.max(np.abs(HS))
    lam = np.linalg.eigvalsh(0.5 * (HS + HS.T))
    assert lam[0] > 0, lam[:3]
    fun.ctx.chol_factor(HS)
    cov = fun.ctx.lrvb_cov(np.eye(995))
    assert np.max(np.abs(cov @ HS - np.eye(995))) < 1e-7 * (lam[-1] / lam[0]) ** 0.5
    # the statistics are additive over a split of the rows (what the 8-GPU all-reduce relies on)
    half = N // 2
    fa = vb.LMMObjective(make_par(p, G), x[:half], y[:half], gid[:half], G)
    fb = vb.LMMObjective(make_par(p, G), x[half:], y[half:], gid[half:], G)
    assert rel_err(fa.local_stats() + fb.local_stats(), fun.local_stats()) < 1e-12


@pytest.mark.parametrize('N,p,G', [(400, 3, 6), (30000, 6, 300), (5000, 2, 1), (20000, 21, 777)])
def test_device_elimination_matches_host_assembly(vb, N, p, G):
    """`global_hessian` with the group sums resident on the device (lrvb_grouped_stats -> lrvb_lmm_group_terms ->
    lrvb_hvec_add_indexed / add_symkron -> finish) against the host assembly from the same statistics copied to numpy
    (`_global_hessian_host`, which the dense AD checks above pin); then the device-resident continuation -- factor
    the matrix where it lies, LRVB covariance -- against numpy on the host copy."""
    rng = np.random.default_rng(N + p + G)
    x, y, gid, par, fun, ft = _problem(vb, rng, N, p, G)
    w = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w)
    theta = _layout(p, G).unconstrain(random_eta(rng, p, G))
    assert fun._device_path
    H_dev = fun.global_hessian(theta)
    H_host = fun._global_hessian_host(theta)
    assert rel_err(H_dev, H_host) < 1e-11
    assert np.max(np.abs(H_dev - H_dev.T)) < 1e-12 * np.max(np.abs(H_dev))
    # the ONE-CALL route (closed forms on the device, round 4) against round 3's call-by-call route with numpy closed forms
    assert rel_err(H_dev, fun._global_hessian_device_stepwise(theta, True)) < 1e-12
    # the grouped statistics call against the two separate calls, and against numpy
    S, gs = fun.ctx.grouped_stats(want_S=True, want_gs=True)
    assert rel_err(S, fun.ctx.weighted_gram()) < 1e-13 and rel_err(gs, fun.ctx.group_sums()) < 1e-13   # other kernels, other summation order
    assert rel_err(np.concatenate([S.ravel(), gs.ravel()]), host_stats(x, y, gid, G, w)) < 1e-12
    # new weights invalidate the resident statistics
    w2 = rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w2)
    H2 = fun.global_hessian(theta)
    assert rel_err(H2, fun._global_hessian_host(theta)) < 1e-11 and rel_err(H2, H_dev) > 1e-6
    # device-resident continuation at an optimum (positive definite there)
    fun.weights_par.set_vector(np.ones(N))
    eta = _cavi_optimum(x, y, gid, G, np.zeros(p), 0.2 * np.eye(p), 0.1, 0.3, (2.0, 1.0), (1.5, 0.5))
    theta_opt = _layout(p, G).unconstrain(eta)
    fun.want_diagnostics = True                                # the sums of the elimination come back with the call
    assert fun.global_hessian(theta_opt, want_host=False) is None
    assert fun.last_local_grad_norm < 1e-6 * N                 # the kernel's own stationarity diagnostic
    fun._gctx.chol_factor_last()
    ng = fun.n_global
    M = rng.normal(size=(5, ng))
    HS = fun._global_hessian_host(theta_opt)
    assert rel_err(fun._gctx.lrvb_cov(M), M @ np.linalg.solve(HS, M.T)) < 1e-8


def test_scattered_block_of_the_device_assembly(vb):
    """lrvb_hvec_add_indexed: a dense block scattered over index lists, accumulated with the other block kinds."""
    rng = np.random.default_rng(8)
    par = vb.ModelParamsDict('p'); par.push_param(vb.VectorParam('a', 9)); par.push_param(vb.VectorParam('b', 5, lb=0.0))
    ctx = vb.DeviceContext(par.layout_blocks(), quad_kind=1)
    rows, cols = np.array([0, 3, 13, 7]), np.array([2, 12, 5])
    B = rng.normal(size=(4, 3))
    base = rng.normal(size=(14, 14))
    ctx.hvec_begin()
    ctx.hvec_add_block(base, 0, 0)
    ctx.hvec_add_indexed(B, rows, cols)
    got = ctx.hvec_finish(np.zeros(14), np.zeros(14), is_free=False)
    want = base.copy(); want[np.ix_(rows, cols)] += B
    assert np.array_equal(got, want)
    ctx.hvec_begin()
    with pytest.raises(ValueError):
        ctx.hvec_add_indexed(B, rows, np.array([2, 12, 14]))        # column outside the matrix
    with pytest.raises(ValueError):
        ctx.hvec_add_indexed(B, rows[:3], cols)                      # block shape


@pytest.mark.parametrize('q', [2, 8, 16, 18, 30, 32, 34, 44, 48, 50, 64, 45, 7])
def test_fused_grouped_statistics_layouts_and_ragged_groups(vb, q):
    """lrvb_grouped_stats, the one-pass kernel over group-sorted rows (even q; odd q takes the two-kernel route): every
    column layout (pair groups of 32 columns + a trailing block of <= 16), groups that are empty, of 1..5 rows, cut by
    one wave boundary, and one group that spans many waves; unsorted group ids; bitwise reproducible."""
    rng = np.random.default_rng(q)
    sizes = np.concatenate([[0, 1, 2, 3, 4, 5, 0, 0, 7, 63, 64, 65, 1000, 129], rng.integers(0, 40, size=200), [0]])
    G = sizes.size
    gid = np.repeat(np.arange(G), sizes).astype(np.int32)
    rng.shuffle(gid)
    N = gid.size
    assert N > 20 * 64                                  # several waves of 64 rows; the 1000-row group spans >= 15 of them
    Z = rng.normal(size=(N, q))
    w = rng.uniform(0.5, 1.5, N)
    blocks = [dict(kind=0, free_size=1, vec_size=1, dim0=1, dim1=0, lb=-np.inf, ub=np.inf)]
    ctx = vb.DeviceContext(blocks, loss='data_only', n_obs=N, n_cols=q)
    ctx.set_groups(gid, G)                               # groups before data: the sorted copy is built lazily
    ctx.set_data(vb._hip.SLOT_X, Z)
    ctx.set_weights(w)
    S, gs = ctx.grouped_stats(want_S=True, want_gs=True)
    want_gs = np.zeros((G, q + 1))
    np.add.at(want_gs[:, 0], gid, w)
    np.add.at(want_gs[:, 1:], gid, w[:, None] * Z)
    assert rel_err(S, Z.T @ (w[:, None] * Z)) < 1e-13
    assert np.max(np.abs(gs - want_gs)) < 1e-12 * np.max(np.abs(want_gs))
    assert np.all(gs[sizes == 0] == 0.0)
    S2, gs2 = ctx.grouped_stats(want_S=True, want_gs=True)
    assert np.array_equal(S, S2) and np.array_equal(gs, gs2)
    # new data in the same context: the sorted copy is rebuilt
    Z2 = rng.normal(size=(N, q))
    ctx.set_data(vb._hip.SLOT_X, Z2)
    S3, gs3 = ctx.grouped_stats(want_S=True, want_gs=True)
    assert rel_err(S3, Z2.T @ (w[:, None] * Z2)) < 1e-13
    # the two-kernel route on the same rows (tuning bit 0)
    ctx.set_tuning(0, 1)
    S4, gs4 = ctx.grouped_stats(want_S=True, want_gs=True)
    ctx.set_tuning(0, 0)
    assert rel_err(S4, S3) < 1e-13 and rel_err(gs4, gs3) < 1e-13


@pytest.mark.parametrize('N,G', [(1, 1), (2, 3), (5, 2), (63, 70), (64, 1), (65, 4), (129, 129)])
def test_fused_grouped_statistics_tiny_inputs(vb, N, G):
    """Fewer rows than one k-step, fewer rows than groups, one group, one row per group."""
    rng = np.random.default_rng(N * 131 + G)
    q = 6
    gid = rng.integers(0, G, size=N).astype(np.int32)
    Z = rng.normal(size=(N, q)); w = rng.uniform(0.5, 1.5, N)
    ctx = vb.DeviceContext([dict(kind=0, free_size=1, vec_size=1, dim0=1, dim1=0, lb=-np.inf, ub=np.inf)], loss='data_only', n_obs=N, n_cols=q)
    ctx.set_data(vb._hip.SLOT_X, Z); ctx.set_groups(gid, G); ctx.set_weights(w)
    S, gs = ctx.grouped_stats(want_S=True, want_gs=True)
    want = np.zeros((G, q + 1))
    np.add.at(want[:, 0], gid, w); np.add.at(want[:, 1:], gid, w[:, None] * Z)
    assert np.max(np.abs(S - Z.T @ (w[:, None] * Z))) < 1e-12 * max(1.0, np.max(np.abs(S)))
    assert np.max(np.abs(gs - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))
