"""BASELINE.json config 3 on the GPU: Dirichlet-multinomial mixture with a SimplexParam row per
observation (LRVB/SimplexParams.py:69-175).  The device eliminates every row's (K-1) x (K-1) simplex
block (one wavefront per row) and reduces the Schur-complement operand with an MFMA GEMM; oracle = the
per-row numpy restatement (oracle/mixture.py) and exact AD of the torch restatement at small sizes,
size-independent identities at N = 1e6, K = 32, V = 31."""
import time

import numpy as np
import pytest
import torch

import torch_ref as tr
from oracle import mixture as om
from helpers import rel_err
from test_mixture_host_math import make_par, problem, near_optimum_problem, clustered_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


@pytest.mark.parametrize('N,V,K', [(1, 2, 2), (37, 3, 2), (50, 4, 3), (61, 5, 4), (40, 7, 5), (33, 6, 6), (45, 9, 8),
                                   (38, 8, 11), (130, 12, 16), (64, 20, 23), (150, 31, 32),
                                   (9, 31, 2), (21, 1, 32), (2, 1, 2), (131, 17, 17)])      # wide x with two categories, one column, two rows, K just past one MFMA block
def test_rows_match_oracle(vb, N, V, K):
    x, w, theta = problem(N, V, K, seed=100 + K)
    par = make_par(N, V, K)
    fun = vb.MixtureObjective(par, x, pi_prior=1.5, phi_prior=0.8)
    fun.weights_par.set_vector(w)
    fg, fz = theta[:fun.n_global], theta[fun.n_global:]
    _, _, lam = fun._lam(np.exp(fg))
    fun._push_state()
    val2, gz, S64, R = fun.ctx.mixture_rows(K, fz, lam)
    o_val2, o_g, _, o_S64, o_R = om.mixture_rows(fz, x, w, lam)
    assert rel_err(val2, o_val2) < 1e-12
    assert rel_err(gz, o_g) < 1e-11
    assert rel_err(S64, o_S64) < 1e-12
    assert rel_err(R, o_R) < 1e-9                      # includes a (K-1) x (K-1) solve per row
    # the dense per-row factorisation (taken when a row is far from its optimum) gives the same operand
    fun.ctx.set_tuning(0, 2)
    R_dense = fun.ctx.mixture_rows(K, fz, lam)[3]
    fun.ctx.set_tuning(0, 0)
    assert rel_err(R_dense, o_R) < 1e-9 and rel_err(R_dense, R) < 1e-9
    # the flat statistics buffer is the same thing
    stats = fun.local_stats(theta)
    assert rel_err(stats, np.concatenate([o_val2, o_S64.ravel(), o_R.ravel()])) < 1e-9


@pytest.mark.parametrize('N,V,K', [(150, 3, 2), (200, 4, 3), (240, 5, 4)])
def test_schur_complement_matches_ad(vb, N, V, K):
    x, w, theta = near_optimum_problem(N, V, K, seed=7 * K)
    par = make_par(N, V, K)
    fun = vb.MixtureObjective(par, x, pi_prior=1.5, phi_prior=0.8, weights=w)
    ft = tr.mixture_objective(x, K, 1.5, 0.8)
    tt, tw = torch.tensor(theta), torch.tensor(w)
    ng = fun.n_global
    v_ad = ft(tt, tw).item()
    assert abs(fun.value(theta) - v_ad) < 1e-11 * abs(v_ad)
    # torch's fp64 trigamma is accurate to ~1e-9, which bounds the agreement of the global block
    assert rel_err(fun.grad(theta), torch.func.grad(ft)(tt, tw).numpy()) < 1e-8
    H_ad = torch.func.hessian(ft)(tt, tw).numpy()
    HS_ad = H_ad[:ng, :ng] - H_ad[:ng, ng:] @ np.linalg.solve(H_ad[ng:, ng:], H_ad[ng:, :ng])
    HS = fun.global_hessian(theta)
    assert rel_err(HS, HS_ad) < 1e-7
    assert np.all(np.linalg.eigvalsh(H_ad) > 0)          # a point where LRVB applies
    cov = fun.global_cov(theta)
    assert rel_err(cov, np.linalg.inv(HS_ad)) < 1e-6
    assert rel_err(cov, np.linalg.inv(H_ad)[:ng, :ng]) < 1e-5     # the full matrix has condition ~1e7
    # moments through a Jacobian: E[pi] sensitivity rows
    M = np.random.default_rng(0).normal(size=(3, ng))
    assert rel_err(fun.global_cov(theta, M), M @ np.linalg.inv(HS_ad) @ M.T) < 1e-6


@pytest.mark.parametrize('K,q', [(2, 3), (3, 5), (5, 4), (8, 16), (6, 32), (32, 32)])      # (6, 32): 192 = 128 + 64 columns, an edge tile half full
def test_device_schur_assembly_matches_oracle(vb, K, q):
    """lrvb_mixture_schur against oracle.mixture_schur: odd sizes take the generic GEMM, (8, 16) and (32, 32)
    the LDS-DMA MFMA kernel; R from the host and R left on the device by lrvb_mixture_rows."""
    rng = np.random.default_rng(K * 100 + q)
    n = K * q
    B = rng.normal(size=(q * q, 7)); C = rng.normal(size=(K * K, 7))
    R = B @ C.T
    R4 = R.reshape(q, q, K, K)
    R = (R4 + R4.transpose(1, 0, 2, 3) + R4.transpose(0, 1, 3, 2) + R4.transpose(1, 0, 3, 2)).reshape(q * q, K * K)
    J = rng.normal(size=(n, n)) * (rng.random((n, n)) < 0.3)
    Hgg = rng.normal(size=(n, n)); Hgg = Hgg + Hgg.T
    sc, dg = rng.uniform(0.5, 2.0, n), rng.normal(size=n)
    ctx = vb.DeviceContext([dict(kind=0, free_size=n, vec_size=n, dim0=n, dim1=0, lb=-np.inf, ub=np.inf)], quad_kind=1)
    for scale, diag in ((sc, dg), (None, None), (sc, None)):
        got = ctx.mixture_schur(K, q, R, J, Hgg, scale=scale, diag_add=diag)
        want = om.mixture_schur(K, q, R, J, Hgg, scale, diag)
        assert rel_err(got, want) < 1e-13
        assert np.array_equal(got, got.T)                  # symmetrised on the device, exactly
    # R uploaded once stays resident: NULL reuses it
    assert np.array_equal(ctx.mixture_schur(K, q, None, J, Hgg, scale=sc, diag_add=dg),
                          ctx.mixture_schur(K, q, R, J, Hgg, scale=sc, diag_add=dg))
    # a fresh context has nothing resident, and a resident operand of another shape is refused
    ctx2 = vb.DeviceContext([dict(kind=0, free_size=n, vec_size=n, dim0=n, dim1=0, lb=-np.inf, ub=np.inf)], quad_kind=1)
    with pytest.raises(RuntimeError):
        ctx2.mixture_schur(K, q, None, J, Hgg)
    with pytest.raises(RuntimeError):
        ctx.mixture_schur(q, K, None, J, Hgg) if K != q else ctx2.mixture_schur(q, K, None, J, Hgg)
    with pytest.raises(ValueError):
        ctx.mixture_schur(K, q, R, J[:-1], Hgg)


def test_saturated_responsibilities(vb):
    """Rows whose responsibilities sit at a vertex (p ~ 1e-12 here; the unscaled H_nn of the reference
    is then singular to rounding).  The oracle follows the reference's unscaled algebra, so agreement
    is limited by ITS conditioning; p -> 0 exactly must stay finite."""
    N, V, K = 60, 6, 8
    x, w, fg, fz, lam = clustered_problem(N, V, K, seed=2, trials=6)
    fz = fz + 0.05 * np.random.default_rng(0).normal(size=fz.shape)   # the gradient vanishes at the optimum
    fun = vb.MixtureObjective(make_par(N, V, K), x, weights=w)
    fun._push_state()
    val2, gz, S64, R = fun.ctx.mixture_rows(K, fz.ravel(), lam)
    o = om.mixture_rows(fz.ravel(), x, w, lam)
    assert rel_err(val2, o[0]) < 1e-12 and rel_err(gz, o[1]) < 1e-10 and rel_err(S64, o[3]) < 1e-12
    assert rel_err(R, o[4]) < 1e-6
    fz2 = fz.copy()
    fz2[::3] *= 200.0                                   # logits of +-1e3: exp underflows to exactly 0
    val2, gz, S64, R = fun.ctx.mixture_rows(K, fz2.ravel(), lam, want_schur=False)
    assert np.all(np.isfinite(val2)) and np.all(np.isfinite(gz)) and np.all(np.isfinite(S64))


def test_indefinite_local_block_is_reported(vb):
    N, V, K = 20, 3, 3
    x, w, theta = problem(N, V, K, seed=5)
    par = make_par(N, V, K)
    fun = vb.MixtureObjective(par, x, weights=w)
    bad = theta.copy()
    bad[fun.n_global:] = -bad[fun.n_global:] * 3.0 + 4.0     # far from the optimal responsibilities
    o = om.mixture_rows(bad[fun.n_global:], x, w, fun._lam(np.exp(bad[:fun.n_global]))[2])
    assert not np.all([np.all(np.linalg.eigvalsh(h) > 0) for h in o[2]])       # the seeded point does have an indefinite block
    with pytest.raises(np.linalg.LinAlgError):
        fun.global_hessian(bad)
    # value and gradient do not need the factorisation
    ft = tr.mixture_objective(x, K, 1.0, 1.0)
    assert abs(fun.value(bad) - ft(torch.tensor(bad), torch.tensor(w)).item()) < 1e-10 * abs(fun.value(bad))


def test_unsupported_shapes_fail_loudly(vb):
    N, V, K = 10, 3, 33                                 # one wavefront holds at most 32 categories (and V + 1 <= 32)
    rng = np.random.default_rng(1)
    x = rng.poisson(2.0, size=(N, V)).astype(np.float64)
    fun = vb.MixtureObjective(make_par(N, V, K), x)
    theta = np.concatenate([np.ones(K + V * K), rng.normal(size=N * (K - 1))])
    with pytest.raises(NotImplementedError):
        fun.value(theta)


def test_full_size_properties(vb):
    """N = 1e6, K = 32, V = 31 (D_global = 1024): identities that hold at any size."""
    N, V, K = 1_000_000, 31, 32
    x, w, fg, fz, lam = clustered_problem(N, V, K, seed=11)
    theta = np.concatenate([fg, fz.ravel()])
    par = vb.ModelParamsDict('params')
    par.push_param(vb.DirichletParamArray('pi', shape=(K,)))
    par.push_param(vb.DirichletParamArray('phi', shape=(V, K)))
    par.push_param(vb.SimplexParam('z', shape=(N, K)))
    fun = vb.MixtureObjective(par, x, pi_prior=1.2, phi_prior=0.9, weights=w)
    t0 = time.perf_counter()
    stats = fun.local_stats(theta)
    t1 = time.perf_counter()
    val2, S64, R = fun._unpack_stats(stats)
    q = V + 1
    # (1) sufficient statistics: sum_k z_nk = 1  ->  C 1 = sum_n w_n x~_n, and the x~ block is X~^T W X~
    xt = np.hstack([np.ones((N, 1)), x])
    assert rel_err(S64[:q, 32:32 + K].sum(axis=1), xt.T @ w) < 1e-11
    assert rel_err(S64[:q, :q], xt.T @ (w[:, None] * xt)) < 1e-12
    # (2) J^T 1 = 0  ->  A_n 1 = 0: the Schur operand annihilates the constant direction of each row
    R4 = R.reshape(q, q, K, K)
    assert np.max(np.abs(R4.sum(axis=3))) < 1e-9 * np.max(np.abs(R4))
    assert rel_err(R4, R4.transpose(1, 0, 3, 2)) < 1e-12          # symmetry in both index pairs
    # (3) additivity over shards of the observation axis (the multi-GPU reduction), on a subsample
    n1 = 30_011
    par1 = make_par(n1, V, K)
    f1 = vb.MixtureObjective(par1, x[:n1], pi_prior=1.2, phi_prior=0.9, weights=w[:n1])
    th1 = np.concatenate([fg, fz[:n1].ravel()])
    par2 = make_par(N - n1, V, K)
    f2 = vb.MixtureObjective(par2, x[n1:], pi_prior=1.2, phi_prior=0.9, weights=w[n1:])
    th2 = np.concatenate([fg, fz[n1:].ravel()])
    assert rel_err(f1.local_stats(th1) + f2.local_stats(th2), stats) < 1e-11
    # (4) the subsample against the oracle rows, row for row
    n2 = 64
    o = om.mixture_rows(fz[:n2], x[:n2], w[:n2], lam)
    par3 = make_par(n2, V, K)
    f3 = vb.MixtureObjective(par3, x[:n2], weights=w[:n2])
    f3._push_state()
    v3, g3, S3, R3 = f3.ctx.mixture_rows(K, fz[:n2].ravel(), lam)
    assert rel_err(R3, o[4]) < 1e-9
    assert np.max(np.abs(g3 - o[1])) < 1e-13             # the local gradient vanishes at the z-optimum
    # (5) Schur complement is symmetric positive definite near the optimum and its Cholesky solve inverts it
    HS = fun.global_hessian(theta)
    assert rel_err(HS, HS.T) < 1e-12
    cov = fun.global_cov(theta)
    assert rel_err(cov @ HS, np.eye(fun.n_global)) < 1e-7
    print('\n[mixture N=1e6 K=32 V=31] local_stats {:.1f} ms (host buffers in and out)'.format(1e3 * (t1 - t0)))


@pytest.mark.parametrize('N,V,K', [(150, 3, 2), (200, 4, 3), (240, 5, 4), (300, 31, 32), (260, 7, 16)])
def test_device_generated_dirichlet_blocks_match_the_uploaded_matrices(vb, N, V, K):
    """`global_hessian` (statistics with the Schur operand left on the device + lrvb_mixture_schur_dirichlet, which writes
    d Lam / d alpha and the global Hessian block from their diagonals and per-Dirichlet constants) against the route
    that builds both n x n matrices with numpy and uploads them (`return_parts=True`), and the resident result against
    its host copy through the device Cholesky."""
    x, w, theta = near_optimum_problem(N, V, K, seed=7 * K)       # the seeds of test_schur_complement_matches_ad: positive definite there
    par = make_par(N, V, K)
    fun = vb.MixtureObjective(par, x, pi_prior=1.5, phi_prior=0.8, weights=w)      # the priors near_optimum_problem fitted with
    assert fun._canonical_order()
    HS_dev = fun.global_hessian(theta)
    HS_up = fun.global_hessian(theta, return_parts=True)[0]
    assert rel_err(HS_dev, HS_up) < 1e-12
    assert np.array_equal(HS_dev, HS_dev.T)
    assert fun.global_hessian(theta, want_host=False) is None
    # the resident matrix is the one that came back: the device Cholesky agrees with numpy's verdict on it either way
    if np.linalg.eigvalsh(HS_up).min() > 1e-9 * np.abs(HS_up).max():
        fun.ctx.chol_factor_last()
        M = np.random.default_rng(1).normal(size=(4, fun.n_global))
        assert rel_err(fun.ctx.lrvb_cov(M), M @ np.linalg.solve(HS_up, M.T)) < 1e-7
    else:
        assert np.linalg.eigvalsh(HS_up).min() < 0
        with pytest.raises(np.linalg.LinAlgError):
            fun.ctx.chol_factor_last()
