"""GPU-vs-oracle parity AT THE BASELINE.json SIZES (round-3 verdict: at N >= 1e5 parity rested on size-independent
properties only).  Each test streams the benchmark-size inputs through the oracle's arithmetic in chunks on the host and
compares the HIP path's result with it directly:

  * headline  N = 1e6 x D = 1024: Hessian, gradient and value against `DeclaredModel.hessian_vec / grad_vec / value_vec`
    accumulated over 65,536-row chunks and `convert_vector_to_free_hessian` (LRVB/Parameters.py:397-424);
  * config 4  sufficient statistics of the 1.25e6-row shard, G = 1e4 (doc/lmm.lyx:105-160) against numpy;
  * config 5  G^T G at N = 1e5, D = 4096 against the per-observation gradient matrix written out in closed form (itself
    pinned against the exact AD cross Hessian d2 f / d theta d w^T at a small size inside the test);
  * config 3  the per-row simplex elimination and the Schur operand on 12,000 rows, K = 32, against the per-row oracle.

Tolerances: 1e-11 relative to the largest entry for sums over observations, 1e-9 through the per-row solves."""
import numpy as np
import pytest
import torch

import torch_ref as tr
from oracle import packing as opk
from oracle import models as om
from helpers import rel_err, on_torch_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def test_headline_hessian_gradient_value_at_1e6_x_1024(vb):
    N, D, n_pos = 1_000_000, 1024, 256
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device=dev); gen.manual_seed(4242)
    X = torch.randn((N, D), dtype=torch.float64, device=dev, generator=gen) / D ** 0.5
    y = torch.randn((N,), dtype=torch.float64, device=dev, generator=gen)
    w = torch.rand((N,), dtype=torch.float64, device=dev, generator=gen) + 0.5
    theta = 0.1 * torch.randn((D,), dtype=torch.float64, device=dev, generator=gen)
    torch.cuda.synchronize()
    prior = np.linspace(0.5, 1.5, D)
    blocks = [dict(kind=0, free_size=D - n_pos, vec_size=D - n_pos, dim0=D - n_pos, dim1=0, lb=-np.inf, ub=np.inf),
              dict(kind=0, free_size=n_pos, vec_size=n_pos, dim0=n_pos, dim1=0, lb=0.0, ub=np.inf)]
    ctx = vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=D, lik_info=2.0, quad_kind=vb._hip.QUAD_DIAG, device=0)
    ctx.set_data_dev(vb._hip.SLOT_X, X.data_ptr(), N, D)
    ctx.set_data_dev(vb._hip.SLOT_Y, y.data_ptr(), N, 1)
    ctx.set_weights_dev(w.data_ptr(), N)
    ctx.set_data(vb._hip.SLOT_QUAD_A, prior)
    H = torch.empty((D, D), dtype=torch.float64, device=dev)
    ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), D)
    ctx.sync()
    th = theta.cpu().numpy()
    H_gpu = H.cpu().numpy()
    g_gpu = ctx.grad(th)
    v_gpu = ctx.value(th)
    # the oracle, chunk by chunk: the data term is a sum over observations (vector coordinates), the prior and the packing
    # conversion are applied once
    lay = opk.Layout([opk.box_block(D - n_pos), opk.box_block(n_pos, lb=0.0)])
    eta = lay.constrain(th)
    S, g, val = np.zeros((D, D)), np.zeros(D), 0.0
    chunk = 65536
    for a in range(0, N, chunk):
        b = min(a + chunk, N)
        mc = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=X[a:b].cpu().numpy(), y=y[a:b].cpu().numpy(), w=w[a:b].cpu().numpy(), lik_info=2.0)
        S += mc.hessian_vec(eta)
        g += mc.grad_vec(eta)
        val += mc.value_vec(eta)
    m0 = om.DeclaredModel(lay, loss=0, quad_A=prior)
    g += m0.grad_vec(eta); val += m0.value_vec(eta)
    H_or = opk.convert_vector_to_free_hessian(lay, th, g, S + m0.hessian_vec(eta))
    g_or = lay.jac(th).T @ g
    assert rel_err(H_gpu, H_or) < 1e-11
    assert rel_err(g_gpu, g_or) < 1e-11
    assert abs(v_gpu - val) < 1e-12 * abs(val)
    assert np.array_equal(H_gpu, H_gpu.T)


def test_config4_statistics_at_shard_size(vb):
    """The one-pass grouped statistics of the hierarchical model at N = 1.25e6, q = 44, G = 1e4 against numpy, with
    non-trivial weights (the size-independent checks of test_gpu_lmm.py stay; this is the direct comparison)."""
    import scipy.sparse
    from test_lmm_host_math import make_par
    rng = np.random.default_rng(404)
    N, p, G = 1_250_000, 43, 10_000
    x = rng.normal(size=(N, p))
    gid = rng.integers(0, G, size=N).astype(np.int32); gid[:G] = np.arange(G)
    y = x @ rng.normal(size=p) + rng.normal(size=G)[gid] * 0.7 + rng.normal(size=N) * 0.5
    w = rng.uniform(0.5, 1.5, N)
    fun = vb.LMMObjective(make_par(p, G), x, y, gid, G, weights=w)
    got = fun.local_stats()
    z = np.hstack([x, y[:, None]])
    q = p + 1
    sel = scipy.sparse.csr_matrix((w, (gid, np.arange(N))), shape=(G, N))          # group-membership matrix, weighted
    want = np.concatenate([(z.T @ (w[:, None] * z)).ravel(), np.hstack([np.asarray(sel.sum(axis=1)), sel @ z]).ravel()])
    assert rel_err(got[:q * q], want[:q * q]) < 1e-12                     # S = Z^T diag(w) Z
    assert rel_err(got[q * q:], want[q * q:]) < 1e-12                     # per-group sum w, sum w x, sum w y
    assert np.max(np.abs(got[q * q:] - want[q * q:]) / (1.0 + np.abs(want[q * q:]))) < 1e-12      # every group, not only the largest


def _wishart_obs_grad_vec(y, eta, d):
    """Row n of G in VECTOR coordinates, written out from the per-observation term of the model
        l_n = 1/2 nu ((y_n - m)^T V (y_n - m) + tr(V Sigma_mu)) - 1/2 E log|Lambda|
    (LRVB/NormalParams.py:6-23, WishartParams.py:6-35) -- an independent restatement for this test; eta = [m, vech(Lambda_mu),
    nu, vech(V)] with off-diagonal vech entries standing for both symmetric matrix entries."""
    from scipy import special
    mm = d * (d + 1) // 2
    r_, c_ = np.tril_indices(d)
    fac = np.where(r_ == c_, 1.0, 2.0)

    def sym(v):
        L = np.zeros((d, d)); L[r_, c_] = v
        return L + L.T - np.diag(np.diag(L))
    m, lam_mu, nu, v = eta[:d], sym(eta[d:d + mm]), eta[d + mm], sym(eta[d + mm + 1:])
    sig = np.linalg.inv(lam_mu)
    vinv = np.linalg.inv(v)
    r = y - m[None, :]
    rv = r @ v
    N = y.shape[0]
    G = np.empty((N, d + 2 * mm + 1))
    G[:, :d] = -nu * rv
    svs = sig @ v @ sig
    G[:, d:d + mm] = (-0.5 * nu * svs[r_, c_] * fac)[None, :]
    kap1 = 0.5 * np.sum(special.polygamma(1, 0.5 * nu - 0.5 * np.arange(d)))
    G[:, d + mm] = 0.5 * (np.sum(rv * r, axis=1) + np.sum(v * sig)) - 0.5 * kap1
    G[:, d + mm + 1:] = (0.5 * nu * (r[:, r_] * r[:, c_] + sig[r_, c_][None, :]) - 0.5 * vinv[r_, c_][None, :]) * fac[None, :]
    return G


def test_config5_gram_at_1e5_x_4096(vb):
    from test_gpu_wishart import _build
    from test_wishart_mvn_host_math import random_point
    # (i) the closed-form rows of G are the exact AD cross Hessian d2 f / d theta d w^T (small size)
    d, N = 3, 40
    rng = np.random.default_rng(55)
    y, par, fun, lay, ft = _build(vb, rng, N, d)
    theta = lay.unconstrain(random_point(rng, d))
    cross = torch.func.jacfwd(torch.func.grad(ft, argnums=0), argnums=1)(torch.tensor(theta), torch.ones(N, dtype=torch.float64)).numpy()
    Gs = _wishart_obs_grad_vec(y, lay.constrain(theta), d) @ lay.jac(theta)
    assert rel_err(Gs.T, cross) < 1e-10
    assert rel_err(fun.gram(theta), Gs.T @ Gs) < 1e-10
    # (ii) at N = 1e5, d = 63 -> D = 4096: G^T G of the device against the same rows contracted on the host
    d, N = 63, 100_000
    rng = np.random.default_rng(20245)
    y, par, fun, lay, ft = _build(vb, rng, N, d)
    assert par.free_size() == 4096
    eta0 = random_point(rng, d)
    eta0[d + d * (d + 1) // 2] = d + 10.0
    theta = lay.unconstrain(eta0)
    got = fun.gram(theta)
    eta = lay.constrain(theta)
    A = np.zeros((eta.size, eta.size))
    for a in range(0, N, 20000):                                         # 20,000 x 4096 rows of G at a time
        Gv = _wishart_obs_grad_vec(y[a:a + 20000], eta, d)
        A += Gv.T @ Gv
    J = lay.jac(theta)
    want = J.T @ A @ J
    assert rel_err(got, want) < 1e-11
    assert np.allclose(got, got.T, rtol=0, atol=1e-12 * np.abs(got).max())


def test_config3_rows_and_schur_operand_on_12000_rows(vb):
    """K = 32, V = 31 (the configuration's shape), 12,000 rows: value partials, the free local gradient, the statistics
    S64 and the Schur operand R of the device against the per-row oracle (SimplexParams.py:33-63 composed through
    convert_vector_to_free_hessian, one (K - 1) x (K - 1) solve per row), then the assembled 1024 x 1024 Schur complement
    against the oracle's assembly from the oracle's operand."""
    from test_mixture_host_math import make_par, near_optimum_problem
    from oracle import mixture as omx
    N, V, K = 12_000, 31, 32
    x, w, theta = near_optimum_problem(N, V, K, seed=132, sweeps=30)      # responsibilities near their row optimum: every local block positive definite
    par = make_par(N, V, K)
    fun = vb.MixtureObjective(par, x, pi_prior=1.5, phi_prior=0.8, weights=w)
    ng = fun.n_global
    assert ng == 1024
    fg, fz = theta[:ng], theta[ng:]
    alpha, beta, lam = fun._lam(np.exp(fg))
    fun._push_state()
    val2, gz, S64, R = fun.ctx.mixture_rows(K, fz, lam)
    o_val2, o_g, _, o_S64, o_R = omx.mixture_rows(fz, x, w, lam)
    assert rel_err(val2, o_val2) < 1e-11
    assert rel_err(gz, o_g) < 1e-11
    assert rel_err(S64, o_S64) < 1e-11
    assert rel_err(R, o_R) < 1e-9
    # the Schur complement assembled on the device from the device operand vs the oracle's assembly of the oracle's operand
    HS = fun.global_hessian(theta)
    C = o_S64[:V + 1, 32:32 + K]
    _, g_vec, Hgg = fun._global_terms(alpha, beta, C)
    jg = np.exp(fg)
    want = omx.mixture_schur(K, V + 1, o_R, fun._dlam(alpha, beta) * jg[None, :], Hgg, scale=jg, diag_add=g_vec * jg)
    assert rel_err(HS, want) < 1e-9
