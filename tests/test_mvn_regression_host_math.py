"""Config 2: the closed forms of MVNRegressionObjective (product host math) against exact AD of a
torch restatement assembled from the reference's building blocks (NormalParams.MVNParam moments,
ExponentialFamilies priors / entropies)."""
import numpy as np
import pytest
import torch

import lrvb_amd as vb
import torch_ref as tr
from lrvb_amd.quadform import duplication_matrix


def _shell(k, mu0, lam0, a0, b0):
    from scipy import special
    f = vb.MVNRegressionObjective.__new__(vb.MVNRegressionObjective)
    mm = k * (k + 1) // 2
    f._special, f.k, f.q = special, k, k + 1
    f._ms, f._ls, f._ia, f._ib = range(0, k), range(k, k + mm), k + mm, k + mm + 1
    f._declare_priors(mu0, lam0, a0, b0)
    f._dup = duplication_matrix(k)
    return f


@pytest.mark.parametrize('k,N', [(1, 30), (3, 50), (5, 80)])
def test_closed_forms_match_ad(k, N):
    rng = np.random.default_rng(k)
    x = rng.normal(size=(N, k)); y = rng.normal(size=N); w = rng.uniform(0.5, 1.5, N)
    mu0 = rng.normal(size=k); a = rng.normal(size=(k, k)); lam0 = a @ a.T / k + np.eye(k)
    a0, b0 = 2.5, 1.3
    mm = k * (k + 1) // 2
    m = rng.normal(size=k)
    c = rng.normal(size=(k, k)); lam = c @ c.T + np.eye(k)
    eta = np.concatenate([m, lam[np.tril_indices(k)], [3.1, 1.7]])
    f = _shell(k, mu0, lam0, a0, b0)
    z = np.hstack([x, y[:, None]])
    S = z.T @ (w[:, None] * z)
    val, g, H = f._terms(eta, S, float(w.sum()))
    ft = tr.mvn_regression_objective(x, y, k, mu0, lam0, a0, b0)
    te, tw = torch.tensor(eta), torch.tensor(w)
    assert abs(val - ft(te, tw).item()) < 1e-11 * max(1.0, abs(val))
    g_ad = torch.func.grad(ft)(te, tw).numpy()
    H_ad = torch.func.hessian(ft)(te, tw).numpy()
    np.testing.assert_allclose(g, g_ad, rtol=0, atol=1e-10 * np.max(np.abs(g_ad)))
    np.testing.assert_allclose(H, H_ad, rtol=0, atol=1e-10 * np.max(np.abs(H_ad)))
    M, cc = f._obs_terms(eta)
    G = 0.5 * np.einsum('na,kab,nb->nk', z, M, z) + cc[None, :]
    cross = torch.func.jacrev(torch.func.grad(ft, argnums=0), argnums=1)(te, tw).numpy()
    np.testing.assert_allclose(G.T, cross, rtol=0, atol=1e-10 * np.max(np.abs(cross)))


@pytest.mark.parametrize('k', [1, 3, 4])
def test_prior_hyper_parameter_closed_forms_match_ad(k):
    """d2 f / d eta d eps^T and d f / d eps for eps = prior mean, prior information (vector form of the symmetric matrix),
    prior shape and rate of `MVNRegressionObjective` against exact AD of the torch restatement with the prior as a
    variable: the cross Hessians `ParametricSensitivityLinearApproximation` needs for PRIOR sensitivity
    (LRVB/ModelSensitivity.py:596-602)."""
    rng = np.random.default_rng(40 + k)
    N = 30
    x = rng.normal(size=(N, k)); y = rng.normal(size=N); w = rng.uniform(0.5, 1.5, N)
    mu0 = rng.normal(size=k); a = rng.normal(size=(k, k)); lam0 = a @ a.T / k + np.eye(k)
    a0, b0 = 2.5, 1.3
    c = rng.normal(size=(k, k)); lam = c @ c.T + np.eye(k)
    eta = np.concatenate([rng.normal(size=k), lam[np.tril_indices(k)], [3.1, 1.7]])
    f = _shell(k, mu0, lam0, a0, b0)
    te, tw = torch.tensor(eta), torch.tensor(w)
    tri = torch.tril_indices(k, k)

    def sym(v):
        L = torch.zeros(k, k, dtype=v.dtype).index_put((tri[0], tri[1]), v)
        return L + L.T - torch.diag(torch.diagonal(L))
    builders = {
        'prior_mean': (mu0, lambda e: tr.mvn_regression_objective(x, y, k, e, lam0, a0, b0)),
        'prior_info': (lam0[np.tril_indices(k)], lambda e: tr.mvn_regression_objective(x, y, k, mu0, sym(e), a0, b0)),
        'prior_shape': (np.array([a0]), lambda e: tr.mvn_regression_objective(x, y, k, mu0, lam0, e[0], b0)),
        'prior_rate': (np.array([b0]), lambda e: tr.mvn_regression_objective(x, y, k, mu0, lam0, a0, e[0])),
    }
    for kind, (e0, build) in builders.items():
        F = lambda point, e: build(e)(point, tw)
        te0 = torch.tensor(e0)
        C = torch.func.jacrev(torch.func.grad(F, argnums=0), argnums=1)(te, te0).numpy()
        g = torch.func.grad(F, argnums=1)(te, te0).numpy()
        np.testing.assert_allclose(f._prior_hyper(kind, eta, 'cross'), C, rtol=0, atol=1e-9 * max(1.0, np.max(np.abs(C))), err_msg=kind)      # torch.polygamma itself is good to ~1e-10
        np.testing.assert_allclose(f._prior_hyper(kind, eta, 'grad'), g, rtol=0, atol=1e-11 * max(1.0, np.max(np.abs(g))), err_msg=kind)
