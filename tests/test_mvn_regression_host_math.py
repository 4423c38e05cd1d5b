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
    f.mu0, f.lam0, f.a0, f.b0 = mu0, lam0, a0, b0
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
