"""Config 5: the closed forms of WishartMVNObjective (product host math) against exact AD of a torch
restatement built from the reference's blocks (WishartParam moments, wishart_entropy,
multivariate_normal_entropy, mvn_prior)."""
import numpy as np
import pytest
import torch

import lrvb_amd as vb
import torch_ref as tr
from lrvb_amd.quadform import duplication_matrix


def shell(d, mu0, lam0, nu0, w0):
    from scipy import special
    f = vb.WishartMVNObjective.__new__(vb.WishartMVNObjective)
    mm = d * (d + 1) // 2
    f._special, f.d, f.q = special, d, d + 1
    f._ms, f._ls, f._inu, f._vs = range(0, d), range(d, d + mm), d + mm, range(d + mm + 1, d + 2 * mm + 1)
    f._declare_priors(mu0, lam0, nu0, w0)
    f._dup = duplication_matrix(d)
    return f


def random_point(rng, d):
    m = rng.normal(size=d)
    a = rng.normal(size=(d, d)); lam_mu = a @ a.T / d + np.eye(d)
    b = rng.normal(size=(d, d)); v = (b @ b.T / d + np.eye(d)) * 0.3
    nu = d + 1.5 + rng.random()
    return np.concatenate([m, lam_mu[np.tril_indices(d)], [nu], v[np.tril_indices(d)]])


@pytest.mark.parametrize('d,N', [(1, 30), (2, 40), (4, 60)])
def test_closed_forms_match_ad(d, N):
    rng = np.random.default_rng(50 + d)
    y = rng.normal(size=(N, d)); w = rng.uniform(0.5, 1.5, N)
    mu0 = rng.normal(size=d); a = rng.normal(size=(d, d)); lam0 = a @ a.T / d + np.eye(d)
    c = rng.normal(size=(d, d)); w0 = c @ c.T / d + np.eye(d)
    nu0 = d + 3.0
    eta = random_point(rng, d)
    f = shell(d, mu0, lam0, nu0, w0)
    z = np.hstack([y, np.ones((N, 1))])
    S = z.T @ (w[:, None] * z)
    val, g, H = f._terms(eta, S, float(w.sum()))
    ft = tr.wishart_mvn_objective(y, d, mu0, lam0, nu0, w0)
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


@pytest.mark.parametrize('d', [1, 2, 4])
def test_prior_hyper_parameter_closed_forms_match_ad(d):
    """Cross Hessians and gradients of `WishartMVNObjective` with respect to its priors -- mean and information of the
    normal prior, degrees of freedom and inverse scale of the Wishart prior -- against exact AD of the torch restatement."""
    import torch
    import torch_ref as tr
    rng = np.random.default_rng(60 + d)
    N = 25
    y = rng.normal(size=(N, d)); w = rng.uniform(0.5, 1.5, N)
    mu0 = rng.normal(size=d); a = rng.normal(size=(d, d)); lam0 = a @ a.T / d + np.eye(d)
    b = rng.normal(size=(d, d)); w0 = b @ b.T / d + np.eye(d)
    nu0 = d + 2.5
    eta = random_point(rng, d)
    f = shell(d, mu0, lam0, nu0, w0)
    te, tw = torch.tensor(eta), torch.tensor(w)
    tri = torch.tril_indices(d, d)

    def sym(v):
        L = torch.zeros(d, d, dtype=v.dtype).index_put((tri[0], tri[1]), v)
        return L + L.T - torch.diag(torch.diagonal(L))
    builders = {
        'prior_mean': (mu0, lambda e: tr.wishart_mvn_objective(y, d, e, lam0, nu0, w0)),
        'prior_info': (lam0[np.tril_indices(d)], lambda e: tr.wishart_mvn_objective(y, d, mu0, sym(e), nu0, w0)),
        'prior_df': (np.array([nu0]), lambda e: tr.wishart_mvn_objective(y, d, mu0, lam0, e[0], w0)),
        'prior_inv_scale': (w0[np.tril_indices(d)], lambda e: tr.wishart_mvn_objective(y, d, mu0, lam0, nu0, sym(e))),
    }
    for kind, (e0, build) in builders.items():
        F = lambda point, e: build(e)(point, tw)
        te0 = torch.tensor(e0)
        C = torch.func.jacrev(torch.func.grad(F, argnums=0), argnums=1)(te, te0).numpy()
        g = torch.func.grad(F, argnums=1)(te, te0).numpy()
        np.testing.assert_allclose(f._prior_hyper(kind, eta, 'cross'), C, rtol=0, atol=1e-9 * max(1.0, np.max(np.abs(C))), err_msg=kind)      # torch.polygamma itself is good to ~1e-10
        np.testing.assert_allclose(f._prior_hyper(kind, eta, 'grad'), g, rtol=0, atol=1e-11 * max(1.0, np.max(np.abs(g))), err_msg=kind)
