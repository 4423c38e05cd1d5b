"""Config 4: the closed forms of LMMObjective (arrow-structured Hessian in vector coordinates, and its
Schur complement algebra) against exact AD of a torch restatement of doc/lmm.lyx:77-87."""
import numpy as np
import pytest
import torch

import lrvb_amd as vb
import torch_ref as tr


def make_par(p, G):
    from synthetic import lmm_par
    return lmm_par(vb, p, G)


def shell(par, p, G, priors):
    f = vb.LMMObjective.__new__(vb.LMMObjective)
    f.par, f.p, f.G = par, p, G
    f._index(par, ('beta', 'mu', 'tau_y', 'tau_mu', 'u'))
    f._declare_priors(*priors)
    return f


def random_eta(rng, p, G):
    m = rng.normal(size=p)
    a = rng.normal(size=(p, p)); lam = a @ a.T / p + np.eye(p)
    return np.concatenate([m, lam[np.tril_indices(p)], [0.3, 1.7, 3.0, 2.2, 2.5, 1.4],
                           rng.normal(size=G) * 0.5, rng.uniform(0.5, 2.0, G)])


def host_stats(x, y, gid, G, w):
    z = np.hstack([x, y[:, None]])
    S = z.T @ (w[:, None] * z)
    gs = np.zeros((G, z.shape[1] + 1))
    np.add.at(gs[:, 0], gid, w)
    np.add.at(gs[:, 1:], gid, w[:, None] * z)
    return np.concatenate([S.ravel(), gs.ravel()])


@pytest.mark.parametrize('p,G,N', [(1, 2, 20), (3, 5, 80)])
def test_arrow_hessian_matches_ad(p, G, N):
    rng = np.random.default_rng(p * 10 + G)
    x = rng.normal(size=(N, p)); y = rng.normal(size=N); w = rng.uniform(0.5, 1.5, N)
    gid = rng.integers(0, G, size=N); gid[:G] = np.arange(G)
    priors = (rng.normal(size=p), np.eye(p) * 0.7, 0.2, 0.5, (2.0, 1.5), (1.5, 0.8))
    par = make_par(p, G)
    f = shell(par, p, G, priors)
    f._external_stats = host_stats(x, y, gid, G, w)
    eta = random_eta(rng, p, G)
    ft = tr.lmm_objective(x, y, gid, G, priors[0], priors[1], priors[2], priors[3], priors[4], priors[5])
    te, tw = torch.tensor(eta), torch.tensor(w)
    assert abs(f.value_vec(eta) - ft(te, tw).item()) < 1e-11 * abs(ft(te, tw).item())
    g, H = f._dense_vec(eta)
    g_ad = torch.func.grad(ft)(te, tw).numpy()
    H_ad = torch.func.hessian(ft)(te, tw).numpy()
    np.testing.assert_allclose(g, g_ad, rtol=0, atol=1e-10 * np.max(np.abs(g_ad)))
    np.testing.assert_allclose(H, H_ad, rtol=0, atol=1e-10 * np.max(np.abs(H_ad)))
    # Schur complement algebra in vector coordinates: inverse of H restricted to the global block
    _, Hgg, (xrows, Hx), dl = f._arrow(eta)          # cross block row-sparse: p + 5 coupled rows
    ng = f.n_global
    assert xrows.size == p + 5 and Hx.shape == (p + 5, 2 * G)
    schur = Hgg.copy()
    schur[np.ix_(xrows, xrows)] -= (Hx / dl[None, :]) @ Hx.T
    np.testing.assert_allclose(np.linalg.inv(schur), np.linalg.inv(H_ad)[:ng, :ng], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('p,G,N', [(1, 2, 20), (3, 4, 60)])
def test_prior_hyper_parameter_closed_forms_match_ad(p, G, N):
    """Cross Hessians and gradients of the hierarchical model with respect to every prior it declares -- mean and information
    of the prior on beta, (mean, information) of the prior on mu, (shape, rate) of the two gamma priors -- against exact AD of
    the torch restatement with the prior as a variable.  The 2 G local rows of every cross Hessian are zero."""
    rng = np.random.default_rng(p * 7 + G)
    x = rng.normal(size=(N, p)); y = rng.normal(size=N); w = rng.uniform(0.5, 1.5, N)
    gid = rng.integers(0, G, size=N); gid[:G] = np.arange(G)
    a = rng.normal(size=(p, p))
    priors = (rng.normal(size=p), a @ a.T / p + np.eye(p) * 0.7, 0.2, 0.5, (2.0, 1.5), (1.5, 0.8))
    par = make_par(p, G)
    f = shell(par, p, G, priors)
    eta = random_eta(rng, p, G)
    te, tw = torch.tensor(eta), torch.tensor(w)
    tri = torch.tril_indices(p, p)

    def sym(v):
        L = torch.zeros(p, p, dtype=v.dtype).index_put((tri[0], tri[1]), v)
        return L + L.T - torch.diag(torch.diagonal(L))
    b0, l0, m0, k0, ty, tm = priors
    builders = {
        'beta_prior_mean': (b0, lambda e: tr.lmm_objective(x, y, gid, G, e, l0, m0, k0, ty, tm)),
        'beta_prior_info': (l0[np.tril_indices(p)], lambda e: tr.lmm_objective(x, y, gid, G, b0, sym(e), m0, k0, ty, tm)),
        'mu_prior': (np.array([m0, k0]), lambda e: tr.lmm_objective(x, y, gid, G, b0, l0, e[0], e[1], ty, tm)),
        'tau_y_prior': (np.array(ty), lambda e: tr.lmm_objective(x, y, gid, G, b0, l0, m0, k0, (e[0], e[1]), tm)),
        'tau_mu_prior': (np.array(tm), lambda e: tr.lmm_objective(x, y, gid, G, b0, l0, m0, k0, ty, (e[0], e[1]))),
    }
    ng = f.n_global
    for kind, (e0, build) in builders.items():
        F = lambda point, e: build(e)(point, tw)
        te0 = torch.tensor(np.asarray(e0, dtype=np.float64))
        C = torch.func.jacrev(torch.func.grad(F, argnums=0), argnums=1)(te, te0).numpy()
        g = torch.func.grad(F, argnums=1)(te, te0).numpy()
        scale = max(1.0, np.max(np.abs(C)))
        assert np.max(np.abs(C[ng:])) == 0.0                                  # priors do not touch the group parameters
        np.testing.assert_allclose(f._prior_hyper(kind, eta[:ng], 'cross'), C[:ng], rtol=0, atol=1e-9 * scale, err_msg=kind)
        np.testing.assert_allclose(f._prior_hyper(kind, eta[:ng], 'grad'), g, rtol=0, atol=1e-9 * max(1.0, np.max(np.abs(g))), err_msg=kind)
