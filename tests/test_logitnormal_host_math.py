"""Oracle of the non-conjugate logistic term (LRVB/Modeling.py:16-52) and of the regression model built on it: pinned by
the reference's own Monte-Carlo check (LRVB/test_exponential_families.py:182-209), by closed-form identities, and by exact
AD of an independent torch restatement."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import logitnormal as ol          # noqa: E402
from scipy import stats                       # noqa: E402


def torch_kl(eta, x, y, w, tau, gx, gw):
    P = x.shape[1]
    mean, info = eta[:P], eta[P:]
    var = 1.0 / info
    mu, v = x @ mean, (x * x) @ var
    t = mu[:, None] + math.sqrt(2.0) * torch.sqrt(v)[:, None] * gx[None, :]
    phi = (gw[None, :] * torch.nn.functional.softplus(t)).sum(dim=1) / math.sqrt(math.pi)
    return (w * (phi - y * mu)).sum() + 0.5 * tau * ((mean ** 2).sum() + var.sum()) + 0.5 * torch.log(info).sum()


def problem(N, P, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(N, P)) / math.sqrt(P)
    beta = rng.normal(size=P)
    y = (rng.uniform(size=N) < 1.0 / (1.0 + np.exp(-x @ beta * 2.0))).astype(np.float64)
    w = rng.uniform(0.5, 1.5, size=N)
    eta = np.concatenate([rng.normal(size=P) * 0.5, rng.uniform(0.5, 3.0, size=P)])
    return x, y, w, eta


def test_reference_monte_carlo_check_of_the_draws_form():
    """The reference's own test of get_e_logistic_term: against 10000 normal draws, tolerance from the draw count."""
    rng = np.random.default_rng(3)
    z_mean = rng.random((3, 2)); z_sd = np.exp(z_mean); y = rng.random((3, 2))
    step = 1.0 / 101.0
    std_draws = stats.norm.ppf(np.linspace(step, 1 - step, 100))           # get_standard_draws(100), Modeling.py:55-58
    z_draws = stats.norm(loc=z_mean, scale=z_sd).rvs((10000, 3, 2), random_state=rng)
    draws = y[None] * z_draws - np.log1p(np.exp(z_draws))
    tol = np.max(3 * np.std(draws, axis=0) / np.sqrt(100))
    assert abs(np.sum(np.mean(draws, axis=0)) - ol.draws_logistic(y, z_mean, z_sd, std_draws)) < tol


def test_quadrature_identities():
    gx, gw = np.polynomial.hermite.hermgauss(40)
    m = np.array([-3.0, -0.4, 0.0, 1.7, 6.0]); s = np.array([0.0, 0.3, 1.0, 2.0, 0.5])
    val = ol.gh_logistic(m, s, gx, gw)
    assert abs(val[0] - np.logaddexp(0.0, -3.0)) < 1e-14                   # sd = 0: the plug-in value
    # log(1 + e^z) - log(1 + e^-z) = z  ->  E softplus(z) - E softplus(-z) = mean (the rule is symmetric)
    assert np.max(np.abs(val - ol.gh_logistic(-m, s, gx, gw) - m)) < 1e-13
    # the draws form is the same rule with nodes d / sqrt(2), weights sqrt(pi) / n
    d = stats.norm.ppf(np.linspace(0.01, 0.99, 99))
    a = ol.draws_logistic(np.zeros(5), m, s, d)
    b = -np.sum(ol.gh_logistic(m, s, d / math.sqrt(2.0), np.full(99, math.sqrt(math.pi) / 99)))
    assert abs(a - b) < 1e-12
    # far tail: no overflow (the reference's log1p(exp(t)) is inf here)
    assert np.isfinite(ol.gh_logistic(np.array([800.0]), np.array([1.0]), gx, gw)).all()


def test_term_derivatives_match_ad():
    gx, gw = np.polynomial.hermite.hermgauss(25)
    rng = np.random.default_rng(5)
    m = rng.normal(size=11) * 2; s = rng.uniform(0.05, 2.5, size=11)
    val, d1, d2 = ol.gh_logistic_derivs(m, s, gx, gw)
    tm, ts = torch.tensor(m, requires_grad=True), torch.tensor(s, requires_grad=True)
    tgx, tgw = torch.tensor(gx), torch.tensor(gw)

    def f(a, b):
        t = a[:, None] + math.sqrt(2.0) * b[:, None] * tgx[None, :]
        return (tgw[None, :] * torch.nn.functional.softplus(t)).sum(dim=1) / math.sqrt(math.pi)
    out = f(tm, ts)
    assert np.max(np.abs(out.detach().numpy() - val)) < 1e-13
    ga, gb = torch.autograd.grad(out.sum(), (tm, ts), create_graph=True)
    assert np.max(np.abs(ga.detach().numpy() - d1[:, 0])) < 1e-13 and np.max(np.abs(gb.detach().numpy() - d1[:, 1])) < 1e-13
    haa, hab = torch.autograd.grad(ga.sum(), (tm, ts), retain_graph=True)
    hbb = torch.autograd.grad(gb.sum(), ts)[0]
    assert np.max(np.abs(haa.numpy() - d2[:, 0])) < 1e-13
    assert np.max(np.abs(hab.numpy() - d2[:, 1])) < 1e-13
    assert np.max(np.abs(hbb.numpy() - d2[:, 2])) < 1e-12


@pytest.mark.parametrize('N,P', [(40, 3), (200, 7)])
def test_model_value_gradient_hessian_match_ad(N, P):
    x, y, w, eta = problem(N, P, seed=N + P)
    gx, gw = np.polynomial.hermite.hermgauss(20)
    val, g, H = ol.kl_terms(eta, x, y, w, 0.7, gx, gw)
    te = torch.tensor(eta, requires_grad=True)
    args = (torch.tensor(x), torch.tensor(y), torch.tensor(w), 0.7, torch.tensor(gx), torch.tensor(gw))
    tv = torch_kl(te, *args)
    assert abs(tv.item() - val) < 1e-11 * max(1.0, abs(val))
    tg = torch.autograd.grad(tv, te)[0].numpy()
    assert np.max(np.abs(tg - g)) < 1e-10 * max(1.0, np.max(np.abs(g)))
    tH = torch.autograd.functional.hessian(lambda e: torch_kl(e, *args), torch.tensor(eta)).numpy()
    assert np.max(np.abs(tH - H)) < 1e-9 * np.max(np.abs(H))
    assert np.max(np.abs(H - H.T)) < 1e-12 * np.max(np.abs(H))
