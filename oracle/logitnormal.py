"""CPU restatement of the non-conjugate logistic term and of the regression model built on it -- TEST INFRASTRUCTURE.

`gh_logistic` restates `get_e_logistic_term_guass_hermite(..., aggregate_all=False)` (LRVB/Modeling.py:36-52):
    z_vals = sqrt(2) * z_sd[..., None] * gh_x + z_mean[..., None];   sum_k gh_w[k] * log1p(exp(z_vals)) / sqrt(pi)
with log1p(exp(t)) written as logaddexp(0, t) (the reference's form is inf past t = 709; equal below).
`draws_logistic` restates `get_e_logistic_term` (:16-32).  The derivative functions are the derivatives of those SUMS
(what autograd returns for the reference's expressions), written out with numpy; `kl_terms` is the objective of a
logistic regression with q(beta_j) = N(mean_j, 1 / info_j) in the vector coordinates (mean, info) of a UVNParamVector
(LRVB/NormalParams.py:51-76) with value, gradient and dense Hessian.  Pinned by exact AD of an independent torch
restatement (tests/test_logitnormal_host_math.py) and by the reference's own Monte-Carlo check of the term
(LRVB/test_exponential_families.py:182-209, ported there).
"""
import math

import numpy as np


def gh_logistic(z_mean, z_sd, gh_x, gh_w):
    z_mean, z_sd = np.asarray(z_mean, dtype=np.float64), np.asarray(z_sd, dtype=np.float64)
    t = z_mean[..., None] + math.sqrt(2.0) * z_sd[..., None] * np.asarray(gh_x, dtype=np.float64)
    return np.sum(np.asarray(gh_w) * np.logaddexp(0.0, t), axis=-1) / math.sqrt(math.pi)


def draws_logistic(y, z_mean, z_sd, std_draws):
    y, z_mean, z_sd = np.asarray(y), np.asarray(z_mean), np.asarray(z_sd)
    d = np.asarray(std_draws, dtype=np.float64)
    t = z_mean[..., None] + z_sd[..., None] * d
    return np.sum(y * z_mean) - np.sum(np.logaddexp(0.0, t)) / d.size


def gh_logistic_derivs(z_mean, z_sd, gh_x, gh_w):
    """(value, d1 (..., 2) = [d/dmean, d/dsd], d2 (..., 3) = [mean mean, mean sd, sd sd]) of the quadrature sum."""
    z_mean, z_sd = np.asarray(z_mean, dtype=np.float64), np.asarray(z_sd, dtype=np.float64)
    xk = math.sqrt(2.0) * np.asarray(gh_x, dtype=np.float64)
    wk = np.asarray(gh_w, dtype=np.float64) / math.sqrt(math.pi)
    t = z_mean[..., None] + z_sd[..., None] * xk
    sg = 0.5 * (1.0 + np.tanh(0.5 * t))                      # sigmoid
    s2 = sg * (1.0 - sg)
    val = np.sum(wk * np.logaddexp(0.0, t), axis=-1)
    d1 = np.stack([np.sum(wk * sg, axis=-1), np.sum(wk * sg * xk, axis=-1)], axis=-1)
    d2 = np.stack([np.sum(wk * s2, axis=-1), np.sum(wk * s2 * xk, axis=-1), np.sum(wk * s2 * xk * xk, axis=-1)], axis=-1)
    return val, d1, d2


def kl_terms(eta, x, y, w, prior_info, gh_x, gh_w):
    """Value, gradient (2 P) and Hessian (2 P x 2 P) in the coordinates eta = [mean | info] of
        sum_n w_n (E log(1 + e^{z_n}) - y_n x_n . mean) + 1/2 prior_info sum_j (mean_j^2 + 1 / info_j) + 1/2 sum_j log info_j,
    z_n ~ N(x_n . mean, x_n^2 . (1 / info))."""
    x = np.asarray(x, dtype=np.float64)
    N, P = x.shape
    mean, info = np.asarray(eta[:P], dtype=np.float64), np.asarray(eta[P:], dtype=np.float64)
    var = 1.0 / info
    x2 = x * x
    mu, v = x @ mean, x2 @ var
    sd = np.sqrt(v)
    phi, d1, d2 = gh_logistic_derivs(mu, sd, gh_x, gh_w)
    val = np.sum(w * (phi - y * mu)) + 0.5 * prior_info * (np.sum(mean ** 2) + np.sum(var)) + 0.5 * np.sum(np.log(info))
    # per observation, in (mu, v): sd = sqrt(v)
    s1, s2 = 0.5 / sd, -0.25 / (sd * v)
    p_mu, p_v = d1[:, 0] - y, d1[:, 1] * s1
    p_mumu, p_muv, p_vv = d2[:, 0], d2[:, 1] * s1, d2[:, 2] * s1 * s1 + d1[:, 1] * s2
    # v_n = sum_j x2_nj / info_j
    dv = -x2 * (var * var)[None, :]                          # d v_n / d info_j
    g_mean = x.T @ (w * p_mu) + prior_info * mean
    g_info = dv.T @ (w * p_v) - 0.5 * prior_info * var * var + 0.5 / info
    H = np.zeros((2 * P, 2 * P))
    H[:P, :P] = x.T @ ((w * p_mumu)[:, None] * x) + prior_info * np.eye(P)
    H[:P, P:] = x.T @ ((w * p_muv)[:, None] * dv)
    H[P:, :P] = H[:P, P:].T
    d2v = 2.0 * x2 * (var ** 3)[None, :]                     # d2 v_n / d info_j^2 (diagonal in j)
    H[P:, P:] = dv.T @ ((w * p_vv)[:, None] * dv) + np.diag(d2v.T @ (w * p_v) + prior_info * var ** 3 - 0.5 / info ** 2)
    return val, np.concatenate([g_mean, g_info]), H
