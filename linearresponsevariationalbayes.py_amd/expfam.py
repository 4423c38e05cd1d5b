"""Closed-form exponential-family quantities: entropies, expectations, expected log priors --
the N-independent scalars of an ELBO (host-side).

Same names and argument meaning as LRVB/ExponentialFamilies.py (multivariate_digamma/gammaln
:5-13; entropies :20-82; e_log_det_wishart :88-94; e_log_inv_wishart_diag :97-102; lognormal
moments :104-109; get_e_log_gamma :111-112; Dirichlet moments :114-120; Gauss-Hermite expectations
:126-168; get_uvn_from_natural_parameters :176-179; priors :186-221).
"""
import math

import numpy as np
from scipy import special


def multivariate_digamma(x, size):
    return np.sum(special.digamma(x - 0.5 * np.arange(int(size))))


def multivariate_gammaln(x, size):
    return np.sum(special.gammaln(x - 0.5 * np.arange(int(size)))) + \
        0.25 * np.log(np.pi) * size * (size - 1.0)


def multinoulli_entropy(p, min_prob=1e-16):
    return -1 * np.sum(p * np.log(p + min_prob), axis=1)


def univariate_normal_entropy(info_obs):
    return 0.5 * np.sum(1 + np.log(2 * math.pi) - np.log(info_obs))


def multivariate_normal_entropy(info_obs):
    sign, logdet = np.linalg.slogdet(info_obs)
    assert sign > 0
    k = info_obs.shape[0]
    return 0.5 * (k + k * np.log(2 * math.pi) - logdet)


def gamma_entropy(shape, rate):
    return np.sum(shape - np.log(rate) + special.gammaln(shape) + (1 - shape) * special.digamma(shape))


def dirichlet_entropy(alpha):
    k = alpha.shape[0]
    sum_alpha = np.sum(alpha, axis=0)
    log_beta = np.sum(special.gammaln(alpha), axis=0) - special.gammaln(sum_alpha)
    return log_beta - (k - sum_alpha) * special.digamma(sum_alpha) - \
        np.sum((alpha - 1) * special.digamma(alpha), axis=0)


def beta_entropy(tau):
    a, b = tau[:, 0], tau[:, 1]
    lbeta = special.gammaln(a) + special.gammaln(b) - special.gammaln(a + b)
    return np.sum(lbeta - (a - 1.) * special.digamma(a) - (b - 1.) * special.digamma(b) +
                  (a + b - 2) * special.digamma(a + b))


def wishart_entropy(df, v):
    k = float(v.shape[0])
    assert v.shape[0] == v.shape[1]
    s, log_det_v = np.linalg.slogdet(v)
    assert s > 0
    return 0.5 * (k + 1) * log_det_v + 0.5 * k * (k + 1) * np.log(2) + \
        multivariate_gammaln(0.5 * df, k) - \
        0.5 * (df - k - 1) * multivariate_digamma(0.5 * df, k) + 0.5 * df * k


def e_log_det_wishart(df, v):
    k = float(v.shape[0])
    assert v.shape[0] == v.shape[1]
    s, log_det_v = np.linalg.slogdet(v)
    assert s > 0
    return multivariate_digamma(0.5 * df, k) + k * np.log(2) + log_det_v


def e_log_inv_wishart_diag(df, v):
    k = float(v.shape[0])
    assert v.shape[0] == v.shape[1]
    return np.log(np.diag(np.linalg.inv(v))) - special.digamma(0.5 * (df - k + 1)) - np.log(2)


def get_e_lognormal(mu, sigma_sq):
    return np.exp(mu + 0.5 * sigma_sq)


def get_var_lognormal(mu, sigma_sq):
    return (np.exp(sigma_sq) - 1) * get_e_lognormal(mu, sigma_sq) ** 2


def get_e_log_gamma(shape, rate):
    return special.digamma(shape) - np.log(rate)


def get_e_dirichlet(alpha):
    return alpha / np.sum(alpha, 0, keepdims=True)


def get_e_log_dirichlet(alpha):
    return special.digamma(alpha) - special.digamma(np.sum(alpha, 0, keepdims=True))


def mvn_prior(prior_mean, prior_info, e_obs, cov_obs):
    obs_diff = e_obs - prior_mean
    return -0.5 * (np.dot(obs_diff, np.matmul(prior_info, obs_diff)) + np.trace(np.matmul(prior_info, cov_obs)))


def uvn_prior(prior_mean, prior_info, e_obs, var_obs):
    return -0.5 * (prior_info * ((e_obs - prior_mean) ** 2 + var_obs))


def gamma_prior(prior_shape, prior_rate, e_obs, e_log_obs):
    return (prior_shape - 1) * e_log_obs - prior_rate * e_obs


def exponential_prior(lambda_par, e_obs):
    return -1 * lambda_par * e_obs


def dirichlet_prior(alpha, log_e_obs):
    assert np.shape(alpha) == np.shape(log_e_obs), 'shape of alpha and log_e_obs do not match'
    return np.dot(alpha - 1, log_e_obs)


def expected_ljk_prior(lkj_param, df, v):
    e_log_r = -1 * e_log_det_wishart(df, v) - np.sum(e_log_inv_wishart_diag(df, v))
    return (lkj_param - 1) * e_log_r


# ---- Gauss-Hermite expectations under arrays of univariate normals --------------------------
def get_e_fun_normal(means, infos, gh_loc, gh_weights, fun):
    """E[fun(X)] elementwise for X ~ N(means, 1 / infos), by Gauss-Hermite quadrature with nodes
    gh_loc and weights gh_weights (numpy.polynomial.hermite.hermgauss convention: weight exp(-x^2)).
    The node axis is appended last and summed out.  LRVB/ExponentialFamilies.py:126-143."""
    means = np.asarray(means, dtype=np.float64)
    infos = np.asarray(infos, dtype=np.float64)
    assert means.shape == infos.shape
    nodes = means[..., None] + math.sqrt(2.0) * np.asarray(gh_loc) / np.sqrt(infos[..., None])
    return np.sum(np.asarray(gh_weights) * fun(nodes), axis=means.ndim) / math.sqrt(math.pi)


def get_e_logitnormal(lognorm_means, lognorm_infos, gh_loc, gh_weights):
    """E[expit(X)].  LRVB/ExponentialFamilies.py:145-150."""
    return get_e_fun_normal(lognorm_means, lognorm_infos, gh_loc, gh_weights, special.expit)


def _log_expit(x):
    # log(expit(x)) without overflow: -log1p(exp(-x)) where that is finite, x itself far in the left tail
    # (the reference switches branches at x = -100 and floors the first at -1e16, :156-157)
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(over='ignore'):
        right = np.maximum(-np.log1p(np.exp(-x)), -1e16)
    return np.where(x > -1e2, right, x)


def get_e_log_logitnormal(lognorm_means, lognorm_infos, gh_loc, gh_weights):
    """(E[log V], E[log(1 - V)]) for V = expit(X); the second follows from log(1 - v) = log v - x.
    LRVB/ExponentialFamilies.py:152-168."""
    e_log_v = get_e_fun_normal(lognorm_means, lognorm_infos, gh_loc, gh_weights, _log_expit)
    return e_log_v, e_log_v - np.asarray(lognorm_means, dtype=np.float64)


def get_uvn_from_natural_parameters(e_term, e2_term):
    """(mean, info) of the normal whose log density is e_term x + e2_term x^2 + const.
    LRVB/ExponentialFamilies.py:176-179."""
    x_info = -2.0 * e2_term
    return e_term / x_info, x_info


def get_e_dp_prior_logitnorm_approx(alpha, lognorm_means, lognorm_infos, gh_loc, gh_weights):
    """Expected Beta(1, alpha) stick prior under logit-normal sticks: (alpha - 1) E[log(1 - V)],
    elementwise.  LRVB/ExponentialFamilies.py:214-221."""
    _, e_log_1mv = get_e_log_logitnormal(lognorm_means, lognorm_infos, gh_loc, gh_weights)
    return (alpha - 1) * e_log_1mv
