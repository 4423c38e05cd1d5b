"""Closed-form exponential-family quantities: entropies, expectations, expected log priors --
the N-independent scalars of an ELBO (host-side).

Same names and argument meaning as LRVB/ExponentialFamilies.py (multivariate_digamma/gammaln
:5-13; entropies :20-82; e_log_det_wishart :88-94; e_log_inv_wishart_diag :97-102; lognormal
moments :104-109; get_e_log_gamma :111-112; Dirichlet moments :114-120; priors :186-212).
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
