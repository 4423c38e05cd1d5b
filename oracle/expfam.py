"""Oracle: exponential-family entropies, expectations and expected log priors used by the
config ELBOs.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates LRVB/ExponentialFamilies.py: multivariate_digamma/gammaln :5-13; entropies :20-82;
e_log_det_wishart :88-94; get_e_log_gamma :111-112; get_e_dirichlet / get_e_log_dirichlet
:114-120; lognormal moments :104-109; priors :186-204.  Pinned against scipy.stats in
tests/test_oracle_known_answers.py, the way LRVB/test_exponential_families.py:16-75 pins the
reference.
"""
import numpy as np
from scipy.special import digamma, gammaln


def multivariate_digamma(x, size):
    return float(np.sum(digamma(x - 0.5 * np.arange(int(size)))))


def multivariate_gammaln(x, size):
    return float(np.sum(gammaln(x - 0.5 * np.arange(int(size))))) + 0.25 * np.log(np.pi) * size * (size - 1.0)


def multinoulli_entropy(p, min_prob=1e-16):
    return -np.sum(p * np.log(p + min_prob), axis=1)


def univariate_normal_entropy(info_obs):
    return 0.5 * np.sum(1.0 + np.log(2.0 * np.pi) - np.log(info_obs))


def multivariate_normal_entropy(info_obs):
    sign, logdet = np.linalg.slogdet(info_obs)
    assert sign > 0
    k = info_obs.shape[0]
    return 0.5 * (k * (1.0 + np.log(2.0 * np.pi)) - logdet)


def gamma_entropy(shape, rate):
    return np.sum(shape - np.log(rate) + gammaln(shape) + (1.0 - shape) * digamma(shape))


def dirichlet_entropy(alpha):
    """alpha: axis 0 is the Dirichlet dimension; returns one entropy per remaining index."""
    alpha = np.asarray(alpha, dtype=np.float64)
    k = alpha.shape[0]
    a0 = np.sum(alpha, axis=0)
    log_beta = np.sum(gammaln(alpha), axis=0) - gammaln(a0)
    return log_beta - (k - a0) * digamma(a0) - np.sum((alpha - 1.0) * digamma(alpha), axis=0)


def wishart_entropy(df, v):
    k = float(v.shape[0])
    sign, log_det_v = np.linalg.slogdet(v)
    assert sign > 0
    return (0.5 * (k + 1.0) * log_det_v + 0.5 * k * (k + 1.0) * np.log(2.0)
            + multivariate_gammaln(0.5 * df, k)
            - 0.5 * (df - k - 1.0) * multivariate_digamma(0.5 * df, k) + 0.5 * df * k)


def e_log_det_wishart(df, v):
    k = float(v.shape[0])
    sign, log_det_v = np.linalg.slogdet(v)
    assert sign > 0
    return multivariate_digamma(0.5 * df, k) + k * np.log(2.0) + log_det_v


def get_e_lognormal(mu, sigma_sq):
    return np.exp(mu + 0.5 * sigma_sq)


def get_var_lognormal(mu, sigma_sq):
    return (np.exp(sigma_sq) - 1.0) * get_e_lognormal(mu, sigma_sq) ** 2


def get_e_log_gamma(shape, rate):
    return digamma(shape) - np.log(rate)


def get_e_dirichlet(alpha):
    return alpha / np.sum(alpha, axis=0, keepdims=True)


def get_e_log_dirichlet(alpha):
    return digamma(alpha) - digamma(np.sum(alpha, axis=0, keepdims=True))


def mvn_prior(prior_mean, prior_info, e_obs, cov_obs):
    d = e_obs - prior_mean
    return -0.5 * (float(d @ prior_info @ d) + float(np.trace(prior_info @ cov_obs)))


def uvn_prior(prior_mean, prior_info, e_obs, var_obs):
    return -0.5 * prior_info * ((e_obs - prior_mean) ** 2 + var_obs)


def gamma_prior(prior_shape, prior_rate, e_obs, e_log_obs):
    return (prior_shape - 1.0) * e_log_obs - prior_rate * e_obs


def dirichlet_prior(alpha, e_log_obs):
    assert np.shape(alpha) == np.shape(e_log_obs)
    return float(np.dot(np.asarray(alpha) - 1.0, e_log_obs))
