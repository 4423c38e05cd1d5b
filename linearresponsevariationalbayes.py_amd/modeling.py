"""Numerical expectations of the non-conjugate logistic term.

Same names and argument meaning as LRVB/Modeling.py:16-60.  With `ctx=` (a models.DeviceContext) the quadrature sums
run on the GPU (`lrvb_gh_logistic`: one thread per element, nodes in LDS); without it they are plain numpy, for
closures evaluated on the host.  `logitnormal.LogitNormalRegressionObjective` is the model these terms are written
for, with its Hessian on the matrix cores.
"""
import math

import numpy as np
from scipy import stats


def get_e_logistic_term(y, z_mean, z_sd, std_draws, ctx=None):
    """sum_n E[y_n z_n - log(1 + exp(z_n))] for z_n ~ N(z_mean_n, z_sd_n^2), the expectation replaced by
    the average over the fixed standard-normal draws `std_draws`.  LRVB/Modeling.py:16-32."""
    y, z_mean, z_sd = np.asarray(y), np.asarray(z_mean), np.asarray(z_sd)
    assert z_sd.ndim == y.ndim
    assert z_mean.ndim == y.ndim
    std_draws = np.asarray(std_draws)
    if ctx is not None:            # the same sum as a quadrature rule: nodes d / sqrt(2), weights sqrt(pi) / n
        d = np.asarray(std_draws, dtype=np.float64).ravel()
        val = ctx.gh_logistic(np.broadcast_to(z_mean, y.shape), np.broadcast_to(z_sd, y.shape), d / math.sqrt(2.0),
                              np.full(d.size, math.sqrt(math.pi) / d.size))
        return np.sum(y * z_mean) - np.sum(val)
    z = z_mean[..., None] + z_sd[..., None] * std_draws
    return np.sum(y * z_mean) - np.sum(np.logaddexp(0.0, z)) / std_draws.size


def get_e_logistic_term_guass_hermite(z_mean, z_sd, gh_x, gh_w, aggregate_all=True, ctx=None):
    """E[log(1 + exp(z))] by Gauss-Hermite quadrature (nodes gh_x, weights gh_w), summed over everything
    or, with aggregate_all=False, per element.  The (misspelt) name is the reference's.
    LRVB/Modeling.py:35-52."""
    z_mean, z_sd = np.asarray(z_mean), np.asarray(z_sd)
    assert z_mean.shape == z_sd.shape
    if ctx is not None:
        term = ctx.gh_logistic(z_mean, z_sd, gh_x, gh_w)
        return np.sum(term) if aggregate_all else term
    z = z_mean[..., None] + math.sqrt(2.0) * z_sd[..., None] * np.asarray(gh_x)
    term = np.asarray(gh_w) * np.logaddexp(0.0, z) / math.sqrt(math.pi)
    return np.sum(term) if aggregate_all else np.sum(term, axis=z_sd.ndim)


def get_standard_draws(num_draws):
    """Standard-normal quantiles at num_draws evenly spaced interior probabilities.  LRVB/Modeling.py:55-58."""
    step = 1.0 / float(num_draws + 1)
    return stats.norm.ppf(np.linspace(step, 1 - step, num_draws))


def univariate_normal_log_prob(u, u_mean, u_info):
    """As the reference writes it (LRVB/Modeling.py:61-63) -- note `+ 0.5 * u_info`, not the log of it."""
    return -0.5 * u_info * (u - u_mean) ** 2 + 0.5 * u_info - 0.5 * np.log(2 * np.pi)
