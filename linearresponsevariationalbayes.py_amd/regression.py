"""Batched conjugate-normal regression helpers (natural parameters of beta from a Gaussian
likelihood and prior; posterior mean / information).

Same functions and argument meaning as LRVB/regression_utils.py: mat_mul_last2dims :8-32,
matvec_mul_last2dims :34-57, get_nat_params_from_likelihood :59-88, get_nat_params_from_prior
:91-106, get_mvn_from_nat_params :109-115, get_regression_coefficients :117-121,
get_posterior_regression_coefficients :124-132.  These locate theta_hat of config 2 in closed
form; they act on small batched arrays on the host (the same X^T Lambda X contraction at scale is
`DeviceContext.weighted_gram`)."""
import numpy as np


def mat_mul_last2dims(x1, x2):
    """x1[..., n, m] @ x2[..., m, p] with broadcasting of the leading dimensions of the smaller one."""
    x1, x2 = np.asarray(x1), np.asarray(x2)
    assert x1.shape[-1] == x2.shape[-2]
    return np.matmul(x1, x2)


def matvec_mul_last2dims(x, y):
    """x[..., n, m] @ y[..., m]; y may carry extra leading dimensions beyond those of x."""
    x, y = np.asarray(x), np.asarray(y)
    assert x.shape[-1] == y.shape[-1]
    extra = y.ndim - (x.ndim - 1)
    assert extra >= 0
    xe = x.reshape(x.shape[:x.ndim - 2] + (1,) * extra + x.shape[-2:]) if extra else x
    if extra:
        # y: [lead..., extra..., m] -> align x's leading dims with y's first dims
        lead = x.ndim - 2
        yy = y.reshape(y.shape[:lead] + y.shape[lead:])
        return np.einsum('...nm,...m->...n', np.broadcast_to(
            xe, y.shape[:lead] + y.shape[lead:lead + extra] + x.shape[-2:]), yy)
    return np.einsum('...nm,...m->...n', x, y)


def get_nat_params_from_likelihood(y, x, info):
    """Coefficients of beta and beta beta^T in the Gaussian log likelihood: (X^T info y,
    -1/2 X^T info X * n_per_dim); `info` is a scalar (homoskedastic) or [..., n_t, n_t]."""
    y, x = np.asarray(y), np.asarray(x)
    assert y.shape[-1] == x.shape[-2]
    d1, d2 = y.ndim - 1, x.ndim - 2
    assert d1 >= d2
    n_per_dim = int(np.prod(y.shape[d2:d1])) if d1 > d2 else 1
    if np.isscalar(info):
        info_x = x * info
    else:
        info = np.asarray(info)
        assert info.shape[-1] == info.shape[-2] == x.shape[-2] and info.ndim == x.ndim
        info_x = mat_mul_last2dims(info, x)
    xt_info = np.swapaxes(info_x, -1, -2)                           # X^T info (info symmetric)
    nat_param1 = matvec_mul_last2dims(xt_info, y)
    nat_param2 = -0.5 * mat_mul_last2dims(np.swapaxes(x, -1, -2), info_x) * n_per_dim
    return nat_param1, nat_param2


def get_nat_params_from_prior(prior_means, prior_infos):
    prior_means, prior_infos = np.asarray(prior_means), np.asarray(prior_infos)
    assert prior_means.shape[-1] == prior_infos.shape[-1] == prior_infos.shape[-2]
    assert prior_means.ndim == prior_infos.ndim - 1
    r = prior_means.shape[-1]
    x = np.tile(np.eye(r), prior_means.shape[:-1] + (1, 1))
    return get_nat_params_from_likelihood(prior_means, x, prior_infos)


def get_mvn_from_nat_params(nat_param1, nat_param2):
    info = -2 * np.asarray(nat_param2)
    mean = matvec_mul_last2dims(np.linalg.inv(info), nat_param1)
    return mean, info


def get_regression_coefficients(y, x, info):
    return get_mvn_from_nat_params(*get_nat_params_from_likelihood(y, x, info))[0]


def get_posterior_regression_coefficients(y, x, info, prior_means, prior_infos):
    n1, n2 = get_nat_params_from_likelihood(y, x, info)
    p1, p2 = get_nat_params_from_prior(prior_means, prior_infos)
    return get_mvn_from_nat_params(n1 + p1, n2 + p2)
