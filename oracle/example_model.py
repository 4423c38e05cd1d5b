"""Oracle: the model of the reference's Example.ipynb (cell 9, lines 247-274), restated in numpy.
TEST INFRASTRUCTURE (see oracle/__init__.py).

    loglik = -1/2 einsum('ni,ij,nj,n', y - x beta, Lambda, y - x beta, w) + 1/2 sum(w) log|Lambda|
    objective(free) = -loglik          (eval_objective, Example.ipynb:269-274)

with beta = ArrayParam((dx, dy), lb=0) and Lambda = PosDefMatrixParam(dy) pushed in that order.
The reference differentiates this closure with autograd; tests/torch_ref.py holds the same
function in torch so that torch.func gives the exact derivatives."""
import numpy as np
from . import packing as opk


def layout(dx, dy):
    return opk.Layout([opk.box_block(dx * dy, lb=0.0, name='beta'), opk.psd_block(dy, name='lambda')])


def unpack(eta, dx, dy):
    beta = eta[:dx * dy].reshape(dx, dy)
    lam = opk.psd_matrix_from_vector(eta[dx * dy:], dy)
    return beta, lam


def objective_vec(eta, x, y, w):
    dx, dy = x.shape[1], y.shape[1]
    beta, lam = unpack(np.asarray(eta, dtype=np.float64), dx, dy)
    r = y - x @ beta
    y_term = -0.5 * np.einsum('ni,ij,nj,n', r, lam, r, w)
    sign, logdet = np.linalg.slogdet(lam)
    assert sign > 0
    return -(y_term + 0.5 * np.sum(w) * logdet)


def objective_free(theta, x, y, w):
    lay = layout(x.shape[1], y.shape[1])
    return objective_vec(lay.constrain(theta), x, y, w)
