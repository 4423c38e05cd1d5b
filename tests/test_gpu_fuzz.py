"""Seeded random shapes through the whole C ABI surface of a declared GLM objective against the numpy
oracle: ragged N around the 8/16-row stages, widths around the 32/64/128-column granules, all three
losses, box / mixed layouts.  One process, one context per case."""
import numpy as np
import pytest

from oracle import models as om
from oracle import solvers as osv
from helpers import make_par, glm_data, rel_err, LOSS_NAME

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 31, 33, 63, 65, 127, 129, 255, 257, 1000, 4097]))
    P = int(rng.choice([1, 2, 3, 8, 15, 16, 31, 32, 33, 62, 64, 66, 100, 126, 128, 130, 192, 254, 256, 258, 384]))
    loss = int(rng.choice([om.GAUSSIAN, om.LOGISTIC, om.POISSON]))
    mixed = bool(rng.integers(0, 2)) and P <= 66
    return rng, N, P, loss, mixed


@pytest.mark.parametrize('seed', range(40))
def test_random_case(vb, seed):
    rng, N, P, loss, mixed = _case(seed)
    if mixed:
        spec = [('box', 'pre', 2, -np.inf, np.inf), ('box', 'beta', P, -1.0, np.inf), ('psd', 'm', 2, 0.1), ('simplex', 's', 1, 3)]
        off = 2
    else:
        p1 = P // 2
        spec = ([('box', 'u', p1, -np.inf, np.inf)] if p1 else []) + [('box', 'pos', P - p1, 0.0, 3.0 if seed % 3 == 0 else np.inf)]
        off = 0
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, loss)
    A = np.full(lay.V, 0.9)
    m = rng.normal(size=lay.V) * 0.1
    kw = dict(glm_param='beta') if mixed else {}
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], lik_info=1.2, quad_A=A, quad_m=m, weights=w, **kw)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, lik_info=1.2, glm_off=off, quad_A=A, quad_m=m)
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=lay.D) * 0.2
    tag = 'seed {} N {} P {} loss {} mixed {}'.format(seed, N, P, loss, mixed)
    assert abs(obj.fun_free(theta) - model.value(theta)) <= 1e-11 * max(1.0, abs(model.value(theta))), tag
    assert rel_err(obj.fun_free_grad(theta), model.grad(theta)) < 1e-10, tag
    Hw = model.hessian(theta)
    v = rng.normal(size=lay.D)
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < 1e-10, tag        # matrix-free: nothing has been built yet
    H = obj.fun_free_hessian(theta)
    assert rel_err(H, Hw) < 1e-10, tag
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < 1e-10, tag        # against the Hessian the build left resident
    G = model.obs_grad(theta)
    assert rel_err(fun.ctx.obs_grad(theta), G) < 1e-10 or np.max(np.abs(G)) == 0.0, tag
    assert rel_err(fun.gram(theta), G.T @ G) < 1e-10 or np.max(np.abs(G)) == 0.0, tag
    # solves on a positive definite shift of H
    ev = np.min(np.linalg.eigvalsh(Hw))
    Hs = Hw + (max(0.0, -ev) + 0.5) * np.eye(lay.D)
    fun.ctx.chol_factor(Hs)
    Q = int(rng.integers(1, 20))
    M = rng.normal(size=(Q, lay.D))
    assert rel_err(fun.ctx.lrvb_cov(M), osv.lrvb_covariance(Hs, M)) < 1e-9, tag
    B = rng.normal(size=(lay.D, 3))
    assert rel_err(fun.ctx.chol_solve(B), np.linalg.solve(Hs, B)) < 1e-9, tag
    want = -(G @ np.linalg.solve(Hs, M.T))
    got = fun.ctx.obs_influence(theta, M)
    assert rel_err(got, want) < 1e-9 or np.max(np.abs(want)) == 0.0, tag
    # blocked CG on the true Hessian when it is positive definite
    if ev > 1e-3 * np.max(np.abs(Hw)):
        Bq = rng.normal(size=(Q, lay.D))
        fun.ctx.set_tuning(0, 8 if seed % 2 else 0)                          # odd seeds: the matrix-free loop; even: the resident Hessian
        X, info, iters = fun.ctx.cg_solve_multi(theta, Bq, tol=1e-10)
        assert np.all(info == 0), tag
        assert rel_err(X, np.linalg.solve(Hw, Bq.T).T) < 1e-6, tag
