"""the D x N weight sensitivity / influence of every observation on a set
of moments, `moment_jac @ (-H^-1 G^T)` (LRVB/ModelSensitivity.py:596-606 with hyper_par = weights;
Example.ipynb:425-441), streamed over the observations from the resident Cholesky factor.
Oracle: the dense formula from the numpy restatement's Hessian and per-observation gradients."""
import time

import numpy as np
import pytest

from oracle import models as om
from helpers import make_par, glm_data, rel_err, LOSS_NAME, on_torch_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


@pytest.mark.parametrize('loss', [om.GAUSSIAN, om.LOGISTIC, om.POISSON])
@pytest.mark.parametrize('N,P,Q', [(1, 3, 2), (37, 5, 5), (1000, 130, 7), (70001, 64, 16),
                                   # n_cols % 128 == 0: the streamed kernel (8-row chunks through LDS)
                                   (1003, 128, 5), (4099, 384, 33), (5, 256, 16)])
def test_box_layout_matches_dense_formula(vb, loss, N, P, Q):
    rng = np.random.default_rng(N + P + loss)
    p1 = P // 3
    par, lay = make_par(vb, [('box', 'u', p1, -np.inf, np.inf), ('box', 'pos', P - p1, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, loss)
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], lik_info=1.3, quad_A=np.full(P, 0.7), weights=w)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, lik_info=1.3, quad_A=np.full(P, 0.7))
    theta = rng.normal(size=lay.D) * 0.3
    M = rng.normal(size=(Q, lay.D))
    H = model.hessian(theta)
    ev = np.min(np.linalg.eigvalsh(H))
    if ev <= 0:                                             # away from an optimum: any SPD hess0 will do
        H = H + (0.1 - ev) * np.eye(lay.D)
    want = -(model.obs_grad(theta) @ np.linalg.solve(H, M.T))
    # through the reference-shaped class ...
    sens = vb.ParametricSensitivityLinearApproximation(
        objective_functor=fun, input_par=par, hyper_par=fun.weights_par, input_val0=theta, hyper_val0=w,
        hess0=H, stream_hyper=True)
    got = sens.get_doutput_dhyper_rows(M)
    assert got.shape == (N, Q)
    assert rel_err(got, want) < 1e-9
    # ... which equals the product the reference would form, and a row window of it
    if N <= 1000:
        full = vb.ParametricSensitivityLinearApproximation(
            objective_functor=fun, input_par=par, hyper_par=fun.weights_par, input_val0=theta, hyper_val0=w, hess0=H)
        assert rel_err(got, (M @ full.get_dinput_dhyper()).T) < 1e-9
    a, b = N // 3, max(N // 3, 2 * N // 3)
    if b > a:
        assert rel_err(sens.get_doutput_dhyper_rows(M, a, b), want[a:b]) < 1e-9
    # vector coordinates
    eta = lay.constrain(theta)
    Hv = model.hessian_vec(eta)
    ev = np.min(np.linalg.eigvalsh(Hv))
    if ev <= 0:
        Hv = Hv + (0.1 - ev) * np.eye(lay.V)
    Mv = rng.normal(size=(Q, lay.V))
    fun.ctx.chol_factor(Hv)
    got_v = fun.ctx.obs_influence(eta, Mv, is_free=False)
    assert rel_err(got_v, -(model.obs_grad_vec(eta) @ np.linalg.solve(Hv, Mv.T))) < 1e-9


def test_general_layout(vb):
    rng = np.random.default_rng(7)
    spec = [('box', 'pre', 2, -np.inf, np.inf), ('box', 'beta', 6, -1.0, np.inf), ('psd', 'm', 3, 0.2), ('simplex', 's', 2, 3)]
    par, lay = make_par(vb, spec)
    N, P, Q = 500, 6, 4
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    A = rng.normal(size=(lay.V, lay.V)); A = A @ A.T / lay.V + np.eye(lay.V)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', glm_param='beta', quad_A=A, weights=w)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, glm_off=2, quad_A=A)
    theta = rng.normal(size=lay.D) * 0.4
    H = model.hessian(theta)
    if np.min(np.linalg.eigvalsh(H)) <= 0:
        H = H + (0.1 - np.min(np.linalg.eigvalsh(H))) * np.eye(lay.D)
    M = rng.normal(size=(Q, lay.D))
    fun._push_state()
    fun.ctx.chol_factor(H)
    got = fun.ctx.obs_influence(theta, M)
    assert rel_err(got, -(model.obs_grad(theta) @ np.linalg.solve(H, M.T))) < 1e-9


def test_errors(vb):
    rng = np.random.default_rng(1)
    par, lay = make_par(vb, [('box', 'beta', 4, -np.inf, np.inf)])
    x, y, w = glm_data(rng, 50, 4, om.GAUSSIAN)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='gaussian', quad_A=np.ones(4))
    fun._push_state()
    with pytest.raises(RuntimeError):                       # no factor yet
        fun.ctx.obs_influence(np.zeros(4), np.eye(4))
    fun.ctx.chol_factor(np.eye(4))
    with pytest.raises(ValueError):
        fun.ctx.obs_influence(np.zeros(4), np.eye(3))
    with pytest.raises(ValueError):
        fun.ctx.obs_influence(np.zeros(4), np.eye(4), n0=10, n1=51)


def test_full_size_rows(vb):
    """N = 1e6 x D = 1024, Q = 16 moments: a row window against the dense formula built from the
    device's own obs_grad rows and Cholesky solve; additivity in M; timing printed."""
    import torch
    N, P, Q = 1_000_000, 1024, 16
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
    yv = torch.randn((N,), dtype=torch.float64, device=dev, generator=g)
    wv = torch.rand((N,), dtype=torch.float64, device=dev, generator=g) + 0.5
    blocks = [dict(kind=0, free_size=P - 256, vec_size=P - 256, dim0=P - 256, dim1=0, lb=-np.inf, ub=np.inf),
              dict(kind=0, free_size=256, vec_size=256, dim0=256, dim1=0, lb=0.0, ub=np.inf)]
    ctx = on_torch_stream(vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, lik_info=2.0, quad_kind=1), dev)
    ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, yv.data_ptr(), N, 1); ctx.set_weights_dev(wv.data_ptr(), N)
    ctx.set_data(2, np.ones(P))
    rng = np.random.default_rng(2)
    theta = rng.normal(size=P) * 0.1
    H = ctx.hessian(theta)
    ctx.chol_factor(H)
    M = rng.normal(size=(Q, P))
    t0 = time.perf_counter()
    out = ctx.obs_influence(theta, M)
    t1 = time.perf_counter()
    assert out.shape == (N, Q) and np.all(np.isfinite(out))
    a, b = 123_456, 123_456 + 300
    G = ctx.obs_grad(theta, a, b)
    want = -(G @ ctx.chol_solve(np.ascontiguousarray(M.T)))
    assert rel_err(out[a:b], want) < 1e-10
    M2 = rng.normal(size=(Q, P))
    both = ctx.obs_influence(theta, M + M2, a, b)
    assert rel_err(both, out[a:b] + ctx.obs_influence(theta, M2, a, b)) < 1e-11
    print('\n[influence N=1e6 D=1024 Q=16] {:.1f} ms for all observations (128 MB result to the host)'.format(1e3 * (t1 - t0)))
