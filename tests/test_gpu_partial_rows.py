"""SURVEY.md section 8(a) rows A10-A12 on the device: `TwoParameterObjective.fun_grad2`
(LRVB/SparseObjectives.py:381-387), `ParameterConverter` (:245-308) and the deprecated all-in-one
`ParametricSensitivity` (:487-573), each against the numpy oracle on the same seeded inputs."""
import warnings

import numpy as np
import pytest

from oracle import models as om, packing as opk
from helpers import make_par, glm_data, rel_err, LOSS_NAME

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


@pytest.mark.parametrize('loss,N,P', [(om.GAUSSIAN, 700, 40), (om.LOGISTIC, 1501, 130), (om.POISSON, 900, 33)])
def test_fun_grad2_weights_and_tilt(vb, loss, N, P):
    rng = np.random.default_rng(N)
    spec = [('box', 'u', P - 12, -np.inf, np.inf), ('box', 'pos', 12, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, loss)
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], lik_info=1.7, quad_A=np.full(P, 0.6))
    # the functor reads its hyper-parameter objects at every evaluation: a lower-bounded weight vector takes the
    # place of the default unconstrained one (free coordinates f = log w)
    fun.weights_par = wpar = vb.VectorParam('weights', N, lb=0.0, val=np.ones(N))
    tpar = fun.tilt_par
    theta = rng.normal(size=P) * 0.2
    eta = lay.constrain(theta)
    losses = om.loss_terms(loss, y, x @ eta, 1.7)[0]
    two = vb.TwoParameterObjective(par, wpar, fun)
    assert rel_err(two.fun_grad2(theta, w, True, False), losses) < 1e-12
    assert rel_err(two.fun_grad2(eta, w, False, False), losses) < 1e-12
    assert rel_err(two.fun_grad2(theta, np.log(w), True, True), losses * w) < 1e-12     # free weights: w = exp(f)
    assert rel_err(wpar.get_vector(), w) < 1e-14 and rel_err(par.get_free(), theta) < 1e-14
    # rows of the loss vector on their own, and the value as their weighted sum
    assert rel_err(fun.ctx.obs_loss(theta, 100, 300), losses[100:300]) < 1e-12
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, lik_info=1.7, quad_A=np.full(P, 0.6))
    assert abs(two.fun_free(theta, np.log(w)) - model.value(theta)) < 1e-11 * abs(model.value(theta))
    two_t = vb.TwoParameterObjective(par, tpar, fun)
    b = rng.normal(size=P)
    assert rel_err(two_t.fun_grad2(theta, b, True, False), eta) < 1e-13
    with pytest.raises(ValueError):
        fun.ctx.obs_loss(theta, 5, N + 1)


def test_parameter_converter_through_device_moments(vb):
    """ParameterConverter whose converter is a device moment map (LinearMoments: its Jacobian comes from the library's
    packing Jacobian), a declared elementwise converter, and an opaque closure."""
    rng = np.random.default_rng(3)
    spec = [('box', 'a', 4, -np.inf, np.inf), ('box', 'b', 3, 0.5, np.inf), ('psd', 'm', 3, 0.2)]
    par, lay = make_par(vb, spec)
    theta = rng.normal(size=lay.D) * 0.4
    par.set_free(theta)
    Q = 5
    B = rng.normal(size=(Q, lay.V))
    out = vb.VectorParam('moments', Q)
    mom = vb.LinearMoments(par, B=B)
    # the device moment functor gives d (B eta) / d theta = B J(theta)
    assert rel_err(mom.jacobian(theta, True), B @ lay.jac(theta)) < 1e-13
    conv = vb.ParameterConverter(par, out, vb.LinearConverter(par, out, B))
    assert rel_err(conv.free_to_vec_jacobian(theta), B @ lay.jac(theta)) < 1e-13
    assert rel_err(conv.vec_to_vec_jacobian(lay.constrain(theta)), B) < 1e-14
    assert rel_err(par.get_free(), theta) < 1e-13                              # inputs restored (:280-292)
    opaque = vb.ParameterConverter(par, out, lambda: out.set_vector(B @ par.get_vector()))
    assert rel_err(opaque.free_to_vec_jacobian(theta), B @ lay.jac(theta)) < 1e-8
    assert rel_err(conv.converter_free_to_vec(theta), B @ lay.constrain(theta)) < 1e-14


def test_deprecated_parametric_sensitivity_on_the_device(vb):
    """The all-in-one class: device Hessian + device Cholesky + cross Hessian with the weights, output map through a
    ParameterConverter; against the oracle's dense formulas  S = -H^-1 G^T,  d out / d w = M S."""
    rng = np.random.default_rng(12)
    N, P = 400, 24
    spec = [('box', 'u', 16, -np.inf, np.inf), ('box', 'pos', 8, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 0.9), weights=w)
    wpar = fun.weights_par
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.9))
    objective = vb.Objective(par, fun)
    theta0, _ = vb.OptimizationUtils.minimize_objective_trust_ncg(objective, np.zeros(P), False, gtol=1e-9, disp=False)
    B = rng.normal(size=(6, P))
    out_par = vb.VectorParam('out', 6)
    conv = vb.LinearConverter(par, out_par, B)
    par.set_free(theta0); conv()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        ps = vb.ParametricSensitivity(fun, par, out_par, wpar, conv, optimal_input_par=theta0)
        assert any(issubclass(c.category, DeprecationWarning) for c in caught)
    H = model.hessian(theta0)
    S = -np.linalg.solve(H, model.obs_grad(theta0).T)
    assert rel_err(ps.objective_hessian, H) < 1e-11
    assert rel_err(ps.get_dinput_dhyper(), S) < 1e-9
    M = B @ lay.jac(theta0)
    assert rel_err(ps.get_doutput_dhyper(), M @ S) < 1e-9
    # leave-one-out style prediction: drop 10 % of observation 7's weight
    w_new = w.copy(); w_new[7] *= 0.9
    pred = ps.predict_input_par_from_hyperparameters(w_new)
    assert rel_err(pred, theta0 + S @ (w_new - w)) < 1e-9
    lin = ps.predict_output_par_from_hyperparameters(w_new, linear=True)
    assert rel_err(lin, B @ lay.constrain(theta0) + M @ S @ (w_new - w)) < 1e-9
    full = ps.predict_output_par_from_hyperparameters(w_new, linear=False)
    assert rel_err(full, B @ lay.constrain(pred)) < 1e-12
    # a Hessian supplied by the caller is used as is (:531-536)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ps2 = vb.ParametricSensitivity(fun, par, out_par, wpar, conv, optimal_input_par=theta0, objective_hessian=H)
    assert rel_err(ps2.get_dinput_dhyper(), S) < 1e-9


def test_north_star_names(vb):
    """BASELINE.json's names for the path resolve and run on the device: get_kl_hessian = Objective.fun_free_hessian,
    get_lrvb_cov = M H^-1 M^T, HyperparameterSensitivityLinearApproximation = the linear-approximation class."""
    rng = np.random.default_rng(5)
    N, P = 600, 20
    par, lay = make_par(vb, [('box', 'u', 12, -np.inf, np.inf), ('box', 'pos', 8, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, om.POISSON)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='poisson', quad_A=np.full(P, 1.1), weights=w)
    model = om.DeclaredModel(lay, loss=om.POISSON, x=x, y=y, w=w, quad_A=np.full(P, 1.1))
    objective = vb.Objective(par, fun)
    theta = rng.normal(size=P) * 0.1
    H = vb.get_kl_hessian(objective, theta)
    assert rel_err(H, model.hessian(theta)) < 1e-11
    M = rng.normal(size=(4, P))
    want = M @ np.linalg.solve(model.hessian(theta), M.T)
    assert rel_err(vb.get_lrvb_cov(objective, theta, M), want) < 1e-9
    assert rel_err(vb.ModelSensitivity.get_lrvb_cov(objective, theta, M, kl_hessian=H), want) < 1e-9
    assert vb.HyperparameterSensitivityLinearApproximation is vb.ParametricSensitivityLinearApproximation


@pytest.mark.parametrize('order', ['library first', 'torch first'])
def test_one_hip_runtime_whatever_the_import_order(order):
    """The library maps the HIP runtime torch bundles (without importing torch), so both import orders end with ONE
    libamdhip64 in the process, a working torch.cuda and a working library."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    first, second = ('import lrvb_amd as vb; vb._hip.load()', 'import torch') if order == 'library first' else \
                    ('import torch', 'import lrvb_amd as vb; vb._hip.load()')
    code = '''
import sys
sys.path.insert(0, {root!r})
{first}
assert ('torch' in sys.modules) == {torch_first}
{second}
import numpy as np, torch
import lrvb_amd as vb
assert torch.cuda.is_available() and vb._hip.device_count() >= 1
par = vb.VectorParam('x', 3)
H = vb.Objective(par, vb.QuadraticObjective(par, A=np.diag([1.0, 2.0, 3.0]))).fun_free_hessian(np.zeros(3))
assert np.allclose(H, np.diag([1.0, 2.0, 3.0]))
t = torch.ones(4, device='cuda') * 2
assert float(t.sum()) == 8.0
copies = sorted({{ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln}})
assert len(copies) == 1, copies
print('ok', copies[0])
'''.format(root=root, first=first, second=second, torch_first=(order == 'torch first'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]


def test_every_cross_hessian_variant_and_vector_jacobian(vb):
    """The six cross-Hessian methods of TwoParameterObjective (LRVB/SparseObjectives.py:389-438), `fun_grad1`, `eval_fun`,
    and `Objective.fun_vector_jacobian`, for the weights in lower-bounded free coordinates (w = exp f) and the tilt."""
    rng = np.random.default_rng(8)
    N, P = 400, 10
    spec = [('box', 'u', 6, -np.inf, np.inf), ('box', 'pos', 4, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 0.6))
    fun.weights_par = wpar = vb.VectorParam('weights', N, lb=0.0, val=np.ones(N))
    theta = rng.normal(size=P) * 0.2
    eta = lay.constrain(theta)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.6))
    G_free = model.obs_grad(theta).T                       # D x N: d2 f / d theta d w^T
    G_vec = model.obs_grad_vec(eta).T                      # V x N
    two = vb.TwoParameterObjective(par, wpar, fun)
    fw = np.log(w)
    assert rel_err(two.fun_hessian_free1_vector2(theta, w), G_free) < 1e-11
    assert rel_err(two.fun_free_hessian12(theta, fw), G_free * w[None, :]) < 1e-11           # d w / d f = w
    assert rel_err(two.fun_free_hessian21(theta, fw), (G_free * w[None, :]).T) < 1e-11
    assert rel_err(two.fun_vector_hessian12(eta, w), G_vec) < 1e-11
    assert rel_err(two.fun_vector_hessian21(eta, w), G_vec.T) < 1e-11
    assert rel_err(two.fun_hessian_vector1_free2(eta, fw), G_vec * w[None, :]) < 1e-11
    assert rel_err(two.fun_grad1(theta, w, True, False), model.grad(theta)) < 1e-11
    assert rel_err(two.fun_grad1(eta, fw, False, True), model.grad_vec(eta)) < 1e-11
    assert abs(two.eval_fun(theta, w, True, False) - model.value(theta)) < 1e-11 * abs(model.value(theta))
    assert rel_err(par.get_free(), theta) < 1e-14 and rel_err(wpar.get_vector(), w) < 1e-14   # both left at the point
    # the tilt: d2 f / d theta d b^T = s J^T, in vector coordinates s I
    two_t = vb.TwoParameterObjective(par, fun.tilt_par, fun)
    b = rng.normal(size=P)
    assert rel_err(two_t.fun_hessian_free1_vector2(theta, b), model.cross_hessian_tilt(theta)) < 1e-13
    assert rel_err(two_t.fun_vector_hessian12(eta, b), np.eye(P)) < 1e-14
    # Objective.fun_vector_jacobian of a moment functor: d (B eta) / d eta = B
    B = rng.normal(size=(3, lay.V))
    mobj = vb.Objective(par, vb.LinearMoments(par, B=B))
    assert rel_err(mobj.fun_vector_jacobian(eta), B) < 1e-14
    assert rel_err(mobj.fun_free_jacobian(theta), B @ lay.jac(theta)) < 1e-13
