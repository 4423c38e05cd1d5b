"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
fp64 tolerance: 1e-11 relative to the largest entry of the expected result for sums over
observations (reduction order differs), 1e-9 for H^-1-type outputs."""
import os
import subprocess

import numpy as np
import pytest

from oracle import models as om
from oracle import solvers as osv
from helpers import make_par, glm_data, rel_err, LOSS_NAME, on_torch_stream

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-11


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1, 'no HIP device visible'
    return lrvb_amd


def test_mfma_f64_operand_maps():
    exe = os.path.join(ROOT, 'tools', 'mfma_f64_probe')
    if not os.path.exists(exe):
        pytest.skip('probe binary not built')
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr


MIXED = [('box', 'a', 3, -np.inf, np.inf), ('box', 'b', 2, 0.5, np.inf), ('box', 'c', 2, -np.inf, 3.0),
         ('box', 'd', 3, -2.0, 5.0), ('psd', 'm', 3, 0.3), ('simplex', 's', 2, 4)]


def test_packing_maps(vb):
    rng = np.random.default_rng(1)
    par, lay = make_par(vb, MIXED)
    ctx = vb.DeviceContext(par.layout_blocks(), quad_kind=1)
    theta = rng.normal(size=lay.D) * 0.7
    eta = lay.constrain(theta)
    assert rel_err(ctx.constrain(theta), eta) < 1e-14
    assert np.max(np.abs(ctx.unconstrain(eta) - theta)) < 1e-12
    assert rel_err(ctx.free_to_vector_jac(theta), lay.jac(theta)) < 1e-14
    # host-side packing classes agree with the device and the oracle
    par.set_free(theta)
    assert rel_err(par.get_vector(), eta) < 1e-14
    assert rel_err(np.asarray(par.free_to_vector_jac(theta).todense()), lay.jac(theta)) < 1e-14
    g = rng.normal(size=lay.V)
    Hv = rng.normal(size=(lay.V, lay.V)); Hv = Hv + Hv.T
    from oracle import packing as opk
    want = opk.convert_vector_to_free_hessian(lay, theta, g, Hv)
    assert rel_err(ctx.free_hessian_from_vector(theta, g, Hv), want) < 1e-13
    assert rel_err(np.asarray(vb.convert_vector_to_free_hessian(par, theta, g, Hv)), want) < 1e-13
    with pytest.raises(ValueError):
        ctx.constrain(theta[:-1])
    bad = eta.copy(); bad[3] = 0.1          # below lb = 0.5 of block b
    with pytest.raises(ValueError):
        ctx.unconstrain(bad)


def test_packing_maps_matrix_arrays(vb):
    """PosDefMatrixParamVector / PosDefMatrixParamArray (LRVB/MatrixParameters.py:201-481): one
    log-Cholesky block per matrix on the device, block-diagonal sparse derivatives on the host."""
    from oracle import packing as opk
    rng = np.random.default_rng(2)
    par = vb.ModelParamsDict('par')
    par.push_param(vb.VectorParam('v', 3, lb=0.0))
    par.push_param(vb.PosDefMatrixParamVector('pv', length=3, matrix_size=3, diag_lb=0.2))
    par.push_param(vb.PosDefMatrixParamArray('pa', array_shape=(2, 2), matrix_size=2))
    lay = opk.Layout([opk.box_block(3, 0.0, np.inf)] + [opk.psd_block(3, 0.2) for _ in range(3)]
                     + [opk.psd_block(2, 0.0) for _ in range(4)])
    assert par.free_size() == lay.D and par.vector_size() == lay.V
    ctx = vb.DeviceContext(par.layout_blocks(), quad_kind=1)
    theta = rng.normal(size=lay.D) * 0.5
    eta = lay.constrain(theta)
    par.set_free(theta)
    assert rel_err(par.get_vector(), eta) < 1e-14
    assert rel_err(ctx.constrain(theta), eta) < 1e-14
    assert np.max(np.abs(ctx.unconstrain(eta) - theta)) < 1e-12
    assert rel_err(ctx.free_to_vector_jac(theta), lay.jac(theta)) < 1e-14
    assert rel_err(np.asarray(par.free_to_vector_jac(theta).todense()), lay.jac(theta)) < 1e-14
    g = rng.normal(size=lay.V)
    Hv = rng.normal(size=(lay.V, lay.V)); Hv = Hv + Hv.T
    want = opk.convert_vector_to_free_hessian(lay, theta, g, Hv)
    assert rel_err(ctx.free_hessian_from_vector(theta, g, Hv), want) < 1e-13
    assert rel_err(np.asarray(vb.convert_vector_to_free_hessian(par, theta, g, Hv)), want) < 1e-13
    # the matrices themselves
    np.testing.assert_allclose(par['pv'].get()[1], opk.psd_matrix_from_vector(eta[3 + 6:3 + 12], 3), atol=1e-14)


@pytest.mark.parametrize('loss', [om.GAUSSIAN, om.LOGISTIC, om.POISSON])
@pytest.mark.parametrize('N,P', [(1, 3), (37, 5), (1000, 130), (4099, 254), (3000, 256), (2500, 300)])
def test_glm_box_layout(vb, loss, N, P):
    rng = np.random.default_rng(100 + N + P + loss)
    p1 = P // 3
    spec = [('box', 'u', p1, -np.inf, np.inf), ('box', 'pos', P - p1, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, loss)
    prior = 0.7
    m_centre = rng.normal(size=P) * 0.1
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], lik_info=1.3,
                             quad_A=np.full(P, prior), quad_m=m_centre, weights=w)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, lik_info=1.3, quad_A=np.full(P, prior), quad_m=m_centre)
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=lay.D) * 0.3
    v = rng.normal(size=lay.D)
    assert abs(obj.fun_free(theta) - model.value(theta)) <= 1e-12 * max(1.0, abs(model.value(theta)))
    assert rel_err(obj.fun_free_grad(theta), model.grad(theta)) < TOL
    Hw = model.hessian(theta)
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL          # before any build: the matrix-free pass over X
    H = obj.fun_free_hessian(theta)
    assert rel_err(H, Hw) < TOL
    assert np.array_equal(H, H.T)
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL          # after the build: against the resident Hessian
    # vector coordinates
    eta = lay.constrain(theta)
    assert rel_err(obj.fun_vector_grad(eta), model.grad_vec(eta)) < TOL
    assert rel_err(obj.fun_vector_hessian(eta), model.hessian_vec(eta)) < TOL
    assert rel_err(obj.fun_vector_hvp(eta, v), model.hessian_vec(eta) @ v) < TOL
    # per-observation gradient matrix and its Gram matrix
    G = fun.ctx.obs_grad(theta)
    Gw = model.obs_grad(theta)
    assert rel_err(G, Gw) < TOL
    assert rel_err(fun.gram(theta), Gw.T @ Gw) < TOL
    # side effect: par holds the evaluation point
    assert np.max(np.abs(par.get_free() - theta)) < 1e-12


def test_glm_general_layout(vb):
    rng = np.random.default_rng(7)
    spec = [('box', 'pre', 2, -np.inf, np.inf), ('box', 'beta', 6, -1.0, np.inf), ('psd', 'm', 3, 0.2), ('simplex', 's', 2, 3)]
    par, lay = make_par(vb, spec)
    N, P = 500, 6
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    A = rng.normal(size=(lay.V, lay.V)); A = A @ A.T / lay.V + np.eye(lay.V)
    m, b = rng.normal(size=lay.V) * 0.2, rng.normal(size=lay.V) * 0.3
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', glm_param='beta', quad_A=A, quad_m=m, quad_b=b, weights=w)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, glm_off=2, quad_A=A, quad_m=m, quad_b=b)
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=lay.D) * 0.4
    v = rng.normal(size=lay.D)
    Hw = model.hessian(theta)
    assert abs(obj.fun_free(theta) - model.value(theta)) < 1e-11 * abs(model.value(theta))
    assert rel_err(obj.fun_free_grad(theta), model.grad(theta)) < TOL
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL          # matrix-free (nothing built yet)
    assert rel_err(obj.fun_free_hessian(theta), Hw) < TOL
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL          # resident Hessian
    assert rel_err(fun.ctx.obs_grad(theta), model.obs_grad(theta)) < TOL
    assert rel_err(fun.gram(theta), model.gram(theta)) < TOL
    assert rel_err(fun.ctx.cross_hessian_tilt(theta), model.cross_hessian_tilt(theta)) < 1e-14


def test_reference_keyword_passthrough_known_answers(vb):
    """LRVB/test_objectives.py:161-243: f = sum(x^2) z y has Hessian 2 z y I, gradient 2 z y x,
    HVP 2 z y v; x16 under the preconditioner 4 I."""
    x = vb.VectorParam('x', size=2)
    fun = vb.QuadraticObjective(x, A=2.0 * np.ones(2), scale_fun=lambda y, z=1.: y * z)
    objective = vb.Objective(par=x, fun=fun)
    x_val = np.array([0., 1.])
    hvp_vec = np.array([2., 3.])
    np.testing.assert_array_almost_equal(1 * 2 * 1, objective.fun_free(x_val, 2))
    np.testing.assert_array_almost_equal(1 * 2 * 3, objective.fun_free(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(1 * 2 * 3, objective.fun_free(x_val, 2, z=3, verbose=True))
    np.testing.assert_array_almost_equal(1 * 2 * 3, objective.fun_vector(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * x_val * 2 * 3, objective.fun_free_grad(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * x_val * 2 * 3, objective.fun_vector_grad(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * x_val * 2 * 3, objective.fun_free_jacobian(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * np.eye(2) * 2 * 3, objective.fun_free_hessian(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * np.eye(2) * 2 * 3, objective.fun_vector_hessian(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * hvp_vec * 2 * 3, objective.fun_free_hvp(x_val, 2, hvp_vec, z=3))
    np.testing.assert_array_almost_equal(2 * hvp_vec * 2 * 3, objective.fun_vector_hvp(x_val, 2, hvp_vec, z=3))
    objective.preconditioner = 4 * np.eye(2)
    np.testing.assert_array_almost_equal(1 * 2 * 3 * 16, objective.fun_free_cond(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * x_val * 2 * 3 * 16, objective.fun_free_grad_cond(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * np.eye(2) * 2 * 3 * 16, objective.fun_free_hessian_cond(x_val, 2, z=3))
    np.testing.assert_array_almost_equal(2 * hvp_vec * 2 * 3 * 16, objective.fun_free_hvp_cond(x_val, 2, hvp_vec, z=3))


def test_cholesky_lrvb_cov_and_cg(vb):
    rng = np.random.default_rng(11)
    N, P = 3000, 200
    spec = [('box', 'beta', P, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, om.POISSON)
    fun = vb.GLMObjective(par, x, y, loss='poisson', prior_info=1.0, weights=w)
    model = om.DeclaredModel(lay, loss=om.POISSON, x=x, y=y, w=w, quad_A=np.ones(P))
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=P) * 0.2
    H = obj.fun_free_hessian(theta)
    Hw = model.hessian(theta)
    # Cholesky solve and LRVB covariance
    B = rng.normal(size=(P, 17))
    fun.ctx.chol_factor(H)
    assert rel_err(fun.ctx.chol_solve(B), np.linalg.solve(Hw, B)) < 1e-9
    M = rng.normal(size=(23, P))
    assert rel_err(fun.ctx.lrvb_cov(M), osv.lrvb_covariance(Hw, M)) < 1e-9
    with pytest.raises(np.linalg.LinAlgError):
        fun.ctx.chol_factor(-np.eye(P))
    # a pivot that fails in a LATER 64-column block (those blocks are factored by the head workgroup of the previous
    # step's trailing update) is reported with its position, and sizes that are not multiples of 64 keep their tail block
    for n, bad_at in ((200, 150), (130, 129), (64, 10), (193, 64)):
        A = rng.normal(size=(n, n))
        S = A @ A.T / n + np.eye(n)
        Lref = np.linalg.cholesky(S)
        fun.ctx.chol_factor(S)
        Bn = rng.normal(size=(n, 5))
        assert rel_err(fun.ctx.chol_solve(Bn), np.linalg.solve(S, Bn)) < 1e-10
        # make the leading minor of order bad_at + 1 singular-negative: subtract more than the pivot from that diagonal entry
        S2 = S.copy()
        S2[bad_at, bad_at] -= 1.5 * Lref[bad_at, bad_at] ** 2
        with pytest.raises(np.linalg.LinAlgError) as err:
            fun.ctx.chol_factor(S2)
        assert str(bad_at + 1) in str(err.value)
    # many right-hand sides: the first block steps have more tiles than the fused solve step takes (they go to the generic
    # GEMM + a head-only step), the last ones are fused -- both routes in one chain, forward and backward
    n, q = 1100, 2100
    A = rng.normal(size=(n, n))
    S = A @ A.T / n + np.eye(n)
    fun.ctx.chol_factor(S)
    Bn = rng.normal(size=(n, q))
    assert rel_err(fun.ctx.chol_solve(Bn), np.linalg.solve(S, Bn)) < 1e-10
    Mq = rng.normal(size=(q, n))
    assert rel_err(fun.ctx.lrvb_cov(Mq), Mq @ np.linalg.solve(S, Mq.T)) < 1e-10
    # CG: device loop vs Cholesky, the reference's own criterion (< 1e-8, test_objectives.py:552-554)
    solver = vb.ConjugateGradientSolver(obj.fun_free_hvp, theta)
    masks = vb.ConjugateGradient.get_masks(P, 40)
    vec = rng.normal(size=P)
    solver.get_hinv_vec_subsets(vec, masks)
    for rhs, sol, info in zip(solver.vecs, solver.hinv_vecs, solver.cg_infos):
        assert info == 0
        assert np.max(np.abs(sol - np.linalg.solve(Hw, rhs))) < 1e-8
    # preconditioned, with a warm start
    Minv = np.diag(1.0 / np.diag(Hw))
    xs, info, iters = fun.ctx.cg_solve(theta, vec, x0=0.1 * vec, Minv=Minv, tol=1e-10)
    assert info == 0 and np.max(np.abs(xs - np.linalg.solve(Hw, vec))) < 1e-8


def test_linear_response_quadratic_model(vb):
    """LRVB/test_model_sensitivity.py:36-88, 367-424: theta_hat(eps) = -A^-1 eps has the free
    form log(theta_hat + 10); dinput/dhyper equals its Jacobian."""
    dim = 3
    param = vb.VectorParam('theta', size=dim, lb=-10.0)
    vec = np.linspace(0.1, 0.3, num=dim)
    matrix = np.outer(vec, vec) + np.eye(dim)
    hyper0 = np.linspace(0.5, 10.0, num=dim)
    fun = vb.QuadraticObjective(param, A=matrix, b=hyper0)
    theta_opt = -np.linalg.solve(matrix, hyper0)
    theta0 = np.log(theta_opt + 10.0)
    sens = vb.ParametricSensitivityLinearApproximation(
        objective_functor=fun, input_par=param, hyper_par=fun.tilt_par, input_val0=theta0, hyper_val0=hyper0)
    # analytic Jacobian of log(-A^-1 eps + 10) w.r.t. eps
    want = np.diag(1.0 / (theta_opt + 10.0)) @ (-np.linalg.inv(matrix))
    np.testing.assert_array_almost_equal(want, sens.get_dinput_dhyper())
    eps = 0.01
    pred = sens.predict_input_par_from_hyperparameters(hyper0 + eps) - theta0
    true = np.log(-np.linalg.solve(matrix, hyper0 + eps) + 10.0) - theta0
    assert np.linalg.norm(true - pred) <= eps * np.linalg.norm(true)


def test_weight_sensitivity_example_shape(vb):
    """Example.ipynb:398-441 restated for a declared model: summary sensitivity operator
    -H^-1 M^T, cross Hessian w.r.t. the weights, influence matrix G H^-1 M^T."""
    rng = np.random.default_rng(5)
    N, P = 400, 4
    par, lay = make_par(vb, [('box', 'beta', P, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, om.GAUSSIAN, scale=1.0)
    fun = vb.GLMObjective(par, x, y, loss='gaussian', glm_param='beta', lik_info=2.0, prior_info=0.1)
    model = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=y, lik_info=2.0, quad_A=np.full(P, 0.1))
    theta = rng.normal(size=P) * 0.2
    w0 = np.ones(N)
    sens = vb.ParametricSensitivityLinearApproximation(
        objective_functor=fun, input_par=par, hyper_par=fun.weights_par, input_val0=theta, hyper_val0=w0)
    Hw = model.hessian(theta)
    Gw = model.obs_grad(theta)
    assert rel_err(sens.get_dinput_dhyper(), -np.linalg.solve(Hw, Gw.T)) < 1e-9
    summary = vb.LinearMoments(par, select='beta')
    M = vb.Objective(par, summary).fun_free_jacobian(theta)
    assert rel_err(M, lay.jac(theta)) < 1e-14
    assert rel_err(sens.get_lrvb_cov(M), osv.lrvb_covariance(Hw, lay.jac(theta))) < 1e-9


def test_errors_and_loud_failures(vb):
    par, lay = make_par(vb, [('box', 'beta', 4, -np.inf, np.inf)])
    rng = np.random.default_rng(3)
    x, y, w = glm_data(rng, 10, 4, om.GAUSSIAN)
    fun = vb.GLMObjective(par, x, y)
    obj = vb.Objective(par, fun)
    with pytest.raises(ValueError):
        obj.fun_free_hessian(np.zeros(5))
    with pytest.raises(ValueError):
        fun.ctx.set_weights(np.ones(9))
    with pytest.raises(AssertionError):
        obj.fun_free_hessian_cond(np.zeros(4))
    # an opaque closure: small D goes through the host's difference fallback (tests/test_host_logic.py pins it with the
    # reference's known answers), a large one is refused -- there is no silent O(D^2)-evaluations route
    closure_obj = vb.Objective(par, lambda: float(np.sum(par.get_vector() ** 2)))
    assert closure_obj.fun_free(np.ones(4)) == 4.0
    np.testing.assert_allclose(closure_obj.fun_free_hessian(np.ones(4)), 2.0 * np.eye(4), atol=1e-8)
    big = vb.VectorParam('big', size=200)
    with pytest.raises(NotImplementedError):
        vb.Objective(big, lambda: float(np.sum(big.get() ** 2))).fun_free_hessian(np.ones(200))


def test_sharded_engine_single_rank_matches_direct_build(vb):
    """The multi-GPU code path (partial statistics -> [all-reduce] -> finish) on one rank equals
    the fused single-GPU entry point and the oracle; also pins the statistics layout."""
    import torch
    from lrvb_amd.distributed import ShardedHessian, DeviceEngine, unpack_tiles, stats_layout
    rng = np.random.default_rng(21)
    N, P = 5000, 300
    spec = [('box', 'u', 200, -np.inf, np.inf), ('box', 'pos', 100, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    fun = vb.GLMObjective(par, x, y, loss='logistic', prior_info=0.3, weights=w)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.3))
    theta = rng.normal(size=P) * 0.2
    dev = torch.device('cuda', 0)
    th = torch.tensor(theta, device=dev)
    H = ShardedHessian(DeviceEngine(fun.ctx, dev)).build(th).cpu().numpy()
    Hw = model.hessian(theta)
    assert rel_err(H, Hw) < TOL
    H2 = torch.empty((P, P), dtype=torch.float64, device=dev)
    on_torch_stream(fun.ctx, dev)
    fun.ctx.hessian_dev(th.data_ptr(), H2.data_ptr(), P); fun.ctx.sync()
    assert np.array_equal(H2.cpu().numpy(), H)          # same kernels, same order: bitwise equal
    eng = DeviceEngine(fun.ctx, dev)
    st = eng.partial(th).cpu().numpy()
    fun.ctx.set_stream(None)
    o_val, o_g, o_t, total = stats_layout(P)
    assert st.size == total == fun.ctx.stats_size()
    eta = lay.constrain(theta)
    z = x @ eta
    l0, l1, l2 = om.loss_terms(om.LOGISTIC, y, z)
    assert abs(st[o_val] - np.sum(w * l0)) < 1e-11 * abs(np.sum(w * l0))
    assert rel_err(st[o_g:o_t], x.T @ (w * l1)) < TOL
    assert rel_err(unpack_tiles(st[o_t:], P), x.T @ ((w * l2)[:, None] * x)) < TOL


@pytest.mark.parametrize('N,P', [(5000, 300), (777, 130), (40000, 1024)])
def test_gaussian_build_reads_x_once(vb, N, P):
    """Gaussian loss: the curvature w tau does not depend on theta, so the build forms the gradient of the data term
    from the SYRK's own sums, d f / d beta = S beta - X^T (c o y) (the column sums ride on the diagonal tiles), and runs
    NO separate pass over X.  The statistics buffer [value | gradient | tiles] must equal the oracle's sums, the pass
    route (forced through the register-staged kernel, tuning bit 0) must give the same Hessian, and the profile must
    show zero pass launches for the Gaussian build and one for a logistic build."""
    import torch
    from lrvb_amd.distributed import DeviceEngine, unpack_tiles, stats_layout
    rng = np.random.default_rng(N + P)
    p1 = P // 3
    spec = [('box', 'u', P - p1, -np.inf, np.inf), ('box', 'pos', p1, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, om.GAUSSIAN)
    y = x @ rng.normal(size=P) + 0.3 * rng.normal(size=N)          # a response the model fits: S beta and r nearly cancel
    fun = vb.GLMObjective(par, x, y, loss='gaussian', lik_info=1.7, prior_info=0.3, weights=w)
    model = om.DeclaredModel(lay, loss=om.GAUSSIAN, x=x, y=y, w=w, lik_info=1.7, quad_A=np.full(P, 0.3))
    theta = rng.normal(size=P) * 0.2
    dev = torch.device('cuda', 0)
    th = torch.tensor(theta, device=dev)
    ctx = fun.ctx
    ctx.profile_enable(True); ctx.profile_reset()
    st = DeviceEngine(ctx, dev).partial(th).cpu().numpy()
    ctx.set_stream(None)
    assert ctx.profile_get()['pass_calls'] == 0 and ctx.profile_get()['wsyrk_calls'] == 1
    o_val, o_g, o_t, total = stats_layout(P)
    eta = lay.constrain(theta)
    l0, l1, l2 = om.loss_terms(om.GAUSSIAN, y, x @ eta, 1.7)
    S = x.T @ ((w * l2)[:, None] * x)
    assert rel_err(unpack_tiles(st[o_t:], P), S) < TOL
    # the gradient is a difference of two O(N) sums: tolerance relative to the sums, not to the (small) difference
    scale = max(np.max(np.abs(S @ eta)), np.max(np.abs(x.T @ (w * 1.7 * y))))
    assert np.max(np.abs(st[o_g:o_t] - x.T @ (w * l1))) < 1e-12 * scale
    assert abs(st[o_val] - np.sum(w * l0)) < 1e-11 * np.sum(w * 1.7 * y * y)
    H = ctx.hessian(theta)
    assert rel_err(H, model.hessian(theta)) < TOL
    ctx.set_tuning(0, 1)                                            # register-staged SYRK: the pass route
    ctx.profile_reset()
    H_pass = ctx.hessian(theta)
    assert ctx.profile_get()['pass_calls'] == 1
    ctx.set_tuning(0, 0)
    ctx.profile_enable(False)
    assert rel_err(H, H_pass) < 1e-12
    obj = vb.Objective(par, fun)
    assert rel_err(obj.fun_free_grad(theta), model.grad(theta)) < TOL     # the gradient entry point still runs the pass


def test_full_size_properties_headline_shape(vb):
    """Size-independent properties at the BASELINE.json shape (D = 1024) with N reduced to what the
    oracle-free checks need: symmetry, linearity in the weights, H v == HVP, G^T G PSD, and the
    registered-staged (generic) kernel agreeing with the LDS-DMA kernel."""
    import torch
    rng = np.random.default_rng(31)
    N, P = 200000, 1024
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
    yv = torch.randn((N,), dtype=torch.float64, device=dev, generator=g)
    w1 = torch.rand((N,), dtype=torch.float64, device=dev, generator=g) + 0.5
    w2 = torch.rand((N,), dtype=torch.float64, device=dev, generator=g) + 0.5
    blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]
    ctx = on_torch_stream(vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, lik_info=1.5, quad_kind=0), dev)
    ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, yv.data_ptr(), N, 1)
    theta = torch.zeros((P,), dtype=torch.float64, device=dev)
    def build(wt):
        H = torch.empty((P, P), dtype=torch.float64, device=dev)
        ctx.set_weights_dev(wt.data_ptr(), N)
        ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), P); ctx.sync()
        return H
    H1, H2, H12 = build(w1), build(w2), build(w1 + w2)
    scale = H12.abs().max().item()
    assert torch.equal(H1, H1.T)
    assert (H1 + H2 - H12).abs().max().item() < 1e-12 * scale            # linear in the weights
    ref = 1.5 * (X[:, :96].T * w1) @ X[:, 900:1024]                        # an off-diagonal block, torch fp64
    assert (H1[:96, 900:1024] - ref).abs().max().item() < 1e-12 * scale
    v = torch.randn((P,), dtype=torch.float64, device=dev, generator=g)
    out = torch.empty_like(v)
    ctx.set_weights_dev(w1.data_ptr(), N)
    ctx.hvp_dev(theta.data_ptr(), v.data_ptr(), out.data_ptr()); ctx.sync()
    assert (out - H1 @ v).abs().max().item() < 1e-11 * (H1 @ v).abs().max().item()
    ctx.set_tuning(0, 1)                                                  # force the register-staged kernel
    Hg = build(w1)
    ctx.set_tuning(0, 0)
    assert (Hg - H1).abs().max().item() < 1e-12 * scale
    # only the four documented bits exist; anything else is refused and changes nothing
    for bad in (16, 1 << 8, 7 << 8, 1 << 16, -1):
        with pytest.raises(ValueError):
            ctx.set_tuning(0, bad)
    assert torch.equal(build(w1), H1)
    G = torch.empty((P, P), dtype=torch.float64, device=dev)
    ctx.gram_dev(theta.data_ptr(), G.data_ptr(), P); ctx.sync()
    assert torch.equal(G, G.T) and torch.linalg.eigvalsh(G).min().item() > -1e-9 * G.abs().max().item()


def test_repeated_hvp_at_one_point(vb):
    """Host-callback optimisers (scipy's cg / trust-ncg, as the reference uses them) call fun_free_hvp many times at
    one point; the library keeps that point's state between consecutive calls.  Every way of changing the
    objective or the point in between must be seen."""
    rng = np.random.default_rng(77)
    spec = [('box', 'u', 9, -np.inf, np.inf), ('box', 'pos', 7, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    N, P = 333, 16
    x, y, w = glm_data(rng, N, P, om.POISSON)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='poisson', quad_A=np.full(P, 0.6), weights=w)
    model = om.DeclaredModel(lay, loss=om.POISSON, x=x, y=y, w=w, quad_A=np.full(P, 0.6))
    obj = vb.Objective(par, fun)
    th1, th2 = rng.normal(size=P) * 0.2, rng.normal(size=P) * 0.2
    vs = rng.normal(size=(4, P))
    H1, H2 = model.hessian(th1), model.hessian(th2)
    for v in vs:                                                  # first call prepares, the others reuse
        assert rel_err(obj.fun_free_hvp(th1, v), H1 @ v) < TOL
    assert rel_err(obj.fun_free_hvp(th2, vs[0]), H2 @ vs[0]) < TOL         # another point
    assert rel_err(obj.fun_free_hvp(th1, vs[1]), H1 @ vs[1]) < TOL         # and back
    # other entry points in between (they overwrite the per-observation scratch) invalidate the state
    obj.fun_free_hvp(th1, vs[0]); fun.gram(th2); obj.fun_free_grad(th2)
    assert rel_err(obj.fun_free_hvp(th1, vs[2]), H1 @ vs[2]) < TOL
    # value, gradient and products at one point share one state, in every order (scipy: fun(x), jac(x), hessp(x, .))
    assert abs(obj.fun_free(th2) - model.value(th2)) <= 1e-12 * abs(model.value(th2))
    assert rel_err(obj.fun_free_grad(th2), model.grad(th2)) < TOL
    assert rel_err(obj.fun_free_hvp(th2, vs[3]), H2 @ vs[3]) < TOL
    assert rel_err(obj.fun_free_grad(th2), model.grad(th2)) < TOL
    assert abs(obj.fun_free(th2) - model.value(th2)) <= 1e-12 * abs(model.value(th2))
    assert rel_err(obj.fun_free_grad(th1), model.grad(th1)) < TOL
    assert rel_err(obj.fun_free_hvp(th1, vs[0]), H1 @ vs[0]) < TOL
    # new weights at the same point
    obj.fun_free_hvp(th1, vs[0])
    w2 = w * rng.uniform(0.5, 1.5, N)
    fun.weights_par.set_vector(w2); model.w = w2
    assert rel_err(obj.fun_free_hvp(th1, vs[3]), model.hessian(th1) @ vs[3]) < TOL
    # vector coordinates keep their own state; switching coordinate systems at numerically equal input is seen
    eta = lay.constrain(th1)
    Hv = model.hessian_vec(eta)
    assert rel_err(obj.fun_vector_hvp(eta, vs[0]), Hv @ vs[0]) < TOL
    assert rel_err(obj.fun_vector_hvp(eta, vs[1]), Hv @ vs[1]) < TOL
    assert rel_err(obj.fun_free_hvp(eta, vs[1]), model.hessian(eta) @ vs[1]) < TOL
    # scipy's cg over a LinearOperator of fun_free_hvp at a fixed point (LRVB/ConjugateGradient.py:63-85)
    import scipy.sparse.linalg as sla
    op = sla.LinearOperator((P, P), matvec=lambda v: obj.fun_free_hvp(th1, v))
    b = rng.normal(size=P)
    sol, info = sla.cg(op, b, rtol=1e-10, atol=0.0)
    assert info == 0 and rel_err(sol, np.linalg.solve(model.hessian(th1), b)) < 1e-8


def test_more_than_2_31_matrix_elements(vb):
    """Maximum sizes: N x D = 2.2e6 x 1024 = 2.25e9 doubles (18 GB) -- element offsets no longer fit in
    32 bits.  Size-independent checks: additivity over a row split (each half indexes below 2^31), the last
    rows of the per-observation gradient against torch, H v == HVP, and the blocked CG against a dense solve."""
    import torch
    N, P = 2_200_000, 1024
    assert N * P > 2 ** 31
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(77)
    X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
    beta = torch.randn((P,), dtype=torch.float64, device=dev, generator=g)
    yv = (torch.sigmoid(X @ beta) > torch.rand((N,), dtype=torch.float64, device=dev, generator=g)).double()
    w = torch.rand((N,), dtype=torch.float64, device=dev, generator=g) + 0.5
    theta = 0.3 * torch.randn((P,), dtype=torch.float64, device=dev, generator=g)
    blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]

    def make(r0, r1, prior):
        ctx = on_torch_stream(vb.DeviceContext(blocks, loss='logistic', n_obs=r1 - r0, n_cols=P, quad_kind=vb._hip.QUAD_DIAG), dev)
        ctx.set_data_dev(0, X[r0:r1].data_ptr(), r1 - r0, P)
        ctx.set_data_dev(1, yv[r0:r1].data_ptr(), r1 - r0, 1)
        ctx.set_weights_dev(w[r0:r1].data_ptr(), r1 - r0)
        ctx.set_data(vb._hip.SLOT_QUAD_A, np.full(P, prior))
        return ctx

    def hessian(ctx):
        H = torch.empty((P, P), dtype=torch.float64, device=dev)
        ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), P); ctx.sync()
        return H

    n1 = 1_100_003
    full, top, bottom = make(0, N, 1.0), make(0, n1, 1.0), make(n1, N, 0.0)
    H = hessian(full)
    Hsum = hessian(top) + hessian(bottom)
    scale = H.abs().max().item()
    assert torch.equal(H, H.T)
    assert (H - Hsum).abs().max().item() < 1e-12 * scale
    th = theta.cpu().numpy()
    v_full, v_parts = full.value(th), top.value(th) + bottom.value(th)
    assert abs(v_full - v_parts) < 1e-12 * abs(v_full)
    g_full, g_parts = full.grad(th), top.grad(th) + bottom.grad(th)
    assert rel_err(g_full, g_parts) < 1e-12
    # rows at the far end of the matrix (element offsets above 2^31)
    z = X[N - 7:] @ theta
    l1 = torch.sigmoid(z) - yv[N - 7:]                     # d2 f / d theta d w_n carries no weight
    assert rel_err(full.obs_grad(th, N - 7, N), (l1[:, None] * X[N - 7:]).cpu().numpy()) < 1e-12
    # Hessian-vector products, one vector and a block of them through CG
    vv = torch.randn((P,), dtype=torch.float64, device=dev, generator=g)
    out = torch.empty_like(vv)
    full.hvp_dev(theta.data_ptr(), vv.data_ptr(), out.data_ptr()); full.sync()
    assert (out - H @ vv).abs().max().item() < 1e-11 * (H @ vv).abs().max().item()
    B = torch.randn((3, P), dtype=torch.float64, device=dev, generator=g)
    Xs, info, iters = full.cg_solve_multi(th, B.cpu().numpy(), tol=1e-10)
    assert np.all(info == 0)
    want = torch.linalg.solve(H, B.T).T.cpu().numpy()
    assert rel_err(Xs, want) < 1e-8


@pytest.mark.parametrize('N,P', [(1, 1), (3, 2), (15, 7), (16, 16), (17, 33), (100, 64), (129, 65), (1000, 127), (700, 129)])
def test_edge_shapes(vb, N, P):
    """Ragged sizes around every internal granule: 16-row stages, 32/64-column narrow kernels, the
    128-column tile edge, odd P (unaligned rows -> generic register-staged kernel)."""
    rng = np.random.default_rng(1000 * N + P)
    par, lay = make_par(vb, [('box', 'b', P, -1.0, np.inf)])
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    fun = vb.GLMObjective(par, x, y, loss='logistic', prior_info=0.5, weights=w)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.5))
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=P) * 0.3
    Hw = model.hessian(theta)
    assert rel_err(obj.fun_free_hessian(theta), Hw) < TOL
    assert rel_err(obj.fun_free_grad(theta), model.grad(theta)) < TOL
    v = rng.normal(size=P)
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL
    assert rel_err(fun.gram(theta), model.gram(theta)) < TOL
    # zero weights on some rows: those observations drop out exactly
    w2 = w.copy(); w2[::2] = 0.0
    fun.weights_par.set_vector(w2)
    model.w = w2
    assert rel_err(obj.fun_free_hessian(theta), model.hessian(theta)) < TOL


@pytest.mark.parametrize('loss,N,P', [(om.GAUSSIAN, 300, 1030), (om.POISSON, 2051, 1153), (om.LOGISTIC, 1000, 2048),
                                      (om.LOGISTIC, 777, 3000), (om.GAUSSIAN, 5, 4096), (om.POISSON, 1, 2050), (om.POISSON, 40, 4100)])
def test_wide_designs(vb, loss, N, P):
    """n_cols > 1024: the row no longer fits in one wavefront's registers.  Up to 4096 columns the four waves of a workgroup
    hold a row between them and X is still read ONCE (round 4); wider still, the two-pass route (row dots, then column
    accumulation).  Both must give the same value, gradient, Hessian, HVP and CG -- and the same as each other (tuning bit 0
    sends every wide design down the two-pass route)."""
    rng = np.random.default_rng(P)
    p1 = P // 2
    par, lay = make_par(vb, [('box', 'u', p1, -np.inf, np.inf), ('box', 'pos', P - p1, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, loss)
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], lik_info=1.3, quad_A=np.full(P, 0.7), weights=w)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, lik_info=1.3, quad_A=np.full(P, 0.7))
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=P) * 0.1
    v = rng.normal(size=P)
    assert abs(obj.fun_free(theta) - model.value(theta)) <= 1e-12 * max(1.0, abs(model.value(theta)))
    assert rel_err(obj.fun_free_grad(theta), model.grad(theta)) < TOL
    Hw = model.hessian(theta)
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL              # matrix-free: before the build leaves its matrix resident
    g_one = obj.fun_free_grad(theta)
    fun.ctx.set_tuning(0, 1)                                             # the two-pass route for the same design
    assert rel_err(obj.fun_free_grad(theta), g_one) < 1e-12 and rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL
    fun.ctx.set_tuning(0, 0)
    assert rel_err(obj.fun_free_hessian(theta), Hw) < TOL
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL
    eta = lay.constrain(theta)
    assert rel_err(obj.fun_vector_hvp(eta, v), model.hessian_vec(eta) @ v) < TOL
    assert rel_err(fun.ctx.obs_grad(theta, 0, min(N, 50)), model.obs_grad(theta, 0, min(N, 50))) < TOL
    # CG needs a positive definite Hessian: the same wide design with every coefficient unconstrained, where
    # H = X^T diag(w loss'') X + prior precision is positive definite by construction at any point
    par_u, lay_u = make_par(vb, [('box', 'u', P, -np.inf, np.inf)])
    fun_u = vb.DeviceObjective(par_u, x=x, y=y, loss=LOSS_NAME[loss], lik_info=1.3, quad_A=np.full(P, 0.7), weights=w)
    model_u = om.DeclaredModel(lay_u, loss=loss, x=x, y=y, w=w, lik_info=1.3, quad_A=np.full(P, 0.7))
    fun_u._push_state()
    H_u = model_u.hessian(theta)
    assert np.min(np.linalg.eigvalsh(H_u)) > 0
    B = rng.normal(size=(3, P))
    X, info, _ = fun_u.ctx.cg_solve_multi(theta, B)
    assert np.all(info == 0) and rel_err(X, np.linalg.solve(H_u, B.T).T) < 1e-6


def test_nan_and_overflow_inputs_do_not_fault(vb):
    """An optimiser's stray probe: nan / overflowing free values give nan results (as the reference's numpy
    closure does) or the documented exceptions -- never a device fault or a hang."""
    rng = np.random.default_rng(0)
    N, P = 1000, 256
    par, lay = make_par(vb, [('box', 'b', P, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, om.POISSON)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='poisson', quad_A=np.ones(P), weights=w)
    obj = vb.Objective(par, fun)
    th = rng.normal(size=P) * 0.1
    th[3] = np.nan
    assert np.isnan(obj.fun_free(th))
    assert np.isnan(obj.fun_free_grad(th)).any()
    H = obj.fun_free_hessian(th)
    assert np.isnan(H).any()
    with pytest.raises(np.linalg.LinAlgError):
        fun.ctx.chol_factor(H)
    X, info, it = fun.ctx.cg_solve_multi(th, rng.normal(size=(3, P)), maxiter=5)
    assert np.all(info == 5)                                 # not converged is reported, not raised (ConjugateGradient.py:82-85)
    th2 = rng.normal(size=P) * 0.1
    th2[5] = 800.0                                           # exp overflows in the box map
    with np.errstate(over='ignore', invalid='ignore'):
        assert not np.isfinite(obj.fun_free(th2))
    good = rng.normal(size=P) * 0.1                          # the context is still usable afterwards
    assert np.isfinite(obj.fun_free(good)) and np.all(np.isfinite(obj.fun_free_hessian(good)))


def test_device_hessian_assembly_from_kronecker_blocks(vb):
    """lrvb_hvec_*: coef * D^T (A (x) B) D blocks written by the device against dense duplication-matrix
    algebra, mirrored off-diagonal blocks, and the free-coordinate conversion of the assembled matrix."""
    from lrvb_amd.quadform import duplication_matrix
    from oracle import packing as opk
    rng = np.random.default_rng(4)
    k1, k2 = 5, 5
    par = vb.ModelParamsDict('p')
    par.push_param(vb.VectorParam('v', 3, lb=0.0))
    par.push_param(vb.PosDefMatrixParam('a', k1))
    par.push_param(vb.PosDefMatrixParam('b', k2, diag_lb=0.1))
    lay = opk.Layout([opk.box_block(3, 0.0, np.inf), opk.psd_block(k1), opk.psd_block(k2, 0.1)])
    ctx = vb.DeviceContext(par.layout_blocks(), quad_kind=1)
    V = lay.V
    m1 = k1 * (k1 + 1) // 2
    A, B, C = (rng.normal(size=(k1, k1)) for _ in range(3))
    Dm = duplication_matrix(k1)
    dense = rng.normal(size=(3, 3)); dense = dense + dense.T
    col = rng.normal(size=(3, m1))
    want = np.zeros((V, V))
    want[:3, :3] += dense
    want[:3, 3:3 + m1] += col; want[3:3 + m1, :3] += col.T
    blk_aa = 0.7 * Dm.T @ np.kron(A, A.T) @ Dm
    want[3:3 + m1, 3:3 + m1] += blk_aa
    blk_ab = -1.3 * Dm.T @ np.kron(B, C) @ Dm
    want[3:3 + m1, 3 + m1:] += blk_ab; want[3 + m1:, 3:3 + m1] += blk_ab.T
    ctx.hvec_begin()
    ctx.hvec_add_block(dense, 0, 0)
    ctx.hvec_add_block(col, 0, 3, mirror=True)
    ctx.hvec_add_symkron(A, A.T, 0.7, 3, 3)
    ctx.hvec_add_symkron(B, C, -1.3, 3, 3 + m1, mirror=True)
    theta = rng.normal(size=lay.D) * 0.4
    Hv = ctx.hvec_finish(lay.constrain(theta), np.zeros(V), is_free=False)
    assert rel_err(Hv, want) < 1e-13
    # same blocks again, free coordinates (the assembled matrix must be symmetric for the conversion)
    S = rng.normal(size=(k1, k1)); S = S + S.T
    g = rng.normal(size=V)
    wantS = np.zeros((V, V)); wantS[:3, :3] = dense
    wantS[3:3 + m1, 3:3 + m1] = Dm.T @ np.kron(S, S) @ Dm
    ctx.hvec_begin()
    ctx.hvec_add_block(dense, 0, 0)
    ctx.hvec_add_symkron(S, S, 1.0, 3, 3)
    Hf = ctx.hvec_finish(theta, g, is_free=True)
    assert rel_err(Hf, opk.convert_vector_to_free_hessian(lay, theta, g, wantS)) < 1e-12
    # misuse is reported
    with pytest.raises(RuntimeError):
        ctx.hvec_add_block(dense, 0, 0)                      # no begin
    ctx.hvec_begin()
    with pytest.raises(ValueError):
        ctx.hvec_add_symkron(A, A, 1.0, V - 2, 0)            # does not fit
    with pytest.raises(ValueError):
        ctx.hvec_add_block(col, 3, 3, mirror=True)           # mirrored block on the diagonal
    ctx.hvec_finish(theta, g, is_free=True, want_host=False)


def test_hvec_program_equals_the_call_per_block_route(vb):
    """`lrvb_hvec_program` (what the Python assembly sends: every block recorded, ONE call) against the begin / add / finish
    entry points of the C ABI: dense, mirrored, indexed and Kronecker blocks, an operand shared by two records, both
    coordinate systems."""
    rng = np.random.default_rng(11)
    k = 6
    spec = [('box', 'a', 5, -np.inf, np.inf), ('psd', 'S', k, 0.0), ('box', 'pos', 4, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    ctx = vb.DeviceContext(par.layout_blocks(), quad_kind=1)
    V, D = ctx.V, ctx.D
    m = k * (k + 1) // 2
    A = rng.normal(size=(k, k)); A = A @ A.T
    B = rng.normal(size=(k, k)); B = B @ B.T
    blk = rng.normal(size=(5, 5)); blk = blk + blk.T
    off = rng.normal(size=(5, 4))
    rows = np.array([0, 2, 5 + m, 5 + m + 3])
    sc = rng.normal(size=(4, 4)); sc = sc + sc.T
    theta = rng.normal(size=D) * 0.3
    g = rng.normal(size=V)

    def assemble(immediate, is_free):
        ctx.hvec_begin(immediate=immediate)
        ctx.hvec_add_block(blk, 0, 0)
        ctx.hvec_add_block(off, 0, 5 + m, mirror=True)
        ctx.hvec_add_indexed(sc, rows, rows)
        ctx.hvec_add_symkron(A, B, 0.5, 5, 5)
        ctx.hvec_add_symkron(B, A, 0.5, 5, 5)               # A and B are sent once
        ctx.hvec_add_symkron(A, A, -0.25, 5, 5)
        return ctx.hvec_finish(theta if is_free else np.zeros(V), g, is_free)

    for is_free in (False, True):
        H_prog, H_imm = assemble(False, is_free), assemble(True, is_free)
        assert np.array_equal(H_prog, H_imm)
        assert np.allclose(H_prog, H_prog.T, rtol=0, atol=1e-12 * np.abs(H_prog).max())
    # operands are copied when they are recorded: a caller that reuses ONE scratch array for several blocks gets each block's
    # contents at the time of its call, as with the call-per-block entry points (advisor finding, round 3)
    def assemble_with_scratch(immediate):
        scratch = np.empty((5, 5))
        ctx.hvec_begin(immediate=immediate)
        scratch[:] = blk
        ctx.hvec_add_block(scratch, 0, 0)
        scratch[:] = 2.0 * blk                                 # overwritten before finish
        ctx.hvec_add_block(scratch[:, :4].copy(), 0, 5 + m, mirror=True)
        scratch[:] = -7.0
        return ctx.hvec_finish(np.zeros(V), g, False)
    assert np.array_equal(assemble_with_scratch(False), assemble_with_scratch(True))
    # an index list that names an element twice is refused (the device adds the entries in parallel)
    ctx.hvec_begin()
    with pytest.raises(ValueError):
        ctx.hvec_add_indexed(sc, np.array([0, 2, 2, 3]), rows)
    # a block that does not fit is refused when it is recorded (and the library checks every record again before it launches anything)
    ctx.hvec_begin()
    with pytest.raises(ValueError):
        ctx.hvec_add_block(blk, V - 2, 0)
    ops = np.array([[0, 0, 5, 5, V - 2, 0, 0, 0]], dtype=np.int64)
    import ctypes
    st = ctx._lib.lrvb_hvec_program(ctx._h, ops.ctypes.data_as(ctypes.c_void_p), 1, vb._hip.ptr(blk.ravel().copy()), 25,
                                    vb._hip.ptr(np.zeros(V)), V, 0, vb._hip.ptr(g), None)
    assert st != 0


@pytest.mark.parametrize('k1,k2', [(1, 2), (5, 17), (33, 16), (63, 3)])
def test_structured_jacobian_products(vb, k1, k2):
    """Layouts of box and log-Cholesky blocks convert to free coordinates without the dense Jacobian (`jt_apply_kernel`:
    J^T A as MFMA tiles over gathered rows): J^T B against the oracle's dense Jacobian, and the Hessian
    J^T H_vec J + sum_k g_k d2 eta_k (LRVB/Parameters.py:397-424) of a model with a data term and a dense quadratic term,
    for block orders from 1 to 63 (the largest the kernel's LDS holds), two blocks per layout, bounded and unbounded boxes."""
    rng = np.random.default_rng(100 + k1)
    spec = [('box', 'pre', 3, 0.0, np.inf), ('psd', 'm1', k1, 0.0), ('box', 'beta', 5, -1.0, 2.0), ('psd', 'm2', k2, 0.3), ('box', 'post', 2, -np.inf, np.inf)]
    par, lay = make_par(vb, spec)
    N, P = 300, 5
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    A = rng.normal(size=(lay.V, lay.V)); A = A @ A.T / lay.V + np.eye(lay.V)
    m, b = rng.normal(size=lay.V) * 0.2, rng.normal(size=lay.V) * 0.3
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', glm_param='beta', quad_A=A, quad_m=m, quad_b=b, weights=w)
    glm_off = 3 + k1 * (k1 + 1) // 2
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, glm_off=glm_off, quad_A=A, quad_m=m, quad_b=b)
    theta = rng.normal(size=lay.D) * 0.3
    J = lay.jac(theta)
    for Q in (1, 70):
        B = rng.normal(size=(lay.V, Q))
        assert rel_err(fun.ctx.jac_t_matmul(theta, B), J.T @ B) < 1e-13
    obj = vb.Objective(par, fun)
    Hw = model.hessian(theta)
    assert rel_err(obj.fun_free_grad(theta), model.grad(theta)) < TOL
    assert rel_err(obj.fun_free_hessian(theta), Hw) < TOL
    v = rng.normal(size=lay.D)
    assert rel_err(obj.fun_free_hvp(theta, v), Hw @ v) < TOL
    assert rel_err(fun.gram(theta), model.gram(theta)) < TOL


@pytest.mark.parametrize('N', [1, 5, 16, 17, 33, 1000, 40003])
@pytest.mark.parametrize('P', [1, 2, 3, 22, 31, 32, 33, 48, 63, 64])
def test_narrow_gram_kernel_edge_shapes(vb, N, P):
    """`lrvb_weighted_gram_sum` (S = X^T diag(w) X and the sum of the weights) on the narrow Gram kernel for every corner of
    its shape handling: fewer rows than a 16-row stage, a ragged last stage, odd and even widths (8- / 16-byte loads), the
    one- and the two-pair instantiation, the weights read in place without padding; against numpy."""
    rng = np.random.default_rng(1000 * P + N)
    x = rng.normal(size=(N, P))
    w = rng.uniform(0.5, 1.5, N)
    blocks = [dict(kind=0, free_size=3, vec_size=3, dim0=3, dim1=0, lb=-np.inf, ub=np.inf)]
    ctx = vb.DeviceContext(blocks, loss='data_only', n_obs=N, n_cols=P, device=0)
    ctx.set_data(vb._hip.SLOT_X, x)
    ctx.set_weights(w)
    S, W = ctx.weighted_gram(with_sum=True)
    want = x.T @ (w[:, None] * x)
    assert np.max(np.abs(S - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))
    assert abs(W - w.sum()) < 1e-12 * w.sum()
    assert np.array_equal(S, S.T)
