"""the optimiser loop on the device.  `minimize_objective_trust_ncg(...,
on_device=True)` must walk the same iterates as the reference's route -- scipy's trust-ncg driving
fun_free / fun_free_grad / fun_free_hvp (LRVB/OptimizationUtils.py:44-75; with a preconditioner the `_cond`
family, LRVB/SparseObjectives.py:202-240) -- and end at a point where the ORACLE's gradient vanishes.
The reference's own check of this wrapper (LRVB/test_objectives.py:400-445) is the same: both routes reach the
known optimum of a quadratic model."""
import numpy as np
import pytest
import scipy.optimize

from oracle import models as om
from helpers import make_par, glm_data, rel_err, LOSS_NAME

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def build(vb, loss, N, spec, seed, prior=0.7, lik_info=1.0):
    rng = np.random.default_rng(seed)
    par, lay = make_par(vb, spec)
    P = lay.V
    x, y, w = glm_data(rng, N, P, loss)
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], lik_info=lik_info, quad_A=np.full(P, prior), weights=w)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, lik_info=lik_info, quad_A=np.full(P, prior))
    return vb.Objective(par, fun), model, rng


@pytest.mark.parametrize('loss,N,spec', [
    (om.GAUSSIAN, 400, [('box', 'b', 12, -np.inf, np.inf)]),
    (om.LOGISTIC, 900, [('box', 'u', 20, -np.inf, np.inf), ('box', 'pos', 12, 0.0, np.inf)]),
    (om.POISSON, 700, [('box', 'lo', 9, -1.0, np.inf), ('box', 'both', 7, -2.0, 3.0)]),
])
def test_device_trust_ncg_walks_the_scipy_iterates(vb, loss, N, spec):
    obj, model, rng = build(vb, loss, N, spec, seed=N)
    D = model.layout.D
    x0 = rng.normal(size=D) * 0.3
    x_host, res_host = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=200, gtol=1e-6, disp=False)
    x_dev, res_dev = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=200, gtol=1e-6, disp=False,
                                                                       on_device=True)
    assert res_host.success and res_dev.success and res_dev.status == 0
    assert res_dev.nit == res_host.nit                       # same algorithm, same constants: same path
    assert rel_err(x_dev, x_host) < 1e-8
    assert abs(res_dev.fun - res_host.fun) <= 1e-11 * max(1.0, abs(res_host.fun))
    # the oracle agrees that this is a stationary point, and a minimum
    g = model.grad(x_dev)
    assert np.linalg.norm(g) < 1e-6                         # gtol, judged by the oracle's gradient
    assert np.min(np.linalg.eigvalsh(model.hessian(x_dev))) > 0
    # side effect of the reference's wrappers: the parameter object sits at the optimum afterwards
    assert rel_err(obj.par.get_free(), x_dev) < 1e-14
    # the callback route pays a gradient pass inside every HVP; the device loop one pass per point
    assert res_dev.nhev >= res_dev.nit and res_dev.nfev <= 2 * res_dev.nit + 2


def test_every_truncated_run_matches(vb):
    """Stopping after k outer iterations gives the same point on both routes for every k (status 1 = maxiter)."""
    obj, model, rng = build(vb, om.LOGISTIC, 600, [('box', 'a', 10, -np.inf, np.inf), ('box', 'b', 6, 0.5, np.inf)], seed=5)
    x0 = rng.normal(size=model.layout.D) * 0.5
    for k in (1, 2, 3, 5):
        xh, rh = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=k, gtol=1e-12, disp=False)
        xd, rd = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=k, gtol=1e-12, disp=False,
                                                                   on_device=True)
        assert rd.nit == rh.nit == k and rd.status == rh.status == 1 and not rd.success
        assert rel_err(xd, xh) < 1e-10
        assert abs(rd.fun - model.value(xd)) <= 1e-12 * max(1.0, abs(rd.fun))


def test_preconditioned_route(vb):
    """`_cond` family: y = A^-1 x, f(A y), A^T g, A^T H A v with A = H^-1/2 from set_objective_preconditioner."""
    obj, model, rng = build(vb, om.POISSON, 800, [('box', 'a', 14, -np.inf, np.inf), ('box', 'b', 10, 0.0, np.inf)], seed=8)
    D = model.layout.D
    x0 = rng.normal(size=D) * 0.2
    vb.OptimizationUtils.set_objective_preconditioner(obj, free_par=x0, ev_min=1e-3)
    xh, rh = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, True, maxiter=100, gtol=1e-6, disp=False)
    xd, rd = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, True, maxiter=100, gtol=1e-6, disp=False,
                                                               on_device=True)
    assert rh.success and rd.success and rd.nit == rh.nit
    assert rel_err(xd, xh) < 1e-8
    assert rel_err(obj.preconditioner @ rd.x, xd) < 1e-13            # result.x is in the optimiser's coordinates
    assert np.linalg.norm(obj.preconditioner.T @ model.grad(xd)) < 1e-6      # gtol applies to the preconditioned gradient
    # Newton from the preconditioned start converges in fewer outer iterations than the plain route
    _, plain = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=100, gtol=1e-6, disp=False,
                                                                 on_device=True)
    assert rd.nit <= plain.nit
    with pytest.raises(AssertionError):                             # `_cond` without a preconditioner: the reference asserts
        obj.preconditioner = None
        vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, True, disp=False, on_device=True)


def test_general_layout_and_quadratic_model(vb):
    """PSD and simplex blocks (dense packing Jacobian, third-order term) and the reference's own test model:
    a quadratic with known optimum (LRVB/test_objectives.py:400-445)."""
    rng = np.random.default_rng(21)
    spec = [('box', 'pre', 2, -np.inf, np.inf), ('box', 'beta', 6, -1.0, np.inf), ('psd', 'm', 3, 0.2), ('simplex', 's', 2, 3)]
    par, lay = make_par(vb, spec)
    N, P = 500, 6
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    A = rng.normal(size=(lay.V, lay.V)); A = A @ A.T / lay.V + 3.0 * np.eye(lay.V)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', glm_param='beta', quad_A=A, weights=w)
    obj = vb.Objective(par, fun)
    x0 = rng.normal(size=lay.D) * 0.05
    xh, rh = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=300, gtol=1e-8, disp=False)
    xd, rd = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=300, gtol=1e-8, disp=False, on_device=True)
    assert rd.status == rh.status and rd.nit == rh.nit
    # both runs stop at |grad| < 1e-8; along the flattest direction of this layout (a simplex logit near -10) that leaves
    # ~1e-6 of freedom in x, and the two routes sum their reductions in different orders
    assert rel_err(xd, xh) < 1e-5
    assert np.max(np.abs(obj.fun_free_grad(xd))) < 1e-7
    # quadratic model: f(x) = 1/2 (x - t)^T Q (x - t), optimum t, reached exactly
    D = 9
    Q = rng.normal(size=(D, D)); Q = Q @ Q.T + np.eye(D)
    t = rng.normal(size=D)
    qpar = vb.VectorParam('x', D)
    qfun = vb.QuadraticObjective(qpar, A=Q, b=-Q @ t)
    qobj = vb.Objective(qpar, qfun)
    xq, rq = vb.OptimizationUtils.minimize_objective_trust_ncg(qobj, np.zeros(D), False, gtol=1e-8, disp=False, on_device=True)
    assert rq.success and rel_err(xq, t) < 1e-7


def test_refusals(vb):
    par = vb.VectorParam('x', 3)
    host_obj = vb.Objective(par, lambda: float(np.sum(par.get() ** 2)))          # opaque closure: host callbacks only
    with pytest.raises(NotImplementedError):
        vb.OptimizationUtils.minimize_objective_trust_ncg(host_obj, np.ones(3), False, disp=False, on_device=True)
    obj, model, rng = build(vb, om.GAUSSIAN, 50, [('box', 'b', 4, -np.inf, np.inf)], seed=2)
    ctx = obj.fun.ctx
    with pytest.raises(ValueError):
        ctx.minimize_trust_ncg(np.zeros(5))
    with pytest.raises(ValueError):
        ctx.minimize_trust_ncg(np.zeros(4), precond=np.eye(3))
    with pytest.raises(ValueError):
        ctx.minimize_trust_ncg(np.zeros(4), initial_trust_radius=0.0)
    with pytest.raises(ValueError):
        ctx.minimize_trust_ncg(np.zeros(4), eta=0.3)
    # a start that already satisfies gtol returns at once
    y, x, info = ctx.minimize_trust_ncg(np.zeros(4), gtol=1e30)
    assert info['nit'] == 0 and info['status'] == 0 and np.array_equal(x, np.zeros(4))


def test_sharded_objective_on_the_device(vb):
    """SURVEY section 8(e), HVP / CG path: with the quadratic term scaled by 1 / world on every shard, value, gradient
    and Hessian-vector products of the shards ADD UP to the full objective's -- on the real device contexts (two
    shards in one process here; the collective itself is covered by the 2-rank gloo test) -- and ShardedObjective over a
    one-rank group reproduces the unsharded answers."""
    import torch.distributed as dist
    from lrvb_amd.distributed import ShardedObjective, shard_rows
    rng = np.random.default_rng(12)
    spec = [('box', 'u', 20, -np.inf, np.inf), ('box', 'pos', 12, 0.0, np.inf)]
    N, P = 1501, 32
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    par, lay = make_par(vb, spec)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.9))
    theta, v = rng.normal(size=P) * 0.2, rng.normal(size=P)
    parts = []
    for r in range(2):
        r0, r1 = shard_rows(N, r, 2)
        p_r, _ = make_par(vb, spec)
        f_r = vb.DeviceObjective(p_r, x=x[r0:r1], y=y[r0:r1], loss='logistic', quad_A=np.full(P, 0.9), weights=w[r0:r1])
        f_r._push_state()
        f_r.ctx.set_quad_scale(0.5)
        parts.append(f_r.ctx)
    assert abs(sum(c.value(theta) for c in parts) - model.value(theta)) < 1e-12 * abs(model.value(theta))
    assert rel_err(sum(c.grad(theta) for c in parts), model.grad(theta)) < 1e-12
    assert rel_err(sum(c.hvp(theta, v) for c in parts), model.hessian(theta) @ v) < 1e-12
    # one-rank process group: the class end to end on a device context
    started = False
    if not dist.is_initialized():
        import os, socket
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=0, world_size=1)
        started = True
    try:
        fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 0.9), weights=w)
        fun._push_state()
        so = ShardedObjective(fun.ctx)
        assert so.world == 1
        assert rel_err(so.hvp(theta, v), model.hessian(theta) @ v) < 1e-12
        b = rng.normal(size=P)
        sol, info = so.cg_solve(theta, b, tol=1e-10)
        assert info == 0 and rel_err(sol, np.linalg.solve(model.hessian(theta), b)) < 1e-8
        fit = so.minimize_trust_ncg(theta, gtol=1e-7, maxiter=100)
        assert fit.status == 0 and np.linalg.norm(model.grad(fit.x)) < 1e-6
    finally:
        if started:
            dist.destroy_process_group()


def test_hessian_is_built_inside_long_cg_runs(vb):
    """After max(8, D / 64) products at one point the optimiser builds that point's Hessian (about D / 86 passes over the
    observations) and serves the rest of the Steihaug-CG run from it: same iterates as the matrix-free route (tuning bit 3),
    `nbuild` reports the builds, and far fewer passes over X are made."""
    spec = [('box', 'u', 230, -np.inf, np.inf), ('box', 'pos', 58, 0.0, np.inf)]       # D = 288 (the route starts at 256 parameters)
    obj, model, rng = build(vb, om.LOGISTIC, 6000, spec, seed=21, prior=0.05)
    D = model.layout.D
    x0 = rng.normal(size=D) * 0.5
    ctx = obj.fun.ctx
    ctx.set_tuning(0, 8)                                       # tuning bit 3: matrix-free products only
    x_free, r_free = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=100, gtol=1e-7, disp=False, on_device=True)
    ctx.set_tuning(0, 0)
    ctx.profile_enable(True); ctx.profile_reset()
    x_res, r_res = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, x0, False, maxiter=100, gtol=1e-7, disp=False, on_device=True)
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    assert r_free.success and r_res.success
    assert r_free.nbuild == 0 and r_res.nbuild >= 1
    assert abs(r_res.nit - r_free.nit) <= 1                                # the same path up to rounding in the last iterations
    assert rel_err(x_res, x_free) < 1e-7
    assert np.linalg.norm(model.grad(x_res)) < 1e-6
    # the products behind the threshold of each built point made no pass over X
    assert r_res.nhev > 8 * r_res.nbuild
    assert prof['wsyrk_calls'] == r_res.nbuild
