"""Blocked conjugate gradients (`lrvb_cg_solve_multi`): the masked right-hand sides of
ConjugateGradientSolver.get_hinv_vec_subsets (LRVB/ConjugateGradient.py:87-105) solved in lockstep.
Each row must reproduce the single-system solver (same recurrence, same stopping rule) and the
reference's own criterion against a direct solve (< 1e-8, LRVB/test_objectives.py:552-554)."""
import numpy as np
import pytest

from oracle import models as om
from helpers import make_par, glm_data, rel_err, LOSS_NAME

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


@pytest.mark.parametrize('loss,N,P,Q', [(om.GAUSSIAN, 2000, 64, 5), (om.POISSON, 3000, 200, 16), (om.LOGISTIC, 1500, 33, 3),
                                        (om.GAUSSIAN, 700, 130, 1),
                                        # n_cols % 128 == 0: the fused multi-vector pass (X read once per iteration)
                                        (om.GAUSSIAN, 1003, 128, 16), (om.POISSON, 4099, 384, 7), (om.LOGISTIC, 2001, 1024, 16),
                                        (om.GAUSSIAN, 5, 256, 20)])
def test_rows_match_single_solver_and_direct_solve(vb, loss, N, P, Q):
    rng = np.random.default_rng(N + Q)
    par, lay = make_par(vb, [('box', 'a', P // 2, -np.inf, np.inf), ('box', 'b', P - P // 2, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, loss)
    fun = vb.DeviceObjective(par, x=x, y=y, loss=LOSS_NAME[loss], quad_A=np.full(P, 1.0), weights=w)
    model = om.DeclaredModel(lay, loss=loss, x=x, y=y, w=w, quad_A=np.full(P, 1.0))
    theta = rng.normal(size=P) * 0.1
    fun._push_state()
    H = model.hessian(theta)
    assert np.min(np.linalg.eigvalsh(H)) > 0
    B = rng.normal(size=(Q, P))
    if Q > 2:
        B[1] = 0.0                                           # a zero right-hand side: solution 0, no iterations
    X, info, iters = fun.ctx.cg_solve_multi(theta, B)
    assert np.all(info == 0)
    want = np.linalg.solve(H, B.T).T
    # residual rule ||r|| < 1e-8 ||b||: the error is that times the conditioning of H
    assert np.max(np.abs(X - want)) < 1e-7 * max(1.0, np.max(np.abs(want)))
    for q in range(Q):
        xq, iq, itq = fun.ctx.cg_solve(theta, B[q])
        assert iq == 0 and itq == iters[q]
        assert rel_err(X[q], xq) < 1e-9 or np.max(np.abs(xq)) == 0.0
    if Q > 2:
        assert iters[1] == 0 and np.all(X[1] == 0.0)
    # preconditioner and warm start
    Minv = np.diag(1.0 / np.diag(H))
    X2, info2, iters2 = fun.ctx.cg_solve_multi(theta, B, X0=0.5 * want, Minv=Minv, tol=1e-10)
    assert np.all(info2 == 0) and np.max(np.abs(X2 - want)) < 1e-7 * max(1.0, np.max(np.abs(want)))
    # iteration cap is reported per row
    X3, info3, iters3 = fun.ctx.cg_solve_multi(theta, B, maxiter=2)
    nz = np.flatnonzero(np.abs(B).sum(axis=1) > 0)
    assert np.all(info3[nz] == 2) and np.all(iters3[nz] == 2)


def test_fused_pass_equals_two_gemm_route(vb):
    """The same blocked solve with the fused kernel switched off (tuning flag bit 0) must agree."""
    rng = np.random.default_rng(9)
    N, P, Q = 3001, 512, 11
    par, lay = make_par(vb, [('box', 'a', P, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, om.POISSON)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='poisson', quad_A=np.full(P, 1.0), weights=w)
    theta = rng.normal(size=P) * 0.1
    fun._push_state()
    B = rng.normal(size=(Q, P))
    X1, info1, it1 = fun.ctx.cg_solve_multi(theta, B)
    fun.ctx.set_tuning(0, 1)
    X2, info2, it2 = fun.ctx.cg_solve_multi(theta, B)
    fun.ctx.set_tuning(0, 0)
    # ~65 iterations here: the two routes sum in different orders, and CG's residual is not monotone near the
    # tolerance, so a system may cross it a few iterations apart (observed: 66 vs 69); the solutions must agree
    assert np.all(info1 == 0) and np.all(info2 == 0) and np.max(np.abs(it1 - it2)) <= 5
    assert rel_err(X1, X2) < 1e-7


def test_four_and_eight_wave_forms_agree(vb):
    """The fused multi-vector pass runs with eight waves per workgroup when the 128-column blocks split evenly over
    them (P = 256, 512, 768, 1024 after padding) and with four otherwise; tuning bit 2 forces four."""
    rng = np.random.default_rng(19)
    for P in (250, 512, 640, 1000):                     # padded to 256 (eight), 512 (eight), 640 (four), 1024 (eight)
        N, Q = 1207, 9
        par, lay = make_par(vb, [('box', 'a', P, -np.inf, np.inf)])
        x, y, w = glm_data(rng, N, P, om.LOGISTIC)
        fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 2.0), weights=w)
        model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 2.0))
        theta = rng.normal(size=P) * 0.05
        fun._push_state()
        B = rng.normal(size=(Q, P))
        X8, info8, _ = fun.ctx.cg_solve_multi(theta, B, tol=1e-11)
        fun.ctx.set_tuning(0, 4)
        X4, info4, _ = fun.ctx.cg_solve_multi(theta, B, tol=1e-11)
        fun.ctx.set_tuning(0, 0)
        assert np.all(info8 == 0) and np.all(info4 == 0)
        want = np.linalg.solve(model.hessian(theta), B.T).T
        assert rel_err(X8, want) < 1e-8 and rel_err(X4, want) < 1e-8
        assert rel_err(X8, X4) < 1e-9


def test_general_layout(vb):
    rng = np.random.default_rng(7)
    spec = [('box', 'pre', 2, -np.inf, np.inf), ('box', 'beta', 6, -1.0, np.inf), ('psd', 'm', 3, 0.2), ('simplex', 's', 2, 3)]
    par, lay = make_par(vb, spec)
    N, P, Q = 500, 6, 4
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    A = rng.normal(size=(lay.V, lay.V)); A = A @ A.T / lay.V + 3.0 * np.eye(lay.V)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', glm_param='beta', quad_A=A, weights=w)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, glm_off=2, quad_A=A)
    theta = rng.normal(size=lay.D) * 0.05
    H = model.hessian(theta)
    assert np.min(np.linalg.eigvalsh(H)) > 0            # dense prior precision >= 3 I dominates at this seeded point
    fun._push_state()
    B = rng.normal(size=(Q, lay.D))
    X, info, iters = fun.ctx.cg_solve_multi(theta, B)
    assert np.all(info == 0)
    assert np.max(np.abs(X - np.linalg.solve(H, B.T).T)) < 1e-8


def test_solver_class_batches_masks(vb):
    rng = np.random.default_rng(3)
    N, P = 4000, 256
    par, lay = make_par(vb, [('box', 'beta', P, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, om.POISSON)
    fun = vb.GLMObjective(par, x, y, loss='poisson', prior_info=1.0, weights=w)
    model = om.DeclaredModel(lay, loss=om.POISSON, x=x, y=y, w=w, quad_A=np.ones(P))
    obj = vb.Objective(par, fun)
    theta = rng.normal(size=P) * 0.2
    H = model.hessian(theta)
    solver = vb.ConjugateGradientSolver(obj.fun_free_hvp, theta)
    masks = vb.ConjugateGradient.get_masks(P, 16)
    vec = rng.normal(size=P)
    solver.get_hinv_vec_subsets(vec, masks)
    assert len(solver.hinv_vecs) == len(masks) == 16 and len(solver.times) == 16
    total = np.zeros(P)
    for rhs, sol, info, mask in zip(solver.vecs, solver.hinv_vecs, solver.cg_infos, solver.masks):
        assert info == 0 and np.array_equal(rhs[mask], vec[mask]) and np.all(rhs[~mask] == 0)
        assert np.max(np.abs(sol - np.linalg.solve(H, rhs))) < 1e-8
        total += sol
    assert np.max(np.abs(total - np.linalg.solve(H, vec))) < 1e-7          # the masks partition vec
    assert np.max(np.abs(par.get_free() - theta)) < 1e-12


def test_errors(vb):
    rng = np.random.default_rng(1)
    par, lay = make_par(vb, [('box', 'beta', 4, -np.inf, np.inf)])
    x, y, w = glm_data(rng, 50, 4, om.GAUSSIAN)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='gaussian', quad_A=np.ones(4))
    fun._push_state()
    with pytest.raises(ValueError):
        fun.ctx.cg_solve_multi(np.zeros(4), np.zeros((3, 5)))
    with pytest.raises(ValueError):
        fun.ctx.cg_solve_multi(np.zeros(3), np.zeros((3, 4)))


def test_point_state_is_reused_only_when_nothing_changed(vb):
    """lrvb_cg_solve / lrvb_cg_solve_multi keep the point state (eta, J, d f / d eta, curvature) of the last solve or product
    and skip the gradient pass when the next solve names the same point with nothing in between -- the reference's
    ConjugateGradientSolver solves many right-hand sides at ONE point (LRVB/ConjugateGradient.py:63-105).  The reuse must
    not survive a new point, new weights or a new quadratic scale; every solve is checked against a direct solve with the
    oracle's Hessian for the state it ran at, and the profile counts the gradient passes."""
    rng = np.random.default_rng(12)
    N, P, Q = 3000, 256, 6
    par, lay = make_par(vb, [('box', 'a', P // 2, -np.inf, np.inf), ('box', 'b', P - P // 2, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 1.0), weights=w)
    fun._push_state()
    ctx = fun.ctx
    ctx.set_tuning(0, 8)                               # matrix-free products only (from 256 parameters on, a long run of products at one point builds the point's Hessian: test_many_products_at_one_point_build_the_hessian)
    B = rng.normal(size=(Q, P))
    th1, th2 = rng.normal(size=P) * 0.1, rng.normal(size=P) * 0.1

    def check(theta, weights, scale=1.0):
        model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=weights, quad_A=np.full(P, scale))
        H = model.hessian(theta)
        X, info, _ = ctx.cg_solve_multi(theta, B, tol=1e-10)
        assert np.all(info == 0) and np.max(np.abs(X - np.linalg.solve(H, B.T).T)) < 1e-7
        x1, i1, _ = ctx.cg_solve(theta, B[0], tol=1e-10)
        assert i1 == 0 and np.max(np.abs(x1 - np.linalg.solve(H, B[0]))) < 1e-7
    ctx.profile_enable(True); ctx.profile_reset()
    check(th1, w)                                     # pass 1 (multi), reused by the single solve
    check(th1, w)                                     # same point again: no pass at all
    assert ctx.profile_get()['pass_calls'] == 1
    check(th2, w)                                     # new point: one more
    check(th1, w)                                     # and back
    assert ctx.profile_get()['pass_calls'] == 3
    w2 = rng.uniform(0.5, 1.5, N)
    ctx.set_weights(w2)
    check(th1, w2)                                    # same point, new weights: the state is rebuilt
    ctx.set_quad_scale(2.5)
    check(th1, w2, scale=2.5)
    assert ctx.profile_get()['pass_calls'] == 5
    ctx.profile_enable(False)


def test_products_use_the_resident_hessian_after_a_build(vb):
    """After a free-coordinate Hessian build the library keeps the matrix: products at the SAME point -- `lrvb_hvp`, `lrvb_cg_solve`,
    `lrvb_cg_solve_multi`, i.e. ConjugateGradientSolver at an optimum after fun_free_hessian (LRVB/ConjugateGradient.py:63-105) --
    run against it with NO pass over the observations (profile: zero pass / SYRK launches), give the results of the matrix-free
    route, and fall back to the passes as soon as the point, the weights, a hyper-parameter or the tuning change."""
    rng = np.random.default_rng(31)
    N, P = 4000, 256
    spec = [('box', 'free', 160, -np.inf, np.inf), ('box', 'pos', 96, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 0.7), weights=w)
    objective = vb.Objective(par, fun)
    ctx = fun.ctx
    theta = rng.normal(size=P) * 0.2
    B = rng.normal(size=(5, P))
    v = rng.normal(size=P)

    def passes(call):
        ctx.profile_enable(True); ctx.profile_reset()
        out = call()
        prof = ctx.profile_get(); ctx.profile_enable(False)
        return out, prof['pass_calls']

    # matrix-free reference results (no build yet: nothing resident)
    (X_mf, info_mf, it_mf), n_mf = passes(lambda: ctx.cg_solve_multi(theta, B, tol=1e-10))
    assert n_mf >= 1 and np.all(info_mf == 0)
    hv_mf = ctx.hvp(theta, v)
    H = objective.fun_free_hessian(theta)                          # the build: its matrix stays in the context
    (X_res, info_res, it_res), n_res = passes(lambda: ctx.cg_solve_multi(theta, B, tol=1e-10))
    assert n_res == 0                                             # no pass over X
    assert np.all(info_res == 0) and rel_err(X_res, np.linalg.solve(H, B.T).T) < 1e-8 and rel_err(X_res, X_mf) < 1e-8
    hv_res, n_hv = passes(lambda: ctx.hvp(theta, v))
    assert n_hv == 0 and rel_err(hv_res, H @ v) < 1e-13 and rel_err(hv_res, hv_mf) < 1e-11
    (x1, info1, _), n1 = passes(lambda: ctx.cg_solve(theta, B[0], tol=1e-10))
    assert n1 == 0 and info1 == 0 and rel_err(x1, np.linalg.solve(H, B[0])) < 1e-8
    # the reference class end to end: ConjugateGradientSolver over fun_free_hvp at the point of the build
    solver = vb.ConjugateGradientSolver(objective.fun_free_hvp, theta)
    masks = vb.ConjugateGradient.get_masks(P, 64)
    _, n_solver = passes(lambda: solver.get_hinv_vec_subsets(v, masks))
    assert n_solver == 0
    for rhs, sol, info in zip(solver.vecs, solver.hinv_vecs, solver.cg_infos):
        assert info == 0 and np.max(np.abs(sol - np.linalg.solve(H, rhs))) < 1e-8
    # another point: the resident matrix is not that point's -- matrix-free again, correct answer
    theta2 = theta + 1e-3
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.7))
    hv2, n2 = passes(lambda: ctx.hvp(theta2, v))
    assert n2 >= 1 and rel_err(hv2, model.hessian(theta2) @ v) < 1e-11
    # ... and the original point is still served from the matrix
    _, n_back = passes(lambda: ctx.hvp(theta, v))
    assert n_back == 0
    # new weights: the matrix is dropped
    w2 = w * rng.uniform(0.9, 1.1, N)
    fun.weights_par.set_vector(w2)
    fun._push_state()
    model.w = w2
    hv3, n3 = passes(lambda: ctx.hvp(theta, v))
    assert n3 >= 1 and rel_err(hv3, model.hessian(theta) @ v) < 1e-11
    # a new prior (hyper-parameter) after a rebuild: dropped again
    H2 = objective.fun_free_hessian(theta)
    assert passes(lambda: ctx.hvp(theta, v))[1] == 0
    fun.prior_info_par.set_vector(np.full(P, 0.9))
    model.quad_A = np.full(P, 0.9)
    hv4, n4 = passes(lambda: objective.fun_free_hvp(theta, v))
    assert n4 >= 1 and rel_err(hv4, model.hessian(theta) @ v) < 1e-11
    # the tuning switch: never use the resident matrix
    objective.fun_free_hessian(theta)
    ctx.set_tuning(0, 8)
    assert passes(lambda: ctx.hvp(theta, v))[1] >= 1
    ctx.set_tuning(0, 0)
    assert passes(lambda: ctx.hvp(theta, v))[1] == 0              # switched back on: the matrix was kept
    # vector coordinates are never served from the free-coordinate matrix
    eta = lay.constrain(theta)
    hvv, nv = passes(lambda: ctx.hvp(eta, v, is_free=False))
    assert nv >= 1 and rel_err(hvv, model.hessian_vec(eta) @ v) < 1e-11


def test_many_products_at_one_point_build_the_hessian(vb):
    """Right-hand sides solved ONE BY ONE at a point (`lrvb_cg_solve`, what the reference's ConjugateGradientSolver does,
    LRVB/ConjugateGradient.py:87-105) and scipy-style product callbacks (`lrvb_hvp`): past max(8, D / 64) matrix-free products
    at the point its Hessian is built and made resident (D >= 256), the remaining products make no pass over the observations,
    the answers are those of the matrix-free route, and a new point starts counting again."""
    rng = np.random.default_rng(77)
    N, P, Q = 5000, 320, 5
    par, lay = make_par(vb, [('box', 'a', P - 64, -np.inf, np.inf), ('box', 'b', 64, 0.0, np.inf)])
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 0.5), weights=w)
    fun._push_state()
    ctx = fun.ctx
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.5))
    B = rng.normal(size=(Q, P))
    for theta in (rng.normal(size=P) * 0.1, rng.normal(size=P) * 0.1):
        H = model.hessian(theta)
        ctx.profile_enable(True); ctx.profile_reset()
        sols = [ctx.cg_solve(theta, B[q], tol=1e-10) for q in range(Q)]
        prof = ctx.profile_get()
        ctx.profile_enable(False)
        for q, (xq, info, its) in enumerate(sols):
            assert info == 0 and np.max(np.abs(xq - np.linalg.solve(H, B[q]))) < 1e-7
        assert sum(s[2] for s in sols) > 40                    # far more products than the threshold ...
        assert prof['wsyrk_calls'] == 1                        # ... one build of the point's Hessian ...
        v = rng.normal(size=P)
        assert rel_err(ctx.hvp(theta, v), H @ v) < 1e-11       # ... which then serves lrvb_hvp too
    # product callbacks alone (scipy's trust-ncg / cg over lrvb_hvp): the tenth product at one point is served by the matrix
    theta = rng.normal(size=P) * 0.1
    H = model.hessian(theta)
    ctx.profile_enable(True); ctx.profile_reset()
    for k in range(12):
        v = rng.normal(size=P)
        assert rel_err(ctx.hvp(theta, v), H @ v) < 1e-11
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    assert prof['wsyrk_calls'] == 1


def test_automatic_build_under_a_reduce_hook(vb):
    """With a sum-over-ranks hook installed (a one-rank identity hook that records what it is handed), the automatic build
    of the point's Hessian inside a run of `lrvb_cg_solve` calls reduces its statistics buffer exactly once -- every rank
    counts the same products, so every rank builds at the same product -- and the products after it hand nothing to the
    hook; the solutions are those of a direct solve."""
    rng = np.random.default_rng(78)
    N, P, Q = 4000, 288, 4
    par, lay = make_par(vb, [('box', 'a', P, -np.inf, np.inf)])
    x, y, w = glm_data(rng, N, P, om.POISSON)
    fun = vb.DeviceObjective(par, x=x, y=y, loss='poisson', quad_A=np.full(P, 0.5), weights=w)
    fun._push_state()
    ctx = fun.ctx
    model = om.DeclaredModel(lay, loss=om.POISSON, x=x, y=y, w=w, quad_A=np.full(P, 0.5))
    theta = rng.normal(size=P) * 0.1
    H = model.hessian(theta)
    B = rng.normal(size=(Q, P))
    calls = []
    ctx.set_reduce_hook(lambda ptr, n, stream: calls.append(n))
    try:
        sols = [ctx.cg_solve(theta, B[q], tol=1e-10) for q in range(Q)]
    finally:
        ctx.set_reduce_hook(None)
    for q, (xq, info, its) in enumerate(sols):
        assert info == 0 and np.max(np.abs(xq - np.linalg.solve(H, B[q]))) < 1e-7
    stats_size = 1 + P + 6 * 128 * 128                   # [value | gradient | lower-triangle tiles] of a 288-column design: 3 x 4 / 2 = 6 tiles
    assert calls.count(stats_size) == 1
    n_products = sum(s[2] for s in sols)
    thr = max(8, P // 64)
    # one gradient pass ([value | gradient]) at the start, `thr` matrix-free products handing their P-vector to the hook, then --
    # in front of product thr + 1 -- the build; nothing after it, however many products follow
    assert n_products > 4 * thr
    assert calls == [1 + P] + [P] * thr + [stats_size]
