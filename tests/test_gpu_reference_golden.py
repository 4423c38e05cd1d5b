"""The device path against fixtures produced by RUNNING the reference's own code (tests/golden/make_golden.py):
`ConjugateGradientSolver` on the reference's `test_cg` problem (LRVB/test_objectives.py:524-554) and the optimiser
wrappers of LRVB/OptimizationUtils.py:25-162 on a small seeded declared model."""
import numpy as np
import pytest

from golden_problems import G, optimiser_problem, check_optimiser_wrappers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    assert lrvb_amd._hip.device_count() >= 1
    return lrvb_amd


def test_device_cg_reproduces_the_reference_solver_run(vb):
    mat, loc, x, masks = G['cg_mat'], G['cg_loc'], G['cg_x'], G['cg_masks']
    K = mat.shape[0]
    ref_vecs, ref_sols, ref_infos = G['cg_vecs'], G['cg_hinv_vecs'], G['cg_infos']
    assert np.all(ref_infos == 0)
    # the reference's problem as a declared objective: f(x) = (x - loc)^T mat (x - loc), HVP = 2 mat v
    par = vb.ModelParamsDict('par')
    par.push_param(vb.VectorParam('x', K))
    fun = vb.QuadraticObjective(par, A=2.0 * mat, m=loc)
    objective = vb.Objective(par, fun)
    v = np.linspace(-1.0, 1.0, K)
    np.testing.assert_allclose(objective.fun_free_hvp(loc, v), 2.0 * (mat @ v), rtol=1e-13, atol=1e-13)

    # (1) the package's ConjugateGradientSolver driving the device HVP, called as the reference's test calls it
    solver = vb.ConjugateGradientSolver(objective.fun_free_hvp, loc)
    solver.get_hinv_vec_subsets(x, [m for m in masks])
    assert np.array_equal(np.array(solver.vecs), ref_vecs)                 # masked right-hand sides
    assert list(solver.cg_infos) == list(ref_infos)
    assert np.max(np.abs(np.array(solver.hinv_vecs) - ref_sols)) < 2e-8    # both stop by the same 1e-8 rule
    exact = np.linalg.solve(2.0 * mat, ref_vecs.T).T
    assert np.max(np.abs(np.array(solver.hinv_vecs) - exact)) < 1e-8       # LRVB/test_objectives.py:552-554

    # (2) the whole loop on the device: one system at a time, all of them in lockstep, and on the resident matrix
    for rhs, want in zip(ref_vecs, ref_sols):
        sol, info, iters = fun.ctx.cg_solve(loc, rhs, tol=1e-8)
        assert info == 0 and np.max(np.abs(sol - want)) < 2e-8
    X, infos, its = fun.ctx.cg_solve_multi(loc, ref_vecs, tol=1e-8)
    assert np.all(infos == 0) and np.max(np.abs(X - ref_sols)) < 2e-8
    for i, (rhs, want) in enumerate(zip(ref_vecs, ref_sols)):
        sol, info, iters = fun.ctx.cg_solve_matrix(2.0 * mat if i == 0 else None, rhs, tol=1e-8)
        assert info == 0 and np.max(np.abs(sol - want)) < 2e-8


def test_device_optimisers_reproduce_the_reference_wrapper_runs(vb):
    par, lay, model, arr = optimiser_problem(vb)
    fun = vb.DeviceObjective(par, x=arr['x'], y=arr['y'], loss='logistic', quad_A=np.full(arr['P'], arr['prior']),
                             weights=arr['w'])
    objective = vb.Objective(par, fun)
    check_optimiser_wrappers(vb, objective, lay, on_device_too=True)
