"""The oracle (and the product's host helpers) against golden vectors produced by running the
reference's own ConjugateGradient.py / OptimizationUtils.py (tests/golden/make_golden.py)."""
import os

import numpy as np

from oracle import solvers as osv

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors.npz'))


def test_masks_and_splits_match_reference():
    import lrvb_amd as vb
    for impl in (osv, vb.ConjugateGradient):
        assert np.array_equal(np.array(impl.get_masks(20, 3)), G['masks_20_3'])
        a, b = impl.split_vector(G['split_in'])
        assert np.array_equal(a, G['split_1']) and np.array_equal(b, G['split_2'])
    assert np.array_equal(np.array(osv.recursive_split(G['rsplit_in'], terminate_len=5)), G['rsplit_out'])
    res = []
    vb.ConjugateGradient.recursive_split(G['rsplit_in'], results=res, terminate_len=5)
    assert np.array_equal(np.array(res), G['rsplit_out'])


def test_sym_matrix_inv_sqrt_matches_reference():
    import lrvb_amd as vb
    h = G['invsqrt_in']
    for tag, kw in (('plain', {}), ('min', {'ev_min': 1.0}), ('max', {'ev_max': 5.0}),
                    ('both', {'ev_min': 1.0, 'ev_max': 5.0})):
        for impl in (osv.sym_matrix_inv_sqrt, vb.OptimizationUtils.get_sym_matrix_inv_sqrt):
            isq, corr = impl(h, **kw)
            np.testing.assert_allclose(isq, G['invsqrt_' + tag], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(corr, G['invsqrt_corr_' + tag], rtol=1e-12, atol=1e-13)


def test_cg_matches_reference_solver():
    mat, loc = G['cg_mat'], G['cg_loc']
    chol = np.linalg.cholesky(2.0 * mat)
    for rhs, ref_sol, info in zip(G['cg_vecs'], G['cg_hinv_vecs'], G['cg_infos']):
        assert info == 0
        x, my_info, iters = osv.cg_solve(lambda v: 2.0 * (mat @ v), rhs, tol=1e-8)
        assert my_info == 0
        # both satisfy the same stopping rule; compare with the reference solver's output and with
        # the exact solve at the reference test's own tolerance (test_objectives.py:552-554)
        exact = np.linalg.solve(chol.T, np.linalg.solve(chol, rhs))
        assert np.max(np.abs(x - exact)) < 1e-8
        assert np.max(np.abs(ref_sol - exact)) < 1e-8
        assert np.max(np.abs(x - ref_sol)) < 2e-8
    # the masked right-hand sides are the reference's
    for m, rhs in zip(G['cg_masks'], G['cg_vecs']):
        want = np.zeros(len(m)); want[m] = G['cg_x'][m]
        assert np.array_equal(rhs, want)


def test_optimiser_wrappers_match_reference_runs():
    """The package's OptimizationUtils (host logic) on the oracle-backed functor against what the reference's own
    functions returned for the same model; the device functor takes the same route in tests/test_gpu_reference_golden.py."""
    import lrvb_amd as vb
    from golden_problems import optimiser_problem, check_optimiser_wrappers
    from oracle_functor import OracleFunctor
    par, lay, model, arr = optimiser_problem(vb)
    objective = vb.Objective(par, OracleFunctor(par, model))
    check_optimiser_wrappers(vb, objective, lay, on_device_too=False)
