"""Host-side utilities either side of the hot path (SURVEY.md section 8(f) "next"): the subspace
parameter (LRVB/ProjectionParams.py, exercised by LRVB/test_variational_bayes.py:179), the Gauss-Hermite
expectations (LRVB/ExponentialFamilies.py:126-221) and the logistic-term expectations (LRVB/Modeling.py).
The reference has no fixtures for the quadrature functions, so they are checked against closed forms and
adaptive quadrature (scipy.integrate.quad) -- parity for them is pinned by those known answers only."""
import numpy as np
import pytest
from scipy import integrate, special, stats

import lrvb_amd as vb
from test_parameter_protocol_suite import required_methods, fd_jacobian


def test_reference_style_module_imports():
    import lrvb_amd.SparseObjectives as obj_lib
    import lrvb_amd.ExponentialFamilies as ef
    import lrvb_amd.Modeling as modeling
    from lrvb_amd import ProjectionParams, ConjugateGradient
    assert obj_lib.Objective is vb.Objective
    assert ef.get_e_fun_normal is vb.ExponentialFamilies.get_e_fun_normal
    assert modeling.get_standard_draws is vb.Modeling.get_standard_draws
    import lrvb_amd.version as version
    assert version.__version__ == vb.__version__ and vb.version is version
    assert ProjectionParams.SubspaceVectorParam is vb.SubspaceVectorParam
    assert ConjugateGradient.ConjugateGradientSolver is vb.ConjugateGradientSolver


# ---- SubspaceVectorParam ------------------------------------------------------------------------
def test_perpendicular_subspace_is_an_orthonormal_complement():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 6))
    basis = vb.ProjectionParams.get_perpendicular_subspace(x)
    assert basis.shape == (6, 4)
    np.testing.assert_allclose(x @ basis, 0.0, atol=1e-12)
    np.testing.assert_allclose(basis.T @ basis, np.eye(4), atol=1e-12)
    with pytest.raises(Exception):          # rank-deficient constraints: singular solve or failed count
        vb.ProjectionParams.get_perpendicular_subspace(np.vstack([x[0], 2 * x[0]]))


def test_subspace_vector_param_protocol():
    required_methods(vb.SubspaceVectorParam())                     # the reference's own call (:179)
    rng = np.random.default_rng(4)
    con = rng.standard_normal((2, 5))
    par = vb.SubspaceVectorParam('s', dim=5, perp_subspace=con)
    required_methods(par)
    assert par.free_size() == 3 and par.vector_size() == 5 and par.dim() == 5
    free = rng.standard_normal(3)
    par.set_free(free)
    np.testing.assert_allclose(con @ par.get(), 0.0, atol=1e-12)   # lands in the subspace
    np.testing.assert_allclose(par.get_free(), free, atol=1e-12)   # and round-trips
    assert par.names() == ['s_%d' % k for k in range(5)]
    # default constraint: zero mean
    zero_mean = vb.SubspaceVectorParam('z', dim=4)
    zero_mean.set_free(np.array([1.0, -2.0, 0.5]))
    assert abs(np.sum(zero_mean.get())) < 1e-12
    # like the reference, `set` checks only the size: an off-subspace value is kept as given
    zero_mean.set(np.ones(4))
    np.testing.assert_array_equal(zero_mean.get_vector(), np.ones(4))
    np.testing.assert_allclose(zero_mean.get_free(), 0.0, atol=1e-12)
    # the constraint rows are copied
    con[0, 0] += 100.0
    par.set_free(free)
    assert abs((con @ par.get())[0]) > 1e-3


def test_subspace_vector_param_errors():
    with pytest.raises(ValueError):
        vb.SubspaceVectorParam(dim=3, perp_subspace=np.ones((1, 4)))
    with pytest.raises(ValueError):
        vb.SubspaceVectorParam(dim=2, perp_subspace=np.eye(2))
    par = vb.SubspaceVectorParam(dim=3)
    with pytest.raises(ValueError):
        par.set(np.zeros(4))
    with pytest.raises(ValueError):
        par.set_free(np.zeros(3))
    with pytest.raises(NotImplementedError):       # no silent host route into a device layout
        par.layout_blocks()


def test_subspace_param_inside_a_dictionary():
    par = vb.ModelParamsDict('p')
    par.push_param(vb.VectorParam('a', 2, lb=0.0))
    par.push_param(vb.SubspaceVectorParam('s', dim=4))
    assert par.free_size() == 5 and par.vector_size() == 6
    free = np.array([0.1, -0.3, 1.0, 2.0, -1.0])
    par.set_free(free)
    np.testing.assert_allclose(par.get_free(), free, atol=1e-12)
    assert abs(np.sum(par['s'].get())) < 1e-12
    jac = np.asarray(par.free_to_vector_jac(free).todense())

    def forward(f):
        par.set_free(f)
        return np.array(par.get_vector())
    np.testing.assert_allclose(jac, fd_jacobian(forward, free), rtol=1e-7, atol=1e-8)


# ---- closed-form PSD transform derivatives ------------------------------------------------------
def test_pos_def_matrix_free_to_vector_and_derivatives():
    mp = vb.MatrixParameters
    rng = np.random.default_rng(5)
    free = 0.3 * rng.standard_normal(6)
    for diag_lb in (0.0, 0.4):
        vec = mp.pos_def_matrix_free_to_vector(free, diag_lb=diag_lb)
        mat = mp.unpack_posdef_matrix(free, diag_lb=diag_lb)
        np.testing.assert_array_equal(vec, mat[np.tril_indices(3)])
        jac = mp.pos_def_matrix_free_to_vector_jac(free, diag_lb=diag_lb)
        np.testing.assert_allclose(
            jac, fd_jacobian(lambda f: mp.pos_def_matrix_free_to_vector(f, diag_lb=diag_lb), free),
            rtol=1e-7, atol=1e-9)
        hess = mp.pos_def_matrix_free_to_vector_hess(free, diag_lb=diag_lb)
        np.testing.assert_allclose(
            hess, fd_jacobian(lambda f: mp.pos_def_matrix_free_to_vector_jac(f, diag_lb=diag_lb), free),
            rtol=1e-6, atol=1e-8)


# ---- Gauss-Hermite expectations -----------------------------------------------------------------
GH_X, GH_W = np.polynomial.hermite.hermgauss(40)


def normal_expectation(fun, mean, info):
    sd = 1.0 / np.sqrt(info)
    val, _ = integrate.quad(lambda x: fun(x) * stats.norm.pdf(x, mean, sd), mean - 12 * sd, mean + 12 * sd,
                            epsabs=1e-13, epsrel=1e-13, limit=400)
    return val


def test_get_e_fun_normal_is_exact_for_polynomials():
    ef = vb.ExponentialFamilies
    means = np.array([[0.3, -1.2], [2.0, 0.0]])
    infos = np.array([[0.5, 2.0], [4.0, 1.0]])
    x, w = np.polynomial.hermite.hermgauss(4)         # exact up to degree 7
    np.testing.assert_allclose(ef.get_e_fun_normal(means, infos, x, w, lambda v: v), means, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(ef.get_e_fun_normal(means, infos, x, w, lambda v: v ** 2),
                               means ** 2 + 1 / infos, rtol=1e-13)
    np.testing.assert_allclose(ef.get_e_fun_normal(means, infos, x, w, lambda v: v ** 4),
                               means ** 4 + 6 * means ** 2 / infos + 3 / infos ** 2, rtol=1e-12)
    with pytest.raises(AssertionError):
        ef.get_e_fun_normal(means, infos[0], x, w, lambda v: v)


def test_logitnormal_expectations_against_adaptive_quadrature():
    ef = vb.ExponentialFamilies
    means = np.array([-1.5, 0.0, 0.7, 3.0])
    infos = np.array([2.0, 1.0, 4.0, 0.8])
    e_v = ef.get_e_logitnormal(means, infos, GH_X, GH_W)
    e_log_v, e_log_1mv = ef.get_e_log_logitnormal(means, infos, GH_X, GH_W)
    for i in range(4):
        assert abs(e_v[i] - normal_expectation(special.expit, means[i], infos[i])) < 1e-9
        assert abs(e_log_v[i] - normal_expectation(lambda x: -np.logaddexp(0, -x), means[i], infos[i])) < 1e-9
        assert abs(e_log_1mv[i] - normal_expectation(lambda x: -np.logaddexp(0, x), means[i], infos[i])) < 1e-9
    # far left tail: log(expit(x)) -> x, no overflow, no nan
    lo, lo1m = ef.get_e_log_logitnormal(np.array([-800.0]), np.array([1.0]), GH_X, GH_W)
    assert np.isfinite(lo[0]) and abs(lo[0] + 800.0) < 1e-6 and abs(lo1m[0]) < 1e-6
    # DP stick prior
    np.testing.assert_allclose(ef.get_e_dp_prior_logitnorm_approx(3.5, means, infos, GH_X, GH_W),
                               2.5 * e_log_1mv, rtol=1e-14)


def test_uvn_from_natural_parameters():
    mean, info = vb.ExponentialFamilies.get_uvn_from_natural_parameters(np.array([3.0, -1.0]), np.array([-0.5, -2.0]))
    np.testing.assert_allclose(info, [1.0, 4.0])
    np.testing.assert_allclose(mean, [3.0, -0.25])


# ---- Modeling.py --------------------------------------------------------------------------------
def test_standard_draws_are_normal_quantiles():
    draws = vb.Modeling.get_standard_draws(9)
    np.testing.assert_allclose(stats.norm.cdf(draws), np.arange(1, 10) / 10.0, atol=1e-14)
    np.testing.assert_allclose(draws, -draws[::-1], atol=1e-14)


def test_logistic_term_expectations():
    md = vb.Modeling
    rng = np.random.default_rng(6)
    z_mean = rng.standard_normal((3, 2))
    z_sd = 0.3 + rng.random((3, 2))
    y = (rng.random((3, 2)) > 0.5).astype(np.float64)
    exact = np.array([[normal_expectation(lambda x: np.logaddexp(0, x), z_mean[i, j], 1 / z_sd[i, j] ** 2)
                       for j in range(2)] for i in range(3)])
    per_elem = md.get_e_logistic_term_guass_hermite(z_mean, z_sd, GH_X, GH_W, aggregate_all=False)
    np.testing.assert_allclose(per_elem, exact, atol=1e-10)
    total = md.get_e_logistic_term_guass_hermite(z_mean, z_sd, GH_X, GH_W)
    assert abs(total - exact.sum()) < 1e-9
    # the draws version converges to sum(y z_mean) - E[log(1 + e^z)] as the draw grid refines
    approx = md.get_e_logistic_term(y, z_mean, z_sd, md.get_standard_draws(4000))
    assert abs(approx - (np.sum(y * z_mean) - exact.sum())) < 2e-3
    # a large z does not overflow
    assert np.isfinite(md.get_e_logistic_term(np.ones(1), np.array([900.0]), np.array([1.0]), md.get_standard_draws(5)))
    with pytest.raises(AssertionError):
        md.get_e_logistic_term(y, z_mean[0], z_sd, md.get_standard_draws(5))
    with pytest.raises(AssertionError):
        md.get_e_logistic_term_guass_hermite(z_mean, z_sd[0], GH_X, GH_W)


def test_univariate_normal_log_prob_follows_the_reference_formula():
    # the reference adds 0.5 * u_info (not its log); mirrored so that code calling it sees the same numbers
    val = vb.Modeling.univariate_normal_log_prob(1.5, 0.5, 2.0)
    assert abs(val - (-0.5 * 2.0 * 1.0 + 1.0 - 0.5 * np.log(2 * np.pi))) < 1e-15
