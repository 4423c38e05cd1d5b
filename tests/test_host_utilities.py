"""Host utilities of the reference that the GPU path does not touch but a switching user may call -- the remaining
functions of ExponentialFamilies, regression_utils, MatrixParameters and Parameters -- each against an INDEPENDENT answer
(scipy.stats / scipy.special, Monte Carlo at three standard errors, or the defining identity)."""
import numpy as np
import scipy.special
import scipy.stats

import lrvb_amd as vb

ef = vb.ExponentialFamilies


def test_multivariate_gamma_functions():
    for x, p in ((3.7, 3), (10.2, 5), (2.1, 1)):
        assert abs(ef.multivariate_gammaln(x, p) - scipy.special.multigammaln(x, p)) < 1e-12
        h = 1e-5
        fd = (scipy.special.multigammaln(x + h, p) - scipy.special.multigammaln(x - h, p)) / (2 * h)
        assert abs(ef.multivariate_digamma(x, p) - fd) < 1e-8


def test_lognormal_and_dirichlet_moments():
    mu, s2 = 0.3, 0.49
    d = scipy.stats.lognorm(s=np.sqrt(s2), scale=np.exp(mu))
    assert abs(ef.get_e_lognormal(mu, s2) - d.mean()) < 1e-12 and abs(ef.get_var_lognormal(mu, s2) - d.var()) < 1e-12
    alpha = np.array([[2.0, 0.7], [3.5, 1.1], [0.9, 4.0]])
    draws = scipy.stats.dirichlet.rvs(alpha[:, 0], size=40000, random_state=1)
    ld = np.log(draws)
    assert np.all(np.abs(ef.get_e_log_dirichlet(alpha)[:, 0] - ld.mean(axis=0)) < 3 * ld.std(axis=0) / np.sqrt(len(ld)))
    np.testing.assert_allclose(np.exp(ef.get_e_log_dirichlet(alpha)).sum(axis=0) < 1.0, True)       # Jensen


def test_wishart_inverse_diagonal_and_lkj_prior():
    rng = np.random.default_rng(2)
    k, df = 3, 9.5
    a = rng.normal(size=(k, k)); v = a @ a.T / k + np.eye(k)
    draws = scipy.stats.wishart.rvs(df, v, size=30000, random_state=3)
    ld = np.log(np.diagonal(np.linalg.inv(draws), axis1=1, axis2=2))
    assert np.all(np.abs(ef.e_log_inv_wishart_diag(df, v) - ld.mean(axis=0)) < 3 * ld.std(axis=0) / np.sqrt(len(ld)))
    # LKJ prior term: (eta - 1) E log|R|, log|R| = -log|W| - sum log diag(W^-1) for the correlation matrix R of W^-1
    lr = -np.linalg.slogdet(draws)[1] - ld.sum(axis=1)
    want, se = 2.5 * lr.mean(), 2.5 * lr.std() / np.sqrt(len(lr))
    assert abs(ef.expected_ljk_prior(3.5, df, v) - want) < 3 * se


def test_prior_terms_are_expected_log_densities():
    rng = np.random.default_rng(4)
    m0, info0, e, var = 0.4, 2.2, -0.3, 0.8
    x = rng.normal(e, np.sqrt(var), size=200000)
    lp = -0.5 * info0 * (x - m0) ** 2
    assert abs(ef.uvn_prior(m0, info0, e, var) - lp.mean()) < 3 * lp.std() / np.sqrt(x.size)
    assert ef.exponential_prior(1.7, 0.6) == -1.7 * 0.6
    alpha = np.array([1.5, 2.0, 0.5]); le = np.log(np.array([0.2, 0.5, 0.3]))
    assert abs(ef.dirichlet_prior(alpha, le) - np.sum((alpha - 1) * le)) < 1e-15


def test_regression_natural_parameters():
    rng = np.random.default_rng(5)
    n, r = 40, 3
    x = rng.normal(size=(n, r)); y = rng.normal(size=n); info = 1.9
    n1, n2 = vb.regression_utils.get_nat_params_from_likelihood(y, x, info)
    np.testing.assert_allclose(n1, info * x.T @ y, rtol=1e-13)
    np.testing.assert_allclose(n2, -0.5 * info * x.T @ x, rtol=1e-13)
    m0 = rng.normal(size=r); a = rng.normal(size=(r, r)); i0 = a @ a.T + np.eye(r)
    p1, p2 = vb.regression_utils.get_nat_params_from_prior(m0, i0)
    np.testing.assert_allclose(p1, i0 @ m0, rtol=1e-13)
    np.testing.assert_allclose(p2, -0.5 * i0, rtol=1e-13)
    mean, post_info = vb.regression_utils.get_mvn_from_nat_params(n1 + p1, n2 + p2)
    np.testing.assert_allclose(post_info, info * x.T @ x + i0, rtol=1e-13)
    np.testing.assert_allclose(mean, np.linalg.solve(info * x.T @ x + i0, info * x.T @ y + i0 @ m0), rtol=1e-12)


def test_matrix_and_offset_helpers():
    mp, pr = vb.MatrixParameters, vb.Parameters
    m = np.array([[0.5, 2.0], [-1.0, 1.5]])
    np.testing.assert_allclose(mp.exp_matrix_diagonal(m), [[np.exp(0.5), 2.0], [-1.0, np.exp(1.5)]])
    np.testing.assert_allclose(mp.log_matrix_diagonal(mp.exp_matrix_diagonal(m)), m)
    sym = np.array([[2.0, 0.3, -0.1], [0.3, 1.0, 0.7], [-0.1, 0.7, 3.0]])
    np.testing.assert_allclose(mp.unvectorize_symmetric_matrix(mp.vectorize_ld_matrix(sym)), sym)
    # offset helpers walk a flat vector parameter by parameter (LRVB/Parameters.py:329-380)
    a, b = vb.VectorParam('a', 2, lb=0.0), vb.ScalarParam('b', lb=-1.0, ub=1.0)
    free = np.array([0.1, -0.4, 0.7])
    off = pr.set_free_offset(a, free, 0); off = pr.set_free_offset(b, free, off)
    assert off == 3
    out = np.zeros(3); off = pr.get_free_offset(a, out, 0); pr.get_free_offset(b, out, off)
    np.testing.assert_allclose(out, free)
    vec = np.zeros(3); off = pr.get_vector_offset(a, vec, 0); pr.get_vector_offset(b, vec, off)
    np.testing.assert_allclose(vec, [np.exp(0.1), np.exp(-0.4), 2.0 / (1.0 + np.exp(-0.7)) - 1.0])
    off = pr.set_vector_offset(a, np.array([1.0, 2.0, 0.5]), 0); pr.set_vector_offset(b, np.array([1.0, 2.0, 0.5]), off)
    np.testing.assert_allclose(a.get(), [1.0, 2.0]); assert abs(b.get() - 0.5) < 1e-15
    fo, vo, jac = pr.free_to_vector_jac_offset(a, free, 0, 0)
    assert (fo, vo) == (2, 2)
    np.testing.assert_allclose(jac.toarray(), np.diag(np.exp(free[:2])))
    hs = []
    assert pr.free_to_vector_hess_offset(b, free, hs, 2, (3, 3)) == 3
    s = 1.0 / (1.0 + np.exp(-0.7))
    assert len(hs) == 1 and abs(hs[0].toarray()[2, 2] - 2.0 * s * (1 - s) * (1 - 2 * s)) < 1e-14
    sp = pr.offset_sparse_matrix(np.array([[1.0, 0.0], [0.0, 2.0]]), (1, 2), (4, 5)).toarray()
    assert sp.shape == (4, 5) and sp[1, 2] == 1.0 and sp[2, 3] == 2.0 and sp.sum() == 3.0


def test_small_family_and_container_methods():
    """E[exp(x)^2] of the univariate normal families (Monte Carlo), the moment array filled from a constant, the Wishart's LKJ
    prior term, array ranges of a stack of PSD matrices, dictionary renaming, the logger's message, the converter forwarders."""
    rng = np.random.default_rng(6)
    u = vb.UVNParamVector('u', length=3)
    u['mean'].set(np.array([0.1, -0.3, 0.5])); u['info'].set(np.array([4.0, 9.0, 2.5]))
    draws = rng.normal(u['mean'].get(), 1 / np.sqrt(u['info'].get()), size=(200000, 3))
    e2 = np.exp(draws) ** 2
    assert np.all(np.abs(u.e2_exp() - e2.mean(axis=0)) < 4 * e2.std(axis=0) / np.sqrt(len(e2)))
    s = vb.UVNParam('s'); s['mean'].set(0.2); s['info'].set(5.0)
    assert abs(s.e2_exp() - np.exp(2 * 0.2 + 2.0 / 5.0)) < 1e-12
    arr = vb.UVNParamArray('a', shape=(2, 2)); arr['mean'].set(np.full((2, 2), 0.1)); arr['info'].set(np.full((2, 2), 4.0))
    np.testing.assert_allclose(arr.e2_exp(), np.exp(0.2 + 0.5))
    mom = vb.UVNMomentParamArray('m', shape=(2, 3))
    const = vb.ArrayParam('c', shape=(2, 3), val=np.arange(6.0).reshape(2, 3) + 1)
    mom.set_from_constant(const)
    np.testing.assert_allclose(mom.e(), const.get()); np.testing.assert_allclose(mom.var(), 0.0, atol=1e-15)
    np.testing.assert_allclose(mom.e2_exp(), mom.e_exp() ** 2 + mom.var_exp())
    wp = vb.WishartParam('w', size=3)
    a = rng.normal(size=(3, 3)); v = a @ a.T / 3 + np.eye(3)
    wp['df'].set(8.5); wp['v'].set(v)
    assert abs(wp.e_log_lkj_inv_prior(2.0) - ef.expected_ljk_prior(2.0, 8.5, v)) < 1e-14
    stack = vb.PosDefMatrixParamArray('ps', array_shape=(2, 2), matrix_size=2)
    ranges = stack.get_array_ranges()
    assert [len(r) for r in ranges] == [2, 2]
    d = vb.ModelParamsDict('old'); d.set_name('new'); assert d.name == 'new'
    lg = vb.Logger(print_every=1)
    seen = []
    lg.callback = lambda logger: seen.append(logger.iter)
    lg.log(1.0, np.zeros(1)); lg.log(0.5, np.ones(1))
    assert seen == [0, 1]
    lg.callback = None
    lg.print_message()                                        # the reference's default message: prints, returns nothing
    # converter forwarders with an opaque closure: values in all four coordinate combinations
    pin, pout = vb.VectorParam('x', 2, lb=0.0), vb.VectorParam('y', 2, lb=1.0)
    conv = vb.ParameterConverter(pin, pout, lambda: pout.set_vector(1.0 + pin.get() ** 2))
    f = np.array([0.2, -0.1]); xv = np.exp(f)
    np.testing.assert_allclose(conv.converter_free_to_vec(f), 1.0 + xv ** 2)
    np.testing.assert_allclose(conv.converter_vec_to_vec(xv), 1.0 + xv ** 2)
    np.testing.assert_allclose(conv.converter_free_to_free(f), np.log(xv ** 2))
    np.testing.assert_allclose(conv.converter_vec_to_free(xv), np.log(xv ** 2))
