"""The Parameter protocol, class by class, in the arrangement of the reference's own suite
(LRVB/test_variational_bayes.py: TestParameterMethods :109-280, TestConstrainingFunctions :282-319,
TestParameters :322-526, TestParameterDictionary :529-640, test_sparse_free_hessians :839-883).
Where the reference compares a derivative with autograd (absent here), this suite compares the
product's analytic sparse Jacobians / Hessians with Richardson-extrapolated central differences of the
product's own forward map (the forward maps themselves are pinned against the oracle elsewhere)."""
import copy

import numpy as np
import pytest
import scipy.sparse
import scipy.stats

import lrvb_amd as vb

LBS = [0.0, -2.0, 1.2, -np.inf]
UBS = [0.0, -1.0, 2.1, np.inf]


def inbounds(lb, ub):
    """A value strictly inside the bounds.  (`get_inbounds_value` follows the reference, which returns
    0.5 * (ub - lb) for two-sided bounds -- outside [lb, ub] when the interval does not straddle it.)"""
    if np.isfinite(lb) and np.isfinite(ub):
        return 0.5 * (lb + ub)
    return vb.Parameters.get_inbounds_value(lb, ub)


def fd_jacobian(fun, x, h=1e-4):
    """Central differences with one Richardson step: O(h^4)."""
    x = np.asarray(x, dtype=np.float64)
    f0 = np.asarray(fun(x))
    J = np.zeros(f0.shape + (x.size,))
    for i in range(x.size):
        e = np.zeros(x.size); e[i] = 1.0
        d1 = (np.asarray(fun(x + h * e)) - np.asarray(fun(x - h * e))) / (2 * h)
        d2 = (np.asarray(fun(x + 2 * h * e)) - np.asarray(fun(x - 2 * h * e))) / (4 * h)
        J[..., i] = (4.0 * d1 - d2) / 3.0
    return J


def check_sparse_transforms(param):
    rng = np.random.default_rng(42)
    free = rng.random(param.free_size())

    def forward(f):
        param.set_free(f)
        return np.array(param.get_vector(), dtype=np.float64)

    jac = param.free_to_vector_jac(free)
    assert scipy.sparse.issparse(jac)
    np.testing.assert_allclose(np.asarray(jac.todense()), fd_jacobian(forward, free), rtol=1e-7, atol=1e-8)
    hess = param.free_to_vector_hess(free)
    assert len(hess) == param.vector_size()

    def jac_dense(f):
        return np.asarray(param.free_to_vector_jac(f).todense())

    fd_h = fd_jacobian(jac_dense, free)                    # (V, D, D)
    for row, h in enumerate(hess):
        assert scipy.sparse.issparse(h)
        np.testing.assert_allclose(np.asarray(h.todense()), fd_h[row], rtol=1e-6, atol=1e-7)
    param.set_free(free)


def required_methods(param, sparse=True):
    param.names()
    param.dictval()
    free = param.get_free()
    param.set_free(free)
    assert np.ndim(free) == 1
    vec = param.get_vector()
    param.set_vector(vec)
    assert np.ndim(vec) == 1
    str(param)
    assert param.free_size() == len(free)
    assert param.vector_size() == len(vec)
    # free -> value has a non-trivial, smooth dependence
    def value(f):
        param.set_free(f)
        return np.array(param.get_vector(), dtype=np.float64)
    assert np.max(np.abs(fd_jacobian(value, free))) > 0
    param.set_free(free)
    if sparse:
        check_sparse_transforms(param)


# ---- TestParameterMethods -------------------------------------------------------------------------
def test_scalar_vector_array_methods():
    required_methods(vb.ScalarParam(lb=0.5, val=2.0))
    required_methods(vb.VectorParam(size=3, lb=-1.0, ub=2.0))
    required_methods(vb.ArrayParam(shape=(2, 3), ub=4.0))


def test_pos_def_matrix_and_simplex_methods():
    required_methods(vb.PosDefMatrixParam(size=3, diag_lb=0.1))
    required_methods(vb.SimplexParam(shape=(3, 4)))


def test_pos_def_matrix_vector_and_array():
    """LRVB/test_variational_bayes.py:124-173."""
    rng = np.random.default_rng(8)
    k, length = 3, 4
    mats = np.array([(lambda a: a @ a.T + 0.5 * np.eye(k))(rng.normal(size=(k, k))) for _ in range(length)])
    pv = vb.PosDefMatrixParamVector('pv', length=length, matrix_size=k, diag_lb=0.1)
    required_methods(pv)
    pv.set(mats)
    np.testing.assert_allclose(pv.get(), mats)
    free = pv.get_free()
    assert free.size == length * k * (k + 1) // 2
    for obs in range(length):
        np.testing.assert_allclose(free[pv.free_obs_slice(obs)],
                                   vb.MatrixParameters.pack_posdef_matrix(mats[obs], diag_lb=0.1))
    pv.set_free(rng.normal(size=free.size))
    pv.set_free(free)
    np.testing.assert_allclose(pv.get(), mats, atol=1e-12)
    vec = pv.get_vector()
    pv.set_free(rng.normal(size=free.size))
    pv.set_vector(vec)
    np.testing.assert_allclose(pv.get(), mats, atol=1e-14)
    with pytest.raises(ValueError):
        pv.set_free(free[:-1])
    with pytest.raises(ValueError):
        pv.set_vector(vec[:-1])
    with pytest.raises(ValueError):
        pv.set(mats[:-1])
    assert pv.length() == length and pv.matrix_size() == k
    shape = (2, 3)
    amats = np.array([[(lambda a: a @ a.T + np.eye(2))(rng.normal(size=(2, 2))) for _ in range(3)] for _ in range(2)])
    pa = vb.PosDefMatrixParamArray('pa', array_shape=shape, matrix_size=2, diag_lb=0.0)
    required_methods(pa)
    pa.set(amats)
    free = pa.get_free()
    for obs in [(0, 0), (1, 2), (0, 1)]:
        np.testing.assert_allclose(free[pa.stacked_obs_slice(obs)], vb.MatrixParameters.pack_posdef_matrix(amats[obs]))
    pa.set_free(np.zeros(free.size))
    pa.set_free(free)
    np.testing.assert_allclose(pa.get(), amats, atol=1e-12)
    dets = pa.apply_matrix_function(np.linalg.det)
    assert dets.shape == shape
    np.testing.assert_allclose(dets, np.linalg.det(amats))
    invs = pa.apply_matrix_function(np.linalg.inv)
    assert invs.shape == shape + (2, 2)
    # inside a dictionary the blocks keep their offsets
    mp = vb.ModelParamsDict('d')
    mp.push_param(vb.VectorParam('v', 2, lb=0.0))
    mp.push_param(pa)
    check_sparse_transforms(mp)
    assert len(mp.layout_blocks()) == 1 + 6


def test_mvn_methods_and_moments():
    par = vb.MVNParam(dim=3)
    required_methods(par)
    par['mean'].set(np.array([0.1, 0.2, 0.3]))
    info = np.array([[2.0, 0.3, 0.0], [0.3, 1.5, 0.2], [0.0, 0.2, 1.0]])
    par['info'].set(info)
    np.testing.assert_allclose(par.e(), [0.1, 0.2, 0.3])
    np.testing.assert_allclose(par.cov(), np.linalg.inv(info))
    np.testing.assert_allclose(par.e_outer(), np.outer(par.e(), par.e()) + np.linalg.inv(info))
    want = scipy.stats.multivariate_normal(mean=par.e(), cov=np.linalg.inv(info)).entropy()
    assert abs(par.entropy() - want) < 1e-12


def test_uvn_family_methods_and_moments():
    par = vb.UVNParam(min_info=0.1)
    required_methods(par)
    par['mean'].set(0.2); par['info'].set(1.2)
    assert abs(par.e() - 0.2) < 1e-15 and abs(par.var() - 1 / 1.2) < 1e-15
    assert abs(par.e_outer() - (0.2 ** 2 + 1 / 1.2)) < 1e-15
    mu, s2 = 0.2, 1 / 1.2
    assert abs(par.e_exp() - np.exp(mu + 0.5 * s2)) < 1e-14
    assert abs(par.var_exp() - (np.exp(s2) - 1) * np.exp(2 * mu + s2)) < 1e-13
    assert abs(par.entropy() - scipy.stats.norm(mu, np.sqrt(s2)).entropy()) < 1e-13
    vec = vb.UVNParamVector(length=3, min_info=0.1)
    required_methods(vec)
    vec['mean'].set(np.array([0.2, 0.5, -1.0])); vec['info'].set(np.array([1.2, 2.1, 0.7]))
    np.testing.assert_allclose(vec.var(), 1 / np.array([1.2, 2.1, 0.7]))
    np.testing.assert_allclose(vec.e_exp(), np.exp(vec.e() + 0.5 * vec.var()))
    assert abs(vec.entropy() - sum(scipy.stats.norm(0, np.sqrt(v)).entropy() for v in vec.var())) < 1e-12
    arr = vb.UVNParamArray(shape=(2, 3), min_info=0.1)
    required_methods(arr)
    arr['mean'].set(np.arange(6.0).reshape(2, 3) * 0.1); arr['info'].set(np.full((2, 3), 2.0))
    np.testing.assert_allclose(arr.e_outer(), arr.e() ** 2 + 0.5)
    mom = vb.UVNMomentParamArray(shape=(2, 3), min_info=0.1)
    required_methods(mom)
    mom.set_from_uvn_param_array(arr)
    np.testing.assert_allclose(mom.e(), arr.e())
    np.testing.assert_allclose(mom.var(), arr.var())
    np.testing.assert_allclose(mom.e_exp(), arr.e_exp())
    np.testing.assert_allclose(mom.entropy(), arr.entropy())


def test_gamma_dirichlet_wishart_methods_and_moments():
    g = vb.GammaParam(min_shape=0.1, min_rate=0.2)
    required_methods(g)
    g['shape'].set(3.0); g['rate'].set(2.0)
    assert abs(g.e() - 1.5) < 1e-15
    draws = scipy.stats.gamma(3.0, scale=0.5)
    assert abs(g.entropy() - draws.entropy()) < 1e-12
    assert abs(g.e_log() - (scipy.special.digamma(3.0) - np.log(2.0))) < 1e-14
    d = vb.DirichletParamArray(shape=(4, 2, 3))
    required_methods(d)
    alpha = np.random.default_rng(0).uniform(0.5, 3.0, size=(4, 2, 3))
    d['alpha'].set(alpha)
    np.testing.assert_allclose(d.e(), alpha / alpha.sum(0, keepdims=True))
    np.testing.assert_allclose(d.e().sum(0), np.ones((2, 3)))
    want_ent = np.array([[scipy.stats.dirichlet(alpha[:, i, j]).entropy() for j in range(3)] for i in range(2)])
    np.testing.assert_allclose(d.entropy(), want_ent, rtol=1e-12)
    w = vb.WishartParam(size=3, diag_lb=0.1)
    required_methods(w)
    v = np.array([[2.0, 0.3, 0.0], [0.3, 1.5, 0.2], [0.0, 0.2, 1.0]])
    w['df'].set(5.5); w['v'].set(v)
    np.testing.assert_allclose(w.e(), 5.5 * v)
    # the reference DEFINES e_inv as df * inv(v) (LRVB/WishartParams.py:22-23), not the inverse-Wishart mean
    np.testing.assert_allclose(w.e_inv(), 5.5 * np.linalg.inv(v))
    assert abs(w.entropy() - scipy.stats.wishart(df=5.5, scale=v).entropy()) < 1e-10


# ---- TestConstrainingFunctions ----------------------------------------------------------------------
@pytest.mark.parametrize('lb', LBS)
@pytest.mark.parametrize('ub', UBS)
def test_scalar_and_array_constraints(lb, ub):
    if lb >= ub:
        return
    free = np.array([-1.3, 0.0, 0.7])
    vec = vb.Parameters.constrain(free, lb, ub)
    assert np.all(vec >= lb) and np.all(vec <= ub)
    np.testing.assert_allclose(vb.Parameters.unconstrain_array(vec, lb, ub), free, atol=1e-12)
    for f, v in zip(free, vec):
        assert abs(vb.Parameters.unconstrain_scalar(v, lb, ub) - f) < 1e-12
    if np.isfinite(lb):
        with pytest.raises(ValueError):
            vb.Parameters.unconstrain_array(np.array([lb - 1.0]), lb, ub)
    if np.isfinite(ub):
        with pytest.raises(ValueError):
            vb.Parameters.unconstrain_scalar(ub + 1.0, lb, ub)


def test_simplex_matrix_constraint():
    rng = np.random.default_rng(0)
    free = rng.normal(size=(5, 3))
    z = vb.SimplexParams.constrain_simplex_matrix(free)
    assert z.shape == (5, 4)
    np.testing.assert_allclose(z.sum(axis=1), np.ones(5), atol=1e-14)
    np.testing.assert_allclose(vb.SimplexParams.unconstrain_simplex_matrix(z), free, atol=1e-12)
    np.testing.assert_allclose(vb.SimplexParams.constrain_simplex_vector(free[0]), z[0])


# ---- TestParameters ---------------------------------------------------------------------------------
@pytest.mark.parametrize('lb', LBS)
@pytest.mark.parametrize('ub', UBS)
def test_vector_array_scalar_param(lb, ub):
    if lb >= ub:
        with pytest.raises(ValueError):
            vb.VectorParam('bad', 3, lb=lb, ub=ub)
        return
    k = 4
    val = np.array([inbounds(lb, ub)] * k) + (0.01 * np.arange(k) if not (np.isfinite(lb) and np.isfinite(ub)) else 0.0)
    vp = vb.VectorParam('test', k, lb=lb, ub=ub)
    vp.set(val)
    np.testing.assert_allclose(vp.get(), val)
    free = vp.get_free()
    vp.set(np.full(k, inbounds(lb, ub)))
    vp.set_free(free)
    np.testing.assert_allclose(vp.get(), val, atol=1e-12)
    vec = vp.get_vector()
    vp.set_vector(vec)
    np.testing.assert_allclose(vp.get(), val, atol=1e-15)
    with pytest.raises(ValueError):
        vp.set(val[:-1])
    with pytest.raises(ValueError):
        vp.set_free(free[:-1])
    with pytest.raises(ValueError):
        vp.set_vector(vec[:-1])
    # as in the reference, `set` does not police the bounds (LRVB/Parameters.py:181-186); the
    # unconstraining map does (Parameters.py:15-28), so an out-of-bounds value surfaces at get_free
    if np.isfinite(lb):
        vp.set(np.full(k, lb - 1.0))
        with pytest.raises(ValueError):
            vp.get_free()
    if np.isfinite(ub):
        vp.set(np.full(k, ub + 1.0))
        with pytest.raises(ValueError):
            vp.get_free()
    vp.set(val)
    ap = vb.ArrayParam('test', shape=(2, 2), lb=lb, ub=ub)
    ap.set(val.reshape(2, 2))
    np.testing.assert_allclose(ap.get_vector(), val)            # C-order flattening
    free = ap.get_free()
    ap.set_free(free)
    np.testing.assert_allclose(ap.get(), val.reshape(2, 2), atol=1e-12)
    with pytest.raises(ValueError):
        ap.set(val)                                              # wrong shape
    sp = vb.ScalarParam('test', lb=lb, ub=ub)
    sp.set(val[0])
    assert abs(sp.get() - val[0]) < 1e-15
    sp.set_free(sp.get_free())
    assert abs(sp.get() - val[0]) < 1e-12
    assert sp.free_size() == 1 and sp.vector_size() == 1
    check_sparse_transforms(vp)
    check_sparse_transforms(ap)
    check_sparse_transforms(sp)


def test_simplex_param():
    rng = np.random.default_rng(1)
    sp = vb.SimplexParam('s', shape=(3, 4))
    val = rng.dirichlet(np.ones(4), size=3)
    sp.set(val)
    np.testing.assert_allclose(sp.get(), val)
    free = sp.get_free()
    assert free.size == 9 and sp.free_shape() == (3, 3)
    sp.set(np.full((3, 4), 0.25))
    sp.set_free(free)
    np.testing.assert_allclose(sp.get(), val, atol=1e-12)
    vec = sp.get_vector()
    np.testing.assert_allclose(vec, val.ravel())
    sp.set_vector(vec)
    np.testing.assert_allclose(sp.get(), val, atol=1e-15)
    with pytest.raises(ValueError):
        sp.set(val[:, :3])
    with pytest.raises(ValueError):
        sp.set_free(free[:-1])
    np.testing.assert_array_equal(sp.get_vector_indices(1), np.arange(4, 8))
    check_sparse_transforms(sp)
    # closed forms from the moments agree with the assembled sparse versions
    z = val[0]
    jac = vb.SimplexParams.constrain_grad_from_moment(z)
    np.testing.assert_allclose(np.asarray(sp.free_to_vector_jac(free).todense())[:4, :3], jac, atol=1e-12)
    hess = vb.SimplexParams.constrain_hess_from_moment(z)
    for k in range(4):
        np.testing.assert_allclose(np.asarray(sp.free_to_vector_hess(free)[k].todense())[:3, :3], hess[k], atol=1e-12)


def test_ld_matrix_helpers_and_pos_def_param():
    mat = np.array([[2.0, 0.0, 0.0], [0.5, 1.5, 0.0], [0.2, -0.3, 1.0]])
    vec = vb.MatrixParameters.vectorize_ld_matrix(mat)
    np.testing.assert_allclose(vec, [2.0, 0.5, 1.5, 0.2, -0.3, 1.0])     # row-major lower triangle
    np.testing.assert_allclose(vb.MatrixParameters.unvectorize_ld_matrix(vec), mat)
    for k1 in range(3):
        for k2 in range(k1 + 1):
            assert vec[vb.MatrixParameters.SymIndex(k1, k2)] == mat[k1, k2]
            assert vb.MatrixParameters.SymIndex(k2, k1) == vb.MatrixParameters.SymIndex(k1, k2)
    a = mat @ mat.T
    free = vb.MatrixParameters.pack_posdef_matrix(a)
    np.testing.assert_allclose(vb.MatrixParameters.unpack_posdef_matrix(free), a, atol=1e-12)
    free_lb = vb.MatrixParameters.pack_posdef_matrix(a, diag_lb=0.3)
    np.testing.assert_allclose(vb.MatrixParameters.unpack_posdef_matrix(free_lb, diag_lb=0.3), a, atol=1e-12)
    pd = vb.PosDefMatrixParam('m', 3)
    pd.set(a)
    np.testing.assert_allclose(pd.get(), a)
    f = pd.get_free()
    pd.set(np.eye(3))
    pd.set_free(f)
    np.testing.assert_allclose(pd.get(), a, atol=1e-12)
    v = pd.get_vector()
    assert v.size == 6
    pd.set(np.eye(3))
    pd.set_vector(v)
    np.testing.assert_allclose(pd.get(), a, atol=1e-15)
    with pytest.raises(ValueError):
        pd.set(np.eye(4))
    with pytest.raises(ValueError):
        pd.set(mat)                                              # not symmetric
    with pytest.raises(ValueError):
        pd.set_free(f[:-1])
    check_sparse_transforms(pd)
    check_sparse_transforms(vb.PosDefMatrixParam('m', 2, diag_lb=0.2))


# ---- TestParameterDictionary -------------------------------------------------------------------------
def make_dict():
    mp = vb.ModelParamsDict('dict')
    mp.push_param(vb.ScalarParam('scalar', lb=0.5, val=1.5))
    mp.push_param(vb.VectorParam('vector', 3, lb=-1.0, ub=4.0, val=np.array([0.0, 1.0, 2.0])))
    mp.push_param(vb.PosDefMatrixParam('mat', 2, val=np.array([[2.0, 0.4], [0.4, 1.0]])))
    mp.push_param(vb.SimplexParam('simplex', (2, 3)))
    mp.push_param(vb.ArrayParam('array', (2, 2), ub=3.0, val=np.array([[0.1, 0.2], [0.3, 0.4]])))
    return mp


def test_model_params_dict_layout_and_errors():
    mp = make_dict()
    required_methods(mp, sparse=False)
    assert mp.free_size() == 1 + 3 + 3 + 4 + 4 and mp.vector_size() == 1 + 3 + 3 + 6 + 4
    # insertion order; every member owns a contiguous range of both vectors
    assert list(mp.free_indices_dict['scalar']) == [0] and list(mp.free_indices_dict['vector']) == [1, 2, 3]
    assert list(mp.vector_indices_dict['simplex']) == list(range(7, 13))
    free = mp.get_free()
    np.testing.assert_allclose(free[mp.free_indices_dict['vector']], mp['vector'].get_free())
    vec = mp.get_vector()
    np.testing.assert_allclose(vec[mp.vector_indices_dict['mat']], mp['mat'].get_vector())
    other = make_dict()
    rng = np.random.default_rng(3)
    new_free = rng.normal(size=mp.free_size()) * 0.3
    other.set_free(new_free)
    mp.set_vector(other.get_vector())
    np.testing.assert_allclose(mp.get_free(), new_free, atol=1e-10)
    np.testing.assert_allclose(mp['array'].get(), other['array'].get(), atol=1e-14)
    with pytest.raises(ValueError):
        mp.set_free(new_free[:-1])
    with pytest.raises(ValueError):
        mp.set_vector(vec[:-1])
    # values survive a deep copy independently (the sensitivity classes rely on it)
    cp = copy.deepcopy(mp)
    cp['scalar'].set(9.0)
    assert mp['scalar'].get() != 9.0
    # dictval mirrors the structure
    dv = mp.dictval()
    assert set(dv.keys()) == {'scalar', 'vector', 'mat', 'simplex', 'array'}


def test_model_params_dict_sparse_transforms():
    check_sparse_transforms(make_dict())


# ---- test_sparse_free_hessians ----------------------------------------------------------------------
def test_convert_vector_to_free_hessian_against_finite_differences():
    mp = make_dict()
    rng = np.random.default_rng(5)
    V = mp.vector_size()
    A = rng.normal(size=(V, V)); A = A + A.T
    b = rng.normal(size=V)

    def f_vec(v):
        return 0.5 * v @ A @ v + b @ v + np.sum(np.sin(v))

    def f_free(free):
        mp.set_free(free)
        return f_vec(np.array(mp.get_vector()))

    free = rng.normal(size=mp.free_size()) * 0.3
    mp.set_free(free)
    v = np.array(mp.get_vector())
    g_vec = A @ v + b + np.cos(v)
    H_vec = A - np.diag(np.sin(v))
    H_free = vb.Parameters.convert_vector_to_free_hessian(mp, free, g_vec, H_vec)
    H_free = np.asarray(H_free.todense()) if scipy.sparse.issparse(H_free) else np.asarray(H_free)

    def grad_free(fr):
        return fd_jacobian(f_free, fr, h=1e-3)

    H_fd = fd_jacobian(grad_free, free, h=1e-3)
    np.testing.assert_allclose(H_free, H_fd, rtol=1e-5, atol=1e-5 * np.max(np.abs(H_fd)))
    np.testing.assert_allclose(H_free, H_free.T, atol=1e-12)
