"""Parameter packing API: structured value <-> flat constrained "vector" <-> flat unconstrained
"free" vector, with sparse Jacobians / Hessians of the free -> vector map.

Host-side mirror of the reference protocol (LRVB/test_variational_bayes.py:74-106 lists the
required methods): get/set, get_free/set_free, get_vector/set_vector, free_size/vector_size,
free_to_vector, free_to_vector_jac, free_to_vector_hess, names, dictval, __str__.

  box       LRVB/Parameters.py:31-61, 82-322          ScalarParam, VectorParam, ArrayParam
  psd       LRVB/MatrixParameters.py:16-198          PosDefMatrixParam
  simplex   LRVB/SimplexParams.py:11-175              SimplexParam
  dict      LRVB/ParameterDictionary.py:19-124        ModelParamsDict

Where the reference asks autograd for the derivative of a constraint (one scalar closure per
element, LRVB/Parameters.py:200-218) this module uses the closed forms; the O(N) arithmetic of
an objective never runs here -- it runs in liblrvb_hip.so, which receives the layout through
`layout_blocks()`.
"""
import copy
import math
import numbers
from collections import OrderedDict

import numpy as np
from copy import deepcopy
from scipy.sparse import coo_matrix, block_diag

from . import _hip

_INF = float('inf')


# ------------------------------------------------------------------------------ box maps
def constrain(free_vec, lb, ub):
    """LRVB/Parameters.py:47-61.  The two-sided case uses the overflow-free logistic (the
    reference's exp(f)/(1+exp(f)) returns nan for f > 709)."""
    if ub <= lb:
        raise ValueError('Upper bound must be greater than lower bound')
    f = np.asarray(free_vec, dtype=np.float64)
    if ub == _INF:
        return f.copy() if lb == -_INF else np.exp(f) + lb
    if lb == -_INF:
        return ub - np.exp(-f)
    ef = np.exp(-np.abs(f))
    return (ub - lb) * np.where(f >= 0, 1.0 / (1.0 + ef), ef / (1.0 + ef)) + lb


def _box_d1_d2(f, lb, ub):
    f = np.asarray(f, dtype=np.float64)
    if ub == _INF and lb == -_INF:
        return np.ones_like(f), np.zeros_like(f)
    if ub == _INF:
        e = np.exp(f)
        return e, e
    if lb == -_INF:
        e = np.exp(-f)
        return e, -e
    ef = np.exp(-np.abs(f))
    s = np.where(f >= 0, 1.0 / (1.0 + ef), ef / (1.0 + ef))
    sp = (ub - lb) * s * (1.0 - s)
    return sp, sp * (1.0 - 2.0 * s)


def unconstrain(vec, lb, ub):
    """LRVB/Parameters.py:31-44."""
    if ub <= lb:
        raise ValueError('Upper bound must be greater than lower bound')
    v = np.asarray(vec, dtype=np.float64)
    if ub == _INF:
        return v.copy() if lb == -_INF else np.log(v - lb)
    if lb == -_INF:
        return -np.log(ub - v)
    return np.log(v - lb) - np.log(ub - v)


def unconstrain_array(vec, lb, ub):
    """LRVB/Parameters.py:15-20."""
    vec = np.asarray(vec)
    if not (vec <= ub).all():
        raise ValueError('Elements larger than the upper bound')
    if not (vec >= lb).all():
        raise ValueError('Elements smaller than the lower bound')
    return unconstrain(vec, lb, ub).flatten()


def unconstrain_scalar(val, lb, ub):
    """LRVB/Parameters.py:23-28."""
    if not val <= ub:
        raise ValueError('Value larger than the upper bound')
    if not val >= lb:
        raise ValueError('Value smaller than the lower bound')
    return unconstrain(val, lb, ub)


def get_inbounds_value(lb, ub):
    """LRVB/Parameters.py:66-79 (note: 0.5*(ub-lb) for two-sided bounds, not the midpoint)."""
    assert lb < ub
    if lb > -_INF and ub < _INF:
        return 0.5 * (ub - lb)
    if lb > -_INF:
        return lb + 1.0
    if ub < _INF:
        return ub - 1.0
    return 0.0


def _check_bounds(lb, ub):
    assert lb >= -_INF
    assert ub <= _INF
    if lb >= ub:
        raise ValueError('Upper bound must strictly exceed lower bound')


class _BoxParam(object):
    """Shared machinery of the three elementwise-constrained parameter types."""

    def _init_box(self, name, lb, ub):
        self.name = name
        _check_bounds(lb, ub)
        self._lb = lb
        self._ub = ub

    def layout_blocks(self):
        n = self.free_size()
        return [dict(kind=_hip.BLOCK_BOX, free_size=n, vec_size=n, dim0=n, dim1=0, lb=self._lb, ub=self._ub)]

    def free_to_vector(self, free_val):
        self.set_free(free_val)
        return self.get_vector()

    def free_to_vector_jac(self, free_val):
        d1, _ = _box_d1_d2(np.asarray(free_val, dtype=np.float64).ravel(), self._lb, self._ub)
        idx = np.arange(self.vector_size())
        return coo_matrix((d1, (idx, idx)), (self.vector_size(), self.free_size()))

    def free_to_vector_hess(self, free_val):
        _, d2 = _box_d1_d2(np.asarray(free_val, dtype=np.float64).ravel(), self._lb, self._ub)
        shape = (self.free_size(), self.free_size())
        return [coo_matrix(([d2[k]], ([k], [k])), shape) for k in range(self.vector_size())]


class ScalarParam(_BoxParam):
    """LRVB/Parameters.py:82-150."""

    def __init__(self, name='', lb=-_INF, ub=_INF, val=None):
        if lb >= ub:
            raise ValueError('Upper bound must strictly exceed lower bound')
        self._init_box(name, lb, ub)
        self.set(val if val is not None else get_inbounds_value(lb, ub))

    def __str__(self):
        return self.name + ': ' + str(self._val)

    def names(self):
        return [self.name]

    def dictval(self):
        return self._val if isinstance(self._val, numbers.Number) else np.asarray(self._val).tolist()

    def set(self, val):
        self._val = val

    def get(self):
        return self._val

    def set_free(self, free_val):
        self.set(constrain(free_val, self._lb, self._ub))

    def get_free(self):
        return np.reshape(unconstrain_scalar(self._val, self._lb, self._ub), 1)

    def set_vector(self, val):
        self.set(val)

    def get_vector(self):
        return np.reshape(self._val, 1)

    def size(self):
        return 1

    def free_size(self):
        return 1

    def vector_size(self):
        return 1


class VectorParam(_BoxParam):
    """LRVB/Parameters.py:154-231."""

    def __init__(self, name='', size=1, lb=-_INF, ub=_INF, val=None):
        self._size = int(size)
        self._init_box(name, lb, ub)
        self.set(val if val is not None else np.full(self._size, get_inbounds_value(lb, ub)))

    def __str__(self):
        return self.name + ':\n' + str(self._val)

    def names(self):
        return [self.name + '_' + str(k) for k in range(self.size())]

    def dictval(self):
        return self._val.tolist()

    def set(self, val):
        if val.size != self.size():
            raise ValueError('Wrong size for vector ' + self.name + '.  Expected: ' + str(self.size()) +
                             ', got ' + str(val.size))
        self._val = val

    def get(self):
        return self._val

    def set_free(self, free_val):
        if free_val.size != self.size():
            raise ValueError('Wrong size for vector ' + self.name)
        self.set(constrain(free_val, self._lb, self._ub))

    def get_free(self):
        return unconstrain_array(self._val, self._lb, self._ub)

    def set_vector(self, val):
        self.set(val)

    def get_vector(self):
        return self._val

    def size(self):
        return self._size

    def free_size(self):
        return self._size

    def vector_size(self):
        return self._size


class HyperVectorParam(VectorParam):
    """The VectorParam a device objective creates for its per-observation weights (and its tilt): same interface, but the
    value is a PRIVATE, READ-ONLY copy and every `set` / `set_vector` / `set_free` stamps a new `version`.  The objective
    keeps the vector resident in HBM and has to know before every evaluation whether that copy is current: comparing
    N = 1e6 weights on the host cost 0.3-0.9 ms per call -- as much as a whole step of configurations 2 and 4 -- where
    comparing two version numbers costs nothing.  Writing into the array `get()` returns raises (numpy: "assignment
    destination is read-only") instead of silently leaving the device copy stale; a plain VectorParam assigned to
    `weights_par` by the caller is still honoured, by the full comparison."""
    _stamp = [0]

    def set(self, val):
        val = np.array(val, dtype=np.float64)                 # private copy
        if val.size != self.size():
            raise ValueError('Wrong size for vector ' + self.name + '.  Expected: ' + str(self.size()) +
                             ', got ' + str(val.size))
        val = val.reshape(self._size)
        val.flags.writeable = False
        self._val = val
        HyperVectorParam._stamp[0] += 1
        self.version = HyperVectorParam._stamp[0]

    def __deepcopy__(self, memo):
        """A copy is a parameter of its own: a fresh version stamp and a frozen private array (copy.deepcopy would
        otherwise hand out the original's stamp with a WRITEABLE array, and an in-place write to the clone would go
        unnoticed by `ResidentVector`)."""
        clone = self.__class__.__new__(self.__class__)
        memo[id(self)] = clone
        for k, v in self.__dict__.items():
            if k not in ('_val', 'version'):
                setattr(clone, k, deepcopy(v, memo))
        clone.set(self._val)
        return clone

    def __setstate__(self, state):
        self.__dict__.update(state)
        if '_val' in state:
            self.set(state['_val'])                              # re-freeze and re-stamp after unpickling


class ResidentVector(object):
    """Whether the device copy of a hyper-parameter vector is current: `changed(par)` returns the vector when it has to
    be uploaded (first use, a new `version`, or -- for parameters without versions -- different contents), else None;
    `key` identifies the current contents cheaply (for result memos)."""

    def __init__(self):
        self.key = None
        self._val = None

    def changed(self, par):
        w = np.asarray(par.get_vector(), dtype=np.float64)
        version = getattr(par, 'version', None)
        if version is not None:
            if self.key == ('v', version):
                return None
            self.key, self._val = ('v', version), None
            return w
        if self._val is not None and np.array_equal(w, self._val):
            return None
        self._val = w.copy()
        self.key = ('b', self._val.tobytes())
        return w


class ArrayParam(_BoxParam):
    """LRVB/Parameters.py:234-322 (C-order flattening)."""

    def __init__(self, name='', shape=(1, 1), lb=-_INF, ub=_INF, val=None):
        self._shape = tuple(shape)
        self._init_box(name, lb, ub)
        self.set(val if val is not None else np.full(self._shape, get_inbounds_value(lb, ub)))

    def __str__(self):
        return self.name + ':\n' + str(self._val)

    def names(self):
        return self.name

    def dictval(self):
        return self._val.tolist()

    def set(self, val):
        if val.shape != self.shape():
            raise ValueError('Wrong size for array ' + self.name + ' Expected shape: ' + str(self.shape()) +
                             ' Got shape: ' + str(val.shape))
        self._val = val

    def get(self):
        return self._val

    def set_free(self, free_val):
        if free_val.size != self.free_size():
            raise ValueError('Wrong size for array {}.  Expected {}, got {}'.format(
                self.name, str(self.free_size()), str(free_val.size)))
        self.set(constrain(free_val, self._lb, self._ub).reshape(self._shape))

    def get_free(self):
        return unconstrain_array(self._val, self._lb, self._ub)

    def set_vector(self, val):
        if val.size != self.vector_size():
            raise ValueError('Wrong size for array {}.  Expected {}, got {}'.format(
                self.name, str(self.vector_size()), str(val.size)))
        self.set(val.reshape(self._shape))

    def get_vector(self):
        return self._val.flatten()

    def shape(self):
        return self._shape

    def free_size(self):
        return int(np.prod(self._shape))

    def vector_size(self):
        return int(np.prod(self._shape))


# ------------------------------------------------------------------------------ psd maps
_TRIL_CACHE = {}


def tril_indices(k):
    """np.tril_indices(k), computed once per k (numpy builds two k x k index grids per call: ~50 us at k = 21, which was a
    third of configuration 2's host time per step)."""
    k = int(k)
    hit = _TRIL_CACHE.get(k)
    if hit is None:
        r, c = np.tril_indices(k)
        r.flags.writeable = False
        c.flags.writeable = False
        hit = _TRIL_CACHE[k] = (r, c)
    return hit


def SymIndex(k1, k2):
    """Index of (k1, k2) in the row-major lower-triangle vector.  MatrixParameters.py:16-23."""
    a, b = (k1, k2) if k2 <= k1 else (k2, k1)
    return int(b + a * (a + 1) // 2)


def vectorize_ld_matrix(mat):
    nrow, ncol = np.shape(mat)
    if nrow != ncol:
        raise ValueError('mat must be square')
    return mat[tril_indices(nrow)]


def _ld_size_to_dim(n):
    k = int(0.5 * (math.sqrt(1 + 8 * n) - 1))
    if k * (k + 1) // 2 != n:
        raise ValueError('Vector is an impossible size')
    return k


def unvectorize_ld_matrix(vec):
    vec = np.asarray(vec)
    k = _ld_size_to_dim(vec.size)
    mat = np.zeros((k, k))
    mat[tril_indices(k)] = vec
    return mat


def unvectorize_symmetric_matrix(vec_val):
    ld = unvectorize_ld_matrix(vec_val)
    return ld + ld.T - np.diag(np.diag(ld))


def exp_matrix_diagonal(mat):
    assert mat.shape[0] == mat.shape[1]
    out = np.array(mat, dtype=np.float64)
    d = np.arange(mat.shape[0])
    out[d, d] = np.exp(out[d, d])
    return out


def log_matrix_diagonal(mat):
    assert mat.shape[0] == mat.shape[1]
    out = np.array(mat, dtype=np.float64)
    d = np.arange(mat.shape[0])
    out[d, d] = np.log(out[d, d])
    return out


def pack_posdef_matrix(mat, diag_lb=0.0):
    """MatrixParameters.py:101-105."""
    k = mat.shape[0]
    return vectorize_ld_matrix(log_matrix_diagonal(np.linalg.cholesky(mat - diag_lb * np.eye(k))))


def unpack_posdef_matrix(free_vec, diag_lb=0.0):
    """MatrixParameters.py:108-112."""
    chol = exp_matrix_diagonal(unvectorize_ld_matrix(free_vec))
    return chol @ chol.T + diag_lb * np.eye(chol.shape[0])


def pos_def_matrix_free_to_vector(free_val, diag_lb=0.0):
    """Lower triangle of the matrix a free vector encodes.  LRVB/MatrixParameters.py:126-128."""
    return vectorize_ld_matrix(unpack_posdef_matrix(free_val, diag_lb=diag_lb))


def pos_def_matrix_free_to_vector_jac(free_val, diag_lb=0.0):
    """Closed-form Jacobian of `pos_def_matrix_free_to_vector` (the reference differentiates it with
    autograd, LRVB/MatrixParameters.py:130-131); diag_lb only shifts the diagonal."""
    free_val = np.asarray(free_val, dtype=np.float64)
    return _psd_jac_dense(free_val, _ld_size_to_dim(free_val.size))


def pos_def_matrix_free_to_vector_hess(free_val, diag_lb=0.0):
    """Closed-form Hessian, indexed [vec, free, free].  LRVB/MatrixParameters.py:132-133."""
    free_val = np.asarray(free_val, dtype=np.float64)
    return _psd_hess_dense(free_val, _ld_size_to_dim(free_val.size))


def _psd_jac_dense(free_val, k):
    """d vec(A)_(ij) / d f_(ab) = dL_ab (d_ia L_jb + d_ja L_ib)."""
    L = exp_matrix_diagonal(unvectorize_ld_matrix(free_val))
    r, c = np.tril_indices(k)
    m = r.size
    dL = np.where(r == c, L[r, r], 1.0)                       # per column (a, b)
    i, j = r[:, None], c[:, None]                              # rows
    a, b = r[None, :], c[None, :]                              # cols
    J = (i == a) * L[j, b] + (j == a) * L[i, b]
    return J * dL[None, :]


def _psd_hess_dense(free_val, k):
    """d2 vec(A)_(ij) / d f_(ab) d f_(cd), shape (m, m, m)."""
    L = exp_matrix_diagonal(unvectorize_ld_matrix(free_val))
    r, c = np.tril_indices(k)
    m = r.size
    dL = np.where(r == c, L[r, r], 1.0)
    H = np.zeros((m, m, m))
    for row in range(m):
        i, j = r[row], c[row]
        for p in range(m):
            a, b = r[p], c[p]
            for q in range(m):
                cc, d = r[q], c[q]
                v = 0.0
                if b == d:
                    v += dL[p] * dL[q] * (float(i == a and j == cc) + float(j == a and i == cc))
                if p == q and a == b:
                    v += L[a, a] * (float(i == a) * L[j, a] + float(j == a) * L[i, a])
                H[row, p, q] = v
    return H


class PosDefMatrixParam(object):
    """LRVB/MatrixParameters.py:137-198."""

    def __init__(self, name='', size=2, diag_lb=0.0, val=None):
        self.name = name
        self._size = int(size)
        self._vec_size = self._size * (self._size + 1) // 2
        self._diag_lb = diag_lb
        assert diag_lb >= 0
        if val is None:
            self._val = np.diag(np.full(self._size, diag_lb + 1.0))
        else:
            self.set(val)

    def layout_blocks(self):
        return [dict(kind=_hip.BLOCK_PSD, free_size=self._vec_size, vec_size=self._vec_size,
                     dim0=self._size, dim1=0, lb=self._diag_lb, ub=_INF)]

    def __str__(self):
        return self.name + ':\n' + str(self._val)

    def names(self):
        return [self.name]

    def dictval(self):
        return self._val.tolist()

    def set(self, val):
        nrow, ncol = np.shape(val)
        if nrow != self._size or ncol != self._size:
            raise ValueError('Matrix is a different size')
        if not (val.transpose() == val).all():
            raise ValueError('Matrix is not symmetric')
        self._val = val

    def get(self):
        return self._val

    def set_free(self, free_val):
        if free_val.size != self._vec_size:
            raise ValueError('Free value is the wrong length')
        mat = unpack_posdef_matrix(free_val, diag_lb=self._diag_lb)
        self.set(0.5 * (mat + mat.T))

    def get_free(self):
        return pack_posdef_matrix(self._val, diag_lb=self._diag_lb)

    def free_to_vector(self, free_val):
        self.set_free(free_val)
        return self.get_vector()

    def free_to_vector_jac(self, free_val):
        return coo_matrix(_psd_jac_dense(free_val, self._size))

    def free_to_vector_hess(self, free_val):
        H = _psd_hess_dense(free_val, self._size)
        return [coo_matrix(H[k]) for k in range(H.shape[0])]

    def set_vector(self, vec_val):
        if vec_val.size != self._vec_size:
            raise ValueError('Vector value is the wrong length')
        self.set(unvectorize_symmetric_matrix(vec_val))

    def get_vector(self):
        return vectorize_ld_matrix(self._val)

    def size(self):
        return self._size

    def free_size(self):
        return self._vec_size

    def vector_size(self):
        return self._vec_size


class PosDefMatrixParamArray(object):
    """An array of positive definite matrices; the last two indices are the matrix indices
    (LRVB/MatrixParameters.py:311-481).  Free and vector forms are the per-matrix forms of
    PosDefMatrixParam stacked in C order of the array index, so the packing Jacobian / Hessians are
    block diagonal; on the device every matrix is one log-Cholesky block."""

    def __init__(self, name='', array_shape=(1,), matrix_size=2, diag_lb=0.0, val=None):
        self.name = name
        self._matrix_size = int(matrix_size)
        self._array_shape = tuple(int(t) for t in np.atleast_1d(array_shape))
        self._array_ranges = [range(0, t) for t in self._array_shape]
        self._array_length = int(np.prod(self._array_shape))
        self._shape = self._array_shape + (self._matrix_size, self._matrix_size)
        self._vec_size = self._matrix_size * (self._matrix_size + 1) // 2
        self._diag_lb = diag_lb
        assert diag_lb >= 0
        if val is None:
            default_val = np.diag(np.full(self._matrix_size, diag_lb + 1.0))
            self._val = np.broadcast_to(default_val, self._shape)
        else:
            self.set(val)

    def layout_blocks(self):
        return [dict(kind=_hip.BLOCK_PSD, free_size=self._vec_size, vec_size=self._vec_size,
                     dim0=self._matrix_size, dim1=0, lb=self._diag_lb, ub=_INF)
                for _ in range(self._array_length)]

    def __str__(self):
        return self.name + ':\n' + str(self._val)

    def names(self):
        return [self.name]

    def dictval(self):
        return np.asarray(self._val).tolist()

    def set(self, val):
        # (the reference only rejects a value whose EVERY dimension differs, :349-354; any mismatch is rejected here)
        if tuple(np.shape(val)) != self._shape:
            raise ValueError('Array is the wrong size')
        self._val = val

    def get(self):
        return self._val

    def stacked_obs_slice(self, obs):
        """Positions in the free / vector form of the matrix at array index `obs` (a tuple)."""
        assert len(obs) == len(self._array_shape)
        start = int(np.ravel_multi_index(obs, self._array_shape)) * self._vec_size
        return slice(start, start + self._vec_size)

    def _matrices(self):
        return np.reshape(np.asarray(self._val), (self._array_length, self._matrix_size, self._matrix_size))

    def set_free(self, free_val):
        free_val = np.asarray(free_val)
        if free_val.size != self.free_size():
            raise ValueError('Free value is the wrong length')
        blocks = np.reshape(free_val, (self._array_length, self._vec_size))
        self._val = np.reshape(np.array([unpack_posdef_matrix(b, diag_lb=self._diag_lb) for b in blocks]), self._shape)

    def get_free(self):
        return np.hstack([pack_posdef_matrix(m, diag_lb=self._diag_lb) for m in self._matrices()])

    def set_vector(self, vec_val):
        vec_val = np.asarray(vec_val)
        if vec_val.size != self.vector_size():
            raise ValueError('Vector value is the wrong length')
        blocks = np.reshape(vec_val, (self._array_length, self._vec_size))
        self._val = np.reshape(np.array([unvectorize_symmetric_matrix(b) for b in blocks]), self._shape)

    def get_vector(self):
        return np.hstack([vectorize_ld_matrix(m) for m in self._matrices()])

    def apply_matrix_function(self, mat_func):
        out = np.array([mat_func(m) for m in self._matrices()])
        return np.reshape(out, self._array_shape + out.shape[1:])

    def free_to_vector(self, free_val):
        self.set_free(free_val)
        return self.get_vector()

    def free_to_vector_jac(self, free_val):
        blocks = np.reshape(np.asarray(free_val, dtype=np.float64), (self._array_length, self._vec_size))
        return block_diag([_psd_jac_dense(b, self._matrix_size) for b in blocks], format='coo')

    def free_to_vector_hess(self, free_val):
        blocks = np.reshape(np.asarray(free_val, dtype=np.float64), (self._array_length, self._vec_size))
        n, v = self.free_size(), self._vec_size
        hessians = []
        for a, b in enumerate(blocks):
            H = _psd_hess_dense(b, self._matrix_size)
            rows, cols = np.meshgrid(np.arange(a * v, (a + 1) * v), np.arange(a * v, (a + 1) * v), indexing='ij')
            for k in range(v):
                hessians.append(coo_matrix((H[k].ravel(), (rows.ravel(), cols.ravel())), (n, n)))
        return hessians

    def matrix_size(self):
        return self._matrix_size

    def length(self):
        return self._array_length

    def get_array_ranges(self):
        return self._array_ranges

    def free_size(self):
        return self._vec_size * self._array_length

    def vector_size(self):
        return self._vec_size * self._array_length


class PosDefMatrixParamVector(PosDefMatrixParamArray):
    """A vector of positive definite matrices, first index = which matrix
    (LRVB/MatrixParameters.py:201-307)."""

    def __init__(self, name='', length=1, matrix_size=2, diag_lb=0.0, val=None):
        super().__init__(name=name, array_shape=(int(length),), matrix_size=matrix_size, diag_lb=diag_lb, val=val)

    def free_obs_slice(self, obs):
        assert obs < self._array_length
        return slice(self._vec_size * obs, self._vec_size * (obs + 1))


# ---------------------------------------------------------------------------------- simplex
def constrain_simplex_matrix(free_mat):
    """LRVB/SimplexParams.py:11-18."""
    free_mat = np.asarray(free_mat, dtype=np.float64)
    aug = np.hstack([np.zeros((free_mat.shape[0], 1)), free_mat])
    aug = aug - aug.max(axis=1, keepdims=True)
    e = np.exp(aug)
    return e / e.sum(axis=1, keepdims=True)


def unconstrain_simplex_matrix(simplex_mat):
    """LRVB/SimplexParams.py:21-23."""
    return np.log(simplex_mat[:, 1:]) - np.log(simplex_mat[:, :1])


def constrain_simplex_vector(free_vec):
    return constrain_simplex_matrix(np.expand_dims(free_vec, 0)).flatten()


def constrain_grad_from_moment(z):
    """J[k, j] = z_k (d_{k,j+1} - z_{j+1}).  LRVB/SimplexParams.py:33-38."""
    z = np.asarray(z, dtype=np.float64)
    K = z.size
    J = -np.outer(z, z[1:])
    J[np.arange(1, K), np.arange(K - 1)] += z[1:]
    return J


def constrain_hess_from_moment(z):
    """H[k, i, j] = z_k[(d_{k,i+1} - z_{i+1})(d_{k,j+1} - z_{j+1}) - z_{i+1}(d_ij - z_{j+1})].
    Same tensor as LRVB/SimplexParams.py:42-63 (doc/simplex_derivatives.lyx)."""
    z = np.asarray(z, dtype=np.float64)
    K = z.size
    q = z[1:]
    E = np.zeros((K, K - 1))
    E[np.arange(1, K), np.arange(K - 1)] = 1.0
    Dm = E - q[None, :]
    common = np.diag(q) - np.outer(q, q)
    return z[:, None, None] * (Dm[:, :, None] * Dm[:, None, :] - common[None, :, :])


class SimplexParam(object):
    """Rows on the simplex.  LRVB/SimplexParams.py:69-175."""

    def __init__(self, name='', shape=(1, 2), val=None):
        self.name = name
        self._shape = tuple(shape)
        self._free_shape = (shape[0], shape[1] - 1)
        self.set(val if val is not None else np.full(shape, 1. / shape[1]))

    def layout_blocks(self):
        return [dict(kind=_hip.BLOCK_SIMPLEX, free_size=self.free_size(), vec_size=self.vector_size(),
                     dim0=self._shape[0], dim1=self._shape[1], lb=0.0, ub=1.0)]

    def __str__(self):
        return self.name + ': ' + str(self._val)

    def names(self):
        return [self.name]

    def dictval(self):
        return self._val.tolist()

    def set(self, val):
        if val.shape != self._shape:
            raise ValueError('Wrong shape for SimplexParam ' + self.name)
        self._val = val

    def get(self):
        return self._val

    def set_free(self, free_val):
        if len(free_val) != self.free_size():
            raise ValueError('Wrong free size for SimplexParam ' + self.name)
        self.set(constrain_simplex_matrix(np.reshape(free_val, self._free_shape)))

    def get_free(self):
        return unconstrain_simplex_matrix(self._val).flatten()

    def free_to_vector(self, free_val):
        self.set_free(free_val)
        return self.get_vector()

    def free_to_vector_jac(self, free_val):
        n, K = self._shape
        P = constrain_simplex_matrix(np.reshape(free_val, self._free_shape))
        rows, cols, vals = [], [], []
        kk, jj = np.meshgrid(np.arange(K), np.arange(K - 1), indexing='ij')
        for r in range(n):
            J = constrain_grad_from_moment(P[r])
            rows.append((r * K + kk).ravel())
            cols.append((r * (K - 1) + jj).ravel())
            vals.append(J.ravel())
        return coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                          (self.vector_size(), self.free_size()))

    def free_to_vector_hess(self, free_val):
        n, K = self._shape
        P = constrain_simplex_matrix(np.reshape(free_val, self._free_shape))
        shape = (self.free_size(), self.free_size())
        ii, jj = np.meshgrid(np.arange(K - 1), np.arange(K - 1), indexing='ij')
        out = []
        for r in range(n):
            H = constrain_hess_from_moment(P[r])
            base = r * (K - 1)
            for k in range(K):
                out.append(coo_matrix((H[k].ravel(), ((base + ii).ravel(), (base + jj).ravel())), shape))
        return out

    def set_vector(self, vec_val):
        if len(vec_val) != self.vector_size():
            raise ValueError('Wrong vector size for SimplexParam ' + self.name)
        self.set(np.reshape(vec_val, self._shape))

    def get_vector(self):
        return self._val.flatten()

    def get_vector_indices(self, row):
        return np.ravel_multi_index([[row], range(self._shape[1])], self._shape)

    def shape(self):
        return self._shape

    def free_shape(self):
        return self._free_shape

    def free_size(self):
        return int(np.prod(self._free_shape))

    def vector_size(self):
        return int(np.prod(self._shape))


# ------------------------------------------------------------------------------ offsets
def get_perpendicular_subspace(x):
    """Orthonormal basis (as columns) of the orthogonal complement of the row space of x.
    LRVB/ProjectionParams.py:13-26: eigenvectors of I - x^T (x x^T)^-1 x with non-zero eigenvalue;
    asserts that exactly x.shape[1] - x.shape[0] of them survive (full row rank)."""
    x = np.asarray(x, dtype=np.float64)
    n_con, dim = x.shape
    complement = np.eye(dim) - x.T @ np.linalg.solve(x @ x.T, x)
    evals, evecs = np.linalg.eigh(complement)
    keep = np.abs(evals) > 1e-8
    assert np.sum(keep) == dim - n_con
    return evecs[:, keep]


class SubspaceVectorParam(object):
    """A vector confined to the subspace orthogonal to the rows of `perp_subspace` (default: one row of
    ones, i.e. zero-mean vectors).  LRVB/ProjectionParams.py:30-100.  The free <-> vector map is linear
    (vector = basis @ free), the Hessians of the map are empty, and `set` only checks the size -- like
    the reference it does not project or reject an off-subspace value."""

    def __init__(self, name='', dim=3, val=None, perp_subspace=None):
        self.name = name
        self._dim = int(dim)
        if perp_subspace is None:
            self._perp = np.full((1, self._dim), 1.0)
        else:
            self._perp = copy.deepcopy(perp_subspace)
        if self._perp.shape[1] != self._dim:
            raise ValueError('The rows of <perp_subspace> must be of length <dim>')
        if self._perp.shape[0] >= self._dim:
            raise ValueError('<perp_subspace> must have strictly fewer rows than <dim>')
        self._basis = get_perpendicular_subspace(self._perp)
        self._free_dim = self._dim - self._perp.shape[0]
        self.set(val if val is not None else np.full(self._dim, 0.))

    def layout_blocks(self):
        raise NotImplementedError(
            'SubspaceVectorParam ' + self.name + ' has no device packing block: the map is linear, so keep the '
            'full vector as a VectorParam in the device objective and apply the basis on the host '
            '(H_free = B^T H_vec B, g_free = B^T g_vec)')

    def __str__(self):
        return self.name + ':\n' + str(self._val)

    def names(self):
        return [self.name + '_' + str(k) for k in range(self.vector_size())]

    def dictval(self):
        return self._val.tolist()

    def set(self, val):
        if val.size != self.dim():
            raise ValueError('Wrong size for vector ' + self.name)
        self._val = val

    def get(self):
        return self._val

    def set_free(self, free_val):
        if free_val.size != self.free_size():
            raise ValueError('Wrong free size for vector ' + self.name)
        self.set(self._basis @ free_val)

    def get_free(self):
        return self._basis.T @ self._val

    def free_to_vector(self, free_val):
        self.set_free(free_val)
        return self.get_vector()

    def free_to_vector_jac(self, free_val):
        return coo_matrix(self._basis)

    def free_to_vector_hess(self, free_val):
        return [coo_matrix((self._free_dim, self._free_dim)) for _ in range(self.vector_size())]

    def set_vector(self, val):
        self.set(val)

    def get_vector(self):
        return self._val

    def dim(self):
        return self._dim

    def free_size(self):
        return self._free_dim

    def vector_size(self):
        return self._dim


def set_free_offset(param, free_vec, offset):
    param.set_free(free_vec[offset:(offset + param.free_size())])
    return offset + param.free_size()


def get_free_offset(param, vec, offset):
    vec[offset:(offset + param.free_size())] = param.get_free()
    return offset + param.free_size()


def set_vector_offset(param, vec, offset):
    param.set_vector(vec[offset:(offset + param.vector_size())])
    return offset + param.vector_size()


def get_vector_offset(param, vec, offset):
    vec[offset:(offset + param.vector_size())] = param.get_vector()
    return offset + param.vector_size()


def offset_sparse_matrix(spmat, offset_shape, full_shape):
    """LRVB/Parameters.py:362-366."""
    spmat = coo_matrix(spmat)
    return coo_matrix((spmat.data, (spmat.row + offset_shape[0], spmat.col + offset_shape[1])),
                      shape=full_shape)


def free_to_vector_jac_offset(param, free_vec, free_offset, vec_offset):
    jac = param.free_to_vector_jac(free_vec[free_offset:free_offset + param.free_size()])
    return free_offset + param.free_size(), vec_offset + param.vector_size(), jac


def free_to_vector_hess_offset(param, free_vec, hessians, free_offset, full_shape):
    hess = param.free_to_vector_hess(free_vec[free_offset:free_offset + param.free_size()])
    for h in hess:
        hessians.append(offset_sparse_matrix(h, (free_offset, free_offset), full_shape))
    return free_offset + param.free_size()


# ------------------------------------------------------------------------------ dictionary
class ModelParamsDict(object):
    """Ordered container that is itself a parameter.  LRVB/ParameterDictionary.py:19-110."""

    def __init__(self, name='ModelParamsDict'):
        self.param_dict = OrderedDict()
        self.free_indices_dict = OrderedDict()
        self.vector_indices_dict = OrderedDict()
        self.name = name
        self._free_size = 0
        self._vector_size = 0
        self.values = ModelParamsDictValues(self)

    def __str__(self):
        return self.name + ':\n' + '\n'.join(['\t' + str(p) for p in self.param_dict.values()])

    def __getitem__(self, key):
        return self.param_dict[key]

    def push_param(self, param):
        self.param_dict[param.name] = param
        self.free_indices_dict[param.name] = range(self._free_size, self._free_size + param.free_size())
        self.vector_indices_dict[param.name] = \
            range(self._vector_size, self._vector_size + param.vector_size())
        self._free_size += param.free_size()
        self._vector_size += param.vector_size()

    def layout_blocks(self):
        out = []
        for p in self.param_dict.values():
            out.extend(p.layout_blocks())
        return out

    def set_name(self, name):
        self.name = name

    def dictval(self):
        return {p.name: p.dictval() for p in self.param_dict.values()}

    def _check(self, vec, want):
        if vec.size != want:
            raise ValueError('Wrong size for parameter {}.  Expected {}, got {}'.format(
                self.name, str(want), str(vec.size)))

    def set_free(self, vec):
        self._check(vec, self._free_size)
        offset = 0
        for p in self.param_dict.values():
            offset = set_free_offset(p, vec, offset)

    def get_free(self):
        return np.hstack([p.get_free() for p in self.param_dict.values()])

    def free_to_vector(self, free_val):
        self.set_free(free_val)
        return self.get_vector()

    def free_to_vector_jac(self, free_val):
        fo = vo = 0
        jacs = []
        for p in self.param_dict.values():
            fo, vo, j = free_to_vector_jac_offset(p, free_val, fo, vo)
            jacs.append(j)
        return block_diag(jacs)

    def free_to_vector_hess(self, free_val):
        fo = 0
        shape = (self.free_size(), self.free_size())
        hessians = []
        for p in self.param_dict.values():
            fo = free_to_vector_hess_offset(p, free_val, hessians, fo, shape)
        return hessians

    def set_vector(self, vec):
        self._check(vec, self._vector_size)
        offset = 0
        for p in self.param_dict.values():
            offset = set_vector_offset(p, vec, offset)

    def get_vector(self):
        return np.hstack([p.get_vector() for p in self.param_dict.values()])

    def names(self):
        return np.concatenate([np.atleast_1d(p.names()) for p in self.param_dict.values()])

    def free_size(self):
        return self._free_size

    def vector_size(self):
        return self._vector_size

    def get(self):
        return self.values


class ModelParamsDictValues(object):
    """Attribute-style access to the values.  LRVB/ParameterDictionary.py:115-124."""

    def __init__(self, param_dict):
        self.param_dict = param_dict

    def __getitem__(self, key):
        return self.param_dict[key].get()

    def __setitem__(self, key, val):
        return self.param_dict[key].set(val)


def convert_vector_to_free_hessian(param, free_val, vector_grad, vector_hess):
    """H_free = J^T H_vec J + sum_k g_k d2 eta_k (LRVB/Parameters.py:397-424), from the sparse
    Jacobian / Hessian lists of `param`.  Host arithmetic on the (small) sparse structure; the
    dense device version is `DeviceLayout.free_hessian_from_vector`."""
    free_val = np.asarray(free_val, dtype=np.float64)
    param.set_free(np.array(free_val))
    J = param.free_to_vector_jac(free_val).tocsr()
    hess_list = param.free_to_vector_hess(free_val)
    D = param.free_size()
    vals, rows, cols = [], [], []
    for k, h in enumerate(hess_list):
        h = coo_matrix(h)
        vals.append(h.data * vector_grad[k])
        rows.append(h.row)
        cols.append(h.col)
    third = coo_matrix((np.hstack(vals), (np.hstack(rows), np.hstack(cols))), (D, D))
    return third + J.T * vector_hess * J


def unvectorize_ld_matrix_vjp(g):
    """Cotangent of `unvectorize_ld_matrix`: the lower-triangle entries of g in packed order
    (LRVB/MatrixParameters.py:70-72; the map is linear, so its vector-Jacobian product is the packing itself)."""
    g = np.asarray(g)
    assert g.shape[0] == g.shape[1]
    return vectorize_ld_matrix(g)


def unvectorize_ld_matrix_jvp(g):
    """Tangent of `unvectorize_ld_matrix` (LRVB/MatrixParameters.py:77-78): the map applied to the tangent."""
    return unvectorize_ld_matrix(g)
