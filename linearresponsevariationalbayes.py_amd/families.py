"""Variational-family containers: a ModelParamsDict plus moment / entropy methods.  They define
the free-vector layout and the moment definitions the config ELBOs use.

Same classes, constructor arguments and method names as the reference:
  MVNParam, UVNParam, UVNParamVector, UVNParamArray, UVNMomentParamArray, MVNArray
                                                        LRVB/NormalParams.py:6-162
  GammaParam                                            LRVB/GammaParams.py:4-16
  WishartParam                                          LRVB/WishartParams.py:6-35
  DirichletParamArray                                   LRVB/DirichletParams.py:11-26
Two reference defects are fixed, not replicated: WishartParam builds its
`v` with the requested `size` (the reference ignores it, WishartParams.py:14-15, and always makes
a 2 x 2), and UVNMomentParamArray.var_exp returns the log-normal VARIANCE (the reference returns
the mean, NormalParams.py:123-124).
"""
import numpy as np

from . import expfam as ef
from .packing import (ModelParamsDict, ScalarParam, VectorParam, ArrayParam, PosDefMatrixParam)


class MVNParam(ModelParamsDict):
    """q(x) = N(mean, info^-1): a mean vector and an information matrix in log-Cholesky coordinates
    (moments as LRVB/NormalParams.py:6-23 defines them)."""

    def __init__(self, name='', dim=2, min_info=0.0):
        ModelParamsDict.__init__(self, name=name)
        self._dim = int(dim)
        for child in (VectorParam('mean', self._dim), PosDefMatrixParam('info', self._dim, diag_lb=min_info)):
            self.push_param(child)

    def e(self):
        return self['mean'].get()

    def cov(self):
        # the information matrix is positive definite by construction: invert through its Cholesky factor
        info = np.asarray(self['info'].get(), dtype=np.float64)
        chol = np.linalg.cholesky(info)
        half = np.linalg.solve(chol, np.eye(info.shape[0]))           # L^-1
        return half.T @ half

    def e_outer(self):
        m = np.asarray(self.e())
        second = self.cov() + m[:, None] * m[None, :]
        return (second + second.T) / 2.0

    def entropy(self):
        return ef.multivariate_normal_entropy(self['info'].get())


class _UVNBase(ModelParamsDict):
    def e(self):
        return self['mean'].get()

    def e_outer(self):
        return self['mean'].get() ** 2 + 1 / self['info'].get()

    def var(self):
        return 1. / self['info'].get()

    def e_exp(self):
        return ef.get_e_lognormal(self['mean'].get(), 1. / self['info'].get())

    def var_exp(self):
        return ef.get_var_lognormal(self['mean'].get(), 1. / self['info'].get())

    def e2_exp(self):
        return self.e_exp() ** 2 + self.var_exp()

    def entropy(self):
        return np.sum(ef.univariate_normal_entropy(self['info'].get()))


class UVNParam(_UVNBase):
    def __init__(self, name='', min_info=0.0):
        super().__init__(name=name)
        self.push_param(ScalarParam('mean'))
        self.push_param(ScalarParam('info', lb=min_info))

    def entropy(self):
        return ef.univariate_normal_entropy(self['info'].get())


class UVNParamVector(_UVNBase):
    def __init__(self, name='', length=2, min_info=0.0):
        super().__init__(name=name)
        self._size = length
        self.push_param(VectorParam('mean', length))
        self.push_param(VectorParam('info', length, lb=min_info))

    def size(self):
        return self._size


class UVNParamArray(_UVNBase):
    def __init__(self, name='', shape=(1, 1), min_info=0.0):
        super().__init__(name=name)
        self._shape = shape
        self.push_param(ArrayParam('mean', shape))
        self.push_param(ArrayParam('info', shape, lb=min_info))

    def shape(self):
        return self._shape


class UVNMomentParamArray(ModelParamsDict):
    """Moment parameterisation (E[x], E[x^2]) of an array of univariate normals."""

    def __init__(self, name='', shape=(2, 3), min_info=0.0):
        super().__init__(name=name)
        self._shape = shape
        self.push_param(ArrayParam('e', shape))
        self.push_param(ArrayParam('e2', shape, lb=min_info))

    def e(self):
        return self['e'].get()

    def e_outer(self):
        return self['e2'].get()

    def var(self):
        return self['e2'].get() - self['e'].get() ** 2

    def e_exp(self):
        return ef.get_e_lognormal(self['e'].get(), self.var())

    def var_exp(self):
        return ef.get_var_lognormal(self['e'].get(), self.var())

    def e2_exp(self):
        return self.e_exp() ** 2 + self.var_exp()

    def entropy(self):
        return np.sum(ef.univariate_normal_entropy(1. / self.var()))

    def shape(self):
        return self._shape

    def set_from_uvn_param_array(self, uvn_par):
        assert uvn_par.shape() == self.shape()
        self['e'].set(uvn_par.e())
        self['e2'].set(uvn_par.e_outer())

    def set_from_constant(self, scalar_array_par):
        assert scalar_array_par.shape() == self.shape()
        self['e'].set(scalar_array_par.get())
        self['e2'].set(scalar_array_par.get() ** 2)


class MVNArray(ModelParamsDict):
    """Rows are multivariate normals with a constant diagonal variance per row."""

    def __init__(self, name='', shape=(2, 2), min_info=0.0):
        super().__init__(name=name)
        self._shape = shape
        self.push_param(ArrayParam('mean', shape=shape))
        self.push_param(VectorParam('info', size=shape[0], lb=min_info))

    def e(self):
        return self['mean'].get()

    def e2(self):
        var = 1 / self['info'].get()
        return self['mean'].get() ** 2 + var[:, None]


class GammaParam(ModelParamsDict):
    def __init__(self, name='', min_shape=0.0, min_rate=0.0):
        super().__init__(name=name)
        self.push_param(ScalarParam('shape', lb=min_shape))
        self.push_param(ScalarParam('rate', lb=min_rate))

    def e(self):
        return self['shape'].get() / self['rate'].get()

    def e_log(self):
        return ef.get_e_log_gamma(shape=self['shape'].get(), rate=self['rate'].get())

    def entropy(self):
        return ef.gamma_entropy(shape=self['shape'].get(), rate=self['rate'].get())


class WishartParam(ModelParamsDict):
    def __init__(self, name='', size=2, diag_lb=0.0, min_df=None):
        super().__init__(name=name)
        self._size = int(size)
        if not min_df:
            min_df = size - 1
        assert min_df >= size - 1
        self.push_param(ScalarParam('df', lb=min_df))
        self.push_param(PosDefMatrixParam('v', size=self._size, diag_lb=diag_lb))

    def e(self):
        return self['df'].get() * self['v'].get()

    def e_log_det(self):
        return ef.e_log_det_wishart(self['df'].get(), self['v'].get())

    def e_inv(self):
        return self['df'].get() * np.linalg.inv(self['v'].get())

    def entropy(self):
        return ef.wishart_entropy(self['df'].get(), self['v'].get())

    def e_log_lkj_inv_prior(self, lkj_param):
        return ef.expected_ljk_prior(lkj_param, self['df'].get(), self['v'].get())


class DirichletParamArray(ModelParamsDict):
    """Axis 0 indexes the Dirichlet dimension; the remaining axes are an array of Dirichlets."""

    def __init__(self, name='', shape=(1, 2), min_alpha=0.0, val=None):
        super().__init__(name=name)
        self._shape = shape
        assert min_alpha >= 0, 'alpha parameter must be non-negative'
        self.push_param(ArrayParam('alpha', shape=shape, lb=min_alpha, val=val))

    def e(self):
        return ef.get_e_dirichlet(self['alpha'].get())

    def e_log(self):
        return ef.get_e_log_dirichlet(self['alpha'].get())

    def entropy(self):
        return ef.dirichlet_entropy(self['alpha'].get())
