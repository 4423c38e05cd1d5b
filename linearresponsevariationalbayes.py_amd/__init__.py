"""MI355X-native linear-response variational Bayes hot path.

Public surface = the reference package's (LRVB/__init__.py:1-24) plus the declared-objective
constructors that replace opaque closures on the device path.  Typical use:

    import lrvb_amd as vb                       # loads this package (see lrvb_amd.py at the repo root)
    par = vb.ModelParamsDict('params'); par.push_param(vb.VectorParam('beta', D, lb=0.))
    fun = vb.GLMObjective(par, x, y, loss='gaussian', glm_param='beta', prior_info=1.0)
    objective = vb.Objective(par, fun)          # same class and methods as the reference
    H = objective.fun_free_hessian(theta)       # one weighted-SYRK pass on the GPU

Reference module names are importable as attributes for drop-in code
(`vb.SparseObjectives.Objective`, `vb.ModelSensitivity...`, `vb.ConjugateGradient...`,
`vb.OptimizationUtils...`, `vb.ExponentialFamilies...`, `vb.Parameters...`).
"""
from .version import __version__
from . import version

from .packing import (ScalarParam, VectorParam, HyperVectorParam, ArrayParam, PosDefMatrixParam, PosDefMatrixParamVector,
                      PosDefMatrixParamArray, SimplexParam, SubspaceVectorParam,
                      ModelParamsDict, ModelParamsDictValues, convert_vector_to_free_hessian)
from .families import (UVNParam, UVNParamVector, UVNParamArray, UVNMomentParamArray, MVNParam,
                       MVNArray, GammaParam, WishartParam, DirichletParamArray)
from .objectives import (Objective, TwoParameterObjective, ParameterConverter, ParametricSensitivity,
                         LinearConverter, ElementwiseConverter, Logger, Timer)
from .sensitivity import (ParametricSensitivityLinearApproximation, HyperparameterSensitivityLinearApproximation,
                          get_kl_hessian, get_lrvb_cov)
from .taylor import ParametricSensitivityTaylorExpansion
from .cg import ConjugateGradientSolver
from .models import (DeviceContext, DeviceObjective, GLMObjective, QuadraticObjective, LinearMoments, scratch_context)
from .quadform import (QuadraticDataObjective, NormalRegressionObjective, MVNRegressionObjective,
                       WishartMVNObjective)

# reference-style module aliases
from . import packing as Parameters
from . import packing as MatrixParameters
from . import packing as SimplexParams
from . import packing as ParameterDictionary
from . import objectives as SparseObjectives
from . import sensitivity as ModelSensitivity
from . import cg as ConjugateGradient
from . import optim as OptimizationUtils
from . import expfam as ExponentialFamilies
from . import families as NormalParams
from .hierarchical import LMMObjective
from .mixture import MixtureObjective
from .logitnormal import LogitNormalRegressionObjective
from .torch_closure import TorchObjective
from . import regression as regression_utils
from . import packing as ProjectionParams
from . import families as GammaParams
from . import families as WishartParams
from . import families as DirichletParams
from . import modeling as Modeling
from . import distributed

# `import lrvb_amd.SparseObjectives as obj_lib` (the reference's usual import style) resolves too
import sys as _sys
for _alias in ('Parameters', 'MatrixParameters', 'SimplexParams', 'ParameterDictionary', 'ProjectionParams',
               'SparseObjectives', 'ModelSensitivity', 'ConjugateGradient', 'OptimizationUtils',
               'ExponentialFamilies', 'NormalParams', 'GammaParams', 'WishartParams', 'DirichletParams',
               'Modeling', 'regression_utils'):
    _sys.modules.setdefault(__name__ + '.' + _alias, globals()[_alias])
del _sys, _alias
