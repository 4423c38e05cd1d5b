"""An objective that is none of the declared models, written as a closure in torch operations, through the reference's classes:
`Objective` (gradient, Hessian, Hessian-vector products in free coordinates), a trust-region fit, the LRVB covariance and
`ParametricSensitivityLinearApproximation` for a hyper-parameter of the closure (LRVB/SparseObjectives.py:95-116, 321-449;
ModelSensitivity.py:555-612 differentiate any autograd closure; here torch.func does in vector coordinates and the HIP library
converts to free coordinates, factors and solves).

    python -c "import __graft_entry__ as g; g.build()"
    python examples/torch_closure.py
"""
import os
import sys

import numpy as np
import scipy.optimize
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrvb_amd as vb                                               # noqa: E402

rng = np.random.default_rng(5)
k, n_lin = 12, 150
par = vb.ModelParamsDict('params')
par.push_param(vb.VectorParam('loc', n_lin))
par.push_param(vb.PosDefMatrixParam('scale', k))
par.push_param(vb.VectorParam('rate', 20, lb=0.0))
ridge = vb.VectorParam('ridge', 3, lb=0.0)                          # the hyper-parameter: three ridge penalties
ridge.set_vector(np.array([1.0, 0.5, 2.0]))
V = par.vector_size()
dev = torch.device('cuda', 0)
Bm = torch.tensor(rng.normal(size=(40, V)) / np.sqrt(V), device=dev)
S0 = rng.normal(size=(k, k)) * 0.1
par['scale'].set(1.5 * np.eye(k) + S0 @ S0.T); par['rate'].set(rng.uniform(1.0, 2.0, 20)); par['loc'].set(rng.normal(size=n_lin) * 0.3)
tgt = torch.tensor(par.get_vector(), device=dev)                    # a target inside the constraint set
seg = torch.tensor(np.repeat([0, 1, 2], [n_lin, k * (k + 1) // 2, 20]), device=dev)


def closure(eta, eps):
    r = eta - tgt
    return 0.2 * torch.logsumexp(Bm @ eta, 0) + 0.5 * torch.sum(eps[seg] * r * r) + 0.05 * torch.sum(r ** 4)


fun = vb.TorchObjective(par, closure, hyper_par=ridge)
objective = vb.Objective(par, fun)
print('D = %d free parameters (%d vector coordinates: a %d-vector, a %d x %d positive definite matrix, %d positive rates)'
      % (par.free_size(), V, n_lin, k, k, 20))
par['scale'].set(np.eye(k)); par['rate'].set(np.ones(20)); par['loc'].set(np.zeros(n_lin))
res = scipy.optimize.minimize(objective.fun_free, par.get_free(), jac=objective.fun_free_grad, hessp=objective.fun_free_hvp,
                              method='trust-ncg', options={'gtol': 1e-8})
theta = res.x
for _ in range(2):
    theta = theta - np.linalg.solve(objective.fun_free_hessian(theta), objective.fun_free_grad(theta))
print('fit: %d iterations, |grad| = %.1e' % (res.nit, np.max(np.abs(objective.fun_free_grad(theta)))))
H = objective.fun_free_hessian(theta)
M = np.eye(par.free_size())[:4]
cov = vb.ModelSensitivity.get_lrvb_cov(objective, theta, M, kl_hessian=H)
print('LRVB covariance of the first four free parameters (device Cholesky): diag =', np.round(np.diag(cov), 5))
sens = vb.ParametricSensitivityLinearApproximation(fun, par, ridge, theta, ridge.get_vector(), hyper_is_free=False)
new = ridge.get_vector() * np.array([1.05, 0.95, 1.02])
pred = sens.predict_input_par_from_hyperparameters(new)
old = ridge.get_vector().copy()
ridge.set_vector(new)
refit = scipy.optimize.minimize(objective.fun_free, theta, jac=objective.fun_free_grad, hess=objective.fun_free_hessian,
                                method='trust-exact', options={'gtol': 1e-10}).x
ridge.set_vector(old)
print('new ridge penalties: predicted move %.3e, actual %.3e, error of the prediction %.1e of the move'
      % (np.linalg.norm(pred - theta), np.linalg.norm(refit - theta), np.linalg.norm(pred - refit) / np.linalg.norm(refit - theta)))
