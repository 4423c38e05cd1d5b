"""Higher-order weight sensitivity at D = 1024 (development aid): time of d^k theta_hat / d w^k [dw] for k = 1, 2, 3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lrvb_amd as vb
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 250000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = np.random.default_rng(0)
x = rng.standard_normal((N, P)) / np.sqrt(P)
beta = rng.standard_normal(P)
y = (1.0 / (1.0 + np.exp(-x @ beta)) > rng.random(N)).astype(np.float64)
par = vb.ModelParamsDict('p')
par.push_param(vb.VectorParam('free', P - P // 4))
par.push_param(vb.VectorParam('pos', P // 4, lb=0.0))
fun = vb.GLMObjective(par, x, y, loss='logistic', prior_info=1.0)
obj = vb.Objective(par, fun)
x0, res = vb.OptimizationUtils.minimize_objective_trust_ncg(obj, par.get_free(), False, gtol=1e-6, maxiter=100, disp=False, on_device=True)
print('fit: nit %d, |g| %.2e' % (res.nit, res.jac_mag))
w0 = np.ones(N)
t0 = time.perf_counter()
tay = vb.ParametricSensitivityTaylorExpansion(fun, par, fun.weights_par, x0, w0, 3)
t1 = time.perf_counter()
print('base values (Hessian + Cholesky): %.1f ms' % ((t1 - t0) * 1e3))
dw = rng.standard_normal(N)
for k in (1, 2, 3):
    tay.evaluate_dkinput_dhyperk(dw, k)
    t0 = time.perf_counter()
    d = tay.evaluate_dkinput_dhyperk(dw, k)
    t1 = time.perf_counter()
    print('k = %d: %.1f ms, |d^k theta| = %.3e' % (k, (t1 - t0) * 1e3, np.linalg.norm(d)))
