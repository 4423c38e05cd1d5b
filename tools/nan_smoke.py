import sys; import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import lrvb_amd as vb
from helpers import make_par, glm_data
from oracle import models as om
rng = np.random.default_rng(0)
N, P = 1000, 256
par, lay = make_par(vb, [('box', 'b', P, 0.0, np.inf)])
x, y, w = glm_data(rng, N, P, om.POISSON)
fun = vb.DeviceObjective(par, x=x, y=y, loss='poisson', quad_A=np.ones(P), weights=w)
obj = vb.Objective(par, fun)
th = rng.normal(size=P) * 0.1
th[3] = np.nan
print('value', obj.fun_free(th))
g = obj.fun_free_grad(th); print('grad nan count', np.isnan(g).sum())
H = obj.fun_free_hessian(th); print('hess nan count', np.isnan(H).sum())
try:
    fun.ctx.chol_factor(H); print('chol ok?!')
except np.linalg.LinAlgError as e:
    print('chol: LinAlgError', str(e)[:60])
X, info, it = fun.ctx.cg_solve_multi(th, rng.normal(size=(3, P)), maxiter=5)
print('cg info', info, 'nan', np.isnan(X).sum())
th2 = rng.normal(size=P) * 0.1
th2[5] = 800.0     # exp overflow in the box map
print('overflow value', obj.fun_free(th2))
# mixture with NaN logits
from test_mixture_host_math import make_par as mk, problem
xm, wm, thm = problem(40, 5, 4, seed=1)
f = vb.MixtureObjective(mk(40, 5, 4), xm, weights=wm)
thm[f.n_global + 2] = np.nan
print('mixture value', f.value(thm))
try:
    f.global_hessian(thm); print('mixture schur computed')
except np.linalg.LinAlgError as e:
    print('mixture: LinAlgError', str(e)[:50])
print('done')
