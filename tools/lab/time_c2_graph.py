"""Lab: a configuration-2 step (lrvb_mvnreg_hessian, result left in HBM) with the launch chain replayed as a captured graph
(profile marks off) against the plain launches (profile marks on: the graph is not used under them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lrvb_amd as vb
rng = np.random.default_rng(0)
N, k = 100000, 21
x = rng.normal(size=(N, k)); y = x @ rng.normal(size=k) + rng.normal(size=N)
par = vb.ModelParamsDict('p'); par.push_param(vb.MVNParam('beta', dim=k)); par.push_param(vb.GammaParam('tau'))
fun = vb.MVNRegressionObjective(par, x, y, prior_mean=np.zeros(k), prior_info=0.1 * np.eye(k), prior_shape=2.0, prior_rate=1.5)
theta = par.get_free()
for prof in (True, False, True, False):
    fun.ctx.profile_enable(prof)
    for _ in range(20): fun.device_hessian(theta, want_host=False)
    fun.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(300): fun.device_hessian(theta, want_host=False)
    fun.ctx.sync()
    t1 = time.perf_counter()
    print('profile marks %-5s (graph %s): %.1f us per step' % (prof, 'off' if prof else 'on', (t1 - t0) / 300 * 1e6))
H = fun.device_hessian(theta)[0]
fun.ctx.profile_enable(True)
H2 = fun.device_hessian(theta)[0]
print('graph vs plain launches: max diff', np.max(np.abs(H - H2)))
