"""Objective-level timings of the BASELINE configurations 2, 4, 5 (Hessian build, LRVB covariance)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import lrvb_amd as vb


def timeit(f, reps=3):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
    return (time.perf_counter() - t0) / reps * 1e3, out


rng = np.random.default_rng(0)
# config 2: MVNParam regression, N = 1e5, k = 21 -> D = 254
N, k = 100_000, 21
x = rng.normal(size=(N, k)); y = x @ rng.normal(size=k) + rng.normal(size=N) / np.sqrt(2.0)
par = vb.ModelParamsDict('p'); par.push_param(vb.MVNParam('beta', dim=k)); par.push_param(vb.GammaParam('tau'))
fun = vb.MVNRegressionObjective(par, x, y, prior_mean=np.zeros(k), prior_info=np.eye(k), prior_shape=2.0, prior_rate=2.0)
obj = vb.Objective(par, fun)
theta = par.get_free()
w = rng.uniform(0.5, 1.5, N)
def build2():
    fun.weights_par.set_vector(w + 0.0 * rng.normal())     # new weights object every call: statistics recomputed
    fun._w_res.key = None
    return obj.fun_free_hessian(theta)
ms, H = timeit(build2)
print('C2  N=1e5 D=%d: Hessian build (statistics on GPU + host closed forms) %.2f ms' % (H.shape[0], ms))
H2s = H + (0.1 - min(0.0, np.linalg.eigvalsh(H).min())) * np.eye(H.shape[0])      # not at an optimum: shift for the timing
ms, _ = timeit(lambda: fun.ctx.chol_factor(H2s)); print('C2  cho_factor (host matrix in) %.2f ms' % ms)

# config 5: Wishart + MVN, d = 63 -> D = 4096, N = 1e6
N5, d = 1_000_000, 63
yy = rng.normal(size=(N5, d))
par5 = vb.ModelParamsDict('p'); par5.push_param(vb.MVNParam('mu', dim=d)); par5.push_param(vb.WishartParam('lambda', size=d))
fun5 = vb.WishartMVNObjective(par5, yy)
obj5 = vb.Objective(par5, fun5)
par5['lambda']['df'].set(d + 5.0)
th5 = par5.get_free()
def build5():
    fun5._w_res.key = None
    return obj5.fun_free_hessian(th5)
ms, H5 = timeit(build5, reps=2)
print('C5  N=1e6 D=%d: exact Hessian build (S on GPU + host closed forms + device free-Hessian conversion) %.1f ms' % (H5.shape[0], ms))
ms, G5 = timeit(lambda: fun5.gram(th5), reps=2)
print('C5  G^T G (Kronecker rows on chip) %.1f ms' % ms)
Hs = H5 + (0.1 - min(0.0, np.linalg.eigvalsh(H5).min())) * np.eye(H5.shape[0])
ms, _ = timeit(lambda: fun5.ctx.chol_factor(Hs), reps=2); print('C5  cho_factor(D=4096, host matrix in) %.1f ms' % ms)
M5 = rng.normal(size=(16, H5.shape[0]))
ms, _ = timeit(lambda: fun5.ctx.lrvb_cov(M5), reps=2); print('C5  lrvb_cov(Q=16) %.1f ms' % ms)
b5 = rng.normal(size=H5.shape[0])
ms, out = timeit(lambda: fun5.ctx.cg_solve_matrix(Hs, b5, tol=1e-8), reps=2); print('C5  CG on the resident dense Hessian: %.1f ms, %d iterations' % (ms, out[2]))

# config 4: one GPU's shard of the hierarchical LMM (N = 1.25e6, p = 43, G = 1e4)
from test_lmm_host_math import make_par as lmm_par, random_eta
from oracle import packing as opk
N4, p4, G4 = 1_250_000, 43, 10_000
x4 = rng.normal(size=(N4, p4)); gid4 = rng.integers(0, G4, size=N4).astype(np.int32); gid4[:G4] = np.arange(G4)
y4 = x4 @ rng.normal(size=p4) + rng.normal(size=G4)[gid4] * 0.7 + rng.normal(size=N4) * 0.5
par4 = lmm_par(p4, G4)
fun4 = vb.LMMObjective(par4, x4, y4, gid4, G4)
th4 = par4.get_free()
def stats4():
    fun4._w_res.key = None; fun4._stats_cache = None
    return fun4.local_stats()
ms, _ = timeit(stats4); print('C4  shard N=1.25e6 p=43 G=1e4: sufficient statistics (host weights in) %.2f ms' % ms)
ms, HS4 = timeit(lambda: fun4.global_hessian(th4), reps=2); print('C4  Schur complement onto the %d global parameters %.1f ms' % (HS4.shape[0], ms))
