"""Times config 3's device pipeline (lrvb_mixture_rows) at N rows, K = 32, V = 31.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import lrvb_amd as vb
from test_mixture_host_math import clustered_problem, make_par
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
V, K = 31, 32
x, w, fg, fz, lam = clustered_problem(N, V, K, seed=11)
theta = np.concatenate([fg, fz.ravel()])
par = vb.ModelParamsDict('params')
par.push_param(vb.DirichletParamArray('pi', shape=(K,)))
par.push_param(vb.DirichletParamArray('phi', shape=(V, K)))
par.push_param(vb.SimplexParam('z', shape=(N, K)))
fun = vb.MixtureObjective(par, x, pi_prior=1.2, phi_prior=0.9, weights=w)
for rep in range(3):
    t0 = time.perf_counter(); stats = fun.local_stats(theta); t1 = time.perf_counter()
    print('local_stats (rows + kron + GEMM + Gram, host in/out): %.1f ms' % ((t1 - t0) * 1e3), flush=True)
t0 = time.perf_counter(); HS = fun.global_hessian(theta); t1 = time.perf_counter()
print('global_hessian (device rows + device Schur assembly, D_g = %d): %.1f ms' % (fun.n_global, (t1 - t0) * 1e3))
t0 = time.perf_counter(); cov = fun.global_cov(theta); t1 = time.perf_counter()
print('global_cov: %.1f ms; min eig H_S %.3e' % ((t1 - t0) * 1e3, np.linalg.eigvalsh(HS).min()))
