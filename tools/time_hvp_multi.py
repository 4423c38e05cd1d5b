"""Blocked CG at the headline shape (16 right-hand sides): wall time of the solve; run under
`rocprofv3 --kernel-trace --stats` for the per-call time of hvp_multi_kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lrvb_amd as vb
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
P = 1024
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
y = torch.randn((N,), dtype=torch.float64, device=dev, generator=g)
w = torch.ones((N,), dtype=torch.float64, device=dev)
blocks = [dict(kind=0, free_size=P - 256, vec_size=P - 256, dim0=P - 256, dim1=0, lb=-np.inf, ub=np.inf),
          dict(kind=0, free_size=256, vec_size=256, dim0=256, dim1=0, lb=0.0, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, lik_info=2.0, quad_kind=1)
ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, y.data_ptr(), N, 1); ctx.set_weights_dev(w.data_ptr(), N)
ctx.set_data(2, np.ones(P))
rng = np.random.default_rng(5)
theta = rng.normal(size=P) * 0.05
rhs = rng.normal(size=(16, P))
ctx.cg_solve_multi(theta, rhs[:2], tol=1e-8)
for rep in range(3):
    t0 = time.perf_counter()
    X_, info, its = ctx.cg_solve_multi(theta, rhs, tol=1e-8)
    t1 = time.perf_counter()
    print('cg_solve_multi 16 rhs: %.2f ms, iterations %s, info %s' % ((t1 - t0) * 1e3, its.max(), info.max()), flush=True)
