"""Lab: 16 right-hand sides solved ONE BY ONE at a point whose Hessian was not built (lrvb_cg_solve), and scipy's cg over
lrvb_hvp callbacks: with and without the automatic build of the point's Hessian (tuning bit 3 switches it off)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lrvb_amd as vb
N, P = 1000000, 1024
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(3)
X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
y = (torch.rand((N,), dtype=torch.float64, device=dev, generator=g) < 0.5).double()
blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='logistic', n_obs=N, n_cols=P, quad_kind=1)
ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, y.data_ptr(), N, 1)
ctx.set_data(2, np.ones(P))
rng = np.random.default_rng(0)
B = rng.normal(size=(16, P))
for flags, name in ((8, 'matrix-free (tuning bit 3)'), (0, 'automatic build')):
    for rep in range(2):
        th = rng.normal(size=P) * 0.05
        ctx.set_tuning(0, flags)
        t0 = time.perf_counter()
        its = [ctx.cg_solve(th, B[q], tol=1e-8)[2] for q in range(16)]
        t1 = time.perf_counter()
    print('%-28s 16 solves one by one: %.1f ms (iterations %s)' % (name, (t1 - t0) * 1e3, its[:4]))
