"""One FRESH blocked-CG solve (new point: the point state is rebuilt) after a warm-up, for a kernel timeline of what a first
solve at a point costs beyond its 13 iterations (tools/lab/prof.sh + tools/lab/trace_gaps.py)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import lrvb_amd as vb
N, P, Q = 1000000, 1024, 16
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(3)
X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
beta = torch.randn((P,), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
y = torch.randn((N,), dtype=torch.float64, device=dev, generator=g)
w = torch.ones((N,), dtype=torch.float64, device=dev)
blocks = [dict(kind=0, free_size=P - 256, vec_size=P - 256, dim0=P - 256, dim1=0, lb=-np.inf, ub=np.inf),
          dict(kind=0, free_size=256, vec_size=256, dim0=256, dim1=0, lb=0.0, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, lik_info=2.0, quad_kind=vb._hip.QUAD_DIAG)
ctx.set_data_dev(vb._hip.SLOT_X, X.data_ptr(), N, P)
ctx.set_data_dev(vb._hip.SLOT_Y, y.data_ptr(), N, 1)
ctx.set_weights_dev(w.data_ptr(), N)
ctx.set_data(vb._hip.SLOT_QUAD_A, np.ones(P))
rng = np.random.default_rng(0)
theta = 0.05 * rng.normal(size=P)
B = rng.normal(size=(Q, P))
ctx.cg_solve_multi(theta, B, tol=1e-8)
ctx.sync()
for rep in range(3):
    th = theta + 1e-3 * (rep + 1)
    t0 = time.perf_counter(); Xs, info, iters = ctx.cg_solve_multi(th, B, tol=1e-8); ctx.sync(); t1 = time.perf_counter()
    print('fresh solve: %.2f ms, iterations %s' % ((t1 - t0) * 1e3, iters[:3]), flush=True)
t0 = time.perf_counter(); ctx.cg_solve_multi(th, B, tol=1e-8); ctx.sync(); t1 = time.perf_counter()
print('same point again: %.2f ms' % ((t1 - t0) * 1e3))
