"""Two fused passes (gradient state + Hessian-vector product) at the headline shape: the timing behind the non-temporal-load change."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lrvb_amd as vb
N, P = 1000000, 1024
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
y = (torch.rand((N,), dtype=torch.float64, device=dev, generator=g) > 0.5).double()
w = torch.ones((N,), dtype=torch.float64, device=dev)
blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='logistic', n_obs=N, n_cols=P, quad_kind=1)
ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, y.data_ptr(), N, 1); ctx.set_weights_dev(w.data_ptr(), N)
ctx.set_data(2, np.ones(P))
th = torch.zeros(P, dtype=torch.float64, device=dev); v = torch.ones(P, dtype=torch.float64, device=dev); o = torch.empty_like(v)
ctx.hvp_dev(th.data_ptr(), v.data_ptr(), o.data_ptr()); ctx.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): ctx.hvp_dev(th.data_ptr(), v.data_ptr(), o.data_ptr())
    ctx.sync()
    print('hvp_dev (2 passes) %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
