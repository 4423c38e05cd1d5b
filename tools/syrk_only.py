"""Runs the Hessian build a few times at the headline shape for each split count given (development aid for PMC runs:
   rocprofv3 --pmc FETCH_SIZE -d out -- python3 tools/syrk_only.py 0,128,256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lrvb_amd as vb
splits = [int(s) for s in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0]
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
P = 1024
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
y = torch.randn((N,), dtype=torch.float64, device=dev, generator=g)
w = torch.ones((N,), dtype=torch.float64, device=dev)
theta = torch.randn((P,), dtype=torch.float64, device=dev, generator=g) * 0.1
H = torch.empty((P, P), dtype=torch.float64, device=dev)
blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, lik_info=2.0, quad_kind=1)
ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, y.data_ptr(), N, 1); ctx.set_weights_dev(w.data_ptr(), N)
ctx.set_data(2, np.ones(P))
for s in splits:
    ctx.set_tuning(s, 0)
    for _ in range(2):
        ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), P)
    ctx.sync()
