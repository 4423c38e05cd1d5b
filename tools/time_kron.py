"""Times the Kronecker-row Gram pipeline of config 5 (d = 63 -> q = 64, D = 4096) at N rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lrvb_amd as vb
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
q, d = 64, 63
V = (d + 1) ** 2
dev = torch.device('cuda:0')
Z = torch.randn((N, q), dtype=torch.float64, device=dev); Z[:, -1] = 1.0
blocks = [dict(kind=0, free_size=V, vec_size=V, dim0=V, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='data_only', n_obs=N, n_cols=q)
ctx.set_data_dev(0, Z.data_ptr(), N, q)
rng = np.random.default_rng(0)
M = rng.normal(size=(V, q, q)) * 0.01; M = M + M.transpose(0, 2, 1)
c = rng.normal(size=V) * 0.01
theta = np.zeros(V)
ctx.profile_enable(True)
for rep in range(2):
    ctx.profile_reset()
    t0 = time.time()
    G = ctx.quadform_gram(M, c, theta)
    t1 = time.time()
    p = ctx.profile_get()
    flops = N * 4096.0 * 4097.0
    print('N=%d: whole G^T G call %.1f ms; MFMA kernels %.2f ms (%d launches: Kronecker SYRK + 4 TN GEMMs of 4096^3) -> >= %.1f TFLOP/s on the SYRK flops alone (%.1f%% of 78.6)' % (
        N, (t1 - t0) * 1e3, p['wsyrk_ms'], p['wsyrk_calls'], flops / (p['wsyrk_ms'] * 1e-3) / 1e12,
        flops / (p['wsyrk_ms'] * 1e-3) / 1e12 / 78.6 * 100), flush=True)
# spot check a few entries against torch
idx = torch.tensor([0, 5, 77, 1234, 4095])
zz = (Z[:, :, None] * Z[:, None, :]).reshape(N, -1) if N <= 20000 else None
print('symmetric:', np.allclose(G, G.T, rtol=0, atol=1e-9 * np.abs(G).max()))
