"""Quick headline timing (development aid, not the bench contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lrvb_amd as vb
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
splits = [int(s) for s in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0]
flag_list = [int(f, 0) for f in sys.argv[4].split(',')] if len(sys.argv) > 4 else [0]
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
y = torch.randn((N,), dtype=torch.float64, device=dev, generator=g)
w = torch.ones((N,), dtype=torch.float64, device=dev)
theta = torch.randn((P,), dtype=torch.float64, device=dev, generator=g) * 0.1
H = torch.empty((P, P), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
blocks = [dict(kind=0, free_size=P - P // 4, vec_size=P - P // 4, dim0=P - P // 4, dim1=0, lb=-np.inf, ub=np.inf),
          dict(kind=0, free_size=P // 4, vec_size=P // 4, dim0=P // 4, dim1=0, lb=0.0, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, lik_info=2.0, quad_kind=1)
ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, y.data_ptr(), N, 1); ctx.set_weights_dev(w.data_ptr(), N)
ctx.set_data(2, np.ones(P))
for s, flags in [(s, f) for s in splits for f in flag_list]:
    ctx.set_tuning(s, flags)
    ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), P); ctx.sync()
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.time(); K = 5
    for _ in range(K):
        ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), P)
    ctx.sync(); t1 = time.time()
    p = ctx.profile_get(); ctx.profile_enable(False)
    ws = p['wsyrk_ms'] / max(p['wsyrk_calls'], 1); ps = p['pass_ms'] / max(p['pass_calls'], 1)
    print('flags=0x%x ' % flags, end='')
    print('splits=%d build %.3f ms  wsyrk %.3f ms (%.1f TF/s, %.1f%% of 78.6)  pass %.3f ms (%.2f TB/s)' % (
        s, (t1 - t0) / K * 1e3, ws, p['wsyrk_flops'] / ws / 1e9, p['wsyrk_flops'] / ws / 1e9 / 78.6 * 100, ps,
        (p['pass_bytes'] / ps / 1e9) if ps > 0 else 0.0), flush=True)
ctx.set_tuning(0, 0)
ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), P); ctx.sync()
# hvp timing
v = torch.randn((P,), dtype=torch.float64, device=dev); out = torch.empty_like(v)
ctx.hvp_dev(theta.data_ptr(), v.data_ptr(), out.data_ptr()); ctx.sync()
t0 = time.time()
for _ in range(10): ctx.hvp_dev(theta.data_ptr(), v.data_ptr(), out.data_ptr())
ctx.sync(); print('hvp_dev (grad pass + hvp pass) %.3f ms' % ((time.time() - t0) / 10 * 1e3))
# check against torch at a subsample of entries
Hs = (X[:, :64].T * (2.0 * w)) @ X[:, :64]
j1 = torch.ones(P, dtype=torch.float64, device=dev); j1[P - P // 4:] = torch.exp(theta[P - P // 4:])
ref = Hs[:64, :64] * j1[:64, None] * j1[None, :64]
# (only the unconstrained leading block: + prior 1 on the diagonal)
ref = ref + torch.eye(64, dtype=torch.float64, device=dev)
print('max rel err vs torch on leading 64x64 block: %.2e' % ((H[:64, :64] - ref).abs().max() / ref.abs().max()).item())
# solve timing
ctx.chol_factor_dev(H.data_ptr(), P, P); ctx.sync()
t0 = time.time(); ctx.chol_factor_dev(H.data_ptr(), P, P); ctx.sync(); t1 = time.time()
M = torch.randn((P, P), dtype=torch.float64, device=dev); cov = torch.empty((P, P), dtype=torch.float64, device=dev)
ctx.lrvb_cov_dev(M.data_ptr(), P, P, cov.data_ptr()); ctx.sync()
t2 = time.time(); ctx.lrvb_cov_dev(M.data_ptr(), P, P, cov.data_ptr()); ctx.sync(); t3 = time.time()
print('chol_factor %.3f ms, lrvb_cov(Q=D) %.3f ms' % ((t1 - t0) * 1e3, (t3 - t2) * 1e3))
