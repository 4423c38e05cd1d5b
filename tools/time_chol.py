"""Times lrvb_chol_factor / lrvb_lrvb_cov at D (development aid; run under rocprofv3 for the kernel split)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lrvb_amd as vb
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.randn((D, D), dtype=torch.float64, device=dev, generator=g)
H = A @ A.T / D + torch.eye(D, dtype=torch.float64, device=dev)
blocks = [dict(kind=0, free_size=D, vec_size=D, dim0=D, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, quad_kind=1)
M = torch.randn((D, D), dtype=torch.float64, device=dev, generator=g)
cov = torch.empty((D, D), dtype=torch.float64, device=dev)
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    ctx.chol_factor_dev(H.data_ptr(), D, D); ctx.sync(); t1 = time.perf_counter()
    ctx.lrvb_cov_dev(M.data_ptr(), D, D, cov.data_ptr()); ctx.sync(); t2 = time.perf_counter()
    print('D=%d chol_factor %.3f ms, lrvb_cov(Q=D) %.3f ms' % (D, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
ref = M @ torch.linalg.solve(H, M.T)
print('cov rel err vs torch: %.2e' % ((cov - ref).abs().max() / ref.abs().max()).item())
# the vendor route for comparison (torch.linalg on ROCm dispatches to rocSOLVER / MAGMA)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Lt = torch.linalg.cholesky(H); torch.cuda.synchronize(); t1 = time.perf_counter()
    Yt = torch.cholesky_solve(M.T.contiguous(), Lt); torch.cuda.synchronize(); t2 = time.perf_counter()
print('torch.linalg.cholesky %.3f ms, torch.cholesky_solve(Q=D) %.3f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
