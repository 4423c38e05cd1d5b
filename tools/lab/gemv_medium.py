"""CG on a resident dense 1024 x 1024 matrix with a dense preconditioner (lrvb_cg_solve_matrix): every iteration is two square
matrix-vector products -- what the medium-size gemv dispatch is for.  Run under tools/lab/prof.sh for the kernel split."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import lrvb_amd as vb
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
A = rng.normal(size=(D, D)); H = A @ A.T / D + np.eye(D)
Minv = np.diag(1.0 / np.diag(H)) + 1e-3 * np.eye(D)
b = rng.normal(size=D)
ctx = vb.DeviceContext([dict(kind=0, free_size=D, vec_size=D, dim0=D, dim1=0, lb=-np.inf, ub=np.inf)], quad_kind=1)
for rep in range(3):
    t0 = time.perf_counter(); x, info, it = ctx.cg_solve_matrix(H, b, Minv=Minv, tol=1e-10); t1 = time.perf_counter()
    print('D=%d: cg_solve_matrix %.2f ms, %d iterations, residual %.1e' % (D, (t1 - t0) * 1e3, it, np.linalg.norm(H @ x - b) / np.linalg.norm(b)), flush=True)
