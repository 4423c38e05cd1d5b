"""Trust-region Newton-CG at the headline shape (N = 1e6, D = 1024, logistic): scipy driving device callbacks
(the reference's structure, LRVB/OptimizationUtils.py:44-75) against the same optimiser as one library call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.optimize
import lrvb_amd as vb
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(3)
X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
beta = torch.randn((P,), dtype=torch.float64, device=dev, generator=g)
y = (torch.sigmoid(X @ beta) > torch.rand((N,), dtype=torch.float64, device=dev, generator=g)).double()
w = torch.ones((N,), dtype=torch.float64, device=dev)
blocks = [dict(kind=0, free_size=P - P // 4, vec_size=P - P // 4, dim0=P - P // 4, dim1=0, lb=-np.inf, ub=np.inf),
          dict(kind=0, free_size=P // 4, vec_size=P // 4, dim0=P // 4, dim1=0, lb=0.0, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='logistic', n_obs=N, n_cols=P, quad_kind=1)
ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, y.data_ptr(), N, 1); ctx.set_weights_dev(w.data_ptr(), N)
ctx.set_data(2, np.ones(P))
x0 = np.zeros(P)
for rep in range(2):
    t0 = time.perf_counter()
    res = scipy.optimize.minimize(ctx.value, x0, jac=ctx.grad, hessp=ctx.hvp, method='trust-ncg',
                                  options=dict(gtol=1e-4, maxiter=100))
    t1 = time.perf_counter()
    ctx.set_tuning(0, 8)                                   # tuning bit 3: every product a pass over X (rounds 2-3)
    ymf, xmf, info_mf = ctx.minimize_trust_ncg(x0, gtol=1e-4, maxiter=100)
    ctx.set_tuning(0, 0)
    t2 = time.perf_counter()
    yd, xd, info = ctx.minimize_trust_ncg(x0, gtol=1e-4, maxiter=100)
    t2b = time.perf_counter()
print('scipy + device callbacks: %.1f ms, nit %d, nfev %d, njev %d, nhev %d, f = %.6f' % ((t1 - t0) * 1e3, res.nit, res.nfev, res.njev, res.nhev, res.fun))
print('device loop, matrix-free: %.1f ms, nit %d, nfev %d, njev %d, nhev %d, f = %.6f' % ((t2 - t1) * 1e3, info_mf['nit'], info_mf['nfev'], info_mf['njev'], info_mf['nhev'], info_mf['fun']))
print('device loop (Hessian built inside long CG runs): %.1f ms, nit %d, nhev %d, nbuild %d, f = %.6f, max |x - x_matrix_free| = %.2e'
      % ((t2b - t2) * 1e3, info['nit'], info['nhev'], info['nbuild'], info['fun'], np.max(np.abs(xd - xmf))))
print('max |x_dev - x_scipy| = %.2e' % np.max(np.abs(xd - res.x)))
# the preconditioned route of the reference (`set_objective_preconditioner` + the `_cond` family, LRVB/OptimizationUtils.py:25-41,
# SparseObjectives.py:202-240, restarted as `repeatedly_optimize` does, :114-162): a few plain iterations, then A = H^-1/2 from ONE
# device Hessian at that point (eigenvalues clamped from below), then the same device loop in y-coordinates (x = A y)
t3 = time.perf_counter()
_, x1, info1 = ctx.minimize_trust_ncg(x0, gtol=1e-4, maxiter=8)
t4 = time.perf_counter()
H1 = ctx.hessian(x1)
lam, Q = np.linalg.eigh(0.5 * (H1 + H1.T))
lam = np.clip(lam, 1e-3 * lam.max(), None)
A = (Q / np.sqrt(lam)) @ Q.T
t5 = time.perf_counter()
yc, xc, infoc = ctx.minimize_trust_ncg(np.linalg.solve(A, x1), precond=A, gtol=1e-4, maxiter=100)
t6 = time.perf_counter()
print('restarted with a preconditioner: 8 plain iterations %.1f ms (nhev %d) + Hessian and eigh (host) %.1f ms + preconditioned fit %.1f ms '
      '(nit %d, nhev %d), f = %.6f, max |x - x_plain| = %.2e'
      % ((t4 - t3) * 1e3, info1['nhev'], (t5 - t4) * 1e3, (t6 - t5) * 1e3, infoc['nit'], infoc['nhev'], infoc['fun'], np.max(np.abs(xc - xd))))
