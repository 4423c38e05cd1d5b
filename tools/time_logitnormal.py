"""Times the logit-normal regression model (LRVB/Modeling.py's non-conjugate logistic term) at N rows,
P coefficients (D = 2 P free parameters): value + gradient, and the Hessian (three weighted MFMA products).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import lrvb_amd as vb
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(0)
x = rng.normal(size=(N, P)) / np.sqrt(P)
y = (rng.uniform(size=N) < 1 / (1 + np.exp(-x @ rng.normal(size=P)))).astype(np.float64)
par = vb.ModelParamsDict('params')
par.push_param(vb.UVNParamVector('beta', length=P))
fun = vb.LogitNormalRegressionObjective(par, x, y, prior_info=1.0, gh_deg=20)
theta = np.concatenate([rng.normal(size=P) * 0.1, np.zeros(P)])
for rep in range(3):
    t0 = time.perf_counter(); g = fun.grad(theta, True); t1 = time.perf_counter()
    H = fun.hessian(theta, True); t2 = time.perf_counter()
    print('N = %d, P = %d: value + gradient %.1f ms, Hessian (D = %d) %.1f ms (host buffers in and out); 3 x 2 N P^2 = %.2e flops'
          % (N, P, (t1 - t0) * 1e3, 2 * P, (t2 - t1) * 1e3, 6.0 * N * P * P), flush=True)
print('min eig H %.3e' % np.linalg.eigvalsh(0.5 * (H + H.T)).min())
