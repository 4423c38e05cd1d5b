"""Does the fused multi-vector pass run into the power cap?  Runs blocked-CG products back to back in a thread and samples
rocm-smi (power, clocks) beside it."""
import sys, os, time, threading, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lrvb_amd as vb
N, P = 1000000, 1024
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, P), dtype=torch.float64, device=dev, generator=g) / P ** 0.5
y = torch.randn((N,), dtype=torch.float64, device=dev, generator=g)
w = torch.ones((N,), dtype=torch.float64, device=dev)
blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, lik_info=2.0, quad_kind=1)
ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, y.data_ptr(), N, 1); ctx.set_weights_dev(w.data_ptr(), N)
ctx.set_data(2, np.ones(P))
rng = np.random.default_rng(5)
theta = rng.normal(size=P) * 0.05
rhs = rng.normal(size=(16, P))
stop = False
def work():
    while not stop:
        ctx.cg_solve_multi(theta, rhs, tol=1e-30, maxiter=200)      # 200 products back to back
t = threading.Thread(target=work); t.start()
def smi():
    out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--json'], capture_output=True, text=True).stdout
    try:
        d = json.loads(out); c = d[sorted(d)[0]]
        return {k: v for k, v in c.items() if any(s in k.lower() for s in ('power', 'sclk', 'mclk', 'fclk', 'junction', 'hotspot'))}
    except Exception as e:
        return {'raw': out[:300], 'err': str(e)}
for i in range(8):
    time.sleep(0.5)
    print(i, smi(), flush=True)
stop = True; t.join()
time.sleep(1.0)
print('idle', smi())
