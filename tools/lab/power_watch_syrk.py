"""Clock and package power while Hessian builds (weighted SYRK) run back to back; then while the fused pass streams."""
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
theta = torch.zeros((P,), dtype=torch.float64, device=dev)
H = torch.empty((P, P), dtype=torch.float64, device=dev)
blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, lik_info=2.0, quad_kind=1)
ctx.set_data_dev(0, X.data_ptr(), N, P); ctx.set_data_dev(1, y.data_ptr(), N, 1); ctx.set_weights_dev(w.data_ptr(), N)
ctx.set_data(2, np.ones(P))
def smi():
    out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True).stdout
    d = json.loads(out); c = d[sorted(d)[0]]
    return c.get('sclk clock speed:'), c.get('Current Socket Graphics Package Power (W)')
for name, fn in (('hessian build (weighted SYRK)', lambda: ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), P)),
                 ('fused pass (HVP)', lambda: ctx.hvp(np.zeros(P), np.ones(P)))):
    stop = False
    def work():
        while not stop:
            for _ in range(10): fn()
            ctx.sync()
    t = threading.Thread(target=work); t.start()
    samples = []
    for i in range(8):
        time.sleep(0.4); samples.append(smi())
    stop = True; t.join()
    print(name, samples, flush=True)
    time.sleep(1.0)
print('idle', smi())
