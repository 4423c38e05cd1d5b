"""Timing lab for the next round: 128 x 128 (four waves, two workgroups per CU) against 256 x 128 (eight waves, one
workgroup per CU) weighted-SYRK tiles on the same off-diagonal rectangle of S (rows 512..1023 x columns 0..511 at
P = 1024: 16 tile units of 128 x 128).  Results of the kernels are discarded (lrvb_lab_syrk is not in the C ABI header)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
lib = vb._hip.load()
fn = lib.lrvb_lab_syrk
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
fn.restype = ctypes.c_int
ms = ctypes.c_double(0.0)
flops = 2.0 * N * 512 * 512
for splits in (32, 64, 128):
    for variant in (2, 4, 2, 4):
        vb._hip.check(fn(ctx._h, variant, splits, 5, ctypes.byref(ms)))
        print('splits %3d  tile %3d x 128 (%d waves): %.3f ms  = %.1f TFLOP/s on 256 CUs' % (
            splits, 64 * variant, 2 * variant, ms.value, flops / ms.value / 1e9), flush=True)
