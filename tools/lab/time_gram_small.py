"""Lab: HIP-event time of the narrow Gram kernel (library profile marks) for a given shape.  usage: python tools/lab/time_gram_small.py N P"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrvb_amd as vb
N, P = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
x = rng.normal(size=(N, P))
blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='gaussian', n_obs=N, n_cols=P, quad_kind=vb._hip.QUAD_DIAG, device=0)
ctx.set_data(vb._hip.SLOT_X, x); ctx.set_data(vb._hip.SLOT_Y, np.zeros(N)); ctx.set_data(vb._hip.SLOT_QUAD_A, np.ones(P))
for _ in range(3): S = ctx.weighted_gram()
ctx.profile_enable(True); ctx.profile_reset()
import time
t0 = time.perf_counter()
for _ in range(20): S = ctx.weighted_gram()
t1 = time.perf_counter()
pr = ctx.profile_get()
print('N=%d P=%d kernel %.1f us  call %.1f us  err %.1e' % (N, P, 1e3 * pr['wsyrk_ms'] / max(pr['wsyrk_calls'], 1), 1e6 * (t1 - t0) / 20, np.max(np.abs(S - x.T @ x)) / np.max(np.abs(S))))
