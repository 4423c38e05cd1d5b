"""Timing of the fused observation pass on wide designs (1024 < n_cols <= 4096): HIP-event time of the gradient pass through the
library profile, against the algorithmic bytes 8 N (P + 3).  usage: python tools/time_wide_pass.py [P ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrvb_amd as vb

dev = torch.device('cuda', 0)
for P in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096]:
    N = int(8.2e9 / 8 / P)
    X = torch.randn((N, P), dtype=torch.float64, device=dev) / P ** 0.5
    y = (torch.rand((N,), dtype=torch.float64, device=dev) < 0.5).double()
    blocks = [dict(kind=0, free_size=P, vec_size=P, dim0=P, dim1=0, lb=-np.inf, ub=np.inf)]
    for flags, name in ((0, 'one pass'), (1, 'two passes (tuning bit 0)')):
        ctx = vb.DeviceContext(blocks, loss='logistic', n_obs=N, n_cols=P, quad_kind=vb._hip.QUAD_DIAG, device=0)
        ctx.set_data_dev(vb._hip.SLOT_X, X.data_ptr(), N, P)
        ctx.set_data_dev(vb._hip.SLOT_Y, y.data_ptr(), N, 1)
        ctx.set_data(vb._hip.SLOT_QUAD_A, np.ones(P))
        ctx.set_tuning(0, flags)
        th = np.random.default_rng(0).normal(size=P) * 0.05
        for _ in range(3):
            ctx.grad(th + 1e-9 * np.random.default_rng(1).normal(size=P))
        ctx.profile_enable(True); ctx.profile_reset()
        for k in range(10):
            ctx.grad(th + 1e-9 * k)
        prof = ctx.profile_get()
        ms = prof['pass_ms'] / max(prof['pass_calls'], 1)
        print('P = {:5d}  N = {:8d}  {:26s} {:7.3f} ms  {:6.2f} TB/s'.format(P, N, name, ms, 8.0 * N * (P + 3) / ms / 1e9), flush=True)
        del ctx
    del X, y
    torch.cuda.empty_cache()
