"""Times the weighted sufficient-statistics kernel (S = Z^T diag(w) Z) at the small-k shapes of
configs 2 and 4 (HBM-bound)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lrvb_amd as vb
for N, q in ((100000, 22), (1000000, 64), (10000000, 45)):
    dev = torch.device('cuda:0')
    Z = torch.randn((N, q), dtype=torch.float64, device=dev)
    w = torch.rand((N,), dtype=torch.float64, device=dev)
    blocks = [dict(kind=0, free_size=q, vec_size=q, dim0=q, dim1=0, lb=-np.inf, ub=np.inf)]
    ctx = vb.DeviceContext(blocks, loss='data_only', n_obs=N, n_cols=q)
    ctx.set_data_dev(0, Z.data_ptr(), N, q); ctx.set_weights_dev(w.data_ptr(), N)
    torch.cuda.synchronize()
    S = ctx.weighted_gram()
    ref = (Z.T * w) @ Z
    err = np.max(np.abs(S - ref.cpu().numpy())) / np.max(np.abs(ref.cpu().numpy()))
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.time()
    for _ in range(5): ctx.weighted_gram()
    t1 = time.time()
    p = ctx.profile_get()
    ms = p['wsyrk_ms'] / max(p['wsyrk_calls'], 1)
    byt = 8.0 * N * (q + 1)
    print('N=%d q=%d: wsyrk kernel %.3f ms (%.2f TB/s of the %.1f MB algorithmic), whole call %.3f ms, rel err %.1e' % (
        N, q, ms, byt / ms / 1e9, byt / 1e6, (t1 - t0) / 5 * 1e3, err), flush=True)
    del ctx, Z, w
