"""Times lrvb_grouped_stats (the fused one-pass kernel) at the config-4 shard shape; run under rocprofv3 for kernel times."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrvb_amd as vb
N, p, G = 1_250_000, 43, 10_000
rng = np.random.default_rng(1)
Z = rng.normal(size=(N, p + 1)); gid = rng.integers(0, G, size=N).astype(np.int32)
blocks = [dict(kind=0, free_size=1, vec_size=1, dim0=1, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, loss='data_only', n_obs=N, n_cols=p + 1)
ctx.set_data(0, Z); ctx.set_groups(gid, G); ctx.set_weights(rng.uniform(0.5, 1.5, N))
for _ in range(3): ctx.grouped_stats(want_S=True, want_gs=False)
t0 = time.perf_counter()
for _ in range(20): ctx.grouped_stats(want_S=True, want_gs=False)
print('grouped_stats call: {:.3f} ms'.format((time.perf_counter() - t0) / 20 * 1e3))
