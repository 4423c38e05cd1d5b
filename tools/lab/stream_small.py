"""Calibration: how fast does this box stream a 440 MB fp64 array (the config-4 shard) with plain kernels?"""
import torch, time
dev = torch.device('cuda', 0)
for mb in (440, 880, 1760, 8200):
    n = mb * 1000 * 1000 // 8
    x = torch.ones(n, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    for fn, name, factor in ((lambda: x.sum(), 'sum', 1), (lambda: torch.mul(x, 2.0, out=y), 'scale (read+write)', 2)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('{:5d} MB {:20s} {:.1f} us  {:.2f} TB/s'.format(mb, name, ms * 1e3, factor * mb * 1e6 / (ms * 1e-3) / 1e12))
