import torch, time
x = torch.randn((1000000, 1024), dtype=torch.float64, device='cuda')
y = torch.empty_like(x)
for name, fn in [('sum', lambda: x.sum()), ('copy', lambda: y.copy_(x)), ('abs().max', lambda: x.abs().max())]:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    gb = x.numel() * 8 / 1e9 * (2 if name == 'copy' else 1)
    print('%s: %.3f ms  %.2f TB/s' % (name, dt * 1e3, gb / dt / 1e3))
