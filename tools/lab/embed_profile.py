"""Where does the host time of an embedded config-2 step go after the CPU-baseline leg?  (cProfile of 20 steps.)"""
import sys, os, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
args = bench.parse(['--steps', '3', '--warmup', '1', '--cpu-budget-s', '6', '--no-configs'])
out = bench.main(args)                       # headline + cpu baseline, as the default run does
import torch
dev = torch.device('cuda', 0)
for rep in range(2):
    pr = cProfile.Profile(); pr.enable()
    r = bench.run_config(args, cfg='c2', steps=20, warmup=3, env=(1, 0, dev, None, False))
    pr.disable()
    print('c2 ms_per_step', r['ms_per_step'], file=sys.stderr)
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500], file=sys.stderr)
