"""cProfile of bench.py --config CFG (host-side cost of a step): python tools/lab/pyprof.py c2 200"""
import cProfile, pstats, sys, os, runpy, io
cfg, steps = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '100'
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py', '--config', cfg, '--steps', steps, '--warmup', '3']
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(root, 'bench.py'), run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
