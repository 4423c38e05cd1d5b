import os, sys, runpy
os.environ['LRVB_BENCH_FORCE_SHARDED'] = '1'
sys.argv = ['bench.py', '--n-obs', '125000', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-configs']
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'bench.py'), run_name='__main__')
