import os, sys, runpy
sys.argv = ['bench.py', '--config', sys.argv[1] if len(sys.argv) > 1 else 'c3', '--steps', '6', '--warmup', '2']
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'bench.py'), run_name='__main__')
