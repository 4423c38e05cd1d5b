"""Five Hessian builds at the headline shape (development aid for rocprofv3 runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ['syrk_only.py', '0,0,0']
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'syrk_only.py'), run_name='__main__')
