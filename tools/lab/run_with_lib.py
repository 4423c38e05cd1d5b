"""Runs a tools/ script against ANOTHER build of liblrvb_hip.so (an ablation or variant built next to the product library):
    python tools/lab/run_with_lib.py <path/to/variant.so relative to tools/lab> <script in tools/> [args...]
The product library is never replaced; the variant is loaded in its place for this process only."""
import sys, os, runpy
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tools'))
import lrvb_amd._hip as h
h.LIB_PATH = os.path.join(root, 'tools', 'lab', sys.argv[1])
sys.argv = [sys.argv[2]] + sys.argv[3:]
runpy.run_path(os.path.join(root, 'tools', sys.argv[0]), run_name='__main__')
