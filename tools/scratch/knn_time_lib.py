import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bridged_gnn_amd import _lib
_lib.SO_PATH = sys.argv[1]
sys.argv = sys.argv[:1] + sys.argv[2:]
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "knn_time.py")).read())
