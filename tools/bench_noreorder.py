"""bench.py with the backward sweep's tile re-ordering toggled (stages._BWD_REORDER): python tools/bench_noreorder.py [bench args]"""
import sys, os, runpy
sys.path.insert(0, os.path.join(os.getcwd(), "3d-gaussian-splatting_amd"))
from gsplat_amd import stages
stages._BWD_REORDER = not stages._BWD_REORDER   # A/B against the default
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
