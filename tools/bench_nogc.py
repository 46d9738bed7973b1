"""bench.py with the cyclic garbage collector off (is the tail of slow steps host-side collection pauses?): python tools/bench_nogc.py [bench args]"""
import gc, os, runpy, sys
gc.disable()
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
