"""bench.py with autograd's worker threads off (backward passes run on the calling thread): what the cross-thread hand-off
of every backward() costs the host.  python tools/bench_singlethread_autograd.py [bench args]"""
import os, runpy, sys
import torch
torch.autograd.set_multithreading_enabled(False)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
