"""Developer aid (GPU box): run bench.py in THIS process with every uninitialised device allocation (torch.empty /
empty_like / new_empty) filled with a byte pattern first -- a kernel chain that reads memory nobody wrote shows up as a different
result, an implausible control word or a fault, on every run instead of once in a while (fresh processes mostly get zero pages;
a process that starts after others have used the card does not).  python tools/poison_run.py [pattern byte] -- bench args..."""
import os
import runpy
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pattern = int(sys.argv[1], 0)
args = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2] == "--" else sys.argv[2:]
_empty, _empty_like = torch.empty, torch.empty_like
count = [0]


def _poison(t):
    if pattern >= 0 and t.is_cuda and t.numel() and t.is_contiguous():
        t.view(-1).view(torch.uint8).fill_(pattern)
        count[0] += 1
    return t


def empty(*a, **k):
    return _poison(_empty(*a, **k))


def empty_like(*a, **k):
    return _poison(_empty_like(*a, **k))


torch.empty, torch.empty_like = empty, empty_like
sys.argv = [os.path.join(ROOT, "bench.py")] + args
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    print(f"[poison_run] pattern {pattern:#x}: {count[0]} allocations filled", file=sys.stderr)
