"""(GPU box) the 4-rank gloo rehearsal of bench.py started from a PARENT THAT HOLDS THE CARD (a HIP context, a few streams, cached
memory) -- what tests/test_scale.py's process is when the rehearsal runs inside the whole suite, where it fails about every
second time (garbage in a projection's output slabs; profiles/r05_four_rank_rehearsal.md).  Variants, a few runs each."""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
# the parent: context, streams of both priorities, some work on them, memory cached and released
x = torch.randn(64 << 20, device="cuda")
streams = [torch.cuda.Stream() for _ in range(6)] + [torch.cuda.Stream(priority=-1) for _ in range(2)]
for s in streams:
    with torch.cuda.stream(s):
        y = x * 2.0
torch.cuda.synchronize()
del y
torch.cuda.empty_cache()
base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "T", "--steps", "3", "--warmup", "1",
        "--views-per-gpu", "2", "--exchange", "factored", "--no-cpu-baseline", "--no-extras", "--views", "1", "--ring", "16",
        "--no-train-step"]
variants = {"streams2": ["--view-streams", "2"], "streams1": ["--view-streams", "1"],
            "streams2_hwq8": ["--view-streams", "2"], "plain_exchange": ["--view-streams", "2", "--exchange", "plain"]}
env0 = dict(os.environ, GSPLAT_AMD_BENCH_BACKEND="gloo", GSPLAT_AMD_DEBUG_BINS="1")
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env0.pop(k, None)
for name, extra in variants.items():
    env = dict(env0)
    if name.endswith("hwq8"):
        env["GPU_MAX_HW_QUEUES"] = "8"
    bad = 0
    for r in range(runs):
        t0 = time.time()
        try:
            res = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=240)
            ok = res.returncode == 0 and "implausible" not in res.stderr
            why = "" if ok else ("implausible" if "implausible" in res.stderr else f"rc {res.returncode}")
        except subprocess.TimeoutExpired:
            ok, why = False, "timeout"
        bad += 0 if ok else 1
        print(f"{name} run {r}: {'ok' if ok else 'FAILED ' + why} in {time.time() - t0:.0f} s", flush=True)
    print(f"== {name}: {bad} of {runs} failed", flush=True)
