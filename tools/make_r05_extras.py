"""Compose the round-5 evidence files under profiles/ from what tools/r05_extras.sh left under gpurun_out/r05x:

  profiles/r05_bench_driver.json        the driver's command (bench.py --steps 20 --warmup 5), un-profiled
  profiles/r05_c5_summary.md            BASELINE config C5 on one GPU: bench line + kernel stats
  profiles/r05_train_step_summary.md    the full optimisation step: tools/bench_train_step.py + kernel stats + host share
  profiles/r05_parity_*.json            per-tensor errors of the config-sized parity tests (incl. the C2 step of 4 cameras)

usage: python tools/make_r05_extras.py [gpurun_out/r05x]"""
import glob
import json
import os
import re
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05x"
out = "profiles"


def line(name):
    return open(f"{src}/{name}").read().strip().splitlines()[-1]


def table(name, rows=24, drop=()):
    tab = open(f"{src}/{name}").read().rstrip().splitlines()
    return "\n".join(ln for ln in tab[:rows] if not any(d in ln for d in drop))


# ---- driver-style line ------------------------------------------------------------------------------------------------
drv = json.loads(line("bench_driver.json"))
json.dump(drv, open(f"{out}/r05_bench_driver.json", "w"), indent=1)

# ---- C5 ---------------------------------------------------------------------------------------------------------------
c5 = json.loads(line("bench_c5.json"))
s5 = c5["stage_ms"]
text = f"""# r05 -- BASELINE config C5 size on one GPU: 5 M gaussians, 3840 x 2160, SH 3, forward + backward

`python bench.py --workload C5 --no-cpu-baseline --no-train-step --steps 20 --warmup 3 --views 1` (4 views per step on two
streams -- 4 consecutive cameras of a ring of {c5['config'].get('ring_cameras', '?')}, the window moving every step -- the speculation checked once per
step; `single_view` is one view per step on one stream):

```
{line("bench_c5.json")}
```

{c5['config']['ms_per_view']} ms per 4K view in the 4-view step ({c5['value']} Mpix/s; fixed cameras: {c5.get('ms_per_view_fixed_cameras')}), {c5['single_view']['ms']} ms for
a view alone (r04: 1.81 / 2.29; r03: 2.10 / 2.35; r02: 2.58).
Stages (single view): binning {s5['binning']} (r04 0.520), forward {s5['gsa_composite_fwd']} (0.414), backward {s5['gsa_composite_bwd']} (0.632),
projection {s5['gsa_project_fwd']} + {s5['gsa_project_bwd']} ms (0.365 + 0.355).
`over_peak` / `roofline_whole_step.frac` > 1: SURVEY's formula prices {c5['config']['intersections_M']} records of the 16 x 16 rule where
{c5['config']['binned_records']} records of 64 x 64 bins are placed, and whole SH rows where the backward reads the digest -- the contractual
fraction flatters at this size; the table below is what the kernels do.

**The depth ordering at this population (VERDICT r04 item 3; target: binning <= 0.40 ms, a view <= 2.15 ms) -- not met.**
What was done: (1) `test_depth_order_at_the_reference_population[5M, 10M]` holds the ordering bit-equal to a stable sort
at the populations the reference names (`trainer.py:92-96`, `training_config.py:22`); (2) a compact LSD scatter
(`radix_scatter_compact_kernel`: 26.6 KB of LDS instead of 38 KB, five workgroups a CU, used above 1024 blocks = ~4 M keys)
and the scan leaving the control words in the pinned slot itself; binning 0.520 -> {s5['binning']} ms.  What bounds the pass: the
scatter ranks 4096 keys per workgroup with sixteen rounds of ballot-based peer ranking (eight ballots per round per wave) --
instruction-bound at ~20 G keys/s over hist + scan + scatter, not HBM-bound (80 MB per pass in ~60 us); the bucketed
ordering's local sorts run at the same rate per key (37.6 us per 1 M keys), so lifting the bucket count (two-level buckets)
buys launches, not throughput, at 5 M: costed at 84 (global pass) + ~200 (local sorts of 5 M keys) us against the three LSD
passes' ~250.  A one-sweep pass (chained scan with look-back) would remove hist + scan (21 of 84 us per pass); not built.
The rest of the stage is M-proportional: `emit_place<8>` places 17.7 M bin records in ~170 us.

`rocprofv3 --kernel-trace --stats -- python3 bench.py --workload C5 --profile-run --steps 10 --warmup 3` (single-view steps):

{table("c5_table.md", 22)}
"""
open(f"{out}/r05_c5_summary.md", "w").write(text)
shutil.copy(f"{src}/stats_c5/p_kernel_stats.csv", f"{out}/r05_c5_kernel_stats.csv")

# ---- train step -------------------------------------------------------------------------------------------------------
log = [ln for ln in open(f"{src}/train_step.log").read().splitlines() if ln.startswith("fused=")]
host = open(f"{src}/train_host.txt").read().strip().splitlines()[-12:] if os.path.exists(f"{src}/train_host.txt") else []
host0 = open(f"{src}/train_host_nochain.txt").read().strip().splitlines()[-12:] if os.path.exists(f"{src}/train_host_nochain.txt") else []
def work(lines):
    m = re.search(r"median ([0-9.]+) ms.*waiting for control words ([0-9.]+) ms", " ".join(lines))
    return f"{float(m.group(1)) - float(m.group(2)):.2f}" if m else "?"


host = [ln for ln in host if "amdgpu.ids" not in ln]
host0 = [ln for ln in host0 if "amdgpu.ids" not in ln]
text = f"""# r05 -- BASELINE config C3 as the loop runs it: full optimisation steps (4 views -> loss -> backward -> Adam), 1 M gaussians, 1080p, SH 3

`python tools/bench_train_step.py` on one MI355X (un-profiled):

```
{chr(10).join(log)}
```

(r04: 11.4 -> 2.53-2.62 ms; r03: 11.06 -> 3.22; r02: 11.3 -> 3.44; r01: 12.5 -> 4.2.  The metrics of the fused path are read
one step late, so its loss is the previous step's.)  `bench.py` reports the same step as `train_step` in its JSON line
(driver-style run of this box: {drv.get('train_step', {}).get('ms_per_step')} ms).

What changed in r05: the fused step queues every view through ONE C call per direction (`gsa_view_fwd`: depth ordering,
placement with the launch order, forward sweep; `gsa_view_bwd`: tile re-ordering, reverse sweep), ABI 13 -- `ViewChain` in
`gsplat_amd/stages.py` -- instead of five ctypes calls with ~45 pointer arguments and a dozen allocations per view.
`tools/train_step_host.py 60` (host time inside `training_step`: queueing vs waiting for the step's control words):

with the per-view calls (default):

```
{chr(10).join(host)}
```

stage by stage (`GSPLAT_AMD_VIEW_CALLS=0`):

```
{chr(10).join(host0)}
```

Host WORK per step = host time in `training_step` minus the wait for the step's control words: {work(host)} ms with the per-view
calls, {work(host0)} ms stage by stage (earlier boxes of the round: 0.81 / 1.11 and 0.83 / 1.06); the step is device-paced either way.

`rocprofv3 --kernel-trace --stats -- python3 tools/bench_train_step.py` (FUSED_ONLY=1: 3 warm-up + 10 timed steps of 4 views;
durations INCLUDE the time a kernel waits beside another stream's sweep, which is why the short kernels read long):

{table("train_table.md", 40, ("at::native::elementwise_kernel_manual", "distribution_"))}
"""
open(f"{out}/r05_train_step_summary.md", "w").write(text)
shutil.copy(f"{src}/stats_train/p_kernel_stats.csv", f"{out}/r05_train_step_kernel_stats.csv")

# ---- parity evidence --------------------------------------------------------------------------------------------------
copied = []
for root in (src, os.path.dirname(src.rstrip("/"))):
    for f in sorted(glob.glob(f"{root}/parity_c2_*.json") + glob.glob(f"{root}/parity_c3crop_cam*.json") +
                    glob.glob(f"{root}/parity_c3_large_coordinates.json")):
        dst = f"{out}/r05_{os.path.basename(f)}"
        if dst not in copied:
            shutil.copy(f, dst)
            copied.append(dst)
stages = {}
for f in sorted(glob.glob(f"{src}/stage_*.json")):
    stages[re.sub(r"\.json$", "", os.path.basename(f))] = json.load(open(f))
if stages:
    worst = max((v2["hip_vs_fp64"] for v in stages.values() for v2 in v.values()), default=0.0)
    json.dump({"_about": "tests/test_hip_parity.py stage tests (GSPLAT_AMD_EVIDENCE): per tensor the HIP path's and the fp32 oracle's "
                         "relative error against the fp64 oracle; the bound is 1e-4 outright on every tensor (STAGE_RELAXED is empty)",
               "worst_hip_vs_fp64": worst, "tests": stages}, open(f"{out}/r05_parity_stages.json", "w"), indent=1)
    print("worst stage error", worst)
if os.path.exists(f"{src}/careful.txt"):
    shutil.copy(f"{src}/careful.txt", f"{out}/r05_careful_path.txt")
print("files:", sorted(x for x in os.listdir(out) if x.startswith("r05")))
