"""Compose profiles/r03_train_step_summary.md, r03_c5_summary.md and r03_views_streams_sweep.md from what
tools/r03_evidence.sh left under gpurun_out/r03x.   usage: python tools/make_r03_extras.py [gpurun_out/r03x]"""
import glob, json, os, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03x"

def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])

def table(path, drop_torch=False):
    rows = open(path).read().rstrip().splitlines()
    return "\n".join(r for r in rows if not (drop_torch and r.startswith("| at::")))

# ---- training step
log = [ln for ln in open(f"{src}/train_step.log").read().splitlines() if ln.startswith("fused=")]
fused_ms = float(log[-1].split(":")[1].split("ms")[0])
open("profiles/r03_train_step_summary.md", "w").write(f"""# r03 -- BASELINE config C3 as the loop runs it: full optimisation steps (4 views -> loss -> backward -> Adam), 1 M gaussians, 1080p, SH 3

`python tools/bench_train_step.py` on one MI355X (un-profiled), round-3 build (one-pass tile binning):

```
{chr(10).join(log)}
```

(r02: 11.3 -> 3.44 ms; r01: 12.5 -> 4.2 ms.)  Per step of the fused path: 4 x (project_fwd, binning, composite_fwd, view loss,
composite_bwd, project_bwd_factored) + activations + SH-gradient rebuild + Adam; view j + 1's front end (projection +
binning) runs on a side stream under view j's sweeps, which is why the binning kernels below show long durations: the
LDS-hungry ones wait for the sweep beside them to end (`profiles/r03_host_pipeline.md`).

`rocprofv3 --kernel-trace --stats -- python3 tools/bench_train_step.py` (FUSED_ONLY=1: 3 warm-up + 10 timed steps of 4 views):

{table(f"{src}/train_step_table.md")}
""")
# ---- C5
c5 = last_json(f"{src}/c5.json")
line = open(f"{src}/c5.json").read().strip().splitlines()[-1]
open("profiles/r03_c5_summary.md", "w").write(f"""# r03 -- BASELINE config C5 size on one GPU: 5 M gaussians, 3840 x 2160, SH 3, forward + backward

`python bench.py --workload C5 --no-cpu-baseline --steps 20 --warmup 3 --views 1` (4 views per step on two streams, the
speculation checked once per step; the `single_view` object is one view per step on one stream):

```
{line}
```

{c5['config']['ms_per_view']} ms per 4K view in the 4-view step ({c5['value']} Mpix/s), {c5['single_view']['ms']} ms for a view alone (r02: 2.58; r01: 3.9).
`over_peak` lists the stages (and the whole step) whose SURVEY-8d "algorithmic" rate exceeds the 8 TB/s peak: the formula
prices 112 M records of the 16 x 16 rule where {c5['config'].get('binned_records')} bin records are placed, and the whole 192-B SH rows where the backward
reads the 48-B digest -- the contractual fraction flatters at this size (`roofline_whole_step.frac` > 1); what the kernels
do is in the table below.  5 M gaussians are beyond the bucketed depth ordering (an average bucket of N / 255 keys no longer
fits the 8192-key LDS capacity of the local sorts, and the 8-bit radix machinery has no more buckets to give): the LSD path
(three `radix_scatter<unsigned int>` passes per view below) orders the depths.  The 8160 bins of 32 x 32 of a 4K frame are
beyond the one-pass placement (2047 bins: its per-bin counters live in LDS), so frames that large are binned at 64 x 64
(`config.bin_tile` = {c5['config'].get('bin_tile')}: 2040 bins, {c5['config'].get('binned_records')} records where 32 x 32 bins have 40.1 M): {c5['stage_ms']['binning']} ms of
binning in all (0.79 with 32 x 32 bins and the two-pass tile sort, the view 2.31 ms).

`rocprofv3 --kernel-trace --stats -- python3 bench.py --workload C5 --profile-run --steps 10 --warmup 3` (single-view steps):

{table(f"{src}/c5_table.md")}
""")
# ---- sweep
rows = []
for f in sorted(glob.glob(f"{src}/sweep_*.json")):
    d = last_json(f)
    c = d["config"]
    rows.append((c["views_per_gpu_per_step"], c["view_streams"], c["ms_per_view"], d["value"], d["single_view"]["ms"], c.get("speculation_check", "")))
rows.sort()
body = "\n".join(f"| {b} | {s} | {ms} | {v} | {sv} | {chk} |" for b, s, ms, v, sv, chk in rows)
open("profiles/r03_views_streams_sweep.md", "w").write(f"""# r03 -- views per step x HIP streams (bench.py, C3, one box, 50 steps each)

A step renders B views and sums their gradients into one bucket; the views alternate on S streams (view j + 1's projection
and binning -- HBM- and latency-bound -- beside view j's sweeps -- vector-issue-bound); with B > 1 the SH gradient is kept
factored per view and rebuilt once per step, and the views' speculation (record capacity, depth-key range) is checked once
per step, behind the last view, so the host queues a step ahead of the device (`profiles/r03_host_pipeline.md`).

| views per step | streams | ms per view | Mpix/s | single view ms (same run) | speculation check |
|---:|---:|---:|---:|---:|---|
{body}

B = 4 on two streams is the default (BASELINE config C3 is "B = 4 views per step"): about 10 % faster per view than one view
per step.  Before the deferred check (the host waited for every view's control words inside its call) the same table read
0.609 for B = 4 / S = 2 against 0.647 for single-view steps on the two-pass binning.
""")
print("wrote profiles/r03_train_step_summary.md, r03_c5_summary.md, r03_views_streams_sweep.md")
