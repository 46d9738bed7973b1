"""Compose profiles/<name>_summary.md (+ copy <name>_kernel_stats.csv / <name>_traffic.json) from what
tools/round_artifacts.sh TAG left under gpurun_out/TAG.

usage: python tools/make_round_summary.py gpurun_out/<TAG> <name> "<title>" """
import json
import os
import shutil
import sys

src, name, title = sys.argv[1], sys.argv[2], sys.argv[3]
tag = os.path.basename(src.rstrip("/"))
out = "profiles"
shutil.copy(f"{src}/stats/p_kernel_stats.csv", f"{out}/{name}_kernel_stats.csv")
shutil.copy(f"{src}/traffic.json", f"{out}/{name}_traffic.json")
profiled = open(f"{src}/bench_profiled.json").read().strip().splitlines()[-1]
default = open(f"{src}/bench_default.json").read().strip().splitlines()[-1]
pj = json.loads(profiled)
table = open(f"{src}/summary_table.md").read().rstrip()
solo = open(f"{src}/summary_table_solo.md").read().rstrip() if os.path.exists(f"{src}/summary_table_solo.md") else ""
if solo:
    shutil.copy(f"{src}/stats_solo/p_kernel_stats.csv", f"{out}/{name}_kernel_stats_solo.csv")
solo = "\n".join(ln for ln in solo.splitlines() if not ln.startswith("| at::") and "rocclr" not in ln)
tr = json.load(open(f"{src}/traffic.json"))
import subprocess
tr["csrc_rev"] = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD:3d-gaussian-splatting_amd/csrc"], text=True).strip()
json.dump(tr, open(f"{out}/{name}_traffic.json", "w"), indent=1)   # (the copy under profiles/ carries the sources' revision)


def counters_row(label, k):
    r = tr["kernels"][k]
    sh = r.get("wave_cycle_shares", {})
    vals = list(sh.values()) + [0, 0, 0]
    return (f"| `{r['kernel']}` | {r.get('valu_insts', 0) / 1e6:.1f} M | {r.get('lds_insts', 0) / 1e6:.1f} M | "
            f"{r.get('wave_cycles', 0) / 1e6:.0f} M | {100 * vals[2]:.0f}% | {100 * vals[1]:.0f}% | {100 * vals[0]:.0f}% | "
            f"{100 * r.get('wait_issue_lds_frac', 0):.0f}% | {r['hbm_bytes_per_launch'] / 1e6:.0f} MB | "
            f"{r['atomic_requests'] / 1e6:.2f} M |")


binning = ", ".join(f"{k} {v['hbm_bytes_per_launch'] / 1e6:.1f}" for k, v in sorted(tr["binning_kernels"].items()))
proj = ", ".join(f"{k} {tr['kernels'][k]['hbm_bytes_per_launch'] / 1e6:.0f}" for k in ("gsa_project_fwd", "gsa_project_bwd")
                 if k in tr["kernels"])
text = f"""# {title} -- rocprofv3 --kernel-trace --stats

Produced by `tools/round_artifacts.sh {tag}` on one MI355X box (default bench, the profiled bench, the PMC passes
behind `profiles/{name}_traffic.json`, in that order) and `tools/make_round_summary.py`.

Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps {pj['steps']} --warmup {pj['warmup']} --no-cpu-baseline`
(every pass of bench.py is traced: the {pj['config']['views_per_gpu_per_step']}-view steps of the timed region -- warm-up, settling, {pj['steps']} timed, {pj['steps']} with one event
per step -- the single-view steps, the instrumented single-view steps, the rotating-view pass;
`profiles/{name}_kernel_stats.csv` is the raw stats file, `profiles/{name}_traffic.json` was measured on csrc revision {tr['csrc_rev']}).

bench.py line of the same (profiled) run:

```
{profiled}
```

Un-profiled default run (`python bench.py`, CPU baseline legs included) on the same build and box:

```
{default}
```

The kernels ALONE -- `rocprofv3 --kernel-trace --stats -- python3 bench.py --profile-run --steps 30 --warmup 5`: single-view
steps on one stream, the protocol `roofline` and `stage_ms` are quoted on (`profiles/{name}_kernel_stats_solo.csv`):

{solo}

The dominant kernel's average duration under the profiler (`composite_bwd_2p_kernel<3>`, this table) agrees with the
HIP-event average bench.py reports in `roofline.avg_launch_ms` (un-profiled; the profiler adds a few percent).

The same kernels in the whole bench (the command at the top: mostly {pj['config']['views_per_gpu_per_step']}-view steps on two streams).  Here a
kernel's duration INCLUDES the time it waits beside another stream's sweep -- the LDS-hungry binning kernels
(`radix_scatter`, `depth_bucket_sort`, `emit_binhist`, `emit_place`, `bin_colscan`) sit out a sweep that leaves a CU no
LDS, `project_fwd` runs beside one at half speed (`profiles/r03_host_pipeline.md`); the training-step leg's kernels
(`loss_strip`, `adam`, `sh_grad_expand`, `activate_*`, `step_report`) are in this table too -- so the averages are longer than
above although the views are shorter:

{table}

Binning is `radix_hist<unsigned int>` + `radix_scan_rows` + `radix_scatter<unsigned int>` (ONE global pass: depth
buckets) + `depth_bucket_sort_kernel` (local sorts fused with the scan and the block-owner table; it also leaves the
control words in the caller's pinned slot -- no copy kernel), then `emit_binhist` + `bin_colscan` + `emit_place` (records
generated from the depth-ordered rectangles -- of the 32 x 32 bins: {pj['config'].get('binned_records')} records where the 16 x 16 rule has
{pj['config']['intersections_M']} -- counted per bin and block, the counts scanned into the lists' offsets, every id stored
at its final place, since r04 with the record's QUADRANT BOX inside its bin in the id's top byte:
`profiles/r03_binning_direct.md`, `profiles/r04_sweep_ablation.md`) + `tile_schedule`: 8 launches, 13 per forward + backward view.
The `scan_*` kernels, `tile_offsets_kernel` and the second / third `radix_*<unsigned int>` calls belong to the first call
of the scene shape (LSD path, which learns the key range); `__amd_rocclr_copyBuffer` is the scene upload and that first
call's read-back.

Counters of the compositor kernels (`profiles/{name}_traffic.json`, separate `--pmc` passes of the same command):

| kernel | VALU wave-instr | LDS wave-instr | wave cycles | issuing | waiting to issue | waiting for data | of which LDS-pipe stall | HBM bytes (FETCH x2 + WRITE) | float-atomic requests |
|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|
{counters_row('bwd', 'gsa_composite_bwd')}
{counters_row('fwd', 'gsa_composite_fwd')}

(`composite_fwd` now also clears the 64 MB of gradient records the backward sweep accumulates into: 64 MB of its writes.)

HBM traffic per launch of the projection kernels (MB): {proj}.
HBM traffic per launch of the binning kernels (MB): {binning}.
"""
if tr.get("step_kernels"):
    sk = tr["step_kernels"]
    total = sum(v["hbm_bytes_per_launch"] * v["launches_per_view"] for v in sk.values())
    rows = "\n".join(f"| `{k}` | {v['hbm_bytes_per_launch'] / 1e6:.1f} | {v['launches_per_view']:g} |"
                     for k, v in sorted(sk.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_view"])
                     if v["hbm_bytes_per_launch"] * v["launches_per_view"] >= 1e5)
    text += f"""
The kernels of the MULTI-VIEW step (`bench.py --profile-run --profile-views 4`: 4 views per step on one stream; same PMC
passes).  The views of a step share ONE projection launch (`project_fwd_kernel<3, true, true, 3>`: geometry and SH rows read
once for the 4 cameras) and ONE projection-backward launch (`project_bwd_batch_kernel<3>`: every gradient written once, the
SH-coefficient gradient formed in registers -- no per-view `project_bwd`, no `sh_grad_expand`):

| kernel | HBM MB per launch | launches per view |
|---|---:|---:|
{rows}

HBM bytes per view of the 4-view step: {total / 1e6:.0f} MB (the single-view step: the sum of the per-launch figures above,
with 376 + 400 MB of projection per view where the shared launches move 199 + 190).
"""
open(f"{out}/{name}_summary.md", "w").write(text)
print(f"wrote {out}/{name}_summary.md ({len(text)} bytes)")
