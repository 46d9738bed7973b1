"""Build profiles/<name>_traffic.json from the rocprofv3 --pmc passes of tools/collect_pmc.sh (fetch / write / insts /
active dirs): HBM traffic per launch of EVERY kernel of the bench step (the four single-kernel stages and every
binning kernel), float-atomic requests, and for the compositor kernels the SQ counters that say what they wait for.

usage: python tools/make_traffic_json.py <pmc-dir> <tag> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(path, pat=".", mean=False):
    """{kernel: {counter: value}} -- last dispatch of each kernel name (or the mean over its dispatches)."""
    rows = list(csv.DictReader(open(path)))
    by = collections.OrderedDict()
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        if not re.search(pat, name):
            continue
        by.setdefault(name, collections.OrderedDict()).setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    out = {}
    for k, disp in by.items():
        vals = list(disp.values())
        if mean:
            keys = vals[-1].keys()
            out[k] = {c: sum(v.get(c, 0.0) for v in vals[1:] or vals) / max(len(vals[1:] or vals), 1) for c in keys}
        else:
            out[k] = vals[-1]
    return out


def first(pattern):
    hits = glob.glob(pattern)
    return hits[0] if hits else None


out_dir, tag = sys.argv[1], sys.argv[2]
OURS = "composite|project|radix|scan_|emit_|tile_|depth_bucket|depth_table|bin_col|bwd_prepare"
fetch = per_kernel(first(f"{out_dir}/fetch/*/*counter_collection.csv"), OURS)
write = per_kernel(first(f"{out_dir}/write/*/*counter_collection.csv"), OURS)
insts = per_kernel(first(f"{out_dir}/insts/*/*counter_collection.csv"), "composite") if first(f"{out_dir}/insts/*/*counter_collection.csv") else {}
active = per_kernel(first(f"{out_dir}/active/*/*counter_collection.csv"), "composite") if first(f"{out_dir}/active/*/*counter_collection.csv") else {}
cal_f = per_kernel(first(f"{out_dir}/fetch_cal/*/*counter_collection.csv"), "reduce_kernel")
cal_fetch_kb = list(cal_f.values())[0].get("FETCH_SIZE", 0) if cal_f else 0

STAGE = {"composite_bwd_2p_kernel": "gsa_composite_bwd", "composite_bwd_scan_kernel": "gsa_composite_bwd",
         "composite_bwd_list_kernel": "gsa_composite_bwd", "composite_bwd_kernel": "gsa_composite_bwd",
         "composite_fwd_kernel": "gsa_composite_fwd", "project_fwd_kernel": "gsa_project_fwd",
         "project_bwd_kernel": "gsa_project_bwd"}
res = {"_about": f"Per launch, from rocprofv3 --pmc in separate passes (FETCH_SIZE; WRITE_SIZE + TCC_EA0_ATOMIC_sum; two SQ "
                 f"passes), bench.py C3 workload, {tag} build; last dispatch of each kernel.  FETCH_SIZE (KB) is doubled per "
                 f"/opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read; calibration in the same "
                 f"pass: a 256 MiB torch reduce read FETCH_SIZE = {cal_fetch_kb:.0f} KB); for scattered 4..16-B accesses that "
                 f"factor is an upper bound.  WRITE_SIZE (KB) is exact for streaming stores and counts 64 B per float-atomic "
                 f"request.  valu_util = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: the share of its resident time a wave spends "
                 f"issuing vector ALU instructions; wave_cycle_shares: the disjoint buckets SQ_WAIT_ANY / SQ_WAIT_INST_ANY / "
                 f"SQ_ACTIVE_INST_ANY of the guide's PMC table over SQ_WAVE_CYCLES; limiter: the largest of them.",
       "kernels": {}, "binning_kernels": {}}
for k in sorted(set(fetch) | set(write)):
    base = k.split("<")[0]
    f = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
    w = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    a = write.get(k, {}).get("TCC_EA0_ATOMIC_sum", 0.0)
    row = {"kernel": k, "fetch_size_bytes_raw": f, "fetch_bytes_corrected": 2 * f, "write_bytes": w, "atomic_requests": a,
           "hbm_bytes_per_launch": 2 * f + w,
           # launches per fwd+bwd view on the steady-state path (one-pass binning: every kernel runs once a view)
           "launches_per_view": 1}
    if base in STAGE:
        i, ac = insts.get(k, {}), active.get(k, {})
        if i and ac:
            wc = i.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            buckets = {"wait_data (s_waitcnt: LDS / memory latency)": i.get("SQ_WAIT_ANY", 0.0) / wc,
                       "wait_issue (VALU / LDS pipe busy)": i.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                       "issuing": ac.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
            row.update({"valu_insts": i.get("SQ_INSTS_VALU"), "lds_insts": i.get("SQ_INSTS_LDS"), "wave_cycles": wc,
                        "valu_util": round(ac.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
                        "wait_issue_lds_frac": round(ac.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4),
                        "wave_cycle_shares": {n: round(v, 4) for n, v in buckets.items()},
                        "limiter": "valu+lds issue (not HBM): " + max(buckets, key=buckets.get)})
        res["kernels"][STAGE[base]] = row
    elif base != "bwd_prepare_kernel":
        res["binning_kernels"][k] = row
    else:
        res["kernels"]["bwd_prepare"] = row
# the kernels of the MULTI-VIEW step (4 views on one stream: tools/collect_pmc.sh's *_step passes): the per-step projection
# launches serve all views of the step (launches_per_view = 1 / views), everything else runs once a view
fs, ws = first(f"{out_dir}/fetch_step/*/*counter_collection.csv"), first(f"{out_dir}/write_step/*/*counter_collection.csv")
if fs and ws:
    VIEWS = 4
    fstep, wstep = per_kernel(fs, OURS), per_kernel(ws, OURS)
    res["step_kernels"] = {}
    for k in sorted(set(fstep) | set(wstep)):
        f = fstep.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
        w = wstep.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        per_step = bool(re.search(r"project_fwd_kernel<[^>]*true>$|project_bwd_batch_kernel|sh_grad_expand", k))
        res["step_kernels"][k] = {"kernel": k, "fetch_bytes_corrected": 2 * f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
                                  "launches_per_view": 1.0 / VIEWS if per_step else 1}
    res["_about_step_kernels"] = (f"the same counters for the kernels of the {VIEWS}-view step (bench.py --profile-run --profile-views "
                                  f"{VIEWS}: one stream); launches_per_view: 1 / {VIEWS} for the per-step projection launches")
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("hbm_bytes_per_launch", "valu_util", "limiter", "atomic_requests")}
                  for k, v in res["kernels"].items()}, indent=1))
print("binning kernels:", {k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in res["binning_kernels"].items()}, "MB")
