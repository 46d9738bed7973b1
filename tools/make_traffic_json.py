"""Build profiles/<name>_traffic.json from the rocprofv3 --pmc passes of tools/collect_pmc.sh (fetch / write dirs)."""
import csv, glob, json, re, sys, collections

def last_dispatch(path, pat):
    rows = list(csv.DictReader(open(path)))
    by = collections.OrderedDict()
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
        if not re.search(pat, name): continue
        by.setdefault(name, {}).setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return {k: list(v.values())[-1] for k, v in by.items()}

out_dir, tag = sys.argv[1], sys.argv[2]
fetch = last_dispatch(glob.glob(f"{out_dir}/fetch/*/*counter_collection.csv")[0], "composite|project")
write = last_dispatch(glob.glob(f"{out_dir}/write/*/*counter_collection.csv")[0], "composite|project")
cal_f = last_dispatch(glob.glob(f"{out_dir}/fetch_cal/*/*counter_collection.csv")[0], "reduce_kernel")
cal_w = last_dispatch(glob.glob(f"{out_dir}/write_cal/*/*counter_collection.csv")[0], "FillFunctor|vectorized_elementwise")
names = {"composite_bwd_kernel": "gsa_composite_bwd", "composite_bwd_list_kernel": "gsa_composite_bwd",
         "composite_fwd_kernel": "gsa_composite_fwd",
         "project_fwd_kernel": "gsa_project_fwd", "project_bwd_kernel": "gsa_project_bwd"}
res = {"_about": f"HBM traffic per launch from rocprofv3 --pmc (separate passes: FETCH_SIZE; WRITE_SIZE + TCC_EA0_ATOMIC_sum), "
                 f"bench.py C3 workload, {tag} build; last dispatch of each kernel. FETCH_SIZE (KB) is doubled per "
                 f"/opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read; calibration in the same "
                 f"passes: a 256 MiB torch reduce read FETCH_SIZE = {list(cal_f.values())[0].get('FETCH_SIZE', 0):.0f} KB); for the "
                 f"compositor's 16-B gathers that factor is an upper bound. WRITE_SIZE (KB) is exact for streaming stores and counts "
                 f"64 B per float-atomic request.", "kernels": {}}
for k, nice in names.items():
    if k not in fetch and k not in write:
        continue
    f = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
    w = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    a = write.get(k, {}).get("TCC_EA0_ATOMIC_sum", 0.0)
    res["kernels"][nice] = {"fetch_size_bytes_raw": f, "fetch_bytes_corrected": 2 * f, "write_bytes": w,
                            "atomic_requests": a, "hbm_bytes_per_launch": 2 * f + w}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
