"""Per-frame GPU idle gaps from a rocprofv3 kernel trace (last frame = between the last two project_bwd launches)."""
import csv, sys, re, glob
rows = sorted(csv.DictReader(open(glob.glob(sys.argv[1])[0])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "project_bwd" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1]
prev_end = int(rows[a - 1]["End_Timestamp"]); t0 = prev_end
busy = 0; gaps = []
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"]).split("(")[0][:40]
    gaps.append(((s - prev_end) / 1e3, n)); busy += e - s; prev_end = max(prev_end, e)
print(f"frame {(prev_end - t0)/1e3:.1f} us, kernels busy {busy/1e3:.1f} us, idle {(prev_end - t0 - busy)/1e3:.1f} us, launches {b - a + 1}")
for g, n in sorted(gaps, reverse=True)[:12]:
    print(f"  gap {g:7.1f} us before {n}")
