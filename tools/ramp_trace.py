"""From a rocprofv3 --kernel-trace csv of `bench.py --steps K ...`: per 4-view step (delimited by the shared projection
launch) of the LAST window of steps, the summed duration of every kernel class and the step's span -- which kernels get
faster while the step time ramps down behind a fence?  usage: python tools/ramp_trace.py trace.csv [steps]"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]))
rows.sort()
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
marks = [i for i, r in enumerate(rows) if r[2].startswith("project_fwd_kernel<3, true, true")]
# the longest run of consecutive steps with gaps < 5 ms between shared projections = a timed window
wins, cur = [], [marks[0]]
for a, b in zip(marks, marks[1:]):
    if rows[b][0] - rows[a][0] < 5_000_000:
        cur.append(b)
    else:
        wins.append(cur); cur = [b]
wins.append(cur)
wins = [w for w in wins if len(w) >= K]
w = wins[0][:K + 1]   # the first long window = the value pass
names = ["composite_fwd", "composite_bwd_2p", "project_fwd", "project_bwd_batch", "depth_bucket", "emit_place", "radix_scatter", "emit_binhist", "bin_colscan", "radix_hist", "radix_scan", "bwd_prepare"]
print("step span_ms " + " ".join(n[:12] for n in names))
for a, b in zip(w, w[1:]):
    acc = collections.Counter()
    for s, e, n in rows[a:b]:
        for k in names:
            if n.startswith(k):
                acc[k] += e - s
    print(f"{w.index(a):3d} {(rows[b][0] - rows[a][0]) / 1e6:7.3f} " + " ".join(f"{acc[k] / 1e3:12.1f}" for k in names))
