"""Developer aid: a window of a rocprofv3 --kernel-trace csv as a two-queue timeline (us).  usage: timeline.py csv [start_fraction] [rows]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]),
                 r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:30]))
rows.sort()
i0 = int(len(rows) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.2))
t0 = rows[i0][0]
qs = sorted({q for _, _, q, _ in rows})
for s, e, q, n in rows[i0:i0 + (int(sys.argv[3]) if len(sys.argv) > 3 else 70)]:
    print(f"{' ' * 44 * qs.index(q)}q{q} {(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f} {n}")
