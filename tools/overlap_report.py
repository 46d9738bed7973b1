"""Developer aid: how much the kernels of the two view streams overlap in time.  Reads a rocprofv3 --kernel-trace csv of
`bench.py` (multi-view steps on several streams): per queue the busy time, the time with >= 2 kernels in flight, and the
average duration of each kernel when it ran alone vs beside a kernel of another queue.
usage: python tools/overlap_report.py p_kernel_trace.csv"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]))
rows.sort()
queues = collections.Counter(q for _, _, q, _ in rows)
print("dispatches per queue:", dict(queues))
# the busiest window with two queues active: take all dispatches of the queues other than the first-seen default one
events = []
for s, e, q, n in rows:
    events.append((s, 1, q)); events.append((e, -1, q))
events.sort()
active, last, t_any, t_two = collections.Counter(), None, 0, 0
for ts, d, q in events:
    if last is not None:
        nq = sum(1 for v in active.values() if v > 0)
        if nq >= 1: t_any += ts - last
        if nq >= 2: t_two += ts - last
    active[q] += d; last = ts
print(f"time with a kernel in flight {t_any / 1e6:.2f} ms, with kernels of two queues in flight {t_two / 1e6:.2f} ms ({100.0 * t_two / max(t_any, 1):.1f} %)")
# per kernel: alone vs overlapped (any overlap with a kernel of another queue)
ivs = collections.defaultdict(list)
for s, e, q, n in rows: ivs[q].append((s, e))
def overlapped(s, e, q):
    tot = 0
    for q2, lst in ivs.items():
        if q2 == q: continue
        import bisect
        i = bisect.bisect_left(lst, (s, 0)) - 1
        i = max(i, 0)
        while i < len(lst) and lst[i][0] < e:
            tot += max(0, min(e, lst[i][1]) - max(s, lst[i][0])); i += 1
    return tot
stat = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for s, e, q, n in rows:
    ov = overlapped(s, e, q)
    st = stat[n]
    if ov > 0.5 * (e - s): st[2] += 1; st[3] += (e - s) / 1e3
    elif ov == 0: st[0] += 1; st[1] += (e - s) / 1e3
print(f"{'kernel':46s} {'alone n':>8s} {'avg us':>8s} {'beside n':>9s} {'avg us':>8s}")
for n, (a, ta, b, tb) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    if a + b < 10: continue
    print(f"{n:46s} {a:8d} {ta / max(a, 1):8.1f} {b:9d} {tb / max(b, 1):8.1f}")
