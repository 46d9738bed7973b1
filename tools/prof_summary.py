"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short markdown table."""
import csv, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    name = m.group(1) if m else name
    return name[:70]

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
print("| kernel | calls | avg us | total ms | % |")
print("|---|---:|---:|---:|---:|")
for r in rows:
    print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.1f} |")
if steps:
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f"\nsum of kernel time / {steps} steps = {tot/steps/1e6:.3f} ms per step")
