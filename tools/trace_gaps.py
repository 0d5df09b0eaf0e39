"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
rows.sort()
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
tot_gap = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > 0:
        tot_gap += g
        k = f"{n0} -> {n1}"
        gaps[k][0] += g
        gaps[k][1] += 1
print(f"kernels {len(rows)} span {span/1e6:.1f} ms busy {busy/1e6:.1f} ms ({busy/span:.1%}) gaps {tot_gap/1e6:.1f} ms")
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {g/1e6:8.1f} ms  n={n:6d}  mean {g/n/1e3:7.1f} us  {k}")
