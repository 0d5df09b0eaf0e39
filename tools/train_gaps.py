"""Idle gaps inside a training step from a rocprofv3 kernel trace (CSV): where the device waits between two kernels of a step.
python tools/train_gaps.py <kernel_trace.csv>"""
import csv, collections, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_step_advance" in r["Kernel_Name"]]
where, tot, pos = collections.Counter(), collections.defaultdict(float), collections.defaultdict(set)
steps = 0
busy = []
for a, b in zip(idx[20:-1], idx[21:]):
    steps += 1
    prev = None
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3)
    for n, r in enumerate(rows[a:b + 1]):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if prev is not None and (st - prev[0]) / 1e3 > 3:
            key = prev[1][:34] + " -> " + r["Kernel_Name"][:34]
            where[key] += 1; tot[key] += (st - prev[0]) / 1e3; pos[key].add(n)
        prev = (en, r["Kernel_Name"])
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:8]:
    print(f"{k:74s} n={where[k]:4d} avg gap {v / where[k]:8.1f} us  per step {v / steps:7.1f}  at kernel # {sorted(pos[k])[:4]}")
d = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in zip(idx[20:-1], idx[21:])]
print(f"step start-to-start: median {statistics.median(d):.1f} us; kernel time per step {statistics.median(busy):.1f} us; kernels per step {idx[30] - idx[29]}")
