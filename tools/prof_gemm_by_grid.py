"""Summarise a rocprofv3 kernel trace by (kernel, grid): usage python tools/prof_gemm_by_grid.py <kernel_trace.csv> [filter]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else "k_gemm_f32"
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if flt in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:34], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, v[0], round(v[1] / v[0], 1), "us")
    tot += v[1]
print("total us per 104 steps/104:", round(tot / 104, 1))
