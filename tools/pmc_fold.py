"""Per-kernel mean of every counter in a rocprofv3 counter_collection CSV: python tools/pmc_fold.py <csv> <out.json> "<command>" """
import csv, json, re, sys
from collections import defaultdict
tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
out = {"command": sys.argv[3] if len(sys.argv) > 3 else "", "per_launch_mean": {k: {c: tot[k][c] / n[k][c] for c in tot[k]} for k in tot}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["per_launch_mean"].items():
    if "conv" in k or "gemm" in k:
        w = v.get("SQ_WAVE_CYCLES", 0) or 1
        print(f"{k:48s} MFMA_BUSY {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3e} WAIT_ANY {v.get('SQ_WAIT_ANY', 0) / w:.2f} WAIT_INST {v.get('SQ_WAIT_INST_ANY', 0) / w:.2f} "
              f"ACTIVE {v.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f} LDS_CONFLICT {v.get('SQ_LDS_BANK_CONFLICT', 0):.0f} BUSY {v.get('SQ_BUSY_CYCLES', 0):.3e}")
