"""Mean of one PMC counter per kernel from a rocprofv3 counter_collection CSV: python3 tools/pmc_one.py <csv> [substr]"""
import csv, sys, collections
t = collections.defaultdict(lambda: collections.defaultdict(list))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        t[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in t.items():
    print(k, {c: f"{sum(x) / len(x):.4g} (n={len(x)})" for c, x in v.items()})
