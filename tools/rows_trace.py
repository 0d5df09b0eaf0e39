"""Per-kernel time vs executed rows from a rocprofv3 kernel trace of tools/rows_sweep.py-like launches.
  run:   rocprofv3 --kernel-trace -d gpurun_out/rt -o rt -- python3 tools/rows_trace.py run 256 8192 256
  fold:  python3 tools/rows_trace.py fold gpurun_out/rt/rt_kernel_trace.csv 256 8192 256"""
import sys, os, csv, re, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
mode = sys.argv[1]
REPS = 3
if mode == "run":
    import numpy as np
    from alphazero_rs_amd import engine as azeng
    from _states import random_states
    lo, hi, step = (int(x) for x in sys.argv[2:5])
    e = azeng.Engine(device=0, max_batch=8192, diag=True)
    e.net_init_random(0, 1)
    for kv in filter(None, os.environ.get("OPT", "").split(",")):
        k, v = kv.split("=")
        e.set_option(k, int(v))
    uniq = random_states(8192, 3)
    for L in range(lo, hi + 1, step):
        for _ in range(REPS):
            e.predict_states(uniq[:L], 0)
else:
    path = sys.argv[2]
    lo, hi, step = (int(x) for x in sys.argv[3:6])
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    fw, cur = [], []
    for r in rows:
        name = r["Kernel_Name"]
        if "az::" not in name: continue
        short = re.sub(r"\(.*", "", name.replace("void ", "").replace("az::", ""))
        cur.append((short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        if short.startswith("k_heads"):
            fw.append(cur); cur = []
    Ls = list(range(lo, hi + 1, step))
    fw = fw[-len(Ls) * REPS:]
    for i, L in enumerate(Ls):
        f = fw[i * REPS + REPS - 1]
        span = (f[-1][3] - f[0][2]) / 1e3
        print(f"rows {L:5d} span {span:7.1f} | " + " | ".join(f"{n[:24]} {t:6.1f}" for n, t, _, _ in f))
