"""conv3 stage and whole-forward time as a function of the executed row count (the staircase of workgroup rounds):
python tools/rows_sweep.py [lo hi step]     (OPT=key=val,key=val sets engine options first)"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
lo, hi, step = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (256, 8192, 256)))
e = azeng.Engine(device=0, max_batch=8192, profile=True, diag=True)
e.net_init_random(0, 1)
for kv in filter(None, os.environ.get("OPT", "").split(",")):
    k, v = kv.split("=")
    e.set_option(k, int(v))
uniq = random_states(8192, 3)
for L in range(lo, hi + 1, step):
    states = uniq[:L]
    e.predict_states(states, 0)
    c3, tot = [], []
    for r in range(5):
        e.reset_stats()
        for _ in range(4):
            e.predict_states(states, 0)
        st = e.stats()
        c3.append(st['net_conv3_ms'] / st['net_launches'])
        tot.append(st['net_total_ms'] / st['net_launches'])
    c, t = np.median(c3) * 1e3, np.median(tot) * 1e3
    print(f"rows {L:5d}  conv3 {c:7.1f} us  {L * 94.371840 / c / 1e3:6.0f} TF   rest {t - c:7.1f} us   forward {t:7.1f} us  {t / L * 1e3:6.1f} ns/row", flush=True)
