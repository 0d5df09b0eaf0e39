"""Forward time vs leaf count for an az_set_option switch: python tools/opt_bench.py conv4_img 0 1"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
key, vals = sys.argv[1], [int(x) for x in sys.argv[2:]]
e = azeng.Engine(device=0, max_batch=8192, profile=True)
e.net_init_random(0, 1)
uniq = random_states(512, 3)
ref = None
for L in (2048, 4096, 5000, 6000, 6700, 7400, 8192):
    states = uniq[np.random.default_rng(0).integers(0, 512, L)]
    row, outs = [], []
    for v in vals:
        e.set_option(key, v)
        outs.append(e.predict_states(states, 0))
        ts = []
        for r in range(5):
            e.reset_stats()
            for _ in range(4):
                e.predict_states(states, 0)
            st = e.stats()
            ts.append(st['net_total_ms'] / st['net_launches'])
        row.append(np.median(ts))
    same = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs[1:])
    print(f"leaves {L}: forward ms " + " ".join(f"{key}={v}: {t:.3f}" for v, t in zip(vals, row)) + f"  bitwise {same}")
