"""Per-layer kernel times of the net forward for a few batch sizes and option settings (rocprof-free: uses
az profile events for conv2/total, so only those two; per-layer needs rocprofv3 on this script)."""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
e = azeng.Engine(device=0, max_batch=8192, profile=True, diag=True)
e.net_init_random(0, 1)
uniq = random_states(512, 3)
for B in (8192, 5000, 2500):
    states = uniq[np.random.default_rng(0).integers(0, 512, B)]
    for opt in (0, 1):
        e.set_option("conv4_big", opt)
        ts = []
        for r in range(4):
            e.reset_stats()
            for _ in range(4):
                e.predict_states(states, 0)
            st = e.stats()
            ts.append(st['net_total_ms'] / st['net_launches'])
        print(f"B={B} conv4_big={opt}: forward ms median {np.median(ts):.4f}")
