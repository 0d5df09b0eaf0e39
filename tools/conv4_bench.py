"""Forward time vs leaf count for conv4 on the 128x128 register-staged kernel (conv4_big 0) or the 256x256 LDS-DMA
kernel (conv4_big 1): tile quantisation of the small layers at ragged batch sizes."""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
e = azeng.Engine(device=0, max_batch=8192, profile=True)
e.net_init_random(0, 1)
uniq = random_states(512, 3)
for L in (4096, 5000, 5600, 6200, 6553, 6700, 7000, 7400, 8192):
    states = uniq[np.random.default_rng(0).integers(0, 512, L)]
    row = []
    for big in (0, 1):
        e.set_option("conv4_big", big)
        e.predict_states(states, 0)
        ts = []
        for r in range(5):
            e.reset_stats()
            for _ in range(4):
                e.predict_states(states, 0)
            st = e.stats()
            ts.append(st['net_total_ms'] / st['net_launches'])
        row.append(np.median(ts))
    print(f"leaves {L}: forward ms conv4_small {row[0]:.3f} conv4_big {row[1]:.3f}  us/leaf {1e3*min(row)/L:.4f}")
