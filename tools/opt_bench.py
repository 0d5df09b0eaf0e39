"""conv2 and whole-forward time vs leaf count for an az_set_option switch, with bitwise agreement:
python tools/opt_bench.py conv1_table 0 1      (LEAVES=2048,4096 to choose the batch sizes)"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
key, vals = sys.argv[1], [int(x) for x in sys.argv[2:]]
e = azeng.Engine(device=0, max_batch=8192, profile=True, diag=True)
e.net_init_random(0, 1)
uniq = random_states(int(os.environ.get("UNIQ", 8192)), 3)
for L in [int(x) for x in os.environ.get("LEAVES", "1400,2048,2700,4096,5000,6700,8192").split(",")]:
    states = uniq[np.random.default_rng(0).integers(0, len(uniq), L)]
    row, outs = [], []
    for v in vals:
        e.set_option(key, v)
        outs.append(e.predict_states(states, 0))
        ts, cs = [], []
        for r in range(5):
            e.reset_stats()
            for _ in range(4):
                e.predict_states(states, 0)
            st = e.stats()
            ts.append(st['net_total_ms'] / st['net_launches'])
            cs.append(st['net_conv2_ms'] / st['net_launches'])
        row.append((np.median(cs), np.median(ts)))
    same = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs[1:])
    print(f"leaves {L}: " + " | ".join(f"{key}={v}: conv2 {c:.3f} ms ({L * 198.180864 / c / 1e3:.0f} TF) forward {t:.3f} ms ({L * 328.986624 / t / 1e3:.0f} TF)"
                                        for v, (c, t) in zip(vals, row)) + f"  bitwise {same}", flush=True)
