"""A/B the net's GEMM variants on a full leaf batch: per-variant conv2 and whole-forward TFLOP/s, and bitwise /
numeric agreement between variants.  Interleaved rounds in one process (variance-correlated)."""
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
e = azeng.Engine(device=0, max_batch=B, profile=True, diag=True)
e.net_init_random(0, 1)
uniq = random_states(512, 3)
states = uniq[np.random.default_rng(0).integers(0, 512, B)]
VARS = tuple(int(x) for x in os.environ.get('VARS', '0,2,3,5').split(','))
outs = {}
for v in VARS:
    e.set_option("gemm_variant", v)
    outs[v] = e.predict_states(states, 0)
v0 = VARS[0]
for v in VARS[1:]:
    print("variant", v, "vs", v0, ": max|dpi|", np.abs(outs[v0][0] - outs[v][0]).max(), "max|dv|", np.abs(outs[v0][1] - outs[v][1]).max(),
          "bitwise", np.array_equal(outs[v0][0], outs[v][0]) and np.array_equal(outs[v0][1], outs[v][1]))
res = {v: [] for v in VARS}
for r in range(rounds):
    for v in VARS:
        e.set_option("gemm_variant", v)
        e.reset_stats()
        for _ in range(4):
            e.predict_states(states, 0)
        st = e.stats()
        res[v].append((st['net_conv2_flops'] / st['net_conv2_ms'] / 1e9, st['net_total_flops'] / st['net_total_ms'] / 1e9,
                       st['net_conv2_ms'] / st['net_launches'], st['net_total_ms'] / st['net_launches']))
for v in VARS:
    a = np.array(res[v])
    print(f"variant {v}: conv2 TFLOP/s median {np.median(a[:,0]):.1f} max {a[:,0].max():.1f} | forward TFLOP/s median {np.median(a[:,1]):.1f} "
          f"| conv2 ms {np.median(a[:,2]):.3f} forward ms {np.median(a[:,3]):.3f}")
