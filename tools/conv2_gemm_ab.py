"""conv2 as the MFMA implicit GEMM ("conv2_table" = 0): k_conv_img2 (conv2_pipe 0) against k_conv_same_pipe (1): bitwise agreement and
conv2-stage / forward time by row count.  python tools/conv2_gemm_ab.py [rows ...]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
e = azeng.Engine(device=0, max_batch=8192, profile=True, diag=True)
e.net_init_random(0, 1)
e.set_option("conv2_table", 0)
uniq = random_states(8192, 3)
for L in [int(x) for x in (sys.argv[1:] or ["1530", "3072", "6700", "8192"])]:
    st = uniq[:L]
    outs, row = [], []
    for v in (0, 1):
        e.set_option("conv2_pipe", v)
        outs.append(e.predict_states(st, 0))
        cs, ts = [], []
        for r in range(5):
            e.reset_stats()
            for _ in range(4):
                e.predict_states(st, 0)
            s = e.stats()
            cs.append(s['net_conv2_ms'] / s['net_launches']); ts.append(s['net_total_ms'] / s['net_launches'])
        row.append((np.median(cs), np.median(ts)))
    same = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    print(f"rows {L}: " + " | ".join(f"conv2_pipe={v}: conv2 {c * 1e3:.1f} us ({L * 198.180864 / c / 1e3:.0f} TF) forward {t * 1e3:.1f} us" for v, (c, t) in zip((0, 1), row)) + f"  bitwise {same}", flush=True)
