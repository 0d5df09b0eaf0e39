"""max |dpi| / |dv| between two values of an az_set_option switch on random states: python tools/diff_opt.py conv3_big 0 1 [rows]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
key, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
e = azeng.Engine(device=0, max_batch=8192, diag=True)
e.net_init_random(0, 1)
st = random_states(n, 3)
e.set_option(key, a); pa, va = e.predict_states(st, 0)
e.set_option(key, b); pb, vb = e.predict_states(st, 0)
pb2, vb2 = e.predict_states(st[::-1].copy(), 0)
print(f"{key} {a} vs {b}: max|dpi| {np.abs(pa - pb).max():.3e} max|dv| {np.abs(va - vb).max():.3e}; {key}={b} row-order independent: "
      f"{np.array_equal(pb2[::-1], pb) and np.array_equal(vb2[::-1], vb)}; finite {np.isfinite(pb).all()}")
