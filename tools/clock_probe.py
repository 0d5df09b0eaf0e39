"""In-kernel clock of the conv2 K loop (diagnostic kernel variant 13): back-to-back launches for ~2 s, then read
the s_memtime / s_memrealtime stamps."""
import sys, os, time, numpy as np, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
e = azeng.Engine(device=0, max_batch=8192, profile=True, diag=True)
e.net_init_random(0, 1)
uniq = random_states(512, 3)
states = uniq[np.random.default_rng(0).integers(0, 512, 8192)]
for variant in (13,):
    e.set_option("gemm_variant", variant)
    t = time.time()
    n = 0
    while time.time() - t < 2.5:
        e.predict_states(states, 0); n += 1
    e.reset_stats()
    for _ in range(8):
        e.predict_states(states, 0)
    st = e.stats()
    e.set_option("print_clock_stamps", 0)
    print("variant", variant, "conv2 TFLOP/s", st['net_conv2_flops'] / st['net_conv2_ms'] / 1e9, "|", azeng._lib.az_last_error(e._h).decode())
