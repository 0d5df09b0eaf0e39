"""Per-segment cycle sums of conv3's K loop (diagnostic build conv3_pipe = 3; wave 0, median over the first 128 workgroups):
python tools/seg_probe.py rows [rows ...]"""
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
e = azeng.Engine(device=0, max_batch=8192, profile=True, diag=True)
e.net_init_random(0, 1)
uniq = random_states(8192, 3)
names = ["reads+MMA1", "wait+bar1", "dmaW", "rd+MMA2", "rd+MMA3", "wait+bar2", "imgswitch", "rd+MMA4", "total", "ticks100MHz"]
for L in [int(x) for x in sys.argv[1:]]:
    e.set_option("conv3_pipe", 3)
    t = time.time()
    while time.time() - t < 1.0:
        e.predict_states(uniq[:L], 0)
    e.set_option("print_seg_stamps", 0)
    v = [int(x) for x in azeng._lib.az_last_error(e._h).decode().split()]
    tot = v[8]
    print(f"rows {L}: clock {100.0 * v[8] / max(1, v[9]):.0f} MHz | " + " | ".join(f"{n} {x / 72:.0f}" for n, x in zip(names[:8], v[:8])) + f" | per step {tot / 72:.0f} cycles", flush=True)
