"""Per-layer device time of NNet::predict at given row counts, from the engine's HIP-event brackets (conv2 stage / conv3 / conv4 /
fc1 + fc2 + heads / whole forward), for option sets given as OPT="k=v,k=v;k=v" (sets separated by ';').
python tools/layer_times.py 2300 2900 8192"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_rs_amd import engine as azeng
from tools._states import random_states
rows = [int(x) for x in sys.argv[1:]] or [2300]
e = azeng.Engine(device=0, max_batch=8192, profile=True, diag=bool(os.environ.get("DIAG")))
e.net_init_random(0, seed=1)
st = random_states(8192, seed=3)
sets = [x for x in os.environ.get("OPT", "").split(";")] or [""]
for n in rows:
    for opt in sets:
        for kv in [x for x in opt.split(",") if x]:
            k, v = kv.split("=")
            e.set_option(k, int(v))
        for _ in range(3):
            e.predict_states(st[:n], 0)
        e.reset_stats()
        reps = 20
        for _ in range(reps):
            e.predict_states(st[:n], 0)
        s = e.stats()
        L = max(1, s["net_launches"])
        print("rows %5d  %-40s conv2 %6.1f  conv3 %6.1f (%.0f TF)  conv4 %6.1f (%.0f TF)  fc+heads %6.1f  total %6.1f us" % (
            n, opt or "(default)", s["net_conv2_ms"] / L * 1e3, s["net_conv3_ms"] / L * 1e3, s["net_conv3_flops"] / max(1e-9, s["net_conv3_ms"]) / 1e9,
            s["net_conv4_ms"] / L * 1e3, s["net_conv4_flops"] / max(1e-9, s["net_conv4_ms"]) / 1e9, s["net_fc_ms"] / L * 1e3, s["net_total_ms"] / L * 1e3), flush=True)
