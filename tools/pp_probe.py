"""Per-segment cycle sums of k_conv3_pp's stage (diagnostic build conv3_pp = 36 / 37: s_memtime stamps of waves 0 and 4, median over the
first 128 workgroups): LOAD until its reads are back | barrier 1 | MFMA issue | wait for the DMA | barrier 2, per stage.
python tools/pp_probe.py rows [rows ...]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
e = azeng.Engine(device=0, max_batch=8192, profile=True, diag=True)
e.net_init_random(0, 1)
uniq = random_states(8192, 3)
names = ["load", "bar1", "mfma", "dmawait", "bar2"]
for L in [int(x) for x in sys.argv[1:]]:
    for var in (36, 37):
        e.set_option("conv3_pp", var)
        t = time.time()
        while time.time() - t < 0.5:
            e.predict_states(uniq[:L], 0)
        e.set_option("print_pp_stamps", 0)
        v = [int(x) for x in azeng._lib.az_last_error(e._h).decode().split()]
        for g in (0, 1):
            w = v[8 * g: 8 * g + 8]
            print(f"rows {L} conv3_pp={var} group {g}: " + " | ".join(f"{n} {x / 144:.0f}" for n, x in zip(names, w[:5])) + f" | per stage {w[7] / 144:.0f} cycles", flush=True)
