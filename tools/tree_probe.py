"""Cycles per phase of k_backup_select (diagnostic build with s_memtime stamps; each stamp drains the wave's memory operations
first, so the phases do not overlap as they do in the shipped kernel): 10th / 50th / 90th percentile over the waves of the last
launch of a short self-play run.  python tools/tree_probe.py"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
e = azeng.Engine(device=0, max_batch=8192, diag=True)
e.net_init_random(0, 1)
e.set_option("tree_stamps", 1)
e.selfplay(n_games=8192 * 2, concurrent=8192, num_sims=100, model_id=0, seed=1, want_boards=False)
e.set_option("print_tree_stamps", 0)
names = ["load head+path", "backup", "wait for its stores", "select", "leaf request", "store head+path", "whole kernel"]
for n, v in zip(names, azeng._lib.az_last_error(e._h).decode().split()):
    print(f"{n:22s} p10/p50/p90 cycles {v}")
