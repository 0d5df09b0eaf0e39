"""Free-running self-play ("selfplay_async") against the lock-step driver on the bench workload: games/s, forwards per call, executed rows
per forward, for several (launches per batch, stages per launch) settings.  E=episodes (default 32768).
python tools/async_probe.py [launches:iters ...]"""
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
G, E = 8192, int(os.environ.get("E", 32768))
e = azeng.Engine(device=0, max_batch=G)
e.net_init_random(0, 1)
sets = [(0, 0, 0)] + [(1,) + tuple(int(x) for x in a.split(":")) for a in (sys.argv[1:] or ["1:6", "2:6", "3:6", "2:3", "2:12"])]
first = 0
for rnd in range(int(os.environ.get("ROUNDS", 2))):
    for mode, L, I in sets:
        e.set_option("selfplay_async", mode)
        if mode:
            e.set_option("selfplay_async_launches", L)
            e.set_option("selfplay_async_iters", I)
        e.reset_stats()
        t = time.time()
        e.selfplay(n_games=E, concurrent=G, num_sims=100, model_id=0, seed=1, first_game_id=first, want_boards=False)
        dt = time.time() - t
        st = e.stats()
        fw = st["tree_launches"] / max(1, L if mode else 1)
        print(f"round {rnd} async={mode} launches={L} iters={I}: {E / dt:7.1f} games/s | forwards {fw:9.0f} | executed rows/forward {st['leaf_rows_executed'] / max(1, fw):7.0f} | "
              f"executed/requested {st['leaf_rows_executed'] / max(1, st['leaf_rows_requested']):.3f} | us per forward {dt / max(1, fw) * 1e6:7.1f}", flush=True)
        first += E
