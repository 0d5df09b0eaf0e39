"""Tree-only rate (stub net: pi = 1/7, v = +1) versus the number of concurrent trees: games/s, sims/s and the
algorithmic tree bytes per second of the select+backup kernels (128 B per level + 356 B per sim)."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
e = azeng.Engine(device=0, max_batch=8192, profile=True)
e.net_set_kind(0, azeng.NET_STUB)
e.selfplay(n_games=256, num_sims=25, model_id=0, want_boards=False, want_states=False)
for G in (2048, 8192, 32768, 65536):
    e.reset_stats()
    t = time.perf_counter()
    r = e.selfplay(n_games=G, num_sims=100, model_id=0, seed=1, want_boards=False, want_states=False, symmetries=False)
    dt = time.perf_counter() - t
    st = e.stats()
    print(json.dumps({"trees": G, "seconds": round(dt, 3), "games_per_sec": round(G / dt, 1), "sims_per_sec": round(st["simulations"] / dt),
                      "expansions_per_sec": round(st["expansions"] / dt), "mean_depth": round(st["depth_sum"] / st["simulations"], 2),
                      "tree_kernel_ms": round(st["tree_ms"], 1), "tree_GBps_algorithmic": round(st["tree_bytes"] / st["tree_ms"] / 1e6, 1),
                      "hbm_frac": round(st["tree_bytes"] / st["tree_ms"] / 1e6 / 8000, 4)}), flush=True)
