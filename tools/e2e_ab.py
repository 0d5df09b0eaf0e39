"""End-to-end A/B of az_set_option switches on the bench workload (8192 slots, E episodes, 100 sims): games/s per setting,
interleaved rounds, with the leaf-row accounting (requested / executed / cache hits / in-batch duplicates).
usage: [NET=stub] [E=32768] [ROUNDS=3] python tools/e2e_ab.py key v0 v1 ... [-- key2=value2 ...fixed options]"""
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
argv = sys.argv[1:]
fixed = []
if "--" in argv:
    i = argv.index("--")
    fixed = [a.split("=") for a in argv[i + 1:]]
    argv = argv[:i]
key, vals = argv[0], [int(x) for x in argv[1:]]
G, E = 8192, int(os.environ.get("E", 16384))
e = azeng.Engine(device=0, max_batch=G, diag=True)
for k, v in fixed:
    e.set_option(k, int(v))
if os.environ.get('NET', 'conv') == 'stub':
    e.net_set_kind(0, azeng.NET_STUB)
else:
    e.net_init_random(0, 1)
res = {v: [] for v in vals}
acct = {}
first = 0
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    for v in vals:
        e.set_option(key, v)
        e.reset_stats()
        t = time.time()
        e.selfplay(n_games=E, concurrent=G, num_sims=100, model_id=0, seed=1, first_game_id=first, want_boards=False)
        res[v].append(E / (time.time() - t))
        st = e.stats()
        rq = max(1, st["leaf_rows_requested"])
        acct[v] = (st["leaf_rows_executed"] / rq, st["eval_cache_hits"] / rq, st["eval_batch_dups"] / rq, st["eval_cache_inserts"])
        first += E
for v in vals:
    a = acct[v]
    print(f"{key}={v}: games/s {np.round(res[v], 1).tolist()} median {np.median(res[v][1:]):.1f} | executed/requested {a[0]:.3f} "
          f"cache hits {a[1]:.3f} batch dups {a[2]:.3f} inserts {a[3]}", flush=True)
