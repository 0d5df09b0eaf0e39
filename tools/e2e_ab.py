"""End-to-end A/B of az_set_option switches on the bench workload (8192 slots, 16384 episodes, 100 sims): games/s per setting,
interleaved rounds.  usage: [NET=stub] python tools/e2e_ab.py gemm_variant 5 6"""
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
key, vals = sys.argv[1], [int(x) for x in sys.argv[2:]]
G, E = 8192, 16384
e = azeng.Engine(device=0, max_batch=G)
if os.environ.get('NET', 'conv') == 'stub':
    e.net_set_kind(0, azeng.NET_STUB)
else:
    e.net_init_random(0, 1)
res = {v: [] for v in vals}
first = 0
for rnd in range(3):
    for v in vals:
        e.set_option(key, v)
        t = time.time()
        e.selfplay(n_games=E, concurrent=G, num_sims=100, model_id=0, seed=1, first_game_id=first, want_boards=False)
        res[v].append(E / (time.time() - t))
        first += E
for v in vals:
    print(f"{key}={v}: games/s {np.round(res[v], 1).tolist()} median {np.median(res[v][1:]):.1f}")
