"""Arena (BASELINE config 3) under several option sets, interleaved in ONE process: games/s per set and round.
OPTS="a=1,b=2;c=3" python tools/arena_ab.py [rounds=3] [games=4096] [sims=400]   (every set starts from the defaults of the keys it names)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_rs_amd import engine as azeng
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
games = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 400
DEFAULTS = {"search_graph": 20, "narrow_rows": 32, "tree_block4": 1, "conv3_small": 1, "conv3_planes": 1, "conv3_tail": 1}
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",") if kv) for s in os.environ.get("OPTS", "").split(";")]
e = azeng.Engine(device=0, max_batch=games)
e.net_init_random(0, 1)
e.net_init_random(1, 2)
e.arena(256, 100, new_model_id=1, old_model_id=0)        # warm-up
ref = None
for r in range(rounds):
    for st in sets:
        for k, v in DEFAULTS.items():
            e.set_option(k, st.get(k, v))
        t = time.perf_counter()
        wld, res = e.arena(games, sims, new_model_id=1, old_model_id=0, seed=3)
        dt = time.perf_counter() - t
        if ref is None:
            ref = res.copy()
        print(f"round {r} {st or 'defaults'}: {games / dt:7.1f} games/s  identical results: {bool((res == ref).all())}", flush=True)
