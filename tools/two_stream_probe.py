"""Proxy for two search pipelines sharing the GPU: K engines (one stream each), K host threads, 8192/K slots and E/K
episodes each, against one engine with all of them.  python tools/two_stream_probe.py [K ...]"""
import sys, os, time, threading, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
G, E = 8192, int(os.environ.get("E", 32768))
for K in [int(x) for x in (sys.argv[1:] or ["1", "2", "1", "2", "4"])]:
    engs = []
    for k in range(K):
        e = azeng.Engine(device=0, max_batch=G // K)
        e.net_init_random(0, 1)
        if K > 1: e.set_option("search_graph", 0)      # capture forbids the other thread's blocking copies (az_engine.h)
        engs.append(e)
    def run(k, n):
        engs[k].selfplay(n_games=n, concurrent=G // K, num_sims=100, model_id=0, seed=1, first_game_id=k * 1000000, want_boards=False)
    for k in range(K): run(k, 256)
    for e in engs: e.reset_stats()
    t = time.time()
    ths = [threading.Thread(target=run, args=(k, E // K)) for k in range(K)]
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.time() - t
    st = [e.stats() for e in engs]
    ex = sum(x["leaf_rows_executed"] for x in st); rq = sum(x["leaf_rows_requested"] for x in st)
    print(f"K={K}: {E / dt:.1f} games/s  executed rows/s {ex / dt / 1e6:.3f} M  executed/requested {ex / max(rq, 1):.3f}", flush=True)
    for e in engs: e.close()
