"""Experiment: ONE GPU driven by K engines on K host threads, each a self-play call of slots / K concurrent games (the calls' tree
kernels, small tail layers and host round trips overlap each other's big GEMMs) against one engine with all the slots.
python tools/two_engines.py [K=2] [slots=8192] [episodes=32768] [sims=100]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_rs_amd import engine as azeng
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
episodes = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
sims = int(sys.argv[4]) if len(sys.argv) > 4 else 100


def run(k_engines):
    es = [azeng.Engine(device=0, max_batch=slots // k_engines) for _ in range(k_engines)]
    for e in es:
        e.net_init_random(0, seed=1)
    res = [None] * k_engines

    def work(i, n, warm):
        res[i] = es[i].selfplay(n_games=n, concurrent=slots // k_engines, num_sims=sims, model_id=0, seed=7, first_game_id=i * (episodes // k_engines),
                                symmetries=False, want_boards=False, want_states=False)
    for warm in (True, False):
        n = (slots if warm else episodes) // k_engines
        ths = [threading.Thread(target=work, args=(i, n, warm)) for i in range(k_engines)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
    st = [e.stats() for e in es]
    ex = sum(s["leaf_rows_executed"] for s in st); rq = sum(s["leaf_rows_requested"] for s in st)
    print(f"{k_engines} engine(s) x {slots // k_engines} slots: {episodes / dt:8.1f} games/s  ({dt:.2f} s; executed/requested rows {ex / max(1, rq):.3f})", flush=True)
    for e in es:
        e.close()


run(1)
run(K)
