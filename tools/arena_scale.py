"""Arena wall time vs the number of paired games (400 sims/move): a time that does not fall with the games is launch / latency, not work.
python tools/arena_scale.py [games ...]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
e = azeng.Engine(device=0, max_batch=4096)
e.net_init_random(0, 1); e.net_init_random(1, 2)
e.arena(64, 25, new_model_id=1, old_model_id=0)
for g in [int(x) for x in (sys.argv[1:] or ["64", "512", "2048", "4096"])]:
    e.reset_stats()
    t = time.perf_counter()
    e.arena(g, 400, new_model_id=1, old_model_id=0, seed=3)
    dt = time.perf_counter() - t
    st = e.stats()
    print(f"games {g:5d}: {dt:.3f} s, {g / dt:7.1f} games/s, {st['tree_launches']} tree launches, {dt / max(1, st['tree_launches']) * 1e6:.1f} us wall per tree launch", flush=True)
