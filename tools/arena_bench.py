"""BASELINE config 3: arena head-to-head, 4096 paired games, 400 sims/move, two independently seeded bf16 nets."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 400
e = azeng.Engine(device=0, max_batch=games, profile=bool(os.environ.get("PROFILE")))
for kv in [x for x in os.environ.get("OPT", "").split(",") if x]:
    e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
e.net_init_random(0, 1)
e.net_init_random(1, 2)
e.arena(64, 25, new_model_id=1, old_model_id=0)        # warm-up
e.reset_stats()
t = time.perf_counter()
wld, res = e.arena(games, sims, new_model_id=1, old_model_id=0, seed=3)
dt = time.perf_counter() - t
st = e.stats()
print(json.dumps({"config": f"arena {games} paired games, {sims} sims/move, bf16 C=512 nets (seeds 1 vs 2), per-game tree pair",
                  "seconds": dt, "games_per_sec": games / dt, "wld_new": wld.tolist(), "simulations_per_sec": st["simulations"] / dt,
                  "node_expansions_per_sec": st["expansions"] / dt, "leaf_evals_per_sec": st["leaf_evals"] / dt,
                  "conv3_tflops": st["net_conv3_flops"] / max(st["net_conv3_ms"], 1e-9) / 1e9,
                  "mfma_fraction_end_to_end": st["leaf_evals"] / dt * 328986624 / 2.5e15}))
