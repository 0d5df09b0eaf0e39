#!/usr/bin/env python3
"""Generates tests/golden/*.json from the CPU oracle (oracle/).

The reference holds NO fixture for best_child / search / get_action_prob / execute_episode / play_games
(parity unpinned by the reference) and cannot be run here (no Rust toolchain, and it panics as
written), so these vectors come from the build's own oracle, frozen here so that (a) a change of the oracle's
behaviour is caught on CPU and (b) the GPU engine is checked against committed numbers, not only against a
binary built on the day.  Floats are stored as their IEEE-754 bit patterns (uint32) to stay bit-exact.

Usage: python tests/golden/make_golden.py   (rewrites the JSON files next to this script)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_py as orc  # noqa: E402


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32).tolist()


def episode_trace(sims, net_kind, salt, seed, game_id, temp_threshold=15):
    """Per-move root statistics of one self-play episode (Coach::execute_episode driven move by move)."""
    t = orc.Tree(sims, net_kind=net_kind, salt=salt)
    s, ply, moves = (0, 0), 0, []
    while True:
        temp = 1.0 if ply + 1 < temp_threshold else 0.0
        pi, counts, q = t.get_action_prob(s[0], s[1], temp, seed=seed, game_id=game_id)
        a = orc.lib().azo_rng_choose_weighted(orc.lib().azo_rng_draw(seed, game_id, ply, 2), pi.ctypes.data, 7)
        moves.append({"state": [str(s[0]), str(s[1])], "temp": temp, "counts": counts.tolist(), "pi": bits(pi), "q": bits(q),
                      "action": int(a)})
        s = orc.c4_play(s[0], s[1], a)
        ply += 1
        if orc.c4_ended(*s) != 0.0:
            return {"moves": moves, "result": bits([orc.c4_ended(*s)])[0], "stats": t.stats()}


def main():
    out = {}
    # config 1 of BASELINE.json: 1 game, 25 sims/move, stub net, cpuct 1, temp_threshold 15
    out["config1_stub_25sims"] = {"sims": 25, "net": "stub", "salt": 0, "seed": 0, "game_id": 0,
                                  "trace": episode_trace(25, orc.NET_STUB, 0, 0, 0)}
    out["hash_100sims"] = [{"sims": 100, "net": "hash", "salt": 4242, "seed": 7, "game_id": g,
                            "trace": episode_trace(100, orc.NET_HASH, 4242, 7, g)} for g in range(3)]
    out["hash_400sims"] = {"sims": 400, "net": "hash", "salt": 99, "seed": 1, "game_id": 5,
                           "trace": episode_trace(400, orc.NET_HASH, 99, 1, 5)}
    with open(os.path.join(HERE, "search_traces.json"), "w") as f:
        json.dump(out, f)

    # self-play tuples (s, pi, z): 6 episodes x 50 sims, hash net
    r = orc.selfplay(6, 50, net_kind=orc.NET_HASH, salt=31337, seed=11, first_game_id=40)
    sp = {"n_games": 6, "sims": 50, "salt": 31337, "seed": 11, "first_game_id": 40, "count": r["count"],
          "game_len": r["game_len"].tolist(), "moves": r["moves"].tolist(),
          "boards_packed": np.packbits(r["boards"].astype(np.uint8).reshape(-1)).tolist(),
          "pis": bits(r["pis"].reshape(-1)), "zs": bits(r["zs"]), "stats": r["stats"]}
    with open(os.path.join(HERE, "selfplay_tuples.json"), "w") as f:
        json.dump(sp, f)

    # arena: 12 games, 50 sims, two hash nets (model ids 1 = new, 0 = old)
    wld, results = orc.arena(12, 50, net_kind=orc.NET_HASH, salt=2024, seed=3, new_model_id=1, old_model_id=0)
    with open(os.path.join(HERE, "arena.json"), "w") as f:
        json.dump({"num": 12, "sims": 50, "salt": 2024, "seed": 3, "new_model_id": 1, "old_model_id": 0,
                   "wld": wld.tolist(), "results": results.tolist()}, f)

    # packed-counter and PUCT known answers at the production scale (WIN_SCALE = 100)
    L = orc.lib()
    c = L.azo_ctr_init()
    seq = []
    for v in (1.0, -1.0, 0.0, 0.5, -0.37, 1e-4, -1e-4, 0.999, -0.004):
        c = L.azo_ctr_visit(c)
        c = L.azo_ctr_unvisit(c, v, 100.0)
        seq.append({"v": bits([v])[0], "ctr": str(c), "w": bits([L.azo_ctr_w(c, 100.0)])[0], "n": L.azo_ctr_n(c),
                    "q": bits([L.azo_ctr_q(c, 100.0)])[0]})
    puct = [{"ctr": str(c), "prior": bits([p])[0], "parent_n": pn, "cpuct": cp,
             "u": bits([L.azo_puct(c, p, pn, cp)])[0]} for p, pn, cp in ((0.142857, 9, 1), (0.5, 100, 1), (0.01, 16800, 2))]
    with open(os.path.join(HERE, "counter_puct.json"), "w") as f:
        json.dump({"sequence": seq, "puct": puct}, f)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
