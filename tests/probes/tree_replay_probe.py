"""Diagnostic: one arena game driven move by move through az_tree_get_action_prob (two conv nets, one tree per player), every call
replayed on an oracle tree from the rows that call recorded; prints the first call whose visit counts differ.
python tools/tree_replay_probe.py [game] [total] [sims]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alphazero_rs_amd import engine as azeng
from oracle import oracle_py as orc
g, total, sims = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 446), (2, 4096), (3, 400)))
e = azeng.Engine(device=0, max_batch=64)
e.net_init_random(22, seed=5); e.net_init_random(23, seed=6)
for k, v in [x.split("=") for x in os.environ.get("OPT", "").split(",") if x]:
    e.set_option(k, int(v))
cap = 22 * (sims + 1) + 8
first = 0 if g < total // 2 else 1                      # model slot of the first seat: 0 = new (23), 1 = old (22)
trees = [e.tree_create(1, reserve=orc.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=m, cpuct=1) for m in (23, 22)]
for t in trees:
    t.record_evals(cap)
otrees = [orc.Tree(sims, net_kind=orc.NET_REPLAY) for _ in range(2)]
done = [0, 0]
s, player = (0, 0), 1
for ply in range(42):
    slot = first if player == 1 else 1 - first
    pi, counts, q = trees[slot].get_action_prob(np.array([s], dtype=np.uint64), 0.0, seed=9, first_game_id=g)
    cnt, st, ps, vs = trees[slot].get_evals()
    n0, n1 = done[slot], int(cnt[0])
    otrees[slot].set_replay(st[0, n0:n1], ps[0, n0:n1], vs[0, n0:n1])
    done[slot] = n1
    opi, ocnt, oq = otrees[slot].get_action_prob(s[0], s[1], 0.0, seed=9, game_id=g)
    ok = np.array_equal(counts[0], ocnt) and np.array_equal(pi[0], opi) and not otrees[slot].replay_bad()
    print("ply", ply, "slot", slot, "rows", n1 - n0, "gpu counts", counts[0].tolist(), "oracle", ocnt.tolist(), "OK" if ok else "DIFF", flush=True)
    if not ok:
        print("   q gpu", q[0], "oracle", oq, "replay_bad", otrees[slot].replay_bad())
        break
    a = int(np.argmax(pi[0]))
    s = orc.c4_play(s[0], s[1], a)
    player = -player
    if orc.c4_ended(*s) != 0.0:
        print("game over after", ply + 1, "plies")
        break
