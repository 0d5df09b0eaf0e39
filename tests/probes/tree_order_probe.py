"""Diagnostic: the order in which one arena game's trees request leaf evaluations, engine log vs an oracle whose net is a callback
into the engine's own NNet::predict.  Prints the first differing request of the first differing call."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alphazero_rs_amd import engine as azeng
from oracle import oracle_py as orc
g, total, sims = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 446), (2, 4096), (3, 400)))
e = azeng.Engine(device=0, max_batch=64)
e.net_init_random(22, seed=5); e.net_init_random(23, seed=6)
cap = 22 * (sims + 1) + 8
first = 0 if g < total // 2 else 1
mids = (23, 22)
trees = [e.tree_create(1, reserve=orc.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=m, cpuct=1) for m in mids]
for t in trees:
    t.record_evals(cap)
cur = {"mid": 0, "req": []}
def predict(boards, model_id):
    b = np.asarray(boards)
    st = np.zeros((b.shape[0], 2), np.uint64)
    for i in range(b.shape[0]):
        for r in range(6):
            for c in range(7):
                bit = 1 << (c * 7 + (5 - r))
                if b[i, 0, r, c] != 0: st[i, 0] |= np.uint64(bit)
                if b[i, 1, r, c] != 0: st[i, 1] |= np.uint64(bit)
    cur["req"].extend([tuple(int(x) for x in s) for s in st])
    return e.predict_states(st, cur["mid"])
orc.set_predict_callback(predict)
otrees = [orc.Tree(sims, net_kind=orc.NET_CALLBACK) for _ in range(2)]
done = [0, 0]
s, player = (0, 0), 1
def show(st):
    m, t = st
    rows = []
    for r in range(5, -1, -1):
        rows.append("".join("X" if (m >> (c * 7 + r)) & 1 else "O" if (t >> (c * 7 + r)) & 1 else "." for c in range(7)))
    return " | ".join(rows)
for ply in range(42):
    slot = first if player == 1 else 1 - first
    pi, counts, q = trees[slot].get_action_prob(np.array([s], dtype=np.uint64), 0.0, seed=9, first_game_id=g)
    cnt, st, ps, vs = trees[slot].get_evals()
    n0, n1 = done[slot], int(cnt[0])
    done[slot] = n1
    cur["mid"], cur["req"] = mids[slot], []
    opi, ocnt, oq = otrees[slot].get_action_prob(s[0], s[1], 0.0, seed=9, game_id=g)
    glog = [tuple(int(x) for x in r) for r in st[0, n0:n1]]
    same = glog == cur["req"]
    print("ply", ply, "slot", slot, "rows", n1 - n0, len(cur["req"]), "counts equal", np.array_equal(counts[0], ocnt), "order equal", same, flush=True)
    if not same:
        k = next(i for i in range(min(len(glog), len(cur["req"]))) if glog[i] != cur["req"][i])
        print(" first differing request", k, "root", show(s))
        print("   engine:", show(glog[k]))
        print("   oracle:", show(cur["req"][k]))
        print("   engine later at", [i for i in range(len(glog)) if glog[i] == cur["req"][k]][:3], " oracle later at", [i for i in range(len(cur["req"])) if cur["req"][i] == glog[k]][:3])
        for j in range(max(0, k - 3), k):
            print("   before:", j, show(glog[j]))
        break
    a = int(np.argmax(pi[0]))
    s = orc.c4_play(s[0], s[1], a)
    player = -player
    if orc.c4_ended(*s) != 0.0:
        break
