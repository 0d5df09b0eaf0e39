"""Diagnostic: one game of the bench-size arena (two conv nets, record_evals) replayed on the oracle, and the same game played
alone (a one-game shard): where do the recorded rows differ?   python tools/arena_replay_probe.py [game] [num] [sims]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alphazero_rs_amd import engine as azeng
from oracle import oracle_py as orc
g, num, sims = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 446), (2, 4096), (3, 400)))
e = azeng.Engine(device=0, max_batch=8192)
e.net_init_random(22, seed=5); e.net_init_random(23, seed=6)
cap = 22 * (sims + 1) + 8
for k, v in [x.split("=") for x in os.environ.get("OPT", "").split(",") if x]:
    e.set_option(k, int(v))
wld, res = e.arena(num, sims, new_model_id=23, old_model_id=22, seed=9, record_evals=cap)
big = [e.arena_get_evals(w, num, cap) for w in (0, 1)]
w1, r1 = e.arena(1, sims, new_model_id=23, old_model_id=22, seed=9, record_evals=cap, first_game=g, total_games=num)
one = [e.arena_get_evals(w, 1, cap) for w in (0, 1)]
print("result big", res[g], "alone", r1[0])
for w in (0, 1):
    cb, co = int(big[w][0][g]), int(one[w][0][0])
    sb, so = big[w][1][g, :cb], one[w][1][0, :co]
    n = min(cb, co)
    d = np.nonzero((sb[:n] != so[:n]).any(axis=1))[0]
    dp = np.nonzero((big[w][2][g, :n] != one[w][2][0, :n]).any(axis=1) | (big[w][3][g, :n] != one[w][3][0, :n]))[0]
    print("tree", w, "records big", cb, "alone", co, "first differing state", d[:3], "first differing (pi, v)", dp[:3])
    if len(dp):
        i = int(dp[0])
        print("   state", sb[i], "pi big", big[w][2][g, i], "alone", one[w][2][0, i], "v", big[w][3][g, i], one[w][3][0, i])
        p2, v2 = e.predict_states(sb[i:i + 1], 23 if w == 0 else 22)
        print("   predict_states now:", p2[0], v2[0])
for name, logs, idx in (("big", big, g), ("alone", one, 0)):
    rn, ro = [(np.array([0, int(l[0][idx])], np.int64), np.ascontiguousarray(l[1][idx, :l[0][idx]]), np.ascontiguousarray(l[2][idx, :l[0][idx]]),
               np.ascontiguousarray(l[3][idx, :l[0][idx]])) for l in logs]
    owld, ores, bad = orc.arena_ex(num, sims, first_game=g, n_games=1, net_kind=orc.NET_REPLAY, seed=9, replay_new=rn, replay_old=ro)
    print(name, "oracle replay: result", ores[0], "bad bits", bad[0] & 255, "first bad record", bad[0] >> 8)
    if bad[0]:
        req = np.zeros(2, np.uint64)
        orc.lib().azo_debug_last_bad_request(req.ctypes.data_as(__import__("ctypes").c_void_p))
        k = int(bad[0] >> 8)
        w = 0 if (bad[0] & 1) else 1
        l = logs[w]
        print("   oracle requested", [hex(int(x)) for x in req], "stones", bin(int(req[0] | req[1])).count("1"))
        for j in range(max(0, k - 2), min(int(l[0][idx]), k + 3)):
            st = l[1][idx, j]
            print("   record", j, [hex(int(x)) for x in st], "stones", bin(int(st[0] | st[1])).count("1"), "v", l[3][idx, j])
        # is the requested state somewhere else in the log?
        hit = np.nonzero((l[1][idx, :l[0][idx]] == req).all(axis=1))[0]
        print("   requested state is record(s)", hit[:5])
