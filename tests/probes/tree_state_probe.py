"""Diagnostic (libaz_engine_diag.so): tree state of engine vs oracle right before the first simulation whose leaf differs."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from alphazero_rs_amd import engine as azeng
from oracle import oracle_py as orc
g, total, sims, bad_ply, bad_sim = 446, 4096, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 306
e = azeng.Engine(device=0, max_batch=64, diag=True)
L = e._lib
L.az_diag_tree_children.restype = C.c_int
L.az_diag_tree_children.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
L.az_diag_tree_set_sims.restype = None
L.az_diag_tree_set_sims.argtypes = [C.c_void_p, C.c_int]
OL = orc.lib()
OL.azo_tree_debug_children.restype = C.c_int
OL.azo_tree_debug_children.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
OL.azo_tree_set_sims.restype = None
OL.azo_tree_set_sims.argtypes = [C.c_void_p, C.c_uint64]
e.net_init_random(22, seed=5); e.net_init_random(23, seed=6)
first = 0 if g < total // 2 else 1
mids = (23, 22)
trees = [e.tree_create(1, reserve=orc.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=m, cpuct=1) for m in mids]
cur = {"mid": 0}
def predict(boards, model_id):
    b = np.asarray(boards)
    st = np.zeros((b.shape[0], 2), np.uint64)
    for i in range(b.shape[0]):
        for r in range(6):
            for c in range(7):
                bit = 1 << (c * 7 + (5 - r))
                if b[i, 0, r, c] != 0: st[i, 0] |= np.uint64(bit)
                if b[i, 1, r, c] != 0: st[i, 1] |= np.uint64(bit)
    return e.predict_states(st, cur["mid"])
orc.set_predict_callback(predict)
otrees = [orc.Tree(sims, net_kind=orc.NET_CALLBACK) for _ in range(2)]
s, player = (0, 0), 1
def children(slot, path):
    p = np.array(path, np.int32)
    a = np.zeros(64, np.uint64); b = np.zeros(64, np.uint64)
    na = L.az_diag_tree_children(trees[slot]._h, 0, p.ctypes.data_as(C.c_void_p), len(path), a.ctypes.data_as(C.c_void_p))
    nb = OL.azo_tree_debug_children(otrees[slot]._h, s[0], s[1], p.ctypes.data_as(C.c_void_p), len(path), b.ctypes.data_as(C.c_void_p))
    return na, a.reshape(8, 8), nb, b.reshape(8, 8)
def u_of(ctr, prior_bits, parent_n):
    w = np.float32(np.int64(ctr >> np.uint64(32)) - 0x7FFFFFFF) / np.float32(100)
    n = int((ctr >> np.uint64(16)) & np.uint64(0xFFFF)); vl = int(ctr & np.uint64(0xFFFF))
    q = np.float32(0) if n == 0 else (w - np.float32(vl)) / np.float32(n)
    p = np.array([prior_bits], np.uint32).view(np.float32)[0]
    return q + (np.float32(1) * p) * np.sqrt(np.float32(parent_n) + np.float32(1e-6)) / np.float32((n + 1) & 0xFFFF), n, w, p
def walk(slot, path, depth_left):
    na, a, nb, b = children(slot, path)
    pn_a = int((a[7, 0] >> np.uint64(16)) & np.uint64(0xFFFF)); pn_b = int((b[7, 0] >> np.uint64(16)) & np.uint64(0xFFFF))
    print("node", path, "engine N", pn_a, "oracle N", pn_b, "children", na, nb)
    for j in range(max(na, nb)):
        ua = u_of(a[j, 2], int(a[j, 3]), pn_a + 1); ub = u_of(b[j, 2], int(b[j, 3]), pn_b + 1)
        flag = "" if (a[j, 2] == b[j, 2] and a[j, 3] == b[j, 3]) else "   <-- DIFFERENT"
        print("   child", j, "a", int(a[j, 1]), int(b[j, 1]), "engine ctr %016x prior %08x link %x u %.9g | oracle ctr %016x prior %08x link %d u %.9g%s" %
              (int(a[j, 2]), int(a[j, 3]), int(a[j, 4]), ua[0], int(b[j, 2]), int(b[j, 3]), -1 if b[j, 4] == np.uint64(2**64 - 1) else 1, ub[0], flag))
    if depth_left > 0:
        for j in range(na):
            walk(slot, path + [j], depth_left - 1)
for ply in range(42):
    slot = first if player == 1 else 1 - first
    cur["mid"] = mids[slot]
    n = bad_sim - 1 if ply == bad_ply else sims          # the differing call stops right before the differing simulation
    L.az_diag_tree_set_sims(trees[slot]._h, n)
    OL.azo_tree_set_sims(otrees[slot]._h, n)
    pi, counts, q = trees[slot].get_action_prob(np.array([s], dtype=np.uint64), 0.0, seed=9, first_game_id=g)
    opi, ocnt, oq = otrees[slot].get_action_prob(s[0], s[1], 0.0, seed=9, game_id=g)
    print("ply", ply, "counts", counts[0].tolist(), ocnt.tolist())
    if ply == bad_ply:
        walk(slot, [], 2)
        break
    s = orc.c4_play(s[0], s[1], int(np.argmax(pi[0])))
    player = -player

# ---- follow the arithmetic's own best path from the root of the differing call ----
def unkey(key):
    m = mask = 0
    for c in range(7):
        col = (key >> (c * 7)) & 0x7F
        top = 1 << (col.bit_length() - 1) if col else 1
        m |= (col ^ top) << (c * 7)
        mask |= (top - 1) << (c * 7)
    return m, mask ^ m
def show(st):
    m, t = st
    return " | ".join("".join("X" if (m >> (c * 7 + r)) & 1 else "O" if (t >> (c * 7 + r)) & 1 else "." for c in range(7)) for r in range(5, -1, -1))
slot = first if player == 1 else 1 - first
path, state = [], s
for level in range(12):
    p = np.array(path, np.int32)
    a = np.zeros(64, np.uint64); b = np.zeros(64, np.uint64)
    if state == s:
        na = L.az_diag_tree_children(trees[slot]._h, 0, p.ctypes.data_as(C.c_void_p), len(path), a.ctypes.data_as(C.c_void_p))
    nb = OL.azo_tree_debug_children(otrees[slot]._h, state[0], state[1], None, 0, b.ctypes.data_as(C.c_void_p))
    b = b.reshape(8, 8)
    pn = int((b[7, 0] >> np.uint64(16)) & np.uint64(0xFFFF))
    us = [u_of(b[j, 2], int(b[j, 3]), pn + 1)[0] for j in range(nb)]
    best = 0
    for j in range(1, nb):
        if not (us[best] > us[j]): best = j
    act = int(b[best, 1])
    nxt = orc.c4_play(state[0], state[1], act)
    kind = "link" if b[best, 4] != np.uint64(2**64 - 1) else ("expanded" if b[best, 5] else "placeholder")
    known = OL.azo_tree_debug_children(otrees[slot]._h, nxt[0], nxt[1], None, 0, np.zeros(64, np.uint64).ctypes.data_as(C.c_void_p))
    print("level", level, "node N", pn, "u", ["%.7g" % x for x in us], "best child", best, "action", act, kind, "| successor in seen:", known >= 0, "|", show(nxt))
    path.append(best)
    state = nxt
    if kind == "placeholder" and known < 0:
        print("  -> expands here")
        break
