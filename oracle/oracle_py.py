"""ctypes wrapper of the CPU oracle (oracle/libaz_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the engine package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "libaz_oracle.so")
TEST_BIN = os.path.join(_DIR, "test_oracle")

NET_STUB, NET_HASH, NET_REPLAY, NET_CALLBACK = 0, 1, 2, 3
GAME_BITS, GAME_ARRAY, GAME_CONNECT3 = 0, 1, 2     # 2 = the engine's second Game policy (three in a row wins)
QUIRK_B1, QUIRK_B2, QUIRK_B4, QUIRK_B6 = 1, 2, 4, 8


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in ("az_oracle.hpp", "az_oracle_games.hpp", "az_oracle_capi.cpp",
                                            "test_oracle.cpp", "Makefile")]
    stale = force or not (os.path.exists(LIB_PATH) and os.path.exists(TEST_BIN))
    if not stale:
        t = min(os.path.getmtime(LIB_PATH), os.path.getmtime(TEST_BIN))
        stale = any(os.path.getmtime(s) > t for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        u64, u32, i32, f32, vp, i64 = C.c_uint64, C.c_uint32, C.c_int32, C.c_float, C.c_void_p, C.c_int64
        L.azo_ctr_init.restype = u64
        L.azo_ctr_visit.restype = u64; L.azo_ctr_visit.argtypes = [u64]
        L.azo_ctr_unvisit.restype = u64; L.azo_ctr_unvisit.argtypes = [u64, f32, f32]
        L.azo_ctr_w.restype = f32; L.azo_ctr_w.argtypes = [u64, f32]
        L.azo_ctr_n.restype = u32; L.azo_ctr_n.argtypes = [u64]
        L.azo_ctr_vloss.restype = u32; L.azo_ctr_vloss.argtypes = [u64]
        L.azo_ctr_q.restype = f32; L.azo_ctr_q.argtypes = [u64, f32]
        L.azo_puct.restype = f32; L.azo_puct.argtypes = [u64, f32, u32, i32]
        L.azo_rng_draw.restype = u64; L.azo_rng_draw.argtypes = [u64, u64, u64, u64]
        L.azo_rng_choose.restype = u32; L.azo_rng_choose.argtypes = [u64, u32]
        L.azo_rng_choose_weighted.restype = i32; L.azo_rng_choose_weighted.argtypes = [u64, vp, i32]
        L.azo_c4_play.restype = None; L.azo_c4_play.argtypes = [u64, u64, i32, vp]
        L.azo_c4_ended.restype = f32; L.azo_c4_ended.argtypes = [u64, u64]
        L.azo_c4_ended_array.restype = f32; L.azo_c4_ended_array.argtypes = [u64, u64, i32]
        L.azo_c4_valid_mask.restype = i32; L.azo_c4_valid_mask.argtypes = [u64, u64]
        L.azo_c4_features.restype = None; L.azo_c4_features.argtypes = [u64, u64, vp]
        L.azo_c4_features_array.restype = None; L.azo_c4_features_array.argtypes = [u64, u64, vp]
        L.azo_c4_mirror.restype = u64; L.azo_c4_mirror.argtypes = [u64]
        L.azo_hashnet.restype = None; L.azo_hashnet.argtypes = [u64, u64, u64, vp, vp]
        L.azo_tree_new.restype = vp
        L.azo_tree_new.argtypes = [i32, i32, u64, u64, u64, u64, u64, u64, i32, i32, u64, u32]
        L.azo_tree_new_mt.restype = vp
        L.azo_tree_new_mt.argtypes = [i32, i32, u64, u64, u64, u64, u64, u64, u64, i32, i32, u64, u32, i32]
        L.azo_tree_set_replay.restype = None; L.azo_tree_set_replay.argtypes = [vp, vp, vp, vp, u64]
        L.azo_tree_replay_bad.restype = i32; L.azo_tree_replay_bad.argtypes = [vp]
        L.azo_tree_free.restype = None; L.azo_tree_free.argtypes = [vp]
        L.azo_tree_get_action_prob.restype = i32
        L.azo_tree_get_action_prob.argtypes = [vp, u64, u64, f32, u64, u64, vp, vp, vp]
        L.azo_tree_stats.restype = None; L.azo_tree_stats.argtypes = [vp, vp]
        L.azo_selfplay.restype = i64
        L.azo_selfplay.argtypes = [i64, u64, u64, u64, i32, u64, u64, u64, i32, u64, i32, u32, i32, vp, vp, vp, i64,
                                   vp, vp, vp, vp, vp, vp, vp, vp, i32]
        L.azo_arena_ex.restype = i32
        L.azo_arena_ex.argtypes = [u64, u64, u64, u64, i32, i32, u64, u64, u64, i32, u64, i32, i32, i32, vp, vp, vp,
                                   vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.azo_arena.restype = i32
        L.azo_arena.argtypes = [u64, u64, i32, u64, u64, u64, i32, u64, i32, i32, i32, vp, vp]
        L.azo_arena_c3.restype = i32
        L.azo_arena_c3.argtypes = L.azo_arena.argtypes
        L.azo_c3_ended.restype = f32; L.azo_c3_ended.argtypes = [u64, u64]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run_known_answer_tests():
    """Runs oracle/test_oracle (the reference's node.rs / connect_four tests re-stated). Returns (rc, stdout)."""
    build()
    r = subprocess.run([TEST_BIN], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode, r.stdout


def default_reserve(sims):
    return 8 + 42 * (7 * sims + 8)


class Tree:
    """One AsyncMcts (src/async_mcts.rs:14-115) on the CPU oracle."""

    def __init__(self, sims, net_kind=NET_STUB, salt=0, cpuct=1, max_depth=1000, reserve=None, model_id=0,
                 game_kind=GAME_BITS, root=None, quirks=0, threads=1, force_lockstep=False):
        """threads = num_threads of AsyncMcts::default (src/async_mcts.rs:27-36): simulations in flight per tree; > 1 runs the
        oracle's lock-step schedule (az_oracle.hpp)."""
        reserve = reserve or default_reserve(sims)
        has_root = 0 if root is None else 1
        m, t = (0, 0) if root is None else root
        self._keep = None
        self._h = lib().azo_tree_new_mt(game_kind, has_root, m, t, reserve, sims, threads, max_depth, model_id, cpuct, net_kind,
                                        salt, quirks, 1 if force_lockstep else 0)
        if not self._h:
            raise RuntimeError("azo_tree_new failed")

    def get_action_prob(self, mine, theirs, temp, seed=0, game_id=0):
        pi = np.zeros(7, np.float32)
        counts = np.zeros(7, np.uint16)
        q = np.zeros(7, np.float32)
        rc = lib().azo_tree_get_action_prob(self._h, int(mine), int(theirs), temp, seed, game_id, _p(pi), _p(counts), _p(q))
        if rc != 0:
            raise RuntimeError("oracle get_action_prob failed (terminal root or reserve exhausted)")
        return pi, counts, q

    def stats(self):
        out = np.zeros(8, np.uint64)
        lib().azo_tree_stats(self._h, _p(out))
        return dict(zip(("sims", "expansions", "leaf_evals", "link_hits", "terminal_hits", "depth_sum", "nodes", "abandoned"),
                        (int(x) for x in out)))

    def set_replay(self, states, pis, vs):
        """Feed recorded (state, pi, v) rows back in order (a tree created with net_kind NET_REPLAY)."""
        self._keep = (np.ascontiguousarray(states, np.uint64), np.ascontiguousarray(pis, np.float32), np.ascontiguousarray(vs, np.float32))
        lib().azo_tree_set_replay(self._h, _p(self._keep[0]), _p(self._keep[1]), _p(self._keep[2]), len(self._keep[2]))

    def replay_bad(self):
        return bool(lib().azo_tree_replay_bad(self._h))

    def close(self):
        if self._h:
            lib().azo_tree_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def selfplay(n_games, sims, net_kind=NET_STUB, salt=0, seed=0, first_game_id=0, temp_threshold=15, cpuct=1,
             max_depth=1000, reserve=None, game_kind=GAME_BITS, quirks=0, threads=1, want_samples=True, replay=None, sim_threads=1):
    """Coach::execute_episode x n_games on the oracle.  replay = (rec_off [n+1] int64, states [N,2] u64 or None,
    pis [N,7], vs [N]) feeds recorded net outputs back (replay parity)."""
    reserve = reserve or default_reserve(sims)
    cap = n_games * 84
    boards = np.zeros((cap, 2, 6, 7), np.float32) if want_samples else None
    pis = np.zeros((cap, 7), np.float32) if want_samples else None
    zs = np.zeros(cap, np.float32) if want_samples else None
    game_len = np.zeros(n_games, np.int32)
    moves = np.zeros((n_games, 42), np.uint8)
    stats = np.zeros(6, np.uint64)
    bad = np.zeros(n_games, np.int32)
    ro = rs = rp = rv = None
    if replay is not None:
        ro = np.ascontiguousarray(replay[0], np.int64)
        rs = None if replay[1] is None else np.ascontiguousarray(replay[1], np.uint64)
        rp = np.ascontiguousarray(replay[2], np.float32)
        rv = np.ascontiguousarray(replay[3], np.float32)
    n = lib().azo_selfplay(n_games, first_game_id, sims, temp_threshold, cpuct, max_depth, reserve, seed, net_kind, salt,
                           game_kind, quirks, threads, _p(boards), _p(pis), _p(zs), cap, _p(game_len), _p(moves),
                           _p(stats), _p(ro), _p(rs), _p(rp), _p(rv), _p(bad), sim_threads)
    if n < 0:
        raise RuntimeError("oracle selfplay failed")
    res = {"count": int(n), "game_len": game_len, "moves": moves, "replay_bad": bad,
           "stats": dict(zip(("sims", "expansions", "leaf_evals", "link_hits", "terminal_hits", "depth_sum"),
                             (int(x) for x in stats)))}
    if want_samples:
        res.update(boards=boards[:n], pis=pis[:n], zs=zs[:n])
    return res


def arena(num, sims, net_kind=NET_HASH, salt=0, seed=0, new_model_id=1, old_model_id=0, cpuct=1, max_depth=1000,
          reserve=None, threads=1, game_kind=GAME_BITS):
    reserve = reserve or default_reserve(sims)
    wld = np.zeros(3, np.uint64)
    results = np.zeros(max(num, 1), np.int8)
    fn = lib().azo_arena_c3 if game_kind == GAME_CONNECT3 else lib().azo_arena
    rc = fn(num, sims, cpuct, max_depth, reserve, seed, net_kind, salt, new_model_id, old_model_id,
            threads, _p(wld), _p(results))
    if rc != 0:
        raise RuntimeError("oracle arena failed")
    return wld, results[: 2 * (num // 2)]


def arena_ex(total, sims, first_game=0, n_games=None, net_kind=NET_HASH, salt=0, seed=0, new_model_id=1, old_model_id=0, cpuct=1,
             max_depth=1000, reserve=None, threads=1, sim_threads=1, start_board=None, replay_new=None, replay_old=None):
    """Games [first_game, first_game + n_games) of a `total`-game arena.  replay_new / replay_old = (off [n+1] int64, states [N,2]
    u64 or None, pis [N,7], vs [N]): the rows each model's trees consumed, per game (net_kind NET_REPLAY).
    Returns (wld, results, replay_bad)."""
    reserve = reserve or default_reserve(sims)
    n_games = total - first_game if n_games is None else n_games
    wld = np.zeros(3, np.uint64)
    results = np.zeros(max(n_games, 1), np.int8)
    bad = np.zeros(max(n_games, 1), np.int32)
    sb = None if start_board is None else np.ascontiguousarray(start_board, np.uint64)
    keep = []

    def unpack(r):
        if r is None:
            return [None] * 4
        a = [np.ascontiguousarray(r[0], np.int64), None if r[1] is None else np.ascontiguousarray(r[1], np.uint64),
             np.ascontiguousarray(r[2], np.float32), np.ascontiguousarray(r[3], np.float32)]
        keep.append(a)
        return a
    rn, ro = unpack(replay_new), unpack(replay_old)
    rc = lib().azo_arena_ex(total, first_game, n_games, sims, sim_threads, cpuct, max_depth, reserve, seed, net_kind, salt, new_model_id,
                            old_model_id, threads, _p(wld), _p(results), _p(sb), _p(rn[0]), _p(rn[1]), _p(rn[2]), _p(rn[3]),
                            _p(ro[0]), _p(ro[1]), _p(ro[2]), _p(ro[3]), _p(bad))
    if rc != 0:
        raise RuntimeError("oracle arena failed")
    return wld, results[:n_games], bad[:n_games]


_cb_keepalive = None


def set_predict_callback(fn):
    """fn(boards [B,2,6,7] f32, model_id) -> (pi [B,7], v [B]); used by net_kind NET_CALLBACK (single thread)."""
    global _cb_keepalive
    CB = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))

    def tramp(boards, B, model_id, pi, v):
        b = np.ctypeslib.as_array(boards, shape=(B, 2, 6, 7))
        p, vv = fn(b, model_id)
        np.ctypeslib.as_array(pi, shape=(B, 7))[:] = p
        np.ctypeslib.as_array(v, shape=(B,))[:] = vv

    _cb_keepalive = CB(tramp)
    lib().azo_set_predict_callback.restype = None
    lib().azo_set_predict_callback.argtypes = [CB]
    lib().azo_set_predict_callback(_cb_keepalive)


def c4_play(mine, theirs, a):
    out = np.zeros(2, np.uint64)
    lib().azo_c4_play(int(mine), int(theirs), a, _p(out))
    return int(out[0]), int(out[1])


def c4_ended(mine, theirs):
    return float(lib().azo_c4_ended(int(mine), int(theirs)))


def c3_ended(mine, theirs):
    """get_game_ended(1) of the Connect Three variant (GAME_CONNECT3)."""
    return float(lib().azo_c3_ended(int(mine), int(theirs)))


def c4_valid_mask(mine, theirs):
    return int(lib().azo_c4_valid_mask(int(mine), int(theirs)))


def c4_features(mine, theirs):
    f = np.zeros((2, 6, 7), np.float32)
    lib().azo_c4_features(int(mine), int(theirs), _p(f))
    return f


def hashnet(mine, theirs, salt=0):
    pi = np.zeros(7, np.float32)
    v = np.zeros(1, np.float32)
    lib().azo_hashnet(int(mine), int(theirs), salt, _p(pi), _p(v))
    return pi, float(v[0])
