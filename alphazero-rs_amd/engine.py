"""ctypes mirror of the C ABI in include/az_engine.h.

Names follow the reference's surface: `Engine.net_*` is the `NNet` trait
(src/nnet.rs:35-45), `TreeBatch` is n x `AsyncMcts` (src/async_mcts.rs:14-115),
`Engine.selfplay` is `Coach::execute_episode` x many (src/coach.rs:104-157) and
`Engine.arena` is `arena::play_games` (src/arena.rs:62-99).

The HIP library is mandatory: if it has not been built this module raises at
import -- there is no CPU path.
"""
import ctypes as C
import os
import weakref

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZ_ENGINE_LIB") or os.path.join(_PKG, "libaz_engine.so")      # AZ_ENGINE_LIB: A/B of two builds (tools/)
# The diagnostic twin (built with -DAZ_DIAG): the same ABI plus the superseded kernel generations, forced tiles, clock-stamp builds and
# timing ablations behind az_set_option.  tools/ and the kernel-family bit-identity tests use it (Engine(diag=True)); nothing else.
DIAG_LIB_PATH = os.path.join(_PKG, "libaz_engine_diag.so")

ACTIONS = 7
FEATURES = 84
MAX_PLIES = 42

AZ_OK = 0
STATUS_NAMES = {
    0: "AZ_OK", 1: "AZ_ERR_BAD_ARGUMENT", 2: "AZ_ERR_CAPACITY", 3: "AZ_ERR_HIP", 4: "AZ_ERR_INVALID_MOVE",
    5: "AZ_ERR_TERMINAL_ROOT", 6: "AZ_ERR_NO_MODEL", 7: "AZ_ERR_IO", 8: "AZ_ERR_UNSUPPORTED",
}
NET_STUB, NET_HASH, NET_CONV = 0, 1, 2
GAME_CONNECT_FOUR, GAME_CONNECT_THREE = 0, 1


class AzError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class az_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_batch", C.c_int32), ("net_channels", C.c_int32), ("profile", C.c_int32),
                ("game", C.c_int32)]


class az_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("games", "moves", "simulations", "expansions", "leaf_evals", "link_hits",
                                          "terminal_hits", "depth_sum", "samples", "net_launches")] + \
               [(n, C.c_double) for n in ("net_conv2_ms", "net_conv2_flops", "net_total_ms", "net_total_flops",
                                          "tree_ms", "tree_bytes", "device_ms")] + \
               [(n, C.c_uint64) for n in ("leaf_rows_requested", "leaf_rows_executed", "eval_cache_hits", "eval_batch_dups",
                                          "eval_cache_inserts")] + \
               [(n, C.c_double) for n in ("net_conv3_ms", "net_conv3_flops", "net_conv2_bytes")] + \
               [(n, C.c_uint64) for n in ("tree_launches", "tree_launches_timed", "tree_arena_allocs")] + \
               [(n, C.c_double) for n in ("net_conv4_ms", "net_conv4_flops", "net_fc_ms", "net_fc_flops", "net_rows_timed")] + \
               [("abandoned_sims", C.c_uint64), ("net_conv3_image_rows", C.c_uint64), ("net_conv3_image_launches", C.c_uint64)]


class az_selfplay_params(C.Structure):
    _fields_ = [("n_games", C.c_int32), ("concurrent", C.c_int32), ("num_sims", C.c_int32),
                ("temp_threshold", C.c_int32), ("max_depth", C.c_int32), ("cpuct", C.c_int32),
                ("model_id", C.c_int32), ("symmetries", C.c_int32), ("reserve", C.c_uint64), ("seed", C.c_uint64),
                ("first_game_id", C.c_uint64), ("record_evals", C.c_int32), ("num_sim_threads", C.c_int32)]


class az_samples(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("count", C.c_int64), ("states", C.c_void_p), ("boards", C.c_void_p),
                ("pis", C.c_void_p), ("zs", C.c_void_p), ("game_len", C.c_void_p), ("moves", C.c_void_p)]


class az_arena_params(C.Structure):
    _fields_ = [("num_games", C.c_int32), ("num_sims", C.c_int32), ("max_depth", C.c_int32), ("cpuct", C.c_int32),
                ("new_model_id", C.c_int32), ("old_model_id", C.c_int32), ("reserve", C.c_uint64), ("seed", C.c_uint64),
                ("first_game", C.c_int32), ("total_games", C.c_int32), ("record_evals", C.c_int32), ("num_sim_threads", C.c_int32),
                ("use_start_board", C.c_int32), ("allreduce_wld", C.c_int32), ("start_board", C.c_uint64 * 2)]


# every symbol include/az_engine.h declares (tests check the library exports all of them)
EXPORTS = [
    "az_create", "az_destroy", "az_last_error", "az_set_option", "az_get_stats", "az_reset_stats", "az_net_set_kind", "az_net_free",
    "az_net_init_random", "az_net_load", "az_net_save", "az_net_param_count", "az_net_set_params",
    "az_net_get_params", "az_net_predict", "az_net_predict_states", "az_net_train", "az_net_train_history",
    "az_net_train_begin", "az_net_train_step", "az_net_train_end", "az_tree_create",
    "az_tree_destroy", "az_tree_reset", "az_tree_get_action_prob", "az_tree_record_evals", "az_tree_get_evals",
    "az_tree_node_counts", "az_selfplay", "az_selfplay_begin", "az_selfplay_next", "az_selfplay_end", "az_selfplay_get_evals", "az_arena", "az_arena_get_evals", "az_arena_get_moves",
    "az_comm_unique_id", "az_comm_init", "az_comm_destroy", "az_gather_samples", "az_allreduce_u64",
]
COMM_ID_BYTES = 128


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP engine has not been built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, u64, i64, f32 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64, C.c_float
    sigs = {
        "az_create": (i32, [C.POINTER(az_config), C.POINTER(vp)]),
        "az_destroy": (None, [vp]),
        "az_last_error": (C.c_char_p, [vp]),
        "az_set_option": (i32, [vp, C.c_char_p, i64]),
        "az_get_stats": (i32, [vp, C.POINTER(az_stats)]),
        "az_reset_stats": (i32, [vp]),
        "az_net_set_kind": (i32, [vp, i32, i32, u64]),
        "az_net_init_random": (i32, [vp, i32, u64]),
        "az_net_free": (i32, [vp, i32]),
        "az_net_load": (i32, [vp, i32, C.c_char_p]),
        "az_net_save": (i32, [vp, i32, C.c_char_p]),
        "az_net_param_count": (i64, [vp]),
        "az_net_set_params": (i32, [vp, i32, vp, i64]),
        "az_net_get_params": (i32, [vp, i32, vp, i64]),
        "az_net_predict": (i32, [vp, i32, vp, i32, vp, vp]),
        "az_net_predict_states": (i32, [vp, i32, vp, i32, vp, vp]),
        "az_net_train": (i32, [vp, i32, i32, vp, vp, vp, i64]),
        "az_net_train_history": (i32, [vp, vp, i32]),
        "az_net_train_begin": (i32, [vp, i32]),
        "az_net_train_step": (i32, [vp, vp, vp, vp, i32, u64, i32, vp, vp]),
        "az_net_train_end": (i32, [vp, i32]),
        "az_tree_create": (i32, [vp, i32, u64, i32, i32, i32, i32, i32, C.POINTER(vp)]),
        "az_tree_destroy": (None, [vp]),
        "az_tree_reset": (i32, [vp, vp]),
        "az_tree_get_action_prob": (i32, [vp, vp, f32, u64, u64, vp, vp, vp]),
        "az_tree_record_evals": (i32, [vp, i32]),
        "az_tree_get_evals": (i32, [vp, vp, vp, vp, vp]),
        "az_tree_node_counts": (i32, [vp, vp]),
        "az_selfplay": (i32, [vp, C.POINTER(az_selfplay_params), C.POINTER(az_samples)]),
        "az_selfplay_begin": (i32, [vp, C.POINTER(az_selfplay_params)]),
        "az_selfplay_next": (i32, [vp, C.c_int32, C.POINTER(az_samples)]),
        "az_selfplay_end": (i32, [vp]),
        "az_selfplay_get_evals": (i32, [vp, vp, vp, vp, vp]),
        "az_arena": (i32, [vp, C.POINTER(az_arena_params), vp, vp]),
        "az_arena_get_evals": (i32, [vp, i32, vp, vp, vp, vp]),
        "az_arena_get_moves": (i32, [vp, vp, vp]),
        "az_comm_unique_id": (i32, [vp, vp]),
        "az_comm_init": (i32, [vp, i32, i32, vp]),
        "az_comm_destroy": (i32, [vp]),
        "az_gather_samples": (i32, [vp, C.POINTER(az_samples), i32, C.POINTER(az_samples), vp]),
        "az_allreduce_u64": (i32, [vp, vp, i32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = load_library()
_lib_diag = None


def diag_library():
    global _lib_diag
    if _lib_diag is None:
        _lib_diag = load_library(DIAG_LIB_PATH)
    return _lib_diag


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _as_ptr(x):
    """numpy array -> host pointer; torch tensor -> its data_ptr() (device or host); int passes through."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class Engine:
    """az_engine handle.  One per process per GPU (one HIP stream)."""

    def __init__(self, device=0, max_batch=8192, net_channels=512, profile=False, game=0, diag=False):
        cfg = az_config(device, max_batch, net_channels, 1 if profile else 0, game)
        h = C.c_void_p()
        self._lib = diag_library() if diag else _lib
        st = self._lib.az_create(C.byref(cfg), C.byref(h))
        if st != AZ_OK:
            raise AzError(st, "az_create failed (no usable HIP device?)")
        self._h = h
        self.net_channels = net_channels
        self._trees = weakref.WeakSet()

    def close(self):
        if self._h:
            for t in list(self._trees):   # a tree must not outlive its engine (it borrows the stream)
                t.close()
            self._lib.az_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != AZ_OK:
            raise AzError(st, self._lib.az_last_error(self._h).decode())

    # ---- NNet ----
    def net_set_kind(self, model_id, kind, salt=0):
        self._check(self._lib.az_net_set_kind(self._h, model_id, kind, salt))

    def net_free(self, model_id):
        """Drop a model id (weights + conv1 table); the per-stream activation workspace stays."""
        self._check(self._lib.az_net_free(self._h, model_id))

    def net_init_random(self, model_id, seed):
        self._check(self._lib.az_net_init_random(self._h, model_id, seed))

    def net_param_count(self):
        return int(self._lib.az_net_param_count(self._h))

    def net_set_params(self, model_id, params):
        p = np.ascontiguousarray(params, dtype=np.float32)
        self._check(self._lib.az_net_set_params(self._h, model_id, _ptr(p), p.size))

    def net_get_params(self, model_id):
        p = np.empty(self.net_param_count(), np.float32)
        self._check(self._lib.az_net_get_params(self._h, model_id, _ptr(p), p.size))
        return p

    def net_save(self, model_id, path):
        self._check(self._lib.az_net_save(self._h, model_id, os.fsencode(path)))

    def net_load(self, model_id, path):
        self._check(self._lib.az_net_load(self._h, model_id, os.fsencode(path)))

    def predict(self, boards, model_id):
        """NNet::predict: boards [B,2,6,7] f32 -> (pi [B,7], v [B])."""
        b = np.ascontiguousarray(boards, dtype=np.float32).reshape(-1, FEATURES)
        pi = np.empty((b.shape[0], ACTIONS), np.float32)
        v = np.empty(b.shape[0], np.float32)
        self._check(self._lib.az_net_predict(self._h, model_id, _ptr(b), b.shape[0], _ptr(pi), _ptr(v)))
        return pi, v

    def predict_states(self, states, model_id):
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 2)
        pi = np.empty((s.shape[0], ACTIONS), np.float32)
        v = np.empty(s.shape[0], np.float32)
        self._check(self._lib.az_net_predict_states(self._h, model_id, _ptr(s), s.shape[0], _ptr(pi), _ptr(v)))
        return pi, v

    def train(self, prev_id, model_id, boards, pis, vs):
        b = np.ascontiguousarray(boards, dtype=np.float32)
        p = np.ascontiguousarray(pis, dtype=np.float32)
        v = np.ascontiguousarray(vs, dtype=np.float32)
        self._check(self._lib.az_net_train(self._h, prev_id, model_id, _ptr(b), _ptr(p), _ptr(v), v.size))
        return self.train_history()

    def train_history(self):
        """[(loss_pi, loss_v)] per epoch of the last train()."""
        n = self._lib.az_net_train_history(self._h, None, 0)
        out = np.zeros((n, 2), np.float32)
        self._lib.az_net_train_history(self._h, _ptr(out), n)
        return [tuple(float(x) for x in r) for r in out]

    def train_begin(self, prev_id):
        self._check(self._lib.az_net_train_begin(self._h, prev_id))

    def train_step(self, boards, pis, vs, mask_seed=0, apply=True, want_grads=False):
        """One optimisation step on an explicit batch -> ((loss_pi, loss_v), grads or None)."""
        b = np.ascontiguousarray(boards, dtype=np.float32)
        p = np.ascontiguousarray(pis, dtype=np.float32)
        v = np.ascontiguousarray(vs, dtype=np.float32)
        loss = np.zeros(2, np.float32)
        grads = np.zeros(self.net_param_count(), np.float32) if want_grads else None
        self._check(self._lib.az_net_train_step(self._h, _ptr(b), _ptr(p), _ptr(v), v.size, int(mask_seed), int(bool(apply)),
                                           _ptr(loss), _ptr(grads) if want_grads else None))
        return (float(loss[0]), float(loss[1])), grads

    def train_end(self, model_id):
        self._check(self._lib.az_net_train_end(self._h, model_id))

    def set_option(self, key, value):
        self._check(self._lib.az_set_option(self._h, key.encode(), int(value)))

    # ---- stats ----
    def stats(self):
        s = az_stats()
        self._check(self._lib.az_get_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in az_stats._fields_}

    def reset_stats(self):
        self._check(self._lib.az_reset_stats(self._h))

    # ---- AsyncMcts ----
    def tree_create(self, n_games, reserve, num_sims, max_depth, model_id, cpuct, num_threads=1):
        return TreeBatch(self, n_games, reserve, num_sims, max_depth, model_id, cpuct, num_threads)

    # ---- Coach::execute_episode x many ----
    def selfplay(self, n_games, num_sims, model_id, seed=0, first_game_id=0, concurrent=0, temp_threshold=15,
                 max_depth=1000, cpuct=1, reserve=1000000, symmetries=True, want_boards=True, want_states=True,
                 record_evals=0, out=None, num_sim_threads=1):
        """Plays n_games episodes; returns dict(states, boards, pis, zs, game_len, moves, count).

        `out` may hold pre-allocated torch CUDA tensors / numpy arrays for states/boards/pis/zs
        (the library writes device or host memory alike)."""
        nsym = 2 if symmetries else 1
        cap = n_games * MAX_PLIES * nsym
        out = dict(out or {})
        if "pis" not in out:
            out["pis"] = np.zeros((cap, ACTIONS), np.float32)
        if "zs" not in out:
            out["zs"] = np.zeros(cap, np.float32)
        if want_states and "states" not in out:
            out["states"] = np.zeros((cap, 2), np.uint64)
        if want_boards and "boards" not in out:
            out["boards"] = np.zeros((cap, 2, 6, 7), np.float32)
        game_len = np.zeros(n_games, np.int32)
        moves = np.zeros((n_games, MAX_PLIES), np.uint8)
        p = az_selfplay_params(n_games, concurrent, num_sims, temp_threshold, max_depth, cpuct, model_id,
                               1 if symmetries else 0, reserve, seed, first_game_id, record_evals, num_sim_threads)
        s = az_samples(cap, 0, _as_ptr(out.get("states")), _as_ptr(out.get("boards")), _as_ptr(out["pis"]),
                       _as_ptr(out["zs"]), _ptr(game_len), _ptr(moves))
        self._check(self._lib.az_selfplay(self._h, C.byref(p), C.byref(s)))
        n = int(s.count)
        res = {"count": n, "game_len": game_len, "moves": moves}
        for k in ("states", "boards", "pis", "zs"):
            if k in out:
                res[k] = out[k][:n]
        return res

    def selfplay_begin(self, n_games, num_sims, model_id, seed=0, first_game_id=0, concurrent=0, temp_threshold=15, max_depth=1000,
                       cpuct=1, reserve=1000000, symmetries=True, record_evals=0, num_sim_threads=1):
        """Opens a self-play session of n_games episodes (az_selfplay_begin): selfplay_next(k) then returns the next k episodes in id
        order while the slots they freed already play later ones."""
        p = az_selfplay_params(n_games, concurrent, num_sims, temp_threshold, max_depth, cpuct, model_id,
                               1 if symmetries else 0, reserve, seed, first_game_id, record_evals, num_sim_threads)
        self._check(self._lib.az_selfplay_begin(self._h, C.byref(p)))
        self._sp_sym = bool(symmetries)

    def selfplay_next(self, n_games, want_boards=True, want_states=True, out=None):
        """The next n_games episodes of the open session: the dict selfplay() returns for the same episodes."""
        nsym = 2 if self._sp_sym else 1
        cap = n_games * MAX_PLIES * nsym
        out = dict(out or {})
        if "pis" not in out:
            out["pis"] = np.zeros((cap, ACTIONS), np.float32)
        if "zs" not in out:
            out["zs"] = np.zeros(cap, np.float32)
        if want_states and "states" not in out:
            out["states"] = np.zeros((cap, 2), np.uint64)
        if want_boards and "boards" not in out:
            out["boards"] = np.zeros((cap, 2, 6, 7), np.float32)
        game_len = np.zeros(n_games, np.int32)
        moves = np.zeros((n_games, MAX_PLIES), np.uint8)
        s = az_samples(cap, 0, _as_ptr(out.get("states")), _as_ptr(out.get("boards")), _as_ptr(out["pis"]),
                       _as_ptr(out["zs"]), _ptr(game_len), _ptr(moves))
        self._check(self._lib.az_selfplay_next(self._h, n_games, C.byref(s)))
        n = int(s.count)
        res = {"count": n, "game_len": game_len, "moves": moves}
        for k in ("states", "boards", "pis", "zs"):
            if k in out:
                res[k] = out[k][:n]
        return res

    def selfplay_end(self):
        self._check(self._lib.az_selfplay_end(self._h))

    def selfplay_get_evals(self, n_games, cap):
        cnt = np.zeros(n_games, np.int32)
        states = np.zeros((n_games, cap, 2), np.uint64)
        pis = np.zeros((n_games, cap, ACTIONS), np.float32)
        vs = np.zeros((n_games, cap), np.float32)
        self._check(self._lib.az_selfplay_get_evals(self._h, _ptr(cnt), _ptr(states), _ptr(pis), _ptr(vs)))
        return cnt, states, pis, vs

    # ---- arena::play_games ----
    def arena(self, num_games, num_sims, new_model_id, old_model_id, seed=0, max_depth=1000, cpuct=1,
              reserve=1000000, first_game=0, total_games=0, record_evals=0, num_sim_threads=1, start_board=None, allreduce_wld=False):
        """play_games: (W, L, D) for the new model + per-game results.  total_games > 0 plays the shard
        [first_game, first_game + num_games) of a total_games arena (seating / RNG by global game index).
        start_board = (first seat's stones, second seat's stones): play_games' `board` argument."""
        sb = (C.c_uint64 * 2)(*(int(x) for x in (start_board if start_board is not None else (0, 0))))
        p = az_arena_params(num_games, num_sims, max_depth, cpuct, new_model_id, old_model_id, reserve, seed,
                            first_game, total_games, record_evals, num_sim_threads, 0 if start_board is None else 1,
                            1 if allreduce_wld else 0, sb)
        wld = np.zeros(3, np.uint64)
        results = np.zeros(max(num_games, 1), np.int8)
        self._check(self._lib.az_arena(self._h, C.byref(p), _ptr(wld), _ptr(results)))
        return wld, results[: (num_games if total_games > 0 else 2 * (num_games // 2))]


    def arena_get_moves(self, n_games):
        """Move record of the last arena(): (game_len [n_games], moves [n_games, 42])."""
        game_len = np.zeros(n_games, np.int32)
        moves = np.zeros((n_games, MAX_PLIES), np.uint8)
        self._check(self._lib.az_arena_get_moves(self._h, _ptr(game_len), _ptr(moves)))
        return game_len, moves

    def arena_get_evals(self, which, n_games, cap):
        """Eval log of the last arena(record_evals=cap): rows the trees of player `which` (0 new, 1 old) consumed, per game."""
        cnt = np.zeros(n_games, np.int32)
        states = np.zeros((n_games, cap, 2), np.uint64)
        pis = np.zeros((n_games, cap, ACTIONS), np.float32)
        vs = np.zeros((n_games, cap), np.float32)
        self._check(self._lib.az_arena_get_evals(self._h, which, _ptr(cnt), _ptr(states), _ptr(pis), _ptr(vs)))
        return cnt, states, pis, vs

    # ---- the collective of the sharded Coach loop (RCCL on the engine's stream) ----
    def comm_unique_id(self):
        buf = np.zeros(COMM_ID_BYTES, np.uint8)
        self._check(self._lib.az_comm_unique_id(self._h, _ptr(buf)))
        return buf

    def comm_init(self, rank, world, unique_id):
        buf = np.ascontiguousarray(unique_id, np.uint8)
        assert buf.size == COMM_ID_BYTES
        self._check(self._lib.az_comm_init(self._h, rank, world, _ptr(buf)))
        self._comm_world = int(world)

    def comm_destroy(self):
        self._check(self._lib.az_comm_destroy(self._h))

    def gather_samples(self, states, pis, zs, dst=0, is_dst=True, capacity=0, out=None):
        """ONE gather of this rank's (s, pi, z) tuples to rank dst (-1: every rank receives); returns (states, pis, zs, counts) on a
        receiving rank, (None, None, None, counts) elsewhere.  counts has one entry per rank of the communicator.
        out = {"states", "pis", "zs"}: receive into these buffers (numpy arrays or torch tensors, host or device; capacity = their
        rows) instead of fresh host arrays -- the returned tuple then holds views of them."""
        n = int(len(zs))
        local = az_samples(n, n, _as_ptr(states), None, _as_ptr(pis), _as_ptr(zs), None, None)
        counts = np.zeros(max(1, getattr(self, "_comm_world", 1)), np.int64)
        if is_dst:
            if out is not None:
                gs, gp, gz = out["states"], out["pis"], out["zs"]
                capacity = int(gz.shape[0])
                g = az_samples(capacity, 0, _as_ptr(gs), None, _as_ptr(gp), _as_ptr(gz), None, None)
            else:
                gs, gp, gz = np.zeros((capacity, 2), np.uint64), np.zeros((capacity, ACTIONS), np.float32), np.zeros(capacity, np.float32)
                g = az_samples(capacity, 0, _ptr(gs), None, _ptr(gp), _ptr(gz), None, None)
            self._check(self._lib.az_gather_samples(self._h, C.byref(local), dst, C.byref(g), _ptr(counts)))
            m = int(g.count)
            return gs[:m], gp[:m], gz[:m], counts
        self._check(self._lib.az_gather_samples(self._h, C.byref(local), dst, None, _ptr(counts)))
        return None, None, None, counts

    def allreduce_u64(self, values):
        v = np.ascontiguousarray(values, np.uint64).copy()
        self._check(self._lib.az_allreduce_u64(self._h, _ptr(v), v.size))
        return v


class TreeBatch:
    """n_games x AsyncMcts::default(..) rooted at the initial board (src/async_mcts.rs:27-48)."""

    def __init__(self, engine, n_games, reserve, num_sims, max_depth, model_id, cpuct, num_threads=1):
        self.engine = engine
        self.n_games = n_games
        h = C.c_void_p()
        engine._check(engine._lib.az_tree_create(engine._h, n_games, reserve, num_sims, num_threads, max_depth, model_id, cpuct, C.byref(h)))
        self._h = h
        self._log_cap = 0
        engine._trees.add(self)

    def close(self):
        if self._h and self.engine._h:
            self.engine._lib.az_tree_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_action_prob(self, states, temp, seed=0, first_game_id=0):
        """states [G,2] uint64 canonical bitboards -> (pi [G,7] f32, counts [G,7] u16, q [G,7] f32)."""
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(self.n_games, 2)
        pi = np.empty((self.n_games, ACTIONS), np.float32)
        counts = np.empty((self.n_games, ACTIONS), np.uint16)
        q = np.empty((self.n_games, ACTIONS), np.float32)
        self.engine._check(self.engine._lib.az_tree_get_action_prob(self._h, _ptr(s), temp, seed, first_game_id, _ptr(pi),
                                                        _ptr(counts), _ptr(q)))
        return pi, counts, q

    def reset(self, root_states=None):
        """AsyncMcts::from_state: re-root every tree (None = the initial board)."""
        s = None if root_states is None else np.ascontiguousarray(root_states, dtype=np.uint64).reshape(self.n_games, 2)
        self.engine._check(self.engine._lib.az_tree_reset(self._h, _ptr(s)))

    def record_evals(self, cap):
        self.engine._check(self.engine._lib.az_tree_record_evals(self._h, cap))
        self._log_cap = cap

    def get_evals(self):
        cap = self._log_cap
        cnt = np.zeros(self.n_games, np.int32)
        states = np.zeros((self.n_games, cap, 2), np.uint64)
        pis = np.zeros((self.n_games, cap, ACTIONS), np.float32)
        vs = np.zeros((self.n_games, cap), np.float32)
        self.engine._check(self.engine._lib.az_tree_get_evals(self._h, _ptr(cnt), _ptr(states), _ptr(pis), _ptr(vs)))
        return cnt, states, pis, vs

    def node_counts(self):
        out = np.zeros(self.n_games, np.uint32)
        self.engine._check(self.engine._lib.az_tree_node_counts(self._h, _ptr(out)))
        return out


# ---- host-side helpers on canonical bitboards (mirror of the Game trait for Connect Four) ----
def c4_play(mine, theirs, a):
    """get_next_state(1, a) then get_canonical_form(next_player) (connect_four_game.rs:90-103, :198-203)."""
    mask = mine | theirs
    nb = (mask + (1 << (a * 7))) & (0x3F << (a * 7))
    return theirs, mine | nb


def c4_features(mine, theirs):
    """to_features: [2,6,7] f32 (connect_four_game.rs:219-237, NCHW per repair S8)."""
    f = np.zeros((2, 6, 7), np.float32)
    for r in range(6):
        for c in range(7):
            bit = 1 << (c * 7 + (5 - r))
            if mine & bit:
                f[0, r, c] = 1.0
            if theirs & bit:
                f[1, r, c] = 1.0
    return f
