"""GPU parity: the HIP tree kernels against the CPU oracle, through the C ABI.

Bit-exact bar (integer / index work and f32 in the reference's operation order):
visit counts, chosen moves, pi, Q and z must be identical to the oracle's on the same
seeds.  The nets used here are the reference's stub (examples/connect_four.rs:12-43)
and the exact-in-f32 hash fixture, so no net numerics enter (the bf16 conv net is
covered by test_net_gpu.py: numerics alone + replay parity).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HASH_SALT = 1234
MODEL_SALT = 0x51ED27


def oracle_salt(model_id):
    return HASH_SALT + model_id * MODEL_SALT


@pytest.fixture(autouse=True, params=[(1, 1), (1, 0), (2, 1)], ids=["fused-fixture-search", "launch-per-simulation", "dedup-every-net"])
def dedup_mode(request, engine):
    """Every test of this module runs three times, each held to the same bit-exact bar against the oracle:
    the default (the stub / hash fixture nets search in ONE launch per move, k_search_fixture), one launch per simulation
    step with a leaf batch (the path the conv net takes, without de-duplication), and that path with leaf de-duplication +
    the evaluation cache switched on for EVERY net (election table, cache, the backup kernels' indirection)."""
    engine.set_option("eval_dedup", request.param[0])
    engine.set_option("fused_search", request.param[1])
    yield request.param[0]
    engine.set_option("eval_dedup", 1)
    engine.set_option("fused_search", 1)


def play_episode_lockstep(engine, oracle, n_games, sims, model_id, okind, osalt, temp_schedule, seed, max_moves=42):
    """Drives az_tree_get_action_prob move by move for n_games trees and the oracle's AsyncMcts alongside;
    every move compares pi / counts / Q, then plays the oracle-agreed move."""
    tb = engine.tree_create(n_games, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=1000,
                            model_id=model_id, cpuct=1)
    trees = [oracle.Tree(sims, net_kind=okind, salt=osalt) for _ in range(n_games)]
    states = [(0, 0)] * n_games
    alive = [True] * n_games
    rng = np.random.default_rng(seed)
    for move in range(max_moves):
        if not any(alive):
            break
        temp = temp_schedule(move)
        # finished games keep searching their last non-terminal position on both sides (the tree keeps growing)
        pi, counts, q = tb.get_action_prob(np.array(states, dtype=np.uint64), temp, seed=seed, first_game_id=100)
        for g in range(n_games):
            opi, ocnt, oq = trees[g].get_action_prob(states[g][0], states[g][1], temp, seed=seed, game_id=100 + g)
            assert np.array_equal(counts[g], ocnt), (move, g, counts[g], ocnt)
            assert np.array_equal(pi[g], opi), (move, g, pi[g], opi)
            assert np.array_equal(q[g], oq), (move, g, q[g], oq)
            if not alive[g]:
                continue
            legal = [a for a in range(7) if opi[a] > 0]
            a = int(rng.choice(legal))
            nxt = oracle.c4_play(states[g][0], states[g][1], a)
            if oracle.c4_ended(*nxt) != 0.0:
                alive[g] = False        # keep the old state so the engine never sees a terminal root
            else:
                states[g] = nxt
    return tb, trees, tb.node_counts()


@pytest.mark.parametrize("n_games,sims", [(1, 25), (5, 25), (16, 100)])
def test_get_action_prob_stub_net(engine, oracle, n_games, sims):
    """config 1 semantics (stub net, cpuct 1): every move of whole games, counts / pi / Q bit-exact."""
    play_episode_lockstep(engine, oracle, n_games, sims, 0, oracle.NET_STUB, 0, lambda m: 1.0 if m < 14 else 0.0, 3)


@pytest.mark.parametrize("n_games,sims", [(3, 25), (33, 100), (8, 400)])
def test_get_action_prob_hash_net(engine, oracle, n_games, sims):
    play_episode_lockstep(engine, oracle, n_games, sims, 10, oracle.NET_HASH, oracle_salt(10),
                          lambda m: 1.0 if m < 14 else 0.0, 11)


def test_get_action_prob_temp0_tiebreak(engine, oracle):
    """temp == 0 from the first move: the uniform pick among equal max counts uses the build's RNG stream."""
    play_episode_lockstep(engine, oracle, 24, 25, 0, oracle.NET_STUB, 0, lambda m: 0.0, 99)


def test_node_counts_and_stats_match(engine, oracle):
    """NodeStore::len and the expansion / link / terminal counters agree with the oracle."""
    sims, n = 100, 6
    engine.reset_stats()
    tb, trees, nodes = play_episode_lockstep(engine, oracle, n, sims, 10, oracle.NET_HASH, oracle_salt(10),
                                             lambda m: 1.0, 5)
    ost = [t.stats() for t in trees]
    assert [int(x) for x in nodes] == [s["nodes"] for s in ost]
    st = engine.stats()
    for mine, theirs in (("simulations", "sims"), ("expansions", "expansions"), ("leaf_evals", "leaf_evals"),
                         ("link_hits", "link_hits"), ("terminal_hits", "terminal_hits"), ("depth_sum", "depth_sum")):
        assert st[mine] == sum(s[theirs] for s in ost), (mine, st[mine])


def test_puct_term_is_ieee_bit_for_bit(engine_mod, oracle):
    """best_child's PUCT term (src/node.rs:352-356, C6) as the selection kernels compute it (az_diag_puct of the diagnostic
    library runs the kernels' own device functions) against the oracle's IEEE f32 arithmetic, bit for bit: EVERY parent N (the
    square root must be correctly rounded: round 3 found __fsqrt_rn to be a bare v_sqrt_f32, 1 ulp off for some N, which flips the
    arg-max of two children that tie to the last bit), random counters and priors, and the tie the 4096 x 400 arena ran into."""
    import ctypes as C
    L = engine_mod.diag_library()
    L.az_diag_puct.restype = C.c_int
    L.az_diag_puct.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(0)
    n = 65536 * 4
    parent = np.tile(np.arange(65536, dtype=np.uint32), 4)
    wins = rng.integers(-4000, 4000, n).astype(np.int64)
    visits = rng.integers(0, 2000, n).astype(np.uint64)
    visits[:65536] = 6
    ctr = (((wins + 0x7FFFFFFF).astype(np.uint64)) << np.uint64(32)) | (visits << np.uint64(16)) | rng.integers(0, 2, n).astype(np.uint64)
    prior = rng.random(n).astype(np.float32)
    # the arena's tie: children (W = 0.05, N = 6, p = 0x3e1373ad) and (W = 0.04, N = 6, p = 0x3e1540be) under a parent with N = 44
    ctr[:2] = [0x8000000400060000, 0x8000000300060000]
    prior[:2] = np.array([0x3e1373ad, 0x3e1540be], np.uint32).view(np.float32)
    parent[:2] = 44
    for cpuct in (1, 3):
        out = np.zeros(n, np.float32)
        assert L.az_diag_puct(ctr.ctypes.data, prior.view(np.uint32).ctypes.data, parent.ctypes.data, cpuct, out.ctypes.data, n) == 0
        OL = oracle.lib()
        ref = np.array([OL.azo_puct(int(ctr[i]), float(prior[i]), int(parent[i]), cpuct) for i in range(n)], np.float32)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), np.nonzero(out.view(np.uint32) != ref.view(np.uint32))[0][:8]
        if cpuct == 1:
            assert out[0] == out[1]                       # an exact tie: max_by keeps the LATER child (C7)


def _compare_selfplay(got, ref):
    assert got["count"] == ref["count"]
    assert np.array_equal(got["game_len"], ref["game_len"])
    assert np.array_equal(got["moves"], ref["moves"])
    assert np.array_equal(got["boards"].reshape(-1, 84), ref["boards"].reshape(-1, 84))
    assert np.array_equal(got["pis"], ref["pis"])
    assert np.array_equal(got["zs"], ref["zs"])


@pytest.mark.parametrize("n_games,sims,model_id", [(1, 25, 0), (7, 25, 0), (64, 100, 10), (256, 25, 10)])
def test_selfplay_matches_execute_episode(engine, oracle, n_games, sims, model_id):
    """az_selfplay == Coach::execute_episode per game id: moves, (s, pi, z) tuples incl. symmetries."""
    okind = oracle.NET_STUB if model_id == 0 else oracle.NET_HASH
    got = engine.selfplay(n_games=n_games, num_sims=sims, model_id=model_id, seed=42, first_game_id=7)
    ref = oracle.selfplay(n_games, sims, net_kind=okind, salt=oracle_salt(model_id) if model_id else 0, seed=42,
                          first_game_id=7, threads=8)
    _compare_selfplay(got, ref)
    # decoded bitboards agree with the feature planes
    from alphazero_rs_amd import engine as azeng
    for i in range(0, got["count"], max(1, got["count"] // 16)):
        m, t = (int(x) for x in got["states"][i])
        assert np.array_equal(azeng.c4_features(m, t), got["boards"][i])


def test_selfplay_refill_is_slot_independent(engine, oracle):
    """Finished slots are refilled with the next episode id; results depend on the episode id only."""
    n, sims = 96, 25
    ref = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=8, threads=8)
    for concurrent in (96, 32, 5):
        got = engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=8, concurrent=concurrent)
        _compare_selfplay(got, ref)


def test_selfplay_shard_invariance(engine, oracle):
    """Sharding by global game id (multi-GPU partitioning) changes nothing: two half-ranges == one full range."""
    n, sims = 64, 25
    full = engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=21, first_game_id=1000)
    lo = engine.selfplay(n_games=n // 2, num_sims=sims, model_id=10, seed=21, first_game_id=1000)
    hi = engine.selfplay(n_games=n // 2, num_sims=sims, model_id=10, seed=21, first_game_id=1000 + n // 2)
    assert full["count"] == lo["count"] + hi["count"]
    assert np.array_equal(full["pis"], np.concatenate([lo["pis"], hi["pis"]]))
    assert np.array_equal(full["zs"], np.concatenate([lo["zs"], hi["zs"]]))
    assert np.array_equal(full["states"], np.concatenate([lo["states"], hi["states"]]))


def test_selfplay_session_delivers_the_same_episodes_in_chunks(engine, oracle, engine_mod):
    """az_selfplay_begin / _next / _end: the slots stay full across the calls that fetch the episodes.  A chunk is exactly what
    az_selfplay returns for the same episode ids (and so what the oracle's execute_episode plays), whatever the chunk sizes, the
    slot count and the order in which the episodes happened to finish; the session's misuse is refused."""
    n, sims = 160, 25
    ref = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=8, threads=8)
    whole = engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=8, concurrent=24)
    _compare_selfplay(whole, ref)
    for concurrent, chunks in ((24, (40, 40, 40, 40)), (24, (1, 7, 100, 52)), (160, (80, 80)), (5, (159, 1))):
        engine.selfplay_begin(n, sims, 10, seed=8, concurrent=concurrent)
        parts = [engine.selfplay_next(k) for k in chunks]
        with pytest.raises(engine_mod.AzError):
            engine.selfplay_next(1)                                  # nothing left
        engine.selfplay_end()
        assert sum(q["count"] for q in parts) == whole["count"]
        for key in ("pis", "zs", "states", "boards", "game_len", "moves"):
            assert np.array_equal(np.concatenate([q[key] for q in parts]), whole[key]), (concurrent, chunks, key)
    # several simulations in flight per tree (the lock-step schedule), chunked: the same episodes as the one-shot call
    whole4 = engine.selfplay(n_games=48, num_sims=24, model_id=10, seed=5, concurrent=16, num_sim_threads=4)
    engine.selfplay_begin(48, 24, 10, seed=5, concurrent=16, num_sim_threads=4)
    parts = [engine.selfplay_next(k) for k in (10, 38)]
    engine.selfplay_end()
    for key in ("pis", "zs", "game_len", "moves"):
        assert np.array_equal(np.concatenate([q[key] for q in parts]), whole4[key]), key
    with pytest.raises(engine_mod.AzError):
        engine.selfplay_next(1)                                      # no session
    engine.selfplay_begin(16, sims, 10, seed=8)
    with pytest.raises(engine_mod.AzError):
        engine.selfplay_begin(16, sims, 10, seed=8)                  # one session per engine
    with pytest.raises(engine_mod.AzError):
        engine.selfplay(n_games=4, num_sims=sims, model_id=10)       # ... and no one-shot call beside it
    a = engine.selfplay_next(16)
    engine.selfplay_end()
    engine.selfplay_end()                                            # idempotent
    b = engine.selfplay(n_games=16, num_sims=sims, model_id=10, seed=8)
    assert np.array_equal(a["pis"], b["pis"]) and np.array_equal(a["moves"], b["moves"])


def test_selfplay_no_symmetries_and_temp_threshold(engine, oracle):
    got = engine.selfplay(n_games=16, num_sims=25, model_id=10, seed=3, symmetries=False, temp_threshold=4)
    ref = oracle.selfplay(16, 25, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=3, temp_threshold=4, threads=8)
    assert got["count"] * 2 == ref["count"]
    assert np.array_equal(got["pis"], ref["pis"][0::2])
    assert np.array_equal(got["zs"], ref["zs"][0::2])
    assert np.array_equal(got["moves"], ref["moves"])


def test_from_arbitrary_root_s10(engine, oracle):
    """S10: a root state the tree has never seen (arena's second player) is pushed + upgraded, then searched."""
    sims = 50
    # position after 1. d1 c1 2. d2: side to move has stones at c1
    s = (0, 0)
    for a in (3, 2, 3):
        s = oracle.c4_play(s[0], s[1], a)
    tb = engine.tree_create(4, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=10, cpuct=1)
    pi, counts, q = tb.get_action_prob(np.array([s] * 4, dtype=np.uint64), 1.0, seed=1, first_game_id=0)
    t = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10))
    opi, ocnt, oq = t.get_action_prob(s[0], s[1], 1.0, seed=1, game_id=0)
    for g in range(4):
        assert np.array_equal(counts[g], ocnt) and np.array_equal(pi[g], opi) and np.array_equal(q[g], oq)
    assert int(tb.node_counts()[0]) == t.stats()["nodes"]


def test_terminal_and_near_terminal_roots(engine, oracle, engine_mod):
    """Edge cases: a root one move from a win / a full board (terminal expansions, S4), and the
    terminal-root error (the reference panics at src/async_mcts.rs:85)."""
    sims = 100
    # three in a column for the side to move, opponent scattered: a win is one ply away
    s = (0, 0)
    for a in (0, 1, 0, 2, 0, 4):
        s = oracle.c4_play(s[0], s[1], a)
    tb = engine.tree_create(2, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=10, cpuct=1)
    pi, counts, q = tb.get_action_prob(np.array([s, s], dtype=np.uint64), 1.0)
    t = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10))
    opi, ocnt, oq = t.get_action_prob(s[0], s[1], 1.0)
    assert np.array_equal(counts[0], ocnt) and np.array_equal(pi[1], opi) and np.array_equal(q[0], oq)
    # finished game as root -> AZ_ERR_TERMINAL_ROOT
    w = oracle.c4_play(s[0], s[1], 0)
    assert oracle.c4_ended(*w) == -1.0
    tb2 = engine.tree_create(1, reserve=1000, num_sims=10, max_depth=1000, model_id=0, cpuct=1)
    with pytest.raises(engine_mod.AzError) as ei:
        tb2.get_action_prob(np.array([w], dtype=np.uint64), 1.0)
    assert ei.value.status == 5


def test_draw_value_and_full_board(engine, oracle):
    """Play a game to a nearly full board with no winner, then search: DRAW_EPS leaves (e = -1e-4) back up
    as 0 / +0.01 per the packed counter's arithmetic (C4)."""
    # column order that fills the board without any four-in-a-row
    seq = [0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0,  2, 3, 2, 3, 3, 2, 2, 3, 2, 3, 3, 2,
           4, 5, 4, 5, 5, 4, 4, 5, 4, 5, 5, 4,  6, 6, 6]
    s = (0, 0)
    for a in seq:
        s = oracle.c4_play(s[0], s[1], a)
        assert oracle.c4_ended(*s) == 0.0
    sims = 30
    tb = engine.tree_create(1, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=10, cpuct=1)
    pi, counts, q = tb.get_action_prob(np.array([s], dtype=np.uint64), 1.0)
    t = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10))
    opi, ocnt, oq = t.get_action_prob(s[0], s[1], 1.0)
    assert np.array_equal(counts[0], ocnt) and np.array_equal(pi[0], opi) and np.array_equal(q[0], oq)
    assert int(ocnt.sum()) == sims


def test_capacity_error(engine, engine_mod):
    """reserve too small -> AZ_ERR_CAPACITY (the reference asserts at src/node.rs:237)."""
    tb = engine.tree_create(2, reserve=40, num_sims=100, max_depth=1000, model_id=0, cpuct=1)
    with pytest.raises(engine_mod.AzError) as ei:
        tb.get_action_prob(np.zeros((2, 2), np.uint64), 1.0)
    assert ei.value.status == 2


def test_bad_arguments(engine, engine_mod):
    with pytest.raises(engine_mod.AzError):
        engine.tree_create(0, reserve=100, num_sims=10, max_depth=10, model_id=0, cpuct=1)
    with pytest.raises(engine_mod.AzError) as ei:
        engine.tree_create(1, reserve=100, num_sims=10, max_depth=10, model_id=999, cpuct=1).get_action_prob(
            np.zeros((1, 2), np.uint64), 1.0)
    assert ei.value.status == 6


def test_full_size_properties(engine):
    """BASELINE config-2 concurrency (8192 games, 100 sims/move) with the stub net: size-independent
    properties -- every pi is a distribution over legal moves, z in {-1, +1, +-1e-4}, every game ends with
    a win or a full board, sample count == 2 * total plies, and simulations == 100 * plies."""
    engine.reset_stats()
    n = 8192
    got = engine.selfplay(n_games=n, num_sims=100, model_id=0, seed=1, want_boards=False)
    plies = got["game_len"].astype(np.int64)
    assert plies.min() >= 7 and plies.max() <= 42
    assert got["count"] == 2 * plies.sum()
    pis, zs = got["pis"], got["zs"]
    assert np.all(np.abs(pis.sum(axis=1) - 1.0) < 1e-5)
    assert np.all((np.abs(zs) == 1.0) | (np.abs(np.abs(zs) - 1e-4) < 1e-9))
    st = engine.stats()
    assert st["games"] == n and st["simulations"] == 100 * plies.sum()
    # mirror symmetry: odd rows are the mirrored twin of even rows
    assert np.array_equal(pis[0::2], pis[1::2, ::-1])
    assert np.array_equal(zs[0::2], zs[1::2])


def test_full_size_spot_check_against_oracle(engine, oracle):
    """BASELINE config-2 size (8192 concurrent games, 100 sims/move) with the hash net: ALL 8192 games are played on the oracle
    too (16 threads, 1024 games per call: tree-only CPU work, a few seconds) and must match move for move, tuple for tuple."""
    n, sims, seed = 8192, 100, 21
    got = engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=seed, want_boards=False)
    offs = np.concatenate([[0], np.cumsum(2 * got["game_len"].astype(np.int64))])
    assert offs[-1] == got["count"]
    for lo in range(0, n, 1024):
        ref = oracle.selfplay(1024, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=seed, first_game_id=lo, threads=16)
        assert np.array_equal(ref["game_len"], got["game_len"][lo:lo + 1024]) and np.array_equal(ref["moves"], got["moves"][lo:lo + 1024]), lo
        a, b = offs[lo], offs[lo + 1024]
        assert np.array_equal(ref["pis"], got["pis"][a:b]) and np.array_equal(ref["zs"], got["zs"][a:b]), lo


# ---- arena::play_games (C16) -----------------------------------------------------------------------
@pytest.mark.parametrize("num,sims", [(2, 25), (12, 50), (64, 100)])
def test_arena_matches_play_games(engine, oracle, engine_mod, num, sims):
    """az_arena == play_games with per-game tree pairs: every game's result and the W/L/D tally for the new
    model, both seatings, temp 0 with the RNG tie-break, S10 fresh roots for the second player's tree."""
    engine.net_set_kind(31, engine_mod.NET_HASH, 555 - 31 * MODEL_SALT)      # "new": oracle model id 1, salt 555
    engine.net_set_kind(30, engine_mod.NET_HASH, 555 + MODEL_SALT - 30 * MODEL_SALT)
    # oracle: HashNet salt = 555 + model_id * MODEL_SALT with new_model_id = 0 -> 555, old_model_id = 1 -> 555 + MODEL_SALT
    wld, results = engine.arena(num, sims, new_model_id=31, old_model_id=30, seed=13)
    owld, oresults = oracle.arena(num, sims, net_kind=oracle.NET_HASH, salt=555, seed=13, new_model_id=0, old_model_id=1, threads=8)
    assert np.array_equal(results, oresults)
    assert wld.tolist() == owld.tolist() and int(wld.sum()) == 2 * (num // 2)


def test_arena_golden(engine, engine_mod):
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arena.json")))
    engine.net_set_kind(33, engine_mod.NET_HASH, g["salt"] + g["new_model_id"] * MODEL_SALT - 33 * MODEL_SALT)
    engine.net_set_kind(32, engine_mod.NET_HASH, g["salt"] + g["old_model_id"] * MODEL_SALT - 32 * MODEL_SALT)
    wld, results = engine.arena(g["num"], g["sims"], new_model_id=33, old_model_id=32, seed=g["seed"])
    assert wld.tolist() == g["wld"] and results.tolist() == g["results"]


def test_arena_odd_and_empty(engine, engine_mod):
    wld, results = engine.arena(1, 10, new_model_id=0, old_model_id=0)       # num/2 == 0 games per seating
    assert wld.tolist() == [0, 0, 0] and len(results) == 0
    wld, results = engine.arena(5, 10, new_model_id=0, old_model_id=0)       # 2 per seating, the odd one is dropped
    assert int(wld.sum()) == 4 and len(results) == 4


def test_arena_full_size_properties(engine, engine_mod, oracle):
    """BASELINE config 3 (4096 paired games, 400 sims/move) with the hash nets: size-independent properties --
    every game ends, W + L + D = 4096, the tally follows the per-game results under the seat swap -- the three-shard
    partitioning, EVERY game against the oracle, and the arena's move record."""
    engine.net_set_kind(41, engine_mod.NET_HASH, 9001)
    engine.net_set_kind(40, engine_mod.NET_HASH, 9001)
    engine.reset_stats()
    wld, res = engine.arena(4096, 400, new_model_id=41, old_model_id=40, seed=5)
    assert int(wld.sum()) == 4096 and len(res) == 4096 and set(np.unique(res).tolist()) <= {-1, 0, 1}
    wins = int((res[:2048] == 1).sum() + (res[2048:] == -1).sum())
    losses = int((res[:2048] == -1).sum() + (res[2048:] == 1).sum())
    assert [wins, losses, 4096 - wins - losses] == wld.tolist()
    st = engine.stats()
    assert st["games"] == 4096 and st["simulations"] % 400 == 0
    # the multi-GPU partitioning at full size: three ragged shards by GLOBAL game index (seating and RNG follow the global
    # index) reproduce the unsharded arena game for game, and their W/L/D counters add up to its tally
    tot, parts = np.zeros(3, np.uint64), []
    for lo, hi in ((0, 1365), (1365, 2731), (2731, 4096)):
        w, r = engine.arena(hi - lo, 400, new_model_id=41, old_model_id=40, seed=5, first_game=lo, total_games=4096)
        tot += w
        parts.append(r)
    assert np.array_equal(np.concatenate(parts), res) and tot.tolist() == wld.tolist()
    # ALL 4096 games played on the oracle as well (16 threads, 512 games per call), game for game; and the move record of the
    # arena (az_arena_get_moves) leads, move by move under the oracle's rules, to exactly those results
    tally = np.zeros(3, np.uint64)
    for lo in range(0, 4096, 512):
        w, ores, _ = oracle.arena_ex(4096, 400, first_game=lo, n_games=512, net_kind=oracle.NET_HASH, salt=9001, seed=5,
                                     new_model_id=41, old_model_id=40, threads=16)
        assert np.array_equal(ores, res[lo:lo + 512]), lo
        tally += w
    assert tally.tolist() == wld.tolist()
    wld2, res2 = engine.arena(4096, 400, new_model_id=41, old_model_id=40, seed=5)
    glen, gmoves = engine.arena_get_moves(4096)
    assert np.array_equal(res2, res) and glen.min() >= 7 and glen.max() <= 42
    for g in range(4096):
        s, player = (0, 0), 1
        for k in range(int(glen[g])):
            s = oracle.c4_play(s[0], s[1], int(gmoves[g, k]))
            player = -player
        e = oracle.c4_ended(*s)
        assert e != 0.0 and int(res[g]) == (-player if e == -1.0 else (player if e == 1.0 else 0)), g


def test_arena_start_board(engine, oracle, engine_mod):
    """play_games' `board: Option<G>` (src/arena.rs:62-67, :12-16): every game starts from the given position with the first
    seat to move; both trees are rooted at the initial board (AsyncMcts::default, src/coach.rs:333-354) and meet the position
    through S10.  A finished start board never enters the loop (src/arena.rs:18): result = round(ended(+1))."""
    engine.net_set_kind(61, engine_mod.NET_HASH, 777 - 61 * MODEL_SALT)
    engine.net_set_kind(60, engine_mod.NET_HASH, 777 + MODEL_SALT - 60 * MODEL_SALT)
    s = (0, 0)
    for a in (3, 3, 2, 4, 4, 2):                      # six plies: the first seat (+1) is to move again, canonical = absolute
        s = oracle.c4_play(s[0], s[1], a)
    wld, res = engine.arena(12, 50, new_model_id=61, old_model_id=60, seed=4, start_board=s)
    owld, ores, _ = oracle.arena_ex(12, 50, net_kind=oracle.NET_HASH, salt=777, seed=4, new_model_id=0, old_model_id=1, threads=4,
                                    start_board=s)
    assert np.array_equal(res, ores) and wld.tolist() == owld.tolist()
    wld0, res0 = engine.arena(12, 50, new_model_id=61, old_model_id=60, seed=4)
    assert not np.array_equal(res0, res) or wld0.tolist() != wld.tolist() or True      # (a different opening; may coincide)
    # a start board on which the SECOND seat has already won: every game is over before its first move
    w = (0, 0)
    for a in (0, 1, 0, 1, 0, 1, 6, 1):                # the second mover completes column 1
        w = oracle.c4_play(w[0], w[1], a)
    assert oracle.c4_ended(*w) == -1.0
    wld, res = engine.arena(6, 50, new_model_id=61, old_model_id=60, seed=4, start_board=w)
    owld, ores, _ = oracle.arena_ex(6, 50, net_kind=oracle.NET_HASH, salt=777, seed=4, new_model_id=0, old_model_id=1, start_board=w)
    assert res.tolist() == ores.tolist() == [-1] * 6 and wld.tolist() == owld.tolist() == [3, 3, 0]
    with pytest.raises(engine_mod.AzError):
        engine.arena(4, 10, new_model_id=61, old_model_id=60, start_board=(3, 1))       # overlapping stones


# ---- several simulations in flight per tree (num_threads > 1, src/async_mcts.rs:191-217) ---------------------------------
@pytest.mark.parametrize("threads,sims,okind", [(2, 100, "hash"), (4, 100, "hash"), (5, 25, "stub"), (8, 96, "hash")])
def test_several_simulations_in_flight(engine, oracle, threads, sims, okind):
    """The reference's tree-parallel search as the deterministic lock-step schedule (oracle/az_oracle.hpp search_lockstep):
    per step T selections in thread order with visit()'s virtual loss (src/node.rs:77-80, :51-58) and the Locked filter
    (C8, src/node.rs:359-365) visible to the later threads, the step's leaves evaluated together, backups in thread order;
    S11 / S12 abandon where the reference panics.  Whole games, every move: counts / pi / Q bit-exact, node counts, every
    counter incl. the abandoned simulations.  PARITY UNPINNED by the reference (it is racy for T > 1): this pins one legal
    execution, the same one on both sides."""
    model, kind, salt = (10, oracle.NET_HASH, oracle_salt(10)) if okind == "hash" else (0, oracle.NET_STUB, 0)
    G = 6
    engine.reset_stats()
    tb = engine.tree_create(G, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=model, cpuct=1,
                            num_threads=threads)
    trees = [oracle.Tree(sims, net_kind=kind, salt=salt, threads=threads) for _ in range(G)]
    states = [(0, 0)] * G
    alive = [True] * G
    rng = np.random.default_rng(threads)
    for move in range(42):
        if not any(alive):
            break
        temp = 1.0 if move < 10 else 0.0
        pi, counts, q = tb.get_action_prob(np.array(states, dtype=np.uint64), temp, seed=7, first_game_id=50)
        for g in range(G):
            opi, ocnt, oq = trees[g].get_action_prob(states[g][0], states[g][1], temp, seed=7, game_id=50 + g)
            assert np.array_equal(counts[g], ocnt), (move, g, counts[g], ocnt)
            assert np.array_equal(pi[g], opi) and np.array_equal(q[g], oq), (move, g)
            if alive[g]:
                nxt = oracle.c4_play(states[g][0], states[g][1], int(rng.choice([a for a in range(7) if opi[a] > 0])))
                if oracle.c4_ended(*nxt) != 0.0:
                    alive[g] = False
                else:
                    states[g] = nxt
    ost = [t.stats() for t in trees]
    assert [int(x) for x in tb.node_counts()] == [s["nodes"] for s in ost]
    st = engine.stats()
    for mine, theirs in (("simulations", "sims"), ("expansions", "expansions"), ("leaf_evals", "leaf_evals"), ("link_hits", "link_hits"),
                         ("terminal_hits", "terminal_hits"), ("depth_sum", "depth_sum"), ("abandoned_sims", "abandoned")):
        assert st[mine] == sum(s[theirs] for s in ost), (mine, st[mine], sum(s[theirs] for s in ost))


def test_several_simulations_in_flight_selfplay_and_arena(engine, oracle, engine_mod):
    """num_sim_threads through az_selfplay (with slot refill) and az_arena (src/coach.rs:246-255, :333-354), and the contracts:
    num_sims % num_threads == 0 (src/async_mcts.rs:192), at most 8 threads."""
    n, sims, T = 72, 48, 4
    ref = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=12, threads=8, sim_threads=T)
    ref1 = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=12, threads=8)
    assert not np.array_equal(ref["moves"], ref1["moves"])                       # a different search than one simulation at a time
    assert ref["stats"]["sims"] == ref1["stats"]["sims"] * 0 + sims * int(ref["game_len"].sum())
    for concurrent in (n, 20):
        _compare_selfplay(engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=12, concurrent=concurrent, num_sim_threads=T), ref)
    engine.net_set_kind(71, engine_mod.NET_HASH, 888 - 71 * MODEL_SALT)
    engine.net_set_kind(70, engine_mod.NET_HASH, 888 + MODEL_SALT - 70 * MODEL_SALT)
    wld, res = engine.arena(20, 60, new_model_id=71, old_model_id=70, seed=2, num_sim_threads=3)
    owld, ores, _ = oracle.arena_ex(20, 60, net_kind=oracle.NET_HASH, salt=888, seed=2, new_model_id=0, old_model_id=1, threads=8, sim_threads=3)
    assert np.array_equal(res, ores) and wld.tolist() == owld.tolist()
    for bad in (dict(num_sims=50, num_sim_threads=4), dict(num_sims=90, num_sim_threads=9)):
        with pytest.raises(engine_mod.AzError) as ei:
            engine.selfplay(n_games=2, model_id=10, **bad)
        assert ei.value.status == 1


def test_from_state_reset(engine, oracle):
    """AsyncMcts::from_state (src/async_mcts.rs:50-72): trees rooted at an arbitrary position hold exactly the nodes the
    oracle's from_root store holds, and search identically; reset(None) returns to the initial board."""
    sims = 60
    s = (0, 0)
    for a in (3, 3, 2, 4, 1):
        s = oracle.c4_play(s[0], s[1], a)
    tb = engine.tree_create(3, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=10, cpuct=1)
    tb.get_action_prob(np.zeros((3, 2), np.uint64), 1.0)                 # dirty the trees first
    tb.reset(np.array([s] * 3, dtype=np.uint64))
    t = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), root=s)
    assert int(tb.node_counts()[1]) == t.stats()["nodes"] == 1 + bin(oracle.c4_valid_mask(*s)).count("1")
    pi, counts, q = tb.get_action_prob(np.array([s] * 3, dtype=np.uint64), 1.0, seed=2)
    opi, ocnt, oq = t.get_action_prob(s[0], s[1], 1.0, seed=2)
    assert np.array_equal(counts[2], ocnt) and np.array_equal(pi[0], opi) and np.array_equal(q[1], oq)
    assert int(tb.node_counts()[0]) == t.stats()["nodes"]
    tb.reset(None)
    assert int(tb.node_counts()[0]) == 8
    t0 = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10))
    pi, counts, q = tb.get_action_prob(np.zeros((3, 2), np.uint64), 1.0, seed=2)
    opi, ocnt, oq = t0.get_action_prob(0, 0, 1.0, seed=2)
    assert np.array_equal(counts[0], ocnt) and np.array_equal(pi[2], opi)


@pytest.mark.parametrize("cpuct,max_depth", [(2, 1000), (5, 1000), (1, 2), (3, 0)])
def test_cpuct_and_max_depth(engine, oracle, cpuct, max_depth):
    """C6: cpuct is an i32 multiplied in f32 inside the PUCT term; B10: depth > max_depth returns eval_heuristic() = 0
    un-negated (src/async_mcts.rs:241-244) -- with max_depth 0/2 most simulations take that exit."""
    sims, G = 80, 6
    tb = engine.tree_create(G, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=max_depth, model_id=10, cpuct=cpuct)
    trees = [oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), cpuct=cpuct, max_depth=max_depth) for _ in range(G)]
    states = [(0, 0)] * G
    rng = np.random.default_rng(cpuct * 100 + max_depth)
    for move in range(8):
        pi, counts, q = tb.get_action_prob(np.array(states, dtype=np.uint64), 1.0, seed=6)
        for g in range(G):
            opi, ocnt, oq = trees[g].get_action_prob(states[g][0], states[g][1], 1.0, seed=6, game_id=g)
            assert np.array_equal(counts[g], ocnt), (move, g)
            assert np.array_equal(pi[g], opi) and np.array_equal(q[g], oq)
        for g, s in enumerate(states):
            nxt = oracle.c4_play(s[0], s[1], int(rng.choice([a for a in range(7) if pi[g][a] > 0])))
            if oracle.c4_ended(*nxt) == 0.0:          # a finished game keeps searching its last position
                states[g] = nxt
    assert [int(x) for x in tb.node_counts()] == [t.stats()["nodes"] for t in trees]


@pytest.mark.parametrize("temp", [0.5, 2.0, 0.25])
def test_fractional_temperature(engine, oracle, temp):
    """S6 (A7): pi = counts^(1/temp) / sum with f32 powf.  Visit counts are bit-exact; pi goes through powf on both
    sides (libm vs the device library), so the bar is the north-star tolerance 1e-5 (the reference itself only ever
    uses temp 0 and 1, src/coach.rs:122-126)."""
    sims, G = 100, 4
    tb = engine.tree_create(G, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=10, cpuct=1)
    pi, counts, q = tb.get_action_prob(np.zeros((G, 2), np.uint64), temp, seed=1)
    t = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10))
    opi, ocnt, oq = t.get_action_prob(0, 0, temp, seed=1)
    for g in range(G):
        assert np.array_equal(counts[g], ocnt) and np.array_equal(q[g], oq)
        assert np.abs(pi[g] - opi).max() <= 1e-5 and abs(float(pi[g].sum()) - 1.0) < 1e-5


def test_ragged_sizes_and_limits(engine, oracle, engine_mod):
    """Sizes that are not multiples of the 8-lane group / 64-lane wave, single slots, and the documented limits."""
    for n, conc in ((3, 2), (13, 1), (67, 9)):
        got = engine.selfplay(n_games=n, num_sims=25, model_id=10, seed=31, concurrent=conc, first_game_id=500)
        ref = oracle.selfplay(n, 25, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=31, first_game_id=500, threads=4)
        _compare_selfplay(got, ref)
    with pytest.raises(engine_mod.AzError) as ei:
        engine.tree_create(65537, reserve=100, num_sims=10, max_depth=10, model_id=0, cpuct=1)
    assert ei.value.status == 1
    with pytest.raises(engine_mod.AzError):
        engine.selfplay(n_games=4, num_sims=0, model_id=0)
    # reserve below the reachable bound -> AZ_ERR_CAPACITY out of az_selfplay too
    with pytest.raises(engine_mod.AzError) as ei:
        engine.selfplay(n_games=8, num_sims=50, model_id=0, reserve=64)
    assert ei.value.status == 2


def test_conv_net_batch_guard(engine_mod):
    """A tree batch larger than az_config.max_batch cannot use the conv net (workspace bound) -> bad argument."""
    e = engine_mod.Engine(device=0, max_batch=128, net_channels=128)
    try:
        e.net_init_random(0, 1)
        with pytest.raises(engine_mod.AzError) as ei:
            e.selfplay(n_games=256, num_sims=10, model_id=0)
        assert ei.value.status == 1
        r = e.selfplay(n_games=256, num_sims=10, model_id=0, concurrent=128, want_boards=False)
        assert r["count"] == 2 * int(r["game_len"].sum())
    finally:
        e.close()


def test_arena_shards_add_up(engine, oracle, engine_mod):
    """Arena games sharded by GLOBAL game index (one rank per GPU): three shards of a 20-game arena reproduce the
    unsharded call game for game, and the W/L/D counters add up (the 3-counter all-reduce of SURVEY.md 8e)."""
    engine.net_set_kind(51, engine_mod.NET_HASH, 4321 - 51 * MODEL_SALT)
    engine.net_set_kind(50, engine_mod.NET_HASH, 4321 + MODEL_SALT - 50 * MODEL_SALT)
    wld, res = engine.arena(20, 50, new_model_id=51, old_model_id=50, seed=9)
    owld, ores = oracle.arena(20, 50, net_kind=oracle.NET_HASH, salt=4321, seed=9, new_model_id=0, old_model_id=1, threads=8)
    assert np.array_equal(res, ores) and wld.tolist() == owld.tolist()
    tot = np.zeros(3, np.uint64)
    parts = []
    for lo, hi in ((0, 7), (7, 13), (13, 20)):
        w, r = engine.arena(hi - lo, 50, new_model_id=51, old_model_id=50, seed=9, first_game=lo, total_games=20)
        tot += w
        parts.append(r)
    assert np.array_equal(np.concatenate(parts), res) and tot.tolist() == wld.tolist()
    with pytest.raises(engine_mod.AzError):
        engine.arena(5, 10, new_model_id=51, old_model_id=50, first_game=18, total_games=20)


def test_dedup_counters_and_cache_corner_cases(engine, oracle, dedup_mode):
    """requested == executed + cache hits + in-batch duplicates == leaf_evals; a tiny cache that overflows its buckets, a
    stone limit, a cache kept across calls and no cache at all change nothing in the results."""
    if dedup_mode != 2:
        pytest.skip("needs de-duplication on the hash net")
    n, sims = 160, 25
    ref = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=8, threads=8)
    try:
        hits_first = None
        for log2, stones, persist in ((24, 42, 0), (10, 42, 0), (24, 5, 0), (0, 42, 0), (12, 42, 1), (12, 42, 1)):
            engine.set_option("eval_cache_log2", log2)
            engine.set_option("eval_cache_max_stones", stones)
            engine.set_option("eval_cache_persist", persist)
            engine.reset_stats()
            got = engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=8, concurrent=48)
            _compare_selfplay(got, ref)
            st = engine.stats()
            assert st["leaf_rows_requested"] == st["leaf_evals"] > 0
            assert st["leaf_rows_requested"] == st["leaf_rows_executed"] + st["eval_cache_hits"] + st["eval_batch_dups"], st
            assert st["eval_batch_dups"] > 0                      # 48 games from the empty board share their openings
            if log2 == 0:
                assert st["eval_cache_hits"] == 0 and st["eval_cache_inserts"] == 0
            elif persist and hits_first is not None:
                # second call on a kept cache: what the first call published is found again, little is left to insert
                assert st["eval_cache_hits"] > hits_first and st["eval_cache_inserts"] <= st["leaf_rows_executed"]
            else:
                assert st["eval_cache_hits"] > 0 and 0 < st["eval_cache_inserts"] <= st["leaf_rows_executed"]
                if persist:
                    hits_first = st["eval_cache_hits"]
            if log2 == 10:
                assert st["eval_cache_inserts"] <= 1024
        # runs of simulation steps are replayed as one hipGraph ("search_graph" steps per launch; the launch that consumes a
        # batch clears its election table, so every step takes the same arguments): any chunking gives the same games
        engine.set_option("eval_cache_persist", 0)
        for chunk in (0, 2, 6, 20):
            engine.set_option("search_graph", chunk)
            _compare_selfplay(engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=8, concurrent=48), ref)
        engine.set_option("eval_cache_persist", 1)
        # a new net under the same model id must not see the old net's cached rows (persist is still on)
        engine.net_set_kind(10, 1, HASH_SALT + 1)
        got = engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=8, concurrent=48)
        ref2 = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10) + 1, seed=8, threads=8)
        _compare_selfplay(got, ref2)
    finally:
        engine.net_set_kind(10, 1, HASH_SALT)
        engine.set_option("eval_cache_log2", 30)
        engine.set_option("eval_cache_max_stones", 42)
        engine.set_option("eval_cache_persist", 0)
        engine.set_option("search_graph", 20)


def test_tree_arena_is_kept_across_calls(engine, oracle, dedup_mode):
    """az_selfplay / az_arena keep their tree arenas for the next call of the same shape (no hipMalloc / hipFree per call) and
    a recycled arena plays exactly what a fresh one does."""
    n, sims = 40, 25
    ref = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=31, threads=8)
    _compare_selfplay(engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=31, concurrent=16), ref)
    a0 = engine.stats()["tree_arena_allocs"]
    for _ in range(3):
        _compare_selfplay(engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=31, concurrent=16), ref)
    assert engine.stats()["tree_arena_allocs"] == a0
    w1, r1 = engine.arena(num_games=24, num_sims=25, new_model_id=10, old_model_id=11, seed=3)
    a1 = engine.stats()["tree_arena_allocs"]
    w2, r2 = engine.arena(num_games=24, num_sims=25, new_model_id=10, old_model_id=11, seed=3)
    assert engine.stats()["tree_arena_allocs"] == a1 and np.array_equal(w1, w2) and np.array_equal(r1, r2)
    _compare_selfplay(engine.selfplay(n_games=n, num_sims=sims, model_id=10, seed=31, concurrent=16), ref)


def test_second_game_through_the_seam(engine_mod, oracle):
    """trait Game is a seam, not a hard-wired rule set (src/game.rs:10-28): an engine created with az_config.game =
    AZ_GAME_CONNECT_THREE (Connect Four's board and moves, three in a row wins) runs the SAME tree kernel templates
    instantiated with another policy struct (csrc/az_game.h) and matches the oracle's generic AsyncMcts<G> / execute_episode /
    play_games on the twin game bit for bit, in all three tree modes; and it does not play Connect Four."""
    e = engine_mod.Engine(device=0, max_batch=256, net_channels=128, game=engine_mod.GAME_CONNECT_THREE)
    try:
        e.net_set_kind(10, engine_mod.NET_HASH, HASH_SALT)
        e.net_set_kind(11, engine_mod.NET_HASH, HASH_SALT)
        n, sims = 96, 25
        ref = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=5, threads=8, game_kind=oracle.GAME_CONNECT3)
        ref4 = oracle.selfplay(n, sims, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=5, threads=8)
        assert ref["game_len"].mean() < ref4["game_len"].mean() and not np.array_equal(ref["moves"], ref4["moves"])
        owld, ores = oracle.arena(16, 25, net_kind=oracle.NET_HASH, salt=HASH_SALT, seed=9, new_model_id=10, old_model_id=11, threads=8,
                                  game_kind=oracle.GAME_CONNECT3)
        for dedup, fused in ((1, 1), (1, 0), (2, 1)):
            e.set_option("eval_dedup", dedup)
            e.set_option("fused_search", fused)
            _compare_selfplay(e.selfplay(n_games=n, num_sims=sims, model_id=10, seed=5, concurrent=40), ref)
            wld, res = e.arena(num_games=16, num_sims=25, new_model_id=10, old_model_id=11, seed=9)
            assert np.array_equal(wld, owld) and np.array_equal(res, ores)
        # the second game with several simulations in flight per tree (the lock-step schedule is a template over the Game policy too)
        e.set_option("eval_dedup", 1)
        e.set_option("fused_search", 1)
        ref_t = oracle.selfplay(n, 24, net_kind=oracle.NET_HASH, salt=oracle_salt(10), seed=6, threads=8, game_kind=oracle.GAME_CONNECT3, sim_threads=4)
        _compare_selfplay(e.selfplay(n_games=n, num_sims=24, model_id=10, seed=6, concurrent=40, num_sim_threads=4), ref_t)
        # fine-grained entry: one search from a position where three in a row is one move away for the side to move
        s = (0, 0)
        for a in (0, 6, 0, 5):
            s = oracle.c4_play(s[0], s[1], a)
        assert oracle.c3_ended(*oracle.c4_play(s[0], s[1], 0)) == -1.0 and oracle.c4_ended(*oracle.c4_play(s[0], s[1], 0)) == 0.0
        tb = e.tree_create(2, reserve=oracle.default_reserve(50), num_sims=50, max_depth=1000, model_id=10, cpuct=1)
        pi, counts, q = tb.get_action_prob(np.array([s, s], dtype=np.uint64), 1.0)
        t = oracle.Tree(50, net_kind=oracle.NET_HASH, salt=oracle_salt(10), game_kind=oracle.GAME_CONNECT3)
        opi, ocnt, oq = t.get_action_prob(s[0], s[1], 1.0)
        assert np.array_equal(counts[0], ocnt) and np.array_equal(pi[1], opi) and np.array_equal(q[0], oq)
        assert int(tb.node_counts()[0]) == t.stats()["nodes"]
    finally:
        e.close()
    with pytest.raises(engine_mod.AzError):
        engine_mod.Engine(device=0, game=7)
