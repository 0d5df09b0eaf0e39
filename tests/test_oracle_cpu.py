"""CPU tests of the oracle: the reference's own known-answer tests, the committed golden vectors, and the
cross-checks between the oracle's two Connect Four representations.  No GPU needed."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def f32(bits):
    return np.asarray(bits, np.uint32).view(np.float32)


def test_reference_known_answer_tests(oracle):
    """src/node.rs:393-655 (13 tests) + connect_four_game.rs:244-264 re-stated in oracle/test_oracle.cpp."""
    rc, out = oracle.run_known_answer_tests()
    assert rc == 0, out
    passed = [l for l in out.splitlines() if l.startswith("PASS")]
    assert len(passed) == 18, out
    for name in ("test_win", "test_loss", "test_winloss", "test_nodestore_upgrade_many_similar", "test_nodestore_upgrade",
                 "test_nodestore_lock", "test_win_diagonal_array", "test_win_diagonal_bits"):
        assert f"PASS {name}" in out


def test_packed_counter_literal_values(oracle):
    """C1-C5 at the test scale the reference uses (1e4): the +1 LSB bias of non-negative backups is kept."""
    L = oracle.lib()
    c = L.azo_ctr_init()
    assert c == 0x7FFFFFFF00000000
    c = L.azo_ctr_visit(c)
    assert (L.azo_ctr_n(c), L.azo_ctr_vloss(c)) == (1, 1) and L.azo_ctr_w(c, 1e4) == 0.0
    c = L.azo_ctr_unvisit(c, 1.0, 1e4)
    assert (L.azo_ctr_n(c), L.azo_ctr_vloss(c)) == (1, 0)
    assert L.azo_ctr_w(c, 1e4) == np.float32(10001) / np.float32(1e4)
    c2 = L.azo_ctr_unvisit(L.azo_ctr_visit(L.azo_ctr_init()), -1.0, 1e4)
    assert L.azo_ctr_w(c2, 1e4) == -1.0
    # production scale 100: v=+1 -> 1.01, v=0 -> +0.01, v=-1e-4 -> unchanged
    c3 = L.azo_ctr_unvisit(L.azo_ctr_visit(L.azo_ctr_init()), 0.0, 100.0)
    assert L.azo_ctr_w(c3, 100.0) == np.float32(1) / np.float32(100)
    c4 = L.azo_ctr_unvisit(L.azo_ctr_visit(L.azo_ctr_init()), -1e-4, 100.0)
    assert L.azo_ctr_w(c4, 100.0) == 0.0 and L.azo_ctr_n(c4) == 1


def test_golden_counter_and_puct(oracle):
    g = json.load(open(os.path.join(GOLD, "counter_puct.json")))
    L = oracle.lib()
    c = L.azo_ctr_init()
    for step in g["sequence"]:
        c = L.azo_ctr_unvisit(L.azo_ctr_visit(c), float(f32([step["v"]])[0]), 100.0)
        assert str(c) == step["ctr"]
        assert np.float32(L.azo_ctr_w(c, 100.0)).view(np.uint32) == step["w"]
        assert np.float32(L.azo_ctr_q(c, 100.0)).view(np.uint32) == step["q"]
    for p in g["puct"]:
        u = L.azo_puct(int(p["ctr"]), float(f32([p["prior"]])[0]), p["parent_n"], p["cpuct"])
        assert np.float32(u).view(np.uint32) == p["u"]


def replay_trace(oracle, entry, game_kind):
    kind = oracle.NET_STUB if entry["net"] == "stub" else oracle.NET_HASH
    t = oracle.Tree(entry["sims"], net_kind=kind, salt=entry["salt"], game_kind=game_kind)
    for mv in entry["trace"]["moves"]:
        m, th = int(mv["state"][0]), int(mv["state"][1])
        pi, counts, q = t.get_action_prob(m, th, mv["temp"], seed=entry["seed"], game_id=entry["game_id"])
        assert counts.tolist() == mv["counts"]
        assert pi.view(np.uint32).tolist() == mv["pi"] and q.view(np.uint32).tolist() == mv["q"]
        assert int(counts.sum()) >= entry["sims"] - 1
    st = t.stats()
    assert st.pop("abandoned") == 0          # only a several-simulations-in-flight search can abandon one
    assert st == entry["trace"]["stats"]


@pytest.mark.parametrize("game_kind", [0, 1])
def test_golden_search_traces(oracle, game_kind):
    """Committed per-move root visit counts / pi / Q (config 1 verbatim, 100 and 400 sims): the bitboard oracle
    and the array oracle (the reference's own i8[6][7] layout) both reproduce them."""
    g = json.load(open(os.path.join(GOLD, "search_traces.json")))
    replay_trace(oracle, g["config1_stub_25sims"], game_kind)
    for e in g["hash_100sims"]:
        replay_trace(oracle, e, game_kind)
    if game_kind == 0:
        replay_trace(oracle, g["hash_400sims"], game_kind)
    assert len(g["config1_stub_25sims"]["trace"]["moves"]) == 40     # config 1: a 40-ply game


def test_golden_selfplay_tuples(oracle):
    g = json.load(open(os.path.join(GOLD, "selfplay_tuples.json")))
    r = oracle.selfplay(g["n_games"], g["sims"], net_kind=oracle.NET_HASH, salt=g["salt"], seed=g["seed"],
                        first_game_id=g["first_game_id"], threads=4)
    assert r["count"] == g["count"] and r["game_len"].tolist() == g["game_len"] and r["moves"].tolist() == g["moves"]
    assert np.packbits(r["boards"].astype(np.uint8).reshape(-1)).tolist() == g["boards_packed"]
    assert r["pis"].reshape(-1).view(np.uint32).tolist() == g["pis"] and r["zs"].view(np.uint32).tolist() == g["zs"]
    assert r["stats"] == g["stats"]
    # structure of the tuples (C15/B4): symmetric pairs, z = +-1 or +-DRAW_EPS, pi sums to 1
    assert np.array_equal(r["pis"][0::2], r["pis"][1::2, ::-1])
    assert np.array_equal(r["boards"][0::2], r["boards"][1::2, :, :, ::-1])
    assert set(np.abs(r["zs"]).tolist()) <= {1.0, float(np.float32(1e-4))}


def test_golden_arena(oracle):
    g = json.load(open(os.path.join(GOLD, "arena.json")))
    wld, results = oracle.arena(g["num"], g["sims"], net_kind=oracle.NET_HASH, salt=g["salt"], seed=g["seed"],
                                new_model_id=g["new_model_id"], old_model_id=g["old_model_id"], threads=4)
    assert wld.tolist() == g["wld"] and results.tolist() == g["results"]
    assert int(wld.sum()) == g["num"]
    # seat-swap accounting (C16): first half is (new, old), second half (old, new)
    half = g["num"] // 2
    wins = int((results[:half] == 1).sum() + (results[half:] == -1).sum())
    assert wins == g["wld"][0]


def test_array_and_bitboard_oracles_agree(oracle):
    a = oracle.selfplay(12, 50, net_kind=oracle.NET_HASH, salt=5, seed=2, game_kind=oracle.GAME_BITS, threads=4)
    b = oracle.selfplay(12, 50, net_kind=oracle.NET_HASH, salt=5, seed=2, game_kind=oracle.GAME_ARRAY, threads=4)
    for k in ("moves", "game_len", "pis", "zs", "boards"):
        assert np.array_equal(a[k], b[k]), k
    assert a["stats"] == b["stats"]


def test_thread_count_does_not_change_results(oracle):
    a = oracle.selfplay(16, 25, net_kind=oracle.NET_STUB, seed=9, threads=1)
    b = oracle.selfplay(16, 25, net_kind=oracle.NET_STUB, seed=9, threads=8)
    assert np.array_equal(a["moves"], b["moves"]) and np.array_equal(a["pis"], b["pis"])


def test_repair_toggles_change_behaviour(oracle):
    """The ref_quirk switches are live: B2 (same-sign backup), B4 (literal z) and B6 (literal win windows) each
    change the outcome, i.e. each repair matters."""
    base = oracle.selfplay(8, 50, net_kind=oracle.NET_HASH, salt=1, seed=4, threads=4)
    b2 = oracle.selfplay(8, 50, net_kind=oracle.NET_HASH, salt=1, seed=4, threads=4, quirks=oracle.QUIRK_B2)
    assert not np.array_equal(base["moves"], b2["moves"])
    b4 = oracle.selfplay(8, 50, net_kind=oracle.NET_HASH, salt=1, seed=4, threads=4, quirks=oracle.QUIRK_B4)
    assert np.array_equal(base["moves"], b4["moves"]) and not np.array_equal(base["zs"], b4["zs"])
    assert set(np.abs(b4["zs"]).tolist()) == {1.0}
    # B6: a bottom-row four starting in column 3 is invisible to the literal loops (`0..W-L`)
    s = (0, 0)
    for a in (3, 3, 4, 4, 5, 5, 6):
        s = oracle.c4_play(s[0], s[1], a)
    L = oracle.lib()
    assert oracle.c4_ended(*s) == -1.0
    assert L.azo_c4_ended_array(s[0], s[1], 0) == -1.0 and L.azo_c4_ended_array(s[0], s[1], 1) == 0.0


def test_terminal_root_and_reserve_errors(oracle):
    s = (0, 0)
    for a in (0, 1, 0, 1, 0, 1, 0):
        s = oracle.c4_play(s[0], s[1], a)
    assert oracle.c4_ended(*s) == -1.0
    with pytest.raises(RuntimeError):
        oracle.Tree(10).get_action_prob(s[0], s[1], 1.0)
    with pytest.raises(RuntimeError):
        oracle.Tree(100, reserve=40).get_action_prob(0, 0, 1.0)


def test_rng_stream(oracle):
    L = oracle.lib()
    draws = {L.azo_rng_draw(1, g, p, k) for g in range(8) for p in range(8) for k in (1, 2)}
    assert len(draws) == 128
    w = np.array([0, 0.25, 0, 0.75, 0, 0, 0], np.float32)
    picks = [L.azo_rng_choose_weighted(L.azo_rng_draw(3, i, 0, 2), w.ctypes.data, 7) for i in range(4000)]
    assert set(picks) == {1, 3} and 0.70 < picks.count(3) / 4000 < 0.80
    assert all(L.azo_rng_choose(L.azo_rng_draw(5, i, 0, 1), 3) in (0, 1, 2) for i in range(100))


def test_lockstep_schedule_with_one_thread_is_the_plain_search(oracle):
    """The oracle's lock-step schedule (num_threads > 1, az_oracle.hpp search_lockstep) run with ONE thread must be the reference's
    search_iteration loop itself: same counts / pi / Q at every move of a whole game, same counters, nothing abandoned.  And with
    several threads it is a different (and deterministic) search that still spends exactly num_sims simulations per move."""
    sims = 60
    a = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=11)
    b = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=11, force_lockstep=True)
    c = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=11, threads=4)
    c2 = oracle.Tree(sims, net_kind=oracle.NET_HASH, salt=11, threads=4)
    s = (0, 0)
    differs = False
    for move in range(42):
        temp = 1.0 if move < 8 else 0.0
        ra = a.get_action_prob(s[0], s[1], temp, seed=3, game_id=1)
        rb = b.get_action_prob(s[0], s[1], temp, seed=3, game_id=1)
        rc = c.get_action_prob(s[0], s[1], temp, seed=3, game_id=1)
        rc2 = c2.get_action_prob(s[0], s[1], temp, seed=3, game_id=1)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y), move
        for x, y in zip(rc, rc2):
            assert np.array_equal(x, y), move
        differs |= not np.array_equal(ra[1], rc[1])
        s = oracle.c4_play(s[0], s[1], int(np.argmax(ra[1])))
        if oracle.c4_ended(*s) != 0.0:
            break
    sa, sb, sc = a.stats(), b.stats(), c.stats()
    assert sa == sb and sb["abandoned"] == 0
    assert sc["sims"] == sa["sims"] and differs
