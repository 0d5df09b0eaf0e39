"""GPU net tests: the bf16 MFMA policy+value net against a plain PyTorch fp32 reference (numerics alone),
and REPLAY PARITY: the oracle consumes the GPU net's recorded (pi, v) so the search stays bit-exact.

Tolerances (floating point, so stated).  Against the bf16-emulating reference the only differences are the
f32 accumulation order inside the MFMA tiles, which now and then flips a bf16 rounding of an activation
(measured on MI355X: max |dpi| 5e-4, max |dv| 2e-3 with O(1) logits) -> bar |dpi| <= 2e-3, |dv| <= 6e-3.
Against the textbook f32 net with explicit BatchNorm the bf16 weight/activation rounding is added
(measured 6e-4 / 3e-3) -> bar |dpi| <= 5e-3, |dv| <= 1.5e-2.  An indexing / tap / fragment-layout bug shows
up as O(0.1-1) errors, far above either bar.
"""
import os

import numpy as np
import pytest

from net_ref import forward_ref, layout, random_params

pytestmark = pytest.mark.gpu
C = 512


def random_states(oracle, n, seed):
    """Legal canonical positions at random depths."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        s = (0, 0)
        for _ in range(int(rng.integers(0, 30))):
            vm = oracle.c4_valid_mask(*s)
            a = int(rng.choice([c for c in range(7) if (vm >> c) & 1]))
            nxt = oracle.c4_play(s[0], s[1], a)
            if oracle.c4_ended(*nxt) != 0.0:
                break
            s = nxt
        out.append(s)
    return np.array(out, dtype=np.uint64)


@pytest.fixture(scope="module")
def conv_engine(engine_mod):
    e = engine_mod.Engine(device=0, max_batch=2048, net_channels=C)
    assert e.net_param_count() == layout(C)[1]
    yield e
    e.close()


@pytest.fixture(scope="module")
def diag_engine(engine_mod):
    """An engine of libaz_engine_diag.so (the same sources built with -DAZ_DIAG): the superseded kernel generations, forced tiles and
    kernel-family switches live there; the shipped library refuses them (test_shipped_library_refuses_diagnostic_options)."""
    e = engine_mod.Engine(device=0, max_batch=2048, net_channels=C, diag=True)
    yield e
    e.close()


@pytest.mark.parametrize("batch", [1, 3, 130, 700])
def test_net_matches_torch_reference(conv_engine, oracle, batch):
    params = random_params(C, seed=batch)
    conv_engine.net_set_params(2, params)
    states = random_states(oracle, batch, seed=100 + batch)
    boards = np.stack([oracle.c4_features(int(m), int(t)) for m, t in states])
    pi, v = conv_engine.predict_states(states, 2)
    pi2, v2 = conv_engine.predict(boards, 2)          # NNet::predict on [B,2,6,7] planes: same rows
    assert np.array_equal(pi, pi2) and np.array_equal(v, v2)
    rpi, rv = forward_ref(params, boards, C, emulate_bf16=True)
    assert np.abs(pi - rpi).max() <= 2e-3, np.abs(pi - rpi).max()
    assert np.abs(v - rv).max() <= 6e-3, np.abs(v - rv).max()
    assert np.all(np.abs(pi.sum(axis=1) - 1) < 1e-5)
    if batch <= 130:
        fpi, fv = forward_ref(params, boards, C, emulate_bf16=False)
        assert np.abs(pi - fpi).max() <= 5e-3, np.abs(pi - fpi).max()
        assert np.abs(v - fv).max() <= 1.5e-2, np.abs(v - fv).max()


def test_net_rows_are_batch_independent(conv_engine, oracle):
    """A row's output does not depend on its position in the batch or on the batch size (BN is folded, every
    row's K-sum has the same order) -- what makes ragged, compacted leaf batches reproducible."""
    conv_engine.net_init_random(3, seed=7)
    states = random_states(oracle, 300, seed=5)
    pi, v = conv_engine.predict_states(states, 3)
    perm = np.random.default_rng(0).permutation(300)
    pi_p, v_p = conv_engine.predict_states(states[perm], 3)
    assert np.array_equal(pi[perm], pi_p) and np.array_equal(v[perm], v_p)
    pi_1, v_1 = conv_engine.predict_states(states[17:18], 3)
    assert np.array_equal(pi[17:18], pi_1) and np.array_equal(v[17:18], v_1)


def test_gemm_kernel_variants_are_bit_identical(diag_engine, conv_engine, oracle):
    """(On the diagnostic library, checked against the shipped one.)  The implicit-GEMM kernel sets (128x128 register-staged, 256x256 LDS-DMA unphased / phased, image-resident conv2
    as one 8-wave or two 4-wave workgroups per CU) sum every output row's K terms in the same order: their results are identical bit for bit, at ragged and
    tile-aligned batch sizes, so the A/B switch never changes what a search sees."""
    shipped = conv_engine
    shipped.net_init_random(5, seed=21)
    conv_engine = diag_engine
    conv_engine.net_init_random(5, seed=21)
    try:
        for table in (1, 0):                            # the shipped library's two kernel sets == the diagnostic library's defaults
            shipped.set_option("conv2_table", table)
            conv_engine.set_option("conv2_table", table)
            st0 = random_states(oracle, 777, seed=3)
            a, b = shipped.predict_states(st0, 5), conv_engine.predict_states(st0, 5)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), table
        shipped.set_option("conv2_table", 1)
        conv_engine.set_option("conv2_table", 0)        # the GEMM kernel sets; conv2 as a table has its own rounding (below)
        for n in (1, 257, 1530):
            states = random_states(oracle, n, seed=900 + n)
            outs = []
            for variant in (0, 1, 2, 3, 5):
                conv_engine.set_option("gemm_variant", variant)
                outs.append(conv_engine.predict_states(states, 5))
            for pi, v in outs[1:]:
                assert np.array_equal(pi, outs[0][0]) and np.array_equal(v, outs[0][1]), n
        for big in (0, 1):                                    # conv4 on the small / the large tile kernel
            conv_engine.set_option("conv4_big", big)
            pi, v = conv_engine.predict_states(states, 5)
            assert np.array_equal(pi, outs[0][0]) and np.array_equal(v, outs[0][1])
        conv_engine.set_option("conv4_big", 0)
        ring_tiles = tuple(layer * 10000 + t for layer in (3, 4, 5) for t in (642, 644, 962, 964, 1282, 1284, 1602, 1922)) + (30000, 40000, 50000)
        for key, vals in (("fc_ring", (0, 2, 3, 1)), ("conv3_ring", (1, 2, 0)), ("conv1_table", (0, 1)),   # the LDS-DMA ring GEMM on fc1 /
                          ("conv3_pipe", (0, 2, 3, 1)), ("conv2_pipe", (0, 1)), ("ring_tile", ring_tiles)):   # fc2 / conv4 / conv3, conv1 as a kernel, the conv3 kernels, every ring tile
            for val in vals:
                conv_engine.set_option(key, val)
                pi, v = conv_engine.predict_states(states, 5)
                assert np.array_equal(pi, outs[0][0]) and np.array_equal(v, outs[0][1]), (key, val)
        small = states[:100]                                  # a small expected batch: conv3 on the 4-stage ring ("conv3_small"), bit-identical too
        ref_small = conv_engine.predict_states(small, 5)
        assert np.array_equal(ref_small[0], outs[0][0][:100]) and np.array_equal(ref_small[1], outs[0][1][:100])
        for key, val in (("conv3_small", 0), ("conv3_ring", 3), ("conv3_ring", 0), ("conv3_small", 1)):
            conv_engine.set_option(key, val)
            pi, v = conv_engine.predict_states(small, 5)
            assert np.array_equal(pi, ref_small[0]) and np.array_equal(v, ref_small[1]), (key, val)
        with pytest.raises(Exception):
            conv_engine.set_option("gemm_variant", 4)         # removed variants are refused, not silently mapped
    finally:
        conv_engine.set_option("gemm_variant", 5)
        conv_engine.set_option("conv4_big", 0)
        conv_engine.set_option("conv2_table", 1)
        conv_engine.set_option("conv3_pipe", 1)
        conv_engine.set_option("conv2_pipe", 1)
        for layer in (3, 4, 5):
            conv_engine.set_option("ring_tile", layer * 10000)
        conv_engine.set_option("fc_ring", 1)
        conv_engine.set_option("conv3_ring", 0)
        conv_engine.set_option("conv1_table", 1)


def test_net_parity_at_bench_scale(engine, engine_mod, oracle):
    """The regime bench.py drives the net in (NNet::predict on the full batch, src/async_mcts.rs:150-151): ONE call of 8192 rows
    and one ragged call of 5003 rows (not a multiple of any tile: 6- and 12-board image tiles, 128/256-row GEMM tiles, the
    8-XCD tile remap with more than one grid round, conv4 on either kernel) must equal, bit for bit, the same rows predicted
    in chunks of 256 -- the chunk size test_net_matches_torch_reference pins to the torch reference -- for the default and
    the plain kernel set; plus one direct comparison with the bf16-emulating torch reference at 4096 rows."""
    engine.net_init_random(20, seed=31)
    states = random_states(oracle, 8192, seed=77)
    diag = engine_mod.Engine(device=0, max_batch=8192, net_channels=C, diag=True)      # the older kernel families, same rows
    try:
        diag.net_set_params(20, engine.net_get_params(20))
        engine.set_option("conv2_table", 0)
        ref_pi = np.empty((8192, 7), np.float32)
        ref_v = np.empty(8192, np.float32)
        for o in range(0, 8192, 256):
            ref_pi[o:o + 256], ref_v[o:o + 256] = engine.predict_states(states[o:o + 256], 20)
        for n in (8192, 5003):
            pi, v = engine.predict_states(states[:n], 20)
            assert np.array_equal(pi, ref_pi[:n]) and np.array_equal(v, ref_v[:n]), n
        diag.set_option("conv2_table", 0)
        for variant, table in ((5, 1), (5, 0), (0, 0)):        # table: conv2 gathers from the conv1 table / reads k_conv1's act1
            for big in (0, 1, 2):
                diag.set_option("gemm_variant", variant)
                diag.set_option("conv1_table", table)
                diag.set_option("conv4_big", big)
                for n in (8192, 5003):
                    pi, v = diag.predict_states(states[:n], 20)
                    assert np.array_equal(pi, ref_pi[:n]) and np.array_equal(v, ref_v[:n]), (variant, table, big, n)
        # a different row order through the full-size kernels (rows land in other tiles / XCDs)
        perm = np.random.default_rng(3).permutation(8192)
        pi, v = engine.predict_states(states[perm], 20)
        assert np.array_equal(pi, ref_pi[perm]) and np.array_equal(v, ref_v[perm])
    finally:
        engine.set_option("conv2_table", 1)
        diag.close()
    n = 4096
    boards = np.stack([oracle.c4_features(int(m), int(t)) for m, t in states[:n]])
    rpi, rv = forward_ref(engine.net_get_params(20), boards, C, emulate_bf16=True)
    assert np.abs(ref_pi[:n] - rpi).max() <= 2e-3, np.abs(ref_pi[:n] - rpi).max()
    assert np.abs(ref_v[:n] - rv).max() <= 6e-3, np.abs(ref_v[:n] - rv).max()
    # the default set: conv2 as nine gathered table rows.  Its own rounding, so its own reference chunks: full-size calls ==
    # chunks of 256 == any row order, bit for bit, and the same tolerance against the torch reference
    tab_pi = np.empty((8192, 7), np.float32)
    tab_v = np.empty(8192, np.float32)
    for o in range(0, 8192, 256):
        tab_pi[o:o + 256], tab_v[o:o + 256] = engine.predict_states(states[o:o + 256], 20)
    for m in (8192, 5003):
        pi, v = engine.predict_states(states[:m], 20)
        assert np.array_equal(pi, tab_pi[:m]) and np.array_equal(v, tab_v[:m]), m
    pi, v = engine.predict_states(states[perm], 20)
    assert np.array_equal(pi, tab_pi[perm]) and np.array_equal(v, tab_v[perm])
    diag = engine_mod.Engine(device=0, max_batch=8192, net_channels=C, diag=True)
    try:                                                            # the gather as whole rows per wave (round 2's first kernel): same bits
        diag.net_set_params(20, engine.net_get_params(20))
        diag.set_option("conv2_table", 2)
        for m in (8192, 5003, 3):
            pi, v = diag.predict_states(states[:m], 20)
            assert np.array_equal(pi, tab_pi[:m]) and np.array_equal(v, tab_v[:m]), m
    finally:
        diag.close()
    assert np.abs(tab_pi[:n] - rpi).max() <= 2e-3, np.abs(tab_pi[:n] - rpi).max()
    assert np.abs(tab_v[:n] - rv).max() <= 6e-3, np.abs(tab_v[:n] - rv).max()
    assert not np.array_equal(tab_pi, ref_pi)                       # (a different rounding, not a different function)
    print("conv2 table vs GEMM sets: max |dpi|", np.abs(tab_pi - ref_pi).max(), "max |dv|", np.abs(tab_v - ref_v).max(),
          "| vs torch: table", np.abs(tab_pi[:n] - rpi).max(), np.abs(tab_v[:n] - rv).max(), "GEMM", np.abs(ref_pi[:n] - rpi).max(), np.abs(ref_v[:n] - rv).max())


def test_init_random_and_checkpoint_roundtrip(conv_engine, oracle, tmp_path):
    conv_engine.net_init_random(4, seed=11)
    p = conv_engine.net_get_params(4)
    off, total = layout(C)
    o, shp = off["conv2_w"]
    lim = np.sqrt(6.0 / (9 * C + 9 * C))
    w = p[o:o + int(np.prod(shp))]
    assert np.abs(w).max() <= lim and np.abs(w).max() > 0.9 * lim and abs(w.mean()) < 1e-3   # Glorot-uniform
    o, shp = off["conv2_bn"]
    assert np.all(p[o:o + C] == 1) and np.all(p[o + C:o + 3 * C] == 0) and np.all(p[o + 3 * C:o + 4 * C] == 1)
    path = os.path.join(tmp_path, "4.aznet")
    conv_engine.net_save(4, path)
    conv_engine.net_load(5, path)
    states = random_states(oracle, 32, seed=1)
    a = conv_engine.predict_states(states, 4)
    b = conv_engine.predict_states(states, 5)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    rpi, rv = forward_ref(p, np.stack([oracle.c4_features(int(m), int(t)) for m, t in states]), C)
    assert np.abs(a[0] - rpi).max() <= 2e-3 and np.abs(a[1] - rv).max() <= 6e-3


def test_replay_parity_selfplay(conv_engine, oracle):
    """Visit counts are discontinuous in (pi, v), so a bf16 net can never match an f32 oracle move for move.
    Replay parity: the engine records every NNet::predict row it consumed; the oracle re-runs the episodes
    feeding those rows back.  Tree logic identical => same states requested in the same order (checked per
    row), same moves, pi and z bit for bit."""
    conv_engine.net_init_random(6, seed=3)
    n, sims, cap = 48, 50, 42 * 51 + 8
    got = conv_engine.selfplay(n_games=n, num_sims=sims, model_id=6, seed=17, record_evals=cap)
    cnt, states, pis, vs = conv_engine.selfplay_get_evals(n, cap)
    assert cnt.max() <= cap
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(cnt)
    fs = np.concatenate([states[g, :cnt[g]] for g in range(n)])
    fp = np.concatenate([pis[g, :cnt[g]] for g in range(n)])
    fv = np.concatenate([vs[g, :cnt[g]] for g in range(n)])
    ref = oracle.selfplay(n, sims, net_kind=oracle.NET_REPLAY, seed=17, threads=8, replay=(off, fs, fp, fv))
    assert not ref["replay_bad"].any()
    assert np.array_equal(got["moves"], ref["moves"]) and np.array_equal(got["game_len"], ref["game_len"])
    assert np.array_equal(got["pis"], ref["pis"]) and np.array_equal(got["zs"], ref["zs"])
    assert np.array_equal(got["boards"].reshape(-1, 84), ref["boards"].reshape(-1, 84))
    # and the recorded rows are what the net returns for those states (the log is the real net output)
    pi2, v2 = conv_engine.predict_states(fs[:256], 6)
    assert np.array_equal(pi2, fp[:256]) and np.array_equal(v2, fv[:256])


def test_replay_parity_get_action_prob(conv_engine, oracle):
    conv_engine.net_init_random(7, seed=9)
    sims, G = 100, 12
    tb = conv_engine.tree_create(G, reserve=oracle.default_reserve(sims), num_sims=sims, max_depth=1000, model_id=7, cpuct=1)
    tb.record_evals(8 * (sims + 1))
    states = [(0, 0)] * G
    rng = np.random.default_rng(2)
    hist = []
    for move in range(6):
        pi, counts, q = tb.get_action_prob(np.array(states, dtype=np.uint64), 1.0, seed=4)
        hist.append((list(states), pi, counts, q))
        states = [oracle.c4_play(s[0], s[1], int(rng.choice([a for a in range(7) if pi[g][a] > 0])))
                  for g, s in enumerate(states)]
    cnt, lstates, lpis, lvs = tb.get_evals()
    for g in range(G):
        # one oracle tree per game, replaying that game's rows through the same six calls
        off = np.array([0, cnt[g]], np.int64)
        import ctypes
        L = oracle.lib()
        t = oracle.Tree(sims, net_kind=oracle.NET_REPLAY)
        # feed the replay buffers through the selfplay-independent path: a ReplayNet lives inside the tree box
        L.azo_tree_set_replay.restype = None
        L.azo_tree_set_replay.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint64]
        fs = np.ascontiguousarray(lstates[g, :cnt[g]]); fp = np.ascontiguousarray(lpis[g, :cnt[g]]); fv = np.ascontiguousarray(lvs[g, :cnt[g]])
        L.azo_tree_set_replay(t._h, fs.ctypes.data, fp.ctypes.data, fv.ctypes.data, int(cnt[g]))
        for (sts, pi, counts, q) in hist:
            opi, ocnt, oq = t.get_action_prob(sts[g][0], sts[g][1], 1.0, seed=4, game_id=g)
            assert np.array_equal(counts[g], ocnt)
            assert np.abs(pi[g] - opi).max() <= 1e-5 and np.abs(q[g] - oq).max() <= 1e-5   # north_star tolerance; in fact equal
            assert np.array_equal(pi[g], opi) and np.array_equal(q[g], oq)


def test_dedup_is_bit_exact_with_the_conv_net(conv_engine, oracle):
    """Leaf de-duplication + the evaluation cache on the real bf16 net: the same episodes with the switch off and on give
    identical moves, pi and z, while the net runs on fewer rows (games from the empty board share their openings)."""
    conv_engine.net_init_random(8, seed=13)
    n, sims = 384, 40
    runs = {}
    try:
        for mode in (0, 1):
            conv_engine.set_option("eval_dedup", mode)
            conv_engine.reset_stats()
            runs[mode] = (conv_engine.selfplay(n_games=n, num_sims=sims, model_id=8, seed=5, concurrent=128), conv_engine.stats())
    finally:
        conv_engine.set_option("eval_dedup", 1)
    (a, sa), (b, sb) = runs[0], runs[1]
    for k in ("moves", "game_len", "pis", "zs", "states"):
        assert np.array_equal(a[k], b[k]), k
    assert sa["leaf_rows_executed"] == sa["leaf_rows_requested"] == sa["leaf_evals"] == sb["leaf_evals"]
    assert sb["leaf_rows_requested"] == sb["leaf_evals"] == sb["leaf_rows_executed"] + sb["eval_cache_hits"] + sb["eval_batch_dups"]
    assert sb["leaf_rows_executed"] < 0.9 * sb["leaf_rows_requested"], sb


def test_model_slots_do_not_leak(engine_mod):
    """A long Coach::learn run trains into a new model id per accepted iteration (src/coach.rs:296-390).  The activation
    workspace belongs to the stream, not to the model, and az_net_free drops a superseded id: device memory stays flat."""
    import torch
    e = engine_mod.Engine(device=0, max_batch=2048, net_channels=C)
    try:
        e.net_init_random(0, seed=1)
        s = np.zeros((64, 2), np.uint64)
        e.predict_states(s, 0)                              # creates the workspace
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        for k in range(1, 13):
            e.net_init_random(k, seed=k)
            e.predict_states(s, k)
            e.net_free(k - 1)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info(0)[0]
        assert abs(free0 - free1) < 64 << 20, (free0, free1)       # one model is 41 MB; 12 leaked workspaces would be 3.6 GB
        with pytest.raises(engine_mod.AzError):
            e.predict_states(s, 3)                          # a freed id is gone
        pi, v = e.predict_states(s, 12)
        assert np.isfinite(pi).all()
    finally:
        e.close()


def test_bench_scale_replay_parity_with_the_conv_net(engine, oracle):
    """The bench configuration itself -- 8192 concurrent games, 100 sims/move, the bf16 C=512 net with conv1 / conv2 as table
    lookups, leaf de-duplication and the evaluation cache on -- held to the oracle: every game's NNet::predict rows are
    recorded and ALL 8192 games are re-played on the oracle from their own records (replay parity: the oracle asks
    for the same states in the same order and gets the engine's rows back), and must match move for move, pi for pi, z for z.
    Game g depends only on (seed, g): what its 8191 neighbours do -- sharing its leaf rows through the election table and the
    cache -- must not show."""
    engine.net_init_random(21, seed=5)
    n, sims, seed = 8192, 100, 33
    cap = 42 * (sims + 1) + 8
    engine.reset_stats()
    got = engine.selfplay(n_games=n, num_sims=sims, model_id=21, seed=seed, want_boards=False, record_evals=cap)
    st = engine.stats()
    assert st["leaf_rows_executed"] < 0.8 * st["leaf_rows_requested"]            # the sharing really happened
    cnt, states, pis, vs = engine.selfplay_get_evals(n, cap)
    assert cnt.max() <= cap and cnt.min() > 0
    _replay_every_episode(oracle, got, (cnt, states, pis, vs), sims, seed, n)
    # and the recorded rows are what NNet::predict returns for those states
    g = 17
    pi2, v2 = engine.predict_states(states[g, :256], 21)
    assert np.array_equal(pi2, pis[g, :256]) and np.array_equal(v2, vs[g, :256])


def test_shipped_kernel_choices_are_bit_identical(engine, oracle):
    """The shipped library's own kernel choices -- conv3 with / without the half-tile tail (k_conv3_auto cuts the tiles of a short last
    round in two along the channels), conv3 / conv4 / fc1 / fc2 as the register-fed skinny GEMM (k_gemm_skinny; forced at every size here), conv3 of a
    small batch on the ring, conv3's LDS image in the bank-conflict-free PLANES layout or with its rows in order -- accumulate every output's K terms in the same order: identical bits at row counts that hit every role
    (no cut, a cut last round, a cut only round, ragged tiles, one row), for both conv2 kernel sets."""
    engine.net_init_random(26, seed=17)
    states = random_states(oracle, 8192, seed=123)
    try:
        for table in (1, 0):
            engine.set_option("conv2_table", table)
            for n in (8192, 5003, 3100, 2304, 1537, 771, 357, 300, 129, 64, 33, 13, 1):
                for key in ("conv3_tail", "narrow_rows", "conv3_small", "conv3_planes", "ring_packed"):
                    engine.set_option(key, 0)
                ref = engine.predict_states(states[:n], 26)
                for t3, nr, s3, pl, rp in ((1, 0, 0, 0, 0), (0, 8192, 0, 0, 0), (1, 32, 1, 0, 1), (0, 100, 1, 1, 0), (1, 16, 0, 1, 1), (0, 0, 0, 1, 1),
                                           (1, 0, 0, 1, 0), (0, 0, 1, 0, 1)):
                    engine.set_option("conv3_tail", t3)
                    engine.set_option("narrow_rows", nr)
                    engine.set_option("conv3_small", s3)
                    engine.set_option("conv3_planes", pl)
                    engine.set_option("ring_packed", rp)     # the ring's weight stages from the packed copy (conv3 small, conv4, fc1, fc2)
                    got = engine.predict_states(states[:n], 26)
                    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (table, n, t3, nr, s3, pl, rp)
    finally:
        for key, val in (("conv2_table", 1), ("conv3_tail", 1), ("narrow_rows", 32), ("conv3_small", 1), ("conv3_planes", 1), ("ring_packed", 1)):
            engine.set_option(key, val)


def _flatten_log(cnt, states, pis, vs, ids):
    """Per-game eval logs [G,cap,..] -> the oracle's replay stream for the games `ids` (offsets + flattened rows)."""
    off = np.zeros(len(ids) + 1, np.int64)
    off[1:] = np.cumsum([cnt[g] for g in ids])
    cat = lambda a: np.ascontiguousarray(np.concatenate([a[g, :cnt[g]] for g in ids]))
    return off, cat(states), cat(pis), cat(vs)


def _replay_every_episode(oracle, got, logs, sims, seed, n, sim_threads=1, chunk=1024):
    """EVERY episode of a recorded self-play call re-played on the oracle from its own recorded rows (replay parity), `chunk`
    episodes per oracle call on 16 threads: moves, pi and z bit for bit.  (Round 3 replayed 24 ids; the one product parity
    bug of three rounds had sat behind a 16-of-4096 sample.)"""
    cnt, states, pis, vs = logs
    offs = np.concatenate([[0], np.cumsum(2 * got["game_len"].astype(np.int64))])
    for lo in range(0, n, chunk):
        ids = list(range(lo, min(n, lo + chunk)))
        ref = oracle.selfplay(len(ids), sims, net_kind=oracle.NET_REPLAY, seed=seed, first_game_id=lo, threads=16, sim_threads=sim_threads,
                              replay=_flatten_log(cnt, states, pis, vs, ids))
        assert not ref["replay_bad"].any(), (lo, np.flatnonzero(ref["replay_bad"])[:5] + lo)
        assert np.array_equal(ref["game_len"], got["game_len"][lo:lo + len(ids)]), lo
        assert np.array_equal(ref["moves"], got["moves"][lo:lo + len(ids)]), lo
        a, b = offs[lo], offs[lo + len(ids)]
        assert np.array_equal(ref["pis"], got["pis"][a:b]) and np.array_equal(ref["zs"], got["zs"][a:b]), lo


def _check_move_record(oracle, game_len, moves, results, start=(0, 0)):
    """az_arena_get_moves: every recorded game, played move by move with the oracle's rules, ends exactly at its recorded length with
    the recorded result (+1 first seat won, -1 second, 0 draw: src/arena.rs:51)."""
    for g in range(len(results)):
        s, player = start, 1
        for k in range(int(game_len[g])):
            assert oracle.c4_ended(*s) == 0.0, (g, k)
            s = oracle.c4_play(s[0], s[1], int(moves[g, k]))
            player = -player
        e = oracle.c4_ended(*s)                        # for the side to move: -1 = the player who just moved won
        assert e != 0.0, g
        want = -player if e == -1.0 else (player if e == 1.0 else 0)
        assert int(results[g]) == want, (g, e, player)


def test_replay_parity_arena_with_two_conv_nets(engine, oracle):
    """BASELINE config 3 AS BENCHMARKED -- az_arena with two bf16 conv nets, the old model's searches on a second stream, the
    model-tagged evaluation cache shared by both, the small-batch conv3 path and the per-ply batch feedback -- held to the
    oracle by replay parity: az_arena records every NNet::predict row each tree consumed (per game, per player); the oracle
    re-plays the games feeding those rows back.  Same states requested in the same order, same results, same W/L/D.
    First a 64-game, 100-sim arena in full, then ALL 4096 games of the 400-sim arena itself (512 games per oracle call)."""
    engine.net_init_random(22, seed=5)
    engine.net_init_random(23, seed=6)
    for num, sims, picks in ((64, 100, None), (4096, 400, 512)):
        cap = 22 * (sims + 1) + 8                                    # a player moves at most 21 times
        engine.reset_stats()
        wld, res = engine.arena(num, sims, new_model_id=23, old_model_id=22, seed=9, record_evals=cap)
        st = engine.stats()
        assert int(wld.sum()) == num and st["leaf_rows_executed"] < st["leaf_rows_requested"]
        logs = [engine.arena_get_evals(w, num, cap) for w in (0, 1)]
        assert max(int(l[0].max()) for l in logs) <= cap
        if picks is None:
            groups = [list(range(num))]                              # the whole small arena in one oracle call
        else:
            groups = [list(range(lo, min(num, lo + picks))) for lo in range(0, num, picks)]      # every game, `picks` per oracle call
        tally = np.zeros(3, np.uint64)
        for sel in groups:
            rn, ro = (_flatten_log(*logs[w], sel) for w in (0, 1))
            owld, ores, bad = oracle.arena_ex(num, sims, first_game=sel[0], n_games=len(sel), net_kind=oracle.NET_REPLAY, seed=9,
                                              threads=16, replay_new=rn, replay_old=ro)
            assert not bad.any(), (num, sel[0], np.flatnonzero(bad)[:5] + sel[0])
            assert np.array_equal(ores, res[sel[0]:sel[0] + len(sel)]), (num, sel[0])
            tally += owld
        assert tally.tolist() == wld.tolist()
        glen, gmoves = engine.arena_get_moves(num)
        _check_move_record(oracle, glen, gmoves, res)
        ids = [g for sel in groups for g in sel]
        # the recorded rows are what NNet::predict returns for those states, under the right model
        for w, mid in ((0, 23), (1, 22)):
            cnt, states, pis, vs = logs[w]
            k = int(min(cnt[ids[0]], 128))
            pi2, v2 = engine.predict_states(states[ids[0], :k], mid)
            assert np.array_equal(pi2, pis[ids[0], :k]) and np.array_equal(v2, vs[ids[0], :k])


def test_replay_parity_selfplay_with_refill_and_the_conv_net(engine, oracle):
    """The bench's shape: more episodes than slots -- 16,384 episodes on 8,192 slots, C = 512, tables + de-duplication + the
    evaluation cache on, the drain of the last 8,192 on shrinking batches (ring / image-resident kernel switching) -- with
    per-EPISODE eval logs that survive the slot refills.  ALL 16,384 episodes (the last 8,192 drain) are replayed on the oracle
    from their own recorded rows: moves, pi and z bit for bit."""
    engine.net_init_random(24, seed=8)
    n, conc, sims, seed = 16384, 8192, 100, 44
    cap = 42 * (sims + 1) + 8
    engine.reset_stats()
    got = engine.selfplay(n_games=n, concurrent=conc, num_sims=sims, model_id=24, seed=seed, want_boards=False, record_evals=cap)
    st = engine.stats()
    assert st["games"] == n and st["leaf_rows_executed"] < 0.8 * st["leaf_rows_requested"]
    logs = engine.selfplay_get_evals(n, cap)
    assert logs[0].max() <= cap and logs[0].min() > 0
    _replay_every_episode(oracle, got, logs, sims, seed, n)


def test_free_running_selfplay_plays_the_same_games(engine, oracle):
    """az_set_option "selfplay_async" = 1: every slot on its own timeline (backup / move / next root / select inside k_async_step,
    several tree launches per leaf batch, moves and refills inside the kernel) -- 4,096 episodes on 1,024 slots with the bf16 net, tables,
    de-duplication and the cache on, for several (launches, stages) settings: the same moves, pi and z as the lock-step driver tuple for
    tuple, and EVERY episode replays on the oracle from its own recorded rows.  The schedule decides when a row is evaluated, never what
    it is."""
    engine.net_init_random(29, seed=14)
    n, conc, sims, seed = 4096, 1024, 100, 52
    cap = 42 * (sims + 1) + 8
    ref = engine.selfplay(n_games=n, concurrent=conc, num_sims=sims, model_id=29, seed=seed, want_boards=False)
    try:
        engine.set_option("selfplay_async", 1)
        for launches, iters in ((2, 6), (1, 2), (3, 16)):
            engine.set_option("selfplay_async_launches", launches)
            engine.set_option("selfplay_async_iters", iters)
            engine.reset_stats()
            got = engine.selfplay(n_games=n, concurrent=conc, num_sims=sims, model_id=29, seed=seed, want_boards=False,
                                  record_evals=cap if launches == 2 else 0)
            st = engine.stats()
            assert st["games"] == n and st["simulations"] == sims * int(got["game_len"].sum())
            for k in ("game_len", "moves", "states", "pis", "zs"):
                assert np.array_equal(got[k], ref[k]), (launches, iters, k)
            if launches == 2:
                _replay_every_episode(oracle, got, engine.selfplay_get_evals(n, cap), sims, seed, n)
    finally:
        engine.set_option("selfplay_async", 0)
        engine.set_option("selfplay_async_launches", 2)
        engine.set_option("selfplay_async_iters", 6)


def test_selfplay_session_with_the_conv_net(engine, oracle):
    """The session form (az_selfplay_begin / _next / _end) with the bf16 net, tables, de-duplication and the cache on: 3,072 episodes on
    1,024 slots fetched as three chunks are tuple for tuple the one-shot call's, with the lock-step and the free-running driver, and every
    episode of the chunked run replays on the oracle from its own recorded rows (the log is the session's, per episode)."""
    engine.net_init_random(31, seed=16)
    n, conc, sims, seed = 3072, 1024, 100, 77
    cap = 42 * (sims + 1) + 8
    ref = engine.selfplay(n_games=n, concurrent=conc, num_sims=sims, model_id=31, seed=seed, want_boards=False)
    try:
        for mode in (0, 1):
            engine.set_option("selfplay_async", mode)
            engine.reset_stats()
            engine.selfplay_begin(n, sims, 31, seed=seed, concurrent=conc, record_evals=cap if mode == 0 else 0)
            parts = [engine.selfplay_next(k, want_boards=False) for k in (1024, 1000, 1048)]
            evals = engine.selfplay_get_evals(n, cap) if mode == 0 else None
            engine.selfplay_end()
            st = engine.stats()
            got = {k: np.concatenate([q[k] for q in parts]) for k in ("game_len", "moves", "states", "pis", "zs")}
            got["count"] = sum(q["count"] for q in parts)
            assert st["games"] == n and st["simulations"] == sims * int(got["game_len"].sum())
            for k in ("game_len", "moves", "states", "pis", "zs"):
                assert np.array_equal(got[k], ref[k]), (mode, k)
            if evals is not None:
                _replay_every_episode(oracle, got, evals, sims, seed, n)
    finally:
        engine.set_option("selfplay_async", 0)


def test_replay_parity_lockstep_threads_with_the_conv_net(engine, oracle):
    """Several simulations in flight per tree (num_sim_threads = 4: the lock-step schedule of DESIGN.md 4.1a) with the real bf16 net,
    tables, de-duplication and the evaluation cache on, 2,048 slots x 4 threads = up to 8,192 rows per step, with slot refill: the
    eval log holds every tree's rows in the order its threads consumed them, and the oracle's lock-step search re-plays ALL 4,096
    episodes from them -- moves, pi and z bit for bit, no simulation spent differently."""
    engine.net_init_random(27, seed=12)
    n, conc, sims, T, seed = 4096, 2048, 100, 4, 21
    cap = 42 * (sims + 1) + 8
    engine.reset_stats()
    got = engine.selfplay(n_games=n, concurrent=conc, num_sims=sims, model_id=27, seed=seed, want_boards=False, record_evals=cap,
                          num_sim_threads=T)
    st = engine.stats()
    assert st["games"] == n and st["simulations"] == sims * int(got["game_len"].sum())
    _replay_every_episode(oracle, got, engine.selfplay_get_evals(n, cap), sims, seed, n, sim_threads=T)


def test_conv3_image_kernel_accounting(engine, oracle):
    """az_stats.net_conv3_image_rows / _launches: what the image-resident conv3 kernel really processed, counted on the device -- a big
    batch is one working launch with all its rows, a small batch (the skinny / ring kernels take it) none."""
    engine.net_init_random(28, seed=3)
    states = random_states(oracle, 4000, seed=77)
    engine.reset_stats()
    engine.predict_states(states, 28)
    st = engine.stats()
    assert (st["net_conv3_image_rows"], st["net_conv3_image_launches"]) == (4000, 1)
    engine.predict_states(states[:13], 28)
    engine.predict_states(states[:3000], 28)
    st = engine.stats()
    assert (st["net_conv3_image_rows"], st["net_conv3_image_launches"]) == (7000, 2)
    engine.reset_stats()
    assert engine.stats()["net_conv3_image_rows"] == 0


def test_set_option_is_per_engine(engine_mod, oracle):
    """az_set_option changes the handle it is given and nothing else: two engines in one process, one with conv2 as the MFMA
    GEMM ("conv2_table" = 0: a different rounding of the same function) and the older kernel families, interleaved
    predict_states calls -- each engine's rows equal its own single-engine result bit for bit."""
    a = engine_mod.Engine(device=0, max_batch=512, net_channels=128)
    b = engine_mod.Engine(device=0, max_batch=512, net_channels=128)
    try:
        st = random_states(oracle, 300, seed=9)
        for e in (a, b):
            e.net_init_random(0, seed=5)
        ref_a = a.predict_states(st, 0)
        b.set_option("conv2_table", 0)
        ref_b = b.predict_states(st, 0)
        assert not np.array_equal(ref_a[0], ref_b[0])               # the two kernel sets round differently
        for k in range(3):
            ga, gb = a.predict_states(st, 0), b.predict_states(st, 0)
            assert np.array_equal(ga[0], ref_a[0]) and np.array_equal(ga[1], ref_a[1]), k
            assert np.array_equal(gb[0], ref_b[0]) and np.array_equal(gb[1], ref_b[1]), k
            b.set_option("conv3_small", k % 2)                      # a bit-identical family switch: b keeps its own results, a never notices
        # the tree options too
        a.net_set_kind(1, engine_mod.NET_HASH, 3)
        b.net_set_kind(1, engine_mod.NET_HASH, 3)
        b.set_option("fused_search", 0)
        b.set_option("tree_block4", 0)
        ra = a.selfplay(n_games=32, num_sims=25, model_id=1, seed=2)
        rb = b.selfplay(n_games=32, num_sims=25, model_id=1, seed=2)
        assert np.array_equal(ra["moves"], rb["moves"]) and np.array_equal(ra["pis"], rb["pis"])
        # the diagnostic library keeps its switches per engine too (one of them on a superseded kernel family)
        c = engine_mod.Engine(device=0, max_batch=512, net_channels=128, diag=True)
        d = engine_mod.Engine(device=0, max_batch=512, net_channels=128, diag=True)
        try:
            for e in (c, d):
                e.net_init_random(0, seed=5)
            d.set_option("conv2_table", 0)
            d.set_option("gemm_variant", 0)
            for k in range(2):
                gc, gd = c.predict_states(st, 0), d.predict_states(st, 0)
                assert np.array_equal(gc[0], ref_a[0]) and np.array_equal(gd[0], ref_b[0]) and np.array_equal(gd[1], ref_b[1]), k
        finally:
            c.close()
            d.close()
    finally:
        a.close()
        b.close()


def test_shipped_library_refuses_diagnostic_options(engine_mod):
    """The timing ablations that compute WRONG results and the clock-stamp builds live in libaz_engine_diag.so (tools/ only):
    the shipped library does not contain them and refuses their option values."""
    e = engine_mod.Engine(device=0, max_batch=64, net_channels=128)
    try:
        for key, val in (("gemm_variant", 12), ("gemm_variant", 17), ("conv3_pipe", 11), ("conv3_pipe", 15), ("conv3_pipe", 3),
                         ("tree_stamps", 1), ("print_seg_stamps", 0)):
            with pytest.raises(engine_mod.AzError) as ei:
                e.set_option(key, val)
            assert ei.value.status == 1, (key, val)
    finally:
        e.close()


def test_evaluation_cache_is_sized_from_the_call(engine_mod):
    """A 1-tree, 25-sim call must not allocate the bench's 43 GB evaluation cache or clear a gigabyte per call: the cache
    is sized from what the call can insert (trees x (sims + 1) x calls, x 4 for the load factor) up to "eval_cache_log2"."""
    import torch
    e = engine_mod.Engine(device=0, max_batch=64, net_channels=128)
    try:
        e.net_init_random(0, seed=1)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        tb = e.tree_create(1, reserve=10000, num_sims=25, max_depth=1000, model_id=0, cpuct=1)
        tb.get_action_prob(np.zeros((1, 2), np.uint64), 1.0)
        r = e.selfplay(n_games=4, num_sims=25, model_id=0, seed=1)
        torch.cuda.synchronize()
        used = free0 - torch.cuda.mem_get_info(0)[0]
        assert used < 512 << 20, used                              # trees + workspace + a small cache, not 43 GB
        assert r["count"] > 0
    finally:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [256, 384])
def test_other_channel_widths(oracle, channels):
    """net_channels other than the two the suite lives on (3 column tiles of conv3 at 384; 4 / 6 table slices): one ragged call equals
    the same rows in chunks, the table gather's two kernels agree bit for bit, the GEMM set stays within the table set's distance, and
    the older kernels reproduce the GEMM set bit for bit."""
    from alphazero_rs_amd import engine as azeng
    e = azeng.Engine(device=0, max_batch=2048, net_channels=channels, diag=True)
    shipped = azeng.Engine(device=0, max_batch=2048, net_channels=channels)
    try:
        e.net_init_random(0, seed=3)
        shipped.net_init_random(0, seed=3)
        st = random_states(oracle, 1201, seed=5)
        sp = shipped.predict_states(st, 0)                          # the shipped library at this width == the diagnostic one's default
        dp = e.predict_states(st, 0)
        assert np.array_equal(sp[0], dp[0]) and np.array_equal(sp[1], dp[1])
        shipped.set_option("conv2_table", 0)
        e.set_option("conv2_table", 0)
        sp, dp = shipped.predict_states(st, 0), e.predict_states(st, 0)
        assert np.array_equal(sp[0], dp[0]) and np.array_equal(sp[1], dp[1])
        e.set_option("conv2_table", 1)
        shipped.close()
        ref = e.predict_states(st, 0)
        parts = [e.predict_states(st[o:o + 250], 0) for o in range(0, 1201, 250)]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), ref[0]) and np.array_equal(np.concatenate([p[1] for p in parts]), ref[1])
        e.set_option("conv2_table", 2)
        a = e.predict_states(st, 0)
        assert np.array_equal(a[0], ref[0]) and np.array_equal(a[1], ref[1])
        e.set_option("conv2_table", 0)
        g = e.predict_states(st, 0)
        assert np.abs(g[0] - ref[0]).max() <= 2e-4 and np.abs(g[1] - ref[1]).max() <= 1e-3
        for key, val in (("conv3_pipe", 0), ("fc_ring", 0), ("gemm_variant", 0)):
            e.set_option(key, val)
            b = e.predict_states(st, 0)
            assert np.array_equal(b[0], g[0]) and np.array_equal(b[1], g[1]), (key, val)
    finally:
        for key, val in (("gemm_variant", 5), ("fc_ring", 1), ("conv3_pipe", 1), ("conv2_table", 1)):
            e.set_option(key, val)
        e.close()
