"""CPU tests of the next-tier host logic: the trainer (NNet::train recipe) and the Coach::learn loop
(replay window, examples files, resume, arena gate) driven by a fake engine that is backed by the oracle."""
import os

import numpy as np
import pytest
import torch

from net_ref import forward_ref, random_params


@pytest.fixture(scope="module")
def mods(engine_mod):
    from alphazero_rs_amd import coach, trainer
    return coach, trainer


def test_trainer_forward_matches_reference(mods):
    coach, trainer = mods
    C = 16
    p = random_params(C, 3)
    boards = (np.random.default_rng(0).random((9, 2, 6, 7)) > 0.7).astype(np.float32)
    net = trainer.PolicyValueNet(C, p, torch.device("cpu"))
    net.eval()
    with torch.no_grad():
        logits, v = net(torch.from_numpy(boards))
    rpi, rv = forward_ref(p, boards, C, emulate_bf16=False)
    assert np.abs(torch.softmax(logits, 1).numpy() - rpi).max() < 1e-5 and np.abs(v.numpy() - rv).max() < 1e-5
    assert np.array_equal(net.flat_params(), p)                       # layout round trip
    assert trainer.layout(512)[1] == 10779144


def test_trainer_reduces_loss_and_updates_bn(mods):
    coach, trainer = mods
    C = 16
    g = np.random.default_rng(1)
    boards = (g.random((256, 2, 6, 7)) > 0.6).astype(np.float32)
    pis = g.random((256, 7)).astype(np.float32)
    pis /= pis.sum(1, keepdims=True)
    vs = np.tanh(boards.reshape(256, -1)[:, :5].sum(1) - 2).astype(np.float32)
    p0 = random_params(C, 2, bn_random=False)
    tr = trainer.Trainer(channels=C, epochs=6, batch_size=32, device=torch.device("cpu"))
    p1 = tr.train(p0, boards, pis, vs, seed=5)
    # training-mode loss falls epoch over epoch (the eval-mode loss lags: BN moving statistics move with momentum
    # 0.99, the tf.layers default, so after a few dozen steps they are still near their initial values)
    assert len(tr.history) == 6 and sum(tr.history[-1]) < sum(tr.history[0]) and tr.history[-1][1] < tr.history[0][1]
    assert np.isfinite(sum(tr.evaluate(p1, boards, pis, vs)))
    off, _ = trainer.layout(C)
    o, _shape = off["conv2_bn"]
    assert not np.array_equal(p1[o + 2 * C:o + 3 * C], p0[o + 2 * C:o + 3 * C])      # moving mean moved
    assert np.array_equal(tr.train(p0, boards, pis, vs, seed=5), p1)                 # deterministic given the seed


class FakeEngine:
    """Stands in for alphazero_rs_amd.engine.Engine on a machine without a GPU: self-play and arena come from the
    oracle with the hash fixture net (salt = a checksum of the model's parameters, so a retrained model plays
    differently)."""

    def __init__(self, oracle, C):
        self.oracle, self.C, self.params = oracle, C, {}
        self.calls = []
        self.freed = {}

    def net_free(self, model_id):
        self.calls.append(("free", model_id))
        self.freed[model_id] = self.params.pop(model_id)

    def net_load(self, model_id, path):
        self.calls.append(("load", model_id))
        self.params[model_id] = np.fromfile(path, np.float32)

    def _salt(self, model_id):
        p = self.params[model_id] if model_id in self.params else self.freed[model_id]
        return int(np.abs(p).sum() * 1e3) % (1 << 30)

    def net_get_params(self, model_id):
        return self.params[model_id].copy()

    def net_set_params(self, model_id, p):
        self.params[model_id] = np.asarray(p, np.float32).copy()

    def net_save(self, model_id, path):
        self.params[model_id].tofile(path)

    def selfplay(self, n_games, num_sims, model_id, seed, first_game_id, concurrent, temp_threshold, max_depth, cpuct,
                 reserve, symmetries, want_boards, num_sim_threads=1):
        self.calls.append(("selfplay", n_games, first_game_id, model_id))
        r = self.oracle.selfplay(n_games, num_sims, net_kind=self.oracle.NET_HASH, salt=self._salt(model_id), seed=seed,
                                 first_game_id=first_game_id, temp_threshold=temp_threshold, threads=4, sim_threads=num_sim_threads)
        from alphazero_rs_amd import dist as azdist
        b = r["boards"][0::2]
        states = np.zeros((b.shape[0], 2), np.uint64)
        for rr in range(6):
            for c in range(7):
                bit = np.uint64(1) << np.uint64(c * 7 + (5 - rr))
                states[:, 0] |= np.where(b[:, 0, rr, c] != 0, bit, np.uint64(0))
                states[:, 1] |= np.where(b[:, 1, rr, c] != 0, bit, np.uint64(0))
        return {"states": states, "pis": r["pis"][0::2], "zs": r["zs"][0::2], "count": b.shape[0], "ref": r}

    def arena(self, num_games, num_sims, new_model_id, old_model_id, seed, max_depth, cpuct, reserve, first_game=0, total_games=0,
              num_sim_threads=1):
        self.calls.append(("arena", num_games, new_model_id, old_model_id))
        total = total_games or 2 * (num_games // 2)
        wld, res, _ = self.oracle.arena_ex(total, num_sims, net_kind=self.oracle.NET_HASH, salt=self._salt(new_model_id) ^ self._salt(old_model_id),
                                           seed=seed, new_model_id=1, old_model_id=0, threads=4, sim_threads=num_sim_threads)
        if not total_games:
            return wld, res
        part = res[first_game:first_game + num_games]          # the shard's games of the full arena
        w = l = d = 0
        for g, r in zip(range(first_game, first_game + num_games), part):
            win = 1 if g < total // 2 else -1
            w, l, d = w + (r == win), l + (r == -win), d + (r == 0)
        return np.array([w, l, d], np.uint64), part


def make_coach(coach, trainer, oracle, tmp, iters, C=16, history=2, queue=10000):
    eng = FakeEngine(oracle, C)
    eng.net_set_params(0, random_params(C, 9, bn_random=False))
    tr = trainer.Trainer(channels=C, epochs=1, batch_size=16, device=torch.device("cpu"))
    msgs = []
    c = coach.Coach.setup(eng, tmp, 1000000, 0.6, 15, history, queue, 1, 64, 6, iters, 5, 10, 1, 1000, 1, trainer=tr,
                          log=msgs.append)
    return c, eng, msgs


def test_coach_learn_loop(mods, oracle, tmp_path):
    coach, trainer = mods
    c, eng, msgs = make_coach(coach, trainer, oracle, str(tmp_path), iters=3)
    rep = c.learn(seed=4)
    assert [r["iteration"] for r in rep] == [0, 1, 2]
    # self-play: num_eps episodes per iteration with global ids iteration*num_eps.., symmetries regenerated (x2)
    sp = [x for x in eng.calls if x[0] == "selfplay"]
    assert [(x[1], x[2]) for x in sp] == [(5, 0), (5, 5), (5, 10)]
    ref0 = oracle.selfplay(5, 10, net_kind=oracle.NET_HASH, salt=eng._salt(0) if rep[0]["model_id"] == 0 else 0, seed=4, threads=4)
    # history window: at most max_history_length (2) iterations are kept and trained on
    assert len(c.history) == 2
    assert rep[2]["samples"] == sum(h[2].shape[0] for h in c.history)
    # gate arithmetic (src/coach.rs:383-390) and the log line (:381)
    for r in rep:
        tot = r["nwins"] + r["pwins"]
        assert r["accepted"] == (tot > 0 and r["nwins"] / tot >= 0.6)
        assert r["nwins"] + r["pwins"] + r["draws"] == 6
    assert any(m.startswith("NEW/PREV WINS : ") for m in msgs)
    # model ids advance only on acceptance
    assert c.model_id == sum(r["accepted"] for r in rep)
    # files: <iter>.examples per iteration, <id>.aznet per candidate
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".examples")) == ["0.examples", "1.examples", "2.examples"]
    z = coach.load_examples(os.path.join(tmp_path, "2.examples"))
    assert [h[2].shape[0] for h in z] == [h[2].shape[0] for h in c.history] and z[0][0].shape[1:] == (2, 6, 7)
    for got, want in zip(z, c.history):
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
    with open(os.path.join(tmp_path, "2.examples"), "rb") as f:
        assert f.read(8) == b"AZEX0001"
    # the stored tuples are the oracle's execute_episode tuples incl. the mirrored twins
    first_iter = coach.load_examples(os.path.join(tmp_path, "0.examples"))
    if rep[0]["model_id"] == 0:
        assert len(first_iter) == 1
        assert np.array_equal(first_iter[0][0], ref0["boards"]) and np.array_equal(first_iter[0][1], ref0["pis"])
        assert np.array_equal(first_iter[0][2], ref0["zs"])


def test_coach_resume_and_queue_limit(mods, oracle, tmp_path):
    coach, trainer = mods
    c, eng, _ = make_coach(coach, trainer, oracle, str(tmp_path), iters=1, queue=40)
    c.learn(seed=1)
    assert c.history[0][2].shape[0] == 40                      # only the newest max_queue_length samples are kept
    open(os.path.join(tmp_path, "notes.txt"), "w").write("ignored")       # the reference would panic on this (A14)
    c2, eng2, _ = make_coach(coach, trainer, oracle, str(tmp_path), iters=1, queue=40)
    assert c2.start_iteration == 1 and len(c2.history) == 1 and c2.history[0][2].shape[0] == 40
    rep = c2.learn(seed=1)
    assert rep[0]["iteration"] == 1 and os.path.exists(os.path.join(tmp_path, "1.examples"))


def test_coach_state_file_skip_first_play_and_model_slots(mods, oracle, tmp_path):
    """(1) skip_first_play still pushes an (empty) history entry, as src/coach.rs:282 does, so the replay window ages exactly
    like the reference's; (2) coach.state names the live model after every gate and a restarted Coach loads it; (3) the model
    slot nobody will read again (the old net after an accept, the candidate after a reject) is dropped."""
    coach, trainer = mods
    c, eng, _ = make_coach(coach, trainer, oracle, str(tmp_path), iters=2, history=3)
    rep = c.learn(seed=4)
    assert coach.read_state(str(tmp_path)) == (1, c.model_id)
    assert os.path.exists(os.path.join(tmp_path, "0.aznet"))                       # the run's initial model
    frees = [x[1] for x in eng.calls if x[0] == "free"]
    assert frees == [r["model_id"] if r["accepted"] else r["model_id"] + 1 for r in rep]
    assert sorted(eng.params) == [c.model_id]                                      # exactly the live model is resident
    # restart: resumes at iteration 2 with the live model loaded from its checkpoint, first play skipped
    c2, eng2, _ = make_coach(coach, trainer, oracle, str(tmp_path), iters=2, history=3)
    assert c2.start_iteration == 2 and c2.model_id == c.model_id and ("load", c.model_id) in eng2.calls
    assert np.array_equal(eng2.params[c.model_id], eng.params[c.model_id])
    n_before = len(c2.history)
    rep2 = c2.learn(skip_first_play=True, seed=4)
    sp = [x for x in eng2.calls if x[0] == "selfplay"]
    assert len(sp) == 1 and sp[0][2] == 3 * 5                                      # only iteration 3 played (global ids 15..)
    z2 = coach.load_examples(os.path.join(tmp_path, "2.examples"))
    assert len(z2) == min(3, n_before + 1) and z2[-1][2].shape[0] == 0            # the skipped play left an empty entry
    assert rep2[0]["iteration"] == 2 and rep2[0]["model_id"] == c.model_id


def test_setup_contracts(mods, oracle, tmp_path):
    coach, trainer = mods
    eng = FakeEngine(oracle, 16)
    with pytest.raises(ValueError):
        coach.Coach.setup(eng, str(tmp_path), 1000, 0.6, 15, 2, 100, 3, 1, 4, 1, 1, 10, 1, 1000, 1)     # 10 % 3 != 0
    with pytest.raises(ValueError):
        coach.Coach.setup(eng, str(tmp_path), 1000, 0.6, 15, 2, 100, 1, 1, 4, 1, 1, 10, 3, 1000, 1)     # 10 % 3 sim threads != 0 (src/async_mcts.rs:192)
    c = coach.Coach.setup(eng, str(tmp_path), 1000, 0.6, 15, 2, 100, 1, 1, 4, 1, 1, 10, 2, 1000, 1)  # two simulations in flight per tree
    assert c.num_sim_threads == 2


def test_states_to_boards(mods, oracle):
    coach, _ = mods
    s = (0, 0)
    sts = []
    for a in (3, 3, 2, 4, 0, 6, 6):
        s = oracle.c4_play(s[0], s[1], a)
        sts.append(s)
    b = coach.states_to_boards(np.array(sts, dtype=np.uint64))
    for i, st in enumerate(sts):
        assert np.array_equal(b[i], oracle.c4_features(*st))


# ---- Coach::learn on two ranks (gloo): episodes shard by global game id, tuples all-gathered, gradients all-reduced ----
def _coach_rank(rank, world, port, tmp, q):
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from alphazero_rs_amd import coach, trainer
    from oracle import oracle_py as orc
    from test_coach_cpu import make_coach
    c, eng, _ = make_coach(coach, trainer, orc, os.path.join(tmp, f"rank{rank}"), iters=1)
    rep = c.learn(seed=4)
    q.put((rank, rep[0]["samples"], c.history[0][0].copy(), c.history[0][1].copy(), c.history[0][2].copy(), eng.params.get(1, eng.freed.get(1)).copy(),
           [x for x in eng.calls if x[0] == "selfplay"], (rep[0]["nwins"], rep[0]["pwins"], rep[0]["draws"]),
           [x for x in eng.calls if x[0] == "arena"]))
    dist.barrier()
    dist.destroy_process_group()


def test_coach_two_ranks_match_single_process(mods, oracle, tmp_path):
    import socket
    import torch.multiprocessing as mp
    coach, trainer = mods
    torch.set_num_threads(1)
    c, eng, _ = make_coach(coach, trainer, oracle, os.path.join(tmp_path, "single"), iters=1)
    rep = c.learn(seed=4)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_coach_rank, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, n, boards, pis, vs, params, calls, wld, acalls in got:
        # the arena was sharded 3 + 3 by global game index and the all-reduced tally equals the single-process one
        assert wld == (rep[0]["nwins"], rep[0]["pwins"], rep[0]["draws"]) and acalls[0][1] == 3
        # every rank ends with the full, identically ordered sample set (rank order == global game-id order)
        assert n == rep[0]["samples"]
        assert np.array_equal(boards, c.history[0][0]) and np.array_equal(pis, c.history[0][1]) and np.array_equal(vs, c.history[0][2])
        # identical data + identical batches + averaged gradients -> the same weights as one process
        assert np.allclose(params, eng.params.get(1, eng.freed.get(1)), atol=1e-5)
    # the 5 episodes were sharded 3 + 2 by global id
    assert got[0][6] == [("selfplay", 3, 0, 0)] and got[1][6] == [("selfplay", 2, 3, 0)]


def test_examples_file_and_shuffle_are_the_documented_formats(mods, tmp_path):
    """`<iter>.examples` = "AZEX0001" | int64 H | int64 lens[H] | boards | pis | vs (raw f32), and the shuffle is the
    Fisher-Yates walk over the counter RNG that include/az_host.hpp implements too (known answers pinned here)."""
    coach, _ = mods
    rng = np.random.default_rng(0)
    hist = [(rng.random((n, 2, 6, 7)).astype(np.float32), rng.random((n, 7)).astype(np.float32), rng.random(n).astype(np.float32))
            for n in (3, 0, 5)]
    path = os.path.join(tmp_path, "7.examples")
    coach.save_examples(path, hist)
    raw = open(path, "rb").read()
    assert raw[:8] == b"AZEX0001" and len(raw) == 8 + 8 * 4 + 8 * 92 * 4
    assert np.frombuffer(raw[8:40], np.int64).tolist() == [3, 3, 0, 5]
    assert np.array_equal(np.frombuffer(raw[40:40 + 3 * 84 * 4], np.float32), hist[0][0].reshape(-1))
    back = coach.load_examples(path)
    assert len(back) == 3 and all(np.array_equal(a, b) for h, g in zip(hist, back) for a, b in zip(h, g))
    with open(path, "wb") as f:
        f.write(b"NOTAZEX0" + raw[8:])
    with pytest.raises(ValueError):
        coach.load_examples(path)
    p = coach.shuffle_permutation(10, seed=3, iteration=2)
    assert sorted(p.tolist()) == list(range(10))
    assert p.tolist() == coach.shuffle_permutation(10, 3, 2).tolist()
    assert p.tolist() != coach.shuffle_permutation(10, 3, 3).tolist() and p.tolist() != coach.shuffle_permutation(10, 4, 2).tolist()
    # scalar restatement of the documented walk
    def mix64(x):
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)
    want = list(range(10))
    for i in range(9, 0, -1):
        r = mix64(mix64(mix64(mix64(3) ^ 2) ^ i) ^ 5)
        j = ((r >> 32) * (i + 1)) >> 32
        want[i], want[j] = want[j], want[i]
    assert p.tolist() == want
    assert coach.shuffle_permutation(0, 1, 1).tolist() == [] and coach.shuffle_permutation(1, 1, 1).tolist() == [0]
