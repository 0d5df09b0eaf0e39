"""Coach::learn end to end on the GPU (config 5 in miniature): self-play with the bf16 MFMA net, NNet::train (the
engine's own f32 trainer, or the torch trainer on the same device), weights stored under the next model id, arena
gate, examples + weights files; the Python host and the C++ host (include/az_host.hpp) must agree byte for byte."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_and_cpp_coach_agree(engine_mod, tmp_path):
    """The same configuration through alphazero-rs_amd/coach.py and through az_host::Coach: same reports, and the
    examples files and candidate weights they write are byte-identical (self-play, shuffle, NNet::train on the device
    and the arena are all deterministic functions of the seed)."""
    from alphazero_rs_amd.coach import Coach
    C, seed = 128, 11
    pydir, cdir = os.path.join(tmp_path, "py"), os.path.join(tmp_path, "cpp")
    e = engine_mod.Engine(device=0, max_batch=256, net_channels=C)
    try:
        e.net_init_random(0, 3)
        e.set_option("train_epochs", 2)
        msgs = []
        coach = Coach.setup(e, pydir, 1000000, 0.55, 15, 3, 100000, 1, 64, 16, 2, 48, 25, 1, 1000, 1, log=msgs.append)
        rep = coach.learn(seed=seed)
    finally:
        e.close()
    exe = os.path.join(tmp_path, "test_coach")
    libdir = os.path.dirname(engine_mod.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_coach.cpp"), "-o", exe, "-L", libdir, "-laz_engine",
                           f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe, cdir, str(C), str(seed)], check=True, stdout=subprocess.PIPE, text=True).stdout
    crep = json.loads([l for l in out.strip().splitlines() if l.startswith("[")][-1])
    assert len(rep) == len(crep) == 2
    for a, b in zip(rep, crep):
        for k in ("iteration", "samples", "nwins", "pwins", "draws", "accepted", "model_id"):
            assert a[k] == b[k], (k, a, b)
        assert np.allclose(np.array(a["losses"]).reshape(-1), b["losses"], rtol=1e-6)
        assert a["nwins"] + a["pwins"] + a["draws"] == 16 and a["samples"] > 0 and len(a["losses"]) == 2
    files = sorted(os.listdir(pydir))
    assert files == sorted(os.listdir(cdir)) and "0.examples" in files and "1.examples" in files and "1.aznet" in files
    for f in files:
        with open(os.path.join(pydir, f), "rb") as x, open(os.path.join(cdir, f), "rb") as y:
            assert x.read() == y.read(), f
    assert any(m.startswith("NEW/PREV WINS : ") for m in msgs)


def _has_model(e, k):
    try:
        e.net_get_params(k)
        return True
    except Exception:
        return False


def test_coach_resume_is_byte_identical(engine_mod, tmp_path):
    """Kill after iteration 0, restart in a fresh engine, run iteration 1: the examples file, the candidate's weights and
    the state file are byte-identical to an uninterrupted two-iteration run (the live model comes back from
    <id>.aznet + coach.state; self-play, shuffle, NNet::train and the arena are deterministic functions of the seed)."""
    from alphazero_rs_amd.coach import Coach
    C, seed = 128, 23
    args = (1000000, 0.55, 15, 3, 100000, 1, 64, 16)
    tail = (40, 25, 1, 1000, 1)

    def run(directory, iters, init_seed):
        e = engine_mod.Engine(device=0, max_batch=256, net_channels=C)
        try:
            e.net_init_random(0, init_seed)
            e.set_option("train_epochs", 1)
            c = Coach.setup(e, directory, *args, iters, *tail, log=lambda m: None)
            return c.learn(seed=seed), c.model_id
        finally:
            e.close()

    da, db = os.path.join(tmp_path, "a"), os.path.join(tmp_path, "b")
    rep_a, mid_a = run(da, 2, 3)
    run(db, 1, 3)
    rep_b, mid_b = run(db, 1, 999)          # a different random init: the restart must take the live model from the checkpoint
    assert rep_b[0]["iteration"] == 1 and mid_a == mid_b
    for k in ("samples", "nwins", "pwins", "draws", "accepted", "model_id", "losses"):
        assert rep_a[1][k] == rep_b[0][k], k
    assert sorted(os.listdir(da)) == sorted(os.listdir(db))
    for f in sorted(os.listdir(da)):
        with open(os.path.join(da, f), "rb") as x, open(os.path.join(db, f), "rb") as y:
            assert x.read() == y.read(), f


def test_coach_learn_on_gpu(engine_mod, tmp_path):
    from alphazero_rs_amd.coach import Coach
    from alphazero_rs_amd.trainer import Trainer
    C = 128
    e = engine_mod.Engine(device=0, max_batch=256, net_channels=C)
    try:
        e.net_init_random(0, 3)
        p0 = e.net_get_params(0)
        msgs = []
        coach = Coach.setup(e, str(tmp_path), 1000000, 0.55, 15, 3, 100000, 1, 64, 16, 2, 48, 25, 1, 1000, 1,
                            trainer=Trainer(channels=C, epochs=2, device=torch.device("cuda", 0)), log=msgs.append)
        rep = coach.learn(seed=11)
        assert len(rep) == 2
        for r in rep:
            assert r["nwins"] + r["pwins"] + r["draws"] == 16
            assert r["samples"] > 0 and len(r["losses"]) == 2
            assert all(np.isfinite(l[0]) and np.isfinite(l[1]) for l in r["losses"])
        assert os.path.exists(os.path.join(tmp_path, "0.examples")) and os.path.exists(os.path.join(tmp_path, "1.examples"))
        assert os.path.exists(os.path.join(tmp_path, "1.aznet"))
        # the loop keeps only the live model resident (the superseded slot is freed): reload both from their checkpoints
        live = coach.model_id
        assert sorted(k for k in (0, 1, 2) if _has_model(e, k)) == [live]
        e.net_load(0, os.path.join(tmp_path, "0.aznet"))
        e.net_load(1, os.path.join(tmp_path, "1.aznet"))
        assert np.array_equal(e.net_get_params(0), p0)
        # the candidate differs from its parent and predicts through the MFMA net like any other model
        p1 = e.net_get_params(1)
        assert p1.shape == p0.shape and not np.array_equal(p1, p0)
        states = np.zeros((4, 2), np.uint64)
        pi0, v0 = e.predict_states(states, 0)
        pi1, v1 = e.predict_states(states, 1)
        assert np.all(np.abs(pi1.sum(1) - 1) < 1e-5) and not np.array_equal(pi0, pi1)
        # the trainer's eval-mode torch net and the engine's bf16 net agree on the trained weights
        from alphazero_rs_amd.trainer import PolicyValueNet
        net = PolicyValueNet(C, p1, torch.device("cpu"))
        net.eval()
        from alphazero_rs_amd.coach import load_examples
        boards = load_examples(os.path.join(tmp_path, "1.examples"))[0][0][:64]
        with torch.no_grad():
            lg, vv = net(torch.from_numpy(boards))
        gpi, gv = e.predict(boards, 1)
        assert np.abs(torch.softmax(lg, 1).numpy() - gpi).max() < 2e-2 and np.abs(vv.numpy() - gv).max() < 5e-2
        # resume continues at iteration 2
        c2 = Coach.setup(e, str(tmp_path), 1000000, 0.55, 15, 3, 100000, 1, 64, 16, 1, 48, 25, 1, 1000, 1,
                         trainer=Trainer(channels=C, epochs=1, device=torch.device("cuda", 0)), log=msgs.append)
        assert c2.start_iteration == 2 and len(c2.history) == 2
    finally:
        e.close()
