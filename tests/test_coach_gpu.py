"""Coach::learn end to end on the GPU (config 5 in miniature): self-play with the bf16 MFMA net, NNet::train with
the torch trainer on the same device, weights uploaded under the next model id, arena gate, examples + weights files."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
        z = np.load(os.path.join(tmp_path, "1.examples"), allow_pickle=False)
        boards = z["boards"][:64]
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
