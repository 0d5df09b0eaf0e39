"""The C++ host-side mirror of the reference interface (include/az_host.hpp: Game / NNet / AsyncMcts /
arena::play_game(s) / Coach::execute_episode) driving the engine through the C ABI, checked against the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL_SALT = 0x51ED27


def test_cpp_host_mirror(engine_mod, oracle, tmp_path):
    exe = os.path.join(tmp_path, "test_host")
    libdir = os.path.dirname(engine_mod.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_host.cpp"), "-o", exe, "-L", libdir, "-laz_engine",
                           f"-Wl,-rpath,{libdir}"])
    sims, episodes = 25, 3
    out = subprocess.run([exe, str(sims), str(episodes)], check=True, stdout=subprocess.PIPE, text=True).stdout
    got = json.loads(out.strip().splitlines()[-1])
    ref = oracle.selfplay(episodes, sims, net_kind=oracle.NET_HASH, salt=1234 + 10 * MODEL_SALT, seed=17, threads=1)
    off = 0
    for ep in range(episodes):
        n = int(ref["game_len"][ep])
        assert got["episodes"][ep]["moves"] == ref["moves"][ep, :n].tolist()
        assert got["episodes"][ep]["samples"] == 2 * n
        z = ref["zs"][off:off + 2 * n].astype(np.float64).sum()
        assert abs(got["episodes"][ep]["zsum"] - z) < 1e-6
        assert abs(got["episodes"][ep]["pisum"] - 2 * n) < 1e-3
        off += 2 * n
    # two simulations in flight per tree through AsyncMcts::default(.., num_threads = 2, ..): the oracle's lock-step schedule
    ref2 = oracle.selfplay(episodes, sims + sims % 2, net_kind=oracle.NET_HASH, salt=1234 + 10 * MODEL_SALT, seed=17, threads=1, sim_threads=2)
    for ep in range(episodes):
        assert got["episodes_t2"][ep] == ref2["moves"][ep, :int(ref2["game_len"][ep])].tolist()
    assert got["odd_sims_panic"] is True
    # arena: new = model 11, old = model 10, same base salt -> oracle salts base + id*MODEL_SALT
    wld, results = oracle.arena(2, sims, net_kind=oracle.NET_HASH, salt=1234, seed=17, new_model_id=11, old_model_id=10)
    assert got["arena"] == results.tolist()
    assert got["terminal_root_panics"] is True and got["ended"] == -1.0
