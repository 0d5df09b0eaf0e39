"""The N>1 path on CPU: world_size-2 gloo run of the shard + pack + gather code that bench.py uses over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_samples(lo, hi):
    """Deterministic per-game tuples keyed on the GLOBAL game id (as the engine's outputs are)."""
    st, pi, z = [], [], []
    for g in range(lo, hi):
        n = 5 + g % 4
        r = np.random.default_rng(g)
        st.append(r.integers(0, 2**48, size=(n, 2), dtype=np.int64))
        p = r.random((n, 7)).astype(np.float32)
        pi.append(p / p.sum(1, keepdims=True))
        z.append(r.choice(np.array([-1, 1, 1e-4], np.float32), size=n))
    return np.concatenate(st), np.concatenate(pi), np.concatenate(z)


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero_rs_amd import dist as azdist
    lo, hi = azdist.shard_range(n_total, rank, world)
    s, p, z = _fake_samples(lo, hi)
    packed = azdist.pack_samples(torch.from_numpy(s), torch.from_numpy(p), torch.from_numpy(z))
    out, counts = azdist.gather_samples(packed, dst=0)
    if rank == 0:
        gs, gp, gz = azdist.unpack_samples(out)
        q.put((gs.numpy(), gp.numpy(), gz.numpy(), counts.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_samples_world2():
    n_total = 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    gs, gp, gz, counts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    es, ep, ez = _fake_samples(0, n_total)          # what a single rank would have produced
    assert np.array_equal(gs, es) and np.array_equal(gp, ep) and np.array_equal(gz, ez)
    assert int(counts.sum()) == es.shape[0]


def test_shard_ranges_partition():
    sys.path.insert(0, ROOT)
    from alphazero_rs_amd import dist as azdist
    for n in (0, 1, 7, 8192, 65536, 65537):
        for w in (1, 2, 4, 8):
            r = [azdist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_pack_roundtrip_and_symmetries():
    sys.path.insert(0, ROOT)
    from alphazero_rs_amd import dist as azdist
    from alphazero_rs_amd import engine  # noqa: F401  (library must load on CPU)
    s, p, z = _fake_samples(3, 6)
    ts, tp, tz = torch.from_numpy(s), torch.from_numpy(p), torch.from_numpy(z)
    a, b, c = azdist.unpack_samples(azdist.pack_samples(ts, tp, tz))
    assert torch.equal(a, ts) and torch.equal(b, tp) and torch.equal(c, tz)
    s2, p2, z2 = azdist.expand_symmetries(ts, tp, tz)
    assert s2.shape[0] == 2 * ts.shape[0]
    assert torch.equal(p2[1::2], tp.flip(1)) and torch.equal(z2[0::2], z2[1::2])
    # mirror is an involution and moves column c to 6-c
    one = torch.tensor([[1 << (2 * 7 + 3), 1 << (6 * 7)]], dtype=torch.int64)
    m, _, _ = azdist.expand_symmetries(one, tp[:1], tz[:1])
    assert m[1, 0].item() == 1 << (4 * 7 + 3) and m[1, 1].item() == 1


def test_bench_self_launch_dry_dist_world2():
    """`python3 bench.py --gpus 2` with no torchrun around it: the parent (which never imports torch or the engine) starts
    the two ranks itself; with --dry-dist gloo they rehearse the process group, the barrier/MAX-over-ranks timing and the
    gather on CPU and rank 0 prints exactly one JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-dist", "gloo"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dry_run"] is True
    assert d["tuples_gathered"] == 2 * (64 * 7) * 2 + 2       # both ranks' tuples of both timed steps reached rank 0's count


def test_bench_parent_does_not_touch_the_gpu():
    """The self-launch branch sits before the first torch / engine import of bench.py (a process that has initialised the GPU
    must not start the ranks)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("self_launch(args.gpus)") < main.index("import torch")
    launch = src[src.index("def self_launch("):src.index("def dry_dist(")]
    assert "import torch" not in launch and "engine" not in launch and "subprocess.call" in launch and "os.exec" not in launch


def test_bench_gathers_through_the_c_abi():
    """bench.py's N-rank step calls the product's collective (az_comm_init + az_gather_samples through the C ABI), not the
    torch.distributed stand-in of alphazero-rs_amd/dist.py -- that one stays for the gloo rehearsal only."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    step = src[src.index("    def step(i):", src.index("def main():")):src.index("    for i in range(args.warmup):", src.index("def main():"))]
    assert "e.gather_samples(" in step and "azdist." not in step
    main = src[src.index("def main():"):]
    assert "e.comm_init(rank, world" in main and "e.comm_unique_id()" in main
    dry = src[src.index("def dry_dist("):src.index("def main():")]
    assert "azdist.gather_samples" in dry


def test_sharded_cpp_example_builds_and_rejects_bad_wiring(tmp_path):
    """examples/connect_four_sharded.cpp (rank, world, id-file: the launcher a host without Python starts once per GPU) compiles
    against the header, and wiring errors end with exit code 2 before any GPU call (the run itself is a GPU test)."""
    import subprocess
    libdir = os.path.join(ROOT, "alphazero-rs_amd")
    exe = os.path.join(tmp_path, "connect_four_sharded")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "connect_four_sharded.cpp"),
                           "-o", exe, "-L", libdir, "-laz_engine", f"-Wl,-rpath,{libdir}"])
    assert subprocess.run([exe], capture_output=True).returncode == 2
    assert subprocess.run([exe, "3", "2", os.path.join(tmp_path, "id"), os.path.join(tmp_path, "ck")], capture_output=True).returncode == 2
