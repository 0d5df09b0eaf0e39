#!/usr/bin/env python3
"""bench.py -- headline benchmark of the async_mcts hot path on MI355X.

Metric (BASELINE.json): self-play games/sec (+ MCTS node-expansions/sec) on connect_four at 100 sims/move,
8192 concurrent games per GPU, bf16 policy+value net (C=512), random-init weights, synthetic = self-generated
positions from the empty board.

A "step" = one episode batch through the hot path: `--episodes` (default 8 x --games) self-play games per GPU played to completion on
`--games` concurrent slots (finished slots are refilled; the last `--games` episodes of a step drain on shrinking batches, which is part of the measurement), then -- when N > 1 -- the ONE RCCL gather of the (s, pi, z)
tuples to rank 0 THROUGH THE C ABI (az_comm_init + az_gather_samples on the engine's stream: the product's collective, not a
torch.distributed stand-in).  Weak scaling: per-GPU work is fixed; `value` = all ranks' games / max-over-ranks time.

Two passes, both named in `config.passes`: the TIMED region runs with the engine's HIP-event brackets off (its search loop replays
hipGraphs, as any embedding host's would); a separate profiled pass of the same call afterwards (brackets on every
`--profile-every`th simulation step) gives the live per-kernel times behind "roofline" (the dominant kernel, conv3 on the matrix
cores) and "kernels".  "cpu_baseline" = the CPU oracle driving the same search with the same f32 net on torch-CPU, bounded
sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0
FLOP_PER_LEAF = 328_986_624  # the net as dense arithmetic
# with conv1 and conv2 as table lookups (az_set_option "conv2_table", the default) those two layers' flops are never executed
FLOP_PER_LEAF_TABLES = FLOP_PER_LEAF - 2 * 42 * 18 * 512 - 2 * 42 * 512 * 4608


def cpu_baseline(params, channels, sims, mean_plies=None, budget_s=20.0, seed=1):
    """The oracle (port of the reference semantics) running the SAME workload on the host cores:
    100 sims/move self-play, NNet::predict = the textbook f32 net on torch-CPU at inference batch 1
    (the reference's own configuration: inference_batch_size 1, examples/connect_four.rs:61)."""
    import numpy as np
    import torch
    from oracle import oracle_py as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from net_ref import unpack, BN_EPS
    import torch.nn.functional as F

    P = unpack(np.asarray(params, np.float32), channels)
    convs = []
    for l in range(4):
        bn = P[f"conv{l+1}_bn"]
        convs.append((P[f"conv{l+1}_w"].permute(3, 2, 0, 1).contiguous(), P[f"conv{l+1}_b"], bn, 1 if l < 2 else 0))

    def predict(boards, model_id):
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(boards))
            for w, b, bn, pad in convs:
                x = torch.relu(F.batch_norm(F.conv2d(x, w, b, padding=pad), bn[2], bn[3], bn[0], bn[1], False, 0.0, BN_EPS))
            x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
            for l in range(2):
                bn = P[f"fc{l+1}_bn"]
                x = torch.relu(F.batch_norm(x @ P[f"fc{l+1}_w"] + P[f"fc{l+1}_b"], bn[2], bn[3], bn[0], bn[1], False, 0.0, BN_EPS))
            pi = torch.softmax(x @ P["pi_w"] + P["pi_b"], dim=1)
            v = torch.tanh(x @ P["v_w"] + P["v_b"]).reshape(-1)
        return pi.numpy(), v.numpy()

    orc.set_predict_callback(predict)
    share = min(os.cpu_count() or 1, 16)          # the box's CPU share for one GPU
    torch.set_num_threads(share)
    # Bounded sample: the opening plies of one episode, move by move, until ~budget_s of CPU work is done.
    tree = orc.Tree(sims, net_kind=orc.NET_CALLBACK)
    s, ply, gid, plies, games_done = (0, 0), 0, 0, 0, 0
    st = {"leaf_evals": 0, "expansions": 0, "sims": 0}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        temp = 1.0 if ply + 1 < 15 else 0.0
        pi, counts, q = tree.get_action_prob(s[0], s[1], temp, seed=seed, game_id=gid)
        a = orc.lib().azo_rng_choose_weighted(orc.lib().azo_rng_draw(seed, gid, ply, 2), pi.ctypes.data, 7)
        s = orc.c4_play(s[0], s[1], a)
        ply += 1
        plies += 1
        if orc.c4_ended(*s) != 0.0:          # episode over: start the next one on a fresh tree
            for k in st:
                st[k] += tree.stats()[k]
            tree, s, ply, gid, games_done = orc.Tree(sims, net_kind=orc.NET_CALLBACK), (0, 0), 0, gid + 1, games_done + 1
    dt = time.perf_counter() - t0
    for k in st:
        st[k] += tree.stats()[k]
    ply = plies
    plies_per_game = mean_plies if mean_plies else 22.6
    # tree-only rate (stub net), one game per thread, for context
    t1 = time.perf_counter()
    rs = orc.selfplay(8 * share, sims, net_kind=orc.NET_STUB, seed=seed, threads=share, want_samples=False)
    dts = time.perf_counter() - t1
    return {
        "value": ply / dt / plies_per_game, "unit": "games/s", "cores": share, "kind": "port",
        "sample": f"{ply} plies ({games_done} finished episodes + the opening of the next) x {sims} sims/move = {st['leaf_evals']} leaf evals of the f32 "
                  f"C={channels} net on torch-CPU ({share} threads, inference batch 1 as examples/connect_four.rs:61) in "
                  f"{dt:.1f} s; games/s = plies/s / {plies_per_game:.1f} mean plies per game of the GPU run",
        "node_expansions_per_sec": st["expansions"] / dt, "simulations_per_sec": st["sims"] / dt,
        "tree_only": {"value": 8 * share / dts, "unit": "games/s", "cores": share,
                      "sample": f"{8 * share} episodes, stub net (pi=1/7, v=+1), one game per thread, {dts:.2f} s",
                      "sims_per_sec": rs["stats"]["sims"] / dts},
    }


def config1_dropin(episodes=20, sims=25):
    import subprocess
    import tempfile
    from alphazero_rs_amd import engine as azeng
    libdir = os.path.dirname(azeng.LIB_PATH)
    exe = os.path.join(tempfile.mkdtemp(prefix="az_dropin_"), "config1_dropin")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "config1_dropin.cpp"),
                           "-o", exe, "-L", libdir, "-laz_engine", f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe, str(episodes), str(sims)], check=True, stdout=subprocess.PIPE, text=True, timeout=300).stdout
    res = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    res["host"] = "C++ host (include/az_host.hpp): Coach::execute_episode over one az_host::AsyncMcts (n_games = 1), move by move through the C ABI"
    # the CPU side of the same shape: the oracle's execute_episode with the stub net, one thread
    from oracle import oracle_py as orc
    t0 = time.perf_counter()
    r = orc.selfplay(episodes, sims, net_kind=orc.NET_STUB, seed=0, threads=1, want_samples=False)
    dt = time.perf_counter() - t0
    res["cpu_port_stub_net"] = {"moves_per_sec": float(r["game_len"].sum()) / dt, "episodes": episodes, "cores": 1,
                                "note": "oracle execute_episode, stub net, one host core (examples/connect_four.rs:55-71 verbatim)"}
    return res


def tree_gbps(st):
    """Algorithmic tree bytes per launch (all launches) over the mean duration of the HIP-event-timed launches."""
    per_launch = st["tree_bytes"] / max(1, st["tree_launches"])
    return per_launch / (st["tree_ms"] * 1e-3 / max(1, st["tree_launches_timed"])) / 1e9


def csrc_sha():
    """Content hash of the kernel sources: what a committed PMC fold must have been collected at to describe THIS build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "alphazero-rs_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_pmc():
    """The newest committed PMC fold (tools/collect_profiles.sh -> profiles/r<NN><x>_pmc_traffic.json): bench.py cannot
    collect hardware counters itself, so per-kernel HBM bytes come from the rocprofv3 --pmc passes of this same command."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        pmc = json.load(open(files[-1]))
        # the fold records the source hash (and commit) it was collected at: anything else is flagged, never quoted silently
        pmc["_stale"] = pmc.get("csrc_sha") != csrc_sha()
        return pmc, os.path.basename(files[-1])
    except Exception:
        return None, None


def self_launch(n):
    """`python3 bench.py --gpus N` without torchrun: start N ranks (one per GPU) under torch.distributed.run as a CHILD
    process -- never an exec, and before anything in this process has initialised the GPU -- and pass its output through.
    Rank 0 of the children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_dist(args, rank, world):
    """--dry-dist gloo: the launcher, process group, barrier/timing protocol and the ONE gather of bench.py's N-rank path
    on CPU with synthetic tuples in place of the engine's output (CPU tests; nothing here is a measurement)."""
    import torch
    import torch.distributed as dist
    from alphazero_rs_amd import dist as azdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(args.dry_dist)
    episodes = args.episodes or 64

    def step(i):
        first = (i * world + rank) * episodes
        g = torch.Generator().manual_seed(first)
        n = episodes * 7 + rank
        packed = azdist.pack_samples(torch.randint(0, 2**40, (n, 2), generator=g), torch.rand((n, 7), generator=g),
                                     torch.rand(n, generator=g))
        gathered, counts = azdist.gather_samples(packed, dst=0)
        if rank == 0 and int(counts.sum()) != gathered.shape[0]:
            raise RuntimeError("gather_samples: count mismatch")
        return n
    for i in range(args.warmup):
        step(i)
    dist.barrier()
    t0 = time.perf_counter()
    tuples = sum(step(args.warmup + i) for i in range(args.steps))
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    agg = torch.tensor([tuples], dtype=torch.float64)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "selfplay_games_per_sec", "value": 0.0, "unit": "games/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": float(t.item()) / max(1, args.steps) * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": f"DRY RUN of the {world}-rank launcher + gather over {args.dry_dist}: no engine, no GPU",
                                     "parallelism": f"games-sharded x{world}"},
                          "dry_run": True, "tuples_gathered": float(agg.item())}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--per-call", action="store_true", help="one az_selfplay call per step instead of one session over all steps")
    ap.add_argument("--games", type=int, default=8192, help="concurrent game slots per GPU")
    ap.add_argument("--episodes", type=int, default=0, help="episodes per GPU per step (0 = 8 x --games)")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--net", default="conv", choices=["conv", "stub"])
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event brackets (roofline = null)")
    ap.add_argument("--profile-every", type=int, default=64, help="bracket every n-th simulation step with HIP events (each bracket idles the GPU a little)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary no-dedup and arena (config 3) measurements")
    ap.add_argument("--no-train-probe", action="store_true", help="skip the NNet::train throughput probe (auxiliary field)")
    ap.add_argument("--conv2-table", type=int, default=1, choices=[0, 1], help="conv1 + conv2 as table lookups (default) / 0 = conv2 as the MFMA implicit GEMM")
    ap.add_argument("--dedup", type=int, default=1, choices=[0, 1], help="leaf de-duplication + per-call evaluation cache (bit-exact); 0 = every requested row runs")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="az_set_option on the timed engine (experiments; named in config.options)")
    ap.add_argument("--force-dist", action="store_true", help="init the process group and run the gather even at world size 1 (rehearsal)")
    ap.add_argument("--dry-dist", default="", choices=["", "gloo"],
                    help="rehearse the N-rank launcher + gather on CPU over gloo with synthetic tuples (no engine, no GPU; value = 0)")
    args = ap.parse_args()
    episodes = args.episodes or 8 * args.games

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python3 bench.py --gpus N`: this parent never touches the GPU (no torch import, no engine); it starts the
        # N ranks as children of torch.distributed.run and relays their output and exit code
        raise SystemExit(self_launch(args.gpus))

    # Libraries write banners to the process's stdout (RCCL's version block at communicator creation, for one): everything but the ONE
    # JSON line goes to stderr -- file descriptor 1 points at stderr until the line is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_dist:
        os.dup2(stdout_fd, 1)
        return dry_dist(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from alphazero_rs_amd import engine as azeng
    from alphazero_rs_amd import dist as azdist
    # the timed region runs WITHOUT the engine's HIP-event brackets (profile mode launches every kernel on its own; un-bracketed the
    # search loop replays hipGraphs: what matters on the shrinking batches of a step's drain); the brackets come on for a separate pass
    e = azeng.Engine(device=local_rank, max_batch=args.games, net_channels=args.channels, profile=False)
    e.set_option("eval_dedup", args.dedup)
    e.set_option("profile_every", args.profile_every)
    e.set_option("conv2_table", args.conv2_table)
    for kv in args.set:
        k, v = kv.split("=")
        e.set_option(k, int(v))
    if args.net == "conv":
        e.net_init_random(0, seed=args.seed)       # identical weights on every rank (replicated, 21.5 MB bf16)
    else:
        e.net_set_kind(0, azeng.NET_STUB)

    cap = episodes * 42
    out = {"states": torch.empty((cap, 2), dtype=torch.int64, device=dev),
           "pis": torch.empty((cap, 7), dtype=torch.float32, device=dev),
           "zs": torch.empty(cap, dtype=torch.float32, device=dev)}

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    gathered = None
    if use_dist:
        # the product's collective: bind the engine to an RCCL communicator (az_comm_init; the unique id travels from rank 0 over
        # the process group that also carries the timing protocol) and gather with az_gather_samples on the engine's stream
        uid = torch.from_numpy(e.comm_unique_id() if rank == 0 else np.zeros(azeng.COMM_ID_BYTES, np.uint8)).to(dev)
        dist.broadcast(uid, src=0)
        e.comm_init(rank, world, uid.cpu().numpy())
        if rank == 0:
            gcap = cap * world
            gathered = {"states": torch.empty((gcap, 2), dtype=torch.int64, device=dev),
                        "pis": torch.empty((gcap, 7), dtype=torch.float32, device=dev),
                        "zs": torch.empty(gcap, dtype=torch.float32, device=dev)}

    # One self-play SESSION over the warm-up and the timed steps (az_selfplay_begin / _next / _end): a step fetches the next `episodes`
    # episodes in id order while the slots they freed already play the following step's -- the slots stay full from step to step, and the
    # drain of the last slots (6 % of a 65536-episode call) is paid once, inside the last timed step.  --per-call: one az_selfplay call
    # per step instead (rounds 1-3 and r04b / r04c: every step ramps up and drains on its own).
    n_steps_total = args.warmup + args.steps
    session = not args.per_call
    if session:
        e.selfplay_begin(n_steps_total * episodes, args.sims, 0, seed=args.seed, first_game_id=rank * n_steps_total * episodes,
                         concurrent=args.games, symmetries=False)

    def step(i):
        if session:
            r = e.selfplay_next(episodes, want_boards=False, out=out)
        else:
            first = (i * world + rank) * episodes          # global game ids: disjoint per (step, rank)
            r = e.selfplay(n_games=episodes, concurrent=args.games, num_sims=args.sims, model_id=0, seed=args.seed,
                           first_game_id=first, symmetries=False, want_boards=False, out=out)
        n = r["count"]
        if use_dist:
            _gs, _gp, gz, counts = e.gather_samples(out["states"][:n], out["pis"][:n], out["zs"][:n], dst=0, is_dst=rank == 0, out=gathered)
            if rank == 0 and int(counts.sum()) != gz.shape[0]:
                raise RuntimeError("az_gather_samples: count mismatch")
        return n, int(r["game_len"].sum())

    for i in range(args.warmup):
        step(i)
    e.reset_stats()
    barrier()
    t0 = time.perf_counter()
    samples = plies = 0
    for i in range(args.steps):
        n, p = step(args.warmup + i)
        samples += n
        plies += p
    barrier()
    dt = time.perf_counter() - t0
    st_timed = e.stats()
    if session:
        e.selfplay_end()
    # the profiled pass (outside the timed region): the same call with the HIP-event brackets on
    prof_s = None
    if not args.no_profile:
        e.set_option("profile", 1)
        e.reset_stats()
        t1 = time.perf_counter()
        r = e.selfplay(n_games=episodes, concurrent=args.games, num_sims=args.sims, model_id=0, seed=args.seed,
                       first_game_id=(args.warmup + args.steps) * world * episodes + rank * episodes, symmetries=False, want_boards=False, out=out)
        torch.cuda.synchronize()
        prof_s = time.perf_counter() - t1
        st_prof = e.stats()
        e.set_option("profile", 0)
    # counts and rates come from the timed region; per-kernel times (net_*_ms, tree_ms, their flops / bytes / launch counts) from the profiled pass
    st = dict(st_timed)
    if prof_s is not None:
        for k, v in st_prof.items():
            if k.startswith("net_") or k in ("tree_ms", "tree_launches_timed", "tree_bytes", "tree_launches", "depth_sum", "simulations_prof"):
                st[k] = v
        st["simulations_for_tree"] = st_prof["simulations"]
    else:
        st["simulations_for_tree"] = st_timed["simulations"]
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        agg = torch.tensor([st["expansions"], st["simulations"], st["leaf_evals"], plies, st["leaf_rows_executed"],
                            st["eval_cache_hits"], st["eval_batch_dups"]], dtype=torch.float64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        expansions, simulations, leaf_evals, plies_all, rows_exec, cache_hits, batch_dups = (float(x) for x in agg.tolist())
    else:
        expansions, simulations, leaf_evals, plies_all = st["expansions"], st["simulations"], st["leaf_evals"], plies
        rows_exec, cache_hits, batch_dups = st["leaf_rows_executed"], st["eval_cache_hits"], st["eval_batch_dups"]

    flop_exec = FLOP_PER_LEAF_TABLES if args.conv2_table else FLOP_PER_LEAF
    if rank == 0:
        games = episodes * args.steps * world
        line = {
            "metric": "selfplay_games_per_sec", "value": games / dt, "unit": "games/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.net == "conv" else "u64", "data": "synthetic",
            "config": {"workload": f"connect_four self-play, {args.games} concurrent games/GPU, {episodes} episodes/GPU/step, "
                                   f"{args.sims} sims/move, temp_threshold 15, cpuct 1, "
                                   + (f"bf16 policy+value net C={args.channels} (random init)" if args.net == "conv" else "stub net"),
                       "concurrent_games_per_gpu": args.games, "episodes_per_gpu_per_step": episodes,
                       "sims_per_move": args.sims, "net": args.net, "parallelism": f"games-sharded x{world}",
                       "symmetries": "identity only in the timed region (the mirrored twin of every tuple is regenerated where the tuples are consumed: "
                                     "k_emit_samples' mirror pass is 0.004 % of device time)",
                       "csrc_sha": csrc_sha(),
                       "options": args.set or None,
                       "session": ("one self-play session over the warm-up and the timed steps (az_selfplay_begin / _next / _end): a step is the next "
                                   f"{episodes} episodes in id order, the slots stay full from step to step, and the timed region ends with the session's "
                                   "drain (every episode of every timed step is finished and emitted inside it); --per-call times one az_selfplay call per step")
                                  if session else "one az_selfplay call per step (--per-call)",
                       "passes": {"timed": f"{args.steps} steps after {args.warmup} warm-up steps, the engine's HIP-event brackets OFF (search loop as hipGraph replays): "
                                           "value, ms_per_step and every */s rate",
                                  "profiled": (f"one more step of the same call afterwards with the brackets ON (every {args.profile_every}th simulation step; "
                                               f"{prof_s:.2f} s): roofline, kernels, conv2_table, tree_hbm") if prof_s is not None else None,
                                  "gather": "az_comm_init + az_gather_samples (RCCL on the engine's stream, through the C ABI) inside the timed step" if use_dist else None}},
            # the same batch as ONE az_selfplay call (the profiled pass: its own ramp and drain, an empty evaluation cache; HIP-event brackets on
            # every 64th simulation step cost it ~0.3 %): what rounds 1-3 reported as `value`, and what a session is measured against
            "per_call": ({"games_per_sec": episodes * world / prof_s, "seconds": prof_s, "episodes_per_gpu": episodes,
                          "executed_over_requested": st_prof["leaf_rows_executed"] / max(1, st_prof["leaf_rows_requested"]),
                          "note": "rank 0's profiled az_selfplay call of one step's episodes x n_gpus"} if prof_s is not None else None),
            "node_expansions_per_sec": expansions / dt, "simulations_per_sec": simulations / dt,
            "leaf_evals_per_sec": leaf_evals / dt, "mean_plies": plies_all / games,
            # leaf de-duplication (bit-exact, az_engine.h "eval_dedup"): the trees REQUEST leaf_evals rows, the net EXECUTES
            # only the distinct states not yet in this call's evaluation cache; every MFMA figure counts executed rows only
            "leaf_rows": {"requested_per_sec": leaf_evals / dt, "executed_per_sec": rows_exec / dt,
                          "executed_over_requested": rows_exec / max(1.0, leaf_evals),
                          "cache_hits_over_requested": cache_hits / max(1.0, leaf_evals),
                          "batch_duplicates_over_requested": batch_dups / max(1.0, leaf_evals), "dedup": args.dedup},
            # MFMA flops the chip really executed (conv3, conv4, fc1, fc2 + heads; conv1 / conv2 are table lookups unless
            # --conv2-table 0) over the dense bf16 peak; "as_dense" prices the same rows as if every layer were dense arithmetic
            "mfma_fraction_end_to_end": (rows_exec / dt) * flop_exec / (MFMA_PEAK_TFLOPS * 1e12 * world) if args.net == "conv" else None,
            "net_flops": {"per_row_executed": flop_exec, "per_row_as_dense": FLOP_PER_LEAF, "conv2_table": args.conv2_table,
                          "fraction_as_dense": (rows_exec / dt) * FLOP_PER_LEAF / (MFMA_PEAK_TFLOPS * 1e12 * world)} if args.net == "conv" else None,
        }
        if args.net == "conv":
            # exact, counted on the device over EVERY forward of the TIMED region (independent of the brackets): what a profiler's total
            # k_conv3_auto time over the same region has to be divided by (the smallest batches run conv3 on the ring / skinny kernels,
            # and the idle half of a dual launch does no work)
            line["k_conv3_auto_accounting_timed_region"] = {
                "rows": st_timed["net_conv3_image_rows"], "working_launches": st_timed["net_conv3_image_launches"],
                "rows_per_working_launch": st_timed["net_conv3_image_rows"] / max(1, st_timed["net_conv3_image_launches"]),
                "flop_per_row": 2.0 * 20 * 512 * 4608}
        roof = None
        if not args.no_profile and args.net == "conv" and st["net_launches"] > 0:
            pmc, pmc_name = committed_pmc()

            def pmc_traffic(kernel_substr, flop_sum, flop_per_row):
                """HBM bytes per launch of the committed PMC passes, scaled to this run's mean rows per launch."""
                if not pmc:
                    return None, None
                for k, v in pmc["kernels"].items():
                    if kernel_substr in k:
                        return v["hbm_bytes_per_leaf"] * (flop_sum / st["net_launches"]) / flop_per_row, pmc_name
                return None, None
            common = {"peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "launches": st["net_launches"],
                      "traffic_source": {"file": f"profiles/{pmc_name}", "csrc_sha": pmc.get("csrc_sha"), "git_sha": pmc.get("git_sha")} if pmc else None,
                      "traffic_stale": bool(pmc["_stale"]) if pmc else None,
                      "launches_are": f"every {args.profile_every}th simulation step of the profiled pass (HIP events on the engine's stream; config.passes)",
                      "net_forward_tflops": st["net_total_flops"] / (st["net_total_ms"] * 1e-3) / 1e12,
                      "net_forward_ms": st["net_total_ms"] / st["net_launches"]}
            if args.conv2_table:
                # conv2 is a table gather now; the dominant kernel (and the dominant MFMA kernel) is conv3
                ach = st["net_conv3_flops"] / (st["net_conv3_ms"] * 1e-3) / 1e12
                traffic, src = pmc_traffic("k_conv3_auto", st["net_conv3_flops"], 2.0 * 20 * 512 * 4608)
                if traffic is None:
                    traffic, src = pmc_traffic("k_conv_valid_", st["net_conv3_flops"], 2.0 * 20 * 512 * 4608)      # folds older than round 3
                roof = {"bound": "mfma", "kernel": "k_conv3_auto<2, true> (conv3: 3x3 valid, 512->512, image-resident implicit GEMM on MFMA, 12 boards x 128 channels per tile "
                                                   "(x 64 in a short last round), two 4-wave workgroups per CU, asm-issued LDS-DMA into a bank-conflict-free LDS image, "
                                                   "fragment reads interleaved into the MFMA clusters; small batches run k_gemm_skinny / the ring instead)",
                        "achieved": ach, "frac": ach / MFMA_PEAK_TFLOPS, "traffic": traffic,
                        "traffic_unit": f"HBM bytes per launch (PMC passes of this command committed as profiles/{src}: bytes per executed row x "
                                        "this run's mean rows per launch)",
                        "avg_launch_ms": st["net_conv3_ms"] / st["net_launches"], "avg_flop_per_launch": st["net_conv3_flops"] / st["net_launches"],
                        # exact, counted on the device over EVERY forward of the profiled pass (not the sampled ones): what a profiler's
                        # total k_conv3_auto time has to be divided by (the smallest batches run conv3 on the ring / skinny kernels, and the
                        # idle half of a dual launch does no work)
                        "k_conv3_auto_accounting": {"rows": st["net_conv3_image_rows"], "working_launches": st["net_conv3_image_launches"],
                                                    "rows_per_working_launch": st["net_conv3_image_rows"] / max(1, st["net_conv3_image_launches"]),
                                                    "flop_per_row": 2.0 * 20 * 512 * 4608},
                        **common}
                gb = st["net_conv2_bytes"] / (st["net_conv2_ms"] * 1e-3) / 1e9
                line["conv2_table"] = {"bound": "hbm", "kernel": "k_conv2_table_x (conv1 + conv2 as nine gathered rows of the per-model U table per output "
                                                                 "position; 181 MB f16 table, served mostly from L2 / Infinity Cache)",
                                       "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                                       "avg_launch_ms": st["net_conv2_ms"] / st["net_launches"],
                                       "algorithmic_bytes_per_row": st["net_conv2_bytes"] / max(1.0, st["net_conv3_flops"] / (2.0 * 20 * 512 * 4608)),
                                       "note": "achieved counts the table rows gathered + activation rows written per launch; above the HBM peak means cache hits"}
            else:
                ach = st["net_conv2_flops"] / (st["net_conv2_ms"] * 1e-3) / 1e12
                traffic, src = pmc_traffic("k_conv_same_pipe<1", st["net_conv2_flops"], 2.0 * 42 * 512 * 4608)
                roof = {"bound": "mfma", "kernel": "k_conv_same_pipe<1, true> (conv2: 3x3 same, 512->512, image-resident implicit GEMM on MFMA, two 4-wave "
                                                   "workgroups per CU, input image gathered from the conv1 pattern table)",
                        "achieved": ach, "frac": ach / MFMA_PEAK_TFLOPS, "traffic": traffic,
                        "traffic_unit": f"HBM bytes per launch (PMC passes of this command committed as profiles/{src})",
                        "avg_launch_ms": st["net_conv2_ms"] / st["net_launches"], "avg_flop_per_launch": st["net_conv2_flops"] / st["net_launches"],
                        **common}
        elif not args.no_profile and st["tree_ms"] > 0:
            ach = tree_gbps(st)
            roof = {"bound": "hbm", "kernel": "k_search_fixture (tree traversal: the whole search of a move in one launch; stub net evaluated in registers)", "achieved": ach,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}
        line["roofline"] = roof
        if not args.no_profile and args.net == "conv" and st["net_launches"] > 0:
            # the step's kernels from the live HIP-event brackets of the profiled pass (the same sample of launches as `roofline`):
            # average duration per launch, share of the bracketed step, and the fraction of the roofline that bounds each
            L = st["net_launches"]
            rows = st["net_rows_timed"] / L
            tree_us = st["tree_ms"] * 1e3 / max(1, st["tree_launches_timed"])
            us = {"conv1+conv2 (k_conv2_table_x)" if args.conv2_table else "conv2 (k_conv_same_pipe)": st["net_conv2_ms"] * 1e3 / L,
                  "conv3": st["net_conv3_ms"] * 1e3 / L, "conv4": st["net_conv4_ms"] * 1e3 / L, "fc1+fc2+heads": st["net_fc_ms"] * 1e3 / L,
                  "tree (k_backup_select)": tree_us}
            step_us = st["net_total_ms"] * 1e3 / L + tree_us
            def mf(flops, ms):
                return flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if ms > 0 else None
            frac = {"conv3": mf(st["net_conv3_flops"], st["net_conv3_ms"]), "conv4": mf(st["net_conv4_flops"], st["net_conv4_ms"]),
                    "fc1+fc2+heads": mf(st["net_fc_flops"], st["net_fc_ms"]), "tree (k_backup_select)": tree_gbps(st) / HBM_PEAK_GBS}
            if args.conv2_table:
                frac["conv1+conv2 (k_conv2_table_x)"] = st["net_conv2_bytes"] / (st["net_conv2_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            else:
                frac["conv2 (k_conv_same_pipe)"] = mf(st["net_conv2_flops"], st["net_conv2_ms"])
            line["kernels"] = {"mean_rows_per_launch": rows, "step_us": step_us,
                               "per_kernel": {k: {"avg_us": v, "share_of_step": v / step_us, "frac_of_roofline": frac.get(k),
                                                  "bound": "hbm (cache-side bytes; > 1 = served from L2 / Infinity Cache)" if "table" in k else ("hbm" if "tree" in k else "mfma")}
                                              for k, v in us.items()},
                               "note": f"HIP events on the engine's stream around every {args.profile_every}th simulation step of the profiled pass"}
        if not args.no_profile and st["tree_ms"] > 0:
            tree_counter = None
            pmc, pmc_name = committed_pmc()
            if pmc:
                for k, v in pmc["kernels"].items():
                    if "k_backup_select" in k:
                        # counter bytes per launch of the committed PMC pass over this run's mean launch time
                        launch_s = st["tree_ms"] * 1e-3 / max(1, st["tree_launches_timed"])
                        tree_counter = {"hbm_bytes_per_launch_pmc": v["hbm_bytes_per_launch"], "source": f"profiles/{pmc_name}",
                                        "source_csrc_sha": pmc.get("csrc_sha"), "source_git_sha": pmc.get("git_sha"), "stale": bool(pmc["_stale"]),
                                        "mean_launch_us_this_run": launch_s * 1e6,
                                        "counter_GBps": v["hbm_bytes_per_launch"] / launch_s / 1e9,
                                        "counter_over_algorithmic": v["hbm_bytes_per_launch"] / max(1.0, st["tree_bytes"] / max(1, st["tree_launches"]))}
            line["tree_hbm"] = {"achieved_GBps": tree_gbps(st), "peak_GBps": HBM_PEAK_GBS,
                                "counter": tree_counter,
                                "frac": tree_gbps(st) / HBM_PEAK_GBS,
                                "frac_of_measured_copy_bw": tree_gbps(st) / 6290.0,
                                "launches_timed": st["tree_launches_timed"], "launches": st["tree_launches"],
                                "algorithmic_bytes_per_sim": st["tree_bytes"] / max(1, st["simulations_for_tree"]),
                                "mean_depth": st["depth_sum"] / max(1, st["simulations_for_tree"])}
        if world == 1 and not args.no_cpu_baseline and args.net == "conv":
            try:
                line["cpu_baseline"] = cpu_baseline(e.net_get_params(0), args.channels, args.sims, mean_plies=plies_all / games)
            except Exception as ex:          # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"error": repr(ex)}
        else:
            line["cpu_baseline"] = None
        if world == 1 and args.net == "conv" and not args.no_train_probe:
            # auxiliary, outside the timed region and not part of `value`: NNet::train (az_net_train, f32 MFMA kernels) on
            # synthetic samples at the recipe's batch of 64 -- the other dense workload of the Coach loop
            try:
                rng = np.random.default_rng(0)
                bs, n_short, n_long = 64, 200, 800
                tb = (rng.random((n_long * bs, 2, 6, 7)) < 0.2).astype(np.float32)
                tp = rng.dirichlet(np.ones(7), n_long * bs).astype(np.float32)
                tv = rng.choice([-1.0, 1.0], n_long * bs).astype(np.float32)
                e.set_option("train_epochs", 1)
                e.set_option("train_batch", bs)
                e.train(0, 1, tb[: 4 * bs], tp[: 4 * bs], tv[: 4 * bs])

                def call(nb):
                    t1 = time.perf_counter()
                    e.train(0, 1, tb[: nb * bs], tp[: nb * bs], tv[: nb * bs])
                    return time.perf_counter() - t1
                t_short, t_long = call(n_short), call(n_long)
                step_s = (t_long - t_short) / (n_long - n_short)          # one optimisation step
                fixed_s = max(0.0, t_short - n_short * step_s)            # per call: upload, graph capture, the trained model's inference tables
                line["nnet_train"] = {"ms_per_step": step_s * 1e3, "samples_per_sec": bs / step_s, "batch": bs,
                                      "steps": [n_short, n_long], "call_ms": [t_short * 1e3, t_long * 1e3], "fixed_ms_per_call": fixed_s * 1e3,
                                      "ms_per_step_of_the_200_step_call": t_short / n_short * 1e3,
                                      "dtype": "f32 parameters / activations / gradients; conv2..conv4 forward as f16 x 3 on the f16 matrix cores, dgrad / wgrad as "
                                               "bf16 x 3 on the bf16 matrix cores, f32 accumulate (gradients within 1e-5 of float64 autograd); "
                                               "conv1 / fc forward on v_mfma_f32_16x16x4_f32",
                                      "tflops": 3 * FLOP_PER_LEAF * bs / step_s / 1e12,
                                      "note": "two az_net_train calls (200 and 800 steps of one epoch): ms_per_step is the slope, i.e. one optimisation step; "
                                              "fixed_ms_per_call is what a call costs besides its steps (sample upload, graph capture, the stored model's "
                                              "BatchNorm fold + conv tables + packed weights) -- the recipe's call has 10 x n / 64 steps (25,600 at the "
                                              "example's literals), so rounds 1-3's figure, the 200-step call divided by 200, overstated a step by "
                                              "fixed / 200; it is kept as ms_per_step_of_the_200_step_call. ~3x forward FLOPs per sample (f32-equivalent)"}
            except Exception as ex:
                line["nnet_train"] = {"error": repr(ex)}
        if world == 1 and args.net == "conv" and not args.no_aux:
            # auxiliary, outside the timed region, not part of `value`
            # (a) the same workload with every requested row executed (de-duplication off): what round 1 measured
            if not args.no_profile:
                e.set_option("profile", 1)
            try:
                e.set_option("eval_dedup", 0)
                e.reset_stats()
                t1 = time.perf_counter()
                r = e.selfplay(n_games=args.games, concurrent=args.games, num_sims=args.sims, model_id=0, seed=args.seed,
                               first_game_id=10**9, symmetries=False, want_boards=False, out=out)
                torch.cuda.synchronize()
                dta = time.perf_counter() - t1
                sa = e.stats()
                line["no_dedup"] = {"games_per_sec": args.games / dta, "episodes": args.games, "leaf_evals_per_sec": sa["leaf_evals"] / dta,
                                    "mfma_fraction_end_to_end": sa["leaf_rows_executed"] / dta * flop_exec / (MFMA_PEAK_TFLOPS * 1e12),
                                    "conv3_tflops": sa["net_conv3_flops"] / (sa["net_conv3_ms"] * 1e-3) / 1e12 if sa["net_conv3_ms"] else None,
                                    "note": "one episode batch of --games episodes (no refill), eval_dedup = 0"}
            except Exception as ex:
                line["no_dedup"] = {"error": repr(ex)}
            finally:
                e.set_option("eval_dedup", args.dedup)
            # (a2) the same workload with conv2 as the MFMA implicit GEMM (round 1's dominant kernel), de-duplication on
            try:
                e.set_option("conv2_table", 0)
                e.reset_stats()
                t1 = time.perf_counter()
                r = e.selfplay(n_games=args.games, concurrent=args.games, num_sims=args.sims, model_id=0, seed=args.seed,
                               first_game_id=2 * 10**9, symmetries=False, want_boards=False, out=out)
                torch.cuda.synchronize()
                dta = time.perf_counter() - t1
                sa = e.stats()
                line["mfma_conv2"] = {"games_per_sec": args.games / dta, "episodes": args.games,
                                      "mfma_fraction_end_to_end": sa["leaf_rows_executed"] / dta * FLOP_PER_LEAF / (MFMA_PEAK_TFLOPS * 1e12),
                                      "conv2_tflops": sa["net_conv2_flops"] / (sa["net_conv2_ms"] * 1e-3) / 1e12 if sa["net_conv2_ms"] else None,
                                      "conv2_frac_of_mfma_peak": sa["net_conv2_flops"] / (sa["net_conv2_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if sa["net_conv2_ms"] else None,
                                      "conv2_avg_launch_ms": sa["net_conv2_ms"] / max(1, sa["net_launches"]),
                                      "note": "one episode batch of --games episodes (no refill), conv2_table = 0: k_conv_same_pipe<1,true> on the matrix cores"}
            except Exception as ex:
                line["mfma_conv2"] = {"error": repr(ex)}
            finally:
                e.set_option("conv2_table", args.conv2_table)
            # (b) BASELINE config 3: arena.rs head-to-head, 4096 paired games new-vs-old net, 400 sims/move, two seeded bf16 nets.
            # On an engine of its own WITHOUT the profiling brackets: the arena is a chain of 33,600 tiny dependent steps whose pace
            # the host's launch calls set, so its searches run as hipGraph launches -- which the brackets of profile mode preclude.
            try:
                ea = azeng.Engine(device=local_rank, max_batch=4096, net_channels=args.channels, profile=False)
                ea.net_set_params(0, e.net_get_params(0))
                ea.net_init_random(2, seed=args.seed + 1)
                # warm-up with the SAME shape (like the warm-up steps of the self-play measurement): the tree arenas of 2 x 4096 trees are
                # allocated and the search graphs captured once per shape and kept (Coach::learn plays an arena of this shape every iteration)
                ea.arena(num_games=4096, num_sims=400, new_model_id=2, old_model_id=0, seed=args.seed + 7)
                ea.reset_stats()
                t1 = time.perf_counter()
                wld, _res = ea.arena(num_games=4096, num_sims=400, new_model_id=2, old_model_id=0, seed=args.seed)
                dta = time.perf_counter() - t1
                sa = ea.stats()
                ea.close()
                line["arena"] = {"games_per_sec": 4096 / dta, "seconds": dta, "games": 4096, "sims_per_move": 400, "wld_new_model": [int(x) for x in wld],
                                 "simulations_per_sec": sa["simulations"] / dta,
                                 "leaf_rows_executed_over_requested": sa["leaf_rows_executed"] / max(1, sa["leaf_rows_requested"]),
                                 "note": "after one warm-up arena of the same shape (arena allocation, graph capture); temp 0 from the first move "
                                         "(src/coach.rs:356-372): games of one seating differ only where the tie-break RNG does, so most leaf "
                                         "rows are duplicates the evaluation cache answers"}
            except Exception as ex:
                line["arena"] = {"error": repr(ex)}
            # (c) BASELINE config 1 as the fine-grained drop-in: one az_tree of ONE game behind az_host::AsyncMcts::get_action_prob,
            # Coach::execute_episode move by move in the C++ host (examples/config1_dropin.cpp), 25 sims/move, stub net and conv net;
            # beside it the oracle's AsyncMcts with the stub net on one host core (the reference's own configuration, examples/connect_four.rs:55-71)
            try:
                line["config1_dropin"] = config1_dropin()
            except Exception as ex:
                line["config1_dropin"] = {"error": repr(ex)}
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    e.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
