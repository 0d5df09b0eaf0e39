"""Coach::learn around the MI355X engine -- SURVEY.md section 8(f1, f3).

Mirrors src/coach.rs: `Coach.setup` takes the reference's 15 parameters (src/coach.rs:38-54) with the same
meaning, `learn` runs the iteration loop of src/coach.rs:169-396: self-play episodes -> replay window
(max_queue_length / max_history_length) -> save examples -> shuffle -> NNet::train -> arena of new vs old ->
accept iff nwins + pwins > 0 and nwins / (nwins + pwins) >= update_threshold (:383-390).  Self-play and the arena
are ONE engine call each (az_selfplay / az_arena); episodes shard across ranks by global game id when a process
group is active and the tuples are gathered once per iteration (alphazero-rs_amd/dist.py).

NNet::train is the engine's own (az_net_train, csrc/az_train.hip) unless a `trainer` object is passed (the PyTorch
autograd restatement of alphazero-rs_amd/trainer.py: what the CPU tests use).  Several ranks are REPLICAS for training:
every rank trains on the same gathered samples with the same seeds and, the kernels being deterministic, ends with the
same weights -- no gradient exchange.
The C++ host (include/az_host.hpp `Coach`) runs the same sequence with the same seeds and writes the same files.

On-disk formats (the reference's are bincode / TF checkpoints, src/coach.rs:159-167 with defect A14; these are the
build's own, documented here):
  <dir>/<iter>.examples   "AZEX0001": char magic[8]; int64 H; int64 lens[H] (samples per history entry);
                          f32 boards[N][2][6][7]; f32 pis[N][7]; f32 vs[N] -- the whole `history` deque, oldest first
  <dir>/<model_id>.aznet  weights file of az_net_save: the initial model and every candidate
  <dir>/coach.state       text "iteration model_id\n": the last finished iteration and the ACCEPTED model id after its gate
                          (a rejected candidate leaves its <id+1>.aznet behind; the state file says which id is live)
Resume picks the largest numeric stem, as Coach::setup does (:55-81); non-numeric files are ignored instead of
panicking; when coach.state names a model whose .aznet exists it is loaded into that engine slot, so iteration k+1 of a
restarted run is byte-identical to an uninterrupted one.
"""
import collections
import os
import time

import numpy as np


class Coach:
    def __init__(self):
        raise TypeError("use Coach.setup(...)")

    @classmethod
    def setup(cls, engine, checkpoint_directory, mcts_reserve_size, update_threshold, temp_threshold,
              max_history_length, max_queue_length, inference_batch_size, num_episode_threads, num_arena_games,
              num_iters, num_eps, num_sims, num_sim_threads, max_depth, cpuct, trainer=None, group=None, log=print):
        self = object.__new__(cls)
        if num_sims % inference_batch_size != 0:                      # assert!, src/coach.rs:83
            raise ValueError("num_sims % inference_batch_size != 0")
        if num_sim_threads < 1 or num_sims % num_sim_threads != 0:    # assert!, src/async_mcts.rs:192
            raise ValueError("num_sims % num_sim_threads != 0")
        self.num_sim_threads = num_sim_threads    # > 1: several simulations in flight per tree (the engine's lock-step schedule)
        self.engine, self.trainer, self.group, self.log = engine, trainer, group, log
        self.dir = str(checkpoint_directory)
        self.mcts_reserve_size, self.update_threshold, self.temp_threshold = mcts_reserve_size, update_threshold, temp_threshold
        self.max_history_length, self.max_queue_length = max_history_length, max_queue_length
        self.num_episode_threads = num_episode_threads    # = concurrent game slots on the GPU (rayon pool size, :202-205)
        self.num_arena_games, self.num_iters, self.num_eps, self.num_sims = num_arena_games, num_iters, num_eps, num_sims
        self.max_depth, self.cpuct = max_depth, cpuct
        self.history = collections.deque()
        self.start_iteration = 0
        os.makedirs(self.dir, exist_ok=True)
        stems = [int(f[:-9]) for f in os.listdir(self.dir) if f.endswith(".examples") and f[:-9].isdigit()]
        self.model_id = 0
        if stems:                                                      # src/coach.rs:55-81
            it = max(stems)
            self.history.extend(load_examples(os.path.join(self.dir, f"{it}.examples")))
            self.start_iteration = it + 1
            st = read_state(self.dir)
            if st is not None and st[0] == it:                         # the live model of the run being resumed
                self.model_id = st[1]
                path = os.path.join(self.dir, f"{self.model_id}.aznet")
                if os.path.exists(path):
                    engine.net_load(self.model_id, path)
        return self

    # ---- rank helpers -------------------------------------------------------------------------------------
    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def save_train_examples(self, iteration):
        """src/coach.rs:159-167 (A14 repaired: <dir>/<iter>.examples, not an absolute path)."""
        path = os.path.join(self.dir, f"{iteration}.examples")
        save_examples(path, self.history)
        return path

    def execute_episodes(self, model_id, iteration, seed):
        """The self-play fan-out of src/coach.rs:241-272: num_eps x execute_episode, sharded by global game id."""
        from . import dist as azdist
        rank, world = self._world()
        lo, hi = azdist.shard_range(self.num_eps, rank, world)
        first = iteration * self.num_eps
        if hi > lo:
            r = self.engine.selfplay(n_games=hi - lo, num_sims=self.num_sims, model_id=model_id, seed=seed,
                                     first_game_id=first + lo, concurrent=min(self.num_episode_threads, hi - lo),
                                     temp_threshold=self.temp_threshold, max_depth=self.max_depth, cpuct=self.cpuct,
                                     reserve=self.mcts_reserve_size, symmetries=False, want_boards=False,
                                     num_sim_threads=self.num_sim_threads)
            states, pis, zs = r["states"], r["pis"], r["zs"]
        else:
            states, pis, zs = np.zeros((0, 2), np.uint64), np.zeros((0, 7), np.float32), np.zeros(0, np.float32)
        if world > 1:
            import torch
            import torch.distributed as tdist
            dev = torch.device("cuda", torch.cuda.current_device()) if tdist.get_backend(self.group) == "nccl" else torch.device("cpu")
            packed = azdist.pack_samples(torch.from_numpy(states.view(np.int64)).to(dev), torch.from_numpy(pis).to(dev),
                                         torch.from_numpy(zs).to(dev))
            counts = torch.zeros(world, dtype=torch.int64, device=dev)
            tdist.all_gather_into_tensor(counts, torch.tensor([packed.shape[0]], dtype=torch.int64, device=dev), group=self.group)
            cmax = int(counts.max())
            padded = torch.zeros((cmax, azdist.TUPLE_WORDS), dtype=torch.int32, device=dev)
            padded[: packed.shape[0]] = packed
            allp = torch.zeros((world * cmax, azdist.TUPLE_WORDS), dtype=torch.int32, device=dev)
            tdist.all_gather_into_tensor(allp, padded, group=self.group)        # every rank trains: all-gather
            allp = torch.cat([allp[r * cmax: r * cmax + int(counts[r])] for r in range(world)])
            s, p, z = azdist.unpack_samples(allp)
        else:
            import torch
            s, p, z = torch.from_numpy(states.view(np.int64)), torch.from_numpy(pis), torch.from_numpy(zs)
        from . import dist as azd
        s2, p2, z2 = azd.expand_symmetries(s.cpu(), p.cpu(), z.cpu())           # get_symmetries at the destination
        boards = states_to_boards(s2.numpy().view(np.uint64))
        return boards, p2.numpy(), z2.numpy()

    def learn(self, skip_first_play=False, seed=0, model_id=None):
        """src/coach.rs:169-396.  Engine model slots: `model_id` is the current net (default: the resumed run's live
        model, else 0), `model_id + 1` the candidate.  Returns a list of per-iteration dicts (wins, accepted, losses)."""
        rank, world = self._world()
        report = []
        if model_id is None:
            model_id = self.model_id
        first = os.path.join(self.dir, f"{model_id}.aznet")
        if rank == 0 and not os.path.exists(first):
            self.engine.net_save(model_id, first)                       # the run's initial model: what a restart would load
        free = getattr(self.engine, "net_free", None)
        for iteration in range(self.start_iteration, self.start_iteration + self.num_iters):
            t_play = t_train = t_arena = 0.0
            boards, pis, vs = np.zeros((0, 2, 6, 7), np.float32), np.zeros((0, 7), np.float32), np.zeros(0, np.float32)
            if not skip_first_play or iteration > self.start_iteration:
                t0 = time.perf_counter()
                boards, pis, vs = self.execute_episodes(model_id, iteration, seed)
                t_play = time.perf_counter() - t0
                if vs.shape[0] > self.max_queue_length:                 # :275-277: keep the newest max_queue_length
                    boards, pis, vs = boards[-self.max_queue_length:], pis[-self.max_queue_length:], vs[-self.max_queue_length:]
            self.history.append((boards, pis, vs))                      # :282: pushed even when the play was skipped (empty entry)
            if len(self.history) > self.max_history_length:             # :285-288
                self.history.popleft()
            if rank == 0:
                self.save_train_examples(iteration)                     # :291-293
            allb = np.concatenate([h[0] for h in self.history])
            allp = np.concatenate([h[1] for h in self.history])
            allv = np.concatenate([h[2] for h in self.history])
            assert allv.shape[0] > 0                                    # :305
            perm = shuffle_permutation(allv.shape[0], seed, iteration)  # :296-297 shuffle
            allb, allp, allv = allb[perm], allp[perm], allv[perm]
            t0 = time.perf_counter()
            if self.trainer is None:                                    # :329 -> NNet::train(samples, id, id + 1)
                self.engine.set_option("train_seed", seed + iteration)
                losses = self.engine.train(model_id, model_id + 1, allb, allp, allv)
            else:
                prev = self.engine.net_get_params(model_id)
                new = self.trainer.train(prev, allb, allp, allv, seed=seed + iteration)
                self.engine.net_set_params(model_id + 1, new)
                losses = list(self.trainer.history)
            t_train = time.perf_counter() - t0
            t0 = time.perf_counter()
            if rank == 0:
                self.engine.net_save(model_id + 1, os.path.join(self.dir, f"{model_id + 1}.aznet"))
            # arena: new (first listed) vs old, both seatings (:333-375); games sharded by global index across ranks,
            # the W/L/D tally is one 3-counter all-reduce
            total = 2 * (self.num_arena_games // 2)
            a_seed = seed + 7919 * (iteration + 1)
            if world > 1 and total > 0:
                from . import dist as azdist
                import torch
                import torch.distributed as tdist
                lo, hi = azdist.shard_range(total, rank, world)
                wld = np.zeros(3, np.uint64)
                if hi > lo:
                    wld, _ = self.engine.arena(hi - lo, self.num_sims, new_model_id=model_id + 1, old_model_id=model_id,
                                               seed=a_seed, max_depth=self.max_depth, cpuct=self.cpuct,
                                               reserve=self.mcts_reserve_size, first_game=lo, total_games=total,
                                               num_sim_threads=self.num_sim_threads)
                dev = torch.device("cuda", torch.cuda.current_device()) if tdist.get_backend(self.group) == "nccl" else torch.device("cpu")
                t = torch.tensor([int(x) for x in wld], dtype=torch.int64, device=dev)
                tdist.all_reduce(t, group=self.group)
                wld = t.cpu().numpy()
            else:
                wld, _ = self.engine.arena(self.num_arena_games, self.num_sims, new_model_id=model_id + 1, old_model_id=model_id,
                                           seed=a_seed, max_depth=self.max_depth, cpuct=self.cpuct,
                                           reserve=self.mcts_reserve_size, num_sim_threads=self.num_sim_threads)
            t_arena = time.perf_counter() - t0
            nwins, pwins, draws = int(wld[0]), int(wld[1]), int(wld[2])
            self.log(f"NEW/PREV WINS : {nwins} / {pwins}; DRAWS : {draws}")            # :381
            accepted = not (pwins + nwins == 0 or nwins / (pwins + nwins) < self.update_threshold)   # :383-390
            self.log("ACCEPTING NEW MODEL" if accepted else "REJECTING NEW MODEL")
            report.append({"iteration": iteration, "samples": int(allv.shape[0]), "nwins": nwins, "pwins": pwins,
                           "draws": draws, "accepted": accepted, "losses": losses, "model_id": model_id,
                           "seconds": {"selfplay": t_play, "train": t_train, "arena": t_arena}})
            # a long run moves to a new model id per accepted iteration: drop the slot nobody will read again
            if free is not None:
                free(model_id if accepted else model_id + 1)
            if accepted:
                model_id += 1
            if rank == 0:
                write_state(self.dir, iteration, model_id)
            self.model_id = model_id
        return report


def write_state(directory, iteration, model_id):
    tmp = os.path.join(directory, "coach.state.tmp")
    with open(tmp, "w") as f:
        f.write(f"{int(iteration)} {int(model_id)}\n")
    os.replace(tmp, os.path.join(directory, "coach.state"))


def read_state(directory):
    """(iteration, model_id) of <dir>/coach.state, or None."""
    try:
        a, b = open(os.path.join(directory, "coach.state")).read().split()
        return int(a), int(b)
    except (OSError, ValueError):
        return None


def states_to_boards(states):
    """to_features (connect_four_game.rs:219-237, S8) for [N,2] uint64 canonical bitboards -> [N,2,6,7] f32."""
    states = np.asarray(states, np.uint64).reshape(-1, 2)
    out = np.zeros((states.shape[0], 2, 6, 7), np.float32)
    for r in range(6):
        for c in range(7):
            bit = np.uint64(1) << np.uint64(c * 7 + (5 - r))
            out[:, 0, r, c] = (states[:, 0] & bit) != 0
            out[:, 1, r, c] = (states[:, 1] & bit) != 0
    return out


def save_examples(path, history):
    """`history`: iterable of (boards [n,2,6,7], pis [n,7], vs [n]) -> "AZEX0001" file (module docstring)."""
    history = list(history)
    with open(path, "wb") as f:
        f.write(b"AZEX0001")
        np.array([len(history)] + [h[2].shape[0] for h in history], np.int64).tofile(f)
        for i in range(3):
            for h in history:
                np.ascontiguousarray(h[i], np.float32).tofile(f)


def load_examples(path):
    with open(path, "rb") as f:
        if f.read(8) != b"AZEX0001":
            raise ValueError(f"{path}: not an AZEX0001 examples file")
        H = int(np.fromfile(f, np.int64, 1)[0])
        lens = [int(x) for x in np.fromfile(f, np.int64, H)]
        boards = [np.fromfile(f, np.float32, n * 84).reshape(n, 2, 6, 7) for n in lens]
        pis = [np.fromfile(f, np.float32, n * 7).reshape(n, 7) for n in lens]
        vs = [np.fromfile(f, np.float32, n) for n in lens]
    if any(v.shape[0] != n for v, n in zip(vs, lens)):
        raise ValueError(f"{path}: truncated")
    return list(zip(boards, pis, vs))


def _mix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def shuffle_permutation(n, seed, iteration):
    """Fisher-Yates with the build's counter RNG (the reference shuffles with SmallRng, src/coach.rs:296-297; B7):
    for i = n-1 .. 1: j = ((draw(seed, iteration, i, 5) >> 32) * (i + 1)) >> 32; swap(perm[i], perm[j]).
    Same permutation as include/az_host.hpp shuffle_permutation."""
    i = np.arange(n, dtype=np.uint64)
    key = _mix64(_mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) ^ np.uint64(iteration))
    r = _mix64(_mix64(key ^ i) ^ np.uint64(5))
    j = ((r >> np.uint64(32)) * (i + np.uint64(1))) >> np.uint64(32)
    perm = list(range(n))
    for k in range(n - 1, 0, -1):
        t = int(j[k])
        perm[k], perm[t] = perm[t], perm[k]
    return np.array(perm, np.int64)
