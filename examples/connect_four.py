#!/usr/bin/env python3
"""The reference's example binary (examples/connect_four.rs:53-77) on the MI355X engine: the same 15 Coach::setup
literals by default (25 sims, 1 episode, 40 arena games, stub net when --net stub), or a real run with the bf16
conv net (--net conv) and bigger numbers.  Needs a GPU."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from alphazero_rs_amd import engine as azeng            # noqa: E402
from alphazero_rs_amd.coach import Coach                # noqa: E402
from alphazero_rs_amd.trainer import Trainer            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default="./checkpoint")
    ap.add_argument("--net", default="conv", choices=["conv"])
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--iters", type=int, default=1)          # num_iters, examples/connect_four.rs:65
    ap.add_argument("--eps", type=int, default=1)            # num_eps, :66
    ap.add_argument("--sims", type=int, default=25)          # num_sims, :67
    ap.add_argument("--arena", type=int, default=40)         # num_arena_games, :64
    ap.add_argument("--slots", type=int, default=8192)       # concurrent games (num_episode_threads, :63)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--trainer", default="engine", choices=["engine", "torch"])   # NNet::train: az_net_train or autograd
    ap.add_argument("--epochs", type=int, default=10)        # connect_four_net.py:13
    a = ap.parse_args()
    e = azeng.Engine(device=0, max_batch=max(a.slots, a.arena, 128), net_channels=a.channels)
    e.net_init_random(0, a.seed)
    e.set_option("train_epochs", a.epochs)
    coach = Coach.setup(e, a.checkpoint,
                        1000000,   # mcts_reserve_size
                        0.6,       # update_threshold
                        15,        # temp_threshold
                        20,        # max_history_length
                        200000,    # max_queue_length
                        1,         # inference_batch_size
                        a.slots,   # num_episode_threads -> concurrent game slots
                        a.arena,   # num_arena_games
                        a.iters,   # num_iters
                        a.eps,     # num_eps
                        a.sims,    # num_sims
                        1,         # num_sim_threads
                        1000,      # max_depth
                        1,         # cpuct
                        trainer=Trainer(channels=a.channels, epochs=a.epochs) if a.trainer == "torch" else None)
    for r in coach.learn(skip_first_play=False, seed=a.seed):
        print(r["iteration"], "samples", r["samples"], "new/prev/draw", r["nwins"], r["pwins"], r["draws"],
              "accepted" if r["accepted"] else "rejected", "loss", r["losses"][-1],
              "seconds", {k: round(v, 2) for k, v in r["seconds"].items()})
    e.close()


if __name__ == "__main__":
    main()
