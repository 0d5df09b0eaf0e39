"""One process per GPU: shard self-play episodes by GLOBAL game id and gather the training tuples.

Self-play games are independent (the reference fans episodes out over a thread pool, src/coach.rs:241-272),
so the path shards with NO data-path collective during play.  The only exchange is at episode-batch end:
one all-gather of per-rank tuple counts (8 B each) and ONE gather of the packed (s, pi, z) payload --
torch.distributed backend "nccl" is RCCL over xGMI on ROCm; the CPU tests run the same code on "gloo".

Packed tuple = 12 x int32 = 48 bytes: state (2 x u64) | pi (7 x f32) | z (f32).  Symmetries are
regenerated at the destination (mirror + reversed pi), halving the bytes on the links.
"""
import numpy as np
import torch
import torch.distributed as dist

TUPLE_WORDS = 12


def shard_range(n_total, rank, world):
    """Global game ids [lo, hi) owned by `rank`: contiguous blocks, remainder spread over the first ranks.
    RNG and results are keyed on the global id, so any world size produces the same games."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_samples(states, pis, zs):
    """states [n,2] u64, pis [n,7] f32, zs [n] f32 (torch tensors, any device) -> [n,12] int32."""
    n = zs.shape[0]
    out = torch.empty((n, TUPLE_WORDS), dtype=torch.int32, device=zs.device)
    out[:, 0:4] = states.contiguous().view(torch.int32).reshape(n, 4)
    out[:, 4:11] = pis.contiguous().view(torch.int32)
    out[:, 11] = zs.contiguous().view(torch.int32)
    return out


def unpack_samples(packed):
    n = packed.shape[0]
    states = packed[:, 0:4].contiguous().view(torch.int64).reshape(n, 2)
    pis = packed[:, 4:11].contiguous().view(torch.float32)
    zs = packed[:, 11].contiguous().view(torch.float32)
    return states, pis, zs


def gather_samples(packed, dst=0, group=None):
    """ONE gather of the ragged per-rank [n_r,12] int32 payloads to `dst` (rank order == global game-id order).
    Returns (packed_all [sum n_r, 12], counts [world]) on dst, (None, counts) elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = packed.device
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([packed.shape[0]], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    cmax = int(counts.max().item())
    padded = torch.zeros((cmax, TUPLE_WORDS), dtype=torch.int32, device=dev)
    padded[: packed.shape[0]] = packed
    if rank == dst:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, bufs, dst=dst, group=group)
        out = torch.cat([bufs[r][: int(counts[r])] for r in range(world)], dim=0)
        return out, counts
    dist.gather(padded, None, dst=dst, group=group)
    return None, counts


def expand_symmetries(states, pis, zs):
    """get_symmetries at the destination (connect_four_game.rs:205-211): identity + left-right mirror with
    reversed pi, interleaved as the reference emits them.  states int64 [n,2] (bit patterns of the u64 boards)."""
    def mirror(b):
        r = torch.zeros_like(b)
        for c in range(7):
            r |= ((b >> (c * 7)) & 0x7F) << ((6 - c) * 7)
        return r
    n = zs.shape[0]
    s2 = torch.stack([states, mirror(states)], dim=1).reshape(2 * n, 2)
    p2 = torch.stack([pis, pis.flip(1)], dim=1).reshape(2 * n, 7)
    z2 = torch.stack([zs, zs], dim=1).reshape(2 * n)
    return s2, p2, z2
