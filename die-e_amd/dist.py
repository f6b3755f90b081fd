"""Multi-GPU plumbing of the self-play path: games are independent, so ranks are independent
data-parallel workers.  No data-path collective exists; torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) only provides the barrier and the reduction of per-rank
timings and counters."""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_first_game_id(rank, games_per_rank):
    """block partition of the global game ids: rank r plays ids [r*G, (r+1)*G) -- the ids key the Philox
    streams, so the union of the shards is the unsharded batch (exactly, with ref_quirks off)"""
    return rank * games_per_rank


def reduce_stats(dist, elapsed, totals, keys, device):
    """MAX of the elapsed time and SUM of the counters over ranks (returns python floats)"""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    v = torch.tensor([float(totals.get(k, 0)) for k in keys], dtype=torch.float64, device=device)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return float(t.item()), {k: float(x) for k, x in zip(keys, v.tolist())}


def gather_counts(dist, count, device):
    """all_gather of one per-rank u64 (the fragment counts of a self-play batch, 8 x u64 on a node): what a
    consumer of the sharded records needs to size its buffers; the records themselves stay on their rank"""
    import torch
    mine = torch.tensor([int(count)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(t.item()) for t in out]
