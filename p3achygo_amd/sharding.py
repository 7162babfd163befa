"""Multi-GPU sharding of the self-play path: one process per GPU, games sharded across ranks,
no data-path collective (the reference launches one selfplay process per GPU,
python/rl_loop/selfplay.py:51-64).  torch.distributed (gloo) is used only for the timing
barrier, the max-over-ranks clock and the sum of per-rank counters that bench.py reports."""
import os
from dataclasses import dataclass


@dataclass
class Shard:
    rank: int
    local_rank: int
    world: int

    @property
    def is_root(self) -> bool:
        return self.rank == 0


def shard_from_env() -> Shard:
    return Shard(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                 int(os.environ.get("WORLD_SIZE", "1")))


def init(shard: Shard) -> None:
    if shard.world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=shard.rank, world_size=shard.world)


def finish(shard: Shard) -> None:
    if shard.world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def barrier(shard: Shard) -> None:
    if shard.world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(shard: Shard, value: float) -> float:
    if shard.world <= 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sum_over_ranks(shard: Shard, values):
    import torch
    t = torch.tensor(list(values), dtype=torch.float64)
    if shard.world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t]


def games_for_rank(total_games: int, shard: Shard) -> int:
    """Weak scaling keeps per-rank games fixed; this helper is for strong-scaling callers."""
    return total_games // shard.world + (1 if shard.rank < total_games % shard.world else 0)


def seed_for_rank(base_seed: int, shard: Shard) -> int:
    """Per-process seed: the reference derives it from the worker id (selfplay/main.cc:244)."""
    return base_seed + 7919 * shard.rank
