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


# ---- host-core placement ("total host cores / 8, NUMA placement of game
# threads and pinned staging next to their GPU" is the scaling risk of the 8-GPU run) ----------

def _parse_cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(pci_bus_id: str):
    """NUMA node of the GPU at `pci_bus_id` ("0000:c1:00.0") from sysfs, or None."""
    try:
        with open(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node") as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError):
        return None


def numa_cpus(node: int):
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            return _parse_cpulist(f.read())
    except OSError:
        return []


def cpus_for_rank(allowed, local_rank: int, local_world: int, node_cpus=None, ranks_on_node=None):
    """The host cores of one rank: an even, contiguous share of the cores this process may use.

    `allowed` is sched_getaffinity(0) (on the GPU box: the job's share).  When the GPU's NUMA
    node is known, the rank takes its share from that node's cores: `node_cpus` (intersected with
    `allowed`) divided among the `ranks_on_node` = (position, count) ranks whose GPUs sit there.
    Otherwise the allowed cores are cut into `local_world` contiguous blocks: with the usual
    enumeration (GPUs 0-3 on socket 0, 4-7 on socket 1; cores numbered socket by socket) block r
    is on GPU r's socket.  Never returns an empty set: small machines share cores."""
    allowed = sorted(allowed)
    pool, pos, cnt = allowed, local_rank, max(local_world, 1)
    if node_cpus and ranks_on_node:
        local = [c for c in node_cpus if c in set(allowed)]
        if len(local) >= ranks_on_node[1]:
            pool, (pos, cnt) = local, ranks_on_node
    if len(pool) < cnt:
        return pool
    per = len(pool) // cnt
    return pool[pos * per:(pos + 1) * per]


def bind_rank_to_local_cpus(shard: Shard, pci_bus_ids=None):
    """Pins this process (hence every thread it starts afterwards: the self-play worker pool,
    the per-group GPU threads, HIP's own helper threads) to its rank's share of the host cores,
    next to its GPU when sysfs says where that is.  Pinned staging buffers are then allocated
    (first touch) on that node.  Call before creating engines or thread pools.  Returns the
    chosen core list.  pci_bus_ids: bus ids of all local GPUs in LOCAL_RANK order (optional)."""
    allowed = sorted(os.sched_getaffinity(0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(shard.world)))
    node_cpus = ranks_on_node = None
    if pci_bus_ids and shard.local_rank < len(pci_bus_ids):
        nodes = [gpu_numa_node(b) for b in pci_bus_ids[:local_world]]
        mine = nodes[shard.local_rank]
        if mine is not None:
            same = [r for r, n in enumerate(nodes) if n == mine]
            node_cpus, ranks_on_node = numa_cpus(mine), (same.index(shard.local_rank), len(same))
    cpus = cpus_for_rank(allowed, shard.local_rank, local_world, node_cpus, ranks_on_node)
    if shard.world > 1 and cpus:
        os.sched_setaffinity(0, cpus)
    return cpus
