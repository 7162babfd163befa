"""N > 1 path on CPU (gloo, world_size 2): every rank runs its own shard of self-play games
with the NullEvaluator behind the same host scheduler bench.py uses, the clock is the max over
ranks, the counters add up, and the shards differ (per-rank seeds) — no data-path collective."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time
    from p3achygo_amd import host_api, sharding
    shard = sharding.shard_from_env()
    sharding.init(shard)
    host_api.set_policy(init_state_sampling=False)
    allowed = sorted(os.sched_getaffinity(0))
    cpus = sharding.bind_rank_to_local_cpus(shard)         # rank's share of the host cores
    assert sorted(os.sched_getaffinity(0)) == cpus and set(cpus) <= set(allowed) and cpus
    sharding.barrier(shard)
    t0 = time.perf_counter()
    moves, b, w, evals = host_api.selfplay_one_game(None, 4, 2, 24, sharding.seed_for_rank(5, shard))
    if rank == 1:
        time.sleep(0.3)                      # the slow rank sets the clock
    dt = time.perf_counter() - t0
    dt_max = sharding.max_over_ranks(shard, dt)
    total_evals, total_moves = sharding.sum_over_ranks(shard, [evals, len(moves)])
    sharding.barrier(shard)
    q.put((rank, dt, dt_max, evals, total_evals, len(moves), total_moves, [int(m) for m in moves[:12]],
           sharding.games_for_rank(5, shard), cpus, allowed))
    sharding.finish(shard)


def test_two_rank_gloo_sharding(built):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, max0, e0, te0, m0, tm0, mv0, g0, c0, a0), (r1, dt1, max1, e1, te1, m1, tm1, mv1, g1, c1, a1) = res
    if len(a0) >= 2:                                                          # ranks own disjoint host cores
        assert not set(c0) & set(c1) and len(c0) == len(c1) == len(a0) // 2
    assert (r0, r1) == (0, 1)
    assert max0 == max1 == pytest.approx(max(dt0, dt1)) and max0 >= 0.3      # clock = slowest rank
    assert te0 == te1 == e0 + e1 and tm0 == tm1 == m0 + m1                    # counters add up
    assert mv0 != mv1                                                         # different shards
    assert g0 + g1 == 5 and g0 == 3                                           # uneven split helper


def test_cpu_placement_rules():
    """Per-rank host cores (SURVEY.md section 8e scaling risk): contiguous even shares of the allowed
    cores, taken from the GPU's NUMA node when sysfs names it; never empty."""
    from p3achygo_amd import sharding as s
    assert s.cpus_for_rank(range(128), 5, 8) == list(range(80, 96))
    # GPUs 4-7 on node 1 (cores 64-127): local rank 5 is the second of four ranks there
    assert s.cpus_for_rank(range(128), 5, 8, node_cpus=list(range(64, 128)), ranks_on_node=(1, 4)) == list(range(80, 96))
    # a node that offers fewer allowed cores than ranks falls back to the plain split
    assert s.cpus_for_rank(range(16), 1, 2, node_cpus=[99], ranks_on_node=(0, 2)) == list(range(8, 16))
    assert s.cpus_for_rank(range(4), 5, 8) == [0, 1, 2, 3]
    assert s._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert s.gpu_numa_node("ffff:ff:1f.0") is None
