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


def test_bench_gpus_flag_starts_that_many_ranks(built):
    """`python bench.py --gpus 2` — the driver's invocation shape when no launcher set WORLD_SIZE — starts two
    rank processes itself (python/rl_loop/sp_loop.py:160-192 starts one self-play process per GPU) and rank 0
    prints one JSON line with n_gpus 2; a WORLD_SIZE that disagrees with --gpus is refused.  Run over the host's
    NullEvaluator (--null-engine: no GPU here), which the line says."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--null-engine", "--steps", "7", "--warmup", "2",
           "--batch", "16", "--groups", "3", "--lanes", "1"]
    one = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run(cmd + ["--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # rank 0 alone prints
    b = json.loads(lines[0])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert a["steps"] == b["steps"] == 7                          # --steps ROUNDS: one batch of each of the 3 groups
    # the window opens and closes on completions of one group, 7 of its batches apart; the other two groups run on
    # their own (over the NullEvaluator nothing paces them to each other — on the GPU the forward passes queue behind
    # one another and every group completes one batch per round): about 7 completions each inside it
    assert 7 + 2 * 4 <= a["engine_batches_completed"] <= 7 + 2 * 14
    assert 2 * (7 + 2 * 4) <= b["engine_batches_completed"] <= 2 * (7 + 2 * 14)    # K rounds per rank, summed
    assert a["positions"] == 16 * a["engine_batches_completed"]   # every batch full
    assert b["positions"] == 16 * b["engine_batches_completed"]   # weak scaling: per-rank work fixed
    assert a["ms_per_step"] == pytest.approx(a["seconds_timed"] / 7 * 1e3)
    assert b["games_past_opening"] == b["games_total"] == 2 * 3 * 16
    assert "NOT a measurement" in b["engine"] and b["roofline"] is None
    bad = subprocess.run(cmd + ["--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True,
                         text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr
