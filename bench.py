#!/usr/bin/env python3
"""Benchmark of the hot path: b12c256btl3 policy/value-net evaluation of leaf positions.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the engine's forward path (init conv -> 12 residual blocks ->
heads) over one batch of 1024 synthetic 19x19 positions per GPU (BASELINE.json configs[2]:
"v3-b12c256btl3 on 1 MI355X, 1024 concurrent games, batch=1024, fp16"), inputs already
resident in HBM when the timed region starts.  Games shard embarrassingly across GPUs
(one engine + one HIP stream per device, no collective on the data path), so scaling is
weak: every rank evaluates its own 1024 positions.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel k_block: conv FLOPs (3x3s + 1x1 reduce/expand) per
launch / HIP-event time, against the 2.5 PFLOP/s dense fp16 MFMA peak) and `cpu_baseline`
(the CPU fp32 oracle timed on this box's host cores over a bounded sample).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

MODEL = "b12c256btl3"
BATCH = 1024
PEAK_FP16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16/FP16 MFMA ~2.5 PF


def make_positions(n, seed):
    from p3achygo_amd import features
    base = features.random_positions(64, seed=seed, n_games=16)
    reps = (n + len(base) - 1) // len(base)
    return np.tile(base, reps)[:n].copy()


def cpu_baseline(path, pos, budget_s=12.0):
    """CPU fp32 oracle (oracle/nn_oracle.c, a port: the reference TF-CPU engine is stale and
    unbuildable) on all host cores, bounded to ~budget_s seconds."""
    from oracle import oracle
    # the GPU box gives a 1-GPU job a 16-CPU share; never oversubscribe it
    cores = min(len(os.sched_getaffinity(0)), 16)
    net = oracle.OracleNet(path)
    t0 = time.perf_counter()
    net.forward_features(pos[:cores], nthreads=cores)
    t1 = time.perf_counter() - t0
    rounds = max(1, min(64, int(budget_s / max(t1, 1e-3))))
    n = cores * rounds
    sample = pos[np.arange(n) % len(pos)]
    t0 = time.perf_counter()
    net.forward_features(sample, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "positions/s", "cores": cores, "kind": "port",
            "sample": f"{n} positions of the same batch, {MODEL} fp32 direct conv, {dt:.1f}s"}


def hbm_traffic(kernel_name, positions):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied; see
    profiles/r01_hbm_fetch_write_pmc.txt).  PMC collection needs the profiler around the whole
    process, so the number is a recorded measurement of this kernel at this batch size, not
    re-measured inside the timed run; null when the profile does not match."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_k_block_hbm_traffic.json")) as f:
            prof = json.load(f)
    except OSError:
        return None
    if prof.get("kernel") != kernel_name or prof.get("positions_per_launch") != positions:
        return None
    return prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--model", default=MODEL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-selfplay", action="store_true")
    ap.add_argument("--selfplay-seconds", type=float, default=8.0)
    args = ap.parse_args()

    from p3achygo_amd import sharding
    shard = sharding.shard_from_env()
    rank, local_rank, world = shard.rank, shard.local_rank, shard.world
    if os.environ.get("P3_BENCH_SINGLE_DEVICE"):   # rehearsal of the N > 1 path on a one-GPU box
        local_rank = 0
    import torch
    sharding.init(shard)
    n_gpus = max(world, 1)

    from p3achygo_amd import engine, netspec
    cfg = netspec.CONFIGS[args.model]
    tmp = tempfile.mkdtemp(prefix="p3bench")
    path = os.path.join(tmp, f"{args.model}.p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))  # fresh-Keras random init
    pos = make_positions(args.batch, seed=1000 + rank)

    torch.cuda.set_device(local_rank)
    eng = engine.create_engine(engine.kind_from_engine_path(path), path, args.batch, 1,
                               device=local_rank)
    eng.load_all(pos)
    eng.upload()  # inputs resident in HBM before the timed region

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        sharding.barrier(shard)

    for _ in range(args.warmup):
        eng.forward_resident(args.batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.forward_resident(args.batch)
    eng.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = sharding.max_over_ranks(shard, dt)
    sharding.barrier(shard)

    total_flops, conv3_flops = eng.flops_per_position()
    eng.close()

    # ---- the same engine driven by the real self-play host (PCIe-inclusive; reported
    # beside `value`, never as `value`): 2 x batch concurrent games, Gumbel n=32 ----------
    selfplay = None
    if not args.no_selfplay:
        from p3achygo_amd import host_api
        cpus = len(os.sched_getaffinity(0))
        threads = max(2, min(16, cpus // max(world, 1)))
        GROUPS = 4   # game groups: three forward passes queued on the GPU while one group is on the host
        host_api.set_groups(GROUPS)
        rates, sp_err, secs = [0.0, 0.0], None, 0.0
        try:   # a side measurement must never cost the headline line
            st = host_api.selfplay_run(path, GROUPS * args.batch, threads, args.selfplay_seconds, default_n=32,
                                       default_k=5, selected_n=32, selected_k=5, warmup_batches=4 * GROUPS,
                                       seed=sharding.seed_for_rank(77, shard), device=local_rank)
            rates, secs = [st.positions / st.seconds, st.moves / st.seconds], st.seconds
        except Exception as ex:   # noqa: BLE001
            sp_err = repr(ex)
        sp = sharding.sum_over_ranks(shard, rates)
        selfplay = {"value": float(sp[0]), "unit": "positions/s", "moves_per_s": float(sp[1]),
                    "concurrent_games_per_gpu": GROUPS * args.batch, "batch": args.batch, "game_groups": GROUPS,
                    "host_threads_per_gpu": threads, "seconds": secs,
                    "gumbel": "n=32 (default k<=5, selected k=5)", "includes": "host MCTS + PCIe + engine"}
        if sp_err:
            selfplay["error"] = sp_err
    eng = engine.create_engine(engine.kind_from_engine_path(path), path, args.batch, 1, device=local_rank)
    eng.load_all(pos)
    eng.upload()
    eng.forward_resident(args.batch)
    eng.sync()
    roof = None
    cpu = None
    if rank == 0:
        ms, flops_launch, kname = eng.time_trunk_kernel(args.batch, 10)
        achieved = flops_launch / (ms * 1e-3) / 1e12
        prof = hbm_traffic(kname, args.batch)
        roof = {"bound": "mfma", "kernel": kname, "achieved": achieved,
                "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP16_MFMA_TFLOPS,
                "traffic": prof["traffic_bytes_per_launch"] if prof else None,   # HBM bytes per launch
                "launch_ms": ms, "algorithmic_flops_per_launch": flops_launch}
        if prof:
            roof["algorithmic_bytes_per_launch"] = prof["algorithmic_bytes_per_launch"]
            roof["traffic_source"] = prof["source"]
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(path, pos)
            except Exception as ex:   # noqa: BLE001
                cpu = {"error": repr(ex)}
    sharding.barrier(shard)

    if rank == 0:
        pps = n_gpus * args.batch * args.steps / dt
        out = {
            "metric": "self-play positions/sec at 1/8 MI355X, b12c256btl3 19x19 n=32",
            "value": pps, "unit": "positions/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.model} random-init, NN evaluation of {args.batch} "
                                   "leaf positions per GPU per step (engine forward, inputs "
                                   "resident in HBM; leaf positions from seeded random-legal "
                                   "playouts)",
                       "batch_per_gpu": args.batch, "parallelism": f"games sharded x{n_gpus}, no collective"},
            "full_net_tflops": pps * total_flops / 1e12,
            "conv3x3_mfma_frac_end_to_end": pps * conv3_flops / 1e12 / (PEAK_FP16_MFMA_TFLOPS * n_gpus),
            "roofline": roof, "cpu_baseline": cpu, "selfplay": selfplay,
        }
        print(json.dumps(out))
    eng.close()
    sharding.finish(shard)


if __name__ == "__main__":
    main()
