#!/usr/bin/env python3
"""Benchmark of the hot path: b12c256btl3 self-play, leaf positions evaluated per second.

    python bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json metric): leaf positions evaluated per second through
`p3hip_run` while the C++ self-play host (libp3host.so: Gumbel MCTS n=32, thousands of concurrent
19x19 games) drives the HIP engine through its C ABI — host search + PCIe + engine, random-init
b12c256btl3 weights, 1024 positions per engine batch (BASELINE.json configs[2]).  The host keeps GROUPS
game groups in flight (GROUPS - 1 forward passes queued on the GPU while one group is on the host), and
one "step" = one ROUND: one engine batch (`p3hip_run` over a group's 1024 leaf positions) of every
group, GROUPS x 1024 positions (round 4; rounds 1-3 timed single batches, and a window of 20 batch
completions on eight interleaving streams opened and closed at arbitrary phases of the streams' cycle:
BENCH_r03's 263.9 k exceeded what its own kernel time allowed).
Untimed: an advance phase that plays every game past its raw-policy opening (up to 30 moves of ONE
evaluation each, self_play_thread.cc:44,363-366; at most --advance-limit batches per group), so that
the timed steps are steady-state Gumbel n=32 search whatever K is, then W warm-up rounds (one batch per
group each).  Then exactly K rounds are timed inside the host: the window opens at the completion of the
last warm-up batch — that group is the ANCHOR — and closes at the completion of the anchor's K-th batch
after it (its own steady clock; every engine run ends with its stream drained), so both ends sit at the
same phase of the groups' cycle; every batch of any group that completes inside is counted (K per group in
steady state).  The line flags itself (`window_artifact`) if its per-batch time is below the dominant
kernel's own launch time.  Barriers bracket the region; the slowest rank's time is the job's time.  Games shard embarrassingly across GPUs (one
process, one engine set, one HIP stream set per device; no data-path collective): weak scaling.
`--gpus N` without WORLD_SIZE in the environment starts the N rank processes itself (before anything
touches the GPU); under torch.distributed.run it checks N against WORLD_SIZE.

Secondary figures in the same JSON line:
  engine_only  — forward pass alone over a resident batch (no host, no PCIe), with the shader clock and socket power
                 amdsmi reports while it loops (`chip_during_loop`);
  engine_benchmark — the reference's nn::Benchmark loop (benchmark_engine.cc:77-108) from C++ over the C ABI;
  as_stated_c3 — BASELINE configs[2] as written: ONE group of 1024 games, batch 1024, the group filling two engine
                 batches in turn with up to 4 playouts of a game's search in flight (round 4: same games, DESIGN.md
                 section 5); `one_lane` inside it = host and GPU alternating, the figure of rounds 1-3;
  roofline     — dominant kernel k_block: conv FLOPs (3x3s + 1x1 reduce/expand) per launch / HIP-event
                 time on the engine's stream, against the 2.5 PFLOP/s dense fp16 MFMA peak;
  cpu_baseline — the CPU fp32 oracle on this box's host cores over a bounded sample: engine-only
                 (`value`), and behind the same self-play host through the same C ABI (`selfplay`).
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
DEFAULT_GROUPS = GROUPS = 8          # game groups per GPU: forward passes of the others queued while one group is on the host
                                     # (round 4: ONE group filling two engine batches in turn — `--groups 1 --lanes 2`, the stated 1024 games
                                     # — measures 219-225 k where eight one-lane groups measure 220-228 k on the same box; the headline stays
                                     # at eight groups because one slow game (an exact ladder read-out) stalls a lone group's pipeline and a
                                     # 20-round window is then 0.2 s long; both are in every line)
                                     # (round 2: 4 / 6 / 8 / 12 groups 217 / 222 / 220 / 220 k positions/s on one box; round 3, the forward
                                     # pass one long launch: 4 / 6 / 8 groups 225 / 237 / 241 k with 241 k engine-only, and the driver's
                                     # 20-step window steadier with eight: 239-242 k against 209-242 k)
PEAK_FP16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16/FP16 MFMA ~2.5 PF
MEASURED_PIPE_CEILING_TFLOPS = 1869.0   # round 4, profiles/r04_mfma_energy_probe.txt: pure 16x16x32 MFMA stream on real-like values
LADDER_BUDGET = 0       # ladder read-out work bound of the host: 0 = the reference's exact read-out (default)


def make_positions(n, seed):
    from p3achygo_amd import features
    base = features.random_positions(64, seed=seed, n_games=16)
    reps = (n + len(base) - 1) // len(base)
    return np.tile(base, reps)[:n].copy()


def cpu_baseline(path, pos, model, budget_s=10.0):
    """CPU fp32 oracle (oracle/nn_oracle.c, a port: the reference TF-CPU engine is stale and
    unbuildable) on the host cores, bounded to ~budget_s seconds per leg."""
    from oracle import oracle
    from p3achygo_amd import host_api
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
    out = {"value": n / dt, "unit": "positions/s", "cores": cores, "kind": "port",
           "what": "engine only: CPU fp32 oracle forward pass, no search",
           "sample": f"{n} positions of the same batch, {model} fp32 direct conv, {dt:.1f}s"}
    # the same self-play host, the CPU engine bound through the same C ABI:
    # 2 groups x `cores` games (every batch gives each core one position), the same advance past the
    # raw-policy openings as the headline run, then the measured batches
    try:
        cpu_lib = os.path.join(ROOT, "oracle", "libp3cpu_engine.so")
        os.environ["P3CPU_THREADS"] = str(cores)
        games = 2 * cores
        per_batch_s = (games / 2) / max(out["value"], 1e-9)
        steps = max(4, int(budget_s / max(per_batch_s, 1e-3)))
        host_api.set_groups(2)
        host_api.set_step_limit(steps)
        host_api.set_advance_limit(32)
        st = host_api.selfplay_run(path, games, cores, 0.0, default_n=32, default_k=5, selected_n=32, selected_k=5,
                                   warmup_batches=1, seed=77, engine_lib=cpu_lib)
        out["selfplay"] = {"value": st.positions / st.seconds, "unit": "positions/s", "cores": cores,
                           "what": "the same self-play host (Gumbel n=32) over the CPU fp32 engine through the "
                                   "same C ABI (oracle/libp3cpu_engine.so)",
                           "evals_per_move": st.positions / max(st.moves, 1),
                           "games_past_opening": st.games_past_opening, "advance_batches_untimed": st.advance_batches,
                           "sample": f"{games} concurrent games, {st.batches} engine batches of {games // 2}, "
                                     f"{st.positions} positions, {st.seconds:.1f}s"}
    except Exception as ex:   # noqa: BLE001
        out["selfplay"] = {"error": repr(ex)}
    finally:
        host_api.set_step_limit(0)
        host_api.set_advance_limit(0)
    return out


def hbm_traffic(kernel_name, positions):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied).  PMC collection
    needs the profiler around the whole process, so the number is a RECORDED measurement of this
    kernel at this batch size, not re-measured inside the timed run; null when no profile of this
    build's kernel matches."""
    for name in ("r04_k_block_hbm_traffic.json", "r03_k_block_hbm_traffic.json", "r02_k_block_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                prof = json.load(f)
        except OSError:
            continue
        if prof.get("kernel") == kernel_name and prof.get("positions_per_launch") == positions:
            return prof
    return None


def measure_hbm_traffic(model, batch, kernel_prefix):
    """HBM bytes per launch of the dominant kernel, measured now: two child runs of three forward passes
    under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes;
    the counters need the profiler around a whole process, hence children of this one).  gfx950
    correction: every read of these kernels is a 16-byte-per-lane access, which FETCH_SIZE tallies at
    half its bytes.  Returns (bytes, detail) or (None, reason)."""
    import csv, glob, shutil, subprocess
    if batch != 1024:
        return None, "the counter run uses batches of 1024"
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "already running under a profiler"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="p3pmc_")
        try:
            subprocess.run([exe, "--output-format", "csv", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--",
                            sys.executable, os.path.join(ROOT, "tools", "gpu_run_forward.py"), "3", model],
                           cwd=tempfile.gettempdir(), env=dict(os.environ, TMPDIR=tempfile.gettempdir()),
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180, check=True)
            xs = []
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for r in csv.DictReader(f):
                        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith(kernel_prefix):
                            xs.append(float(r["Counter_Value"]))
            if not xs:
                return None, f"no {counter} rows for {kernel_prefix}"
            vals[counter] = sum(xs) / len(xs)
        except Exception as ex:   # noqa: BLE001
            return None, f"{counter} pass failed: {ex!r}"[:200]
        finally:
            shutil.rmtree(out, ignore_errors=True)
    total = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return total, {"fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
                   "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two child runs of 3 forward passes; reads = 2 x FETCH_SIZE (gfx950 half-count of 16-byte accesses); mean over the kernel's launches"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N rank processes here,
    one per GPU (the reference starts one self-play process per GPU: python/rl_loop/sp_loop.py:160-192,
    selfplay.py:51-64).  Nothing in this parent has touched torch or the GPU.  Rank 0's child prints the
    JSON line on the inherited stdout; the first failing rank takes the others down with it."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in live:      # a dead rank leaves the others in a barrier: end exactly the ones started here
                    other.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512,
                    help="timed ROUNDS per GPU: one engine batch (p3hip_run) of every game group each")
    ap.add_argument("--warmup", type=int, default=16, help="untimed warm-up rounds (one batch per game group each)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--groups", type=int, default=DEFAULT_GROUPS, help="game groups per GPU (each `--batch` games)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="engine instances (HIP streams) per game group, filled in turn by the group's games; 1 (default) = "
                         "host and GPU alternate within a group and the other groups keep the GPU busy; `--groups 1 --lanes 2` "
                         "is BASELINE configs[2] as stated, the `as_stated_c3` leg of the default run")
    ap.add_argument("--inflight", type=int, default=4,
                    help="playouts of one game's search that may wait for results at once when --lanes > 1")
    ap.add_argument("--model", default=MODEL)
    ap.add_argument("--advance-limit", type=int, default=64,
                    help="untimed batches per group, at most, that play every game past its raw-policy opening "
                         "before the warm-up (0 = start timing inside the openings, as rounds 1-2 did)")
    ap.add_argument("--engine-steps", type=int, default=200, help="timed steps of the engine-only leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live HBM-traffic measurement (two rocprofv3 --pmc child runs)")
    ap.add_argument("--ladder-budget", type=int, default=LADDER_BUDGET,
                    help="ladder read-out work bound of the self-play host (0 = reference-exact, the default; "
                         "> 0 = the opt-in throughput mode, hits are reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the engine_benchmark and as_stated_c3 legs")
    ap.add_argument("--null-engine", action="store_true",
                    help="plumbing rehearsal without a GPU: the host over its NullEvaluator (uniform policy); "
                         "the line it prints is marked as not a measurement")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch one rank per GPU "
                 f"(torch.distributed.run --nproc-per-node {args.gpus}) or drop WORLD_SIZE and let bench.py start them")

    from p3achygo_amd import sharding
    shard = sharding.shard_from_env()
    rank, local_rank, world = shard.rank, shard.local_rank, shard.world
    if os.environ.get("P3_BENCH_SINGLE_DEVICE"):   # rehearsal of the N > 1 path on a one-GPU box
        local_rank = 0
    import torch
    sharding.init(shard)
    n_gpus = max(world, 1)
    use_gpu = not args.null_engine
    if use_gpu and local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank} but the node shows {torch.cuda.device_count()}")
    # host cores of this rank: its share of the allowed cores, on its GPU's NUMA node when known
    bus_ids = None
    try:
        bus_ids = []
        for d in range(torch.cuda.device_count() if use_gpu else 0):
            p = torch.cuda.get_device_properties(d)
            bus_ids.append(f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0")
    except Exception:   # noqa: BLE001
        bus_ids = None
    cpus = sharding.bind_rank_to_local_cpus(shard, bus_ids)

    from p3achygo_amd import host_api, netspec
    cfg = netspec.CONFIGS[args.model]
    tmp = tempfile.mkdtemp(prefix="p3bench")
    path = os.path.join(tmp, f"{args.model}.p3w")
    if use_gpu:
        from p3achygo_amd import engine
        netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))  # fresh-Keras random init
        pos = make_positions(args.batch, seed=1000 + rank)
        torch.cuda.set_device(local_rank)

    def sync():
        if use_gpu:
            torch.cuda.synchronize()

    # ---- headline: self-play through the C ABI ----------------------------------------------
    threads = max(2, min(16, len(cpus)))
    GROUPS = args.groups   # local from here on
    LANES = max(1, min(4, args.lanes))
    steps = args.steps
    host_api.set_groups(GROUPS)
    host_api.set_lanes(LANES, args.inflight if LANES > 1 else 1)
    host_api.set_ladder_budget(args.ladder_budget)
    host_api.set_step_rounds(steps)
    host_api.set_advance_limit(args.advance_limit)
    ladder0 = host_api.ladder_stats()
    sharding.barrier(shard)
    sync()
    wall0 = time.perf_counter()
    st = host_api.selfplay_run(path if use_gpu else None, GROUPS * args.batch, threads, 0.0, default_n=32, default_k=5,
                               selected_n=32, selected_k=5, warmup_batches=args.warmup * LANES,
                               seed=sharding.seed_for_rank(77, shard), device=local_rank)
    sync()
    wall = time.perf_counter() - wall0
    host_api.set_step_rounds(0)
    host_api.set_advance_limit(0)
    host_api.set_lanes(1, 1)
    ladder1 = host_api.ladder_stats()
    dt = sharding.max_over_ranks(shard, st.seconds)
    dt_fit = sharding.max_over_ranks(shard, st.seconds_fit)
    sharding.barrier(shard)
    totals = sharding.sum_over_ranks(shard, [st.positions, st.moves, st.games, st.batches, st.cache_hits,
                                             ladder1[0] - ladder0[0], ladder1[3] - ladder0[3],
                                             st.games_past_opening, st.advance_batches])

    def headline():
        pps = totals[0] / dt
        return {
            "metric": "self-play positions/sec at 1/8 MI355X, b12c256btl3 19x19 n=32",
            "value": pps, "unit": "positions/s", "n_gpus": n_gpus, "steps": steps,
            "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.model} random-init, self-play: C++ host (Gumbel MCTS n=32, default k<=5, "
                                   f"selected k=5) -> p3hip_run over batches of {args.batch} leaf positions -> "
                                   f"results back to the search; one step = one round = one engine batch of each of the "
                                   f"{GROUPS} game groups" + (f" x {LANES} lanes" if LANES > 1 else "") +
                                   f" = {GROUPS * LANES * args.batch} positions",
                       "batch_per_gpu": args.batch, "concurrent_games_per_gpu": GROUPS * args.batch,
                       "game_groups_per_gpu": GROUPS, "lanes_per_group": LANES, "host_threads_per_gpu": threads,
                       "host_cpus_of_rank0": [cpus[0], cpus[-1]] if cpus else None,
                       "ladder_node_budget": args.ladder_budget,
                       "parallelism": f"games sharded x{n_gpus}, no collective"},
            "includes": "host MCTS + PCIe (1,860 B up / 7,556 B down per position) + engine",
            "positions": totals[0], "moves_per_s": totals[1] / dt,
            "evals_per_move": totals[0] / max(totals[1], 1.0),
            "games_past_opening": totals[7], "games_total": n_gpus * GROUPS * args.batch,
            "advance_batches_untimed": totals[8],
            "finished_games": totals[2],
            "engine_batches_completed": totals[3], "mean_batch_fill": totals[0] / max(totals[3], 1) / args.batch,
            "ms_per_batch": dt / max(totals[3] / n_gpus, 1) * 1e3,
            "eval_cache_hits": totals[4], "ladder_readouts": totals[5], "ladder_budget_hits": totals[6],
            "seconds_timed": dt, "wall_s_incl_setup_advance_and_warmup": wall,
            # the same K rounds, per-round time from a least-squares line through the anchor group's K + 1 completion
            # instants instead of the first and the last
            "value_fit": totals[0] / dt_fit, "seconds_timed_fit": dt_fit,
        }

    if not use_gpu:
        if rank == 0:
            out = headline()
            out["engine"] = "NullEvaluator (uniform policy): a plumbing rehearsal of the launch path, NOT a measurement"
            out["roofline"] = out["cpu_baseline"] = None
            print(json.dumps(out), flush=True)
        sharding.finish(shard)
        return

    # ---- secondary: the engine alone over a resident batch ----------------------------------
    eng = engine.create_engine(engine.kind_from_engine_path(path), path, args.batch, 1, device=local_rank)
    eng.load_all(pos)
    eng.upload()
    for _ in range(5):
        eng.forward_resident(args.batch)
    eng.sync()
    sharding.barrier(shard)
    sampler = None
    if rank == 0:   # shader clock and socket power beside the loop (a side thread reading amdsmi every 20 ms)
        try:
            from p3achygo_amd.power_sampler import PowerSampler
            sampler = PowerSampler(local_rank)
            sampler.start()
        except Exception:   # noqa: BLE001
            sampler = None
    t0 = time.perf_counter()
    for _ in range(args.engine_steps):
        eng.forward_resident(args.batch)
    eng.sync()
    torch.cuda.synchronize()
    dt_e = sharding.max_over_ranks(shard, time.perf_counter() - t0)
    power = None
    if sampler is not None:
        try:
            power = sampler.stop()
        except Exception:   # noqa: BLE001
            power = None
    sharding.barrier(shard)
    total_flops, conv3_flops = eng.flops_per_position()

    roof = None
    cpu = None
    extras = {}
    if rank == 0:
        ms, flops_launch, kname = eng.time_trunk_kernel(args.batch, 10)
        achieved = flops_launch / (ms * 1e-3) / 1e12
        prof = hbm_traffic(kname, args.batch)
        roof = {"bound": "mfma", "kernel": kname, "achieved": achieved,
                "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP16_MFMA_TFLOPS,
                "traffic": prof["traffic_bytes_per_launch"] if prof else None,   # HBM bytes per launch (recorded)
                "launch_ms": ms, "algorithmic_flops_per_launch": flops_launch,
                # information beside the contract's `peak`: what a stream of nothing but v_mfma_f32_16x16x32_f16 on register
                # operands with this net's value statistics holds on this pool under the socket's power limit
                # (tools/gpu_mfma_energy_probe.py, profiles/r04_mfma_energy_probe.txt: 0.75 of the peak; 0.98 on zeros)
                "matrix_pipe_ceiling_measured": {"tflops": MEASURED_PIPE_CEILING_TFLOPS,
                                                 "frac_of_it": achieved / MEASURED_PIPE_CEILING_TFLOPS}}
        if prof:
            roof["algorithmic_bytes_per_launch"] = prof["algorithmic_bytes_per_launch"]
            roof["traffic_source"] = "recorded: " + prof["source"]
        if not args.no_pmc and n_gpus == 1:   # counter passes only on a single-GPU run: at N > 1 the recorded figure stands
            prefix = "void p3::k_block<" if kname.startswith("k_block") else "void p3::k_lconv<3,"
            live, detail = measure_hbm_traffic(args.model, args.batch, prefix)
            if live is not None:
                roof["traffic"] = live
                roof["traffic_source"] = "measured in this run: " + detail["how"]
                roof["traffic_counters"] = {k: detail[k] for k in ("fetch_size_kib_raw", "write_size_kib")}
            else:
                roof["traffic_live_unavailable"] = detail
        if kname.startswith("k_block<256,128,btl"):
            roof["launch_contents"] = ("ONE launch per forward pass: every residual block + the broadcast blocks' 1x1 convs "
                                       "(HBM-bound) and dense (LDS-bound) between them, 8.1 % of the FLOPs together; "
                                       "blocks_only = one launch per run of residual blocks with all of those as their own "
                                       "launches (P3HIP_NO_BFUSE=1: slower forward pass, comparable with earlier rounds)")
        elif kname.startswith("k_block"):
            roof["launch_contents"] = ("residual blocks + the 1x1 convs of the neighbouring broadcast blocks (HBM-bound); "
                                       "blocks_only = the same launches with those convs as their own launches "
                                       "(P3HIP_NO_BFUSE=1: slower forward pass, comparable with earlier rounds)")
        else:
            roof["launch_contents"] = ("one 3x3 layer conv (C_b -> C_b) of a layer-wise trunk: 4-wave workgroups of one position, "
                                       "two per CU; activations round-trip HBM between layers")
        if kname.startswith("k_block"):
            os.environ["P3HIP_NO_BFUSE"] = "1"       # read by p3hip_create: this instance only
            try:
                eng2 = engine.create_engine(engine.kind_from_engine_path(path), path, args.batch, 1, device=local_rank)
                eng2.load_all(pos)
                eng2.upload()
                for _ in range(5):
                    eng2.forward_resident(args.batch)
                eng2.sync()
                ms2, fl2, _ = eng2.time_trunk_kernel(args.batch, 10)
                roof["blocks_only"] = {"achieved": fl2 / (ms2 * 1e-3) / 1e12, "frac": fl2 / (ms2 * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                                       "launch_ms": ms2, "algorithmic_flops_per_launch": fl2}
                eng2.close()
            finally:
                del os.environ["P3HIP_NO_BFUSE"]
    eng.close()
    if rank == 0 and n_gpus == 1 and not args.no_extras:
        # the reference's engine micro-benchmark, from C++
        try:
            eb = host_api.engine_benchmark(path, pos, args.batch, warmup_runs=100, max_rounds=1001, device=local_rank)
            extras["engine_benchmark"] = {
                "value": args.batch / (eb.avg_run_us * 1e-6), "unit": "positions/s", "avg_run_us": eb.avg_run_us,
                "rounds": eb.rounds, "loop_positions_per_s": eb.positions / eb.loop_seconds,
                "what": "nn::Benchmark (benchmark_engine.cc:77-108) in C++ over the C ABI on one thread: 100 warm-up runs, "
                        f"{eb.rounds} rounds of p3hip_load_slot x {args.batch} -> p3hip_run -> p3hip_get_slot x {args.batch}; "
                        "value = batch / mean p3hip_run time (H2D + forward pass + D2H, the reference's clock placement); "
                        "loop_positions_per_s also counts the loads and gets"}
        except Exception as ex:   # noqa: BLE001
            extras["engine_benchmark"] = {"error": repr(ex)}
        # BASELINE configs[2] exactly as written: 1024 concurrent games, batch 1024 = ONE game group.  With one lane the
        # GPU idles while the host advances the games and the host idles during the forward pass (`one_lane`); with two
        # lanes (two engine instances filled in turn by the same 1024 games, up to 4 playouts of a search waiting for
        # results at once — the games and every result are the same, tests/test_selfplay_cpu.py) the group overlaps
        # its host work with its own forward passes (`value`)
        try:
            host_api.set_groups(1)
            host_api.set_step_limit(512)
            host_api.set_advance_limit(args.advance_limit)
            host_api.set_lanes(2, 4)
            s2 = host_api.selfplay_run(path, args.batch, threads, 0.0, default_n=32, default_k=5, selected_n=32,
                                       selected_k=5, warmup_batches=8, seed=177, device=local_rank)
            host_api.set_lanes(1, 1)
            s1 = host_api.selfplay_run(path, args.batch, threads, 0.0, default_n=32, default_k=5, selected_n=32,
                                       selected_k=5, warmup_batches=8, seed=177, device=local_rank)
            extras["as_stated_c3"] = {
                "value": s2.positions / s2.seconds, "unit": "positions/s", "steps": s2.batches,
                "concurrent_games": args.batch, "game_groups": 1, "lanes": 2, "playouts_in_flight_per_game": 4,
                "mean_batch_fill": s2.positions / max(s2.batches, 1) / args.batch,
                "evals_per_move": s2.positions / max(s2.moves, 1),
                "host_share_of_time": s2.host_seconds / s2.seconds,
                "one_lane": {"value": s1.positions / s1.seconds, "steps": s1.batches,
                             "gpu_share_of_time": s1.gpu_seconds / s1.seconds,
                             "host_share_of_time": s1.host_seconds / s1.seconds},
                "what": "BASELINE configs[2] as stated (1024 concurrent games, inference batch 1024): ONE game group whose "
                        "games fill two engine batches in turn, so its host work overlaps its own forward passes; "
                        "`one_lane` = the same group with host and GPU alternating (rounds 1-3); the headline keeps "
                        "`game_groups_per_gpu` one-lane groups in flight instead"}
        except Exception as ex:   # noqa: BLE001
            extras["as_stated_c3"] = {"error": repr(ex)}
            host_api.set_lanes(1, 1)
        finally:
            host_api.set_groups(GROUPS)
            host_api.set_step_limit(0)
            host_api.set_advance_limit(0)
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(path, pos, args.model)
        except Exception as ex:   # noqa: BLE001
            cpu = {"error": repr(ex)}
    sharding.barrier(shard)

    if rank == 0:
        out = headline()
        pps = out["value"]
        eng_pps = n_gpus * args.batch * args.engine_steps / dt_e
        out["conv3x3_mfma_frac_end_to_end"] = pps * conv3_flops / 1e12 / (PEAK_FP16_MFMA_TFLOPS * n_gpus)
        out["engine_only"] = {"value": eng_pps, "unit": "positions/s", "steps": args.engine_steps,
                              "ms_per_step": dt_e / args.engine_steps * 1e3,
                              "what": "forward pass over a resident batch: no host, no PCIe",
                              "full_net_tflops": eng_pps * total_flops / 1e12,
                              "conv3x3_mfma_frac": eng_pps * conv3_flops / 1e12 / (PEAK_FP16_MFMA_TFLOPS * n_gpus),
                              # the chip's state while this loop ran (rank 0's GPU): the forward pass sits on the socket power
                              # cap, and the clock the chip grants under it (2.0-2.2 GHz of 2.4) is in every number of this line
                              "chip_during_loop": power}
        out.update(extras)
        # self-check: a step cannot be shorter than the ONE trunk launch every batch must execute
        if roof and roof.get("launch_ms") and roof.get("kernel", "").startswith("k_block"):
            out["window_artifact"] = bool(out["ms_per_batch"] < 0.98 * roof["launch_ms"])
        out["roofline"] = roof
        out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    sharding.finish(shard)


if __name__ == "__main__":
    main()
