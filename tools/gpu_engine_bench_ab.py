#!/usr/bin/env python3
"""The reference's Benchmark() loop (benchmark_engine.cc:77-108) from C++ over the C ABI (p3host_engine_benchmark), with and
without the result records written by the heads kernel (P3HIP_NO_DIRECT_RESULTS=1 = the strided D2H copy of rounds 1-3)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os, tempfile
sys.path.insert(0, %r)
import numpy as np
from p3achygo_amd import host_api, netspec, features
cfg = netspec.CONFIGS["b12c256btl3"]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:1024].copy()
eb = host_api.engine_benchmark(path, pos, 1024, warmup_runs=100, max_rounds=601, device=0)
print("RESULT avg_run_us %%.1f  positions/s %%.0f  loop positions/s %%.0f" %% (eb.avg_run_us, 1024 / (eb.avg_run_us * 1e-6), eb.positions / eb.loop_seconds), flush=True)
"""
VARIANTS = (("direct", {}), ("copied", {"P3HIP_NO_DIRECT_RESULTS": "1"}))
for rnd in range(2):
    for label, extra in VARIANTS:
        env = dict(os.environ); env.pop("P3HIP_NO_DIRECT_RESULTS", None); env.update(extra)
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=300)
        print(label, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
