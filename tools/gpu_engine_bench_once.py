"""One engine_benchmark run (the reference's Benchmark() loop over the C ABI) for profiling: python tools/gpu_engine_bench_once.py [rounds]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from p3achygo_amd import host_api, netspec, features
cfg = netspec.CONFIGS["b12c256btl3"]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:1024].copy()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eb = host_api.engine_benchmark(path, pos, 1024, warmup_runs=50, max_rounds=rounds, device=0)
print("RESULT avg_run_us %.1f  positions/s %.0f  loop positions/s %.0f" % (eb.avg_run_us, 1024 / (eb.avg_run_us * 1e-6), eb.positions / eb.loop_seconds), flush=True)
