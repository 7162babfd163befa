"""Aggregate forward rate of N engine instances (own HIP stream each) fed round-robin from one host
thread, resident batches: does kernel-level overlap between streams (one stream's HBM-bound phases under
another's MFMA-bound ones) raise the GPU's throughput?  Usage: gpu_multi_stream.py [net]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
net = sys.argv[1] if len(sys.argv) > 1 else "b12c256btl3"
cfg = netspec.CONFIGS[net]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
for n_eng, batch in ((1, 1024), (2, 1024), (4, 1024), (2, 512), (4, 512), (4, 256), (8, 256), (1, 2048), (1, 4096)):
    pos = np.tile(features.random_positions(64, seed=1, n_games=16), (batch + 63) // 64)[:batch].copy()
    engs = [engine.HipEngine(path, batch) for _ in range(n_eng)]
    for e in engs:
        e.load_all(pos); e.upload()
    for _ in range(10):
        for e in engs:
            e.forward_resident(batch)
    for e in engs:
        e.sync()
    reps = max(20, 200 * 1024 // (batch * n_eng))
    t0 = time.perf_counter()
    for _ in range(reps):
        for e in engs:
            e.forward_resident(batch)
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    print(f"{net}: {n_eng} engines x batch {batch}: {reps * n_eng * batch / dt:9.0f} positions/s", flush=True)
    for e in engs:
        e.close()
