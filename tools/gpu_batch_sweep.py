"""k_block launch time and forward time vs batch size: separates per-launch fixed cost from per-position cost."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
cfg = netspec.CONFIGS["b12c256btl3"]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
base = features.random_positions(64, seed=1, n_games=16)
for batch in [256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096]:
    pos = np.tile(base, (batch + 63) // 64)[:batch].copy()
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.upload()
    for _ in range(5): eng.forward_resident(batch)
    eng.sync()
    ms, fl, kn = eng.time_trunk_kernel(batch, 20)
    t0 = time.perf_counter()
    for _ in range(20): eng.forward_resident(batch)
    eng.sync()
    fw = (time.perf_counter() - t0) / 20 * 1e3
    print(f"batch {batch:5d}: k_block {ms*1e3:7.1f} us ({ms*1e3/batch*256:6.1f} us per position-round)  forward {fw:7.3f} ms  {batch/fw:7.1f} k positions/s", flush=True)
    eng.close()
