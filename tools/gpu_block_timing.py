"""Forward time and fused-block kernel time of the headline net (and others given on the command line)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
batch = int(os.environ.get("BATCH", "1024"))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), (batch + 63) // 64)[:batch].copy()
for name in sys.argv[1:] or ["b12c256btl3"]:
    cfg = netspec.CONFIGS[name]
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.upload()
    for _ in range(10): eng.forward_resident(batch)
    eng.sync()
    t0 = time.perf_counter(); n = 50
    for _ in range(n): eng.forward_resident(batch)
    eng.sync()
    ms = (time.perf_counter() - t0) / n * 1e3
    line = f"{name:18s} forward {ms:7.3f} ms  {batch / ms:8.1f} k positions/s"
    try:
        kms, fl, kname = eng.time_trunk_kernel(batch, 10)
        line += f"  {kname} {kms:.4f} ms/launch {fl / kms / 1e9:.1f} TFLOP/s frac {fl / kms / 1e9 / 2500:.3f}"
    except Exception as ex:
        line += f"  ({ex})"
    print(line, flush=True)
    eng.close()
