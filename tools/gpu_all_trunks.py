"""Resident forward time of every supported trunk at batch 1024 (one engine each)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
batch = 1024
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
names = sys.argv[1:] or ["b12c256btl3", "b12c128btl3", "b8c128nbt", "b12c256nbt", "b15c192_classic", "b10c384nbt", "b14c384btl3"]
for name in names:
    cfg = netspec.CONFIGS[name]
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.upload()
    for _ in range(5): eng.forward_resident(batch)
    eng.sync()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): eng.forward_resident(batch)
    eng.sync()
    ms = (time.perf_counter() - t0) / n * 1e3
    total, conv3 = eng.flops_per_position()
    try:
        kms, fl, kname = eng.time_trunk_kernel(batch, 5)
        kern = f"{kname} {kms * 1e3:7.1f} us/launch = {fl / kms / 1e9 / 2500:.3f} of 2.5 PFLOP/s"
    except Exception as ex:  # noqa: BLE001
        kern = "-"
    print(f"{name:18s} forward {ms:7.3f} ms  {batch / ms:8.1f} k positions/s  whole net {total * batch / ms / 1e9:6.0f} TFLOP/s  {kern}", flush=True)
    eng.close()
