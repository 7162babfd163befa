"""DVFS check: the same binary on the headline net with all-zero weights (every MFMA operand zero)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
batch = 1024
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
cfg = netspec.CONFIGS["b12c256btl3"]
for label in ("random", "zeros", "random"):
    W = netspec.generate_weights(cfg)
    if label == "zeros":
        W = {k: (np.zeros_like(v) if not k.endswith(".var") else v) for k, v in W.items()}
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, W)
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.upload()
    for _ in range(200): eng.forward_resident(batch)
    eng.sync()
    kms, fl, kname = eng.time_trunk_kernel(batch, 20)
    print(f"{label:7s} {kname} {kms:.4f} ms/launch frac {fl / kms / 1e9 / 2500:.3f}", flush=True)
    eng.close()
