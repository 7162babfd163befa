"""Runs a few resident forwards of a net at batch 1024 (target for rocprofv3 runs).
Usage: gpu_run_forward.py [iterations] [net]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
batch = 1024
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
cfg = netspec.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "b12c256btl3"]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
eng = engine.HipEngine(path, batch)
eng.load_all(pos); eng.upload()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng.forward_resident(batch)
eng.sync()
eng.close()
