"""Runs the first btl block of a 3-block C=256 net repeatedly (for rocprofv3 --pmc runs)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
batch = 1024
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
cfg = netspec.NetConfig("L3", 3, 256, 128, 32, 64, 3, 3, "btl")
path = os.path.join(tempfile.mkdtemp(), "L3.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
eng = engine.HipEngine(path, batch)
eng.load_all(pos); eng.upload(); eng.forward_resident(batch); eng.sync()
ms, fl, kn = eng.time_trunk_kernel(batch, 5)
print(kn, ms)
eng.close()
