"""Race screen for the hand-synchronised kernels: the same batch evaluated many times must give
bit-identical results every time (a premature ring read or an unordered store would show up as
a run that differs).  Usage: gpu_determinism.py [net] [iterations] [batch]"""
import os, sys, tempfile, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
name = sys.argv[1] if len(sys.argv) > 1 else "b12c256btl3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
cfg = netspec.CONFIGS[name]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg, randomize=True))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), (batch + 63) // 64)[:batch].copy()
eng = engine.HipEngine(path, batch)
ref = None
bad = 0
for it in range(iters):
    eng.load_all(pos)
    eng.RunInference()
    h = hashlib.sha1()
    for s in (0, 1, batch // 2, batch - 1, 17 % batch, 333 % batch):
        h.update(eng.get_raw(s).tobytes())
    d = h.hexdigest()
    if ref is None:
        ref = d
    elif d != ref:
        bad += 1
        print("iteration", it, "differs", flush=True)
print(f"{name}: {iters} runs of {batch} positions, {bad} differing", flush=True)
eng.close()
sys.exit(1 if bad else 0)
