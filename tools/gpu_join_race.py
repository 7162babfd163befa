"""Diagnostic: tests/test_engine_gpu.py::test_repeated_and_concurrent_runs_are_bit_identical with a report of what differs."""
import os, sys, threading, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
name = "b12c256btl3"
batch = 512
cfg = netspec.CONFIGS[name]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg, randomize=True))
pos = np.tile(features.random_positions(64, seed=2, n_games=16), batch // 64).copy()
SLOTS = (0, 1, batch // 2, batch - 1, 77)
ref = {}
report = []
def worker(tag, iters):
    eng = engine.HipEngine(path, batch)
    for it in range(iters):
        eng.load_all(pos)
        eng.RunInference()
        rows = {s: eng.get_raw(s).copy() for s in SLOTS}
        if not ref:
            ref.update(rows)
        for s in SLOTS:
            d = np.abs(rows[s] - ref[s])
            if (d > 0).any() or np.isnan(rows[s]).any():
                cols = np.nonzero(~(d == 0))[0]
                report.append((tag, it, s, len(cols), cols[:6].tolist(), float(np.nanmax(d)), bool(np.isnan(rows[s]).any())))
    eng.close()
worker("solo", 20)
print("after solo:", len(report))
ths = [threading.Thread(target=worker, args=("t%d" % t, 20)) for t in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
print("mismatches:", len(report))
for r in report[:30]:
    print(r)
