"""Engine-only micro-benchmark in the shape of the reference's Benchmark()
(cc/nn/engine/benchmark_engine.cc:77-108): LoadBatch x B -> RunInference -> GetBatch x B, 100
warm-up rounds, 1000 timed; single engine instance, single stream, PCIe included."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
name = sys.argv[1] if len(sys.argv) > 1 else "b12c256btl3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
timed = int(sys.argv[3]) if len(sys.argv) > 3 else 300
cfg = netspec.CONFIGS[name]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), (batch + 63) // 64)[:batch].copy()
eng = engine.create_engine(engine.kind_from_engine_path(path), path, batch, 1)
res = engine.Result()
def round_():
    eng.load_all(pos)            # LoadBatch for every slot (one ctypes call per slot)
    eng.RunInference()
    eng.GetBatch(0, res); eng.GetBatch(batch - 1, res)
for _ in range(30): round_()
t0 = time.perf_counter()
for _ in range(timed): round_()
dt = time.perf_counter() - t0
print(f"{name} batch {batch}: {batch*timed/dt:,.0f} positions/s  ({dt/timed*1e3:.3f} ms per round, PCIe + host LoadBatch loop included)")
