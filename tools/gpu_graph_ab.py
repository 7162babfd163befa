"""p3hip_run over a full batch with and without P3HIP_FLAG_LAUNCH_GRAPH (ms per run, H2D + forward + D2H)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
for name, batch in (("b12c256btl3", 1024), ("b12c256btl3", 64), ("b12c128btl3", 256)):
    cfg = netspec.CONFIGS[name]
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
    pos = np.tile(features.random_positions(64, seed=1, n_games=16), (batch + 63) // 64)[:batch].copy()
    engs = {"launches": engine.HipEngine(path, batch), "graph": engine.HipEngine(path, batch, flags=engine.FLAG_LAUNCH_GRAPH)}
    for rnd in range(3):
        for label, eng in engs.items():
            for _ in range(5):
                eng.load_all(pos); eng.RunInference()
            t0 = time.perf_counter()
            for _ in range(50):
                eng.load_all(pos); eng.RunInference()
            ms = (time.perf_counter() - t0) / 50 * 1e3
            print(f"{name} batch {batch:5d} {label:9s} {ms:7.3f} ms per load_all + p3hip_run (graph state {eng.graph_state()})", flush=True)
    for e in engs.values():
        e.close()
