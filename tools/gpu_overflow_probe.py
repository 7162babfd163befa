"""Robustness probe: a net whose activations overflow fp16 (inf / NaN everywhere) must run to
completion and hand back NaNs, not fault.  Run under `timeout`."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
name = sys.argv[1] if len(sys.argv) > 1 else "test_b3c256btl1"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1e3
cfg = netspec.CONFIGS[name]
W = netspec.generate_weights(cfg)
for k in W:
    if k.endswith(".w") and "conv" in k:
        W[k] = (W[k] * scale).astype(np.float32)
path = os.path.join(tempfile.mkdtemp(), "ovf.p3w")
netspec.save_p3w(path, cfg, W)
pos = features.random_positions(8, seed=3, n_games=8)
eng = engine.HipEngine(path, 8)
for i in range(8):
    eng.LoadBatch(i, pos[i:i + 1])
print("running", name, "scale", scale, flush=True)
eng.RunInference()
r = eng.GetBatch(0)
ml = np.ctypeslib.as_array(r.move_logits)
print("done: finite logits", int(np.isfinite(ml).sum()), "of", ml.size, flush=True)
eng.close()
