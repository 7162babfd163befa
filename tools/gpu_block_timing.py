"""Times the fused block kernel for L = 1, 2, 3 inner 3x3 convs (and nbt) to split its cost
into per-3x3-layer time and fixed (stage-in + 1x1 reduce/expand + epilogues) time."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pos = np.tile(features.random_positions(64, seed=1, n_games=16), (batch + 63) // 64)[:batch].copy()
res = {}
for name, C, Cb, L, bt in [("L1", 256, 128, 1, "btl"), ("L2", 256, 128, 2, "btl"), ("L3", 256, 128, 3, "btl"),
                           ("nbt", 256, 128, 2, "nbt"), ("c128L3", 128, 64, 3, "btl")]:
    cfg = netspec.NetConfig(name, 3, C, Cb, 32, 64, 3, L, bt)
    d = tempfile.mkdtemp()
    path = os.path.join(d, name + ".p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.upload(); eng.forward_resident(batch); eng.sync()
    ms, fl, kn = eng.time_trunk_kernel(batch, 10)
    res[name] = ms
    print(f"{name}: {ms*1e3:.1f} us/launch  ({fl/ms/1e9:.0f} TFLOP/s on 3x3 flops)  {kn}")
    eng.close()
per3 = (res["L3"] - res["L1"]) / 2
print(f"per 3x3 layer: {per3*1e3:.1f} us  ({2*batch*361*9*128*128/per3/1e9:.0f} TFLOP/s = {2*batch*361*9*128*128/per3/1e9/2500*100:.1f}% of peak); fixed part: {(res['L1']-per3)*1e3:.1f} us")
