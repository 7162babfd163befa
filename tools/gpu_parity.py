"""Diagnostic: HIP engine vs CPU oracle / golden fixtures on the GPU box (prints stats)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle
from p3achygo_amd import engine, features, netspec

names = sys.argv[1:] or ["test_b3c256btl1", "test_b3c128btl2", "test_b3c256nbt", "test_b3c128nbt"]
for name in names:
    cfg = netspec.CONFIGS[name]
    W = netspec.generate_weights(cfg, randomize=True)
    d = tempfile.mkdtemp()
    path = os.path.join(d, name + ".p3w")
    netspec.save_p3w(path, cfg, W)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"nn_{name}.npz"))
    n = int(g["n_pos"])
    pos = np.frombuffer(g["features"].tobytes(), dtype=features.features_dtype())
    t0 = time.time()
    eng = engine.HipEngine(path, 8)
    eng.load_all(pos)
    eng.RunInference()
    print(name, "create+run s", round(time.time() - t0, 2))
    for i in range(n):
        raw = eng.get_raw(i)
        ref = g["raw"][i]
        segs = {"pi": (0, 362), "opt": (362, 724), "outcome": (724, 726), "score": (726, 1526),
                "own": (1526, 1887), "q6err": (1887, 1888), "gamma": (1888, 1889)}
        msg = []
        for k, (a, b) in segs.items():
            dlt = np.abs(raw[a:b] - ref[a:b]).max()
            msg.append(f"{k}:{dlt:.4f}/{np.abs(ref[a:b]).max():.2f}")
        r = eng.GetBatch(i)
        mp = np.ctypeslib.as_array(r.move_probs)
        msg.append(f"probs:{np.abs(mp - g['move_probs'][i]).max():.5f} argmax:{mp.argmax()==g['move_probs'][i].argmax()}")
        print("  pos", i, " ".join(msg), "nan" if np.isnan(raw).any() else "")
    eng.close()
