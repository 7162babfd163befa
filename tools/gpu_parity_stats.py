"""Diagnostic: worst-case errors of the HIP engine against every committed golden fixture
(per net: max |d logit| per head segment, max |d prob|, max KL) — the numbers the tolerances
in tests/test_engine_gpu.py are set from."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec

names = sys.argv[1:] or ["test_b3c128btl2", "test_b3c128nbt", "test_b3c256btl1", "test_b3c256nbt", "test_b3c384btl3",
                         "test_b3c384nbt", "test_b3c192classic", "b8c128nbt", "b12c128btl3", "b12c256btl3",
                         "b12c256btl3_peaked", "b10c384nbt", "b14c384btl3"]
SEGS = {"pi": (0, 362), "opt": (362, 724), "outcome": (724, 726), "score": (726, 1526), "own": (1526, 1887),
        "q6err": (1887, 1888), "gamma": (1888, 1889)}


def kl(p, q):
    p = np.asarray(p, np.float64); q = np.maximum(np.asarray(q, np.float64), 1e-300)
    m = p > 0
    return float((p[m] * np.log(p[m] / q[m])).sum())


for name in names:
    g = np.load(os.path.join(ROOT, "tests", "golden", f"nn_{name}.npz"))
    peak = float(g["peak"])
    base = name[:-len("_peaked")] if peak else name
    cfg = netspec.CONFIGS[base]
    W = netspec.generate_weights(cfg, randomize=True)
    if peak:
        W = netspec.peak_policy(W, peak)
    path = os.path.join(tempfile.mkdtemp(), name + ".p3w")
    netspec.save_p3w(path, cfg, W)
    pos = np.frombuffer(g["features"].tobytes(), dtype=features.features_dtype())
    eng = engine.HipEngine(path, max(8, len(pos)))
    eng.load_all(pos)
    eng.RunInference()
    worst = {k: 0.0 for k in SEGS}
    wp = {k: 0.0 for k in ("move_probs", "value_probs", "score_probs", "opt_move_probs")}
    wk = dict(wp)
    agree = 0
    for i in range(len(pos)):
        raw = eng.get_raw(i)
        for k, (a, b) in SEGS.items():
            worst[k] = max(worst[k], float(np.abs(raw[a:b] - g["raw"][i][a:b]).max()))
        r = eng.GetBatch(i)
        for k in wp:
            got = np.ctypeslib.as_array(getattr(r, k))
            wp[k] = max(wp[k], float(np.abs(got - g[k][i]).max()))
            wk[k] = max(wk[k], kl(g[k][i], got))
        agree += int(np.ctypeslib.as_array(r.move_probs).argmax() == g["move_probs"][i].argmax())
    print(f"{name:20s} n={len(pos):2d} logits " + " ".join(f"{k}:{v:.2e}" for k, v in worst.items()))
    print(f"{'':20s} probs  " + " ".join(f"{k}:{v:.2e}" for k, v in wp.items()))
    print(f"{'':20s} KL     " + " ".join(f"{k}:{v:.2e}" for k, v in wk.items()) + f"  argmax {agree}/{len(pos)}", flush=True)
    eng.close()
