"""Cost of the on-device NN cache on p3hip_run at 1024 positions (b12c256btl3): plain run, keyed run with every key
new (probe + forward + store), keyed run with every key known (probe + fill only), and a half/half mix."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
net = sys.argv[1] if len(sys.argv) > 1 else "b12c256btl3"
B = 1024
cfg = netspec.CONFIGS[net]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:B].copy()
sz = pos.dtype.itemsize


def timed(eng, load, reps=20):
    for _ in range(3):
        load(-1); eng.RunInference()
    t = 0.0
    for r in range(reps):
        load(r)
        t0 = time.perf_counter(); eng.RunInference(); t += time.perf_counter() - t0
    return t / reps * 1e3


plain = engine.HipEngine(path, B)
ms_plain = timed(plain, lambda r: plain.load_all(pos))
plain.close()
eng = engine.HipEngine(path, B)
eng.EnableCache(18)
L, h = eng._L, eng._h
ctr = [1]


def load_new(r):
    for i in range(B):
        ctr[0] += 1
        L.p3hip_load_slot_keyed(h, i, pos.ctypes.data + i * sz, ctr[0] * 0x9E3779B97F4A7C15 & (2**64 - 1), ctr[0], 0)


def load_known(r):
    for i in range(B):
        L.p3hip_load_slot_keyed(h, i, pos.ctypes.data + i * sz, (i + 1) * 0xD6E8FEB86659FD93 & (2**64 - 1), 7, 0)


def load_mix(r):
    for i in range(B):
        if i % 2:
            ctr[0] += 1
            L.p3hip_load_slot_keyed(h, i, pos.ctypes.data + i * sz, ctr[0] * 0x9E3779B97F4A7C15 & (2**64 - 1), ctr[0], 0)
        else:
            L.p3hip_load_slot_keyed(h, i, pos.ctypes.data + i * sz, (i + 1) * 0xD6E8FEB86659FD93 & (2**64 - 1), 7, 0)


ms_new = timed(eng, load_new)
load_known(0); eng.RunInference()
ms_known = timed(eng, load_known)
ms_mix = timed(eng, load_mix)
print(f"{net} p3hip_run of {B} positions: plain {ms_plain:.3f} ms; keyed, all new {ms_new:.3f} ms; all known {ms_known:.3f} ms; half known {ms_mix:.3f} ms; {eng.cache_stats()}")
eng.close()
