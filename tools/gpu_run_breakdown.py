#!/usr/bin/env python3
"""Where p3hip_run's time goes for a single caller: host wall time of upload (gather + H2D + sync), one forward pass +
sync, 20 forward passes back to back, and the whole p3hip_run, 1024 positions of b12c256btl3."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
cfg = netspec.CONFIGS["b12c256btl3"]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
B = 1024
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:B].copy()
eng = engine.HipEngine(path, B)
L = eng._L
h = eng._h
def t(f, n=50):
    for _ in range(5): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
eng.load_all(pos)
def load_only(): eng.load_all(pos)
def upload(): eng.load_all(pos); L.p3hip_upload(h)
def fwd1(): L.p3hip_forward_resident(h, B); L.p3hip_sync(h)
def fwd20():
    for _ in range(20): L.p3hip_forward_resident(h, B)
    L.p3hip_sync(h)
def run(): eng.load_all(pos); L.p3hip_run(h)
tl = t(load_only)
print("load_all (1024 ctypes calls)      %.3f ms" % tl)
print("upload - load                     %.3f ms" % (t(upload) - tl))
print("one forward + sync                %.3f ms" % t(fwd1))
print("20 forwards + sync, per forward   %.3f ms" % (t(fwd20, 10) / 20))
print("p3hip_run - load                  %.3f ms" % (t(run) - tl))
eng.close()
# stage by stage inside p3hip_run (drained after each stage: P3HIP_TIME_RUN)
import subprocess
child = r"""
import sys, os, tempfile
sys.path.insert(0, %r)
import numpy as np
from p3achygo_amd import engine, features, netspec
cfg = netspec.CONFIGS["b12c256btl3"]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:1024].copy()
eng = engine.HipEngine(path, 1024)
for r in range(60):
    eng.load_all(pos); eng.RunInference()
print(os.environ.get("P3HIP_NO_DIRECT_RESULTS", "direct"), eng._L.p3hip_last_error(eng._h).decode())
""" % ROOT
for extra in ({}, {"P3HIP_NO_DIRECT_RESULTS": "copied"}):
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, P3HIP_TIME_RUN="1", **extra), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-300:])
