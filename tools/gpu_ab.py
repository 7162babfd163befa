"""A/B timing of kernel builds on one box: tools/gpu_ab.py libA.so libB.so [...]
Each build is timed in its own subprocess, alternating, three rounds; prints the trunk
kernel's us/launch and the resident forward's ms for b12c256btl3 at batch 1024."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time, tempfile
sys.path.insert(0, %r)
import numpy as np
from p3achygo_amd import engine, features, netspec
batch = 1024
cfg = netspec.CONFIGS[os.environ.get("AB_NET", "b12c256btl3")]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
eng = engine.HipEngine(path, batch)
eng.load_all(pos); eng.upload()
for _ in range(5): eng.forward_resident(batch)
eng.sync()
try:
    ms, fl, kn = eng.time_trunk_kernel(batch, 20)
except Exception:      # layer-wise trunks have no fused block kernel to time
    ms, fl = float("nan"), 0.0
t0 = time.perf_counter()
for _ in range(20): eng.forward_resident(batch)
eng.sync()
fw = (time.perf_counter() - t0) / 20 * 1e3
print("%%-40s k_block %%.1f us  (%%.0f TFLOP/s)  forward %%.3f ms" %% (os.path.basename(os.environ["P3HIP_LIB"]), ms * 1e3, fl / ms / 1e9, fw))
""" % ROOT
for rnd in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, P3HIP_LIB=os.path.abspath(lib))
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
