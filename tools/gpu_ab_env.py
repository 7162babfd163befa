"""A/B timing of one build under different environment switches on one box:
  tools/gpu_ab_env.py [net] VAR1=val,VAR2=val ... (an empty spec "-" = no switch)
Each variant is timed in its own subprocess, alternating, three rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time, tempfile
sys.path.insert(0, %r)
import numpy as np
from p3achygo_amd import engine, features, netspec
batch = int(os.environ.get("AB_BATCH", "1024"))
cfg = netspec.CONFIGS[os.environ.get("AB_NET", "b12c256btl3")]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), (batch + 63) // 64)[:batch].copy()
eng = engine.HipEngine(path, batch)
eng.load_all(pos); eng.upload()
for _ in range(30): eng.forward_resident(batch)
eng.sync()
t0 = time.perf_counter()
for _ in range(100): eng.forward_resident(batch)
eng.sync()
fw = (time.perf_counter() - t0) / 100 * 1e3
try:
    ms, fl, kn = eng.time_trunk_kernel(batch, 20)
except Exception:
    ms, fl = float("nan"), 0.0
print("%%-28s %%s forward %%.3f ms (%%.0f pos/s); block launch %%.1f us (%%.0f TFLOP/s)" %% (os.environ["AB_LABEL"], cfg.name if hasattr(cfg, "name") else "", fw, batch / fw * 1e3, ms * 1e3, fl / ms / 1e9), flush=True)
""" % ROOT
args = sys.argv[1:]
net = "b12c256btl3"
if args and "=" not in args[0] and args[0] != "-":
    net, args = args[0], args[1:]
for rnd in range(3):
    for spec in args:
        env = dict(os.environ, AB_NET=net, AB_LABEL=spec)
        if spec != "-":
            for kv in spec.split(","):
                k, v = kv.split("=")
                env[k] = v
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
