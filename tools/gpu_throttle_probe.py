"""Which limiter holds the clock while the forward pass loops: amdsmi's violation status (power / current / thermal) before and
after 3 s of resident forward passes, plus clock and power samples.  python tools/gpu_throttle_probe.py"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import amdsmi
from p3achygo_amd import engine, features, netspec
from p3achygo_amd.power_sampler import PowerSampler

amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]


def show(tag):
    for fn in ("amdsmi_get_violation_status", "amdsmi_get_power_cap_info", "amdsmi_get_gpu_metrics_info"):
        try:
            r = getattr(amdsmi, fn)(h)
            if fn == "amdsmi_get_gpu_metrics_info":
                r = {k: v for k, v in r.items() if any(t in k for t in ("throttle", "power", "gfxclk", "temperature", "activity", "residency", "acc"))
                     and not isinstance(v, (list, tuple))}
            print(tag, fn, r, flush=True)
        except Exception as e:   # noqa: BLE001
            print(tag, fn, "unavailable:", repr(e)[:160], flush=True)


cfg = netspec.CONFIGS["b12c256btl3"]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:1024].copy()
eng = engine.HipEngine(path, 1024)
eng.load_all(pos); eng.upload()
show("idle  ")
smp = PowerSampler(0)
smp.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 3.0:
    for _ in range(20):
        eng.forward_resident(1024)
    eng.sync(); n += 20
    if n % 200 == 0:
        show(f"t={time.perf_counter() - t0:4.1f}")
print("forward ms", (time.perf_counter() - t0) / n * 1e3, smp.stop(), flush=True)
show("after ")
eng.close()
