"""Shader clock, socket power and limiter residencies while each BASELINE trunk's forward pass loops (resident batch):
which trunks run into the package power limit and which do not.  python tools/gpu_chip_state_trunks.py [seconds per net]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
from p3achygo_amd.power_sampler import PowerSampler

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
smp = PowerSampler(0)
for name, batch in (("b12c256btl3", 1024), ("b12c128btl3", 1024), ("b12c128btl3", 256), ("b8c128nbt", 1024), ("b14c384btl3", 1024),
                    ("b10c384nbt", 1024)):
    cfg = netspec.CONFIGS[name]
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
    pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.upload()
    for _ in range(30):
        eng.forward_resident(batch)
    eng.sync()
    smp.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            eng.forward_resident(batch)
        eng.sync(); n += 50
    dt = time.perf_counter() - t0
    st = smp.stop() or {}
    total_flops, _ = eng.flops_per_position()
    r = st.get("limiter_residency") or {}
    print(f"{name:14s} batch {batch:5d}  forward {dt / n * 1e3:7.3f} ms  {batch * n / dt / 1e3:7.1f} k positions/s  "
          f"whole net {batch * n / dt * total_flops / 1e15:5.3f} PFLOP/s  clock {st.get('gfx_clock_mhz_mean', 0):5.0f} MHz  "
          f"power {st.get('socket_power_w_mean', 0):5.0f} W  ppt residency {r.get('ppt', float('nan')):.2f}  "
          f"thermal {max(r.get('socket_thm', 0), r.get('vr_thm', 0), r.get('hbm_thm', 0)):.2f}", flush=True)
    eng.close()
