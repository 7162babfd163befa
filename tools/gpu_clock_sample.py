"""Shader clock and socket power while the headline net's forward pass loops (amdsmi samples every
20 ms from a side thread): random weights, then the same binary on all-zero weights.
Usage: gpu_clock_sample.py [seconds per leg] [net]"""
import os, sys, tempfile, threading, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


from p3achygo_amd.power_sampler import PowerSampler as Sampler   # noqa: E402


smp = Sampler()
print("sampler:", smp.kind, "idle:", smp.read(), flush=True)
batch = 1024
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
cfg = netspec.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "b12c256btl3"]
for label in ("random", "zeros", "random"):
    W = netspec.generate_weights(cfg)
    if label == "zeros":
        W = {k: (np.zeros_like(v) if not k.endswith(".var") else v) for k, v in W.items()}
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, W)
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.upload()
    for _ in range(50):
        eng.forward_resident(batch)
    eng.sync()
    samples, stop = [], False

    def loop():
        while not stop:
            try:
                samples.append(smp.read())
            except Exception as e:  # noqa: BLE001
                samples.append((0.0, 0.0))
            time.sleep(0.02)

    th = threading.Thread(target=loop); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            eng.forward_resident(batch)
        eng.sync(); n += 20
    dt = time.perf_counter() - t0
    stop = True; th.join()
    s = np.array(samples[len(samples) // 4:])   # steady state
    try:
        kms, fl, kname = eng.time_trunk_kernel(batch, 20)
    except Exception:   # layer-wise trunks have no fused block kernel
        kms, fl, kname = float("nan"), 0.0, "-"
    print(f"{label:7s} forward {dt / n * 1e3:.3f} ms; {kname} {kms:.4f} ms/launch frac {fl / kms / 1e9 / 2500:.3f}; "
          f"gfx clock mean {s[:, 0].mean():.0f} MHz (min {s[:, 0].min():.0f}, max {s[:, 0].max():.0f}); "
          f"socket power mean {s[:, 1].mean():.0f} W (max {s[:, 1].max():.0f}); {len(s)} samples", flush=True)
    eng.close()
