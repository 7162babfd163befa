"""Where a k_block launch's wall time goes beyond its steady-state phases (DIAGNOSTIC build, make -C
p3achygo_amd/csrc diag): every workgroup's entry / ring-ready / end-of-position / exit instants of launch
P3DIAG_LAUNCH (default 1) on the shader clock (durations inside a workgroup) and on the 100 MHz device-wide
counter (start skew and finish skew across workgroups).
Usage: P3HIP_LIB=build/libp3hip_diag.so python tools/gpu_spans.py [net] [batch]"""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("P3HIP_LIB", os.path.join(ROOT, "build", "libp3hip_diag.so"))
import numpy as np
from p3achygo_amd import engine, features, netspec
net = sys.argv[1] if len(sys.argv) > 1 else "b12c256btl3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = netspec.CONFIGS[net]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
pos = np.tile(features.random_positions(64, seed=1, n_games=16), (batch + 63) // 64)[:batch].copy()
eng = engine.HipEngine(path, batch)
eng.load_all(pos); eng.upload()
for _ in range(30):
    eng.forward_resident(batch)
eng.sync()
ms, fl, kn = eng.time_trunk_kernel(batch, 20)
for _ in range(3):
    eng.forward_resident(batch)
eng.sync()
L = eng._L
L.p3hip_debug_block_spans.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
WG, SL = 512, 16
buf = np.zeros(WG * SL, np.uint64)
assert L.p3hip_debug_block_spans(eng._h, buf.ctypes.data, buf.size) == 0
sp = buf.reshape(WG, SL).astype(np.int64)
live = sp[:, 0] > 0
sp = sp[live]
n = sp.shape[0]
clk, rt = sp[:, :8], sp[:, 8:]
print(f"{net} batch {batch}: launch {os.environ.get('P3DIAG_LAUNCH', '1')}, {n} workgroups; average k_block launch (HIP events, all launches) {ms * 1e3:.1f} us")
tot_c, tot_r = clk[:, 7] - clk[:, 0], (rt[:, 7] - rt[:, 0]) / 100.0
print(f"  workgroup entry -> exit: {tot_c.mean():.0f} cycles (min {tot_c.min()}, max {tot_c.max()}) = {tot_r.mean():.1f} us (min {tot_r.min():.1f}, max {tot_r.max():.1f}); shader clock {tot_c.mean() / tot_r.mean():.0f} MHz")
print(f"  entry -> ring ready: {(clk[:, 1] - clk[:, 0]).mean():.0f} cycles")
prev = clk[:, 1]
for p in range(5):
    cur = clk[:, 2 + p]
    ok = cur > 0
    if not ok.any():
        break
    d = (cur - prev)[ok]
    print(f"  position {p}: {d.mean():.0f} cycles (min {d.min()}, max {d.max()}), {ok.sum()} workgroups")
    prev = np.where(ok, cur, prev)
t0 = rt[:, 0].min()
start, end = (rt[:, 0] - t0) / 100.0, (rt[:, 7] - t0) / 100.0
print(f"  start skew across workgroups: mean {start.mean():.1f} us, max {start.max():.1f} us; first exit {end.min():.1f} us, last exit {end.max():.1f} us (launch span on the device {end.max():.1f} us)")
if n == 512:
    for p in range(2):
        a0 = clk[:256, 2 + p] - clk[:256, 1 + p]
        a1 = clk[256:, 2 + p] - clk[256:, 1 + p]
        print(f"  position {p}: first workgroup of a CU {a0.mean():.0f} cycles, second {a1.mean():.0f}")
    print(f"  first 256 workgroups start at {start[:256].mean():.1f} us, the second of each CU at {start[256:].mean():.1f} us; exits {end[:256].mean():.1f} / {end[256:].mean():.1f} us")
eng.close()
