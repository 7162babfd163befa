"""Phase breakdown of the fused btl block kernel from in-kernel s_memtime stamps."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from p3achygo_amd import engine, features, netspec
batch = 1024
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
cfg = netspec.NetConfig("L3", 3, 256, 128, 32, 64, 3, 3, "btl")
path = os.path.join(tempfile.mkdtemp(), "L3.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
eng = engine.HipEngine(path, batch)
eng.load_all(pos); eng.upload(); eng.forward_resident(batch); eng.sync()
st = eng.debug_block_stamps(batch).astype(np.int64)   # [wg][wave][posidx][stamp]
names = ["stage_store0+load1", "seg slice0", "barrier+stage_store1", "seg slice1",
         "epi->act1", "conv3x3 #1", "epi->act2", "conv3x3 #2", "epi->act3", "conv3x3 #3", "epi->act4",
         "stage_load next", "expand seg0", "expand epi0", "expand seg1", "expand epi1", "final barrier"]
for w in (0, 7):
    d = np.diff(st[:, w, 1:3, :18], axis=-1)   # positions 1..2 (steady state)
    med = np.median(d.reshape(-1, 17), axis=0)
    tot = np.median((st[:, w, 1:3, 17] - st[:, w, 1:3, 0]).reshape(-1))
    print(f"wave {w}: total per position {tot:.0f} cycles")
    for n, m in zip(names, med):
        print(f"   {n:24s} {m:8.0f}  {100*m/tot:5.1f}%")
    wc = st[:, w, 1:3, 20:24]
    print("   acquire-wait cycles inside conv3x3 #1..#3:", np.median(np.diff(wc, axis=-1).reshape(-1, 3), axis=0))
conv = st[:, :, 1:3, 6] - st[:, :, 1:3, 5]
print("conv3x3 #1 cycles per wave 0..7:", np.median(conv.transpose(1, 0, 2).reshape(8, -1), axis=1))
t0 = st[:, :, 1:3, 5] - st[:, 0:1, 1:3, 5]
print("conv3x3 #1 start skew vs wave 0:", np.median(t0.transpose(1, 0, 2).reshape(8, -1), axis=1))
span = np.median(st[:, 0, 3, 17] - st[:, 0, 0, 0])
print("4 positions span (cycles):", span)
