"""Phase timing of k_block from the DIAGNOSTIC build (make -C p3achygo_amd/csrc diag; never the shipped
library): per-wave s_memtime stamps of one launch (P3DIAG_LAUNCH, default 1 = the launch with a fused
broadcast conv on both sides), second position of workgroups 0..7.  Prints shader-clock cycles per phase.
Usage: P3HIP_LIB=build/libp3hip_diag.so python tools/gpu_stamps.py [net] [zeros]"""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("P3HIP_LIB", os.path.join(ROOT, "build", "libp3hip_diag.so"))
import numpy as np
from p3achygo_amd import engine, features, netspec
net = sys.argv[1] if len(sys.argv) > 1 else "b12c256btl3"
zeros = len(sys.argv) > 2 and sys.argv[2] == "zeros"
batch = 1024
cfg = netspec.CONFIGS[net]
W = netspec.generate_weights(cfg)
if zeros:
    W = {k: (np.zeros_like(v) if not k.endswith(".var") else v) for k, v in W.items()}
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, W)
pos = np.tile(features.random_positions(64, seed=1, n_games=16), 16)[:batch].copy()
eng = engine.HipEngine(path, batch)
eng.load_all(pos); eng.upload()
for _ in range(30):
    eng.forward_resident(batch)
eng.sync()
L = eng._L
L.p3hip_debug_block_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
WG, WV, SEC, SL = 8, 8, 8, 32
buf = np.zeros(WG * WV * SEC * SL, np.uint64)
assert L.p3hip_debug_block_stamps(eng._h, buf.ctypes.data, buf.size) == 0
st = buf.reshape(WG, WV, SEC, SL).astype(np.int64)
nw = 4 if cfg.channels == 128 and not os.environ.get("P3HIP_C128_WG8") else 8
st = st[:, :nw]
NAMES_BLK = {1: "x load/activate/write A0 (first block only)", 2: "reduce seg K0", 3: "barrier + write A1", 4: "reduce seg K1",
             5: "epilogue reduce", 6: "3x3 #1", 7: "epilogue 1", 8: "3x3 #2", 9: "epilogue 2", 10: "3x3 #3", 11: "epilogue 3",
             12: "residual load issue (pass 0)", 13: "expand seg pass 0", 14: "expand epilogue pass 0 (+stores)",
             15: "expand seg pass 1", 16: "residual load + barrier + write A0", 17: "expand epilogue pass 1 (+stores)"}
NAMES_HEAD = {1: "z loads + write slice 0", 2: "seg p0 K0", 4: "write slice 1 + x load issue", 5: "seg p0 K1", 6: "epilogue p0",
              7: "z slice 0 reload issue", 8: "seg p1 K1", 10: "write slice 0 + x load issue", 11: "seg p1 K0", 12: "epilogue p1",
              13: "x' half 0 reload + activate + write"}
NAMES_TAIL = {1: "(entry)", 2: "seg p0 K0", 4: "barrier + write A1", 5: "seg p0 K1", 6: "mish + store p0", 7: "x' half 0 reload issue",
              8: "seg p1 K1", 10: "activate + write half 0", 11: "seg p1 K0", 12: "barrier + mish + store p1"}

# the tail with the dense fused in (tail_dense): the same eight phases per channel half, slots 16 * half + k
NAMES_TAIL_DENSE = {}
for _h in range(2):
    for _k, _n in {1: "x' reload, activate, write half 0 (second half only)", 2: "conv_first seg K0", 3: "barrier + write A1",
                   4: "conv_first seg K1", 5: "barrier + mish -> Tt + dense parameters", 6: "dense: 3 column passes + u stores",
                   7: "barrier + zero halo"}.items():
        NAMES_TAIL_DENSE[16 * _h + _k] = f"half {_h}: {_n}"
NAMES_TAIL_DENSE[16] = "(between the halves)"


def report(title, sec, names):
    s = st[:, :, sec, :]
    ks = [k for k in range(SL) if (s[:, :, k] > 0).all()]
    if len(ks) < 2:
        return 0.0
    print(f"--- {title}: cycles (mean over {s.shape[0]} workgroups x {s.shape[1]} waves; min..max)")
    tot = 0.0
    for a, b in zip(ks[:-1], ks[1:]):
        d = s[:, :, b] - s[:, :, a]
        tot += d.mean()
        print(f"  {a:2d}->{b:2d} {names.get(b, ''):48s} {d.mean():9.0f}  ({d.min():6d}..{d.max():6d})")
    print(f"  total {tot:9.0f}")
    return tot


grand = report("head (conv_last of the broadcast block before the run)", 6, NAMES_HEAD)
for blk in range(6):
    grand += report(f"block {blk}", blk, NAMES_BLK)
dense_tail = (st[:, :, 7, 16] > 0).all()
grand += report("tail (conv_first%s of the broadcast block after the run)" % (" + dense" if dense_tail else ""), 7,
                NAMES_TAIL_DENSE if dense_tail else NAMES_TAIL)
# whole position: first stamp of the first section to last stamp of the last
print(f"sum of phases {grand:.0f} cycles per position ({'zero' if zeros else 'random'} weights)")
eng.close()
