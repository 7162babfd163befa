"""numpy restatement of what k_blockw computes, and of engine.cpp's packing of its weight stream / parameter table
(test infrastructure: tests/test_blockw_asm_cpu.py runs the generated assembly in sim.py against this)."""
import numpy as np

LOG2E = np.float32(1.4426950408889634)
LN2 = np.float32(0.6931471805599453)


def mish_l2(t):
    """t = log2(e) * y (float32) -> mish(y) as float32, the kernel's arithmetic (conv_core.h mish_t2)"""
    t = t.astype(np.float32)
    with np.errstate(over="ignore", divide="ignore"):
        e = np.exp2(t.astype(np.float64)).astype(np.float32)
        d = (e.astype(np.float64) * (e + np.float32(2.0)).astype(np.float64) + 2.0).astype(np.float32)
        r = (1.0 / d.astype(np.float64)).astype(np.float32)
        u = (r.astype(np.float64) * np.float64(np.float32(-2.0) * LN2) + np.float64(LN2)).astype(np.float32)
    return (t.astype(np.float64) * u.astype(np.float64)).astype(np.float32)


def bn_mish16(v, sc, sh):
    """v [C][361] float32, sc/sh [C] already times log2(e) -> fp16"""
    t = (v.astype(np.float64) * sc[:, None].astype(np.float64) + sh[:, None].astype(np.float64)).astype(np.float32)
    return mish_l2(t).astype(np.float16)


def conv(a16, w, k):
    """a16 [Cin][361] fp16, w [k][k][Cin][Cout] (values already fp16-representable) -> [Cout][361] float32, SAME padding"""
    cin = a16.shape[0]
    g = np.zeros((cin, 19 + 2, 19 + 2), np.float64)
    g[:, 1:20, 1:20] = a16.astype(np.float64).reshape(cin, 19, 19)
    out = np.zeros((w.shape[3], 19, 19), np.float64)
    p = k // 2
    for ky in range(k):
        for kx in range(k):
            patch = g[:, 1 + ky - p:20 + ky - p, 1 + kx - p:20 + kx - p].reshape(cin, 361)
            out += (w[ky, kx].astype(np.float64).T @ patch).reshape(-1, 19, 19)
    return out.reshape(-1, 361).astype(np.float32)


def block_ref(x16, W, bn, L):
    """one btl block: x16 [256][361] fp16; W[j] j = 0..L+1 HWIO float32 (fp16-representable); bn[j] = (scale, shift) float32
    (NOT yet times log2(e)) -> x' fp16"""
    a = bn_mish16(x16.astype(np.float32), bn[0][0] * LOG2E, bn[0][1] * LOG2E)
    r = conv(a, W[0], 1)
    a = bn_mish16(r, bn[1][0] * LOG2E, bn[1][1] * LOG2E)
    for j in range(1, L + 1):
        r = conv(a, W[j], 3)
        a = bn_mish16(r, bn[j + 1][0] * LOG2E, bn[j + 1][1] * LOG2E)
    y = conv(a, W[L + 1], 1)
    return (y.astype(np.float64) + x16.astype(np.float64)).astype(np.float32).astype(np.float16)


def granule(Wf, tap, k0, cout0, scale=None):
    """Wf [taps][cin][cout] -> 2048 fp16: [k16 half j][cout tile c][h][n][8] = W[tap][k0 + 16 j + 8 h + e][cout0 + 32 c + n]
    (times scale[cout]: the folded BN scale of the layer that follows, times log2 e), rounded to fp16 once"""
    out = np.zeros((2, 2, 2, 32, 8), np.float16)
    for j in range(2):
        for c in range(2):
            for h in range(2):
                k = k0 + 16 * j + 8 * h
                blk = Wf[tap, k:k + 8, cout0 + 32 * c:cout0 + 32 * c + 32].astype(np.float32)      # [8][32]
                if scale is not None:
                    blk = blk * scale[cout0 + 32 * c:cout0 + 32 * c + 32][None, :]
                out[j, c, h] = blk.T.astype(np.float16)
    return out.reshape(-1)


def pack_block(W, bn, L):
    """-> (weight stream fp16, parameter floats) of one block, engine.cpp build_plan's order"""
    ws = []
    sc = lambda j: (bn[j][0] * LOG2E).astype(np.float32)      # noqa: E731
    w0 = W[0].reshape(1, 256, 128)
    for half in range(2):                     # x quarters (0, 1) then (2, 3); inside: set A's four granules, then set B's
        for s0 in (0, 64):
            for st in range(4):
                ws.append(granule(w0, 0, 128 * half + 32 * st, s0, sc(1)))
    for j in range(1, L + 1):
        wj = W[j].reshape(9, 128, 128)
        for ph in range(4):
            s0, half = (ph & 1) * 64, ph >> 1
            for ky in range(3):
                for q in range(2):
                    for kx in range(3):
                        ws.append(granule(wj, ky * 3 + kx, 64 * half + 32 * q, s0, sc(j + 1)))
    we = W[L + 1].reshape(1, 128, 256)
    for qo in range(4):
        for c in range(4):
            ws.append(granule(we, 0, 32 * c, 64 * qo))
    prm = [(bn[0][0] * LOG2E).astype(np.float32), (bn[0][1] * LOG2E).astype(np.float32)]
    for j in range(1, L + 2):
        prm.append((bn[j][1] * LOG2E).astype(np.float32))
    return np.concatenate(ws), np.concatenate(prm)


def x_to_device(x16):
    """[256][361] -> device layout [32][361][8]"""
    return x16.reshape(32, 8, 361).transpose(0, 2, 1).copy()


def x_from_device(d):
    return d.reshape(32, 361, 8).transpose(0, 2, 1).reshape(256, 361)


def random_block(rng, L):
    W = [None] * (L + 2)
    W[0] = (rng.standard_normal((1, 1, 256, 128)) * (1.0 / 16)).astype(np.float16).astype(np.float32)
    for j in range(1, L + 1):
        W[j] = (rng.standard_normal((3, 3, 128, 128)) * (1.0 / 34)).astype(np.float16).astype(np.float32)
    W[L + 1] = (rng.standard_normal((1, 1, 128, 256)) * (1.0 / 11)).astype(np.float16).astype(np.float32)
    bn = []
    for j in range(L + 2):
        c = 256 if j == 0 else 128
        bn.append((rng.uniform(0.5, 1.5, c).astype(np.float32), (rng.standard_normal(c) * 0.1).astype(np.float32)))
    return W, bn
