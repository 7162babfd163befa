"""Network specification, deterministic random-init weights and the `.p3w` weight file.

The arithmetic spec is the reference's Keras graph (python/model.py); this module only
holds the *shape* information (python/model_config.py:62-172) and the initialiser
distributions (model.py:101-126 conv/dense factories, :218-232 ConvBlock, :681-683
gpool-bias dense, :872 gamma output zeros, :1155-1164 init conv / game layer), so that
weights of the named architecture can be produced on a box with no reference checkout.

Tensor layouts in the file are the Keras ones (conv kernels HWIO, Dense (in, out), BN
gamma/beta/moving_mean/moving_variance) so that a future `.keras` importer is a plain
copy; engines repack to their own device layouts at load time.

File format (`.p3w`, little endian):
    char  magic[4] = "P3W1"
    i32   version, nblocks, C, Cb, H, V, bcast_interval, inner_layers, block_type, ntensors
    ntensors x { char name[48]; i32 ndim; i32 dims[4]; i64 offset (in floats) }
    pad to 64 B, then float32 data
"""
from __future__ import annotations

import dataclasses
import struct
from typing import Dict, List, Tuple

import numpy as np

NUM_PLANES = 15
NUM_SCALARS = 8
NUM_LOCS = 361
NUM_MOVES = 362
SCORE_RANGE = 800
NUM_V_BUCKETS = 51
BN_EPS = 1e-3  # model.py:231 BatchNormalization(momentum=0.99, epsilon=1e-3)

BLOCK_TYPES = {"btl": 0, "nbt": 1, "classic": 2}
WEIGHT_SEED = 0x70336163  # "p3ac" 


@dataclasses.dataclass(frozen=True)
class NetConfig:
    name: str
    blocks: int
    channels: int
    bottleneck_channels: int
    head_channels: int
    c_val: int
    broadcast_interval: int
    inner_layers: int
    block_type: str

    def block_kind(self, i: int) -> str:
        """model.py:1002-1011: block i is a broadcast block iff i % interval == interval-1."""
        if i % self.broadcast_interval == self.broadcast_interval - 1:
            return "broadcast"
        return self.block_type


# Table captured from python/model_config.py:62-172 (see tests/golden/model_configs.json);
# b12c128btl3 is named by config/v2-b12c128btl3.json but absent from ModelConfig.from_str,
# so it is defined here by analogy with b12c256btl3.
CONFIGS: Dict[str, NetConfig] = {
    c.name: c
    for c in [
        NetConfig("tiny", 6, 16, 8, 8, 16, 4, 1, "btl"),
        NetConfig("small", 16, 128, 64, 32, 64, 8, 2, "btl"),
        NetConfig("b10c128btl3", 10, 128, 64, 32, 64, 4, 3, "btl"),
        NetConfig("b12c128btl3", 12, 128, 64, 32, 64, 5, 3, "btl"),
        NetConfig("b12c256btl3", 12, 256, 128, 32, 64, 5, 3, "btl"),
        NetConfig("b14c384btl3", 14, 384, 192, 32, 80, 6, 3, "btl"),
        NetConfig("b15c192_classic", 15, 192, 64, 32, 80, 6, 2, "classic"),
        NetConfig("b8c128nbt", 8, 128, 64, 32, 64, 3, 2, "nbt"),
        NetConfig("b12c256nbt", 12, 256, 128, 32, 80, 3, 2, "nbt"),
        NetConfig("b10c384nbt", 10, 384, 192, 32, 80, 4, 2, "nbt"),
        # not in the reference: shallow nets of the supported widths for fast parity tests
        NetConfig("test_b3c128btl2", 3, 128, 64, 32, 32, 3, 2, "btl"),
        NetConfig("test_b3c128nbt", 3, 128, 64, 32, 48, 3, 2, "nbt"),
        NetConfig("test_b3c256btl1", 3, 256, 128, 32, 64, 3, 1, "btl"),
        NetConfig("test_b3c256nbt", 3, 256, 128, 32, 80, 3, 2, "nbt"),
        NetConfig("test_b3c384btl3", 3, 384, 192, 32, 80, 3, 3, "btl"),
        NetConfig("test_b3c384nbt", 3, 384, 192, 32, 64, 3, 2, "nbt"),
        NetConfig("test_b3c192classic", 3, 192, 64, 32, 80, 3, 2, "classic"),
        # broadcast blocks in the MIDDLE of the trunk (interval 2: blocks 1 and 3), so that block
        # launches with a broadcast block before them, after them and on both sides all occur
        NetConfig("test_b5c256nbt_i2", 5, 256, 128, 32, 48, 2, 2, "nbt"),
        NetConfig("test_b5c128btl1_i2", 5, 128, 64, 32, 32, 2, 1, "btl"),
        # C = 256 btl with two inner layers and broadcast blocks 1 and 3: three one-block runs joined into ONE
        # k_block launch (conv_last head, fused dense tail on both sides of a run), V = 48 heads
        NetConfig("test_b5c256btl2_i2", 5, 256, 128, 32, 48, 2, 2, "btl"),
        # ten blocks, every second one a broadcast block, the LAST block among them: five one-block runs — more than a
        # joined launch takes (kMaxRuns = 4), so the first launch ends with a tail read from outside and the second
        # begins with a head fed from outside (every run of such a launch stores to / loads from the SAME u buffer),
        # and the trunk ends with a stand-alone conv_last
        NetConfig("test_b10c256btl1_i2", 10, 256, 128, 32, 32, 2, 1, "btl"),
    ]
}


def tensor_specs(cfg: NetConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Canonical (name, shape, init) list.  init in {glorot, vs_tn, vs1, zeros, bn}."""
    C, Cb, H, V = cfg.channels, cfg.bottleneck_channels, cfg.head_channels, cfg.c_val
    t: List[Tuple[str, Tuple[int, ...], str]] = []

    def conv(name, k, cin, cout, init="glorot"):
        t.append((name + ".w", (k, k, cin, cout), init))

    def dense(name, cin, cout, init="glorot"):
        t.append((name + ".w", (cin, cout), init))
        t.append((name + ".b", (cout,), "zeros"))

    def bn(name, c):
        for f in ("gamma", "beta", "mean", "var"):
            t.append((f"{name}.{f}", (c,), "bn_" + f))

    conv("init_conv", 5, NUM_PLANES, C, "vs_tn")  # model.py:1155-1163
    dense("init_game", NUM_SCALARS, C)  # model.py:1164
    for i in range(cfg.blocks):
        p = f"blocks.{i}"
        kind = cfg.block_kind(i)
        if kind == "broadcast":  # model.py:570-606
            bn(p + ".bn0", C)
            conv(p + ".conv0", 1, C, C)
            dense(p + ".dense", NUM_LOCS, NUM_LOCS)
            bn(p + ".bn1", C)
            conv(p + ".conv1", 1, C, C)
        elif kind == "btl":  # model.py:372-425
            L = cfg.inner_layers
            bn(p + ".bn0", C)
            conv(p + ".conv0", 1, C, Cb)
            for j in range(1, L + 1):
                bn(f"{p}.bn{j}", Cb)
                conv(f"{p}.conv{j}", 3, Cb, Cb)
            bn(f"{p}.bn{L + 1}", Cb)
            conv(f"{p}.conv{L + 1}", 1, Cb, C)
        elif kind == "nbt":  # model.py:430-486
            bn(p + ".bn0", C)
            conv(p + ".conv0", 1, C, Cb)
            for j in range(1, 5):
                bn(f"{p}.bn{j}", Cb)
                conv(f"{p}.conv{j}", 3, Cb, Cb)
            bn(p + ".bn5", Cb)
            conv(p + ".conv5", 1, Cb, C)
        elif kind == "classic":  # model.py:329-368
            for j in range(2):
                bn(f"{p}.bn{j}", C)
                conv(f"{p}.conv{j}", 3, C, C)
        else:
            raise ValueError(kind)
    # policy head, model.py:725-812
    conv("policy.conv_p", 1, C, H)
    conv("policy.conv_g", 1, C, H)
    bn("policy.gpool_bn", H)
    dense("policy.gpool_dense", 2 * H, H, "vs1")  # model.py:681-683
    conv("policy.out_moves", 1, H, 2)
    dense("policy.out_pass", 2 * H, 2)
    conv("policy.soft_moves", 1, H, 1)
    dense("policy.soft_pass", 2 * H, 1)
    conv("policy.opt_moves", 1, H, 1)
    dense("policy.opt_pass", 2 * H, 1)
    # value head, model.py:824-979
    conv("value.conv", 1, C, H)
    dense("value.oq_embed", 2 * H, V)
    dense("value.oq_out", V, 14)
    dense("value.mcts_dist", V, NUM_V_BUCKETS)
    conv("value.own", 1, H, 1)
    dense("value.gamma_pre", 2 * H, V)
    dense("value.gamma_out", V, 1, "zeros")  # model.py:872
    dense("value.score_pre", 2 * H + 1, V)
    dense("value.score_out", V, 1)
    return t


def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = shape[0] * shape[1]
    return rf * shape[2], rf * shape[3]


def generate_weights(cfg: NetConfig, seed: int = WEIGHT_SEED, randomize: bool = False
                     ) -> Dict[str, np.ndarray]:
    """Random-init weights of architecture `cfg` following the Keras initialisers.

    randomize=False: a fresh Keras model (BN gamma=1 beta=0 mean=0 var=1, zero biases).
    randomize=True : BN statistics, biases and the zero-initialised gamma output are drawn
    at random as well, so that every term of the forward pass is exercised.
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape, init in tensor_specs(cfg):
        if init == "glorot":
            fi, fo = _fans(shape)
            lim = np.sqrt(6.0 / (fi + fo))
            w = rng.uniform(-lim, lim, size=shape)
        elif init in ("vs_tn", "vs1"):
            # VarianceScaling(scale=1, fan_in, truncated_normal): stddev/.87962566, cut at 2σ
            fi, _ = _fans(shape)
            std = np.sqrt(1.0 / fi) / 0.87962566103423978
            w = rng.normal(0.0, 1.0, size=shape)
            bad = np.abs(w) > 2.0
            while bad.any():
                w[bad] = rng.normal(0.0, 1.0, size=int(bad.sum()))
                bad = np.abs(w) > 2.0
            w = w * std
        elif init == "zeros":
            w = np.zeros(shape)
            if randomize:
                w = rng.normal(0.0, 0.1, size=shape)
        elif init == "bn_gamma":
            w = rng.uniform(0.5, 1.5, size=shape) if randomize else np.ones(shape)
        elif init == "bn_beta":
            w = rng.normal(0.0, 0.1, size=shape) if randomize else np.zeros(shape)
        elif init == "bn_mean":
            w = rng.normal(0.0, 0.1, size=shape) if randomize else np.zeros(shape)
        elif init == "bn_var":
            w = rng.uniform(0.5, 2.0, size=shape) if randomize else np.ones(shape)
        else:
            raise ValueError(init)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


POLICY_OUT_TENSORS = ("policy.out_moves.w", "policy.out_pass.w", "policy.out_pass.b",
                      "policy.opt_moves.w", "policy.opt_pass.w", "policy.opt_pass.b")


def peak_policy(weights: Dict[str, np.ndarray], scale: float) -> Dict[str, np.ndarray]:
    """Scales the last linear layer of the policy / optimistic-policy outputs by `scale`.

    Random-init nets have near-uniform policies (logit sigma ~0.35, max prob ~0.01), on which
    probability tolerances and argmax checks say little.  Scaling the final 1x1 conv / dense
    multiplies the logits (pass offset excluded) by `scale`, i.e. gives the peaked policies of
    a trained net while every other tensor keeps its initialiser distribution.
    """
    out = dict(weights)
    for k in POLICY_OUT_TENSORS:
        out[k] = (weights[k] * np.float32(scale)).astype(np.float32)
    return out


_HDR = struct.Struct("<4s10i")
_ENT = struct.Struct("<48si4iq")


def save_p3w(path: str, cfg: NetConfig, weights: Dict[str, np.ndarray], version: int = 1) -> None:
    specs = tensor_specs(cfg)
    ents = []
    off = 0
    for name, shape, _ in specs:
        w = weights[name]
        assert tuple(w.shape) == tuple(shape), (name, w.shape, shape)
        dims = list(shape) + [1] * (4 - len(shape))
        ents.append(_ENT.pack(name.encode(), len(shape), *dims, off))
        off += int(np.prod(shape))
    hdr = _HDR.pack(b"P3W1", version, cfg.blocks, cfg.channels, cfg.bottleneck_channels,
                    cfg.head_channels, cfg.c_val, cfg.broadcast_interval, cfg.inner_layers,
                    BLOCK_TYPES[cfg.block_type], len(specs))
    blob = hdr + b"".join(ents)
    blob += b"\0" * ((-len(blob)) % 64)
    with open(path, "wb") as f:
        f.write(blob)
        for name, _, _ in specs:
            f.write(np.ascontiguousarray(weights[name], dtype="<f4").tobytes())


def load_p3w(path: str) -> Tuple[NetConfig, Dict[str, np.ndarray], int]:
    with open(path, "rb") as f:
        data = f.read()
    magic, version, nb, C, Cb, H, V, bi, il, bt, nt = _HDR.unpack_from(data, 0)
    if magic != b"P3W1":
        raise ValueError("not a .p3w file: " + path)
    btype = {v: k for k, v in BLOCK_TYPES.items()}[bt]
    cfg = NetConfig("file", nb, C, Cb, H, V, bi, il, btype)
    pos = _HDR.size
    ents = []
    for _ in range(nt):
        name, nd, d0, d1, d2, d3, off = _ENT.unpack_from(data, pos)
        pos += _ENT.size
        ents.append((name.rstrip(b"\0").decode(), (d0, d1, d2, d3)[:nd], off))
    base = pos + ((-pos) % 64)
    arr = np.frombuffer(data, dtype="<f4", offset=base)
    w = {n: arr[o:o + int(np.prod(s))].reshape(s).copy() for n, s, o in ents}
    return cfg, w, version


def flops_per_position(cfg: NetConfig) -> Tuple[float, float]:
    """(total, 3x3-trunk-conv-only) algorithmic FLOPs (2*MAC) per position, SURVEY.md §8d."""
    C, Cb, H, V = cfg.channels, cfg.bottleneck_channels, cfg.head_channels, cfg.c_val
    mac = NUM_LOCS * 25 * NUM_PLANES * C + NUM_SCALARS * C
    mac3 = 0
    for i in range(cfg.blocks):
        kind = cfg.block_kind(i)
        if kind == "broadcast":
            mac += NUM_LOCS * 2 * C * C + C * NUM_LOCS * NUM_LOCS
        elif kind == "btl":
            mac += NUM_LOCS * 2 * C * Cb
            mac3 += NUM_LOCS * cfg.inner_layers * 9 * Cb * Cb
        elif kind == "nbt":
            mac += NUM_LOCS * 2 * C * Cb
            mac3 += NUM_LOCS * 4 * 9 * Cb * Cb
        elif kind == "classic":
            mac3 += NUM_LOCS * 2 * 9 * C * C
    mac += NUM_LOCS * 3 * C * H + NUM_LOCS * H * 5  # head 1x1 convs
    mac += 2 * H * H + 2 * H * 4 + 2 * H * V * 2 + V * (14 + NUM_V_BUCKETS + 1)
    mac += (2 * H + 1) * V + SCORE_RANGE * V
    mac += mac3
    return 2.0 * mac, 2.0 * mac3
