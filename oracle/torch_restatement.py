"""Independent PyTorch (CPU, float64 by default) restatement of the reference network.

TEST INFRASTRUCTURE ONLY — used by oracle/make_golden.py (to emit tests/golden/*.npz) and
by tests to cross-check oracle/nn_oracle.c.  Never imported by the product path.

It follows python/model.py of the reference (file:line cited per function) but shares no
code with oracle/nn_oracle.c: convolutions go through torch.nn.functional.conv2d in NCHW,
the broadcast dense through a batched matmul, so that a layout or indexing mistake in
either restatement shows up as a disagreement.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # model.py:231


def _t(w, dtype):
    return torch.from_numpy(np.asarray(w)).to(dtype)


def _conv(x, w_hwio):
    """make_conv (model.py:101-117): Conv2D, padding="same", no bias.  x is NCHW."""
    k = w_hwio.shape[0]
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1).contiguous(), padding=k // 2)


def _mish(x):
    return x * torch.tanh(F.softplus(x, threshold=30.0))


def _bn(x, W, prefix, dtype):
    g, b, m, v = (_t(W[f"{prefix}.{f}"], dtype) for f in ("gamma", "beta", "mean", "var"))
    s = g / torch.sqrt(v + BN_EPS)
    return (x - m[None, :, None, None]) * s[None, :, None, None] + b[None, :, None, None]


def _preact(x, W, blk, idx, dtype):
    """ConvPreActivation.call (model.py:276-282)."""
    return _conv(_mish(_bn(x, W, f"blocks.{blk}.bn{idx}", dtype)),
                 _t(W[f"blocks.{blk}.conv{idx}.w"], dtype))


def _gpool(x):
    """GlobalPool.call (model.py:641-645)."""
    return torch.cat([x.mean(dim=(2, 3)), x.amax(dim=(2, 3))], dim=1)


def _dense(x, W, name, dtype):
    return x @ _t(W[name + ".w"], dtype) + _t(W[name + ".b"], dtype)


def forward(cfg, W: Dict[str, np.ndarray], planes_nhwc: np.ndarray, feats: np.ndarray,
            dtype=torch.float64) -> Dict[str, np.ndarray]:
    """P3achyGoModel.call (model.py:1222-1295).  planes_nhwc: [N,19,19,15], feats: [N,8]."""
    x = _t(planes_nhwc, dtype).permute(0, 3, 1, 2)
    gs = _dense(_t(feats, dtype), W, "init_game", dtype)
    x = _conv(x, _t(W["init_conv.w"], dtype)) + gs[:, :, None, None]  # model.py:1230-1237
    N = x.shape[0]
    for i in range(cfg.blocks):
        kind = cfg.block_kind(i)
        if kind == "broadcast":  # model.py:556-606
            t = _preact(x, W, i, 0, dtype)
            t = _mish(t).reshape(N, cfg.channels, 361)
            t = t @ _t(W[f"blocks.{i}.dense.w"], dtype) + _t(W[f"blocks.{i}.dense.b"], dtype)
            t = _preact(t.reshape(N, cfg.channels, 19, 19), W, i, 1, dtype)
            x = x + t
        elif kind == "btl":  # model.py:372-425
            t = x
            for j in range(cfg.inner_layers + 2):
                t = _preact(t, W, i, j, dtype)
            x = x + t
        elif kind == "nbt":  # model.py:430-486
            t = _preact(x, W, i, 0, dtype)
            for r in range(2):
                u = _preact(_preact(t, W, i, 1 + 2 * r, dtype), W, i, 2 + 2 * r, dtype)
                t = t + u
            x = x + _preact(t, W, i, 5, dtype)
        else:  # classic, model.py:329-368
            x = x + _preact(_preact(x, W, i, 0, dtype), W, i, 1, dtype)
    trunk = x

    # PolicyHead.call (model.py:783-812) + GlobalPoolBias.call (:696-706)
    p = _conv(x, _t(W["policy.conv_p.w"], dtype))
    g = _mish(_bn(_conv(x, _t(W["policy.conv_g.w"], dtype)), W, "policy.gpool_bn", dtype))
    gp = _gpool(g)
    p = _mish(p + _dense(gp, W, "policy.gpool_dense", dtype)[:, :, None, None])
    pi2 = _conv(p, _t(W["policy.out_moves.w"], dtype)).reshape(N, 2, 361)
    pass2 = _dense(gp, W, "policy.out_pass", dtype) - 3
    pi_logits = torch.cat([pi2[:, 0], pass2[:, 0:1]], dim=1)
    opt = _conv(p, _t(W["policy.opt_moves.w"], dtype)).reshape(N, 361)
    opt_logits = torch.cat([opt, _dense(gp, W, "policy.opt_pass", dtype) - 3], dim=1)

    # ValueHead.call (model.py:887-979)
    v = _conv(x, _t(W["value.conv.w"], dtype))
    vp = _gpool(v)
    emb = _mish(_dense(vp, W, "value.oq_embed", dtype))
    go = _dense(emb, W, "value.oq_out", dtype)
    own = torch.tanh(_conv(v, _t(W["value.own.w"], dtype))).reshape(N, 361)
    gamma = _dense(_mish(_dense(vp, W, "value.gamma_pre", dtype)), W, "value.gamma_out", dtype)
    scores = 0.05 * torch.arange(-400, 400, dtype=dtype) + 0.025  # model.py:1223-1228
    vs = torch.cat([vp[:, None, :].expand(N, 800, vp.shape[1]),
                    scores[None, :, None].expand(N, 800, 1)], dim=2)
    sl = _dense(_mish(_dense(vs, W, "value.score_pre", dtype)), W, "value.score_out", dtype)
    score_logits = torch.clamp(F.softplus(gamma), max=10.0) * sl.reshape(N, 800)

    raw = torch.cat([pi_logits, opt_logits, go[:, 0:2], score_logits, own,
                     4 * torch.sigmoid(go[:, 5:6]), gamma], dim=1)
    return {
        "raw": raw.double().numpy(),
        "move_probs": torch.softmax(pi_logits, 1).double().numpy(),
        "value_probs": torch.softmax(go[:, 0:2], 1).double().numpy(),
        "score_probs": torch.softmax(score_logits, 1).double().numpy(),
        "opt_move_probs": torch.softmax(opt_logits, 1).double().numpy(),
        "trunk_nhwc": trunk.permute(0, 2, 3, 1).double().numpy(),
    }
