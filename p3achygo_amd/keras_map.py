"""Keras layer/variable names of the reference model -> `.p3w` tensor names.

The reference saves `.keras` archives (zip -> model.weights.h5; python/model_utils.py:197-204).
Two maps are kept here, both taken from python/model.py (cited per row) and both a pure renaming — `.p3w`
keeps the Keras tensor layouts (conv kernels HWIO, Dense (in, out), BatchNormalization gamma / beta /
moving_mean / moving_variance), so nothing is transposed:
  * `name_map`: layer `name=` arguments joined by "/", for weight files keyed by layer names;
  * `object_path_map`: the dataset paths of a Keras 3 `model.weights.h5` (attribute names, see below).
p3achygo_amd/keras_import.py reads the archive (zipfile + h5lite, no Keras / h5py) and applies them.  No
checkpoint saved by the reference exists in this build's environment, so neither map has met a real file.

Path convention used here: layer names joined by "/" from the model root, then the Keras variable
name, e.g. "bottleneck_res_3/res_id_inner_0/conv/kernel".  A ConvBlock (model.py:203-292) owns
`conv` (Conv2D, no bias) and `norm_layer` (BatchNormalization when use_var_norm=False, which is
what every shipped config uses); ConvPreActivation applies norm_layer -> mish -> conv.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import netspec

BN_VARS = (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "mean"), ("moving_variance", "var"))


def _conv_block(rows, keras_prefix: str, p3w_block: str, idx: int):
    """ConvBlock `keras_prefix` = conv `idx` of a `.p3w` block: its pre-activation BN and kernel."""
    for kv, pv in BN_VARS:
        rows.append((f"{keras_prefix}/norm_layer/{kv}", f"{p3w_block}.bn{idx}.{pv}"))
    rows.append((f"{keras_prefix}/conv/kernel", f"{p3w_block}.conv{idx}.w"))


def name_map(cfg: netspec.NetConfig) -> List[Tuple[str, str]]:
    """(keras path, p3w tensor name) for every tensor of architecture `cfg`."""
    rows: List[Tuple[str, str]] = []
    # P3achyGoModel.__init__, model.py:1155-1164
    rows.append(("init_board_conv/kernel", "init_conv.w"))
    rows.append(("init_game_layer/kernel", "init_game.w"))
    rows.append(("init_game_layer/bias", "init_game.b"))
    for i in range(cfg.blocks):
        b = f"blocks.{i}"
        kind = cfg.block_kind(i)
        if kind == "broadcast":      # BroadcastResidualBlock, model.py:570-608; names :592-603, :522-526
            k = f"broadcast_res_{i}"
            _conv_block(rows, f"{k}/broadcast_conv_first", b, 0)
            rows.append((f"{k}/broadcast_mix/broadcast_linear/kernel", f"{b}.dense.w"))
            rows.append((f"{k}/broadcast_mix/broadcast_linear/bias", f"{b}.dense.b"))
            _conv_block(rows, f"{k}/broadcast_conv_last", b, 1)
        elif kind == "btl":          # BottleneckResidualConvBlock, model.py:372-425; names :400-409
            k = f"bottleneck_res_{i}"
            _conv_block(rows, f"{k}/res_id_reduce_dim_begin", b, 0)
            for j in range(cfg.inner_layers):
                _conv_block(rows, f"{k}/res_id_inner_{j}", b, 1 + j)
            _conv_block(rows, f"{k}/res_id_expand_dim_end", b, cfg.inner_layers + 1)
        elif kind == "nbt":          # NbtResidualBlock, model.py:430-486; names :447-469, inner :350
            k = f"nbt_res_{i}"
            _conv_block(rows, f"{k}/nbt_reduce_dim", b, 0)
            for r in range(2):
                for j in range(2):
                    _conv_block(rows, f"{k}/nbt_res{r}/res_id_inner_{j}", b, 1 + 2 * r + j)
            _conv_block(rows, f"{k}/nbt_expand_dim", b, 5)
        elif kind == "classic":      # ClassicResidualBlock, model.py:329-368; name :1030
            k = f"classic_res_{i}"
            for j in range(2):
                _conv_block(rows, f"{k}/res_id_inner_{j}", b, j)
        else:
            raise ValueError(kind)
    # PolicyHead, model.py:745-778; GlobalPoolBias :670-683
    ph = "policy_head"
    rows.append((f"{ph}/policy_conv_p/kernel", "policy.conv_p.w"))
    rows.append((f"{ph}/policy_conv_g/kernel", "policy.conv_g.w"))
    for kv, pv in BN_VARS:
        rows.append((f"{ph}/policy_gpool/batch_norm_gpool/{kv}", f"policy.gpool_bn.{pv}"))
    rows.append((f"{ph}/policy_gpool/dense/kernel", "policy.gpool_dense.w"))
    rows.append((f"{ph}/policy_gpool/dense/bias", "policy.gpool_dense.b"))
    rows.append((f"{ph}/policy_output_moves/kernel", "policy.out_moves.w"))
    for keras, p3w in (("policy_output_pass", "policy.out_pass"), ("policy_soft_pass", "policy.soft_pass"),
                       ("policy_optimistic_pass", "policy.opt_pass")):
        rows.append((f"{ph}/{keras}/kernel", p3w + ".w"))
        rows.append((f"{ph}/{keras}/bias", p3w + ".b"))
    rows.append((f"{ph}/policy_soft_moves/kernel", "policy.soft_moves.w"))
    rows.append((f"{ph}/policy_optimistic_moves/kernel", "policy.opt_moves.w"))
    # ValueHead, model.py:846-879
    vh = "value_head"
    rows.append((f"{vh}/value_conv/kernel", "value.conv.w"))
    rows.append((f"{vh}/value_conv_ownership/kernel", "value.own.w"))
    for keras, p3w in (("value_outcome_q_biases", "value.oq_embed"), ("value_outcome_q_output", "value.oq_out"),
                       ("value_outcome_mcts_dist", "value.mcts_dist"), ("value_gamma_pre", "value.gamma_pre"),
                       ("value_gamma_output", "value.gamma_out"), ("value_score_distribution_pre", "value.score_pre"),
                       ("value_score_distribution_output", "value.score_out")):
        rows.append((f"{vh}/{keras}/kernel", p3w + ".w"))
        rows.append((f"{vh}/{keras}/bias", p3w + ".b"))
    return rows


# ---- Keras 3 object paths ---------------------------------------------------------------------------
# `model.weights.h5` inside a `.keras` archive is not keyed by layer names.  keras.src.saving.saving_lib walks
# the object graph: every KerasSaveable child is stored under the ATTRIBUTE name that holds it, attributes in
# sorted order, each object once (first visit wins); a list attribute is a container whose members are named by
# the snake-cased class name with a per-container counter ("conv_pre_activation", "conv_pre_activation_1", ...);
# a layer's variables are datasets vars/0, vars/1, ... in `weights` order (Conv2D: kernel; Dense: kernel, bias;
# BatchNormalization: gamma, beta, moving_mean, moving_variance).  For a subclassed keras.Model the `layers`
# property is walked as such a container, which is why the reference's own migration script addresses the value
# head as "layers/value_head/<attribute>" (python/scripts/migrate_checkpoint.py:3-9,45-47) while the trunk, held
# by the earlier-sorting attribute `blocks` (model.py:1166), sits under "blocks/".  Attribute names below are the
# reference's: ConvBlock.conv / .norm_layer (model.py:227-232), ResidualBlock.blocks (:318), Broadcast.dense
# (:524), GlobalPoolBias.g_norm_layer / .dense (:673-683), PolicyHead (:748-778), ValueHead (:852-879),
# P3achyGoModel.init_board_conv / .init_game_layer / .blocks / .policy_head / .value_head (:1155-1193).
# No `.keras` file of the reference exists in this build's environment: the walk is restated from the Keras
# sources' published behaviour and pinned only by that one documented key.
BN_ORDER = ("gamma", "beta", "mean", "var")
_CLASS = {"btl": "bottleneck_residual_conv_block", "nbt": "nbt_residual_block", "classic": "classic_residual_block",
          "broadcast": "broadcast_residual_block"}


class _Counter:
    """Names of the members of one saved container: snake-cased class name, `_n` from the second on."""

    def __init__(self):
        self.used: Dict[str, int] = {}

    def __call__(self, cls: str) -> str:
        if cls in self.used:
            self.used[cls] += 1
            return f"{cls}_{self.used[cls]}"
        self.used[cls] = 0
        return cls


def _conv_block_obj(rows, prefix: str, p3w_block: str, idx: int):
    rows.append((f"{prefix}/conv/vars/0", f"{p3w_block}.conv{idx}.w"))
    for n, pv in enumerate(BN_ORDER):
        rows.append((f"{prefix}/norm_layer/vars/{n}", f"{p3w_block}.bn{idx}.{pv}"))


def object_path_map(cfg: netspec.NetConfig) -> List[Tuple[str, str]]:
    """(dataset path inside model.weights.h5, p3w tensor name) for every tensor of architecture `cfg`."""
    rows: List[Tuple[str, str]] = []
    top = _Counter()
    for i in range(cfg.blocks):
        b = f"blocks.{i}"
        kind = cfg.block_kind(i)
        k = "blocks/" + top(_CLASS[kind]) + "/blocks"
        inner = _Counter()
        if kind == "broadcast":
            _conv_block_obj(rows, f"{k}/{inner('conv_pre_activation')}", b, 0)
            mix = f"{k}/{inner('broadcast_pre_act')}/dense"
            rows.append((f"{mix}/vars/0", f"{b}.dense.w"))
            rows.append((f"{mix}/vars/1", f"{b}.dense.b"))
            _conv_block_obj(rows, f"{k}/{inner('conv_pre_activation')}", b, 1)
        elif kind == "btl":
            for j in range(cfg.inner_layers + 2):
                _conv_block_obj(rows, f"{k}/{inner('conv_pre_activation')}", b, j)
        elif kind == "nbt":
            _conv_block_obj(rows, f"{k}/{inner('conv_pre_activation')}", b, 0)
            for r in range(2):
                res = f"{k}/{inner('classic_residual_block')}/blocks"
                pair = _Counter()
                for j in range(2):
                    _conv_block_obj(rows, f"{res}/{pair('conv_pre_activation')}", b, 1 + 2 * r + j)
            _conv_block_obj(rows, f"{k}/{inner('conv_pre_activation')}", b, 5)
        else:   # classic
            for j in range(2):
                _conv_block_obj(rows, f"{k}/{inner('conv_pre_activation')}", b, j)
    rows.append(("init_board_conv/vars/0", "init_conv.w"))
    rows.append(("init_game_layer/vars/0", "init_game.w"))
    rows.append(("init_game_layer/vars/1", "init_game.b"))
    ph = "layers/policy_head"
    rows.append((f"{ph}/conv_g/vars/0", "policy.conv_g.w"))
    rows.append((f"{ph}/conv_p/vars/0", "policy.conv_p.w"))
    rows.append((f"{ph}/gpool/dense/vars/0", "policy.gpool_dense.w"))
    rows.append((f"{ph}/gpool/dense/vars/1", "policy.gpool_dense.b"))
    for n, pv in enumerate(BN_ORDER):
        rows.append((f"{ph}/gpool/g_norm_layer/vars/{n}", f"policy.gpool_bn.{pv}"))
    rows.append((f"{ph}/optimistic_policy_moves/vars/0", "policy.opt_moves.w"))
    rows.append((f"{ph}/optimistic_policy_pass/vars/0", "policy.opt_pass.w"))
    rows.append((f"{ph}/optimistic_policy_pass/vars/1", "policy.opt_pass.b"))
    rows.append((f"{ph}/output_moves/vars/0", "policy.out_moves.w"))
    rows.append((f"{ph}/output_pass/vars/0", "policy.out_pass.w"))
    rows.append((f"{ph}/output_pass/vars/1", "policy.out_pass.b"))
    rows.append((f"{ph}/soft_policy_moves/vars/0", "policy.soft_moves.w"))
    rows.append((f"{ph}/soft_policy_pass/vars/0", "policy.soft_pass.w"))
    rows.append((f"{ph}/soft_policy_pass/vars/1", "policy.soft_pass.b"))
    vh = "layers/value_head"
    rows.append((f"{vh}/conv/vars/0", "value.conv.w"))
    rows.append((f"{vh}/conv_ownership/vars/0", "value.own.w"))
    for attr, p3w in (("gamma_output", "value.gamma_out"), ("gamma_pre", "value.gamma_pre"),
                      ("outcome_mcts_dist", "value.mcts_dist"), ("outcome_q_embed", "value.oq_embed"),
                      ("outcome_q_output", "value.oq_out"), ("score_output", "value.score_out"),
                      ("score_pre", "value.score_pre")):
        rows.append((f"{vh}/{attr}/vars/0", p3w + ".w"))
        rows.append((f"{vh}/{attr}/vars/1", p3w + ".b"))
    return rows


def rename(cfg: netspec.NetConfig, keras_vars: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """{keras path: array} -> {p3w name: float32 array}, checked against netspec.tensor_specs(cfg):
    every tensor present exactly once with the expected shape (Keras layouts are kept)."""
    specs = {n: s for n, s, _ in netspec.tensor_specs(cfg)}
    out: Dict[str, np.ndarray] = {}
    for kpath, pname in name_map(cfg):
        if kpath not in keras_vars:
            raise KeyError(f"checkpoint lacks {kpath} (-> {pname})")
        a = np.asarray(keras_vars[kpath], dtype=np.float32)
        if tuple(a.shape) != tuple(specs[pname]):
            raise ValueError(f"{kpath}: shape {a.shape}, {pname} expects {specs[pname]}")
        out[pname] = a
    missing = set(specs) - set(out)
    if missing:
        raise KeyError(f"name map does not cover {sorted(missing)}")
    return out
