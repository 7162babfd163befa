"""Emits the committed golden fixtures under tests/golden/ (run in the build container).

    python oracle/make_golden.py

1. model_configs.json — the architecture table read from the reference's dependency-free
   python/model_config.py + python/constants.py (imported from /root/reference/python; the
   rest of the reference's Python imports TensorFlow and cannot be imported here).
2. nn_<config>.npz — seeded inputs (Features records) and float64 outputs of the
   independent PyTorch restatement (oracle/torch_restatement.py) for seeded random-init
   weights (netspec.generate_weights(cfg, seed, randomize=True)); a weight checksum guards
   against generator drift.  These pin oracle/nn_oracle.c and the HIP engine.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_restatement as tr  # noqa: E402
from p3achygo_amd import features, netspec  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def dump_model_configs():
    ref = "/root/reference/python"
    if not os.path.isdir(ref):
        print("reference not present; keeping committed model_configs.json")
        return
    sys.path.insert(0, ref)
    import constants as rc  # noqa
    import model_config as mc  # noqa
    out = {"constants": {"BOARD_LEN": rc.BOARD_LEN, "NUM_MOVES": rc.NUM_MOVES,
                         "SCORE_RANGE": rc.SCORE_RANGE, "NUM_V_BUCKETS": rc.NUM_V_BUCKETS,
                         "num_input_planes": rc.num_input_planes(1),
                         "num_input_features": rc.num_input_features(1)},
           "configs": {}}
    for name in mc.CONFIG_OPTIONS:
        c = mc.ModelConfig.from_str(name)
        if c.is_transformer:
            continue
        out["configs"][name] = {
            "blocks": c.kBlocks, "channels": c.kChannels,
            "bottleneck_channels": c.kBottleneckChannels, "head_channels": c.kHeadChannels,
            "c_val": c.kCVal, "broadcast_interval": c.kBroadcastInterval,
            "inner_layers": c.kInnerBottleneckLayers, "block_type": c.kTrunkBlockType}
    with open(os.path.join(GOLD, "model_configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def wide_positions(n_pos, seed):
    """Mid / late-game, pass-heavy and both-komi-sign positions, both colours to move."""
    q = n_pos // 4
    parts = [
        features.random_positions(q, seed=seed, max_moves=220, n_games=q),
        features.random_positions(q, seed=seed + 1, min_moves=120, max_moves=340, n_games=q,
                                  komis=(7.5, -7.5, 0.5, 6.5, -20.5, 40.5)),
        features.random_positions(q, seed=seed + 2, min_moves=20, max_moves=300, n_games=q, pass_prob=0.2,
                                  komis=(7.5, -7.5, 5.5)),
        features.random_positions(n_pos - 3 * q, seed=seed + 3, min_moves=250, max_moves=420,
                                  n_games=n_pos - 3 * q, pass_prob=0.05, komis=(7.5, -0.5)),
    ]
    return np.concatenate(parts)


def dump_nn(name, n_pos, seed, wide=False, peak=0.0, tag="", store=np.float64):
    cfg = netspec.CONFIGS[name]
    W = netspec.generate_weights(cfg, seed=netspec.WEIGHT_SEED, randomize=True)
    if peak:
        W = netspec.peak_policy(W, peak)
    if wide:
        pos = wide_positions(n_pos, seed)
    else:
        pos = features.random_positions(n_pos, seed=seed, max_moves=220, n_games=n_pos)
    # independent (numpy) statement of LoadPlanes/LoadFeatures for the fixture inputs
    planes = np.zeros((n_pos, 19, 19, 15), np.float32)
    sc = np.zeros((n_pos, 8), np.float32)
    for k in range(n_pos):
        f = pos[k]
        col = int(f["color"])
        for ch, key in ((0, "board"), (7, "stones_atari"), (9, "stones_two_liberties"),
                        (11, "stones_three_liberties"), (13, "stones_laddered")):
            g = f[key].reshape(19, 19)
            planes[k, :, :, ch] = (g == col)
            planes[k, :, :, ch + 1] = (g == -col)
        for t in range(5):
            i, j = int(f["last_moves"][t]["i"]), int(f["last_moves"][t]["j"])
            if (i, j) == (19, 0):
                sc[k, 2 + t] = 1
            elif (i, j) != (-1, -1):
                planes[k, i, j, 2 + t] = 1
        sc[k, 0 if col == 1 else 1] = 1
        sc[k, 7] = (-1.0 if col == 1 else 1.0) * float(f["komi"]) / 15.0
    ref = tr.forward(cfg, W, planes, sc)
    wsum = float(sum(float(w.astype(np.float64).sum()) for w in W.values()))
    wsq = float(sum(float((w.astype(np.float64) ** 2).sum()) for w in W.values()))
    np.savez_compressed(
        os.path.join(GOLD, f"nn_{name}{tag}.npz"),
        features=np.frombuffer(pos.tobytes(), np.uint8), n_pos=n_pos, planes=planes.astype(np.uint8),
        scalars=sc, raw=ref["raw"].astype(store), move_probs=ref["move_probs"].astype(store),
        value_probs=ref["value_probs"].astype(store),
        score_probs=ref["score_probs"].astype(store), opt_move_probs=ref["opt_move_probs"].astype(store),
        weight_checksum=np.array([wsum, wsq]), peak=np.array(float(peak)))
    print(name + tag, "ok", ref["raw"].shape, "max move prob", float(ref["move_probs"].max()))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    dump_model_configs()
    dump_nn("tiny", 4, 11)
    dump_nn("test_b3c128btl2", 3, 12)
    dump_nn("test_b3c128nbt", 3, 13)
    dump_nn("test_b3c256btl1", 3, 16)
    dump_nn("test_b3c256nbt", 3, 17)
    dump_nn("test_b3c384btl3", 3, 18)
    dump_nn("test_b3c384nbt", 3, 19)
    dump_nn("test_b3c192classic", 3, 20)
    dump_nn("test_b5c256nbt_i2", 3, 25)
    dump_nn("test_b5c128btl1_i2", 3, 26)
    dump_nn("test_b5c256btl2_i2", 3, 27)
    dump_nn("test_b10c256btl1_i2", 3, 28)
    # full-size BASELINE architectures: wide position sets; outputs stored as float32 (the
    # float64 results rounded once: 6e-8 relative, three orders below any tolerance)
    dump_nn("b8c128nbt", 8, 14, wide=True, store=np.float32)                  # C1
    dump_nn("b12c128btl3", 8, 21, wide=True, store=np.float32)                # C2
    dump_nn("b12c256btl3", 32, 15, wide=True, store=np.float32)               # C3 / C4 (headline)
    dump_nn("b12c256btl3", 8, 22, wide=True, peak=12.0, tag="_peaked", store=np.float32)
    dump_nn("b10c384nbt", 4, 23, wide=True, store=np.float32)                 # C5
    dump_nn("b14c384btl3", 4, 24, wide=True, store=np.float32)                # C5
