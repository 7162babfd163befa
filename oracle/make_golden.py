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


def dump_nn(name, n_pos, seed):
    cfg = netspec.CONFIGS[name]
    W = netspec.generate_weights(cfg, seed=netspec.WEIGHT_SEED, randomize=True)
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
        os.path.join(GOLD, f"nn_{name}.npz"),
        features=np.frombuffer(pos.tobytes(), np.uint8), n_pos=n_pos, planes=planes.astype(np.uint8),
        scalars=sc, raw=ref["raw"], move_probs=ref["move_probs"], value_probs=ref["value_probs"],
        score_probs=ref["score_probs"], opt_move_probs=ref["opt_move_probs"],
        weight_checksum=np.array([wsum, wsq]))
    print(name, "ok", ref["raw"].shape)


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
    dump_nn("b8c128nbt", 2, 14)
    dump_nn("b12c256btl3", 2, 15)
