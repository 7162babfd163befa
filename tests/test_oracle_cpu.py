"""CPU suite: the oracle against the committed golden vectors, the net spec, the weight
file, and that the C-ABI library loads and exports every declared symbol (no GPU calls)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

SMALL = ["tiny", "test_b3c128btl2", "test_b3c128nbt", "test_b3c256btl1", "test_b3c256nbt", "test_b3c384btl3",
         "test_b3c384nbt", "test_b3c192classic", "test_b5c256btl2_i2", "test_b10c256btl1_i2"]


def test_model_configs_match_reference_table():
    """netspec.CONFIGS reproduces python/model_config.py:62-172 (captured fixture)."""
    from p3achygo_amd import netspec
    with open(os.path.join(ROOT, "tests", "golden", "model_configs.json")) as f:
        ref = json.load(f)
    assert ref["constants"] == {"BOARD_LEN": 19, "NUM_MOVES": 362, "SCORE_RANGE": 800,
                                "NUM_V_BUCKETS": 51, "num_input_planes": 15,
                                "num_input_features": 8}
    for name, c in ref["configs"].items():
        mine = netspec.CONFIGS[name]
        assert (mine.blocks, mine.channels, mine.head_channels, mine.c_val,
                mine.broadcast_interval, mine.block_type) == (
            c["blocks"], c["channels"], c["head_channels"], c["c_val"], c["broadcast_interval"],
            c["block_type"]), name
        if c["block_type"] != "classic":
            assert mine.bottleneck_channels == c["bottleneck_channels"], name
        if c["block_type"] == "btl":
            assert mine.inner_layers == c["inner_layers"], name


def test_flops_match_survey():
    """b12c256btl3 = 4.077 GFLOP/position, 3x3 convs 3.194 GFLOP."""
    from p3achygo_amd import netspec
    t, c3 = netspec.flops_per_position(netspec.CONFIGS["b12c256btl3"])
    assert abs(t / 1e9 - 4.077) < 2e-3 and abs(c3 / 1e9 - 3.194) < 1e-3
    t, _ = netspec.flops_per_position(netspec.CONFIGS["b8c128nbt"])
    assert abs(t / 1e9 - 0.868) < 2e-3
    t, _ = netspec.flops_per_position(netspec.CONFIGS["b10c384nbt"])
    assert abs(t / 1e9 - 9.274) < 3e-3


def test_p3w_roundtrip(tmp_path):
    from p3achygo_amd import netspec
    cfg = netspec.CONFIGS["tiny"]
    W = netspec.generate_weights(cfg, randomize=True)
    p = str(tmp_path / "t.p3w")
    netspec.save_p3w(p, cfg, W)
    cfg2, W2, ver = netspec.load_p3w(p)
    assert ver == 1 and cfg2.blocks == cfg.blocks and cfg2.channels == cfg.channels
    assert set(W2) == set(W)
    for k in W:
        assert np.array_equal(W[k], W2[k]), k


@pytest.mark.parametrize("name", SMALL)
def test_weight_generator_is_stable(name):
    """The seeded generator reproduces the weights the golden vectors were made with."""
    from p3achygo_amd import netspec
    g, _ = load_golden(name)
    W = netspec.generate_weights(netspec.CONFIGS[name], randomize=True)
    wsum = sum(float(w.astype(np.float64).sum()) for w in W.values())
    wsq = sum(float((w.astype(np.float64) ** 2).sum()) for w in W.values())
    assert np.allclose([wsum, wsq], g["weight_checksum"], rtol=1e-9)


def test_feature_planes_channel_map(built):
    """Plane/scalar layout of LoadPlanes/LoadFeatures (go_features.cc:10-61); channel map
    known answers from cc/nn/__tests__/nn_board_utils_test.cc:84-112: 0/1 stones, 2-6 last
    five moves, 7/8 atari, 9/10 two libs, 11/12 three libs, (v1) 13/14 ladder; scalars 0/1
    colour, 2-6 pass flags, 7 komi."""
    from oracle import oracle
    from p3achygo_amd import features
    f = np.zeros(1, dtype=features.features_dtype())
    f["bsize"], f["color"], f["komi"] = 19, -1, 7.5     # white to move
    f["board"][0][3 * 19 + 4] = -1                      # own (white) stone at (3,4)
    f["board"][0][5 * 19 + 6] = 1                       # opponent (black) stone at (5,6)
    f["stones_atari"][0][3 * 19 + 4] = -1
    f["stones_two_liberties"][0][5 * 19 + 6] = 1
    f["stones_three_liberties"][0][0] = -1
    f["stones_laddered"][0][360] = 1
    moves = [(-1, -1), (19, 0), (2, 2), (19, 0), (10, 11)]  # noop, pass, move, pass, move
    for t, (i, j) in enumerate(moves):
        f["last_moves"][0][t]["i"], f["last_moves"][0][t]["j"] = i, j
    planes, sc = oracle.OracleNet.fill_inputs(f)
    p = planes[0]
    assert p[3, 4, 0] == 1 and p[5, 6, 1] == 1 and p[..., 0].sum() == 1 and p[..., 1].sum() == 1
    assert p[3, 4, 7] == 1 and p[..., 8].sum() == 0
    assert p[5, 6, 10] == 1 and p[..., 9].sum() == 0
    assert p[0, 0, 11] == 1 and p[..., 12].sum() == 0
    assert p[18, 18, 14] == 1 and p[..., 13].sum() == 0
    assert p[2, 2, 4] == 1 and p[10, 11, 6] == 1 and p[..., 2:7].sum() == 2
    assert list(sc[0][:7]) == [0, 1, 0, 1, 0, 1, 0]
    assert abs(sc[0][7] - 7.5 / 15.0) < 1e-7             # +komi/15 for white
    f["color"] = 1
    _, sc = oracle.OracleNet.fill_inputs(f)
    assert list(sc[0][:2]) == [1, 0] and abs(sc[0][7] + 0.5) < 1e-7


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_golden(built, weight_files, name):
    """oracle/nn_oracle.c (fp32) vs the float64 PyTorch restatement's committed outputs."""
    from oracle import oracle
    g, pos = load_golden(name)
    net = oracle.OracleNet(weight_files(name))
    planes, sc = net.fill_inputs(pos)
    assert np.array_equal(planes.astype(np.uint8), g["planes"]) and np.allclose(sc, g["scalars"])
    res, raw = net.forward_features(pos, nthreads=4)
    assert np.abs(raw - g["raw"]).max() < 2e-5
    for i in range(len(pos)):
        for key in ("move_probs", "value_probs", "score_probs", "opt_move_probs"):
            got = np.ctypeslib.as_array(getattr(res[i], key))
            assert np.abs(got - g[key][i]).max() < 1e-6, (key, i)
        assert abs(res[i].err2_outcome - g["raw"][i][1887]) < 1e-5
        assert np.array_equal(np.ctypeslib.as_array(res[i].move_logits), raw[i][:362])


@pytest.mark.parametrize("name", ["b8c128nbt", "b12c128btl3", "b12c256btl3", "b12c256btl3_peaked", "b10c384nbt",
                                  "b14c384btl3"])
def test_oracle_matches_golden_full_size(built, weight_files, name):
    """The BASELINE architectures at full depth (C1, C2, C3 incl. the peaked-policy set, C5):
    fp32 C oracle vs the float64 restatement's committed outputs (stored as float32)."""
    from oracle import oracle
    from p3achygo_amd import netspec
    g, pos = load_golden(name)
    peak = float(g["peak"])
    base = name[:-len("_peaked")] if peak else name
    W = netspec.generate_weights(netspec.CONFIGS[base], randomize=True)
    if peak:
        W = netspec.peak_policy(W, peak)
    wsum = sum(float(w.astype(np.float64).sum()) for w in W.values())
    wsq = sum(float((w.astype(np.float64) ** 2).sum()) for w in W.values())
    assert np.allclose([wsum, wsq], g["weight_checksum"], rtol=1e-9)
    net = oracle.OracleNet(weight_files(base, peak=peak))
    res, raw = net.forward_features(pos, nthreads=8)
    tol = np.full(1889, 1e-4)
    if peak:
        tol[:724] *= peak
    assert (np.abs(raw - g["raw"]) <= tol).all()
    for i in range(len(pos)):
        for key in ("move_probs", "value_probs", "score_probs", "opt_move_probs"):
            got = np.ctypeslib.as_array(getattr(res[i], key))
            assert np.abs(got - g[key][i]).max() < (2e-4 if peak else 2e-6), (key, i)


def test_oracle_agrees_with_torch_restatement_fresh_init(built, weight_files):
    """Fresh-Keras init (BN identity, zero biases), independent float64 restatement."""
    from oracle import oracle, torch_restatement as tr
    from p3achygo_amd import features, netspec
    name = "test_b3c128nbt"
    cfg = netspec.CONFIGS[name]
    W = netspec.generate_weights(cfg, randomize=False)
    pos = features.random_positions(2, seed=77)
    net = oracle.OracleNet(weight_files(name, randomize=False))
    planes, sc = net.fill_inputs(pos)
    _, raw, trunk = net.forward_planes(planes, sc, nthreads=2, want_trunk=True)
    ref = tr.forward(cfg, W, planes, sc)
    assert np.abs(raw - ref["raw"]).max() < 2e-5
    assert np.abs(trunk - ref["trunk_nhwc"].reshape(2, 361, -1)).max() < 2e-5


def test_cabi_exports_every_declared_symbol(built):
    """libp3hip.so loads without a GPU and exports each function include/p3hip.h declares."""
    from p3achygo_amd import engine
    hdr = open(os.path.join(ROOT, "include", "p3hip.h")).read()
    declared = set(re.findall(r"\b(p3hip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    L = C.CDLL(engine.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), sym


def test_pod_layouts_match_header(built):
    from p3achygo_amd import features
    assert C.sizeof(features.Features) == 1860 and C.sizeof(features.Result) == 7568
    assert features.Features.board.offset == 12 and features.Features.last_moves.offset == 376
    assert features.Features.stones_atari.offset == 416
    assert features.Features.stones_laddered.offset == 1499
    assert features.Result.opt_move_probs.offset == 6112


def test_engine_path_rules(tmp_path):
    """KindFromEnginePath (engine_factory.cc:16-35) + the `.p3w` rule; VERSION default 1."""
    from p3achygo_amd import engine
    p = tmp_path / "m.p3w"; p.write_bytes(b"x")
    t = tmp_path / "m.trt"; t.write_bytes(b"x")
    o = tmp_path / "m.bin"; o.write_bytes(b"x")
    (tmp_path / "_trt").mkdir()
    assert engine.kind_from_engine_path(str(p)) == engine.Kind.kHip
    assert engine.kind_from_engine_path(str(t)) == engine.Kind.kTrt
    assert engine.kind_from_engine_path(str(o)) == engine.Kind.kUnknown
    assert engine.kind_from_engine_path(str(tmp_path / "_trt")) == engine.Kind.kTFTrt
    assert engine.kind_from_engine_path(str(tmp_path)) == engine.Kind.kTF
    assert engine.get_version_from_model_path(str(p)) == 1
    (tmp_path / "VERSION").write_text("0\n")
    assert engine.get_version_from_model_path(str(p)) == 0
    with pytest.raises(engine.EngineError):
        engine.create_engine(engine.Kind.kUnknown, str(o), 4, 1)


def test_engine_fails_loudly_without_gpu(built, weight_files):
    """No CPU fallback: on a box without a HIP device engine creation must raise."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from p3achygo_amd import engine
    with pytest.raises(engine.EngineError):
        engine.HipEngine(weight_files("test_b3c128btl2"), 4)


def test_bad_weight_files_fail_creation_with_a_message(built, weight_files, tmp_path):
    """A truncated file, a file whose header names an architecture its tensor table lacks, and a
    mis-shaped tensor all fail p3hip_create with a message (the library never aborts the host);
    the weight plan is built before any HIP call, so this holds with and without a GPU."""
    from p3achygo_amd import engine, netspec
    good = open(weight_files("test_b3c128btl2"), "rb").read()
    cut = tmp_path / "cut.p3w"
    cut.write_bytes(good[:len(good) // 2])
    with pytest.raises(engine.EngineError, match="truncated"):
        engine.HipEngine(str(cut), 4)
    cfg = netspec.CONFIGS["test_b3c128btl2"]
    W = netspec.generate_weights(cfg)
    # header says 3 blocks, table holds the tensors of a 2-block net: blocks.2.* are missing
    cfg2 = netspec.NetConfig("two", 2, 128, 64, 32, 32, 3, 2, "btl")
    W2 = netspec.generate_weights(cfg2)
    p = tmp_path / "short.p3w"
    netspec.save_p3w(str(p), cfg2, W2)
    blob = bytearray(p.read_bytes())
    blob[8:12] = (3).to_bytes(4, "little")          # nblocks field of the header
    p.write_bytes(bytes(blob))
    with pytest.raises(engine.EngineError, match=r"lacks tensors.*blocks\.2\."):
        engine.HipEngine(str(p), 4)
    # a tensor of the wrong size
    Wbad = dict(W)
    p3 = tmp_path / "shape.p3w"
    specs = netspec.tensor_specs(cfg)
    netspec.save_p3w(str(p3), cfg, Wbad)
    blob = bytearray(p3.read_bytes())
    ent0 = 44 + [n for n, _, _ in specs].index("policy.conv_p.w") * 76
    blob[ent0 + 52 + 12:ent0 + 52 + 16] = (16).to_bytes(4, "little")   # dims[3]: 32 -> 16
    p3.write_bytes(bytes(blob))
    with pytest.raises(engine.EngineError, match=r"policy\.conv_p\.w \(wrong size\)"):
        engine.HipEngine(str(p3), 4)


@pytest.mark.parametrize("name", ["b12c256btl3", "b8c128nbt", "b15c192_classic"])
def test_keras_name_map_covers_every_tensor(name):
    """Keras layer/variable paths (python/model.py `name=` arguments) -> .p3w tensor names: one row
    per tensor of the architecture, no duplicates on either side, and the renaming of a synthetic
    checkpoint (arrays of the Keras shapes under the Keras paths) reproduces the .p3w tensors.  A
    real `.keras` archive cannot be read here (no h5py / Keras): the map is the deliverable."""
    from p3achygo_amd import keras_map, netspec
    cfg = netspec.CONFIGS[name]
    rows = keras_map.name_map(cfg)
    specs = netspec.tensor_specs(cfg)
    assert len(rows) == len(specs) == len({k for k, _ in rows}) == len({p for _, p in rows})
    assert {p for _, p in rows} == {n for n, _, _ in specs}
    W = netspec.generate_weights(cfg, randomize=True)
    back = {p: k for k, p in rows}
    ckpt = {back[n]: W[n] for n in W}
    out = keras_map.rename(cfg, ckpt)
    assert all(np.array_equal(out[n], W[n]) for n in W)
    assert ("bottleneck_res_0/res_id_inner_1/conv/kernel", "blocks.0.conv2.w") in rows or cfg.block_type != "btl"
    ckpt.pop(next(iter(ckpt)))
    with pytest.raises(KeyError):
        keras_map.rename(cfg, ckpt)
