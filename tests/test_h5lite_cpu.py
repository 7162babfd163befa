"""p3achygo_amd/h5lite.py (numpy-only HDF5 subset reader) against files written by the HDF5
library itself (tests/golden/h5/make_h5_fixtures.py, h5py 3.3.0 / HDF5 1.10.6) — bit-exact, dtype and shape included."""
import io
import os
import zipfile

import numpy as np
import pytest

from p3achygo_amd import h5lite

H5 = os.path.join(os.path.dirname(__file__), "golden", "h5")
EXPECTED = np.load(os.path.join(H5, "expected.npz"))


def _check(tag, datasets):
    want = {k.split(":", 1)[1]: EXPECTED[k] for k in EXPECTED.files if k.startswith(tag + ":")}
    assert set(datasets) == set(want)
    for path, arr in want.items():
        got = datasets[path]
        assert got.dtype == arr.dtype and got.shape == arr.shape, path
        assert got.tobytes() == arr.tobytes(), path


def test_default_h5py_file_symbol_table_groups_and_every_layout():
    """libver "earliest" (what Keras writes): superblock 0, B-tree/heap groups (one of them spanning several
    symbol-table nodes), contiguous / compact / chunked data, gzip + shuffle + fletcher32, f16/f32/f64, ints,
    big-endian floats, a scalar, an empty array, a two-level chunk B-tree."""
    f = h5lite.File(os.path.join(H5, "earliest.h5"))
    ds = f.datasets()
    _check("e", ds)
    assert list(ds)[:3] == ["layers/batch_normalization/vars/0", "layers/batch_normalization/vars/1", "layers/conv2d/vars/0"]
    np.testing.assert_array_equal(f.read("/misc/u8"), EXPECTED["e:misc/u8"])
    with pytest.raises(KeyError):
        f.read("layers/nothing")


def test_latest_libver_file_link_messages_and_single_chunk_index():
    _check("l", h5lite.File(os.path.join(H5, "latest.h5")).datasets())


def test_bytes_input_and_keras_archive_container():
    with zipfile.ZipFile(os.path.join(H5, "tiny.keras")) as z:
        assert {"metadata.json", "config.json", "model.weights.h5"} <= set(z.namelist())
        _check("k", h5lite.File(z.read("model.weights.h5")).datasets())


def test_what_the_subset_does_not_cover_is_refused_not_misread():
    with pytest.raises(NotImplementedError, match="dense link storage"):
        h5lite.File(os.path.join(H5, "dense_group.h5")).datasets()
    with pytest.raises(h5lite.H5Error):
        h5lite.File(b"not an hdf5 file at all" * 100)
    raw = open(os.path.join(H5, "earliest.h5"), "rb").read()
    with pytest.raises(h5lite.H5Error):
        h5lite.File(raw[: len(raw) // 3]).datasets()    # truncated: an error, never garbage


# ---- `.keras` -> `.p3w` (p3achygo_amd/keras_import.py) ------------------------------------------------------
def test_keras_archive_import_writes_the_p3w_file_of_the_architecture_in_config_json(tmp_path):
    """tiny_p3achygo.keras: the reference's "tiny" architecture in the Keras 3 object-path layout (written with
    h5py from keras_map.object_path_map; tensor hashes recorded at generation).  The importer finds the
    architecture in config.json, maps all 123 tensors, leaves the optimizer / extra variables aside, and the
    `.p3w` it writes holds exactly the generated tensors."""
    import dataclasses
    import hashlib
    import json

    from p3achygo_amd import keras_import, netspec

    src = os.path.join(H5, "tiny_p3achygo.keras")
    dst = str(tmp_path / "tiny.p3w")
    assert keras_import.main([src, dst]) == 0
    cfg, unused = keras_import.import_checkpoint(src, dst)
    assert cfg == netspec.CONFIGS["tiny"]
    assert unused == ["layers/value_head/outcome_q_extra/vars/0", "optimizer/vars/0"]
    cfg2, tensors, _version = netspec.load_p3w(dst)
    assert dataclasses.replace(cfg2, name=cfg.name) == cfg    # the file carries the shape fields, not the name
    want = json.load(open(os.path.join(H5, "tiny_p3achygo_sha256.json")))
    assert set(tensors) == set(want) == {n for n, _, _ in netspec.tensor_specs(cfg)}
    for name, (shape, digest) in want.items():
        a = tensors[name]
        assert list(a.shape) == shape and a.dtype == np.float32
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == digest, name


def test_keras_import_refuses_what_does_not_fit():
    from p3achygo_amd import keras_import, keras_map, netspec

    ds, config = keras_import.read_archive(os.path.join(H5, "tiny_p3achygo.keras"))
    cfg = keras_import.config_from_arguments(keras_import.model_arguments({"a": [{"b": config}]}))
    assert cfg.name == "tiny"
    with pytest.raises(KeyError, match="does not hold the tensors of b8c128nbt"):
        keras_import.convert(ds, netspec.CONFIGS["b8c128nbt"])
    # a checkpoint whose groups are named differently from the map: the message shows the nearest dataset paths
    renamed = {k.replace("init_board_conv", "init_board_conv2d"): v for k, v in ds.items()}
    with pytest.raises(KeyError, match="nearest unmatched dataset paths: .*init_board_conv2d"):
        keras_import.convert(renamed, cfg)
    bad = dict(ds)
    first = keras_map.object_path_map(cfg)[0][0]
    bad[first] = bad[first][..., :-1]
    with pytest.raises(ValueError, match="shape"):
        keras_import.convert(bad, cfg)
    # a weights file keyed by layer names ("<layer>/<variable>:0") goes through keras_map.name_map
    by_name = {k: ds[p] for (k, n), (p, n2) in zip(sorted(keras_map.name_map(cfg), key=lambda r: r[1]),
                                                   sorted(keras_map.object_path_map(cfg), key=lambda r: r[1]))}
    out, unused = keras_import.convert({k + ":0": v for k, v in by_name.items()}, cfg)
    ref, _ = keras_import.convert(ds, cfg)
    assert unused == [] and all(np.array_equal(out[n], ref[n]) for n in ref)
    with pytest.raises(ValueError, match="--config"):
        keras_import.import_checkpoint(os.path.join(H5, "tiny.keras"), "/dev/null")


def test_object_paths_cover_every_architecture_and_carry_the_reference_documented_key():
    """One dataset path per tensor for every architecture; the value head sits under "layers/value_head/<attribute>",
    the key python/scripts/migrate_checkpoint.py:45-47 renames, the trunk under the `blocks` attribute."""
    from p3achygo_amd import keras_map, netspec

    for cfg in netspec.CONFIGS.values():
        rows = keras_map.object_path_map(cfg)
        assert len({k for k, _ in rows}) == len(rows)
        assert {p for _, p in rows} == {n for n, _, _ in netspec.tensor_specs(cfg)}
    rows = dict((p, k) for k, p in keras_map.object_path_map(netspec.CONFIGS["b12c256btl3"]))
    assert rows["value.oq_embed.w"] == "layers/value_head/outcome_q_embed/vars/0"
    assert rows["blocks.0.conv0.w"] == "blocks/bottleneck_residual_conv_block/blocks/conv_pre_activation/conv/vars/0"
    assert rows["blocks.6.bn4.var"] == "blocks/bottleneck_residual_conv_block_5/blocks/conv_pre_activation_4/norm_layer/vars/3"
    assert rows["blocks.9.dense.b"] == "blocks/broadcast_residual_block_1/blocks/broadcast_pre_act/dense/vars/1"
    nbt = dict((p, k) for k, p in keras_map.object_path_map(netspec.CONFIGS["b8c128nbt"]))
    assert nbt["blocks.1.conv4.w"] == "blocks/nbt_residual_block_1/blocks/classic_residual_block_1/blocks/conv_pre_activation_1/conv/vars/0"
