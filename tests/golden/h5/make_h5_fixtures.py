"""Writes the HDF5 fixtures of tests/test_h5lite_cpu.py with the HDF5 library itself.

Run with an interpreter that has h5py (in this image: /opt/conda/bin/python3.9, h5py 3.3.0 / HDF5 1.10.6):
    /opt/conda/bin/python3.9 tests/golden/h5/make_h5_fixtures.py
The interpreter the test-suite runs under has no h5py; the tests read these files with p3achygo_amd/h5lite.py and
compare with expected.npz (the arrays as numpy wrote them).
"""
import io
import json
import os
import zipfile

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20260)
expected = {}


def put(f, tag, path, arr, **kw):
    f.create_dataset(path, data=arr, **kw)
    expected[f"{tag}:{path}"] = np.asarray(arr).astype(np.asarray(arr).dtype.newbyteorder("="))


# 1. what h5py writes by default (libver "earliest": superblock 0, symbol-table groups, layout version 3)
with h5py.File(os.path.join(HERE, "earliest.h5"), "w") as f:
    put(f, "e", "layers/conv2d/vars/0", rng.standard_normal((3, 3, 4, 5)).astype(np.float32))
    put(f, "e", "layers/batch_normalization/vars/0", rng.standard_normal(5).astype(np.float32))
    put(f, "e", "layers/batch_normalization/vars/1", rng.standard_normal(5).astype(np.float32))
    put(f, "e", "layers/dense/vars/0", rng.standard_normal((7, 3)).astype(np.float16))
    put(f, "e", "layers/dense/vars/1", rng.standard_normal(3).astype(np.float64))
    put(f, "e", "optimizer/vars/0", np.array(12345678901, np.int64))                      # scalar
    put(f, "e", "misc/u8", rng.integers(0, 255, (4, 6)).astype(np.uint8))
    put(f, "e", "misc/i32", rng.integers(-1000, 1000, 9).astype(np.int32))
    put(f, "e", "misc/big_endian", rng.standard_normal((2, 5)).astype(">f4"))
    put(f, "e", "misc/empty", np.zeros((0, 4), np.float32))
    put(f, "e", "misc/gzip_shuffle", rng.standard_normal((20, 30)).astype(np.float32), chunks=(7, 11), compression="gzip", shuffle=True)
    put(f, "e", "misc/chunked_plain", rng.standard_normal((5, 33)).astype(np.float32), chunks=(2, 8))
    put(f, "e", "misc/fletcher", rng.integers(0, 1 << 40, (6, 6)).astype(np.int64), chunks=(3, 6), fletcher32=True)
    put(f, "e", "misc/many_chunks", rng.standard_normal((64, 70)).astype(np.float16), chunks=(4, 5), compression="gzip")  # a two-level chunk B-tree
    for i in range(40):                                                                       # more than one symbol-table node
        put(f, "e", f"wide/vars/{i}", rng.standard_normal(i % 5 + 1).astype(np.float32))
    # compact layout through the low-level API
    arr = rng.standard_normal((3, 4)).astype(np.float32)
    space = h5py.h5s.create_simple(arr.shape)
    dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
    dcpl.set_layout(h5py.h5d.COMPACT)
    did = h5py.h5d.create(f["misc"].id, b"compact", h5py.h5t.IEEE_F32LE, space, dcpl)
    did.write(h5py.h5s.ALL, h5py.h5s.ALL, arr)
    expected["e:misc/compact"] = arr
    f.attrs["keras_version"] = "3.3.3"

# 2. libver "latest": superblock 3, version-2 object headers, link messages, layout version 4
with h5py.File(os.path.join(HERE, "latest.h5"), "w", libver="latest") as f:
    put(f, "l", "layers/conv2d/vars/0", rng.standard_normal((1, 1, 6, 2)).astype(np.float32))
    put(f, "l", "layers/dense/vars/0", rng.standard_normal((4, 4)).astype(np.float32))
    put(f, "l", "layers/dense/vars/1", rng.standard_normal(4).astype(np.float32))
    put(f, "l", "one_chunk", rng.standard_normal((6, 5)).astype(np.float32), chunks=(6, 5), compression="gzip")
    put(f, "l", "scalar", np.float32(2.5))

# 3. a group beyond eight links under libver "latest" is stored densely (fractal heap): must be refused, not misread
with h5py.File(os.path.join(HERE, "dense_group.h5"), "w", libver="latest") as f:
    for i in range(12):
        f.create_dataset(f"g/{i}", data=np.float32(i))

# 4. a `.keras` archive in the Keras 3 layout (zip: metadata.json, config.json, model.weights.h5; datasets
#    <object path>/vars/<n>), for a made-up two-layer model: the container format, not the reference's model
buf = io.BytesIO()
with h5py.File(buf, "w") as f:
    put(f, "k", "layers/conv2d/vars/0", rng.standard_normal((3, 3, 2, 4)).astype(np.float32))
    put(f, "k", "layers/dense/vars/0", rng.standard_normal((4, 2)).astype(np.float32))
    put(f, "k", "layers/dense/vars/1", rng.standard_normal(2).astype(np.float32))
    f.create_group("optimizer/vars")
with zipfile.ZipFile(os.path.join(HERE, "tiny.keras"), "w", zipfile.ZIP_DEFLATED) as z:
    z.writestr("metadata.json", json.dumps({"keras_version": "3.3.3", "date_saved": "2026-01-01@00:00:00"}))
    z.writestr("config.json", json.dumps({"class_name": "Sequential", "config": {"name": "tiny", "layers": []}}))
    z.writestr("model.weights.h5", buf.getvalue())

np.savez(os.path.join(HERE, "expected.npz"), **expected)
print(len(expected), "arrays")

# 5. an archive in the layout keras_map.object_path_map describes, for the reference's "tiny" architecture
#    (python/model_config.py), weights from netspec.generate_weights with randomised BN statistics; plus the
#    optimizer group and a non-inference variable a real checkpoint also carries.  Written FROM the map: it pins the
#    importer's plumbing (archive, config.json lookup, shapes, dtypes, BN order), not Keras's naming.
import sys
sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..", "..", "..")))
from p3achygo_amd import keras_map, netspec   # numpy only

cfg = netspec.CONFIGS["tiny"]
W = netspec.generate_weights(cfg, randomize=True)
buf = io.BytesIO()
with h5py.File(buf, "w") as f:
    for path, name in keras_map.object_path_map(cfg):
        f.create_dataset(path, data=W[name].astype(np.float32))
    f.create_dataset("optimizer/vars/0", data=np.int64(1234))
    f.create_dataset("layers/value_head/outcome_q_extra/vars/0", data=np.zeros((3, 3), np.float32))
config = {"module": "model", "class_name": "P3achyGoModel", "registered_name": "p3achygo>P3achyGoModel",
          "config": {"board_len": 19, "num_input_planes": 15, "num_input_features": 8, "num_blocks": cfg.blocks,
                     "num_channels": cfg.channels, "num_bottleneck_channels": cfg.bottleneck_channels,
                     "num_head_channels": cfg.head_channels, "c_val": cfg.c_val, "bottleneck_length": cfg.inner_layers + 2,
                     "conv_size": 3, "broadcast_interval": cfg.broadcast_interval, "trunk_block_type": cfg.block_type,
                     "name": "tiny"}}
with zipfile.ZipFile(os.path.join(HERE, "tiny_p3achygo.keras"), "w", zipfile.ZIP_DEFLATED) as z:
    z.writestr("metadata.json", json.dumps({"keras_version": "3.3.3"}))
    z.writestr("config.json", json.dumps(config))
    z.writestr("model.weights.h5", buf.getvalue())
import hashlib
with open(os.path.join(HERE, "tiny_p3achygo_sha256.json"), "w") as f:
    json.dump({n: [list(W[n].shape), hashlib.sha256(np.ascontiguousarray(W[n], np.float32).tobytes()).hexdigest()] for n in sorted(W)}, f, indent=0)
print("tiny_p3achygo.keras", os.path.getsize(os.path.join(HERE, "tiny_p3achygo.keras")), "bytes")
