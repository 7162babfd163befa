"""`.keras` checkpoint -> `.p3w` weight file.

    python -m p3achygo_amd.keras_import model.keras model.p3w [--config b12c256btl3]

The reference saves `model.save(path)` archives (python/model_utils.py:197-204): a zip of `config.json`,
`metadata.json` and `model.weights.h5`.  This importer reads the archive with zipfile + p3achygo_amd/h5lite.py
(no Keras, no h5py), finds the architecture in `config.json` (the P3achyGoModel constructor arguments, as
python/scripts/migrate_checkpoint.py:50-68 does), maps every dataset to its `.p3w` tensor through
keras_map.object_path_map (Keras 3 object paths) or, for weight files keyed by layer names, keras_map.name_map,
checks every shape against netspec.tensor_specs, and writes the `.p3w` file the engine loads.  `.p3w` keeps the
Keras tensor layouts, so no tensor is transposed.

Validation status: the container and HDF5 reading are pinned by files the HDF5 library wrote
(tests/test_h5lite_cpu.py); the path map is pinned by an archive of the Keras 3 layout written from this map
(tests/golden/h5/tiny_p3achygo.keras) and by the one key the reference documents — no archive saved by the
reference's own Keras model exists in this environment.
"""
from __future__ import annotations

import argparse
import json
import sys
import zipfile
from typing import Dict, Optional, Tuple

import numpy as np

from . import h5lite, keras_map, netspec


def read_archive(path: str) -> Tuple[Dict[str, np.ndarray], Optional[dict]]:
    """({dataset path: array}, config.json or None) of a `.keras` archive or a bare weights `.h5`."""
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            names = set(z.namelist())
            if "model.weights.h5" not in names:
                raise ValueError(f"{path}: no model.weights.h5 in the archive")
            config = json.loads(z.read("config.json")) if "config.json" in names else None
            return h5lite.File(z.read("model.weights.h5")).datasets(), config
    return h5lite.File(path).datasets(), None


def model_arguments(config: Optional[dict]) -> Optional[dict]:
    """The P3achyGoModel constructor arguments inside config.json, wherever they are nested."""
    if isinstance(config, dict):
        if config.get("class_name") == "P3achyGoModel":
            return config.get("config", {})
        for v in config.values():
            r = model_arguments(v)
            if r:
                return r
    elif isinstance(config, list):
        for v in config:
            r = model_arguments(v)
            if r:
                return r
    return None


def config_from_arguments(args: dict) -> netspec.NetConfig:
    """NetConfig of a checkpoint's constructor arguments (model.py:1128-1150; bottleneck_length counts the two
    1x1 convs, model.py:403)."""
    block_type = args.get("trunk_block_type", "btl")
    inner = int(args["bottleneck_length"]) - 2 if block_type == "btl" else 2
    want = dict(blocks=int(args["num_blocks"]), channels=int(args["num_channels"]),
                bottleneck_channels=int(args["num_bottleneck_channels"]), head_channels=int(args["num_head_channels"]),
                c_val=int(args["c_val"]), broadcast_interval=int(args["broadcast_interval"]), inner_layers=inner,
                block_type=block_type)
    for cfg in netspec.CONFIGS.values():
        if all(getattr(cfg, k) == v for k, v in want.items()):
            return cfg
    return netspec.NetConfig(name=str(args.get("name", "imported")), **want)


def _strip(path: str) -> str:
    return path[:-2] if path.endswith(":0") else path


def convert(datasets: Dict[str, np.ndarray], cfg: netspec.NetConfig) -> Tuple[Dict[str, np.ndarray], list]:
    """({p3w name: float32 array}, datasets left unused) — every tensor of `cfg` exactly once, shapes checked."""
    specs = {n: tuple(s) for n, s, _ in netspec.tensor_specs(cfg)}
    ds = {_strip(k): v for k, v in datasets.items()}
    tried = []
    for label, rows in (("Keras 3 object paths", keras_map.object_path_map(cfg)), ("layer-name paths", keras_map.name_map(cfg))):
        missing = [k for k, _ in rows if k not in ds]
        if missing:
            # name the dataset paths closest to the first absent keys: a real checkpoint whose groups are
            # named differently from this map (it has met no file the reference saved) shows here how
            import difflib
            spare = sorted(set(ds) - {k for k, _ in rows})
            near = "; ".join(f"{k} ~ {difflib.get_close_matches(k, spare, n=2, cutoff=0.3) or 'nothing alike'}" for k in missing[:4])
            tried.append(f"{label}: {len(missing)} of {len(rows)} absent (nearest unmatched dataset paths: {near})")
            continue
        out: Dict[str, np.ndarray] = {}
        for k, name in rows:
            a = np.asarray(ds[k])
            if tuple(a.shape) != specs[name]:
                raise ValueError(f"{k}: shape {tuple(a.shape)}, {name} of {cfg.name} is {specs[name]}")
            if not np.issubdtype(a.dtype, np.floating):
                raise ValueError(f"{k}: dtype {a.dtype}")
            out[name] = a.astype(np.float32)
        lacking = set(specs) - set(out)
        if lacking:
            raise KeyError(f"path map does not cover {sorted(lacking)}")
        used = {k for k, _ in rows}
        return out, sorted(k for k in ds if k not in used)
    raise KeyError(f"checkpoint does not hold the tensors of {cfg.name} ({'; '.join(tried)}); it has e.g. {sorted(ds)[:4]}")


def import_checkpoint(src: str, dst: str, config_name: Optional[str] = None) -> Tuple[netspec.NetConfig, list]:
    datasets, config = read_archive(src)
    if config_name:
        cfg = netspec.CONFIGS[config_name]
    else:
        args = model_arguments(config)
        if not args:
            raise ValueError("no P3achyGoModel arguments in config.json: name the architecture with --config")
        cfg = config_from_arguments(args)
    tensors, unused = convert(datasets, cfg)
    netspec.save_p3w(dst, cfg, tensors)
    return cfg, unused


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--config", default=None, help="architecture name (netspec.CONFIGS); default: read config.json")
    a = ap.parse_args(argv)
    cfg, unused = import_checkpoint(a.src, a.dst, a.config)
    print(f"{a.dst}: {cfg.name}, {len(netspec.tensor_specs(cfg))} tensors" + (f"; {len(unused)} datasets not part of the inference graph" if unused else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
