"""Python mirror of the reference's engine interface over the p3hip C ABI.

    nn::Engine            cc/nn/engine/engine.h:22-43      -> class HipEngine
    CreateEngine          cc/nn/engine/engine_factory.cc:56-73 -> create_engine
    KindFromEnginePath    cc/nn/engine/engine_factory.cc:16-35 -> kind_from_engine_path
    GetVersionFromModelPath  engine_factory.cc:37-53       -> get_version_from_model_path

Method names, argument meaning and failure behaviour follow the reference: the reference
aborts (CHECK / LOG(FATAL)) on engine failures, so every non-zero status of the C ABI is
raised as `EngineError` here.
"""
from __future__ import annotations

import ctypes as C
import enum
import os

import numpy as np

from .features import RAW_LEN, Features, Result

_HERE = os.path.dirname(os.path.abspath(__file__))
# P3HIP_LIB lets tools/gpu_ab.py time two builds of the kernels side by side.
LIB_PATH = os.environ.get("P3HIP_LIB") or os.path.join(_HERE, "csrc", "libp3hip.so")

EXPORTS = [
    "p3hip_create", "p3hip_create_error", "p3hip_destroy", "p3hip_kind", "p3hip_path",
    "p3hip_batch_size", "p3hip_load_slot", "p3hip_run", "p3hip_get_slot", "p3hip_get_ownership",
    "p3hip_last_error", "p3hip_forward_resident", "p3hip_upload", "p3hip_sync", "p3hip_get_raw",
    "p3hip_time_trunk_kernel", "p3hip_flops_per_position", "p3hip_graph_state",
    "p3hip_cache_enable", "p3hip_load_slot_keyed", "p3hip_get_slot_keyed", "p3hip_cache_stats",
    "p3hip_blockw_stamps", "p3hip_debug_x",
]

FLAG_RUN_ALL_SLOTS = 2
FLAG_SHARED_DEVICE = 4
FLAG_LAUNCH_GRAPH = 8


class EngineError(RuntimeError):
    pass


class Kind(enum.IntEnum):
    """Engine::Kind (engine.h:24-30) extended with kHip."""
    kUnknown = 0
    kTrt = 1
    kTF = 2
    kTFTrt = 3
    kTFXla = 4
    kHip = 5


_lib = None


def lib():
    """Loads libp3hip.so; fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.p3hip_create.restype = C.c_void_p
        L.p3hip_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.p3hip_create_error.restype = C.c_char_p
        L.p3hip_destroy.argtypes = [C.c_void_p]
        L.p3hip_kind.argtypes = [C.c_void_p]
        L.p3hip_path.restype = C.c_char_p
        L.p3hip_path.argtypes = [C.c_void_p]
        L.p3hip_batch_size.argtypes = [C.c_void_p]
        L.p3hip_load_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.p3hip_run.argtypes = [C.c_void_p]
        L.p3hip_get_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.p3hip_get_ownership.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.p3hip_last_error.restype = C.c_char_p
        L.p3hip_last_error.argtypes = [C.c_void_p]
        L.p3hip_forward_resident.argtypes = [C.c_void_p, C.c_int]
        L.p3hip_upload.argtypes = [C.c_void_p]
        L.p3hip_sync.argtypes = [C.c_void_p]
        L.p3hip_get_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.p3hip_cache_enable.argtypes = [C.c_void_p, C.c_int]
        L.p3hip_load_slot_keyed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
        L.p3hip_get_slot_keyed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.p3hip_cache_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.p3hip_time_trunk_kernel.restype = C.c_double
        L.p3hip_time_trunk_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double),
                                              C.POINTER(C.c_char_p)]
        L.p3hip_flops_per_position.argtypes = [C.c_void_p, C.POINTER(C.c_double),
                                               C.POINTER(C.c_double)]
        _lib = L
    return _lib


def kind_from_engine_path(path: str) -> Kind:
    """engine_factory.cc:16-35 with the added rule `*.p3w` file -> kHip."""
    if os.path.isfile(path):
        ext = os.path.splitext(path)[1]
        if ext == ".p3w":
            return Kind.kHip
        if ext == ".trt":
            return Kind.kTrt
        if ext == ".pb":
            return Kind.kTFXla
        return Kind.kUnknown
    if os.path.basename(os.path.normpath(path)) == "_trt":
        return Kind.kTFTrt
    return Kind.kTF


def get_version_from_model_path(path: str) -> int:
    """engine_factory.cc:37-53: sibling VERSION file, default 1."""
    parent = os.path.dirname(path) if os.path.isfile(path) else path
    vf = os.path.join(parent, "VERSION")
    if os.path.isfile(vf):
        try:
            return int(open(vf).read().split()[0])
        except (ValueError, IndexError):
            pass
    return 1


class HipEngine:
    """nn::Engine over libp3hip.so (engine.h:22-43)."""

    def __init__(self, path: str, batch_size: int, version: int = 1, device: int = 0,
                 flags: int = 0):
        self._L = lib()
        self._h = self._L.p3hip_create(path.encode(), batch_size, version, device, flags)
        if not self._h:
            raise EngineError("p3hip_create: " + self._L.p3hip_create_error().decode())
        self.batch_size = batch_size

    # -- reference surface --------------------------------------------------------------
    def kind(self) -> Kind:
        return Kind(self._L.p3hip_kind(self._h))

    def path(self) -> str:
        return self._L.p3hip_path(self._h).decode()

    def LoadBatch(self, batch_id: int, features) -> None:
        ptr = features.ctypes.data if isinstance(features, np.ndarray) else C.addressof(features)
        self._ck(self._L.p3hip_load_slot(self._h, batch_id, ptr), "LoadBatch")

    def RunInference(self) -> None:
        self._ck(self._L.p3hip_run(self._h), "RunInference")

    def GetBatch(self, batch_id: int, result: Result = None) -> Result:
        result = result if result is not None else Result()
        self._ck(self._L.p3hip_get_slot(self._h, batch_id, C.addressof(result)), "GetBatch")
        return result

    def GetOwnership(self, batch_id: int) -> np.ndarray:
        own = np.zeros(361, np.float32)
        self._ck(self._L.p3hip_get_ownership(self._h, batch_id, own.ctypes.data), "GetOwnership")
        return own

    # -- on-device NN cache (include/p3hip.h) ---------------------------------------------
    def EnableCache(self, log2_entries: int) -> None:
        self._ck(self._L.p3hip_cache_enable(self._h, log2_entries), "EnableCache")

    def LoadBatchKeyed(self, batch_id: int, features, key_lo: int, key_hi: int, symmetry: int = 0) -> None:
        ptr = features.ctypes.data if isinstance(features, np.ndarray) else C.addressof(features)
        self._ck(self._L.p3hip_load_slot_keyed(self._h, batch_id, ptr, key_lo, key_hi, symmetry), "LoadBatchKeyed")

    def GetBatchKeyed(self, batch_id: int, result: Result = None):
        """(result, symmetry of the returned result, came from the table)"""
        result = result if result is not None else Result()
        sym, hit = C.c_int(0), C.c_int(0)
        self._ck(self._L.p3hip_get_slot_keyed(self._h, batch_id, C.addressof(result), C.byref(sym), C.byref(hit)), "GetBatchKeyed")
        return result, sym.value, bool(hit.value)

    def cache_stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        self._L.p3hip_cache_stats(self._h, out)
        return {"lookups": out[0], "hits": out[1], "stored": out[2], "entries": out[3]}

    # -- measurement / test hooks -------------------------------------------------------
    def load_all(self, feats_rec: np.ndarray) -> None:
        feats_rec = np.ascontiguousarray(feats_rec)
        sz = feats_rec.dtype.itemsize
        for i in range(len(feats_rec)):
            self._ck(self._L.p3hip_load_slot(self._h, i, feats_rec.ctypes.data + i * sz), "LoadBatch")

    def upload(self) -> None:
        self._ck(self._L.p3hip_upload(self._h), "upload")

    def forward_resident(self, n: int) -> None:
        self._ck(self._L.p3hip_forward_resident(self._h, n), "forward_resident")

    def sync(self) -> None:
        self._ck(self._L.p3hip_sync(self._h), "sync")

    def get_raw(self, batch_id: int) -> np.ndarray:
        raw = np.zeros(RAW_LEN, np.float32)
        self._ck(self._L.p3hip_get_raw(self._h, batch_id, raw.ctypes.data), "get_raw")
        return raw

    def time_trunk_kernel(self, n_positions: int, iters: int):
        fl = C.c_double(0)
        name = C.c_char_p()
        ms = self._L.p3hip_time_trunk_kernel(self._h, n_positions, iters, C.byref(fl), C.byref(name))
        if ms < 0:
            raise EngineError("time_trunk_kernel: " + self._L.p3hip_last_error(self._h).decode())
        return ms, fl.value, (name.value or b"").decode()

    def debug_x(self, n: int, channels: int) -> np.ndarray:
        """residual stream after the last forward pass as [n][channels][361] floats"""
        out = np.zeros(n * channels * 361, np.float32)
        self._L.p3hip_debug_x.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self._ck(self._L.p3hip_debug_x(self._h, out.ctypes.data, n), "debug_x")
        return out.reshape(n, channels // 8, 361, 8).transpose(0, 1, 3, 2).reshape(n, channels, 361)

    def blockw_stamps(self) -> np.ndarray:
        """s_memtime stamps of k_blockw's _diag twin (P3HIP_BLOCKW=1 P3HIP_BLOCKW_DIAG=1): [wg 8][block 16][wave 4][24]"""
        out = np.zeros(8 * 16 * 4 * 24, np.uint64)
        self._L.p3hip_blockw_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if self._L.p3hip_blockw_stamps(self._h, out.ctypes.data, out.size) != 0:
            raise EngineError("no k_blockw stamps (engine not created with P3HIP_BLOCKW_DIAG)")
        return out.reshape(8, 16, 4, 24)

    def graph_state(self) -> int:
        """P3HIP_FLAG_LAUNCH_GRAPH: 1 replaying the captured forward pass, 0 not (yet), -1 capture failed."""
        self._L.p3hip_graph_state.argtypes = [C.c_void_p]
        return int(self._L.p3hip_graph_state(self._h))

    def flops_per_position(self):
        t, c = C.c_double(0), C.c_double(0)
        self._L.p3hip_flops_per_position(self._h, C.byref(t), C.byref(c))
        return t.value, c.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.p3hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int, what: str) -> None:
        if rc != 0:
            raise EngineError(f"{what} failed (rc={rc}): " + self._L.p3hip_last_error(self._h).decode())


def create_engine(kind: Kind, path: str, batch_size: int, version: int, device: int = 0,
                  flags: int = 0) -> HipEngine:
    """CreateEngine (engine_factory.cc:56-73); only kHip is served by this package."""
    if kind == Kind.kHip:
        return HipEngine(path, batch_size, version, device, flags)
    raise EngineError(f"Unknown Engine Kind {kind!r} (the reference LOG(FATAL)s here)")
