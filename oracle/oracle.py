"""ctypes loader for oracle/liboracle.so (TEST INFRASTRUCTURE ONLY; see nn_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "nn_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.oracle_load.restype = C.c_void_p
        L.oracle_load.argtypes = [C.c_char_p]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_config.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.oracle_fill_inputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_forward_features.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


class OracleNet:
    def __init__(self, path: str):
        self._h = lib().oracle_load(path.encode())
        if not self._h:
            raise RuntimeError("oracle_load failed: " + path)
        cfg = (C.c_int * 9)()
        lib().oracle_config(self._h, cfg)
        self.channels = cfg[2]

    def close(self):
        if self._h and _LIB is not None:
            _LIB.oracle_free(self._h)
        self._h = None

    def __del__(self):
        self.close()

    @staticmethod
    def fill_inputs(feats_rec: np.ndarray):
        """LoadPlanes/LoadFeatures restatement: Features records -> (planes NHWC, scalars)."""
        n = len(feats_rec)
        feats_rec = np.ascontiguousarray(feats_rec)
        planes = np.zeros((n, 19, 19, 15), np.float32)
        sc = np.zeros((n, 8), np.float32)
        rec_size = feats_rec.dtype.itemsize
        for i in range(n):
            lib().oracle_fill_inputs(feats_rec.ctypes.data + i * rec_size,
                                     planes[i].ctypes.data, sc[i].ctypes.data)
        return planes, sc

    def forward_planes(self, planes: np.ndarray, scalars: np.ndarray, nthreads: int = 8,
                       want_trunk: bool = False):
        from p3achygo_amd.features import RAW_LEN, Result
        n = len(planes)
        planes = np.ascontiguousarray(planes, np.float32)
        scalars = np.ascontiguousarray(scalars, np.float32)
        res = (Result * n)()
        raw = np.zeros((n, RAW_LEN), np.float32)
        trunk = np.zeros((n, 361, self.channels), np.float32) if want_trunk else None
        lib().oracle_forward(self._h, n, planes.ctypes.data, scalars.ctypes.data,
                             C.addressof(res), raw.ctypes.data,
                             trunk.ctypes.data if want_trunk else None, nthreads)
        return res, raw, trunk

    def forward_features(self, feats_rec: np.ndarray, nthreads: int = 8):
        from p3achygo_amd.features import RAW_LEN, Result
        n = len(feats_rec)
        feats_rec = np.ascontiguousarray(feats_rec)
        res = (Result * n)()
        raw = np.zeros((n, RAW_LEN), np.float32)
        lib().oracle_forward_features(self._h, n, feats_rec.ctypes.data, C.addressof(res),
                                      raw.ctypes.data, nthreads)
        return res, raw
