import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds libp3hip.so / liboracle.so in-tree if they are missing (cross-compiles on CPU)."""
    import __graft_entry__ as g
    from p3achygo_amd import engine
    need = [engine.LIB_PATH, os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "p3achygo_amd", "host", "libp3host.so")]
    if not all(os.path.exists(p) for p in need):
        g.build()
    return True


@pytest.fixture(scope="session")
def weight_files(tmp_path_factory):
    """name -> path of a seeded random-init .p3w (randomised BN stats), generated on demand."""
    from p3achygo_amd import netspec
    d = tmp_path_factory.mktemp("weights")
    cache = {}

    def get(name, randomize=True, peak=0.0):
        key = (name, randomize, peak)
        if key not in cache:
            cfg = netspec.CONFIGS[name]
            p = os.path.join(d, f"{name}_{int(randomize)}_{peak:g}.p3w")
            W = netspec.generate_weights(cfg, randomize=randomize)
            if peak:
                W = netspec.peak_policy(W, peak)
            netspec.save_p3w(p, cfg, W)
            cache[key] = p
        return cache[key]
    return get


def load_golden(name):
    import numpy as np
    from p3achygo_amd import features
    g = np.load(os.path.join(ROOT, "tests", "golden", f"nn_{name}.npz"))
    pos = np.frombuffer(g["features"].tobytes(), dtype=features.features_dtype()).copy()
    return g, pos
