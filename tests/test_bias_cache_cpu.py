"""The bias cache of the serial search (p3achygo_amd/host/bias_cache.h; cc/mcts/bias_cache.h:51-206,
tree.h:23-32, gumbel.cc:729-777): local-pattern keys, the weighted observed-error arithmetic as a
known answer, the node-death hook, and a self-play game with it on (config/v4.json: lambda 0.3,
alpha 0.8 — BASELINE configs[0])."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
from p3achygo_amd import host_api

B, W = 1, -1


@pytest.fixture(scope="module")
def L(built):
    lib = host_api.lib()
    lib.p3host_test_local_pattern.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.p3host_test_bias_cache_math.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_void_p]
    return lib


def _game(L, moves):
    g = L.p3host_game_new(7.5)
    for i, j, c in moves:
        assert L.p3host_game_play(g, i, j, c)
    return g


def _pattern(L, g, other=None):
    grid, atari, ko = (np.zeros(25, np.int8) for _ in range(3))
    same = np.zeros(4, np.int32)
    ok = L.p3host_test_local_pattern(g, other, grid.ctypes.data, atari.ctypes.data, ko.ctypes.data, same.ctypes.data)
    return ok, grid.reshape(5, 5), atari.reshape(5, 5), ko.reshape(5, 5), same


def test_local_pattern_maps_and_keys(L):
    """5x5 window around the last move: stones / off-board, groups in atari, empty points the side to
    move may not play (bias_cache.h:64-124).  No pattern after a pass or before the second move."""
    g = _game(L, [(0, 1, B)])
    assert _pattern(L, g)[0] == 0                       # no move two plies back
    g = _game(L, [(0, 1, B), (19, 0, W)])
    assert _pattern(L, g)[0] == 0                       # last move is a pass
    # white stone at (0,0) in atari after B (0,1); last move W (5,5) elsewhere, then B (1,0) captures?
    # build: B(0,1) W(0,0) -> white corner stone has one liberty (1,0)
    g = _game(L, [(0, 1, B), (0, 0, W)])
    ok, grid, atari, ko, _ = _pattern(L, g)
    assert ok == 1
    # the window is centred on (0,0): rows/cols -2,-1 are off the board
    assert (grid[:2, :] == 2).all() and (grid[:, :2] == 2).all()
    assert grid[2, 2] == W and grid[2, 3] == B and grid[3, 2] == 0
    assert atari[2, 2] == 1 and atari[2, 3] == 0 and atari.sum() == 1
    assert ko.sum() == 0
    # a suicide point for the side to move counts as "ko" (illegal empty point):
    # black stones around (1,1)... W to move may not play (0,0) when B holds (0,1),(1,0) and it captures nothing
    g = _game(L, [(0, 1, B), (10, 10, W), (1, 0, B), (10, 11, W), (2, 2, B)])
    ok, grid, atari, ko, _ = _pattern(L, g)           # centred on (2,2), white to move
    assert ok == 1 and grid[2, 2] == B
    assert ko[0, 0] == 1 and ko.sum() == 1              # (0,0): suicide for white
    # same local situation reached in another game (stones far away differ): same hashes, same key
    h = _game(L, [(0, 1, B), (12, 12, W), (1, 0, B), (12, 13, W), (2, 2, B)])
    ok, *_rest, same = _pattern(L, g, h)
    assert list(same) == [1, 1, 1, 0]                   # maps equal; the key differs only by two-moves-ago
    h2 = _game(L, [(1, 0, B), (10, 10, W), (0, 1, B), (10, 11, W), (2, 2, B)])
    assert list(_pattern(L, g, h2)[4]) == [1, 1, 1, 1]


def _expected(alpha, lam, init, cv, cvv, extra):
    cv = [list(cv[:3]), list(cv[3:])]
    cvv = [list(cvv[:3]), list(cvv[3:])]
    err = wt = 0.0
    last = [[0.0, 0.0], [0.0, 0.0]]

    def update(k):
        nonlocal err, wt
        visits = sum(cv[k])
        util = sum(n * -v for n, v in zip(cv[k], cvv[k]) if n > 0)
        obs = init[k] - util / visits
        w = visits ** alpha
        err += obs * w - last[k][0]
        wt += w - last[k][1]
        last[k] = [obs * w, w]
        return lam * err / wt
    out = [update(0), update(1)]
    cv[0][0] += extra
    out.append(update(0))
    n0 = 1 + sum(cv[0])
    v0 = (init[0] - out[2] - sum(n * v for n, v in zip(cv[0], cvv[0]))) / n0
    err -= 0.8 * last[1][0]
    wt -= 0.8 * last[1][1]
    out += [lam * err / wt, 0.0, err, wt, v0, 0.0]
    return out


@pytest.mark.parametrize("alpha,lam", [(0.8, 0.3), (0.85, 0.45)])
def test_bias_cache_known_answer(L, alpha, lam):
    """Two nodes of one local pattern share an entry: each update replaces the node's own
    contribution (observed error x visits^alpha), the bias is lambda x error / weight, a dying node
    takes 0.8 of its contribution with it, RecomputeNodeStats takes the bias off the node's own
    estimate, unused entries are pruned."""
    init = np.array([0.30, -0.20], np.float32)
    cv = np.array([3, 0, 2, 1, 4, 0], np.int32)
    cvv = np.array([-0.10, 0.9, 0.25, 0.40, -0.35, 0.0], np.float32)
    out = np.zeros(9, np.float32)
    L.p3host_test_bias_cache_math(alpha, lam, init.ctypes.data, cv.ctypes.data, cvv.ctypes.data, 5, out.ctypes.data)
    want = _expected(alpha, lam, [float(x) for x in init], [int(x) for x in cv], [float(x) for x in cvv], 5)
    assert np.allclose(out, want, rtol=2e-5, atol=2e-6), (out, want)


def test_selfplay_game_with_bias_cache_on_cpu_engine(built, tmp_path):
    """Config C1's search settings (n=8, k=4, bias_cache_lambda 0.3 / alpha 0.8, config/v4.json) on
    a tiny random-init net through the CPU engine behind the C ABI: the game is reproducible,
    roots get a non-zero adjustment, entries are pruned as the tree is reaped, the search differs
    from the one without the cache, and lambda = 0 switches everything off."""
    from p3achygo_amd import netspec
    lib = os.path.join(ROOT, "oracle", "libp3cpu_engine.so")
    os.environ["P3CPU_THREADS"] = "1"
    cfg = netspec.CONFIGS["tiny"]
    w = str(tmp_path / "tiny.p3w")
    netspec.save_p3w(w, cfg, netspec.generate_weights(cfg, randomize=True))
    try:
        host_api.set_bias_cache(0.3, 0.8)
        mv1, b1, w1, ev1 = host_api.selfplay_one_game(w, 8, 4, 70, seed=11, engine_lib=lib)
        pruned, adj = host_api.last_bias_counters()
        mv2, *_ = host_api.selfplay_one_game(w, 8, 4, 70, seed=11, engine_lib=lib)
        assert len(mv1) == 70 and np.array_equal(mv1, mv2)
        assert pruned > 100 and adj > 0.05
    finally:
        host_api.set_bias_cache(0.0, 0.8)
    mv3, *_ = host_api.selfplay_one_game(w, 8, 4, 70, seed=11, engine_lib=lib)
    assert host_api.last_bias_counters() == (0, 0.0)
    assert len(mv3) == 70 and not np.array_equal(mv1, mv3)
