"""Search and self-play host on the CPU with the NullEvaluator (uniform policy — the
reference's NullEngine, cc/mcts/__tests__/search_test.cc:49-65) and a scripted evaluator."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def host(built):
    from p3achygo_amd import host_api
    return host_api


def test_gumbel_known_answer(host):
    """Scenario of cc/mcts/__tests__/gumbel_test.cc:74-123 (n=8, k=4, no noise): sequential
    halving spends exactly n visits, splits them 1/1/3/3, keeps the best two after round 1
    and picks (0,3); the network's own move is (0,0); child Qs reproduce the scripted values
    (plus the small score-utility term, |.| < 0.02)."""
    L = host.lib()
    cv = (C.c_int * 4)()
    cq = (C.c_float * 4)()
    nn, mc, rn = C.c_int(), C.c_int(), C.c_int()
    rc = L.p3host_test_scripted_search(8, 4, cv, cq, C.byref(nn), C.byref(mc), C.byref(rn))
    assert rc >> 8 == 8                       # visits spent == n
    assert list(cv) == [1, 1, 3, 3]
    assert nn.value == 0 and mc.value == 3
    assert rn.value == 8                      # root n = sum of child visits (gumbel.cc:550-554)
    for i in range(4):
        assert abs(cq[i] - (-0.5 + i / 3.0)) < 0.02


@pytest.mark.parametrize("n,k,expect,best", [(32, 4, 32, 3), (16, 2, 16, 1), (128, 8, 126, 3), (32, 5, 31, 3)])
def test_gumbel_visit_budget(host, n, k, expect, best):
    """Visits spent = sum over halving rounds of round(n / (rounds * k_r)) * k_r
    (gumbel.cc:388-473); the budget invariant the reference checks in search_test.cc:130-222."""
    L = host.lib()
    cv = (C.c_int * 4)()
    cq = (C.c_float * 4)()
    nn, mc, rn = C.c_int(), C.c_int(), C.c_int()
    rc = L.p3host_test_scripted_search(n, k, cv, cq, C.byref(nn), C.byref(mc), C.byref(rn))
    rounds = max(int(np.log2(k)), 1)
    spent, kk = 0, k
    while kk > 1:
        spent += int(np.floor(n / (rounds * kk) + 0.5)) * kk
        kk //= 2
    assert rc >> 8 == spent == expect
    assert mc.value == best                   # best-valued move among the top-k by prior


def _replay(host, moves):
    b = host.Board()
    passes = 0
    for m in moves:
        col = 1 if m > 0 else -1
        idx = abs(int(m)) - 1
        if idx == 361:
            b.pass_(col)
            passes += 1
        else:
            assert b.play(idx // 19, idx % 19, col), "illegal move in recorded game"
            passes = 0
    return b


def test_one_game_deterministic_and_legal(host):
    """BASELINE configs[0] plumbing case (1 thread, n=8 k=4): a whole game is reproducible
    from its seed, alternates colours, contains only legal moves and ends by two passes or the
    move cap; the recorded score equals a fresh rules-engine replay."""
    mv, bs, ws, ev = host.selfplay_one_game(None, 8, 4, 120, seed=42)
    mv2, bs2, ws2, ev2 = host.selfplay_one_game(None, 8, 4, 120, seed=42)
    assert np.array_equal(mv, mv2) and (bs, ws, ev) == (bs2, ws2, ev2)
    mv3, *_ = host.selfplay_one_game(None, 8, 4, 120, seed=43)
    assert not np.array_equal(mv, mv3)
    assert len(mv) <= 120 and ev > len(mv)
    assert all((m > 0) == (i % 2 == 0) for i, m in enumerate(mv))
    b = _replay(host, mv)
    assert len(mv) == 120 or b.is_game_over()
    rb, rw, _ = b.scores()
    assert (rb, rw) == (bs, ws)


def test_scheduler_with_null_engine(host):
    """The double-buffered scheduler drives many games on several threads; every evaluation
    is counted once and all games make progress."""
    st = host.selfplay_run(None, num_games=64, num_threads=4, seconds=2.5, default_n=8, default_k=4,
                           selected_n=8, selected_k=4, max_moves=16, warmup_batches=1, seed=3)
    assert st.positions > 64 and st.moves > 0 and st.batches > 2
    assert st.games > 0                       # 16-move cap: games finish and restart
    # one batch = one evaluation per game of the half
    assert abs(st.positions - st.batches * 32) <= 64


def test_puct_root_search(host):
    """SearchRootPuct (gumbel.cc:563-666) on the scripted position: exactly n playouts, the
    root's visit count grows by n (it is on the backup path, unlike the Gumbel root), the
    move is the arg-max of the new visit counts, and with enough playouts PUCT concentrates
    on the best-valued move (0,3)."""
    L = host.lib()
    cv = (C.c_int * 4)()
    cq = (C.c_float * 4)()
    nn, mc, rn = C.c_int(), C.c_int(), C.c_int()
    for n in (8, 64):
        rc = L.p3host_test_scripted_search(n, 0, cv, cq, C.byref(nn), C.byref(mc), C.byref(rn))
        assert rc >> 8 == n
        assert rn.value == 1 + n              # EvaluateRoot sets n = 1, then one per playout
        assert sum(cv) <= n and nn.value == 0
        assert mc.value == int(np.argmax(list(cv)))
    assert mc.value == 3 and cv[3] > cv[0]


def test_softmax_known_answers(host):
    """core::SoftmaxV known answers of cc/core/__tests__/vmath_test.cc:42-107."""
    L = host.lib()
    L.p3host_softmax.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    cases = [
        ([1.0, 1.5, 2.0], [0.18632372, 0.30719589, 0.50648039]),
        ([5.2, 1.3, -4.4, 4.5, -3.0, 7.2, 2.1, -4.0],
         [1.11714669e-01, 2.26131844e-03, 7.56629338e-06, 5.54758628e-02, 3.06828327e-05, 8.25465956e-01,
          5.03265673e-03, 1.12875833e-05]),
        ([5.2, 1.3, -4.4, 4.5, -3.0, 7.2, 2.1, -4.0, 9.8, 4.6],
         [9.17561645e-03, 1.85732016e-04, 6.21452908e-07, 4.55647628e-03, 2.52011581e-06, 6.77991447e-02,
          4.13354202e-04, 9.27098797e-07, 9.12829923e-01, 5.03568507e-03]),
        ([-149.944, -157.025, -158.732, -158.947, -160.693, -161.818, -161.623],
         [9.9884784e-01, 8.3996827e-04, 1.5237786e-04, 1.2289763e-04, 2.1442285e-05, 6.9612906e-06,
          8.4600651e-06]),
        ([1.0, 2.0, 3.0, 4.0], [0.03205860, 0.08714432, 0.23688282, 0.64391426]),
        ([2.0, 2.0, 2.0, 2.0], [0.25, 0.25, 0.25, 0.25]),
        ([100.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]),
    ]
    for x, want in cases:
        a = np.array(x, np.float32)
        out = np.zeros_like(a)
        L.p3host_softmax(a.ctypes.data, out.ctypes.data, len(a))
        assert np.allclose(out, want, rtol=2e-5, atol=1e-9), x


def _mv(color, i, j):
    return (i * 19 + j + 1) * color


def test_sgf_golden_strings(host):
    """Known answers of cc/sgf/__tests__/sgf_serializer_test.cc:47-112 (minus the trailing
    newline, which the recorder appends: sgf_recorder.cc:282-285).  Coordinates are emitted
    row letter first (sgf_serializer.cc:27-32)."""
    assert host.sgf_from_moves([]) == "(;FF[4]GM[1]KM[7.5]RE[?]PB[testB]PW[testW])"
    assert host.sgf_from_moves([], write_result=True) == "(;FF[4]GM[1]KM[7.5]RE[W+7.5]PB[testB]PW[testW])"
    moves = [_mv(1, 0, 0), _mv(-1, 1, 0), _mv(1, 0, 1), _mv(-1, 1, 1)]
    assert host.sgf_from_moves(moves) == "(;FF[4]GM[1]KM[7.5]RE[?]PB[testB]PW[testW];B[aa];W[ba];B[ab];W[bb])"
    white_win = [_mv(1, 0, 2), _mv(-1, 0, 3), _mv(1, 1, 2), _mv(-1, 1, 3), _mv(1, 2, 2), _mv(-1, 2, 4),
                 _mv(1, 2, 0), _mv(-1, 1, 5), _mv(1, 2, 1), _mv(-1, 0, 5)]
    assert host.sgf_from_moves(white_win, write_result=True) == (
        "(;FF[4]GM[1]KM[7.5]RE[W+5.5]PB[testB]PW[testW];B[ac];W[ad];B[bc];W[bd];B[cc];W[ce];B[ca];W[bf];B[cb];W[af])")
    black_win = [_mv(1, 0, 2), _mv(1, 1, 2), _mv(1, 2, 2), _mv(1, 2, 0), _mv(1, 2, 1), _mv(-1, 2, 3)]
    assert host.sgf_from_moves(black_win, write_result=True) == (
        "(;FF[4]GM[1]KM[7.5]RE[B+0.5]PB[testB]PW[testW];B[ac];B[bc];B[cc];B[ca];B[cb];W[cd])")
    assert host.sgf_from_moves([_mv(1, 3, 3), 362 * -1]) .endswith(";B[dd];W[])")   # pass = empty value


def test_sgf_recorder_files(host, tmp_path):
    """Batch files follow cc/data/filename_format.h:27-31: gen%03d_b%03d_g%03d_%s.sgf with one
    game per line and a matching .done file (sgf_recorder.cc:266-326)."""
    import os
    import re
    host.set_recorder(str(tmp_path), gen=7, worker_id="w3", flush_interval=4)
    try:
        st = host.selfplay_run(None, num_games=16, num_threads=2, seconds=1.5, default_n=4, default_k=2,
                               selected_n=4, selected_k=2, max_moves=12, warmup_batches=0, seed=11)
    finally:
        host.set_recorder("")
    files = sorted(os.listdir(tmp_path))
    sgfs = [f for f in files if f.endswith(".sgf")]
    assert sgfs and all(re.fullmatch(r"gen007_b\d{3}_g\d{3}_w3\.sgf", f) for f in sgfs)
    total = 0
    for f in sgfs:
        assert f[:-4] + ".done" in files
        lines = open(os.path.join(tmp_path, f)).read().split("\n")
        assert lines[-1] == ""
        n_games = int(re.search(r"_g(\d{3})_", f).group(1))
        assert len(lines) - 1 == n_games
        for ln in lines[:-1]:
            assert ln.startswith("(;FF[4]GM[1]KM[7.5]RE[") and "PB[p3achygo]PW[p3achygo]" in ln and ln.endswith(")")
        total += n_games
    assert total >= st.games
