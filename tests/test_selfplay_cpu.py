"""Search and self-play host on the CPU with the NullEvaluator (uniform policy — the
reference's NullEngine, cc/mcts/__tests__/search_test.cc:49-65) and a scripted evaluator."""
import ctypes as C

import numpy as np
import pytest



@pytest.fixture(autouse=True, scope="module")
def _ladder_throughput_mode(built):
    """These tests exercise scheduling and search plumbing over the uniform NullEvaluator, whose
    search trees wander into chaotic positions where the exact ladder read-out takes seconds
    (millions of nodes): they run the host's opt-in ladder work bound.  Bit-exactness of the
    default mode is pinned in tests/test_rules_cpu.py."""
    from p3achygo_amd import host_api as _h
    _h.set_ladder_budget(20000)
    yield
    _h.set_ladder_budget(0)


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


def test_game_recorder_files(host, tmp_path):
    """<dir>/sgf batch files follow cc/data/filename_format.h:27-31 (gen%03d_b%03d_g%03d_%s.sgf,
    one game per line, matching .done; only games that started on an empty board:
    game_recorder.cc:101-108); <dir>/chunks holds the training chunks and their side files."""
    import os
    import re
    host.set_recorder(str(tmp_path), gen=7, worker_id="w3", flush_interval=4)
    try:
        st = host.selfplay_run(None, num_games=16, num_threads=2, seconds=1.5, default_n=4, default_k=2,
                               selected_n=4, selected_k=2, max_moves=12, warmup_batches=0, seed=11)
    finally:
        host.set_recorder("")
    files = sorted(os.listdir(tmp_path / "sgf"))
    sgfs = [f for f in files if f.endswith(".sgf")]
    assert sgfs and all(re.fullmatch(r"gen007_b\d{3}_g\d{3}_w3\.sgf", f) for f in sgfs)
    total = 0
    for f in sgfs:
        assert f[:-4] + ".done" in files
        lines = open(tmp_path / "sgf" / f).read().split("\n")
        assert lines[-1] == ""
        n_games = int(re.search(r"_g(\d{3})_", f).group(1))
        assert len(lines) - 1 == n_games
        for ln in lines[:-1]:
            # komi = round(7 + clamp(N(0,1), -3, 3)) +- 0.5 (self_play_thread.cc:204-206)
            m = re.match(r"\(;FF\[4\]GM\[1\]KM\[(-?\d+\.5)\]RE\[", ln)
            assert m and 3.5 <= float(m.group(1)) <= 10.5
            assert "PB[p3achygo]PW[p3achygo]" in ln and ln.endswith(")")
        total += n_games
    assert 0 < total <= st.games + 16
    chunks = sorted(os.listdir(tmp_path / "chunks"))
    zz = [f for f in chunks if f.endswith(".tfrecord.zz")]
    assert zz and all(re.fullmatch(r"gen007_b\d{3}_g\d{3}_n\d{5}_t\d+_w3\.tfrecord\.zz", f) for f in zz)
    for f in zz:
        stem = f[:-len(".tfrecord.zz")]
        assert {stem + ".done", stem + ".stats", stem + ".visit_count"} <= set(chunks)
    n_examples = sum(int(re.search(r"_n(\d{5})_", f).group(1)) for f in zz)
    assert n_examples == host.last_run_counters()[1] > 0


def test_init_states_forks_and_reuse_buffer(host):
    """Default policy (selfplay/main.cc:48-50,191-193): fork managers feed the GoExploit buffer
    and new games draw from it; with init-state sampling off nothing is added."""
    st = host.selfplay_run(None, num_games=64, num_threads=4, seconds=2.5, default_n=4, default_k=2,
                           selected_n=4, selected_k=2, max_moves=40, warmup_batches=0, seed=5)
    added, _ = host.last_run_counters()
    assert st.games > 64 and added > 0
    host.set_policy(init_state_sampling=False)
    try:
        host.selfplay_run(None, num_games=16, num_threads=2, seconds=0.5, default_n=4, default_k=2,
                          selected_n=4, selected_k=2, max_moves=12, warmup_batches=0, seed=5)
        assert host.last_run_counters()[0] == 0
    finally:
        host.set_policy()


def _cround(x):
    """std::round: halves away from zero"""
    import math
    return math.copysign(math.floor(abs(x) + 0.5), x)


def _fork(host, kind, n_moves, seed, p_win=0.5, score=0):
    import ctypes as C
    L = host.lib()
    L.p3host_test_fork.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    out = np.zeros(8, np.int32)
    komi = np.zeros(1, np.float32)
    last5 = np.zeros(5, np.int32)
    board = np.zeros(361, np.int8)
    L.p3host_test_fork(kind, n_moves, seed, p_win, score, out.ctypes.data, komi.ctypes.data, last5.ctypes.data,
                       board.ctypes.data)
    return out, float(komi[0]), last5, board


def test_fork_manager_late_and_random(host):
    """ForkManager kLate / kSampleUniform (fork_manager.h:196-385): exactly one fork at the
    sampled move number; the stored state is one or two plies past it, the colour to move and
    stone count match, the alternative moves end the last-move list, and komi is the
    score-neutral one: the scripted net says the side to move at the fork leads by 4.5, so
    komi moves by round(+-4.5) from 7.5 (ComputeAdjKomi, :520-533)."""
    seen_single = seen_double = False
    for seed in range(1, 40):
        for kind in (1, 4):
            out, komi, last5, board = _fork(host, kind, 260, seed, score=4)
            assert out[7] == kind and 10 <= out[6] <= 250
            assert out[0] == 1
            fork_mv = out[6]
            plies = out[1] - fork_mv
            assert plies in (1, 2)
            mover = 1 if fork_mv % 2 == 0 else -1          # colour that was to move at the fork
            assert out[2] == (mover if plies == 2 else -mover)
            assert out[3] in (1, 2) and (kind != 4 or out[3] == 2)   # kSampleUniform forces a full search
            assert out[4] <= fork_mv + plies and out[4] >= 1
            assert all(0 <= x <= 361 for x in last5[-plies:])
            if kind == 1:
                assert 5 + 1 <= out[5] <= 2 * 36 + 1        # best-of-n candidates (+ second ply) + komi eval
            else:
                assert out[5] == 1                           # only the komi evaluation
            # E[score] = +4.5 for the side to move at the evaluated position
            if plies == 1:      # P': opponent to move and 4.5 ahead => original mover 4.5 behind
                want = 7.5 + _cround(-4.5 if mover == 1 else 4.5)
                assert komi == want
                seen_single = True
            else:               # P'': komi adjusted with probability 1/2
                want = 7.5 + _cround(4.5 if mover == 1 else -4.5)
                assert komi in (7.5, want)
                seen_double = True
    assert seen_single and seen_double


def test_fork_manager_uniform_kind(host):
    """kUniform (fork_manager.h:183-211, 387-394): positions are sampled with probability 5 %
    per move and exactly one of them reaches the buffer at the end of the game, unchanged
    (first_move_behavior kSample, same colour to move, move number = number of stones here)."""
    added = 0
    for seed in range(1, 30):
        out, komi, last5, board = _fork(host, 6, 120, seed)
        assert out[0] in (0, 1) and out[5] == 0
        if out[0]:
            added += 1
            assert out[3] == 0 and out[4] == out[1]
            assert out[2] == (1 if out[1] % 2 == 0 else -1)
            # mcts_score 3.0 for the mover: komi either kept or shifted by round(+-3)
            assert komi in (7.5, 7.5 + 3, 7.5 - 3)
    assert added >= 20


def test_init_state_distribution(host):
    """GetInitState (self_play_thread.cc:202-252): 5 % handicap games (2-4 stones, White to
    move, komi (h-2)*14+20.5); otherwise komi = round(7 + clamp(N(0,1),-3,3)) +- 0.5."""
    import ctypes as C
    L = host.lib()
    L.p3host_test_init_states.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    hc = np.zeros(1, np.int32)
    hist = np.zeros(64, np.int32)
    hs = np.zeros(3, np.int32)
    n = 20000
    L.p3host_test_init_states(7, n, hc.ctypes.data, hist.ctypes.data, hs.ctypes.data)
    assert abs(hc[0] / n - 0.05) < 0.01
    assert hs.sum() == hc[0] and all(abs(x / hc[0] - 1 / 3) < 0.08 for x in hs)
    komis = {k / 2: c for k, c in enumerate(hist) if c}
    assert min(komis) >= 3.5 and max(komis) <= 10.5 and all(k % 1 == 0.5 for k in komis)
    assert hist.sum() == n - hc[0]
    # round(7 + z) = 7 with probability P(|z| < 0.5) = 0.383, split evenly between 6.5 and 7.5
    assert abs((komis[6.5] + komis[7.5] - (0.383 + 0.242) * hist.sum()) / hist.sum()) < 0.02


def test_sel_mult_calibration_file(host, tmp_path):
    """--sel_mult_calibration_file (selfplay/main.cc:64-67,71-118): per-generation thresholds replace the
    built-in ones of MoveSelManager; comments, malformed lines and unknown fields are skipped, a missing
    file means the defaults."""
    import ctypes as C
    L = host.lib()
    L.p3host_test_move_sel_file.argtypes = [C.c_char_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                            C.c_void_p, C.c_void_p]
    f = tmp_path / "calib.txt"
    f.write_text("# gen 42\nv_outcome_stddev.p50=0.09\nv_outcome_stddev_adj.p80=1.4\nv_outcome_stddev_adj.p99=4.0\n"
                 "pre_kld.p05=0.01\npre_kld.p70=0.25\nnn_mcts_diff.p70=0.2\nnn_mcts_diff.p99=0.8\n"
                 "expected_std.n0=0.3\nexpected_std.n5=0.25\nexpected_std.nx=1\nnodot=3\nbogus.p50=7\nno equals sign\n"
                 "pre_kld.p95=abc\n")

    def sel(path, n_pre, std, kld, diff, q):
        out, cnt = np.zeros(6, np.float32), np.zeros(5, np.int32)
        L.p3host_test_move_sel_file(path, n_pre, std, kld, diff, q, 1.0, out.ctypes.data, cnt.ctypes.data)
        return out, list(cnt)

    o, cnt = sel(str(f).encode(), 20, 0.1, 0.03, 0.5, 0.0)
    assert cnt == [1, 2, 2, 2, 2]
    assert o[4] == pytest.approx(1 - 0.7 * (0.06 - 0.03) / (0.06 - 0.01), rel=1e-5)      # KLD penalty with p05 = 0.01
    assert o[5] == pytest.approx(1 + 0.6 * (0.5 - 0.2) / (0.8 - 0.2), rel=1e-5)          # NN-MCTS bonus with p70 / p99 of the file
    d, cnt = sel(str(tmp_path / "missing.txt").encode(), 20, 0.1, 0.03, 0.5, 0.0)
    assert cnt == [0, 0, 0, 0, 0]
    assert d[4] == pytest.approx(1 - 0.7 * (0.06 - 0.03) / (0.06 - 0.0001), rel=1e-5)
    assert d[5] == pytest.approx(1 + 0.6 * (0.5 - 0.1463) / (0.65 - 0.1463), rel=1e-5)
    # and a self-play run takes the file
    try:
        host.set_calibration_file(str(f))
        host.set_groups(2)
        st = host.selfplay_run(None, 8, 2, 0.3, default_n=8, default_k=4, selected_n=8, selected_k=4, max_moves=12,
                               warmup_batches=1)
    finally:
        host.set_calibration_file("")
    assert st.positions > 0


def test_move_sel_manager_reference_cases(host, tmp_path):
    """All 14 cases of cc/selfplay/__tests__/move_sel_manager_test.cc, with its MakeCalibration() thresholds
    handed over as a calibration file."""
    import ctypes as C
    L = host.lib()
    L.p3host_test_move_sel_ex.argtypes = [C.c_uint, C.c_char_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_void_p]
    SB, SP, KB, KP, NM = 1, 2, 4, 8, 16
    base = ("v_outcome_stddev_adj.p01=0.02\nv_outcome_stddev_adj.p50=0.64\nv_outcome_stddev_adj.p80=1.52\n"
            "v_outcome_stddev_adj.p99=4.96\nnn_mcts_diff.p70=0.15\nnn_mcts_diff.p99=0.65\n" +
            "".join(f"expected_std.n{n}=0.16\n" for n in range(0, 201, 5)))
    files = {}
    for name, extra in (("base", ""), ("kldb", "pre_kld.p70=0.1\npre_kld.p95=0.5\n"), ("kldp", "pre_kld.p05=0.001\n")):
        files[name] = tmp_path / (name + ".txt")
        files[name].write_text(base + extra)
    K = ("modifier", "modifier_unscaled", "sel_bonus", "sel_penalty", "sel_std_bonus", "sel_std_penalty", "sel_kld_bonus",
         "sel_kld_penalty", "sel_nn_mcts_bonus", "sel_q_adjust", "std_adj", "std_adj_att")

    def compute(flags, cal, n_pre, std, kld, diff, q, scale):
        out = np.zeros(12, np.float32)
        L.p3host_test_move_sel_ex(flags, str(files[cal]).encode() if cal else b"", n_pre, std, kld, diff, q, scale,
                                  out.ctypes.data)
        return dict(zip(K, (float(x) for x in out)))
    ap = lambda x: pytest.approx(x, rel=1e-5)
    r = compute(SB | SP, None, 32, 0.15, 0, 0, 0, 1)                      # uncalibrated: std_adj = 0 => modifier 1
    assert r["std_adj"] == 0 and r["std_adj_att"] == 0 and r["sel_std_bonus"] == ap(1) and r["sel_std_penalty"] == ap(1)
    assert r["modifier"] == ap(1)
    r = compute(SB | SP, "base", 128, 0.16, 0, 0, 0, 1)                   # neutral position
    assert r["std_adj"] == ap(1) and r["sel_std_bonus"] == ap(1) and r["sel_std_penalty"] == ap(1) and r["modifier"] == ap(1)
    r = compute(SB | SP, "base", 128, 0.48, 0, 0, 0, 1)                   # high std_adj => bonus
    assert r["std_adj"] == ap(3) and r["std_adj_att"] == ap(3) and r["sel_std_bonus"] > 1 and r["sel_std_penalty"] == ap(1)
    assert r["modifier"] > 1 and r["sel_std_bonus"] == pytest.approx(1 + 0.5 * (3.0 - 1.52) / (4.96 - 1.52), rel=1e-4)
    r = compute(SB | SP, "base", 128, 0.016, 0, 0, 0, 1)                  # low std_adj => penalty
    assert r["std_adj"] < 0.64 and r["sel_std_penalty"] < 1 and r["sel_std_bonus"] == ap(1) and r["modifier"] < 1
    lo, hi = compute(SB | SP, "base", 5, 0.48, 0, 0, 0, 1), compute(SB | SP, "base", 128, 0.48, 0, 0, 0, 1)
    assert lo["std_adj_att"] < hi["std_adj_att"] and lo["modifier"] < hi["modifier"]       # low n_pre attenuates
    con, dec = compute(SB | SP, "base", 128, 0.48, 0, 0, 0, 1), compute(SB | SP, "base", 128, 0.48, 0, 0, 0.95, 1)
    assert dec["sel_q_adjust"] < con["sel_q_adjust"] and dec["sel_bonus"] < con["sel_bonus"]   # decisive position
    r = compute(SB, "base", 128, 0.016, 0, 0, 0, 1)                       # flag off: computed, not applied
    assert r["sel_std_penalty"] < 1 and r["sel_penalty"] == ap(1) and r["modifier"] == ap(1)
    r = compute(SB | SP, "base", 128, 0.48, 0, 0, 0, 0)                   # scale factor 0
    assert r["modifier"] == ap(1) and r["modifier_unscaled"] > 1
    both, std, kld = (compute(f, "kldb", 128, 0.48, 0.3, 0, 0, 1) for f in (SB | KB, SB, KB))   # bonus = max, not product
    assert both["sel_std_bonus"] > 1 and both["sel_kld_bonus"] > 1
    assert both["sel_bonus"] == pytest.approx(max(std["sel_bonus"], kld["sel_bonus"]), rel=1e-3)
    assert both["sel_bonus"] < std["sel_bonus"] * kld["sel_bonus"]
    both, std, kld = (compute(f, "kldp", 128, 0.016, 0.005, 0, 0, 1) for f in (SP | KP, SP, KP))   # penalty = min
    assert both["sel_std_penalty"] < 1 and both["sel_kld_penalty"] < 1
    assert both["sel_penalty"] == pytest.approx(min(std["sel_penalty"], kld["sel_penalty"]), rel=1e-3)
    assert both["sel_penalty"] > std["sel_penalty"] * kld["sel_penalty"]
    r = compute(NM, "base", 128, 0, 0, 0.5, 0, 1)                         # NN-MCTS bonus: 1 + 0.6 * 0.35 / 0.5
    assert r["sel_nn_mcts_bonus"] == pytest.approx(1.42, rel=1e-4) and r["sel_bonus"] > 1 and r["modifier"] > 1
    r = compute(NM, "base", 0, 0, 0, 0, 0, 1)                             # uninitialised root
    assert r["sel_nn_mcts_bonus"] == ap(1) and r["modifier"] == ap(1)
    r = compute(NM, "base", 128, 0, 0, 0.05, 0, 1)                        # below the lower bound
    assert r["sel_nn_mcts_bonus"] == ap(1) and r["modifier"] == ap(1)
    both, nm, kld = (compute(f, "kldb", 128, 0, 0.3, 0.5, 0, 1) for f in (NM | KB, NM, KB))
    assert both["sel_bonus"] == pytest.approx(max(nm["sel_bonus"], kld["sel_bonus"]), rel=1e-3)


def test_move_sel_manager_known_answers(host):
    """MoveSelManager::Compute (move_sel_manager.h:41-77) with the default calibration
    constants (:131-179) and self-play's flags kNnMctsBonus | kKldPenalty."""
    import ctypes as C
    L = host.lib()
    L.p3host_test_move_sel.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]

    def sel(n_pre, std, kld, diff, q, scale=1.0):
        out = np.zeros(6, np.float32)
        L.p3host_test_move_sel(n_pre, std, kld, diff, q, scale, out.ctypes.data)
        return out

    o = sel(0, 0.0, 0.0, 0.0, 0.0)                       # uninitialised root: neutral
    assert np.allclose(o[:3], 1.0) and o[3] == pytest.approx(1.0)
    o = sel(20, 0.1, 0.03, 0.0, 0.0)                     # low pre-KLD: penalty 1 - 0.7*(0.06-0.03)/(0.06-0.0001)
    pen = 1 - 0.7 * (0.06 - 0.03) / (0.06 - 0.0001)
    assert o[4] == pytest.approx(pen, rel=1e-5) and o[0] == pytest.approx(pen, rel=1e-5)
    o = sel(20, 0.1, 0.5, 0.3982, 0.0)                   # |NN-MCTS| halfway p70..p99: bonus 1.3
    assert o[5] == pytest.approx(1.3, rel=1e-3) and o[0] == pytest.approx(1.3, rel=1e-3)
    o = sel(20, 0.1, 0.5, 0.3982, 0.7)                   # q attenuation: (1 - (0.7-0.5)/0.4)^0.4
    qa = (1 - 0.5) ** 0.4
    assert o[3] == pytest.approx(qa, rel=1e-5) and o[0] == pytest.approx(1 + qa * 0.3, rel=1e-3)
    o = sel(20, 0.1, 0.5, 0.3982, 0.95)                  # decided position: signals off
    assert o[0] == pytest.approx(1.0)
    o = sel(20, 0.1, 0.5, 0.3982, 0.0, scale=0.5)        # sel_mult_scale_factor halves the effect
    assert o[0] == pytest.approx(1.15, rel=1e-3)
    o = sel(20, 0.1, 0.00005, 5.0, 0.0)                  # floor 0.3 and bonus cap: 1 + 0.6*(5-0.1463)/(0.65-0.1463) capped 2.5
    assert o[4] == pytest.approx(0.3) and o[1] == pytest.approx(2.5) and o[0] == pytest.approx(0.75)


def test_selfplay_host_over_the_cpu_engine_and_step_limit(built, weight_files):
    """The CPU baseline leg of bench.py: the same self-play host binds the CPU fp32 oracle through
    the same C ABI (oracle/libp3cpu_engine.so, loaded with dlopen exactly like libp3hip.so), and
    with a step limit the timed region is exactly `steps` engine batches (each group its share):
    `steps` batches' worth of leaf positions are loaded and evaluated in it."""
    import os
    from conftest import ROOT
    from p3achygo_amd import host_api
    lib = os.path.join(ROOT, "oracle", "libp3cpu_engine.so")
    assert os.path.exists(lib)
    os.environ["P3CPU_THREADS"] = "4"
    host_api.set_groups(2)
    host_api.set_step_limit(6)
    try:
        st = host_api.selfplay_run(weight_files("test_b3c128btl2"), num_games=16, num_threads=2, seconds=0.0,
                                   default_n=8, default_k=4, selected_n=8, selected_k=4, max_moves=40,
                                   warmup_batches=1, seed=3, engine_lib=lib)
    finally:
        host_api.set_step_limit(0)
    assert st.batches == 6                           # 3 batches for each of the 2 groups
    assert 0 < st.positions <= 6 * 8                 # at most one full batch of 8 games per step
    assert st.positions >= 6 * 8 - 16                # terminal / cached leaves leave a few slots empty
    assert st.seconds > 0


def test_round_anchored_window_counts_whole_rounds(built):
    """bench.py's timed region (round 4): `rounds` ROUNDS, opened and closed by completions of the SAME group (the one
    that finishes its warm-up last), so both ends sit at the same phase of the groups' cycle; every batch of any group
    completing inside is counted with the positions its host advance loaded.  The anchor contributes exactly `rounds`
    batches; the other groups, which nothing paces over the NullEvaluator, about as many each (on the GPU the
    forward passes queue behind one another: exactly one per group and round)."""
    from p3achygo_amd import host_api
    host_api.set_groups(4)
    host_api.set_step_rounds(6)
    host_api.set_step_limit(3)           # ignored while rounds are set
    try:
        st = host_api.selfplay_run(None, num_games=64, num_threads=4, seconds=0.0, default_n=16, default_k=4,
                                   selected_n=16, selected_k=4, warmup_batches=2, seed=5)
        host_api.set_groups(1)           # one group: a round is a batch
        solo = host_api.selfplay_run(None, num_games=16, num_threads=2, seconds=0.0, default_n=16, default_k=4,
                                     selected_n=16, selected_k=4, warmup_batches=2, seed=5)
    finally:
        host_api.set_step_rounds(0)
        host_api.set_step_limit(0)
        host_api.set_groups(2)
    assert st.rounds == 6 and 6 + 3 * 3 <= st.batches <= 6 + 3 * 12
    assert st.positions == 16 * st.batches and st.seconds > 0
    assert 0.5 * st.seconds < st.seconds_fit < 2.0 * st.seconds      # the anchor's 7 completion instants, fitted
    assert solo.rounds == solo.batches == 6 and solo.positions == 6 * 16


def test_advance_phase_plays_every_game_past_its_opening_before_the_timed_steps(built):
    """bench.py's untimed advance phase: a game samples its first moves from the raw policy,
    one evaluation per move, for up to 30 moves (self_play_thread.cc:44,363-366); with the advance limit set every
    group runs untimed batches until all its games are past that, so a SHORT timed window is Gumbel search
    (n = 16 here: close to 16 evaluations per move, not 1) and it holds exactly the requested number of
    batches, whichever groups they fall in (5 is not a multiple of the 3 groups)."""
    from p3achygo_amd import host_api
    host_api.set_groups(3)
    host_api.set_step_limit(5)
    try:
        host_api.set_advance_limit(0)
        cold = host_api.selfplay_run(None, num_games=96, num_threads=4, seconds=0.0, default_n=16, default_k=4,
                                     selected_n=16, selected_k=4, warmup_batches=1, seed=12)
        host_api.set_advance_limit(64)
        warm = host_api.selfplay_run(None, num_games=96, num_threads=4, seconds=0.0, default_n=16, default_k=4,
                                     selected_n=16, selected_k=4, warmup_batches=1, seed=12)
        host_api.set_groups(1)           # BASELINE configs[2] as written: one group, host and engine alternate
        solo = host_api.selfplay_run(None, num_games=32, num_threads=4, seconds=0.0, default_n=16, default_k=4,
                                     selected_n=16, selected_k=4, warmup_batches=1, seed=12)
    finally:
        host_api.set_step_limit(0)
        host_api.set_advance_limit(0)
        host_api.set_groups(2)
    assert cold.batches == warm.batches == solo.batches == 5
    assert cold.positions == warm.positions == 5 * 32 and solo.positions == 5 * 32
    assert cold.advance_batches == 0 and cold.games_past_opening < 96      # most games still in their openings
    assert cold.moves > 0.5 * cold.positions                               # ~1 evaluation per move there
    assert 0 < warm.advance_batches <= 3 * 64 and warm.games_past_opening == 96
    assert warm.moves * 8 < warm.positions                                 # search: many evaluations per move
    assert solo.games_past_opening == 32


def test_baseline_config_c1_plumbing_on_the_cpu_engine(built, weight_files):
    """BASELINE configs[0]: "v4-b8 random-init net, 1 self-play thread, n=8 k=4 Gumbel, TF-CPU inference
    (plumbing, no GPU)".  v4 = b8c128nbt with bias_cache_lambda 0.3 / alpha 0.8 (config/v4.json); the
    reference's TF-CPU engine does not build, its stand-in is the CPU fp32
    oracle behind the same C ABI.  One game thread, the first moves of a game (a full-size fp32 forward
    pass takes 0.2 s on one core): every move costs at most n = 8 leaf evaluations plus its root.
    Reproducibility of whole games is pinned on the tiny net (tests/test_bias_cache_cpu.py) and on
    the GPU engine (tests/test_engine_gpu.py::test_config_c1_game_with_bias_cache_on_hip_engine)."""
    import os
    from conftest import ROOT
    from p3achygo_amd import host_api
    lib = os.path.join(ROOT, "oracle", "libp3cpu_engine.so")
    os.environ["P3CPU_THREADS"] = "1"
    host_api.set_ladder_budget(0)                        # reference-exact features
    try:
        host_api.set_bias_cache(0.3, 0.8)
        w = weight_files("b8c128nbt")
        mv, b, wsc, ev = host_api.selfplay_one_game(w, 8, 4, 6, seed=21, engine_lib=lib)
    finally:
        host_api.set_bias_cache(0.0, 0.8)
        host_api.set_ladder_budget(20000)
    assert len(mv) == 6 and len(set(int(m) for m in mv)) == 6 and all(m != 0 for m in mv)
    assert 6 <= ev <= 6 * 9                              # root + at most n leaves per move


def test_gumbel_early_stopping(host):
    """GumbelSearchParams::early_stopping_enabled (gumbel.cc:323-351,396-466; off by default): every
    ceil(v / 4) sweeps of a halving round the candidates are re-ranked, and the round ends as soon
    as — every candidate having 10 visits — no upper confidence bound of the half to be dropped
    reaches the best lower bound of the half that stays.  The scripted scenario of the known-answer
    test separates its four candidates by a third of the value range, so with a large budget the
    search stops far short of n and still picks the same move."""
    L = host.lib()
    cv = (C.c_int * 4)()
    cq = (C.c_float * 4)()
    nn, mc, rn = C.c_int(), C.c_int(), C.c_int()
    rc = L.p3host_test_scripted_search(400, 4, cv, cq, C.byref(nn), C.byref(mc), C.byref(rn))
    assert rc >> 8 == 400 and mc.value == 3
    full = list(cv)
    L.p3host_test_scripted_early_stopping(1)
    try:
        rc = L.p3host_test_scripted_search(400, 4, cv, cq, C.byref(nn), C.byref(mc), C.byref(rn))
    finally:
        L.p3host_test_scripted_early_stopping(0)
    assert 40 <= rc >> 8 < 250 and mc.value == 3       # both rounds end at their first or second check
    assert all(v >= 10 for v in cv) and sum(cv) < sum(full)


def test_opening_book_start(host):
    """The book branch of GetInitState (self_play_thread.cc:216-233, cc/selfplay/book.h; its
    probability constant is 0, so it is taken only when the uniform draw is exactly 0): a random
    prefix of 0..4 moves of one of six openings, played from the empty board, colours alternating,
    the last-move window filled from the back, move_num left at 0."""
    L = host.lib()
    L.p3host_test_book_state.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p]
    book_points = {3 * 19 + 3, 15 * 19 + 15, 15 * 19 + 4, 4 * 19 + 15, 16 * 19 + 4, 15 * 19 + 16, 2 * 19 + 15, 16 * 19 + 15}
    lens = set()
    for seed in range(60):
        out, last5 = (C.c_int * 5)(), (C.c_int * 5)()
        L.p3host_test_book_state(seed, out, last5)
        stones, color, kind, move_num, nm = list(out)
        assert kind == 1 and move_num == 0 and 0 <= nm <= 4
        assert color == (1 if nm % 2 == 0 else -1)
        assert stones in (nm, nm - 1)                      # line 3 repeats a point: that stone is rejected
        moves = [m for m in last5 if m != -20]
        assert list(last5[:5 - nm]) == [-20] * (5 - nm) and all(m in book_points for m in moves)
        if nm:
            assert moves[0] == 3 * 19 + 3
        lens.add(nm)
    assert lens == {0, 1, 2, 3, 4}


def _inflight(host, seed, depth, sched, n=32, k=5, sn=32, sk=5, max_moves=60, games=3, cache=64, init=1, es=0):
    L = host.lib()
    f = L.p3host_test_game_inflight
    f.argtypes = [C.c_int] * 5 + [C.c_uint64] + [C.c_int] * 5 + [C.c_uint64, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 5)()
    assert f(n, k, sn, sk, max_moves, seed, depth, games, cache, init, es, sched, out) == 0
    return list(out)


@pytest.mark.parametrize("cfg", [
    dict(),                                                                  # bench.py's search: n = 32, k = 5
    dict(n=64, k=8, sn=128, sk=8, max_moves=80, games=2, cache=4, es=1),     # early stopping on, a four-entry cache
    dict(n=16, k=4, sn=48, sk=6, max_moves=150, games=3, cache=2, init=0),
    dict(n=8, k=2, sn=8, sk=2, max_moves=300, games=5),                      # two considered actions: every other playout waits
    dict(n=32, k=5, max_moves=250, games=2, cache=1, es=1),
])
def test_playouts_in_flight_play_the_same_games(host, cfg):
    """Round 4: up to four playouts of one Gumbel search wait for their evaluations at once
    (GumbelSearch::IssueNext, GameRunner::TryAdvance) so that ONE group of games can fill a second engine batch while
    its first is on the GPU.  A playout starts early only if no playout in flight shares its considered action and no
    read of the root children's statistics (a round closing, an early-stopping check) lies in between; the evaluation
    cache reserves a missed entry at request time.  Over a position-dependent evaluator (HashEvaluator) one game runner
    plays several consecutive games — restart states, forks, raw-policy openings, PUCT fast moves and tree reuse
    included — at depth 1 (the reference's order) and at depths 2 and 4 under two result-arrival schedules: the
    moves, the scores, the number of evaluations and of cache hits must be identical, and requests must really have
    been outstanding together."""
    for seed in (11, 12):
        base = _inflight(host, seed, 1, 1, **cfg)
        assert base[4] == 1
        for depth in (2, 4):
            for sched in (3, 9):
                got = _inflight(host, seed, depth, sched, **cfg)
                assert got[:4] == base[:4], (seed, depth, sched)
                assert got[4] == min(depth, cfg.get("k", 5))


def test_playouts_in_flight_are_off_where_playouts_share_state(host):
    """With a bias cache every playout reads and writes the cache's shared entries, so the search keeps one
    evaluation in flight whatever depth the scheduler allows (as for a PUCT root and in graph mode)."""
    host.set_bias_cache(0.3, 0.8)
    try:
        base = _inflight(host, 5, 1, 1, max_moves=40, games=1)
        got = _inflight(host, 5, 4, 3, max_moves=40, games=1)
    finally:
        host.set_bias_cache(0.0, 0.8)
    assert got[:4] == base[:4] and got[4] == 1


def test_eval_cache_reserves_a_missed_entry_and_fills_it_later(host):
    """EvalCache::Probe / InsertPending / Fill against Find / Insert: random lookups over a few keys, once one at a
    time and once with up to four results outstanding (filled out of order, after their entries may have been
    evicted): the same lookups hit, with the same results, and the caches end with the same contents — at every
    capacity from 'evicts on every miss' to 'never evicts', and with the cache off."""
    f = host.lib().p3host_test_eval_cache_pipeline
    f.argtypes = [C.c_int] * 4 + [C.c_uint64]
    f.restype = C.c_long
    for cap, nkeys, depth in ((1, 3, 2), (2, 5, 4), (4, 6, 3), (8, 12, 4), (64, 80, 4), (3, 3, 4), (0, 4, 2)):
        for seed in (1, 2, 3):
            assert f(cap, nkeys, 4000, depth, seed) == 0, (cap, nkeys, depth, seed)


def test_two_lanes_of_one_group_play_the_same_games(host):
    """The scheduler's two-lane form (host_api.set_lanes; BASELINE configs[2] as stated): ONE group of games fills
    two engine batches in turn — results of one lane are consumed and its next batch is filled while the other
    lane's batch is being evaluated — with rows left empty by games that have to wait handed to games that can start
    another playout.  Every game runner's first game is the same as with one lane, for every depth, and with three
    or more playouts in flight the batches stay (nearly) full where depth 1 leaves every other batch empty."""
    host.set_policy(init_state_sampling=False)
    host.set_groups(1)

    def run(lanes, depth, steps):
        host.set_lanes(lanes, depth)
        host.set_step_limit(steps)
        st = host.selfplay_run(None, 48, 4, 0.0, default_n=16, default_k=4, selected_n=16, selected_k=4, max_moves=24,
                               warmup_batches=1, seed=5, engine_lib="hash")
        return host.last_first_game_digests(), st.positions / st.batches / 48

    try:
        base, fill = run(1, 1, 560)
        assert fill == 1.0 and (base != 0).all()
        for depth, steps, min_fill in ((1, 1120, 0.49), (2, 800, 0.7), (4, 700, 0.85)):
            got, fill = run(2, depth, steps)
            both = (got != 0) & (base != 0)
            assert both.sum() >= 44 and (got[both] == base[both]).all(), depth
            assert fill >= min_fill, (depth, fill)
    finally:
        host.set_lanes(1, 1)
        host.set_step_limit(0)
        host.set_groups(2)
        host.set_policy()


def test_a_batch_does_not_wait_for_a_last_few_slow_games(host):
    """Two lanes: a host phase whose last few games are slow (an exact ladder read-out can take tens of milliseconds)
    lets its batch leave without them once another lane's run has come back; the stragglers finish on the pool, the
    group's next host phase waits for them first and loads what they asked for into ITS batch.  With every 61st
    (game, phase) pair asleep for 2 ms nearly every phase hands games over — and every game runner still plays the
    same first game as the one-lane, one-at-a-time schedule, at every depth."""
    host.set_policy(init_state_sampling=False)
    host.set_groups(1)

    def run(lanes, depth, steps, slow_us):
        host.set_lanes(lanes, depth)
        host.set_step_limit(steps)
        host.set_test_slow_games(slow_us)
        host.selfplay_run(None, 48, 4, 0.0, default_n=16, default_k=4, selected_n=16, selected_k=4, max_moves=24,
                          warmup_batches=1, seed=5, engine_lib="hash")
        return host.last_first_game_digests(), host.last_handed_over()

    try:
        base, (phases0, games0) = run(1, 1, 560, 0)
        assert (base != 0).all() and phases0 == 0 and games0 == 0          # one lane never hands over
        for depth, steps in ((4, 800), (2, 900), (1, 1200)):
            got, (phases, games) = run(2, depth, steps, 2000)
            both = (got != 0) & (base != 0)
            assert both.sum() >= 40 and (got[both] == base[both]).all(), depth
            assert phases >= steps // 4 and games >= phases, (depth, phases, games)
    finally:
        host.set_test_slow_games(0)
        host.set_lanes(1, 1)
        host.set_step_limit(0)
        host.set_groups(2)
        host.set_policy()
