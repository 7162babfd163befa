"""Rules engine / RNG / symmetry of the self-play host against the reference's own known
answers (cc/game/__tests__/board_test.cc, symmetry_test.cc transcribed to data by
tests/golden/make_rules_fixtures.py) and independently derived PCG known answers.
Integer work: every comparison is exact."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
with open(os.path.join(GOLD, "board_cases.json")) as f:
    BOARD_CASES = json.load(f)


@pytest.fixture(scope="module")
def host(built):
    from p3achygo_amd import host_api
    return host_api


def _dsl_board(host, text):
    """ParseBoardDSL (board_dsl.cc:91-117): all black stones in index order, then white."""
    cells = [c for c in text if not c.isspace()]
    assert len(cells) == 361
    b = host.Board()
    for col, chars in ((1, "xX"), (-1, "oO")):
        for idx, c in enumerate(cells):
            if c in chars:
                assert b.play(idx // 19, idx % 19, col), (idx, col)
    return b


@pytest.mark.parametrize("case", BOARD_CASES, ids=[f"{i}-{c['name']}" for i, c in enumerate(BOARD_CASES)])
def test_reference_board_case(host, case):
    b = host.Board()
    scores = None
    for op in case["ops"]:
        k = op[0]
        if k == "dsl":
            b = _dsl_board(host, op[1])
        elif k == "parse_seq":   # ParseBoard (board_test.cc:10-31): sequential, failures ignored
            b = host.Board()
            for idx, c in enumerate(op[1]):
                if c in "XO":
                    b.play(idx // 19, idx % 19, 1 if c == "X" else -1)
        elif k == "play":
            b.play(op[2], op[3], op[1])
        elif k == "play_expect":
            assert b.play(op[2], op[3], op[1]) == op[4], op
        elif k == "dry_expect":
            assert b.dry(op[2], op[3], op[1]) == op[4], op
        elif k == "pass":
            b.pass_(op[1])
        elif k == "at":
            assert b.position()[op[1], op[2]] == op[3], op
        elif k == "all_empty":
            assert not b.position().any()
        elif k == "raw":
            b.place_raw(op[2], op[3], op[1])
        elif k == "libs":
            assert b.group_liberties(*op[1]) == op[2], op
        elif k == "libs_eq":
            assert b.group_liberties(*op[1]) == b.group_liberties(*op[2]), op
        elif k == "same_group":
            assert b.group_id(*op[1]) == b.group_id(*op[2]) >= 0, op
        elif k == "calc_pa":
            b.calc_pass_alive(op[1])
        elif k == "pa_region":
            want = np.zeros((19, 19), bool)
            for i, j in op[2]:
                want[i, j] = True
            assert np.array_equal(b.pass_alive() == op[1], want), op
        elif k == "calc_all_pa":
            b.is_all_pass_alive()
        elif k == "all_pass_alive":
            assert b.is_all_pass_alive() == op[1]
        elif k == "libs_plane":
            assert b.liberties_plane(op[1])[op[2], op[3]] == op[4], op
        elif k == "libs_plane_nonempty":
            assert b.liberties_plane(op[1])[op[2], op[3]] != 0, op
        elif k == "ladder_plane":
            before = b.position().copy()
            assert b.laddered()[op[1], op[2]] == op[3], op
            assert np.array_equal(before, b.position())   # CHECK_EQ(board, board_copy)
        elif k == "get_scores":
            scores = b.scores()
        elif k == "score":
            assert scores[0 if op[1] == "black" else 1] == op[2], op
        elif k == "ownership":
            want = np.zeros((19, 19), np.int8)
            for i, j in op[1]:
                want[i, j] = 1
            for i, j in op[2]:
                want[i, j] = -1
            assert np.array_equal(scores[2], want), op
        else:
            raise AssertionError("unknown op " + k)


def test_symmetry_golden_grids(host):
    """symmetry_test.cc:10-104: eight 5x5 golden grids, inverse round trips, Loc images."""
    L = host.lib()
    with open(os.path.join(GOLD, "symmetry_cases.json")) as f:
        g = json.load(f)
    n = g["grid_len"]
    base = g["grids"][0]
    for sym in range(8):
        out = [None] * (n * n)
        for i in range(n * n):
            out[L.p3host_transform_index(sym, i, n)] = base[i]
        assert out == g["grids"][sym], sym
        for i in range(n * n):
            assert L.p3host_transform_inv(sym, L.p3host_transform_index(sym, i, n), n) == i
            assert L.p3host_transform_index(sym, L.p3host_transform_inv(sym, i, 19), 19) == i
    m = g["loc_grid_len"]
    li, lj = g["loc"]
    for sym in range(8):
        t = L.p3host_transform_index(sym, li * m + lj, m)
        assert [t // m, t % m] == g["loc_images"][sym], sym


def _pcg_ref(state, inc):
    """Textbook PCG-XSH-RR 64/32 step (constants of cc/core/rand.cc:7-14), in Python ints."""
    M = (1 << 64) - 1
    x = state
    rot = x >> 59
    state = (x * 6364136223846793005 + inc) & M
    x ^= x >> 18
    v = (x >> 27) & 0xFFFFFFFF
    return state, ((v >> rot) | (v << ((-rot) & 31))) & 0xFFFFFFFF


INC = [1442695040888963407, 6364136223846793007, 1865811235122147685, 7664345821815920749]


def test_prng_known_answers(host):
    """PRng streams vs an independent big-int PCG: next, next64, next128, RandRange, Uniform."""
    L = host.lib()
    import ctypes as C
    for seed in (0, 1, 42, 0xDEADBEEFCAFEF00D):
        h = L.p3host_prng_new(seed, seed + 1, seed + 2, seed + 3)
        st = [(seed + k + INC[k]) & ((1 << 64) - 1) for k in range(4)]

        def step(k):
            st[k], r = _pcg_ref(st[k], INC[k])
            return r
        for _ in range(50):
            assert L.p3host_prng_next(h) == step(0)
        for _ in range(20):
            a, bb = step(0), step(1)
            assert L.p3host_prng_next64(h) == (a << 32) | bb
        hi, lo = C.c_uint64(), C.c_uint64()
        for _ in range(10):
            r = [step(k) for k in range(4)]
            L.p3host_prng_next128(h, C.byref(hi), C.byref(lo))
            assert hi.value == (r[0] << 32) | r[1] and lo.value == (r[2] << 32) | r[3]
        for lo_, hi_ in ((0, 8), (0, 362), (5, 6), (-3, 100)):   # RandRange, rand.cc:100-121
            width = hi_ - lo_
            mask = (1 << width.bit_length()) - 1
            r = step(0)
            while (r & mask) >= width:
                r = step(0)
            got = L.p3host_rand_range(h, lo_, hi_)
            assert got == (r & mask) + lo_ and lo_ <= got < hi_
        assert L.p3host_rand_range(h, 7, 7) == 7
        L.p3host_prng_free(h)
    p = L.p3host_prob_new(123)
    s = (123 + INC[0]) & ((1 << 64) - 1)
    for _ in range(100):   # Probability::Uniform (probability.cc:23-36): 1.man - 1, man = rand >> 9
        s, r = _pcg_ref(s, INC[0])
        want = np.frombuffer(np.uint32((127 << 23) | (r >> 9)).tobytes(), np.float32)[0] - np.float32(1)
        assert L.p3host_prob_uniform(p) == want
    s, r = _pcg_ref(s, INC[0])
    u = np.frombuffer(np.uint32((127 << 23) | (r >> 9)).tobytes(), np.float32)[0] - np.float32(1)
    assert abs(L.p3host_prob_gumbel(p) - (-np.log(-np.log(np.float64(u))))) < 1e-5
    L.p3host_prob_free(p)


def test_probability_moments_and_prng_determinism(host):
    """cc/core/__tests__/probability_test.cc (uniform / Gaussian / Gumbel sample moments over 10^6 draws from
    seed 42, the file's tolerances) and rand_test.cc (same seeds give the same streams through every
    constructor arity; RandRange stays in range)."""
    import ctypes as C
    L = host.lib()
    L.p3host_prob_moments.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p]
    out = np.zeros(3, np.float32)
    L.p3host_prob_moments(42, 0, 1000000, out.ctypes.data)
    assert out[2] == 0 and out[0] == pytest.approx(0.5, rel=0.002) and out[1] == pytest.approx(1 / 12, rel=0.002)
    L.p3host_prob_moments(42, 1, 1000000, out.ctypes.data)
    assert abs(out[0]) <= 0.005 and out[1] == pytest.approx(1.0, rel=0.005)
    L.p3host_prob_moments(42, 2, 1000000, out.ctypes.data)
    assert out[0] == pytest.approx(0.5772156649, rel=0.007) and out[1] == pytest.approx(np.pi ** 2 / 6, rel=0.01)
    L.p3host_prng_new.restype = C.c_void_p
    L.p3host_prng_new.argtypes = [C.c_uint64] * 4
    for seeds in ((7, 0, 0, 0), (11, 12, 0, 0), (17, 13, 4, 5), (0, 1, 2, 3)):
        a, b = L.p3host_prng_new(*seeds), L.p3host_prng_new(*seeds)
        hi0, lo0, hi1, lo1 = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        for _ in range(4):
            L.p3host_prng_next128(C.c_void_p(a), C.byref(hi0), C.byref(lo0))
            L.p3host_prng_next128(C.c_void_p(b), C.byref(hi1), C.byref(lo1))
            assert (hi0.value, lo0.value) == (hi1.value, lo1.value)
        L.p3host_prng_free(C.c_void_p(a)); L.p3host_prng_free(C.c_void_p(b))
    h = L.p3host_prng_new(0, 1, 2, 3)
    for lo, hi in ((0, 5), (-80, -40), (10, 50), (-40, 30)):
        assert lo <= L.p3host_rand_range(C.c_void_p(h), lo, hi) < hi
    L.p3host_prng_free(C.c_void_p(h))


def test_superko_and_ko(host):
    """Positional superko (board.cc:617-621): the ko recapture recreates an earlier
    whole-board position -> kRepeatedPosition; after an exchange elsewhere the same point is
    legal again.  Also the status codes of MoveStatus (board.h:50-58)."""
    b = host.Board()
    for (i, j, c) in [(2, 1, 1), (2, 2, -1), (3, 2, 1), (3, 3, -1), (2, 3, 1), (2, 4, -1), (1, 2, 1), (1, 3, -1)]:
        assert b.play(i, j, c)
    assert b.position()[2, 2] == 0            # white stone captured by black's (1,2)
    assert b.play(2, 2, -1)                   # white retakes the ko, capturing (2,3)
    assert b.position()[2, 3] == 0
    assert b.dry_status(2, 3, 1) == 6         # kRepeatedPosition
    assert b.play(10, 10, 1) and b.play(10, 11, -1)
    assert b.dry_status(2, 3, 1) == 0         # position differs now: legal
    assert b.dry_status(2, 2, 1) == 3         # kLocNotEmpty
    assert b.dry_status(19, 19, 1) == 2       # kOutOfBounds
    assert b.dry_status(5, 5, 0) == 1         # kUnknownColor
    c = host.Board()
    for (i, j) in [(0, 1), (1, 0)]:
        c.play(i, j, 1)
    assert c.dry_status(0, 0, -1) == 5        # kSelfCapture


def test_pass_alive_update_points(host):
    """pass_alive_ is only refreshed by the 3rd+ pass, GetScores and IsAllPassAlive
    (board.cc:562-572,627-629,502-507) and prohibits moves from then on (board.cc:589-591)."""
    b = host.Board()
    for (i, j) in [(0, 1), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3)]:
        assert b.play(i, j, 1)
    assert b.dry(0, 0, -1) is False           # suicide for white anyway
    assert b.dry(0, 0, 1) and b.dry(0, 2, 1)  # not yet marked
    b.pass_(-1); b.pass_(1)                   # two passes: game over, no Benson yet
    assert b.is_game_over() and not b.pass_alive().any()
    b.play(10, 10, -1)
    b.pass_(1)                                # third pass overall -> Benson runs
    assert (b.pass_alive()[0:2, 0:4] == 1).all()
    assert b.dry_status(0, 0, 1) == 4         # kPassAliveRegion


def test_ladder_readout_exact_mode_equals_naive_reader_and_budget_is_a_switch(built):
    """Planes 13/14 (Board::GetLadderedStones, cc/game/board.cc:692-899).  Default mode = the
    reference's (depth bound only): over 20,000 random-playout positions it agrees, bit for bit,
    with an independent naive read-out written against the public board API (flood fills, boards
    copied by value).  The node budget is an opt-in throughput mode of the self-play host
    (host_api.set_ladder_budget): off by default, and with 20,000 nodes it never fires on this
    position distribution (it does on search-tree positions: its hits are counted in
    ladder_stats and reported by bench.py)."""
    import ctypes as C
    from p3achygo_amd import host_api
    L = host_api.lib()
    L.p3host_ladder_budget.restype = C.c_long
    assert L.p3host_ladder_budget() == 0                      # reference-exact unless asked otherwise
    L.p3host_test_ladder_modes.argtypes = [C.c_int, C.c_uint64, C.c_long, C.c_int, C.c_int, C.c_long, C.c_void_p]
    out = (C.c_long * 7)()
    L.p3host_test_ladder_modes(20000, 11, 20000, 10, 1, 0, out)
    naive_diff, budget_diff, budget_hits, max_nodes, with_ladder, readouts, skipped = list(out)
    assert naive_diff == 0 and skipped == 0
    assert budget_diff == 0 and budget_hits == 0 and max_nodes < 20000
    assert with_ladder > 5000 and readouts > 100000             # the sample is not vacuous
    # a budget small enough to fire changes planes (so the switch is live) and is counted
    L.p3host_test_ladder_modes(3000, 12, 40, 60, 1, 0, out)
    assert out[0] == 0 and out[2] > 0 and out[1] > 0
    assert L.p3host_ladder_budget() == 0                      # restored
    before = host_api.ladder_stats()
    host_api.set_ladder_budget(25)
    L.p3host_ladder_budget.restype = C.c_long
    assert L.p3host_ladder_budget() == 25
    host_api.set_ladder_budget(0)
    assert host_api.ladder_stats()[0] == before[0]
