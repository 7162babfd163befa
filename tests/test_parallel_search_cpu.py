"""Batch parallel search and evaluation matches (p3achygo_amd/host/parallel_search.h,
eval_match.h).  The search cases are the reference's cc/mcts/__tests__/search_test.cc:130-222
(NullEngine, pre-evaluated root, same budgets and invariants) plus tree-consistency checks."""
import ctypes as C

import numpy as np
import pytest

from p3achygo_amd import host_api


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
def L(built):
    lib = host_api.lib()
    lib.p3host_t_quantile.restype = C.c_float
    lib.p3host_test_batch_search.argtypes = [C.c_int, C.c_int, C.c_void_p]
    return lib


def run(L, batch, budget, mode=1, q_fn=0, n_fn=0, collision=0, detector=0):
    out = np.zeros(10, np.int32)
    L.p3host_test_batch_search_ex.argtypes = [C.c_int] * 7 + [C.c_void_p]
    assert L.p3host_test_batch_search_ex(batch, budget, mode, q_fn, n_fn, collision, detector, out.ctypes.data) == 0
    return out


def test_t_quantiles_match_scipy(L):
    """tree.cc:16-33 builds the table with boost::math::students_t: two-sided alpha = 0.05."""
    from scipy import stats
    for dof in (1, 2, 3, 5, 10, 30, 100, 999, 1000):
        assert L.p3host_t_quantile(dof) == pytest.approx(stats.t.ppf(1 - 0.025, dof), rel=2e-6)
    assert L.p3host_t_quantile(0) == L.p3host_t_quantile(1)             # CachedQuantile clamps
    assert L.p3host_t_quantile(5000) == L.p3host_t_quantile(1000)
    assert L.p3host_t_quantile(1) == pytest.approx(12.7062, rel=1e-5)   # textbook values
    assert L.p3host_t_quantile(10) == pytest.approx(2.22814, rel=1e-5)


@pytest.mark.parametrize("batch,budget", [(4, 32), (1, 16), (1, 24), (2, 24), (4, 24), (16, 10000)])
def test_search_terminates_and_respects_budget(L, batch, budget):
    """search_test.cc:130-222: visits in [budget, budget + threads), a move is returned, the
    root was visited."""
    o = run(L, batch, budget)
    visits, aborted, collisions, rounds, root_n, move = o[:6]
    assert budget <= visits < budget + batch
    assert move >= 0 and root_n > 1
    assert o[7] == 0 and o[8] == 0                   # n = 1 + sum(child visits) everywhere, nothing in flight
    assert root_n == 1 + o[6] == 1 + visits          # every completed descent passes the root
    assert rounds * batch == visits + aborted and collisions == aborted
    assert o[9] <= visits                            # terminal leaves need no evaluation


def test_search_is_deterministic_and_forks_cover_runner_ups(L):
    """With a uniform prior the first round of batch B visits B different root children (the
    later descents fork the first path at its smallest PUCT gap), and the run is reproducible."""
    a, b = run(L, 8, 64), run(L, 8, 64)
    assert np.array_equal(a, b)
    one = run(L, 1, 8)
    assert one[0] == 8 and one[1] == 0               # a single descent per round never collides


def test_eval_match_null_engines(built):
    """Plumbing of the match runner (eval.cc:103-518) without a GPU: every game finishes,
    colours alternate (cur is Black in even games), results add up, both engines were used."""
    st = host_api.eval_match(None, None, num_games=6, visits_per_move=16, leaves_per_round=4, max_moves=30,
                             num_threads=2, seed=3)
    assert st.games == 6 and st.cur_wins + st.cand_wins + st.draws == 6
    assert st.moves == 6 * 30 and st.resignations == 0          # uniform nets never resign or pass out early
    assert st.visits >= 16 * st.moves and st.positions > 0 and st.batches > 0
    assert st.positions <= st.visits + st.moves


def test_eval_match_legacy_single_thread_path(built):
    """num_threads_per_game = 1 takes GumbelEvaluator::SearchRootPuct with LCB move choice
    (eval.cc:99-101,262-281): one evaluation per step, n visits per move."""
    st = host_api.eval_match(None, None, num_games=4, visits_per_move=12, leaves_per_round=1, max_moves=16,
                             num_threads=2, seed=9)
    assert st.games == 4 and st.cur_wins + st.cand_wins + st.draws == 4 and st.moves == 4 * 16
    assert st.visits == 12 * st.moves and st.collisions == 0
    assert st.positions <= st.visits + st.moves


@pytest.mark.parametrize("q_fn,n_fn,collision,detector", [
    (2, 1, 0, 0),    # defaults of player_config.h: virtual_loss_soft, virtual_visit, abort, noop
    (1, 1, 0, 0),    # hard virtual loss
    (0, 1, 2, 0),    # search_test.cc MakeParams: identity Q, virtual visits, smart retry
    (0, 0, 1, 0),    # plain retry
    (2, 1, 2, 1), (2, 1, 0, 2), (2, 1, 2, 3),   # n-in-flight / level-saturation / product detectors
])
@pytest.mark.parametrize("batch,budget", [(4, 32), (1, 16), (2, 24), (16, 2000)])
def test_concurrent_round_mode(L, batch, budget, q_fn, n_fn, collision, detector):
    """SearchTask's round structure (search.cc:336-458) with the virtual-loss Q / N functions,
    collision policies and detectors (search.h:28-39,247-485, search_policy.h:400-459): same
    invariants as the reference's tests (search_test.cc:130-222), plus a consistent tree."""
    if detector in (1, 3) and batch < 4:
        pytest.skip("threshold log2(batch) = 1: the n-in-flight detector fires on every descent")
    o = run(L, batch, budget, 0, q_fn, n_fn, collision, detector)
    visits, aborted, collisions, rounds, root_n, move = o[:6]
    assert budget <= visits < budget + batch
    assert move >= 0 and root_n > 1
    assert o[7] == 0 and o[8] == 0
    assert root_n == 1 + o[6] == 1 + visits
    assert rounds * batch == visits + aborted and aborted <= collisions
    if batch == 1:
        assert collisions == 0


def test_virtual_loss_spreads_a_round(L):
    """Identity Q / N in a concurrent round sends every worker down the same path (all but the
    first collide at the pending leaf and abort); virtual visits + virtual loss make the
    workers of one round take different root moves, so far fewer descents are wasted."""
    plain = run(L, 8, 64, 0, 0, 0, 0, 0)
    vl = run(L, 8, 64, 0, 2, 1, 0, 0)
    assert plain[1] > 5 * max(vl[1], 1) and vl[3] < plain[3]


def test_fruitless_rounds_end_the_search(L):
    """Two workers with the n-in-flight detector: threshold max(1, log2(2)) = 1 fires on every
    entry into an existing child, so once the root's legal moves are expanded no descent can
    complete; the search gives up instead of spinning (the reference would not terminate)."""
    o = run(L, 2, 5000, 0, 2, 1, 0, 1)
    assert 0 < o[0] < 5000 and o[5] >= 0 and o[8] == 0


# ---- round 2: Elo, player configs, NN cache, SGF output, BuUct / graph search, threaded search ------

def test_relative_elo_and_match_summary(L):
    """core::RelativeElo (cc/core/elo.h) and the summary eval/main.cc:459-471 prints."""
    out = np.zeros(4, np.float32)
    L.p3host_match_summary.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.p3host_match_summary(60, 100, out.ctypes.data)
    wr = 0.6
    c95 = 1.96 * np.sqrt(wr * (1 - wr) / 100)
    assert out[0] == pytest.approx(wr) and out[1] == pytest.approx(c95, rel=1e-6)
    assert out[2] == pytest.approx(400 * np.log10(wr / (1 - wr)), rel=1e-6)          # +70.4 Elo
    assert out[3] == pytest.approx(400 * np.log10((.5 + c95) / (.5 - c95)), rel=1e-5)
    L.p3host_match_summary(50, 100, out.ctypes.data)
    assert out[2] == 0.0


def test_player_config_file(L, tmp_path):
    """ParsePlayerConfigFile (player_config.h:133-244): key: value lines, comments, every PlayerSearchConfig
    field by name, unknown keys ignored, enum strings falling back as MakeSearchParams does; and the
    parameter sets derived from it (MakeSearchParams / the SearchRootPuct call of eval.cc:241-258)."""
    L.p3host_parse_player_config.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p]
    p = tmp_path / "cand.cfg"
    p.write_text("# candidate\nn: 200\nnum_threads_per_game: 16\nc_puct: 1.25\n\nq_fn: virtual_loss\nn_fn: identity\n"
                 "collision_policy: smart_retry\ncollision_detector: product\nsearch_mode: batch\n"
                 "descent_policy: bu_uct\nmax_o_ratio: 0.7\nuse_mcgs: true\nuse_bias_cache: 1\ntime_ms: 250\n"
                 "k: 4\nnoise_scaling: 0.5\nearly_stopping_for_gumbel: true\nuse_puct: false\nuse_puct_v: true\n"
                 "c_puct_v_2: 2.5\ntau: 0.75\nuse_lcb: false\nscore_weight: 0.25\nscore_utility_mode: integral\n"
                 "enable_m3_bonus: true\nvar_scale_prior_visits: 7\nm3_prior_visits: 11\np_opt_weight: 0.3\n"
                 "enable_pondering: true\ntime_control_flags: all\nvl_delta: -2.0\nmax_collision_retries: 9\n"
                 "puct_root_policy: visit_count_sample\nname: ignored\nno_such_key: 1\nline without a colon\n")
    out = np.zeros(40, np.float32)
    err = C.create_string_buffer(256)
    assert L.p3host_parse_player_config(str(p).encode(), out.ctypes.data, err) == 0, err.value
    assert list(out[:2]) == [200, 16] and out[2] == pytest.approx(1.25)
    assert list(out[6:12]) == [1, 0, 2, 3, 1, 1] and out[12] == pytest.approx(0.7)
    assert list(out[13:16]) == [1, 1, 250]
    assert list(out[16:21]) == [4, 0.5, 1, 0, 1] and out[21] == pytest.approx(2.5) and out[22] == pytest.approx(0.75)
    assert out[23] == 0 and out[24] == pytest.approx(0.25) and out[25] == 1          # use_lcb, score weight, integral
    assert list(out[26:29]) == [1, 7, 11] and out[29] == pytest.approx(0.3)
    assert out[30] == 1 and out[31] == np.float32(2 ** 32 - 1) and out[32] == -2.0 and out[33] == 9
    assert out[34] == 2                      # parallel search: puct_root_policy wins (visit_count_sample)
    assert out[35] == 0                      # SearchRootPuct: use_lcb false -> visit count
    assert out[36] == pytest.approx(0.45) and out[37] == 1.0   # ... and the defaults eval.cc does not override there
    assert out[38] == 1
    # defaults (player_config.h:20-108; this repository's drivers set num_threads_per_game themselves), enum
    # fall-backs (player_config.cc:56-95) and the root policy derived from use_lcb when no policy is named
    p.write_text("q_fn: bogus\nn_fn: bogus\ncollision_policy: bogus\ncollision_detector: bogus\nsearch_mode: bogus\n"
                 "descent_policy: bogus\ntime_ms: auto\n")
    assert L.p3host_parse_player_config(str(p).encode(), out.ctypes.data, err) == 0
    assert list(out[6:12]) == [1, 1, 0, 0, 0, 0] and out[15] == -1
    assert out[1] == 8 and out[38] == 1      # this repository's default worker count: the parallel search
    assert list(out[16:21]) == [8, 1.0, 0, 1, 0] and out[23] == 1 and out[24] == 0.5 and out[25] == 0
    assert list(out[26:30]) == [0, 0, 20, 0.0] and out[34] == 1 and out[35] == 1 and out[39] == pytest.approx(0.8)
    p.write_text("n: ten\n")
    assert L.p3host_parse_player_config(str(p).encode(), out.ctypes.data, err) == 1


def _puct_scores_restated(n, v, v_var, mp, op, children, pp, fns, is_root):
    """search_policy.h:159-316 in numpy float32 / float64 as the C++ types dictate; children = list of
    (action, visits, v, v_var, v_m3, in_flight)."""
    f = np.float32
    c_puct, cvs, c_v2, use_v, var_scale, var_prior, m3_on, m3_prior, p_opt, root_fpu = pp
    qk, nk, vl = fns
    if p_opt == 0:
        prob = mp.astype(f)
    elif p_opt == 1:
        prob = op.astype(f)
    else:
        prob = (mp + f(p_opt) * (op - mp)).astype(f)
    A = len(mp)
    cv, fl = np.zeros(A, int), np.zeros(A, int)
    qs, qvars, qm3 = np.zeros(A, f), np.zeros(A, f), np.zeros(A, np.float64)
    qstd_w, qm3_w = f(0), 0.0
    for a, vis, cvv, cvar, cm3, inf in children:
        cv[a], fl[a] = vis, inf
        if vis > 0:
            qs[a] = -f(cvv)
        if vis >= 3:
            qvars[a], qm3[a] = f(cvar), -cm3
            qstd_w = f(qstd_w + np.sqrt(f(cvar)) * f(vis))
            qm3_w += np.cbrt(qm3[a]) * vis
    qstd_mean, qm3_mean = f(qstd_w / f(n)), qm3_w / n
    p_expl = f(0)
    for a in range(A):
        if cv[a] + fl[a] > 0:
            p_expl = f(p_expl + prob[a])
    v_fpu = f(f(v) - f(root_fpu if is_root else 0.2) * np.sqrt(p_expl))
    scale = lambda c: f(f(c) + f(cvs) * np.log(f(f(n + f(500)) / f(500))))
    cp, cp2 = scale(c_puct), scale(c_v2)
    N = lambda a: f(cv[a] + fl[a]) if nk == 1 else f(cv[a])
    def Q(q, a):
        if qk == 1:
            return f(q + f(fl[a]) * f(vl))
        if qk == 2:
            return q if fl[a] == 0 else f(f(q * f(cv[a]) + f(fl[a]) * f(vl)) / f(cv[a] + fl[a]))
        return q
    total_n = f(1)
    for a in range(A):
        total_n = f(total_n + N(a))
    out = np.zeros(A, f)
    for a in range(A):
        sc = f(1)
        if var_scale and cv[a] >= 3 and qstd_mean != 0:
            sc = f(f(f(var_prior) + f(cv[a]) * f(np.sqrt(qvars[a]) / qstd_mean)) / f(f(var_prior) + f(cv[a])))
        cn = N(a)
        q = Q(qs[a] if cv[a] > 0 else v_fpu, a)
        m3b = 0.0
        if m3_on and cv[a] >= 3:
            m3b = (float(f(m3_prior)) + (np.cbrt(qm3[a]) - qm3_mean)) / float(f(m3_prior) + f(cv[a]))
        if use_v:
            var = (f(1) if n < 3 else f(v_var)) if cv[a] < 3 else qvars[a]
            sd = np.sqrt(f(var))
            vs = f(f(prob[a] * sd) * f(np.sqrt(total_n) / f(1 + cn)))
            ns = f(f(prob[a] * np.log(total_n)) / f(1 + cn))
            ex = f(f(cp * vs) + f(cp2 * ns))
        else:
            ex = f(f(f(cp * sc) * prob[a]) * f(np.sqrt(total_n) / f(1 + cn)))
        out[a] = f(float(f(ex + q)) + m3b)
    return out


@pytest.mark.parametrize("variant", ["puct", "puct_v", "m3_var_popt", "virtual"])
def test_puct_scorer_known_answers(L, variant):
    """PuctScorer::ComputeScores with every term of search_policy.h:159-316 — optimistic-policy blend, FPU,
    visit-scaled c_puct, variance scaling, PUCT-V, the third-moment bonus, virtual-loss Q / N — against an
    independent numpy restatement on a hand-built node."""
    rng = np.random.default_rng(5)
    A = 362
    mp = rng.dirichlet(np.full(A, 0.3)).astype(np.float32)
    op = rng.dirichlet(np.full(A, 0.3)).astype(np.float32)
    acts = [3, 40, 41, 100, 200, 361]
    children = [(3, 12, 0.21, 0.05, 0.004, 0), (40, 3, -0.4, 0.2, -0.02, 1), (41, 1, 0.6, 0.0, 0.0, 0),
                (100, 7, -0.1, 0.09, 0.011, 2), (200, 0, 0.0, 0.0, 0.0, 1), (361, 2, 0.05, 0.01, 0.0, 0)]
    n, v, v_var = 26, 0.07, 0.12
    pp = {"puct": [1.0, 0.45, 3.0, 0, 0, 0, 0, 20, 0.0, 0.2],
          "puct_v": [1.1, 0.3, 2.5, 1, 0, 0, 0, 20, 1.0, 0.05],
          "m3_var_popt": [0.9, 0.45, 3.0, 0, 1, 4, 1, 11, 0.35, 0.1],
          "virtual": [1.0, 0.45, 3.0, 0, 1, 0, 0, 20, 0.0, 0.2]}[variant]
    fns = [2, 1, -1.5] if variant == "virtual" else ([1, 0, -0.5] if variant == "puct_v" else [0, 0, -1.5])
    L.p3host_test_puct_scores.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + \
        [C.c_int, C.c_void_p]
    ia = lambda k: np.array([c[k] for c in children], np.int32)
    fa = lambda k: np.array([c[k] for c in children], np.float32)
    m3 = np.array([c[4] for c in children], np.float64)
    ppv, fv = np.array(pp, np.float32), np.array(fns, np.float32)
    for is_root in (0, 1):
        got = np.zeros(A, np.float32)
        a_, vis_, v_, var_, fl_ = ia(0), ia(1), fa(2), fa(3), ia(5)
        L.p3host_test_puct_scores(n, v, v_var, mp.ctypes.data, op.ctypes.data, len(children), a_.ctypes.data, vis_.ctypes.data,
                                  v_.ctypes.data, var_.ctypes.data, m3.ctypes.data, fl_.ctypes.data, ppv.ctypes.data,
                                  fv.ctypes.data, is_root, got.ctypes.data)
        want = _puct_scores_restated(n, v, v_var, mp, op, children, pp, fns, bool(is_root))
        assert np.abs(got - want).max() < 2e-6, (variant, is_root, np.abs(got - want).max())
        assert int(np.argmax(got)) == int(np.argmax(want))


def test_leaf_evaluator_score_sign_convention(L):
    """cc/mcts/__tests__/leaf_evaluator_test.cc (compiled out upstream, its four cases still state the
    convention): the root's score estimate is seen from the leaf's side to move — same colour as the root
    keeps its sign, the other colour flips it — and a neutral evaluation's utility is the score transform
    of (0 - that)."""
    L.p3host_test_evaluate_leaf.restype = C.c_float
    L.p3host_test_evaluate_leaf.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_float]
    BLACK, WHITE = 1, -1
    st = lambda root: 0.5 * (2 / np.pi) * np.arctan((0 - root) / 19)
    for c, rc, sign in ((BLACK, BLACK, 1), (BLACK, WHITE, -1), (WHITE, WHITE, 1), (WHITE, BLACK, -1)):
        assert L.p3host_test_evaluate_leaf(c, rc, 15.0, 0, 0.5) == pytest.approx(st(sign * 15.0), abs=1e-6)


def _score_table_entry(score_idx, stddev):
    f = np.float32
    score_mean = f(score_idx - 400 + 0.5)
    mass, acc, z = f(0), f(0), f(-5.0)
    while z <= f(5.0):
        pdf = f(np.exp(-0.5 * float(z) * float(z)))
        st = f((2.0 / np.pi) * float(np.arctan(f(f(score_mean + f(z * f(stddev))) / f(19)))))
        mass, acc = f(mass + pdf), f(acc + f(st * pdf))
        z = f(z + f(0.1))
    return f(acc / mass)


def test_score_utility_modes(L):
    """LeafEvaluator::ScoreUtility (leaf_evaluator.cc:12-132): the direct arctan transform and its
    Gaussian-smoothed `integral` form (table over score mean x stddev, bilinear interpolation, 0.75 of the
    root's score estimate as the reference point), against a numpy restatement of the same loops."""
    L.p3host_test_score_utility.restype = C.c_float
    L.p3host_test_score_utility.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    f = np.float32
    for w, s, sd, root in ((0.5, 3.2, 0.0, 1.0), (0.5, -12.75, 4.6, 2.5), (0.25, 40.1, 17.3, -6.0), (1.0, 0.49, 0.2, 0.0),
                           (0.5, 450.0, 500.0, 0.0), (0.5, -450.0, 1.5, 0.0)):
        direct = f(f(w) * f(2.0 / np.pi) * np.arctan(f(f(f(s) - f(root)) / f(19))))
        assert L.p3host_test_score_utility(0, w, s, sd, root) == pytest.approx(float(direct), abs=1e-6)
        mean = f(f(s) - f(0.75) * f(root))
        sf, df = int(np.floor(f(mean - f(0.5)))), int(np.floor(f(sd)))
        xi, yi = int(np.clip(sf + 400, 0, 798)), int(np.clip(df, 0, 398))
        md, sdl = f(f(mean - f(0.5)) - f(sf)), f(f(sd) - f(df))
        a00, a01, a10, a11 = (_score_table_entry(xi, yi), _score_table_entry(xi, yi + 1), _score_table_entry(xi + 1, yi),
                              _score_table_entry(xi + 1, yi + 1))
        b0, b1 = f(a00 + sdl * f(a01 - a00)), f(a10 + sdl * f(a11 - a10))
        want = f(f(w) * f(b0 + md * f(b1 - b0)))
        assert L.p3host_test_score_utility(1, w, s, sd, root) == pytest.approx(float(want), abs=3e-6)
    # a flat score distribution around the reference point is worth nothing; a wide one shrinks the utility
    assert abs(L.p3host_test_score_utility(1, 0.5, 0.0, 3.0, 0.0)) < 0.02
    assert abs(L.p3host_test_score_utility(1, 0.5, 20.0, 30.0, 0.0)) < abs(L.p3host_test_score_utility(0, 0.5, 20.0, 30.0, 0.0))


def test_eval_match_player_configs_cover_every_search_path(built, tmp_path):
    """A match between two player config files (eval/main.cc --cur_config / --cand_config) through the paths
    of eval.cc:229-269: the parallel search with PUCT-V, the third-moment bonus, the optimistic-policy blend
    and the integral score utility on one side; on the other the legacy single-thread paths — Gumbel
    (use_puct false) and SearchRootPuct with visit-count move choice (use_lcb false).  Reproducible, every
    game finishes, both players search."""
    cur, cand = tmp_path / "cur.cfg", tmp_path / "cand.cfg"
    cur.write_text("n: 24\nnum_threads_per_game: 4\nuse_puct_v: true\nc_puct_v_2: 2.0\nenable_m3_bonus: true\n"
                   "m3_prior_visits: 8\np_opt_weight: 0.5\nscore_utility_mode: integral\nscore_weight: 0.4\n"
                   "var_scale_cpuct: true\nvar_scale_prior_visits: 3\nroot_fpu: 0.05\n")
    res = []
    for cand_text in ("n: 16\nk: 4\nnum_threads_per_game: 1\nuse_puct: false\nnoise_scaling: 0.5\n",
                      "n: 16\nnum_threads_per_game: 1\nuse_puct: true\nuse_lcb: false\np_opt_weight: 1.0\n"
                      "score_utility_mode: integral\n"):
        cand.write_text(cand_text)
        try:
            host_api.eval_set_paths(cur_config=str(cur), cand_config=str(cand))
            a = host_api.eval_match(None, None, num_games=4, visits_per_move=999, leaves_per_round=9, max_moves=20,
                                    num_threads=2, seed=11)
            b = host_api.eval_match(None, None, num_games=4, visits_per_move=999, leaves_per_round=9, max_moves=20,
                                    num_threads=2, seed=11)
        finally:
            host_api.eval_set_paths()
        assert a.games == 4 and a.moves == b.moves and a.visits == b.visits and a.positions == b.positions
        assert a.moves == 4 * 20 and a.visits >= 4 * 10 * (16 + 24) * 0.8   # ten moves each side, the files' budgets
        res.append((a.visits, a.positions))
    assert res[0] != res[1]
    # a player's command-line flags win over its file (eval/main.cc:146-246): --cand_n 4 against the file's 16
    try:
        host_api.eval_set_paths(cur_config=str(cur), cand_config=str(cand))
        host_api.eval_set_player_flags(cand="n: 4\n")
        c = host_api.eval_match(None, None, num_games=4, visits_per_move=999, leaves_per_round=9, max_moves=20,
                                num_threads=2, seed=11)
    finally:
        host_api.eval_set_paths()
        host_api.eval_set_player_flags()
    assert c.moves == 4 * 20 and c.visits < res[1][0] - 4 * 10 * 8


def test_eval_match_cache_sgf_and_result_file(built, tmp_path):
    """The scheduler-driven match with the per-game NN cache on, SGF recording and the result file
    (eval/main.cc: --cache_size, --recorder_path, --res_write_path): cached positions cost no engine
    slot, one SGF line per game with the players' names by colour, relative Elo written as %f."""
    rec, res = tmp_path / "rec", tmp_path / "elo.txt"
    try:
        host_api.eval_set_search(cache_entries_per_game=512)
        host_api.eval_set_paths(recorder_dir=str(rec), res_write_path=str(res))
        st = host_api.eval_match(None, None, num_games=6, visits_per_move=16, leaves_per_round=4, max_moves=24,
                                 num_threads=2, seed=3)
        host_api.eval_set_search(cache_entries_per_game=0)
        host_api.eval_set_paths()
        st0 = host_api.eval_match(None, None, num_games=6, visits_per_move=16, leaves_per_round=4, max_moves=24,
                                  num_threads=2, seed=3)
    finally:
        host_api.eval_set_search()
        host_api.eval_set_paths()
    assert st.games == 6 and st.moves == 6 * 24
    # with tree reuse and five last moves in the key (NNKey, nn_interface.h:206-228) a repeat needs the
    # same recent history, so hits are rare; what must hold is the accounting: same searches, and
    # every cached evaluation is one engine slot less
    assert st.cache_hits >= 0 and st0.cache_hits == 0
    assert st.positions + st.cache_hits == st0.positions
    assert st.winrate == pytest.approx(st.cand_wins / 6)
    files = sorted((rec / "sgf").iterdir())
    assert [f.name for f in files] == ["gen000_b000_g006_EVAL_cur_cand.done", "gen000_b000_g006_EVAL_cur_cand.sgf"]
    lines = files[1].read_text().splitlines()
    assert len(lines) == 6 and all(l.startswith("(;FF[4]GM[1]KM[7.5]RE[") for l in lines)
    assert "PB[cur]PW[cand]" in lines[0] and "PB[cand]PW[cur]" in lines[1]      # cur is Black in even games
    assert lines[0].count(";B[") + lines[0].count(";W[") == 24
    assert float(res.read_text()) == pytest.approx(st.rel_elo, abs=1e-4) or not np.isfinite(st.rel_elo)


@pytest.mark.parametrize("cap", [1, 2])
def test_eval_match_with_a_cache_smaller_than_a_round(built, cap):
    """--cache_size below the evaluations of one round (leaves_per_round = 8): a round's hits are copied
    out when they are found, because the deliveries of the same round insert into the cache and may evict
    the entry a later evaluation hit (the round-2 advisor's case: cap 1, miss then hit).  Same searches as
    without a cache, every hit one engine slot less."""
    try:
        host_api.eval_set_search(cache_entries_per_game=0)
        st0 = host_api.eval_match(None, None, num_games=6, visits_per_move=24, leaves_per_round=8, max_moves=30,
                                  num_threads=2, seed=11)
        host_api.eval_set_search(cache_entries_per_game=cap)
        st = host_api.eval_match(None, None, num_games=6, visits_per_move=24, leaves_per_round=8, max_moves=30,
                                 num_threads=2, seed=11)
    finally:
        host_api.eval_set_search()
    assert st.games == 6 and st.moves == st0.moves == 6 * 30
    assert st.positions + st.cache_hits == st0.positions and st.visits == st0.visits


@pytest.mark.parametrize("mcgs,descent,bias", [(True, 0, 0.0), (False, 1, 0.0), (True, 1, 0.3)])
def test_eval_match_graph_search_buuct_and_bias_cache(built, mcgs, descent, bias):
    """use_mcgs (McgsNodeTable, node_table.h:77-118), descent_policy bu_uct (search.h:174-251) and
    use_bias_cache on the eval path: the matches finish with consistent counts."""
    try:
        host_api.eval_set_search(descent=descent, max_o_ratio=0.8, use_mcgs=mcgs, bias_lambda=bias)
        st = host_api.eval_match(None, None, num_games=4, visits_per_move=24, leaves_per_round=4, max_moves=20,
                                 num_threads=2, seed=5)
    finally:
        host_api.eval_set_search()
    assert st.games == 4 and st.moves == 4 * 20 and st.cur_wins + st.cand_wins + st.draws == 4
    assert st.visits >= 24 * st.moves


THREADED = [
    # threads, budget, q_fn, n_fn, collision, detector, descent, graph, time_ms
    (8, 64, 2, 1, 0, 0, 0, 0, 0),      # the defaults of player_config.h
    (8, 64, 1, 1, 2, 1, 0, 0, 0),      # hard virtual loss, smart retry, n-in-flight detector
    (4, 48, 0, 1, 1, 3, 0, 0, 0),      # search_test.cc MakeParams + retry + product detector
    (8, 64, 2, 1, 0, 0, 1, 0, 0),      # BuUct descent
    (8, 64, 2, 1, 0, 0, 0, 1, 0),      # graph search (McgsNodeTable)
    (6, 0, 2, 1, 0, 0, 1, 1, 40),      # time control instead of a visit budget
]


@pytest.mark.parametrize("cfg", THREADED)
def test_threaded_concurrent_search(L, cfg):
    """mcts::Search in Mode::kConcurrent as the reference runs it (threaded_search.h: worker threads,
    node locks, two barriers per round, async NNInterface slot with kExplicit signalling, bias
    cache on): six searches of a game with tree reuse; after every one nothing is left in flight,
    n = 1 + sum(child visits) on every inner node, the visit count is in [budget, budget +
    threads) (search_test.cc:130-222) and the move is legal."""
    L.p3host_test_threaded_search.argtypes = [C.c_int] * 10 + [C.c_uint64, C.c_void_p]
    st = (C.c_long * 5)()
    mask = L.p3host_test_threaded_search(*cfg, 6, 7, st)
    assert mask == 0, f"invariant mask {mask:#x}"
    assert st[4] == 6 and st[0] >= (6 * cfg[1] if cfg[8] == 0 else 6)
    assert st[1] <= st[2] + 6 * cfg[0]                         # aborts come from collisions


def test_eval_match_thread_per_game_with_two_nn_interfaces(built):
    """eval/main.cc:380-452 as written there: a thread per game, each player's engine behind its own
    kExplicit NNInterface (num_shared_search_tasks = games, NN cache on), the threaded search per
    move with slot range [game * T, (game + 1) * T)."""
    st = host_api.eval_match_threads(None, None, num_games=4, visits_per_move=24, threads_per_game=4, max_moves=16,
                                     cache_size=1 << 14, seed=2)
    assert st.games == 4 and st.cur_wins + st.cand_wins + st.draws == 4 and st.moves == 4 * 16
    assert st.visits >= 24 * st.moves and st.batches > 0
