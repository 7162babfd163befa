"""Batch parallel search and evaluation matches (p3achygo_amd/host/parallel_search.h,
eval_match.h).  The search cases are the reference's cc/mcts/__tests__/search_test.cc:130-222
(NullEngine, pre-evaluated root, same budgets and invariants) plus tree-consistency checks."""
import ctypes as C

import numpy as np
import pytest

from p3achygo_amd import host_api


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
