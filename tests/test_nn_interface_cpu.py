"""The thread-per-game NNInterface (p3achygo_amd/host/nn_interface.h), the reference's
second boundary (cc/nn/nn_interface.h:85-202).  The stress cases are the reference's
cc/nn/__tests__/nn_interface_sync_test.cc (same CountingEngine, jitter, slow threads, both
wake strategies, single and dual interfaces) shortened from 120 s to a few seconds each —
the container has 8 cores, so the 128-thread case still oversubscribes 16:1."""
import ctypes as C

import numpy as np
import pytest

from p3achygo_amd import host_api
from p3achygo_amd.engine import Result

MUTEX, GEN_COUNTER = 0, 1


@pytest.fixture(scope="module")
def L(built):
    lib = host_api.lib()
    lib.p3host_test_nn_sync.argtypes = [C.c_int] * 5 + [C.c_void_p]
    lib.p3host_test_nn_async.argtypes = [C.c_int] * 4 + [C.c_void_p]
    lib.p3host_test_nn_compacting.argtypes = [C.c_int] * 5 + [C.c_void_p] * 3
    lib.p3host_nn_new.restype = C.c_void_p
    lib.p3host_nn_new.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int,
                                  C.c_char_p, C.c_int]
    lib.p3host_nn_free.argtypes = [C.c_void_p]
    lib.p3host_nn_num_inferences.restype = C.c_long
    lib.p3host_nn_num_inferences.argtypes = [C.c_void_p]
    lib.p3host_nn_set_num_cache_last_moves.argtypes = [C.c_void_p, C.c_int]
    lib.p3host_nn_load_and_get_inference.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.p3host_nn_play_threads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p]
    lib.p3host_game_new.restype = C.c_void_p
    lib.p3host_game_new.argtypes = [C.c_float]
    lib.p3host_game_free.argtypes = [C.c_void_p]
    lib.p3host_game_play.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.p3host_prob_new.restype = C.c_void_p
    lib.p3host_prob_new.argtypes = [C.c_uint64]
    lib.p3host_prob_free.argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("strategy", [MUTEX, GEN_COUNTER])
@pytest.mark.parametrize("cache", [0, 1024])
def test_sync_no_race_no_stale_no_mixup(L, strategy, cache):
    """RunSyncTest: 128 threads, 200 us timeout, every 8th thread sleeps past several batches."""
    calls = C.c_long(0)
    mask = L.p3host_test_nn_sync(strategy, 128, 2500, cache, 0, C.byref(calls))
    assert mask == 0, f"failure mask {mask:#x} (1 race, 2 stale, 4 wrong slot, 8 wrong value)"
    assert calls.value > 128 * 20     # the interface made progress on every thread's behalf


def test_engine_slot_states_hand_over_rules(L):
    """The engine's dirty-slot rules (csrc/slot_state.h), replayed step by step: a slot stays in
    the run set until its result is fetched, so the load-during-run interleaving of
    nn_interface.cc:351-361 (picked up by run N, counted for run N+1) still yields a result."""
    assert L.p3host_test_slot_states() == 0


@pytest.mark.parametrize("strategy", [MUTEX, GEN_COUNTER])
def test_sync_over_a_compacting_engine(L, strategy):
    """The stress loop over an evaluator with the HIP engine's compaction rules whose Run() takes
    300 us with the interface lock held while workers keep loading, 100 us batch timeout: every
    result handed out was evaluated by the last run, from the caller's latest load, in its slot."""
    calls, rows, runs = C.c_long(0), C.c_long(0), C.c_long(0)
    mask = L.p3host_test_nn_compacting(strategy, 64, 2500, 100, 300, C.byref(calls), C.byref(rows), C.byref(runs))
    assert mask == 0, f"failure mask {mask:#x} (2 stale, 4 wrong slot, 8 wrong value, 16 slot not evaluated)"
    assert calls.value > 64 * 20 and runs.value > 100
    # every call was evaluated at least once; the evaluator stretches its gather with yields to
    # provoke loads landing inside it, each of which costs one re-evaluated row in the next run
    assert rows.value >= calls.value


@pytest.mark.parametrize("strategy", [MUTEX, GEN_COUNTER])
def test_sync_dual_interfaces(L, strategy):
    """RunDualInterfaceTest: 64 game threads alternate between two interfaces every ply."""
    calls = C.c_long(0)
    mask = L.p3host_test_nn_sync(strategy, 64, 2500, 0, 1, C.byref(calls))
    assert mask == 0, f"failure mask {mask:#x}"
    assert calls.value > 64 * 20


@pytest.mark.parametrize("strategy", [MUTEX, GEN_COUNTER])
@pytest.mark.parametrize("tasks,workers", [(1, 1), (1, 8), (4, 8), (16, 4)])
def test_async_explicit_signalling(L, strategy, tasks, workers):
    """LoadEntry x W -> SignalReadyForInference -> FetchEntry x W per search task, kExplicit."""
    infs = C.c_long(0)
    rounds = 200
    mask = L.p3host_test_nn_async(strategy, tasks, workers, rounds, C.byref(infs))
    assert mask == 0, f"failure mask {mask:#x}"
    # at least one inference per round; the 400 us timeout may split a round into partial
    # batches (nn_interface.cc:293-318), never more than one per loaded slot
    assert rounds <= infs.value <= rounds * tasks * workers


def _eval(L, nn, tid, game, color, prob):
    r = Result()
    L.p3host_nn_load_and_get_inference(nn, tid, game, color, prob, C.byref(r))
    return r


def test_cache_keys_and_hits(L):
    """nn_interface.cc:92-133: an identical (colour, board, last moves, komi) is served from the
    cache without an inference; a different colour or move history is a miss; with
    SetNumCacheLastMoves(1) (selfplay/main.cc:177) only the last move is part of the key."""
    nn = L.p3host_nn_new(0, None, None, 0, 1, 400, 64, GEN_COUNTER, None, 0)
    prob = L.p3host_prob_new(7)
    g = L.p3host_game_new(7.5)
    try:
        _eval(L, nn, 0, g, 1, prob)
        assert L.p3host_nn_num_inferences(nn) == 1
        r = _eval(L, nn, 0, g, 1, prob)
        assert L.p3host_nn_num_inferences(nn) == 1          # hit
        assert r.move_probs[0] == pytest.approx(1 / 362)
        _eval(L, nn, 0, g, -1, prob)
        assert L.p3host_nn_num_inferences(nn) == 2          # other colour to move
        # two move orders reaching the same board with the same last move
        a, b = L.p3host_game_new(7.5), L.p3host_game_new(7.5)
        for (i, j, c) in [(3, 3, 1), (15, 15, -1), (3, 15, 1), (15, 3, -1), (9, 9, 1)]:
            assert L.p3host_game_play(a, i, j, c)
        for (i, j, c) in [(3, 15, 1), (15, 3, -1), (3, 3, 1), (15, 15, -1), (9, 9, 1)]:
            assert L.p3host_game_play(b, i, j, c)
        _eval(L, nn, 0, a, -1, prob)
        n0 = L.p3host_nn_num_inferences(nn)
        _eval(L, nn, 0, b, -1, prob)
        assert L.p3host_nn_num_inferences(nn) == n0 + 1     # five last moves differ: miss
        L.p3host_nn_set_num_cache_last_moves(nn, 1)
        _eval(L, nn, 0, a, -1, prob)
        n1 = L.p3host_nn_num_inferences(nn)
        assert n1 == n0 + 2                                  # the key changed shape: miss once
        _eval(L, nn, 0, b, -1, prob)
        assert L.p3host_nn_num_inferences(nn) == n1          # same board, same last move: hit
        L.p3host_game_free(a)
        L.p3host_game_free(b)
    finally:
        L.p3host_game_free(g)
        L.p3host_prob_free(prob)
        L.p3host_nn_free(nn)


def test_lru_eviction(L):
    """core::LRUCache (lru_cache.h:17-64): capacity 2 keeps the two most recently used keys."""
    nn = L.p3host_nn_new(0, None, None, 0, 1, 400, 2, GEN_COUNTER, None, 0)
    prob = L.p3host_prob_new(1)
    games = [L.p3host_game_new(k) for k in (5.5, 6.5, 7.5)]
    try:
        inf = lambda: L.p3host_nn_num_inferences(nn)
        _eval(L, nn, 0, games[0], 1, prob)
        _eval(L, nn, 0, games[1], 1, prob)
        assert inf() == 2
        _eval(L, nn, 0, games[0], 1, prob)       # hit; 0 becomes most recent
        assert inf() == 2
        _eval(L, nn, 0, games[2], 1, prob)       # evicts 1
        assert inf() == 3
        _eval(L, nn, 0, games[0], 1, prob)
        assert inf() == 3
        _eval(L, nn, 0, games[1], 1, prob)       # was evicted
        assert inf() == 4
    finally:
        for g in games:
            L.p3host_game_free(g)
        L.p3host_prob_free(prob)
        L.p3host_nn_free(nn)


def test_thread_per_game_driver_with_unregistering_threads(L):
    """32 game threads of different lengths would deadlock a batch that waits for everyone:
    finished threads unregister (nn_interface.cc:160-170) and the rest keep being served."""
    T, M = 32, 12
    nn = L.p3host_nn_new(0, None, None, 0, T, 400, 0, MUTEX, None, 0)
    out = (Result * (T * M))()
    try:
        L.p3host_nn_play_threads(nn, T, M, 100, C.byref(out))
        probs = np.array([out[i].move_probs[361] for i in range(T * M)])
        assert np.allclose(probs, 1 / 362)
        # batching happened: far fewer inferences than evaluations
        assert M <= L.p3host_nn_num_inferences(nn) < T * M
    finally:
        L.p3host_nn_free(nn)


def test_nn_interface_over_an_engine_side_cache_unrotates_by_the_stored_symmetry():
    """NNInterface with the cache in the engine (EnableDeviceCache; include/p3hip.h p3hip_cache_*) behind its
    per-thread LRUs (cc/nn/nn_interface.cc:107-132).  A fake engine with the HIP engine's cache rules returns, on a
    hit, the result it stored under the symmetry of THAT evaluation; every call draws a fresh random symmetry, so a
    hit whose result were un-rotated by the caller's own symmetry would show rotated stones.  200 positions x 6
    rounds through the blocking and the async entry points: every result reads as the game's own stones, each
    distinct key is evaluated once, every later request is a hit."""
    import ctypes as C
    from p3achygo_amd import host_api
    L = host_api.lib()
    out = (C.c_long * 5)()
    L.p3host_test_nn_device_cache.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_long)]
    assert L.p3host_test_nn_device_cache(200, 6, 12345, 0, out) == 0       # no host LRU: every repeat reaches the table
    bad, evaluated, hits, counted, keys = list(out)
    assert bad == 0
    assert evaluated == keys and 150 <= keys <= 200          # random playouts may repeat a position
    assert hits == counted == 200 * 6 - evaluated
    # with the interface's own LRU in front (the reference's order, nn_interface.cc:112-118): a thread's own
    # repeats are answered at once and never reach the engine; only first sightings are evaluated
    assert L.p3host_test_nn_device_cache(200, 6, 12345, 1 << 10, out) == 0
    bad, evaluated, hits, counted, keys = list(out)
    assert bad == 0 and evaluated == keys and hits == counted == 0
