"""ctypes access to libp3host.so (rules engine, RNG, symmetry, features, self-play host)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libp3host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        vp, i32, u64, f32 = C.c_void_p, C.c_int, C.c_uint64, C.c_float
        sig = {
            "p3host_prng_new": (vp, [u64, u64, u64, u64]), "p3host_prng_free": (None, [vp]),
            "p3host_prng_next": (C.c_uint32, [vp]), "p3host_prng_next64": (u64, [vp]),
            "p3host_prng_next128": (None, [vp, C.POINTER(u64), C.POINTER(u64)]),
            "p3host_rand_range": (i32, [vp, i32, i32]),
            "p3host_prob_new": (vp, [u64]), "p3host_prob_free": (None, [vp]),
            "p3host_prob_uniform": (f32, [vp]), "p3host_prob_gumbel": (f32, [vp]),
            "p3host_transform_index": (i32, [i32, i32, i32]), "p3host_transform_inv": (i32, [i32, i32, i32]),
            "p3host_board_new": (vp, [f32, i32]), "p3host_board_handicap": (vp, [i32, f32]),
            "p3host_board_copy": (vp, [vp]), "p3host_board_free": (None, [vp]),
            "p3host_board_play": (i32, [vp, i32, i32, i32]), "p3host_board_dry": (i32, [vp, i32, i32, i32]),
            "p3host_board_pass": (i32, [vp, i32]), "p3host_board_place_raw": (None, [vp, i32, i32, i32]),
            "p3host_board_is_game_over": (i32, [vp]), "p3host_board_is_all_pass_alive": (i32, [vp]),
            "p3host_board_move_count": (i32, [vp]), "p3host_board_hash": (u64, [vp]),
            "p3host_board_position": (None, [vp, vp]), "p3host_board_pass_alive": (None, [vp, vp]),
            "p3host_board_calc_pass_alive": (None, [vp, i32]),
            "p3host_board_scores": (None, [vp, C.POINTER(f32), C.POINTER(f32), vp]),
            "p3host_board_liberties_plane": (None, [vp, i32, vp]), "p3host_board_laddered": (None, [vp, vp]),
            "p3host_board_group_liberties": (i32, [vp, i32, i32]),
            "p3host_board_group_id": (i32, [vp, i32, i32]),
            "p3host_game_new": (vp, [f32]), "p3host_game_free": (None, [vp]),
            "p3host_game_play": (i32, [vp, i32, i32, i32]), "p3host_game_num_moves": (i32, [vp]),
            "p3host_game_features": (None, [vp, i32, i32, vp]),
            "p3host_unapply_symmetry": (None, [i32, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class Board:
    """Thin handle over p3::Board (cc/game/board.h surface)."""

    def __init__(self, komi: float = 7.5, prohibit_pass_alive: bool = True, _h=None):
        self._L = lib()
        self._h = _h or self._L.p3host_board_new(komi, int(prohibit_pass_alive))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.p3host_board_free(self._h)
            self._h = None

    def copy(self):
        return Board(_h=self._L.p3host_board_copy(self._h))

    def play(self, i, j, color) -> bool:
        return self._L.p3host_board_play(self._h, i, j, color) == 0

    def play_status(self, i, j, color) -> int:
        return self._L.p3host_board_play(self._h, i, j, color)

    def dry(self, i, j, color) -> bool:
        return self._L.p3host_board_dry(self._h, i, j, color) == 0

    def dry_status(self, i, j, color) -> int:
        return self._L.p3host_board_dry(self._h, i, j, color)

    def pass_(self, color):
        self._L.p3host_board_pass(self._h, color)

    def place_raw(self, i, j, color):
        self._L.p3host_board_place_raw(self._h, i, j, color)

    def _grid(self, fn, *a):
        out = np.zeros(361, np.int8)
        fn(self._h, *a, out.ctypes.data)
        return out.reshape(19, 19)

    def position(self):
        return self._grid(self._L.p3host_board_position)

    def pass_alive(self):
        return self._grid(self._L.p3host_board_pass_alive)

    def calc_pass_alive(self, color=0):
        self._L.p3host_board_calc_pass_alive(self._h, color)

    def is_all_pass_alive(self) -> bool:
        return bool(self._L.p3host_board_is_all_pass_alive(self._h))

    def is_game_over(self) -> bool:
        return bool(self._L.p3host_board_is_game_over(self._h))

    def scores(self):
        b, w = C.c_float(), C.c_float()
        own = np.zeros(361, np.int8)
        self._L.p3host_board_scores(self._h, C.byref(b), C.byref(w), own.ctypes.data)
        return b.value, w.value, own.reshape(19, 19)

    def liberties_plane(self, n):
        return self._grid(self._L.p3host_board_liberties_plane, n)

    def laddered(self):
        return self._grid(self._L.p3host_board_laddered)

    def group_liberties(self, i, j) -> int:
        return self._L.p3host_board_group_liberties(self._h, i, j)

    def group_id(self, i, j) -> int:
        return self._L.p3host_board_group_id(self._h, i, j)

    def hash(self) -> int:
        return self._L.p3host_board_hash(self._h)


class SelfPlayStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("positions", C.c_long), ("moves", C.c_long),
                ("games", C.c_long), ("black_wins", C.c_long), ("batches", C.c_long),
                ("gpu_seconds", C.c_double), ("host_seconds", C.c_double), ("cache_hits", C.c_long),
                ("advance_batches", C.c_long), ("games_past_opening", C.c_long), ("seconds_fit", C.c_double), ("rounds", C.c_long)]


def selfplay_run(weights: str | None, num_games: int, num_threads: int, seconds: float,
                 default_n: int = 32, default_k: int = 5, selected_n: int = 32, selected_k: int = 5,
                 max_moves: int = 600, warmup_batches: int = 4, seed: int = 1, device: int = 0,
                 engine_lib: str | None = None) -> SelfPlayStats:
    """Runs the self-play scheduler.  weights=None -> NullEvaluator (uniform policy, no GPU)."""
    L = lib()
    L.p3host_selfplay_run.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint64,
                                      C.POINTER(SelfPlayStats), C.c_char_p]
    st = SelfPlayStats()
    err = C.create_string_buffer(256)
    if weights is None:
        elib = engine_lib.encode() if engine_lib == "hash" else None   # "hash": HashEvaluator, position-dependent test results
    else:
        elib = (engine_lib or os.path.join(_HERE, "csrc", "libp3hip.so")).encode()
    rc = L.p3host_selfplay_run(elib, weights.encode() if weights else None, device, num_games, num_threads,
                               default_n, default_k, selected_n, selected_k, max_moves, seconds,
                               warmup_batches, seed, C.byref(st), err)
    if rc != 0:
        raise RuntimeError(f"selfplay_run rc={rc}: {err.value.decode()}")
    return st


class EngineBenchmarkStats(C.Structure):
    _fields_ = [("rounds", C.c_long), ("positions", C.c_long), ("avg_run_us", C.c_double),
                ("loop_seconds", C.c_double), ("checksum", C.c_double)]


def engine_benchmark(weights: str, positions: np.ndarray, batch: int, warmup_runs: int = 100, max_rounds: int = 1001,
                     device: int = 0, engine_lib: str | None = None) -> EngineBenchmarkStats:
    """nn::Benchmark (cc/nn/engine/benchmark_engine.cc:77-108) in C++ over the C ABI: 100 warm-up runs, then
    <= 1001 rounds of LoadBatch x B -> RunInference -> GetBatch x B, timed around RunInference.
    `positions`: array of p3hip_features records (features.FEATURES_DTYPE)."""
    L = lib()
    L.p3host_engine_benchmark.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.POINTER(EngineBenchmarkStats), C.c_char_p]
    st = EngineBenchmarkStats()
    err = C.create_string_buffer(256)
    pos = np.ascontiguousarray(positions)
    elib = (engine_lib or os.path.join(_HERE, "csrc", "libp3hip.so")).encode()
    rc = L.p3host_engine_benchmark(elib, weights.encode(), device, batch, pos.ctypes.data, len(pos), warmup_runs,
                                   max_rounds, C.byref(st), err)
    if rc != 0:
        raise RuntimeError(f"engine_benchmark rc={rc}: {err.value.decode()}")
    return st


def set_bias_cache(bias_cache_lambda: float = 0.0, bias_cache_alpha: float = 0.8) -> None:
    """--bias_cache_lambda / --bias_cache_alpha of subsequent self-play runs (selfplay/main.cc:58-61);
    lambda 0 = off (the reference default); config/v4.json (BASELINE configs[0]) sets 0.3 / 0.8."""
    L = lib()
    L.p3host_selfplay_set_bias_cache.argtypes = [C.c_float, C.c_float]
    L.p3host_selfplay_set_bias_cache(bias_cache_lambda, bias_cache_alpha)


def set_calibration_file(path: str = "") -> None:
    """--sel_mult_calibration_file of subsequent self-play runs (selfplay/main.cc:64-67): per-generation
    thresholds of the training-move selection multiplier; empty = the built-in ones."""
    L = lib()
    L.p3host_selfplay_set_calibration_file.argtypes = [C.c_char_p]
    L.p3host_selfplay_set_calibration_file(path.encode())


def set_early_stopping(enabled: bool) -> None:
    """--early_stopping_enabled of subsequent self-play runs (selfplay/main.cc:68; off by default)."""
    L = lib()
    L.p3host_selfplay_set_early_stopping.argtypes = [C.c_int]
    L.p3host_selfplay_set_early_stopping(int(enabled))


def last_bias_counters():
    """(bias-cache entries pruned, sum over moves of |root adjustment|) of the last run / game."""
    L = lib()
    L.p3host_selfplay_last_bias_pruned.restype = C.c_long
    L.p3host_selfplay_last_bias_adj.restype = C.c_double
    return L.p3host_selfplay_last_bias_pruned(), L.p3host_selfplay_last_bias_adj()


def selfplay_one_game(weights: str | None, default_n: int, default_k: int, max_moves: int, seed: int,
                      engine_lib: str | None = None):
    """One complete game on one thread; returns (moves, black_score, white_score, evals)."""
    L = lib()
    L.p3host_selfplay_one_game.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                           C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                           C.POINTER(C.c_long), C.c_char_p]
    mv = np.zeros(1024, np.int32)
    b, w, ev = C.c_float(), C.c_float(), C.c_long()
    err = C.create_string_buffer(256)
    elib = (engine_lib or os.path.join(_HERE, "csrc", "libp3hip.so")).encode() if weights else None
    n = L.p3host_selfplay_one_game(elib, weights.encode() if weights else None, default_n, default_k,
                                   max_moves, seed, mv.ctypes.data, len(mv), C.byref(b), C.byref(w),
                                   C.byref(ev), err)
    if n < 0:
        raise RuntimeError(f"selfplay_one_game rc={n}: {err.value.decode()}")
    return mv[:n].copy(), b.value, w.value, ev.value


def set_policy(init_state_sampling: bool = True, use_seen_state_prob: float = 0.5, sel_mult_base: float = 0.0,
               sel_mult_scale_factor: float = 1.0) -> None:
    """Game-loop policy of subsequent selfplay_run calls (reference defaults, selfplay/main.cc:48-57);
    init_state_sampling=False: every game from the empty board at komi 7.5, no forks."""
    L = lib()
    L.p3host_selfplay_set_policy.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
    L.p3host_selfplay_set_policy(int(init_state_sampling), use_seen_state_prob, sel_mult_base, sel_mult_scale_factor)


def set_ladder_budget(nodes: int) -> None:
    """Work bound of one ladder read-out (process-wide).  0, the default, is the reference's
    behaviour (depth bound only, cc/game/board.cc:780-783): planes 13/14 bit-exact.  A positive
    budget is the opt-in throughput mode of the self-play host: an exhausted read-out reads "not
    laddered" and is counted (ladder_stats()[3])."""
    L = lib()
    L.p3host_set_ladder_budget.argtypes = [C.c_long]
    L.p3host_set_ladder_budget(int(nodes))


def ladder_stats():
    """(read-outs, nodes, max nodes of one read-out, budget hits) since process start."""
    L = lib()
    out = (C.c_long * 4)()
    L.p3host_ladder_stats(out)
    return tuple(out)


def set_groups(n: int) -> None:
    """Game groups (engine instances) of subsequent selfplay_run calls; default 2."""
    L = lib()
    L.p3host_selfplay_set_groups.argtypes = [C.c_int]
    L.p3host_selfplay_set_groups(n)


def set_lanes(lanes: int = 1, max_inflight: int = 1) -> None:
    """Lanes per game group (engine instances whose batches the group's games fill in turn) and how many playouts of
    one search may wait for results at once (GumbelSearch::IssueNext; at most 4).  1 / 1, the default: host and GPU
    alternate within a group.  2 / 4: one group overlaps its host work with its own forward passes — BASELINE
    configs[2] as stated (1024 games, batch 1024) without a second game group.  The games are the same either way."""
    L = lib()
    L.p3host_selfplay_set_lanes.argtypes = [C.c_int, C.c_int]
    L.p3host_selfplay_set_lanes(int(lanes), int(max_inflight))


def set_test_slow_games(us: int) -> None:
    """Tests: every 61st (game, host phase) pair of subsequent selfplay_run calls sleeps `us` microseconds before it
    advances — a stand-in for a slow exact ladder read-out; 0 = off."""
    L = lib()
    L.p3host_selfplay_set_test_slow_games.argtypes = [C.c_long]
    L.p3host_selfplay_set_test_slow_games(int(us))


def last_handed_over():
    """(host phases, games) of the last selfplay_run whose batch left without a last few slow games (two or more lanes:
    the stragglers finish on the pool and load into the group's next batch)."""
    L = lib()
    L.p3host_selfplay_last_handed_over.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
    a, b = C.c_long(), C.c_long()
    L.p3host_selfplay_last_handed_over(C.byref(a), C.byref(b))
    return a.value, b.value


def last_first_game_digests() -> np.ndarray:
    """Per game runner of the last selfplay_run: a digest of its first finished game (moves and score), 0 if none."""
    L = lib()
    L.p3host_selfplay_last_first_game_digests.argtypes = [C.c_void_p, C.c_int]
    n = L.p3host_selfplay_last_first_game_digests(None, 0)
    out = np.zeros(max(n, 1), np.uint64)
    L.p3host_selfplay_last_first_game_digests(out.ctypes.data, n)
    return out[:n]


def set_advance_limit(max_batches: int) -> None:
    """> 0: before the warm-up every group of subsequent selfplay_run calls runs untimed batches, at most
    this many, until all its games have left their raw-policy opening (self_play_thread.cc:44,363-366);
    0 = off (the default)."""
    L = lib()
    L.p3host_selfplay_set_advance_limit.argtypes = [C.c_int]
    L.p3host_selfplay_set_advance_limit(int(max_batches))


def set_step_limit(batches: int) -> None:
    """> 0: subsequent selfplay_run calls time exactly `batches` engine batches (bench.py --steps),
    whichever game groups they fall in, instead of running for `seconds`; 0 = time limit."""
    L = lib()
    L.p3host_selfplay_set_step_limit.argtypes = [C.c_long]
    L.p3host_selfplay_set_step_limit(int(batches))


def set_step_rounds(rounds: int) -> None:
    """> 0: subsequent selfplay_run calls time `rounds` ROUNDS (bench.py --steps): the window opens and closes on
    completions of the SAME game group (the one that finishes its warm-up last), `rounds` of its batches apart, and
    counts every batch of any group that completes inside — one per group and round in steady state.  Takes
    precedence over set_step_limit; 0 = off."""
    L = lib()
    L.p3host_selfplay_set_step_rounds.argtypes = [C.c_long]
    L.p3host_selfplay_set_step_rounds(int(rounds))


def last_run_counters():
    """(reuse-buffer insertions, training examples written) of the last selfplay_run."""
    L = lib()
    L.p3host_selfplay_last_reuse_added.restype = C.c_long
    L.p3host_selfplay_last_examples.restype = C.c_long
    return L.p3host_selfplay_last_reuse_added(), L.p3host_selfplay_last_examples()


def set_recorder(directory: str, gen: int = 0, worker_id: str = "0", flush_interval: int = 128) -> None:
    """Game recording (<dir>/sgf, <dir>/chunks) for subsequent selfplay_run calls ('' disables)."""
    L = lib()
    L.p3host_selfplay_set_recorder.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.p3host_selfplay_set_recorder(directory.encode(), gen, worker_id.encode(), flush_interval)


def sgf_from_moves(moves, komi: float = 7.5, write_result: bool = False, b_name: str = "testB",
                   w_name: str = "testW") -> str:
    L = lib()
    L.p3host_sgf_from_moves.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_char_p, C.c_char_p,
                                        C.c_char_p, C.c_int]
    mv = np.asarray(moves, np.int32)
    out = C.create_string_buffer(1 << 16)
    n = L.p3host_sgf_from_moves(mv.ctypes.data if len(mv) else None, len(mv), komi, int(write_result),
                                b_name.encode(), w_name.encode(), out, len(out))
    assert n >= 0
    return out.value.decode()


class EvalStats(C.Structure):
    _fields_ = [("games", C.c_int), ("cur_wins", C.c_int), ("cand_wins", C.c_int), ("draws", C.c_int),
                ("resignations", C.c_int), ("moves", C.c_long), ("visits", C.c_long), ("collisions", C.c_long),
                ("positions", C.c_long), ("batches", C.c_long), ("seconds", C.c_double), ("cache_hits", C.c_long),
                ("winrate", C.c_float), ("c95", C.c_float), ("rel_elo", C.c_float), ("elo_c95", C.c_float)]


def eval_set_search(mode: int = 0, q_fn: int = 2, n_fn: int = 1, collision: int = 0, detector: int = 0,
                    descent: int = 0, max_o_ratio: float = 1.0, use_mcgs: bool = False, bias_lambda: float = 0.0,
                    bias_alpha: float = 0.8, cache_entries_per_game: int = 256) -> None:
    """Parallel-search knobs of subsequent eval matches (PlayerSearchConfig, player_config.h:60-108):
    mode 0 concurrent / 1 batch; q_fn 0 identity / 1 virtual_loss / 2 virtual_loss_soft; n_fn 0
    identity / 1 virtual_visit; collision 0 abort / 1 retry / 2 smart_retry; detector 0 noop / 1
    n_in_flight / 2 level_saturation / 3 product; descent 0 deterministic / 1 bu_uct."""
    L = lib()
    L.p3host_eval_set_search.argtypes = [C.c_int] * 5
    L.p3host_eval_set_search(mode, q_fn, n_fn, collision, detector)
    L.p3host_eval_set_search_ex.argtypes = [C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int]
    L.p3host_eval_set_search_ex(descent, max_o_ratio, int(use_mcgs), bias_lambda, bias_alpha, cache_entries_per_game)


def eval_set_paths(cur_config: str = "", cand_config: str = "", recorder_dir: str = "", res_write_path: str = "") -> None:
    """--cur_config / --cand_config / --recorder_path / --res_write_path of subsequent eval matches."""
    L = lib()
    L.p3host_eval_set_paths.argtypes = [C.c_char_p] * 4
    L.p3host_eval_set_paths(cur_config.encode(), cand_config.encode(), recorder_dir.encode(), res_write_path.encode())


def eval_set_player_flags(cur: str = "", cand: str = "") -> None:
    """The per-player command-line flags of eval/main.cc (--cur_n, --cand_use_puct_v, ...) of subsequent
    matches, as "key: value" lines; they override the player's config file (main.cc:146-246)."""
    L = lib()
    L.p3host_eval_set_player_flags.argtypes = [C.c_char_p, C.c_char_p]
    L.p3host_eval_set_player_flags(cur.encode(), cand.encode())


def eval_match(cur_weights: str | None, cand_weights: str | None, num_games: int, visits_per_move: int = 128,
               leaves_per_round: int = 8, max_moves: int = 600, num_threads: int = 8, seed: int = 1,
               device: int = 0, engine_lib: str | None = None) -> EvalStats:
    """Model-vs-model games with the batch parallel search (cc/eval), all games advanced together by
    the batching scheduler.  weights=None -> NullEvaluator."""
    L = lib()
    L.p3host_eval_match.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_uint64, C.POINTER(EvalStats), C.c_char_p]
    st = EvalStats()
    err = C.create_string_buffer(256)
    elib = (engine_lib or os.path.join(_HERE, "csrc", "libp3hip.so")).encode() if cur_weights else None
    rc = L.p3host_eval_match(elib, cur_weights.encode() if cur_weights else None,
                             cand_weights.encode() if cand_weights else None, device, num_games, visits_per_move,
                             leaves_per_round, max_moves, num_threads, seed, C.byref(st), err)
    if rc != 0:
        raise RuntimeError(f"eval_match rc={rc}: {err.value.decode()}")
    return st


def set_device_nn_cache(log2_entries: int) -> None:
    """Thread-per-game drivers: NN cache in the engine's HBM table (2^log2_entries entries per engine); 0 = host LRUs."""
    lib().p3host_set_device_nn_cache(int(log2_entries))


def device_nn_cache_hits() -> int:
    L = lib()
    L.p3host_device_nn_cache_hits.restype = C.c_long
    return int(L.p3host_device_nn_cache_hits())


def device_nn_cache_lookups() -> int:
    L = lib()
    L.p3host_device_nn_cache_lookups.restype = C.c_long
    return int(L.p3host_device_nn_cache_lookups())


def eval_match_threads(cur_weights: str | None, cand_weights: str | None, num_games: int, visits_per_move: int = 128,
                       threads_per_game: int = 8, max_moves: int = 600, cache_size: int = 1 << 20, seed: int = 1,
                       device: int = 0, engine_lib: str | None = None) -> EvalStats:
    """The reference's own shape of the match (eval/main.cc:380-452): one thread per game, two
    NNInterfaces (kExplicit signalling, NN cache on), the threaded mcts::Search per move."""
    L = lib()
    L.p3host_eval_match_threads.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_long, C.c_uint64, C.POINTER(EvalStats), C.c_char_p]
    st = EvalStats()
    err = C.create_string_buffer(256)
    elib = (engine_lib or os.path.join(_HERE, "csrc", "libp3hip.so")).encode() if cur_weights else None
    rc = L.p3host_eval_match_threads(elib, cur_weights.encode() if cur_weights else None,
                                     cand_weights.encode() if cand_weights else None, device, num_games,
                                     visits_per_move, threads_per_game, max_moves, cache_size, seed, C.byref(st), err)
    if rc != 0:
        raise RuntimeError(f"eval_match_threads rc={rc}: {err.value.decode()}")
    return st
