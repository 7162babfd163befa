"""ctypes mirrors of the C-ABI PODs (include/p3hip.h) and a synthetic position source.

`Features` mirrors nn::GoFeatures (cc/nn/engine/go_features.h:12-22) and `Result` mirrors
nn::NNInferResult (cc/nn/engine/engine.h:12-20) field for field.

`random_positions` produces seeded random-legal playout positions (move number uniform in
[0, max_moves], komi 7.5) with true stone / liberty planes, the synthetic input shape
SURVEY.md §8d prescribes for the engine benchmark.  Ladder planes are a seeded subset of
the 1-2 liberty groups (the network only needs the value distribution of real planes;
the exact ladder reader lives in the C++ host rules engine).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np

NUM_LOCS = 361
NUM_MOVES = 362
BL = 19
RAW_LEN = 1889


class Loc(C.Structure):
    _fields_ = [("i", C.c_int32), ("j", C.c_int32)]


class Features(C.Structure):
    _fields_ = [
        ("bsize", C.c_int32),
        ("color", C.c_int8),
        ("komi", C.c_float),
        ("board", C.c_int8 * NUM_LOCS),
        ("last_moves", Loc * 5),
        ("stones_atari", C.c_int8 * NUM_LOCS),
        ("stones_two_liberties", C.c_int8 * NUM_LOCS),
        ("stones_three_liberties", C.c_int8 * NUM_LOCS),
        ("stones_laddered", C.c_int8 * NUM_LOCS),
    ]


class _OptProbs(C.Structure):
    _fields_ = [("v", C.c_float * NUM_MOVES)]


class Result(C.Structure):
    _fields_ = [
        ("move_logits", C.c_float * NUM_MOVES),
        ("move_probs", C.c_float * NUM_MOVES),
        ("value_probs", C.c_float * 2),
        ("score_probs", C.c_float * 800),
        ("_pad", C.c_float * 2),  # alignas(16) opt_move_probs: 1526 floats -> 1528
        ("opt_move_probs", C.c_float * NUM_MOVES),
        ("err2_outcome", C.c_float),
        ("_tail", C.c_float * 1),  # struct size rounds up to a multiple of 16 bytes
    ]


assert C.sizeof(Features) == 1860, C.sizeof(Features)
assert C.sizeof(Result) == 4 * 1892, C.sizeof(Result)
assert Result.opt_move_probs.offset % 16 == 0


def features_dtype() -> np.dtype:
    return np.dtype(Features)


def result_to_dict(r: Result) -> dict:
    return {
        "move_logits": np.ctypeslib.as_array(r.move_logits).copy(),
        "move_probs": np.ctypeslib.as_array(r.move_probs).copy(),
        "value_probs": np.ctypeslib.as_array(r.value_probs).copy(),
        "score_probs": np.ctypeslib.as_array(r.score_probs).copy(),
        "opt_move_probs": np.ctypeslib.as_array(r.opt_move_probs).copy(),
        "err2_outcome": float(r.err2_outcome),
    }


# ------------------------------------------------------------------ synthetic positions

_NBRS = None


def _neighbors():
    global _NBRS
    if _NBRS is None:
        nb = []
        for p in range(NUM_LOCS):
            i, j = divmod(p, BL)
            l = []
            if i > 0: l.append(p - BL)
            if i < BL - 1: l.append(p + BL)
            if j > 0: l.append(p - 1)
            if j < BL - 1: l.append(p + 1)
            nb.append(l)
        _NBRS = nb
    return _NBRS


def _group(board, p):
    nb = _neighbors()
    col = board[p]
    stones = {p}
    libs = set()
    stack = [p]
    while stack:
        q = stack.pop()
        for r in nb[q]:
            if board[r] == 0:
                libs.add(r)
            elif board[r] == col and r not in stones:
                stones.add(r)
                stack.append(r)
    return stones, libs


def _play(board, p, col):
    """Plays col at p if legal (no suicide); returns captured-stone count or -1."""
    nb = _neighbors()
    if board[p] != 0:
        return -1
    board[p] = col
    cap = 0
    for r in nb[p]:
        if board[r] == -col:
            st, lb = _group(board, r)
            if not lb:
                for s in st:
                    board[s] = 0
                cap += len(st)
    st, lb = _group(board, p)
    if not lb:
        board[p] = 0
        return -1
    return cap


def _liberty_planes(board):
    planes = [np.zeros(NUM_LOCS, np.int8) for _ in range(3)]
    seen = set()
    for p in range(NUM_LOCS):
        if board[p] == 0 or p in seen:
            continue
        st, lb = _group(board, p)
        seen |= st
        n = len(lb)
        if 1 <= n <= 3:
            for s in st:
                planes[n - 1][s] = board[p]
    return planes


def random_positions(n: int, seed: int = 0, max_moves: int = 250, n_games: int = 0,
                     min_moves: int = 0, pass_prob: float = 0.02, komis=(7.5,)) -> np.ndarray:
    """n seeded random-legal playout positions as a numpy array of `Features` records.

    The defaults reproduce the round-1 generator bit for bit (fixtures depend on it);
    min_moves / pass_prob / komis widen the distribution (late-game, pass-heavy, komi of
    either sign: one komi per game, drawn after the stops so the default stream is unchanged).
    """
    rng = np.random.default_rng(seed)
    out = np.zeros(n, dtype=features_dtype())
    n_games = n_games or max(1, min(n, 16))
    per_game: List[List[int]] = [[] for _ in range(n_games)]
    for k in range(n):
        per_game[k % n_games].append(k)
    for g in range(n_games):
        idxs = per_game[g]
        if not idxs:
            continue
        stops = sorted(int(rng.integers(min_moves, max_moves + 1)) for _ in idxs)
        komi = float(komis[int(rng.integers(0, len(komis)))]) if len(komis) > 1 else float(komis[0])
        board = np.zeros(NUM_LOCS, np.int8)
        hist = []  # (i, j) or pass
        col = 1
        move_no = 0
        si = 0
        while si < len(stops):
            while si < len(stops) and stops[si] == move_no:
                k = idxs[si]
                f = out[k]
                f["bsize"] = BL
                f["color"] = col
                f["komi"] = komi
                f["board"] = board
                lm = [(-1, -1)] * 5 + hist
                for t in range(5):
                    f["last_moves"][t]["i"], f["last_moves"][t]["j"] = lm[len(lm) - 5 + t]
                a, b2, c3 = _liberty_planes(board)
                f["stones_atari"], f["stones_two_liberties"], f["stones_three_liberties"] = a, b2, c3
                lad = np.where(rng.random(NUM_LOCS) < 0.5, a + b2, 0).astype(np.int8)
                f["stones_laddered"] = lad
                si += 1
            # play one random legal move (pass with small probability)
            played = False
            if rng.random() > pass_prob:
                for p in rng.permutation(NUM_LOCS)[:40]:
                    if _play(board, int(p), col) >= 0:
                        hist.append(divmod(int(p), BL))
                        played = True
                        break
            if not played:
                hist.append((19, 0))
            col = -col
            move_no += 1
    return out
