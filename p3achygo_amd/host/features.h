// features.h — GoFeatures of a position, restating NNInterface::LoadBatch
// (cc/nn/nn_interface.cc:245-277): last five moves, stones and the four liberty/ladder
// grids, all mapped through the chosen D4 symmetry; and the inverse mapping of the three
// output grids NNInterface::GetBatch applies (cc/nn/nn_interface.h:251-290).
#pragma once
#include "../../include/p3hip.h"
#include "board.h"
#include "symmetry.h"

namespace p3 {

inline void FillFeatures(const Position& pos, Color color_to_move, Symmetry sym, p3hip_features* f) {
  f->bsize = kBoardLen;
  f->color = color_to_move;
  f->komi = pos.komi();
  for (int i = 0; i < P3HIP_NUM_LAST_MOVES; ++i) {
    Loc l = pos.last[i].loc;   // noop-padded at the front exactly like Game::moves()
    if (l != kPassLoc && l != kNoopLoc) l = AsLoc(TransformIndex(sym, Idx(l), kBoardLen));
    f->last_moves[i].i = l.i;
    f->last_moves[i].j = l.j;
  }
  const Board& b = pos.board;
  ApplySymmetry(sym, b.position().data(), f->board, kBoardLen);
  ApplySymmetry(sym, b.GetStonesInAtari().data(), f->stones_atari, kBoardLen);
  ApplySymmetry(sym, b.GetStonesWithLiberties(2).data(), f->stones_two_liberties, kBoardLen);
  ApplySymmetry(sym, b.GetStonesWithLiberties(3).data(), f->stones_three_liberties, kBoardLen);
  ApplySymmetry(sym, b.GetLadderedStones().data(), f->stones_laddered, kBoardLen);
}

inline void FillFeatures(const Game& game, Color color_to_move, Symmetry sym, p3hip_features* f) {
  FillFeatures(Position(game), color_to_move, sym, f);
}

// undo the symmetry on the 361-point parts of the result (pass entry 361 is untouched)
inline void UnapplySymmetry(Symmetry sym, p3hip_result* r) {
  float tmp[kNumLocs];
  ApplyInverse(sym, r->move_logits, tmp, kBoardLen);
  std::memcpy(r->move_logits, tmp, sizeof tmp);
  ApplyInverse(sym, r->move_probs, tmp, kBoardLen);
  std::memcpy(r->move_probs, tmp, sizeof tmp);
  ApplyInverse(sym, r->opt_move_probs, tmp, kBoardLen);
  std::memcpy(r->opt_move_probs, tmp, sizeof tmp);
}

}  // namespace p3
