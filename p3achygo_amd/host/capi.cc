// capi.cc — C entry points of libp3host.so used by the tests (ctypes) to drive the rules
// engine, RNG, symmetry and feature extraction.  Not part of the engine ABI.
#include <cstring>

#include "board.h"
#include "features.h"
#include "rng.h"
#include "symmetry.h"

using namespace p3;

extern "C" {

// ---- PRng / Probability ----------------------------------------------------------------
void* p3host_prng_new(uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3) { return new PRng(s0, s1, s2, s3); }
void p3host_prng_free(void* p) { delete (PRng*)p; }
uint32_t p3host_prng_next(void* p) { return ((PRng*)p)->next(); }
uint64_t p3host_prng_next64(void* p) { return ((PRng*)p)->next64(); }
void p3host_prng_next128(void* p, uint64_t* hi, uint64_t* lo) { ((PRng*)p)->next128(*hi, *lo); }
int p3host_rand_range(void* p, int lo, int hi) { return RandRange(*(PRng*)p, lo, hi); }
void* p3host_prob_new(uint64_t seed) { return new Probability(seed); }
void p3host_prob_free(void* p) { delete (Probability*)p; }
float p3host_prob_uniform(void* p) { return ((Probability*)p)->Uniform(); }
float p3host_prob_gumbel(void* p) { return ((Probability*)p)->GumbelSample(); }

// ---- symmetry --------------------------------------------------------------------------
int p3host_transform_index(int sym, int idx, int n) { return TransformIndex((Symmetry)sym, idx, n); }
int p3host_transform_inv(int sym, int idx, int n) { return TransformInv((Symmetry)sym, idx, n); }

// ---- board -----------------------------------------------------------------------------
void* p3host_board_new(float komi, int prohibit_pass_alive) { return new Board(komi, prohibit_pass_alive != 0); }
void* p3host_board_handicap(int handicap, float komi) { return new Board(handicap, komi); }
void* p3host_board_copy(void* b) { return new Board(*(Board*)b); }
void p3host_board_free(void* b) { delete (Board*)b; }
int p3host_board_play(void* b, int i, int j, int color) { return (int)((Board*)b)->PlayMove(Loc{i, j}, (Color)color); }
int p3host_board_dry(void* b, int i, int j, int color) { return (int)((Board*)b)->PlayMoveDry(Loc{i, j}, (Color)color); }
int p3host_board_pass(void* b, int color) { return (int)((Board*)b)->Pass((Color)color); }
void p3host_board_place_raw(void* b, int i, int j, int color) { ((Board*)b)->PlaceRaw(Loc{i, j}, (Color)color); }
int p3host_board_is_game_over(void* b) { return ((Board*)b)->IsGameOver(); }
int p3host_board_is_all_pass_alive(void* b) { return ((Board*)b)->IsAllPassAlive(); }
int p3host_board_move_count(void* b) { return ((Board*)b)->move_count(); }
uint64_t p3host_board_hash(void* b) { return ((Board*)b)->hash(); }
void p3host_board_position(void* b, int8_t* out) { std::memcpy(out, ((Board*)b)->position().data(), kNumLocs); }
void p3host_board_pass_alive(void* b, int8_t* out) { std::memcpy(out, ((Board*)b)->pass_alive().data(), kNumLocs); }
void p3host_board_calc_pass_alive(void* b, int color) {
  if (color == 0) ((Board*)b)->CalculatePassAliveRegions();
  else ((Board*)b)->CalculatePassAliveRegionForColor((Color)color);
}
void p3host_board_scores(void* b, float* black, float* white, int8_t* ownership) {
  Scores s = ((Board*)b)->GetScores();
  *black = s.black_score;
  *white = s.white_score;
  std::memcpy(ownership, s.ownership.data(), kNumLocs);
}
void p3host_board_liberties_plane(void* b, int liberties, int8_t* out) {
  Grid g = ((Board*)b)->GetStonesWithLiberties(liberties);
  std::memcpy(out, g.data(), kNumLocs);
}
void p3host_board_laddered(void* b, int8_t* out) {
  Grid g = ((Board*)b)->GetLadderedStones();
  std::memcpy(out, g.data(), kNumLocs);
}
int p3host_board_group_liberties(void* b, int i, int j) { return ((Board*)b)->LibertiesAt(i * kBoardLen + j); }
int p3host_board_group_id(void* b, int i, int j) { return ((Board*)b)->GroupIdAt(i * kBoardLen + j); }

// ---- game + features -------------------------------------------------------------------
void* p3host_game_new(float komi) { return new Game(komi, true); }
void p3host_game_free(void* g) { delete (Game*)g; }
int p3host_game_play(void* g, int i, int j, int color) { return ((Game*)g)->PlayMove(Loc{i, j}, (Color)color); }
int p3host_game_num_moves(void* g) { return ((Game*)g)->num_moves(); }
void p3host_game_features(void* g, int color, int sym, p3hip_features* out) {
  FillFeatures(*(Game*)g, (Color)color, (Symmetry)sym, out);
}
void p3host_unapply_symmetry(int sym, p3hip_result* r) { UnapplySymmetry((Symmetry)sym, r); }

}  // extern "C"
