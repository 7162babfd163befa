// capi.cc — C entry points of libp3host.so used by the tests (ctypes) to drive the rules
// engine, RNG, symmetry and feature extraction.  Not part of the engine ABI.
#include <algorithm>
#include <array>
#include <cstring>
#include <vector>

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
// the loops of cc/core/__tests__/probability_test.cc: float sums of n samples (kind 0 uniform, 1 Gaussian,
// 2 Gumbel with non-finite samples counted as 0); out = mean, variance, samples outside [0, 1) (uniform)
void p3host_prob_moments(uint64_t seed, int kind, int n, float* out) {
  Probability p(seed);
  float sum = 0.0f, sum_sq = 0.0f;
  int out_of_range = 0;
  for (int i = 0; i < n; ++i) {
    float x = kind == 0 ? p.Uniform() : kind == 1 ? p.Gaussian() : p.GumbelSample();
    if (kind == 0 && (x < 0.0f || x >= 1.0f)) ++out_of_range;
    if (kind == 2 && !std::isfinite(x)) x = 0.0f;
    sum += x;
    sum_sq += x * x;
  }
  const float mean = sum / n;
  out[0] = mean; out[1] = sum_sq / n - mean * mean; out[2] = (float)out_of_range;
}

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

// ---- scripted search (known-answer test of the Gumbel root search) ------------------------
// Scenario of the reference's (stale) cc/mcts/__tests__/gumbel_test.cc:20-123, driven through
// the real search instead of a hand-built tree: the prior prefers move (0,0) (logit 2) over
// (0,1),(0,2),(0,3) (logit 1), everything else is negligible; every position after Black's
// first move at (0,i) is worth q_i = -0.5 + i/3 to Black.  No Gumbel noise.  With n=8, k=4
// sequential halving must visit the children 1,1,3,3 times and pick (0,3).
#include "search.h"

extern "C" void p3host_softmax(const float* in, float* out, int n) { SoftmaxN(in, out, n); }

// ---- bias cache (bias_cache.h) ------------------------------------------------------------------
extern "C" {
// LocalPattern::FromCurrentPosition of a game handle: returns 0 when the position has no pattern
// (no stone as last move, or no move two plies back); otherwise fills the three 5x5 maps (grid:
// 1 black, -1 white, 2 off board) and sets same[0..2] to whether the grid / atari / ko hashes
// equal those of the game `other` (may be null).
int p3host_test_local_pattern(void* game, void* other, int8_t* grid, int8_t* atari, int8_t* ko, int* same) {
  const auto p = LocalPattern::FromCurrentPosition(Position(*(Game*)game));
  if (!p) return 0;
  std::memcpy(grid, p->grid.data(), 25);
  std::memcpy(atari, p->atari.data(), 25);
  std::memcpy(ko, p->ko.data(), 25);
  if (other && same) {
    const auto q = LocalPattern::FromCurrentPosition(Position(*(Game*)other));
    same[0] = q && q->grid_hash == p->grid_hash;
    same[1] = q && q->atari_hash == p->atari_hash;
    same[2] = q && q->ko_hash == p->ko_hash;
    same[3] = q && q->last_move.loc == p->last_move.loc && q->last_move.color == p->last_move.color &&
              q->two_moves_ago.loc == p->two_moves_ago.loc;
  }
  return 1;
}
// Scripted BiasCache arithmetic on two nodes that share one entry.  Node k (k = 0, 1) has own
// estimate init[k], n[k] visits and three children with visits cv[k][0..2] and values cvv[k][0..2].
// Steps: update node 0, update node 1, give node 0's first child `extra` more visits and update it
// again, release node 1 (its destructor), read-only fetch through node 0.  out[0..4] = the five
// returned biases; out[5], out[6] = the entry (error, weight) at the end; out[7] = node 0's v after
// RecomputeNodeStats with the last bias; out[8] = entries left after node 0 is released and the
// cache pruned.
void p3host_test_bias_cache_math(float alpha, float lambda, const float* init, const int* cv, const float* cvv,
                                 int extra, float* out) {
  BiasCache cache(alpha, lambda);
  NodePool pool;
  TreeNode* nodes[2] = {pool.Create(), pool.Create()};
  Game g(7.5f, true);
  g.PlayMove(Loc{3, 3}, kBlack);
  g.PlayMove(Loc{15, 15}, kWhite);
  const Position pos(g);
  for (int k = 0; k < 2; ++k) {
    TreeNode* nd = nodes[k];
    nd->evaluated = true;
    nd->init_util_est = init[k];
    nd->n = 1;
    for (int c = 0; c < 3; ++c) {
      TreeNode* ch = pool.Create();
      ch->v = cvv[k * 3 + c];
      ch->n = cv[k * 3 + c];
      nd->children.push_back(ChildEdge{(int16_t)c, cv[k * 3 + c], ch});
      nd->n += cv[k * 3 + c];
    }
    AssignBiasCacheEntry(&cache, pos, nd);   // same position: both get the same entry
  }
  out[0] = UpdateAndFetchObsBias(&cache, nodes[0]);
  out[1] = UpdateAndFetchObsBias(&cache, nodes[1]);
  nodes[0]->children[0].visits += extra;
  nodes[0]->n += extra;
  out[2] = UpdateAndFetchObsBias(&cache, nodes[0]);
  RecomputeNodeStats(nodes[0], out[2]);
  out[7] = nodes[0]->v;
  nodes[1]->bias.Release();
  out[3] = cache.Fetch(nodes[0]->bias);
  out[4] = cache.Fetch(nodes[1]->bias);   // no entry any more: 0
  out[5] = nodes[0]->bias.bias_cache_entry->err;
  out[6] = nodes[0]->bias.bias_cache_entry->weight;
  nodes[0]->bias.Release();
  cache.PruneUnused();
  out[8] = (float)cache.size();
}
}

// k > 0: Gumbel root search (n, k); k == 0: SearchRootPuct with n playouts (self-play's
// fast-move parameters).
static bool g_scripted_early_stopping = false;
extern "C" void p3host_test_scripted_early_stopping(int on) { g_scripted_early_stopping = on != 0; }
extern "C" int p3host_test_scripted_search(int n, int k, int* child_visits, float* child_q, int* nn_move,
                                           int* mcts_move, int* root_n) {
  Game game(7.5f, true);
  NodePool pool;
  TreeNode* root = pool.Create();
  Probability prob(0);
  GumbelParams p;
  p.n = n; p.k = k; p.noise_scaling = 0.0f; p.tau = 0.0f;
  p.early_stopping_enabled = g_scripted_early_stopping;
  GumbelSearch search;
  if (k > 0) {
    search.Begin(&game, &pool, root, kBlack, p, &prob);
  } else {
    PuctParams pp;
    pp.c_puct = 1.05f; pp.c_puct_visit_scaling = 0.28f; pp.root_fpu = 0.05f;
    search.BeginPuct(&game, &pool, root, kBlack, n, pp, 0.5f, &prob);
  }
  auto eval = [&](const Position& pos, Color to_move, p3hip_result& r) {
    float logits[kNumMoves];
    for (int i = 0; i < kNumMoves; ++i) logits[i] = -30.0f;
    logits[0] = 2.0f;
    logits[1] = logits[2] = logits[3] = 1.0f;
    std::memcpy(r.move_logits, logits, sizeof logits);
    SoftmaxN(logits, r.move_probs, kNumMoves);
    std::memcpy(r.opt_move_probs, r.move_probs, sizeof r.move_probs);
    int first = -1;   // Black's first move of the game = first non-noop entry of the move window
    for (int i = 0; i < 5 && first < 0; ++i)
      if (pos.last[i].color == kBlack && pos.last[i].loc.i == 0 && pos.last[i].loc.j < 4) first = pos.last[i].loc.j;
    float q_black = first < 0 ? 0.0f : -0.5f + first / 3.0f;
    float q = to_move == kBlack ? q_black : -q_black;
    r.value_probs[1] = 0.5f * (1 + q);
    r.value_probs[0] = 0.5f * (1 - q);
    for (int i = 0; i < P3HIP_NUM_SCORE_LOGITS; ++i) r.score_probs[i] = 0.0f;
    r.score_probs[399] = r.score_probs[400] = 0.5f;   // E[score] = 0
    r.err2_outcome = 0.0f;
  };
  int guard = 0;
  while (search.Step() == GumbelSearch::Status::kNeedEval) {
    p3hip_result r;
    eval(*search.eval_game(), search.eval_color(), r);
    search.Resume(r);
    if (++guard > 10000) return 1;
  }
  for (int i = 0; i < 4; ++i) {
    child_visits[i] = root->child_visits(i);
    child_q[i] = Q(root, i);
  }
  *nn_move = MoveIdx(search.result().nn_move);
  *mcts_move = MoveIdx(search.result().mcts_move);
  *root_n = root->n;
  return (int)search.result().visits << 8;
}

// ---- TF recorder (tests mirror cc/recorder/__tests__/{tf_recorder,sel_mult}_test.cc) --------
#include "tf_recorder.h"

extern "C" {
void* p3host_tfrec_new(const char* dir, int gen, const char* worker) { return new TfRecorder(dir, gen, worker); }
void p3host_tfrec_free(void* r) { delete (TfRecorder*)r; }
int p3host_tfrec_flush(void* r) { return ((TfRecorder*)r)->Flush(); }
const char* p3host_tfrec_last_chunk(void* r) { return ((TfRecorder*)r)->last_chunk().c_str(); }
// One finished game given as a move list (encoding of p3host_sgf_from_moves) plus per-move
// record columns; null columns take the defaults of the reference tests' SimpleRecord.
// stats rows: sampled_raw_policy, nn_q, mcts_q, nn_mcts_diff, v_outcome_stddev, prior_entropy,
// nn_uncertainty, kld, pre_kld, sel_mult_modifier, sel_mult_modifier_weight, visit_count,
// visit_count_pre.
int p3host_tfrec_record(void* r, const int* moves, int n, float komi, const float* pi, const uint8_t* trainable,
                        const float* root_q, const float* root_score, const float* kld,
                        const uint32_t* value_dist, const float* stats) {
  Game g(komi, true);
  for (int i = 0; i < n; ++i) {
    Color c = moves[i] > 0 ? kBlack : kWhite;
    int idx = (moves[i] > 0 ? moves[i] : -moves[i]) - 1;
    if (!g.PlayMove(MoveLoc(idx), c)) return -1;
  }
  g.WriteResult();
  std::vector<MoveSearchRecord> infos(n);
  for (int i = 0; i < n; ++i) {
    MoveSearchRecord& m = infos[i];
    for (int a = 0; a < kNumMoves; ++a) m.mcts_pi[a] = pi ? pi[i * kNumMoves + a] : 1.0f / kNumMoves;
    m.move_trainable = trainable ? trainable[i] : 1;
    m.root_q_outcome = root_q ? root_q[i] : 0.5f;
    m.root_score = root_score ? root_score[i] : 0.0f;
    m.kld = kld ? kld[i] : 0.0f;
    if (value_dist) std::memcpy(m.mcts_value_dist, value_dist + i * kNumVBuckets, sizeof m.mcts_value_dist);
    if (stats) {
      const float* s = stats + i * 13;
      m.move_stats = MoveSearchStats{s[0] != 0.0f, s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], s[9], s[10], s[11], s[12]};
    }
  }
  ((TfRecorder*)r)->RecordGame(Board(komi, true), g, std::move(infos));
  return 0;
}
uint32_t p3host_crc32c(const void* p, size_t n) { return Crc32c(p, n); }
void p3host_ladder_stats(long* out) { LadderStats(out); }
void p3host_set_ladder_budget(long nodes) { SetLadderNodeBudget(nodes); }
long p3host_ladder_budget() { return LadderNodeBudget(); }
}

// ---- ladder read-out: exact mode vs an independent naive read-out vs the budgeted mode --------
// The naive reader follows cc/game/board.cc:692-899 through the PUBLIC board API only (flood
// fills over position(), boards copied by value at every ply), sharing nothing with
// board.cc's LadderSolver (which walks the circular group lists and the liberty counters).
namespace {
struct NaiveLadder {
  static void Flood(const Board& b, int p, std::vector<int>& stones, std::vector<int>& libs) {
    const Grid& g = b.position();
    const Color c = g[p];
    std::array<uint8_t, kNumLocs> seen{};
    stones.assign(1, p);
    libs.clear();
    seen[p] = 1;
    for (size_t h = 0; h < stones.size(); ++h) {
      const int s = stones[h], i = s / kBoardLen, j = s % kBoardLen;
      const int nb[4] = {i > 0 ? s - kBoardLen : -1, i < kBoardLen - 1 ? s + kBoardLen : -1, j > 0 ? s - 1 : -1,
                         j < kBoardLen - 1 ? s + 1 : -1};
      for (int q : nb) {
        if (q < 0 || seen[q]) continue;
        if (g[q] == kEmpty) { seen[q] = 1; libs.push_back(q); }
        else if (g[q] == c) { seen[q] = 1; stones.push_back(q); }
      }
    }
  }
  static bool Solve(Board board, Color g_color, Color to_move, int root, int last_move, int depth) {
    if (depth > 300) return false;
    if (!MoveOk(board.PlayMove(AsLoc(last_move), Opp(to_move)))) return g_color != to_move;
    if (board.position()[root] == kEmpty) return true;
    std::vector<int> stones, libs;
    Flood(board, root, stones, libs);
    if (g_color != to_move) {   // attacker: needs exactly two liberties to keep chasing
      if (libs.size() > 2) return false;
      if (libs.size() <= 1) return true;
      return Solve(board, g_color, Opp(to_move), root, libs[0], depth + 1) ||
             Solve(board, g_color, Opp(to_move), root, libs[1], depth + 1);
    }
    if (libs.size() > 1) return false;   // defender already out
    if (!Solve(board, g_color, Opp(to_move), root, libs[0], depth + 1)) return false;   // extend
    // or capture an adjacent attacker group that is in atari
    std::vector<int> done;   // one stone of every enemy group handled
    for (int s : stones) {
      const int i = s / kBoardLen, j = s % kBoardLen;
      const int nb[4] = {i > 0 ? s - kBoardLen : -1, i < kBoardLen - 1 ? s + kBoardLen : -1, j > 0 ? s - 1 : -1,
                         j < kBoardLen - 1 ? s + 1 : -1};
      for (int q : nb) {
        if (q < 0 || board.position()[q] != Opp(g_color)) continue;
        std::vector<int> es, el;
        Flood(board, q, es, el);
        if (el.size() != 1) continue;
        bool dup = false;
        for (int d : done) dup |= std::find(es.begin(), es.end(), d) != es.end();
        if (dup) continue;
        done.push_back(q);
        if (!Solve(board, g_color, Opp(to_move), root, el[0], depth + 1)) return false;
      }
    }
    return true;
  }
  static Grid Laddered(const Board& b) {
    Grid out{};
    std::array<uint8_t, kNumLocs> handled{};
    for (int p = 0; p < kNumLocs; ++p) {
      if (b.position()[p] == kEmpty || handled[p]) continue;
      std::vector<int> stones, libs;
      Flood(b, p, stones, libs);
      for (int s : stones) handled[s] = 1;
      if (libs.size() != 1) continue;
      int empty_nb = 0;   // IsLaddered's pre-check: the liberty has three or more empty neighbours
      {
        const int s = libs[0], i = s / kBoardLen, j = s % kBoardLen;
        const int nb[4] = {i > 0 ? s - kBoardLen : -1, i < kBoardLen - 1 ? s + kBoardLen : -1, j > 0 ? s - 1 : -1,
                           j < kBoardLen - 1 ? s + 1 : -1};
        for (int q : nb) empty_nb += q >= 0 && b.position()[q] == kEmpty;
      }
      if (empty_nb >= 3) continue;
      const Color c = b.position()[p];
      if (Solve(b, c, Opp(c), p, libs[0], 0))
        for (int s : stones) out[s] = c;
    }
    return out;
  }
};
}  // namespace

extern "C" {
// Uniformly random legal playouts (seeded), one position every `stride` moves from move
// `first_move` on, `n_positions` in all.  out[0] = positions where the exact mode (no budget)
// and the naive reader disagree on the ladder plane, out[1] = positions where the budgeted mode
// (`budget` nodes) differs from the exact mode, out[2] = read-outs that exhausted the budget,
// out[3] = largest node count of one exact read-out, out[4] = positions with a laddered stone,
// out[5] = total exact read-outs.  max_exact_nodes > 0 skips the (exponential) naive read of
// positions whose exact read-out took more nodes than that; out[6] counts those.
void p3host_test_ladder_modes(int n_positions, uint64_t seed, long budget, int first_move, int stride,
                              long max_exact_nodes, long* out) {
  for (int i = 0; i < 7; ++i) out[i] = 0;
  const long saved = LadderNodeBudget();
  PRng prng(seed, seed ^ 0x9e3779b97f4a7c15ull, seed * 3 + 1, seed * 7 + 5);
  int done = 0;
  while (done < n_positions) {
    Game game(7.5f, true);
    Color c = kBlack;
    int passes = 0;
    for (int mv = 0; mv < 420 && done < n_positions && passes < 2; ++mv) {
      Loc pick = kPassLoc;
      for (int tries = 0; tries < 30; ++tries) {
        const int idx = RandRange(prng, 0, kNumLocs);
        if (game.IsValidMove(AsLoc(idx), c)) { pick = AsLoc(idx); break; }
      }
      passes = pick == kPassLoc ? passes + 1 : 0;
      game.PlayMove(pick, c);
      c = Opp(c);
      if (mv < first_move || (mv - first_move) % stride != 0) continue;
      long s0[4], s1[4];
      SetLadderNodeBudget(0);
      LadderStats(s0);
      const Grid exact = game.board().GetLadderedStones();
      LadderStats(s1);
      out[5] += s1[0] - s0[0];
      const long nodes = s1[1] - s0[1];
      if (nodes > out[3]) out[3] = nodes;
      bool any = false;
      for (Color v : exact) any |= v != kEmpty;
      out[4] += any;
      if (max_exact_nodes > 0 && nodes > max_exact_nodes) ++out[6];
      else if (NaiveLadder::Laddered(game.board()) != exact) ++out[0];
      SetLadderNodeBudget(budget);
      LadderStats(s0);
      const Grid fast = game.board().GetLadderedStones();
      LadderStats(s1);
      out[2] += s1[3] - s0[3];
      out[1] += fast != exact;
      ++done;
    }
  }
  SetLadderNodeBudget(saved);
}
}

// ---- fork manager / init-state / move-selection tests ----------------------------------------
#include "selfplay_policy.h"

extern "C" {
// A book start (BookInitState): out[0] stones on the board, out[1] colour to move, out[2] kind,
// out[3] move_num, out[4] number of non-noop last moves; last5 = encoded last moves (i*19+j, noop -20).
void p3host_test_book_state(uint64_t seed, int* out, int* last5) {
  Probability prob(seed);
  InitState s0;
  s0.board = Board(7.5f, true);
  const InitState s = BookInitState(prob, s0);
  int stones = 0, nm = 0;
  for (Color c : s.board.position()) stones += c != kEmpty;
  for (int i = 0; i < 5; ++i) {
    last5[i] = s.last_moves[i].loc == kNoopLoc ? -20 : s.last_moves[i].loc.i * 19 + s.last_moves[i].loc.j;
    nm += s.last_moves[i].loc != kNoopLoc;
  }
  out[0] = stones; out[1] = s.color_to_move; out[2] = (int)s.kind; out[3] = s.move_num; out[4] = nm;
}

// Plays `n_moves` scripted legal moves (first empty point scanning from a seed-dependent
// offset), calling MaybeFork before each; network evaluations are answered by a scripted
// result whose policy is uniform over the board, win probability `p_win` and a one-hot score
// distribution at margin `score` (bin index score + 400, so E[score] = score + 0.5).
// kind: index into (early, late, t1, t2, random, regret, uniform) given probability 1.
// Outputs the InitState that reached the buffer (if any): out[0]=#states added,
// [1]=move_num, [2]=colour to move, [3]=first_move_behavior, [4]=#stones on its board,
// [5]=#evaluations requested, [6]=fork move number, [7]=kind; komi_out = its komi; last5 =
// encoded last moves (i*19+j, pass 361, noop -20).
int p3host_test_fork(int kind, int n_moves, uint64_t seed, float p_win, int score, int* out, float* komi_out,
                     int* last5, int8_t* board_out) {
  ForkParams fp;
  fp.early = fp.late = fp.t1 = fp.t2 = fp.random = fp.regret = fp.uniform = 0.0f;
  float* slots[7] = {&fp.early, &fp.late, &fp.t1, &fp.t2, &fp.random, &fp.regret, &fp.uniform};
  *slots[kind] = 1.0f;
  ReuseBuffer buf(seed);
  Probability prob(seed);
  ForkManager fm(fp, &buf, prob, false);
  Game g(7.5f, true);
  Color c = kBlack;
  p3hip_result r;
  std::memset(&r, 0, sizeof r);
  for (int a = 0; a < kNumMoves; ++a) { r.move_logits[a] = 0; r.move_probs[a] = 1.0f / kNumMoves; r.opt_move_probs[a] = 1.0f / kNumMoves; }
  r.value_probs[0] = 1 - p_win; r.value_probs[1] = p_win;
  r.score_probs[score + 400] = 1.0f;
  int evals = 0;
  for (int m = 0; m < n_moves && !g.IsGameOver(); ++m) {
    Loc mv = kPassLoc;
    for (int t = 0; t < kNumLocs; ++t) {
      int idx = (int)((seed * 31 + m * 7 + t * 13) % kNumLocs);
      if (g.IsValidMove(AsLoc(idx), c)) { mv = AsLoc(idx); break; }
    }
    ForkManager::MoveData md{&g.board(), c, mv, 0.1f, 0.2f, 3.0f, true};
    if (fm.MaybeFork(g, md, prob)) {
      Position pos;
      Color ec;
      while (fm.NextEval(&pos, &ec)) { ++evals; fm.Deliver(r); }
    }
    g.PlayMove(mv, c);
    c = Opp(c);
  }
  g.WriteResult();
  fm.FinalizeGame(g, prob);
  out[0] = (int)buf.added();
  out[5] = evals;
  out[6] = fm.fork_move_num();
  out[7] = (int)fm.kind();
  if (auto s = buf.Get()) {
    out[1] = s->move_num;
    out[2] = s->color_to_move;
    out[3] = (int)s->first_move_behavior;
    int stones = 0;
    for (int i = 0; i < kNumLocs; ++i) { stones += s->board.at(i) != kEmpty; if (board_out) board_out[i] = s->board.at(i); }
    out[4] = stones;
    *komi_out = s->board.komi();
    for (int i = 0; i < 5; ++i) last5[i] = TfRecorder::EncodeLoc16(s->last_moves[i].loc);
  }
  return 0;
}

// GetInitState statistics over `n` draws with an empty buffer: counts of handicap games and
// the histogram of komi*2 (index komi*2, 0..63) for the non-handicap ones.
void p3host_test_init_states(uint64_t seed, int n, int* handicap_count, int* komi2_hist, int* handicap_stones) {
  Probability prob(seed);
  ReuseBuffer buf(seed);
  *handicap_count = 0;
  for (int i = 0; i < 64; ++i) komi2_hist[i] = 0;
  for (int i = 0; i < 3; ++i) handicap_stones[i] = 0;
  for (int i = 0; i < n; ++i) {
    InitState s = GetInitState(prob, &buf, 0.5f);
    if (s.kind == InitState::Kind::kHandicap) {
      ++*handicap_count;
      int stones = 0;
      for (int p = 0; p < kNumLocs; ++p) stones += s.board.at(p) != kEmpty;
      if (stones >= 2 && stones <= 4 && s.color_to_move == kWhite && s.board.komi() == (stones - 2) * 14 + 20.5f) ++handicap_stones[stones - 2];
    } else {
      int k2 = (int)std::lround(s.board.komi() * 2);
      if (k2 >= 0 && k2 < 64) ++komi2_hist[k2];
    }
  }
}

// MoveSelManager::Compute with the default calibration and the self-play flags
// (kNnMctsBonus | kKldPenalty); out = modifier, bonus, penalty, q_adjust, kld_penalty, nn_mcts_bonus
// the same with the thresholds of a calibration file (ParseCalibrationFile); counts[0..4] = entries parsed per field
void p3host_test_move_sel_file(const char* path, int n_pre, float std_dev, float pre_kld, float nn_mcts_diff, float q,
                               float scale, float* out, int* counts) {
  SelMultCalibration cal = ParseCalibrationFile(path ? path : "");
  counts[0] = (int)cal.v_outcome_stddev.size(); counts[1] = (int)cal.v_outcome_stddev_adj.size();
  counts[2] = (int)cal.pre_kld.size(); counts[3] = (int)cal.nn_mcts_diff.size(); counts[4] = (int)cal.expected_std_by_n.size();
  MoveSelManager m(kNnMctsBonus | kKldPenalty, cal);
  MoveSelResult r = m.Compute(n_pre, std_dev, pre_kld, nn_mcts_diff, q, scale);
  out[0] = r.modifier; out[1] = r.sel_bonus; out[2] = r.sel_penalty; out[3] = r.sel_q_adjust;
  out[4] = r.sel_kld_penalty; out[5] = r.sel_nn_mcts_bonus;
}
// any flag set, thresholds from a calibration file ("" = none); out = the twelve MoveSelResult fields in order
void p3host_test_move_sel_ex(unsigned flags, const char* path, int n_pre, float std_dev, float pre_kld, float nn_mcts_diff,
                             float q, float scale, float* out) {
  MoveSelManager m(flags, ParseCalibrationFile(path ? path : ""));
  const MoveSelResult r = m.Compute(n_pre, std_dev, pre_kld, nn_mcts_diff, q, scale);
  const float v[12] = {r.modifier, r.modifier_unscaled, r.sel_bonus, r.sel_penalty, r.sel_std_bonus, r.sel_std_penalty,
                       r.sel_kld_bonus, r.sel_kld_penalty, r.sel_nn_mcts_bonus, r.sel_q_adjust, r.std_adj, r.std_adj_att};
  std::memcpy(out, v, sizeof v);
}
void p3host_test_move_sel(int n_pre, float std_dev, float pre_kld, float nn_mcts_diff, float q, float scale, float* out) {
  SelMultCalibration cal;
  MoveSelManager m(kNnMctsBonus | kKldPenalty, cal);
  MoveSelResult r = m.Compute(n_pre, std_dev, pre_kld, nn_mcts_diff, q, scale);
  out[0] = r.modifier; out[1] = r.sel_bonus; out[2] = r.sel_penalty; out[3] = r.sel_q_adjust;
  out[4] = r.sel_kld_penalty; out[5] = r.sel_nn_mcts_bonus;
}
}

// ---- parallel (batch) search tests: cc/mcts/__tests__/search_test.cc:130-222 -------------------
#include "parallel_search.h"

extern "C" {
float p3host_t_quantile(int dof) { return CachedTQuantile(dof); }

// NullEngine of the reference test (uniform policy, value 0.5/0.5, uniform score bins) and its
// MakeEvaluatedRoot; runs one BatchSearch from the empty board.
// out: [0] visits [1] aborted [2] collisions [3] rounds [4] root n [5] move index (-1 noop)
//      [6] sum of root child visits [7] nodes whose n != 1 + sum(child visits) (must be 0)
//      [8] nodes left with n_in_flight != 0 (must be 0) [9] evaluations requested
int p3host_test_batch_search_ex(int batch, int budget, int mode, int q_fn, int n_fn, int collision, int detector,
                                int* out);
int p3host_test_batch_search(int batch, int budget, int* out) {
  return p3host_test_batch_search_ex(batch, budget, /*batch mode*/ 1, 0, 0, 0, 0, out);
}
// mode: 0 concurrent rounds, 1 batch; q_fn 0/1/2 identity / virtual loss / soft; n_fn 0/1 identity /
// virtual visit; collision 0/1/2 abort / retry / smart retry; detector 0..3 noop / n-in-flight / level / product
int p3host_test_batch_search_ex(int batch, int budget, int mode, int q_fn, int n_fn, int collision, int detector,
                                int* out) {
  Game game(7.5f, true);
  NodePool pool;
  TreeNode* root = pool.Create();
  root->evaluated = true;
  root->color_to_move = kBlack;
  root->n = 1;
  for (int a = 0; a < kNumMoves; ++a) { root->move_probs[a] = 1.0f / kNumMoves; root->move_logits[a] = 0; root->opt_probs[a] = 1.0f / kNumMoves; }
  p3hip_result r;
  std::memset(&r, 0, sizeof r);
  for (int a = 0; a < kNumMoves; ++a) { r.move_probs[a] = 1.0f / kNumMoves; r.opt_move_probs[a] = 1.0f / kNumMoves; }
  r.value_probs[0] = r.value_probs[1] = 0.5f;
  for (int i = 0; i < P3HIP_NUM_SCORE_LOGITS; ++i) r.score_probs[i] = 1.0f / P3HIP_NUM_SCORE_LOGITS;
  ParallelSearchParams p;
  p.batch = batch;
  p.visit_budget = budget;
  p.mode = (SearchMode)mode;
  p.fns = VirtualFns{(QFn)q_fn, (NFn)n_fn, -1.5f};
  p.collision = (CollisionPolicy)collision;
  p.detector = (CollisionDetector)detector;
  BatchSearch s;
  s.Begin(&game, &pool, root, kBlack, p);
  int evals = 0;
  for (int n; (n = s.Step()) > 0;) {
    for (int i = 0; i < n; ++i) s.Deliver(i, r);
    evals += n;
  }
  const ParallelSearchResult& res = s.result();
  out[0] = res.num_visits; out[1] = res.num_aborted; out[2] = res.num_collisions; out[3] = res.rounds;
  out[4] = root->n;
  out[5] = res.move == kNoopLoc ? -1 : MoveIdx(res.move);
  int sum = 0;
  for (const ChildEdge& e : root->children) sum += e.visits;
  out[6] = sum;
  int bad_n = 0, bad_if = 0;
  std::vector<TreeNode*> stack{root};
  while (!stack.empty()) {
    TreeNode* nd = stack.back();
    stack.pop_back();
    int cs = 0;
    for (const ChildEdge& e : nd->children) { cs += e.visits; if (e.node) stack.push_back(e.node); }
    if (nd->n != 1 + cs && !(nd->is_terminal && cs == 0)) ++bad_n;
    if (nd->n_in_flight != 0) ++bad_if;
  }
  out[7] = bad_n; out[8] = bad_if; out[9] = evals;
  return 0;
}
}
