// eval_match.h — model-vs-model evaluation games (cc/eval/eval.cc:103-518, the parallel
// mcts::Search path: PlayerSearchConfig::num_threads_per_game > 1, eval.cc:99-101,254-261).
//
// Each game keeps one search tree per player (eval.cc:130-147); the side to move runs a
// BatchSearch on its own tree with its own network, both trees follow the played move
// (eval.cc:318-352), a player resigns when its root outcome estimate drops below
// kResignThreshold (eval.cc:28,277-282).  `cur` plays Black in even-numbered games
// (eval.cc:110).  All games of a match advance together: every scheduler step gathers the
// leaves that the games whose side to move uses engine e want evaluated, runs engine e once,
// and hands the results back — no thread per game, no slot signalling.
#pragma once
#include <memory>
#include <vector>

#include "features.h"
#include "parallel_search.h"
#include "symmetry.h"

namespace p3 {

constexpr float kResignThreshold = -0.92f;   // eval.cc:28

struct EvalPlayerConfig {   // PlayerSearchConfig (player_config.h:20-108)
  int n = 128;                   // visit budget per move
  // > 1: parallel mcts::Search with this many leaves per round; 1: the legacy path,
  // GumbelEvaluator::SearchRootPuct with LCB move choice (eval.cc:99-101,262-281, use_puct /
  // use_lcb defaults of player_config.h:29,44)
  int num_threads_per_game = 8;
  float c_puct = 1.0f, c_puct_visit_scaling = 0.45f, root_fpu = 0.2f;
  bool var_scale_cpuct = false;
  int var_scale_prior_visits = 0;
  // parallel-search knobs, defaults of player_config.h:76-108
  SearchMode search_mode = SearchMode::kConcurrent;
  QFn q_fn = QFn::kVirtualLossSoft;
  NFn n_fn = NFn::kVirtualVisit;
  float vl_delta = -1.5f;
  CollisionPolicy collision_policy = CollisionPolicy::kAbort;
  CollisionDetector collision_detector = CollisionDetector::kNoOp;
  int max_collision_retries = 4;
};

class EvalGame {
 public:
  EvalGame(int game_id, const EvalPlayerConfig& cur, const EvalPlayerConfig& cand, int max_moves, uint64_t seed)
      : id_(game_id), cur_is_black_(game_id % 2 == 0), max_moves_(max_moves), prob_(seed), game_(7.5f, true) {
    cfg_[0] = cur_is_black_ ? cur : cand;   // index 0 = black's player, 1 = white's
    cfg_[1] = cur_is_black_ ? cand : cur;
    for (int s = 0; s < 2; ++s) tree_[s] = pool_[s].Create();
    BeginSearch();
  }
  bool done() const { return done_; }
  // 0 = cur's engine, 1 = cand's engine — for the side to move
  int active_engine() const { return (color_ == kBlack) == cur_is_black_ ? 0 : 1; }
  // Advances the game until it wants evaluations from active_engine(); returns their number
  // (0 when the game has finished).
  int Step() {
    for (;;) {
      if (done_) return 0;
      if (parallel_) {
        const int n = search_.Step();
        if (n > 0) return n;
      } else if (puct_.Step() == GumbelSearch::Status::kNeedEval) {
        return 1;
      }
      FinishMove();
    }
  }
  void FillEval(int i, p3hip_features* f) {
    sym_[i] = RandomSymmetry(prob_.prng());
    if (parallel_) FillFeatures(search_.eval_pos(i), search_.eval_color(i), sym_[i], f);
    else FillFeatures(*puct_.eval_game(), puct_.eval_color(), sym_[i], f);
  }
  void Deliver(int i, p3hip_result& r) {
    UnapplySymmetry(sym_[i], &r);
    if (parallel_) search_.Deliver(i, r);
    else puct_.Resume(r);
  }
  // +1 cur won, -1 cand won, 0 draw
  int cur_result() const {
    const Color w = winner_;
    if (w == kEmpty) return 0;
    return (w == kBlack) == cur_is_black_ ? 1 : -1;
  }
  Color winner() const { return winner_; }
  bool resigned() const { return resigned_; }
  int num_moves() const { return game_.num_moves(); }
  long visits() const { return visits_; }
  long collisions() const { return collisions_; }
  const Game& game() const { return game_; }

 private:
  void BeginSearch() {
    const int side = color_ == kBlack ? 0 : 1;
    ParallelSearchParams p;
    p.batch = cfg_[side].num_threads_per_game;
    p.visit_budget = cfg_[side].n;
    p.puct.c_puct = cfg_[side].c_puct;
    p.puct.c_puct_visit_scaling = cfg_[side].c_puct_visit_scaling;
    p.puct.root_fpu = cfg_[side].root_fpu;
    p.puct.enable_var_scaling = cfg_[side].var_scale_cpuct;
    p.puct.var_scale_prior_visits = cfg_[side].var_scale_prior_visits;
    p.mode = cfg_[side].search_mode;
    p.fns = VirtualFns{cfg_[side].q_fn, cfg_[side].n_fn, cfg_[side].vl_delta};
    p.collision = cfg_[side].collision_policy;
    p.detector = cfg_[side].collision_detector;
    p.max_collision_retries = cfg_[side].max_collision_retries;
    parallel_ = cfg_[side].num_threads_per_game > 1;   // UsesParallelSearch, eval.cc:99-101
    if (parallel_) {
      search_.Begin(&game_, &pool_[side], tree_[side], color_, p);
    } else {
      p.puct.kind = PuctRootSelection::kLcb;
      puct_.BeginPuct(&game_, &pool_[side], tree_[side], color_, cfg_[side].n, p.puct, /*tau=*/1.0f, &prob_);
    }
  }
  void FinishMove() {
    const int side = color_ == kBlack ? 0 : 1;
    Loc move;
    if (parallel_) {
      const ParallelSearchResult& r = search_.result();
      visits_ += r.num_visits;
      collisions_ += r.num_collisions;
      move = r.move;
    } else {
      visits_ += puct_.result().visits;
      move = puct_.result().mcts_move;
    }
    if (VOutcome(tree_[side]) < kResignThreshold) {   // eval.cc:277-282
      resigned_ = true;
      winner_ = Opp(color_);
      done_ = true;
      return;
    }
    game_.PlayMove(move, color_);
    color_ = Opp(color_);
    for (int s = 0; s < 2; ++s) {   // both trees follow the move (eval.cc:318-352)
      TreeNode* next = tree_[s]->child(MoveIdx(move));
      if (!next) next = pool_[s].Create();
      pool_[s].Reap(next);
      tree_[s] = next;
    }
    if (game_.IsGameOver() || game_.num_moves() >= max_moves_) {
      game_.WriteResult();
      winner_ = game_.result().winner;
      done_ = true;
      return;
    }
    BeginSearch();
  }

  int id_;
  bool cur_is_black_;
  int max_moves_;
  Probability prob_;
  Game game_;
  EvalPlayerConfig cfg_[2];
  NodePool pool_[2];
  TreeNode* tree_[2];
  Color color_ = kBlack;
  BatchSearch search_;
  GumbelSearch puct_;
  bool parallel_ = true;
  Symmetry sym_[64];
  bool done_ = false, resigned_ = false;
  Color winner_ = kEmpty;
  long visits_ = 0, collisions_ = 0;
};

}  // namespace p3
