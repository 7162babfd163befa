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
#include <cstdio>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "features.h"
#include "parallel_search.h"
#include "symmetry.h"

namespace p3 {

constexpr float kResignThreshold = -0.92f;   // eval.cc:28

// core::RelativeElo (cc/core/elo.h:8-12) and the match summary of eval/main.cc:133-135,459-479.
inline float RelativeElo(float winrate) { return 400 * std::log10(winrate / (1.0f - winrate)); }
inline float ConfidenceDelta(float z_score, float num_sims, float wr) { return z_score * std::sqrt(wr * (1 - wr) / num_sims); }
struct MatchSummary {
  float winrate = 0, c95 = 0, rel_elo = 0, elo_c95 = 0;
};
inline MatchSummary SummarizeMatch(int num_cand_won, int num_games) {
  MatchSummary m;
  m.winrate = (float)num_cand_won / (float)num_games;
  m.rel_elo = RelativeElo(m.winrate);
  m.c95 = ConfidenceDelta(1.96f, (float)num_games, m.winrate);
  m.elo_c95 = RelativeElo(.5f + m.c95);
  return m;
}
// --res_write_path: the relative Elo as "%f" (eval/main.cc:473-477)
inline bool WriteMatchResult(const std::string& path, float rel_elo) {
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return false;
  fprintf(f, "%f", rel_elo);
  fclose(f);
  return true;
}

struct EvalPlayerConfig {   // PlayerSearchConfig (player_config.h:20-108), same fields and defaults
  // Gumbel (the legacy path with use_puct = false)
  int n = 128;                   // visit budget per move
  int k = 8;
  float noise_scaling = 1.0f;
  bool early_stopping_for_gumbel = false;
  // PUCT (GumbelEvaluator::SearchRootPuct and the parallel mcts::Search)
  bool use_puct = true;
  float c_puct = 1.0f, c_puct_visit_scaling = 0.45f;
  bool var_scale_cpuct = false;
  bool use_puct_v = false;
  float c_puct_v_2 = 3.0f;
  float tau = 1.0f;
  std::string puct_root_policy;   // "visit_count" | "lcb" | "visit_count_sample"; empty: from use_lcb
  bool use_lcb = true;
  // score utility
  float score_weight = kDefaultScoreWeight;
  std::string score_utility_mode = "direct";   // | "integral"
  // other
  bool use_mcgs = false;                                          // McgsNodeTable for this player's tree
  bool use_bias_cache = false;
  float bias_cache_alpha = 0.8f, bias_cache_lambda = 0.4f;
  bool enable_m3_bonus = false;
  int var_scale_prior_visits = 0;
  int m3_prior_visits = 20;
  float p_opt_weight = 0.0f;
  float root_fpu = 0.2f;
  // parallel-search knobs (player_config.h:63-108).  num_threads_per_game > 1 (or time_ms > 0): parallel
  // mcts::Search with this many workers / leaves per round; otherwise the legacy path, Gumbel or
  // SearchRootPuct (eval.cc:99-101,229-269).  The match drivers of this repository set it from their
  // own argument (reference default: 1).
  int num_threads_per_game = 8;
  int time_ms = 0;                                                // threaded driver only: time control ("auto" = -1)
  bool enable_pondering = false;                                  // parsed; GTP only in the reference
  uint32_t time_control_flags = 0;                                // parsed; cc/gtp/time_control only
  QFn q_fn = QFn::kVirtualLossSoft;
  NFn n_fn = NFn::kVirtualVisit;
  CollisionPolicy collision_policy = CollisionPolicy::kAbort;
  CollisionDetector collision_detector = CollisionDetector::kNoOp;
  float vl_delta = -1.5f;
  int max_collision_retries = 4;
  SearchMode search_mode = SearchMode::kConcurrent;
  DescentPolicy descent_policy = DescentPolicy::kDeterministic;   // "deterministic" | "bu_uct"
  float max_o_ratio = 1.0f;
};

inline bool UsesParallelSearch(const EvalPlayerConfig& c) { return c.num_threads_per_game > 1 || c.time_ms > 0; }   // eval.cc:99-101

// MakeScoreUtilityParams (player_config.cc:4-12)
inline ScoreUtilityParams MakeScoreUtility(const EvalPlayerConfig& c) {
  return ScoreUtilityParams{c.score_weight, c.score_utility_mode == "integral" ? ScoreUtilityMode::kIntegral : ScoreUtilityMode::kDirect};
}
// The PuctParams of the parallel search (MakeSearchParams, player_config.cc:40-54,100-113) ...
inline PuctParams MakeSearchPuctParams(const EvalPlayerConfig& c) {
  PuctParams p;
  if (c.puct_root_policy == "visit_count") p.kind = PuctRootSelection::kVisitCount;
  else if (c.puct_root_policy == "visit_count_sample") p.kind = PuctRootSelection::kVisitCountSample;
  else if (c.puct_root_policy == "lcb" || c.puct_root_policy.empty())
    p.kind = (c.puct_root_policy == "lcb" || c.use_lcb) ? PuctRootSelection::kLcb : PuctRootSelection::kVisitCount;
  else p.kind = PuctRootSelection::kLcb;
  p.c_puct = c.c_puct; p.c_puct_visit_scaling = c.c_puct_visit_scaling;
  p.c_puct_v_2 = c.c_puct_v_2; p.use_puct_v = c.use_puct_v;
  p.enable_var_scaling = c.var_scale_cpuct; p.var_scale_prior_visits = c.var_scale_prior_visits;
  p.tau = c.tau;
  p.enable_m3_bonus = c.enable_m3_bonus; p.m3_prior_visits = c.m3_prior_visits;
  p.p_opt_weight = c.p_opt_weight; p.root_fpu = c.root_fpu;
  return p;
}
// ... and of the legacy SearchRootPuct call (eval.cc:241-258: no visit scaling or tau override there)
inline PuctParams MakeRootPuctParams(const EvalPlayerConfig& c) {
  PuctParams p;
  p.kind = c.use_lcb ? PuctRootSelection::kLcb : PuctRootSelection::kVisitCount;
  p.c_puct = c.c_puct;
  p.c_puct_v_2 = c.c_puct_v_2; p.use_puct_v = c.use_puct_v;
  p.enable_var_scaling = c.var_scale_cpuct; p.var_scale_prior_visits = c.var_scale_prior_visits;
  p.enable_m3_bonus = c.enable_m3_bonus; p.m3_prior_visits = c.m3_prior_visits;
  p.p_opt_weight = c.p_opt_weight; p.root_fpu = c.root_fpu;
  return p;
}
inline GumbelParams MakeGumbelParams(const EvalPlayerConfig& c) {   // eval.cc:259-267
  GumbelParams g;
  g.n = c.n; g.k = c.k; g.noise_scaling = c.noise_scaling;
  g.early_stopping_enabled = c.early_stopping_for_gumbel;
  return g;
}

// ParsePlayerConfigFile (cc/eval/player_config.h:133-244): "key: value" lines, '#' comments and blank
// lines skipped, every PlayerSearchConfig field by its name, unknown keys ignored, enum-valued fields fall
// back as MakeSearchParams does (player_config.cc:56-95: an unknown q_fn is virtual_loss, the others their
// defaults).  Returns false only when the file cannot be opened or a number does not parse.
inline bool ParsePlayerConfigStream(std::istream& in, EvalPlayerConfig* cfg, std::string* err) {
  auto trim = [](std::string x) {
    const size_t b = x.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return std::string();
    return x.substr(b, x.find_last_not_of(" \t\r\n") - b + 1);
  };
  std::string line;
  try {
    while (std::getline(in, line)) {
      line = trim(line);
      if (line.empty() || line[0] == '#') continue;
      const size_t colon = line.find(':');
      if (colon == std::string::npos) continue;
      const std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
      auto b = [&] { return val == "true" || val == "1"; };
      if (key == "n") cfg->n = std::stoi(val);
      else if (key == "k") cfg->k = std::stoi(val);
      else if (key == "noise_scaling") cfg->noise_scaling = std::stof(val);
      else if (key == "early_stopping_for_gumbel") cfg->early_stopping_for_gumbel = b();
      else if (key == "use_puct") cfg->use_puct = b();
      else if (key == "c_puct") cfg->c_puct = std::stof(val);
      else if (key == "c_puct_visit_scaling") cfg->c_puct_visit_scaling = std::stof(val);
      else if (key == "var_scale_cpuct") cfg->var_scale_cpuct = b();
      else if (key == "use_puct_v") cfg->use_puct_v = b();
      else if (key == "c_puct_v_2") cfg->c_puct_v_2 = std::stof(val);
      else if (key == "tau") cfg->tau = std::stof(val);
      else if (key == "puct_root_policy") cfg->puct_root_policy = val;
      else if (key == "use_lcb") cfg->use_lcb = b();
      else if (key == "score_weight") cfg->score_weight = std::stof(val);
      else if (key == "score_utility_mode") cfg->score_utility_mode = val;
      else if (key == "use_mcgs") cfg->use_mcgs = b();
      else if (key == "use_bias_cache") cfg->use_bias_cache = b();
      else if (key == "bias_cache_alpha") cfg->bias_cache_alpha = std::stof(val);
      else if (key == "bias_cache_lambda") cfg->bias_cache_lambda = std::stof(val);
      else if (key == "enable_m3_bonus") cfg->enable_m3_bonus = b();
      else if (key == "var_scale_prior_visits") cfg->var_scale_prior_visits = std::stoi(val);
      else if (key == "m3_prior_visits") cfg->m3_prior_visits = std::stoi(val);
      else if (key == "p_opt_weight") cfg->p_opt_weight = std::stof(val);
      else if (key == "root_fpu") cfg->root_fpu = std::stof(val);
      else if (key == "num_threads_per_game") cfg->num_threads_per_game = std::stoi(val);
      else if (key == "time_ms") cfg->time_ms = val == "auto" ? -1 : std::stoi(val);
      else if (key == "enable_pondering") cfg->enable_pondering = b();
      else if (key == "time_control_flags") cfg->time_control_flags = val == "all" ? ~0u : (uint32_t)std::stoul(val);
      else if (key == "q_fn")
        cfg->q_fn = val == "identity" ? QFn::kIdentity : val == "virtual_loss_soft" ? QFn::kVirtualLossSoft : QFn::kVirtualLoss;
      else if (key == "n_fn") cfg->n_fn = val == "identity" ? NFn::kIdentity : NFn::kVirtualVisit;
      else if (key == "collision_policy")
        cfg->collision_policy = val == "retry" ? CollisionPolicy::kRetry : val == "smart_retry" ? CollisionPolicy::kSmartRetry : CollisionPolicy::kAbort;
      else if (key == "collision_detector")
        cfg->collision_detector = val == "n_in_flight" ? CollisionDetector::kNInFlight
                                  : val == "level_saturation" ? CollisionDetector::kLevelSaturation
                                  : val == "product" ? CollisionDetector::kProduct : CollisionDetector::kNoOp;
      else if (key == "vl_delta") cfg->vl_delta = std::stof(val);
      else if (key == "max_collision_retries") cfg->max_collision_retries = std::stoi(val);
      else if (key == "search_mode") cfg->search_mode = val == "batch" ? SearchMode::kBatch : SearchMode::kConcurrent;
      else if (key == "descent_policy") cfg->descent_policy = val == "bu_uct" ? DescentPolicy::kBuUct : DescentPolicy::kDeterministic;
      else if (key == "max_o_ratio") cfg->max_o_ratio = std::stof(val);
      // unknown keys are ignored (player_config.h:243)
    }
  } catch (const std::exception& e) {   // the reference lets std::stoi / std::stof throw out of main
    if (err) *err = "bad value in line: " + line;
    return false;
  }
  return true;
}
inline bool ParsePlayerConfig(const std::string& path, EvalPlayerConfig* cfg, std::string* err) {
  std::ifstream in(path);
  if (!in) { if (err) *err = "cannot open " + path; return false; }
  return ParsePlayerConfigStream(in, cfg, err);
}
// The per-player command-line flags of eval/main.cc (--cur_n, --cand_use_puct_v, ...; ApplyCur/
// CandCommandLineFlags, main.cc:146-246) override the config file: here the same "key: value" lines,
// applied after the file.
inline bool ApplyPlayerConfigText(const std::string& text, EvalPlayerConfig* cfg, std::string* err) {
  std::istringstream in(text);
  return ParsePlayerConfigStream(in, cfg, err);
}

class EvalGame {
 public:
  EvalGame(int game_id, const EvalPlayerConfig& cur, const EvalPlayerConfig& cand, int max_moves, uint64_t seed)
      : id_(game_id), cur_is_black_(game_id % 2 == 0), max_moves_(max_moves), prob_(seed), game_(7.5f, true) {
    cfg_[0] = cur_is_black_ ? cur : cand;   // index 0 = black's player, 1 = white's
    cfg_[1] = cur_is_black_ ? cand : cur;
    for (int s = 0; s < 2; ++s) {   // eval.cc:124-160: node table kind and bias cache per player
      pool_[s].set_graph(cfg_[s].use_mcgs);
      if (cfg_[s].use_bias_cache) bias_[s].reset(new BiasCache(cfg_[s].bias_cache_alpha, cfg_[s].bias_cache_lambda));
      tree_[s] = pool_[s].GetOrCreate(game_.board().hash(), kBlack, false);
    }
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
  // the engine's NN cache answered (or evaluated) under symmetry `sym`, which need not be the one FillEval drew
  Symmetry eval_symmetry(int i) const { return sym_[i]; }
  void DeliverUnder(int i, p3hip_result& r, Symmetry sym) {
    sym_[i] = sym;
    Deliver(i, r);
  }
  // +1 cur won, -1 cand won, 0 draw
  int cur_result() const {
    const Color w = winner_;
    if (w == kEmpty) return 0;
    return (w == kBlack) == cur_is_black_ ? 1 : -1;
  }
  Color winner() const { return winner_; }
  bool cur_is_black() const { return cur_is_black_; }
  bool resigned() const { return resigned_; }
  int num_moves() const { return game_.num_moves(); }
  long visits() const { return visits_; }
  // the position / colour of evaluation i of the current Step() (for the caller's NN cache)
  const Position& eval_position(int i) const { return parallel_ ? search_.eval_pos(i) : *puct_.eval_game(); }
  Color eval_color_of(int i) const { return parallel_ ? search_.eval_color(i) : puct_.eval_color(); }
  // a cached result is handed over without a symmetry (it was stored un-symmetrised)
  void DeliverCached(int i, const p3hip_result& r) {
    if (parallel_) search_.Deliver(i, r);
    else puct_.Resume(r);
  }
  long collisions() const { return collisions_; }
  const Game& game() const { return game_; }

 private:
  void BeginSearch() {
    const int side = color_ == kBlack ? 0 : 1;
    const EvalPlayerConfig& c = cfg_[side];
    puct_.set_bias_cache(bias_[side].get());
    puct_.set_score_utility(MakeScoreUtility(c));
    // this scheduler has no clock: time_ms belongs to the thread-per-game driver
    parallel_ = c.num_threads_per_game > 1;   // UsesParallelSearch, eval.cc:99-101
    if (parallel_) {
      ParallelSearchParams p;
      p.batch = c.num_threads_per_game;
      p.visit_budget = c.n;
      p.puct = MakeSearchPuctParams(c);
      p.score_util = MakeScoreUtility(c);
      p.mode = c.search_mode;
      p.fns = VirtualFns{c.q_fn, c.n_fn, c.vl_delta};
      p.collision = c.collision_policy;
      p.detector = c.collision_detector;
      p.max_collision_retries = c.max_collision_retries;
      p.descent = c.descent_policy;
      p.max_o_ratio = c.max_o_ratio;
      p.bias_cache = bias_[side].get();
      search_.Begin(&game_, &pool_[side], tree_[side], color_, p);
    } else if (c.use_puct) {   // GumbelEvaluator::SearchRootPuct, eval.cc:241-258
      puct_.BeginPuct(&game_, &pool_[side], tree_[side], color_, c.n, MakeRootPuctParams(c), /*tau=*/1.0f, &prob_);
    } else {                   // GumbelEvaluator::SearchRoot, eval.cc:259-267
      puct_.Begin(&game_, &pool_[side], tree_[side], color_, MakeGumbelParams(c), &prob_);
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
      if (!next) next = pool_[s].GetOrCreate(game_.board().hash(), color_, game_.IsGameOver());
      pool_[s].Reap(next);
      tree_[s] = next;
      if (bias_[s]) bias_[s]->PruneUnused();
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
  std::unique_ptr<BiasCache> bias_[2];   // before the pools: nodes release their entries first
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
