// parallel_search.h — the multi-leaf PUCT search used by the evaluation path.
//
// Restates mcts::Search (cc/mcts/search.{h,cc}) in its batch form, BatchSearch
// (search.cc:487-640): every round makes `batch` descents from the root — the first follows
// the best PUCT action everywhere, each later one forks an earlier path of the round at the
// node whose runner-up action has the smallest PUCT gap — collects the new leaves, has them
// evaluated TOGETHER, and backs the round up deepest-node-first (TopologicalBackup,
// search.cc:247-270) with RecomputeNodeStats (tree.h:175-241).  The final move is the legal
// move with the best lower confidence bound (Search::Run, search.cc:742-760,833).
//
// The reference's concurrent mode runs the same round structure on `num_threads` OS threads
// per game with per-node mutexes and two barriers per round; its batch mode is the
// single-threaded statement of it and is the one that maps onto this host, where a search
// never blocks: Step() returns the leaves of a round, the scheduler evaluates them in the
// engine's next batch together with every other game's, Deliver()/Step() continue.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <unordered_set>
#include <vector>

#include "board.h"
#include "search.h"   // includes confidence bounds (Lcb / Ucb)

namespace p3 {


// ---- top-4 PUCT scores (PuctScorer::TopScores, search_policy.h:318-351; the scorer is in search.h) ----
using TopActions = std::array<std::pair<int, float>, 4>;   // (action or -1, score)
inline TopActions PuctTopScores(const TreeNode* node, const Board& board, Color color, const PuctParams& pp,
                                bool is_root, const VirtualFns& vf = VirtualFns{}) {   // search_policy.h:318-351
  float scores[kNumMoves];
  PuctScoresAll(node, pp, is_root, scores, vf);
  TopActions top;
  top.fill({-1, -1e6f});
  for (int a = 0; a < kNumMoves; ++a) {
    int rank = 0;
    while (rank < 4 && !(scores[a] > top[rank].second)) ++rank;
    if (rank >= 4 || !board.IsValidMove(MoveLoc(a), color)) continue;
    for (int r = 3; r > rank; --r) top[r] = top[r - 1];
    top[rank] = {a, scores[a]};
  }
  return top;
}

// BuUctDescentPolicy::Run (search.h:174-251): the PUCT ranking with children whose share of
// in-flight descents O'(s, a) = sum_n_in_flights / (n + n_in_flight) exceeds `max_o` pushed out of
// it; when that leaves nothing, the plain ranking.
inline TopActions BuUctTopScores(const TreeNode* node, const Board& board, Color color, const PuctParams& pp,
                                 bool is_root, const VirtualFns& vf, float max_o) {
  float scores[kNumMoves];
  PuctScoresAll(node, pp, is_root, scores, vf);
  TopActions top, fallback;
  top.fill({-1, -1000.0f});
  fallback.fill({-1, -1000.0f});
  auto ranking = [](const TopActions& t, float score) {
    for (int r = 0; r < 4; ++r)
      if (score >= t[r].second) return r;
    return 4;
  };
  auto sift = [](TopActions& t, int r, std::pair<int, float> e) {
    if (r < 0 || r >= 4) return;
    for (int i = 3; i > r; --i) t[i] = t[i - 1];
    t[r] = e;
  };
  for (int a = 0; a < kNumMoves; ++a) {
    if (!board.IsValidMove(MoveLoc(a), color)) continue;
    const int fr = ranking(fallback, scores[a]);
    int r = ranking(top, scores[a]);
    if (const TreeNode* child = node->child(a)) {
      const float n = (float)child->n, nf = (float)child->n_in_flight.load(std::memory_order_acquire);
      const float o = (float)child->sum_n_in_flights.load(std::memory_order_acquire) / (n + nf);
      if (o > max_o) r = 4;
    }
    sift(fallback, fr, {a, scores[a]});
    sift(top, r, {a, scores[a]});
  }
  return top[0].first >= 0 ? top : fallback;
}

enum class DescentPolicy : uint8_t { kDeterministic = 0, kBuUct = 1 };                    // search.h:26-29
enum class SearchMode : uint8_t { kConcurrent = 0, kBatch = 1 };                           // Search::Mode
enum class CollisionPolicy : uint8_t { kAbort = 0, kRetry = 1, kSmartRetry = 2 };          // search.h:28-32
enum class CollisionDetector : uint8_t { kNoOp = 0, kNInFlight = 1, kLevelSaturation = 2, kProduct = 3 };   // :34-39

struct ParallelSearchParams {   // Search::Params (search.h:92-107)
  int batch = 8;                // num_threads: descents (leaves) per round
  int visit_budget = 128;       // total_visit_budget
  PuctParams puct;
  // kBatch: priority-first forks with identity Q / N (BatchSearch, search.cc:487-640, which
  // "currently ignores policies").  kConcurrent: the round structure of SearchTask
  // (search.cc:336-458) — every worker descends through the marks (n_in_flight) the earlier
  // workers of the round left, with the Q / N functions, collision policy and detector below;
  // the reference runs the workers on threads, here they descend one after the other.
  SearchMode mode = SearchMode::kBatch;
  VirtualFns fns;               // q_fn_kind / n_fn_kind / vl_delta
  ScoreUtilityParams score_util;   // score_util_params (search.h Params; player_config.cc:128)
  CollisionPolicy collision = CollisionPolicy::kAbort;
  int max_collision_retries = 4;
  CollisionDetector detector = CollisionDetector::kNoOp;
  DescentPolicy descent = DescentPolicy::kDeterministic;   // descent_policy_kind (concurrent mode)
  float max_o_ratio = 0.8f;                                  // BuUct: max_o = ratio * (workers - 1) / 2, search.cc:651-653
  BiasCache* bias_cache = nullptr;                           // optional (Search's second constructor)
};
struct ParallelSearchResult {   // Search::Result
  Loc move = kPassLoc;
  int num_visits = 0, num_aborted = 0, num_collisions = 0, rounds = 0;
};

class BatchSearch {
 public:
  // `game`, `pool` and `root` (a node of `pool` for the position of `game`) must outlive the search.
  void Begin(const Game* game, NodePool* pool, TreeNode* root, Color color, const ParallelSearchParams& p) {
    game_ = game; pool_ = pool; root_ = root; color_ = color; p_ = p;
    res_ = ParallelSearchResult{};
    root_pos_ = Position(*game);
    if (p_.batch > kMaxBatch) p_.batch = kMaxBatch;
    workers_.assign(p_.batch, Worker{});
    pending_.resize(p_.batch);
    n_evals_ = 0;
    stalled_rounds_ = 0;
    state_ = game->IsGameOver() ? State::kDone : (root->evaluated ? State::kRound : State::kRootEval);
  }

  // Advances until network evaluations are needed; returns how many (0: the search is done).
  int Step() {
    for (;;) {
      switch (state_) {
        case State::kDone:
          return 0;
        case State::kRootEval:   // search.cc:781-787
          eval_pos_[0] = &root_pos_;
          eval_color_[0] = color_;
          eval_worker_[0] = -1;
          n_evals_ = 1;
          state_ = State::kRootEvalWait;
          return 1;
        case State::kRootEvalWait:
          EvaluateRoot(pending_[0], root_, color_);
          AssignBiasCacheEntry(p_.bias_cache, root_pos_, root_);   // search.cc:787
          state_ = State::kRound;
          break;
        case State::kRound:
          if (res_.num_visits >= p_.visit_budget) { Finish(); return 0; }
          DescendRound();
          state_ = State::kRoundWait;
          if (n_evals_ > 0) return n_evals_;
          break;
        case State::kRoundWait: {
          const int before = res_.num_visits;
          FetchAndBackup();
          // Guard (not in the reference, which would spin): a detector that fires on every
          // descent (n-in-flight with fewer than four workers: threshold log2(batch) = 1) leaves
          // rounds without a single completed visit; give up after 32 of them in a row.
          stalled_rounds_ = res_.num_visits == before ? stalled_rounds_ + 1 : 0;
          if (stalled_rounds_ >= 32) { Finish(); return 0; }
          state_ = State::kRound;
          break;
        }
      }
    }
  }
  const Position& eval_pos(int i) const { return *eval_pos_[i]; }
  Color eval_color(int i) const { return eval_color_[i]; }
  void Deliver(int i, const p3hip_result& r) { pending_[i] = r; }
  const ParallelSearchResult& result() const { return res_; }

 private:
  enum class State { kRootEval, kRootEvalWait, kRound, kRoundWait, kDone };
  struct PathElem { TreeNode* node; int action; TopActions top; };   // action -1 = leaf marker
  struct Worker {
    std::vector<PathElem> path;
    Position pos;
    bool aborted = false, needs_eval = false;
    Color leaf_color = kBlack;
    int eval_slot = -1;
  };
  struct Fork { float diff; int path_num, path_index, puct_index; };

  bool DetectorFires(int in_flight_old, int level) const {   // search.h:447-485
    const int base = std::max(1, (int)std::log2((double)std::max(p_.batch, 1)));   // RunWithDetector
    const bool nf = in_flight_old + 1 >= base;
    const bool lv = level >= 0 && level < 8 && pending_level_[level] >= base * (level + 1);
    switch (p_.detector) {
      case CollisionDetector::kNInFlight: return nf;
      case CollisionDetector::kLevelSaturation: return lv;
      case CollisionDetector::kProduct: return nf && lv;
      default: return false;
    }
  }

  // Descend (search.cc:84-245) for one worker; false on collision (the marks are undone, the
  // colliding path stays in w.path for the collision policy).
  bool Descend(Worker& w, const std::vector<PathElem>& prefix) {
    const VirtualFns fns = p_.mode == SearchMode::kConcurrent ? p_.fns : VirtualFns{};
    w.pos = root_pos_;
    w.path.clear();
    Color c = color_;
    root_->sum_n_in_flights += in_flight(root_)++;   // search.cc:110-113
    TreeNode* cur = root_;
    auto collide = [&]() {   // on_collision, search.cc:96-106
      for (PathElem& e : w.path) {
        const int old = in_flight(e.node)--;
        e.node->sum_n_in_flights -= old - 1;
      }
      return false;
    };
    const bool bu_uct = p_.mode == SearchMode::kConcurrent && p_.descent == DescentPolicy::kBuUct;
    const float max_o = p_.max_o_ratio * (float)(p_.batch - 1) / 2.0f;
    for (size_t idx = 0;; ++idx) {
      if (!cur->evaluated) {   // a leaf claimed earlier in this round
        w.path.push_back({cur, -1, {}});
        return collide();
      }
      int action;
      TopActions top;
      if (idx < prefix.size()) { action = prefix[idx].action; top = prefix[idx].top; }
      else {
        top = bu_uct ? BuUctTopScores(cur, w.pos.board, c, p_.puct, cur == root_, fns, max_o)
                     : PuctTopScores(cur, w.pos.board, c, p_.puct, cur == root_, fns);
        action = top[0].first;
      }
      if (action < 0) action = kPassEncoding;   // no legal scored move: pass is always legal
      w.pos.PlayMove(MoveLoc(action), c);
      c = Opp(c);
      w.path.push_back({cur, action, top});
      TreeNode* child = cur->child(action);
      int child_in_flight;
      if (child) {
        child_in_flight = in_flight(child)++;
        child->sum_n_in_flights += child_in_flight;
      } else {
        // leaf case (search.cc:156-177).  In graph mode the table may hand back a node another
        // path already owns: claimed in this round -> collision; evaluated -> keep descending.
        child = pool_->GetOrCreate(w.pos.board.hash(), c, w.pos.IsGameOver());
        child->color_to_move = c;
        if (w.pos.IsGameOver()) child->is_terminal = true;
        child_in_flight = in_flight(child)++;
        child->sum_n_in_flights += child_in_flight;
        cur->children.push_back(ChildEdge{(int16_t)action, 0, child});
        if (!child->evaluated) {
          if (child_in_flight > 0) {
            w.path.push_back({child, -1, {}});
            return collide();
          }
          cur = child;   // a new node: claimed by this descent
          break;
        }
      }
      if (p_.mode == SearchMode::kConcurrent && DetectorFires(child_in_flight, (int)w.path.size() + 1)) {
        w.path.push_back({child, -1, {}});
        return collide();
      }
      cur = child;
      if (w.pos.IsGameOver() || child->is_terminal) {
        if (child_in_flight == 0) break;      // this descent claims the terminal node
        w.path.push_back({child, -1, {}});    // already claimed in this round
        return collide();
      }
    }
    w.path.push_back({cur, -1, {}});
    {
      const int level = (int)w.path.size() - 2;   // inc_pending_at_level, search.cc:212
      if (level >= 0 && level < 8) ++pending_level_[level];
    }
    if (w.pos.IsGameOver()) cur->is_terminal = true;
    w.needs_eval = !cur->evaluated && !w.pos.IsGameOver();
    w.leaf_color = c;
    return true;
  }

  // SmartRetryCollisionPolicy::Handle (search.h:311-357): refork the colliding path where the
  // runner-up is closest; false = abort.
  static bool SmartRetryPrefix(const std::vector<PathElem>& path, std::vector<PathElem>* prefix) {
    if (path.size() <= 1) { prefix->clear(); return true; }
    int min_index = -1;
    float min_diff = std::numeric_limits<float>::max();
    for (int i = 0; i < (int)path.size(); ++i) {
      if (path[i].action < 0 || path[i].top[1].first < 0) continue;
      const float diff = std::abs(path[i].top[0].second - path[i].top[1].second);
      if (diff < min_diff) { min_index = i; min_diff = diff; }
    }
    if (min_index < 0) return false;
    prefix->assign(path.begin(), path.begin() + min_index + 1);
    PathElem& e = prefix->back();
    const TopActions t = e.top;
    e.action = t[1].first;
    e.top = TopActions{t[0], t[2], t[3], {-1, -10000.0f}};
    return true;
  }

  void DescendRoundConcurrent() {   // SearchTask, search.cc:336-401, workers in order
    backup_.clear();
    n_evals_ = 0;
    for (int& x : pending_level_) x = 0;
    for (int wi = 0; wi < p_.batch; ++wi) {
      Worker& w = workers_[wi];
      w.aborted = false;
      w.needs_eval = false;
      std::vector<PathElem> prefix;
      int retries = 0;
      bool ok = Descend(w, prefix);
      if (!ok) ++res_.num_collisions;
      while (!ok) {
        bool retry = false;
        if (p_.collision != CollisionPolicy::kAbort && retries < p_.max_collision_retries) {
          ++retries;
          if (p_.collision == CollisionPolicy::kRetry) { prefix.clear(); retry = true; }
          else retry = SmartRetryPrefix(w.path, &prefix);
        }
        if (!retry) break;
        ok = Descend(w, prefix);
      }
      if (!ok) {
        w.aborted = true;
        w.path.clear();
        ++res_.num_aborted;
        continue;
      }
      for (int pi = 0; pi < (int)w.path.size(); ++pi)
        backup_.push_back(BackupElem{pi, w.path[pi].node, w.path[pi].action, pi == (int)w.path.size() - 1});
      if (w.needs_eval) {
        w.eval_slot = n_evals_;
        eval_pos_[n_evals_] = &w.pos;
        eval_color_[n_evals_] = w.leaf_color;
        eval_worker_[n_evals_] = wi;
        ++n_evals_;
      }
    }
  }

  void DescendRound() {   // search.cc:536-597
    if (p_.mode == SearchMode::kConcurrent) { DescendRoundConcurrent(); return; }
    forks_.clear();
    stored_forks_.clear();
    backup_.clear();
    n_evals_ = 0;
    for (int& x : pending_level_) x = 0;
    for (int wi = 0; wi < p_.batch; ++wi) {
      Worker& w = workers_[wi];
      w.aborted = false;
      w.needs_eval = false;
      std::vector<PathElem> prefix;
      if (!forks_.empty()) {   // fork with the smallest PUCT gap first
        auto it = std::min_element(forks_.begin(), forks_.end(), [](const Fork& a, const Fork& b) { return a.diff < b.diff; });
        const Fork f = *it;
        forks_.erase(it);
        const std::vector<PathElem>& src = workers_[f.path_num].path;
        prefix.assign(src.begin(), src.begin() + f.path_index + 1);
        prefix.back().action = prefix.back().top[f.puct_index].first;
      }
      if (!Descend(w, prefix)) {
        w.aborted = true;
        w.path.clear();
        ++res_.num_aborted;
        ++res_.num_collisions;
        continue;
      }
      for (int pi = 0; pi < (int)w.path.size(); ++pi) {
        const PathElem& e = w.path[pi];
        backup_.push_back(BackupElem{pi, e.node, e.action, pi == (int)w.path.size() - 1});
        if (e.action < 0) continue;   // the leaf has no scored actions
        if (stored_forks_.count(e.node)) continue;
        for (int k = 1; k < 4; ++k) {
          if (e.top[k].first < 0) continue;
          forks_.push_back(Fork{std::abs(e.top[0].second - e.top[k].second), wi, pi, k});
        }
        stored_forks_.insert(e.node);
      }
      if (w.needs_eval) {
        w.eval_slot = n_evals_;
        eval_pos_[n_evals_] = &w.pos;
        eval_color_[n_evals_] = w.leaf_color;
        eval_worker_[n_evals_] = wi;
        ++n_evals_;
      }
    }
  }

  void FetchAndBackup() {   // search.cc:603-640
    for (int wi = 0; wi < p_.batch; ++wi) {
      Worker& w = workers_[wi];
      if (w.aborted) continue;
      TreeNode* leaf = w.path.back().node;
      if (w.needs_eval) EvaluateLeaf(pending_[w.eval_slot], leaf, w.leaf_color, color_, root_->init_score_est, p_.score_util);
      if (w.pos.IsGameOver()) {
        Scores s = w.pos.GetScores();
        EvaluateTerminal(s, leaf, w.leaf_color, color_, root_->init_score_est, p_.score_util);
        leaf->evaluated = true;
      }
      AssignBiasCacheEntry(p_.bias_cache, w.pos, leaf);   // search.cc:251
      ++res_.num_visits;
    }
    // deepest first; a node is finalised on the last entry that passes through it
    std::stable_sort(backup_.begin(), backup_.end(), [](const BackupElem& a, const BackupElem& b) { return a.depth > b.depth; });
    for (const BackupElem& b : backup_) {
      TreeNode* node = b.node;
      node->n += 1;
      if (!b.is_leaf) {
        ChildEdge* e = node->edge(b.action);
        if (e) e->visits += 1;
      }
      if (--in_flight(node) == 0) {
        if (b.is_leaf) {
          node->w = node->v = node->init_util_est;
          node->w_outcome = node->v_outcome = node->init_outcome_est;
        } else {
          RecomputeNodeStats(node, UpdateAndFetchObsBias(p_.bias_cache, node));   // search.cc:277-279
        }
      }
    }
    ++res_.rounds;
    n_evals_ = 0;
  }

  void Finish() {   // best_lcb_move, search.cc:742-760
    std::array<std::pair<int, float>, kNumMoves> lcbs;
    for (int a = 0; a < kNumMoves; ++a) lcbs[a] = {a, Lcb(root_, a)};
    std::stable_sort(lcbs.begin(), lcbs.end(), [](const auto& x, const auto& y) { return y.second < x.second; });
    res_.move = kPassLoc;
    for (const auto& [a, lcb] : lcbs)
      if (root_pos_.board.IsValidMove(MoveLoc(a), color_)) { res_.move = MoveLoc(a); break; }
    state_ = State::kDone;
  }

  std::atomic<int>& in_flight(TreeNode* n) { return n->n_in_flight; }

  struct BackupElem { int depth; TreeNode* node; int action; bool is_leaf; };
  static constexpr int kMaxBatch = 64;

  const Game* game_ = nullptr;
  NodePool* pool_ = nullptr;
  TreeNode* root_ = nullptr;
  Color color_ = kBlack;
  ParallelSearchParams p_;
  ParallelSearchResult res_;
  State state_ = State::kDone;
  Position root_pos_;
  std::vector<Worker> workers_;
  std::vector<Fork> forks_;
  std::unordered_set<TreeNode*> stored_forks_;
  std::vector<BackupElem> backup_;
  const Position* eval_pos_[kMaxBatch];
  Color eval_color_[kMaxBatch];
  int eval_worker_[kMaxBatch];
  std::vector<p3hip_result> pending_;
  int n_evals_ = 0;
  int pending_level_[8] = {};   // GlobalSearchState::pending_each_level
  int stalled_rounds_ = 0;
};

}  // namespace p3
