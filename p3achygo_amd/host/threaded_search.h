// threaded_search.h — the reference's concurrent parallel search as it is written there: one OS
// thread per worker, per-node locks, two barriers per round (mcts::Search::Run in
// Mode::kConcurrent: cc/mcts/search.cc:84-253 Descend, :286-334 Backup, :336-458 SearchTask,
// :460-484 SpawnSearchTasks, :737-838 Run; policies cc/mcts/search.h:132-485).
//
// The scheduler-friendly form of the same search is BatchSearch (parallel_search.h): a resumable
// state machine whose workers descend one after the other, which is what the batched evaluation
// host drives.  This class is the thread-per-worker form for callers written against the
// reference's blocking API (cc/eval's PlayEvalGame, gtp): every worker thread descends through
// the marks (n_in_flight) the others are leaving at the same time, queues its leaf through the
// async NNInterface slot (LoadEntry / SignalReadyForInference / FetchEntry, kExplicit
// signalling), waits at barrier 1 until every worker has descended and every result is in, backs
// its own path up — a node is finalised by the last worker to come back through it — and waits
// at barrier 2 for the round to end.  Time control: a timer thread raises should_stop after
// total_visit_time_ms (search.cc:799-809).
#pragma once
#include <chrono>
#include <condition_variable>
#include <future>
#include <thread>

#include "nn_interface.h"
#include "parallel_search.h"

namespace p3 {

class ThreadedSearch {
 public:
  struct Params {   // Search::Params, search.h:88-107
    int num_threads = 8;
    int total_visit_budget = 128;
    int total_visit_time_ms = 0;
    PuctParams puct;
    ScoreUtilityParams score_util;
    VirtualFns fns{QFn::kVirtualLossSoft, NFn::kVirtualVisit, -1.5f};
    DescentPolicy descent = DescentPolicy::kDeterministic;
    CollisionPolicy collision = CollisionPolicy::kAbort;
    CollisionDetector detector = CollisionDetector::kNoOp;
    int max_collision_retries = 4;
    float max_o_ratio = 0.8f;
  };
  struct Result {   // Search::Result
    Loc move = kPassLoc;
    int num_visits = 0, num_aborted = 0, num_collisions = 0;
    long time_ms = 0;
  };

  // The slot's worker ids 0 .. num_threads-1 are this search's (task_offset = game * num_threads).
  explicit ThreadedSearch(NNInterface::Slot slot, BiasCache* bias_cache = nullptr) : slot_(slot), bias_cache_(bias_cache) {}
  void StopSearch() { g_.should_stop.store(true, std::memory_order_relaxed); }

  Result Run(Probability& probability, Game& game, NodePool* table, TreeNode* root, Color color_to_move, const Params& p) {
    const auto begin = std::chrono::steady_clock::now();
    p_ = p;
    game_ = &game; table_ = table; root_ = root; root_color_ = color_to_move;
    g_.did_signal = false;
    g_.round_parity = false;
    g_.num_workers = p.num_threads;
    g_.visit_budget = p.total_visit_budget;
    g_.descent_remaining = g_.round_remaining = p.num_threads;
    g_.pending = 0;
    g_.total_num_visits = g_.total_num_aborted = g_.total_num_collisions = 0;
    g_.should_stop = false;
    g_.should_stop_this_round = false;
    for (auto& x : g_.pending_each_level) x.store(0, std::memory_order_relaxed);
    if (game.IsGameOver()) return Result{};   // not a well-defined search
    if (!root->evaluated) {   // search.cc:781-787
      slot_.LoadEntry(0, game, color_to_move, probability);
      slot_.SignalReadyForInference();
      const p3hip_result r = slot_.FetchEntry(0, game, color_to_move);
      EvaluateRoot(r, root, color_to_move);
      AssignBiasCacheEntry(bias_cache_, Position(game), root);
    }
    std::promise<void> done;
    std::future<void> done_f = done.get_future();
    std::thread timer([&] {
      if (p.total_visit_time_ms <= 0) return;
      if (done_f.wait_for(std::chrono::milliseconds(p.total_visit_time_ms)) == std::future_status::timeout)
        g_.should_stop.store(true, std::memory_order_relaxed);
    });
    {
      std::vector<std::thread> workers;
      for (int w = 0; w < p.num_threads; ++w)
        workers.emplace_back([this, w, seed = probability.prng().next64()] {
          Probability prob(seed);   // the reference's workers seed from the clock (search.cc:476)
          SearchTask(w, prob);
        });
      for (auto& t : workers) t.join();
    }
    done.set_value();
    timer.join();
    Result res;
    res.num_visits = g_.total_num_visits.load();
    res.num_aborted = g_.total_num_aborted.load();
    res.num_collisions = g_.total_num_collisions.load();
    res.time_ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - begin).count();
    // best_lcb_move, search.cc:742-760
    std::array<std::pair<int, float>, kNumMoves> lcbs;
    for (int a = 0; a < kNumMoves; ++a) lcbs[a] = {a, Lcb(root, a)};
    std::stable_sort(lcbs.begin(), lcbs.end(), [](const auto& x, const auto& y) { return y.second < x.second; });
    for (const auto& al : lcbs)
      if (game.IsValidMove(MoveLoc(al.first), color_to_move)) { res.move = MoveLoc(al.first); break; }
    return res;
  }

 private:
  struct PathElem { TreeNode* node; int action; TopActions top; };   // action -1 = leaf marker
  using SearchPath = std::vector<PathElem>;

  struct GlobalSearchState {   // search.h:41-79
    std::mutex mu;
    std::condition_variable cv;
    bool did_signal = false, round_parity = false;
    int descent_remaining = 0, pending = 0, round_remaining = 0;
    int num_workers = 0, visit_budget = 0;
    std::array<std::atomic<int>, 8> pending_each_level{};
    std::atomic<int> total_num_visits{0}, total_num_aborted{0}, total_num_collisions{0};
    std::atomic<bool> should_stop{false};
    bool should_stop_this_round = false;
  };

  bool DetectorFires(int child_n_in_flight, int level) const {   // search.h:447-485
    const int base = std::max(1, (int)std::log2((double)std::max(p_.num_threads, 1)));
    const bool nf = child_n_in_flight + 1 >= base;
    const bool lv = level >= 0 && level < 8 && g_.pending_each_level[level].load(std::memory_order_acquire) >= base * (level + 1);
    switch (p_.detector) {
      case CollisionDetector::kNInFlight: return nf;
      case CollisionDetector::kLevelSaturation: return lv;
      case CollisionDetector::kProduct: return nf && lv;
      default: return false;
    }
  }

  // One descent (search.cc:84-253).  Returns false on a collision: the marks are undone and
  // `path` holds the colliding path for the collision policy.  On success the leaf is queued for
  // evaluation (when it needs one) and, once every worker has descended, fetched.
  bool Descend(int worker_id, Probability& prob, Game& game, const SearchPath& prefix, SearchPath& path,
               Color* leaf_color, bool* needs_eval) {
    auto on_collision = [&]() {
      for (PathElem& e : path) {
        const int old = e.node->n_in_flight.fetch_sub(1, std::memory_order_release);
        e.node->sum_n_in_flights.fetch_sub(old - 1, std::memory_order_release);
      }
      return false;
    };
    path.clear();
    Color c = root_color_;
    root_->sum_n_in_flights.fetch_add(root_->n_in_flight.fetch_add(1, std::memory_order_release), std::memory_order_release);
    const float max_o = p_.max_o_ratio * (float)(g_.num_workers - 1) / 2.0f;
    TreeNode* cur = root_;
    for (size_t idx = 0;; ++idx) {
      if (!cur->evaluated.load(std::memory_order_acquire)) {   // pending in another worker
        path.push_back({cur, -1, {}});
        return on_collision();
      }
      int child_in_flight;
      TreeNode* child;
      {
        std::lock_guard<std::mutex> node_lock(cur->mu);
        int action;
        TopActions top;
        if (idx < prefix.size()) {
          action = prefix[idx].action;
          top = prefix[idx].top;
        } else {
          top = p_.descent == DescentPolicy::kBuUct
                    ? BuUctTopScores(cur, game.board(), c, p_.puct, cur == root_, p_.fns, max_o)
                    : PuctTopScores(cur, game.board(), c, p_.puct, cur == root_, p_.fns);
          action = top[0].first;
        }
        if (action < 0) action = kPassEncoding;
        game.PlayMove(MoveLoc(action), c);
        c = Opp(c);
        path.push_back({cur, action, top});
        child = cur->child(action);
        if (child) {
          child_in_flight = child->n_in_flight.fetch_add(1, std::memory_order_acq_rel);
          child->sum_n_in_flights.fetch_add(child_in_flight, std::memory_order_release);
        } else {
          child = table_->GetOrCreateGuarded(game.board().hash(), c, game.IsGameOver());
          if (!child->evaluated.load(std::memory_order_acquire)) {
            child->color_to_move = c;
            if (game.IsGameOver()) child->is_terminal = true;
          }
          child_in_flight = child->n_in_flight.fetch_add(1, std::memory_order_acq_rel);
          child->sum_n_in_flights.fetch_add(child_in_flight, std::memory_order_release);
          cur->children.push_back(ChildEdge{(int16_t)action, 0, child});
          if (!child->evaluated.load(std::memory_order_acquire)) {
            if (child_in_flight > 0) {   // graph search: another worker transposed here first
              path.push_back({child, -1, {}});
              return on_collision();
            }
            cur = child;   // truly a leaf: claimed
            break;
          }
        }
      }
      if (DetectorFires(child_in_flight, (int)path.size() + 1)) {
        path.push_back({child, -1, {}});
        return on_collision();
      }
      cur = child;
      if (game.IsGameOver() || child->is_terminal) {
        if (child_in_flight == 0) break;   // this worker claims the terminal node
        path.push_back({child, -1, {}});
        return on_collision();
      }
    }
    path.push_back({cur, -1, {}});
    {
      const int level = (int)path.size() - 2;
      if (level >= 0 && level < 8) g_.pending_each_level[level].fetch_add(1, std::memory_order_release);
    }
    TreeNode* leaf = cur;
    if (game.IsGameOver()) leaf->is_terminal = true;
    *needs_eval = !leaf->evaluated.load(std::memory_order_acquire) && !game.IsGameOver();
    *leaf_color = c;
    if (*needs_eval) slot_.LoadEntry(worker_id, game, c, prob);   // LeafEvaluator::QueueEval
    {
      std::lock_guard<std::mutex> l(g_.mu);
      g_.descent_remaining--;
      if (*needs_eval) g_.pending++;
      if (g_.descent_remaining == 0 && g_.pending > 0 && !g_.did_signal) {
        slot_.SignalReadyForInference();
        g_.did_signal = true;
      }
    }
    g_.cv.notify_all();
    // FetchLeafEval, search.cc:54-71
    if (*needs_eval) {
      const p3hip_result r = slot_.FetchEntry(worker_id, game, c);
      EvaluateLeaf(r, leaf, c, root_color_, root_->init_score_est, p_.score_util);
      {
        std::lock_guard<std::mutex> l(g_.mu);
        g_.pending--;
      }
      g_.cv.notify_all();
    }
    if (game.IsGameOver()) {
      Scores s = game.GetScores();
      EvaluateTerminal(s, leaf, c, root_color_, root_->init_score_est, p_.score_util);
      leaf->evaluated.store(true, std::memory_order_release);
    }
    AssignBiasCacheEntry(bias_cache_, Position(game), leaf);
    return true;
  }

  void BackupStep(TreeNode* node, int action, bool is_leaf) {   // search.cc:286-325
    {
      std::lock_guard<std::mutex> l(node->mu);
      node->n += 1;
      if (!is_leaf)
        if (ChildEdge* e = node->edge(action)) e->visits += 1;
    }
    const int old = node->n_in_flight.fetch_sub(1, std::memory_order_acq_rel);
    if (old == 1) {   // last worker back through this node: its children are final
      if (is_leaf) {
        node->w = node->v = node->init_util_est;
        node->w_outcome = node->v_outcome = node->init_outcome_est;
        node->v_err = node->init_err_est;
      } else {
        RecomputeNodeStats(node, UpdateAndFetchObsBias(bias_cache_, node));
      }
    }
  }

  // SmartRetryCollisionPolicy::Handle (search.h:311-357)
  static bool SmartRetryPrefix(const SearchPath& path, SearchPath* prefix) {
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

  void ResetRoundLocked() {   // UnsafeGlobalStateReset, search.cc:73-83
    g_.descent_remaining = g_.num_workers;
    g_.did_signal = false;
    for (auto& x : g_.pending_each_level) x.store(0, std::memory_order_relaxed);
    g_.round_remaining = g_.num_workers;
    g_.round_parity = !g_.round_parity;
    g_.should_stop_this_round = g_.should_stop.load(std::memory_order_relaxed);
  }

  void SearchTask(int worker_id, Probability& prob) {   // search.cc:336-458
    auto should_stop = [&] {
      return g_.should_stop_this_round || g_.total_num_visits.load(std::memory_order_relaxed) >= g_.visit_budget;
    };
    SearchPath path, prefix;
    while (!should_stop()) {
      bool this_parity;
      {
        std::lock_guard<std::mutex> l(g_.mu);
        this_parity = g_.round_parity;
      }
      int retries = 0;
      Color leaf_color = kBlack;
      bool needs_eval = false;
      Game search_game = *game_;
      prefix.clear();
      bool ok = Descend(worker_id, prob, search_game, prefix, path, &leaf_color, &needs_eval);
      if (!ok) {
        g_.total_num_collisions.fetch_add(1, std::memory_order_relaxed);
        while (!ok) {
          bool retry = false;
          if (p_.collision != CollisionPolicy::kAbort && retries < p_.max_collision_retries) {
            ++retries;
            if (p_.collision == CollisionPolicy::kRetry) { prefix.clear(); retry = true; }
            else retry = SmartRetryPrefix(path, &prefix);
          }
          if (!retry) {   // abort: this worker sits the round out
            {
              std::lock_guard<std::mutex> l(g_.mu);
              g_.total_num_aborted.fetch_add(1, std::memory_order_relaxed);
              g_.descent_remaining--;
              if (g_.descent_remaining == 0 && g_.pending > 0 && !g_.did_signal) {
                slot_.SignalReadyForInference();
                g_.did_signal = true;
              }
            }
            g_.cv.notify_all();
            break;
          }
          Game retry_game = *game_;
          ok = Descend(worker_id, prob, retry_game, prefix, path, &leaf_color, &needs_eval);
        }
      }
      {   // barrier 1: every worker has descended, every queued evaluation is in
        std::unique_lock<std::mutex> l(g_.mu);
        g_.cv.wait(l, [&] { return g_.descent_remaining == 0 && g_.pending == 0; });
      }
      if (ok) {
        for (int i = (int)path.size() - 1; i >= 0; --i) BackupStep(path[i].node, path[i].action, i == (int)path.size() - 1);
        g_.total_num_visits.fetch_add(1, std::memory_order_relaxed);
      }
      {   // barrier 2: the last worker resets the round and flips the parity
        std::unique_lock<std::mutex> l(g_.mu);
        if (--g_.round_remaining == 0) {
          ResetRoundLocked();
          g_.cv.notify_all();
        }
        g_.cv.wait(l, [&] { return g_.round_parity != this_parity; });
      }
    }
  }

  NNInterface::Slot slot_;
  BiasCache* bias_cache_;
  Params p_;
  GlobalSearchState g_;
  Game* game_ = nullptr;
  NodePool* table_ = nullptr;
  TreeNode* root_ = nullptr;
  Color root_color_ = kBlack;
};

}  // namespace p3
