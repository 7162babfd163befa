// search.h — serial Gumbel MCTS of the self-play host as a RESUMABLE state machine.
//
// Algorithmically this restates the reference's serial search:
//   GumbelEvaluator::SearchRoot           cc/mcts/gumbel.cc:260-559  (Gumbel top-k + sequential
//                                         halving, improved policy, tau sampling, root update)
//   GumbelEvaluator::Search/Backward/SingleBackup   gumbel.cc:674-821 (descent, incremental
//                                         mean / variance / 3rd moment / histogram backup)
//   PuctScorer::ComputeScores/TopMove     cc/mcts/search_policy.h:159-368 (non-root PUCT with FPU,
//                                         visit-scaled c_puct, child variance scaling)
//   LeafEvaluator::{EvaluateRoot,EvaluateLeaf,EvaluateTerminal,InitFields}
//                                         cc/mcts/leaf_evaluator.cc:83-215
//   TreeNode / accessors                  cc/mcts/tree.h:21-148
// What is new is the control flow.  The reference blocks one OS thread per game inside
// LeafEvaluator (nn_interface.cc:107-132), which caps a process at 256 games; here a search
// is an object whose Step() runs until it needs a network evaluation, hands the position
// out, and continues when Resume() delivers the result, so one host thread can interleave
// hundreds of games and every GPU batch is filled from all of them.
// Tree nodes keep children sparsely (a search at n = 32 touches a handful of the 362 moves).
#pragma once
#ifdef __AVX2__
#include <immintrin.h>
#endif
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <array>
#include <cmath>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/p3hip.h"
#include "bias_cache.h"
#include "board.h"
#include "rng.h"

namespace p3 {

constexpr float kDefaultScoreWeight = 0.5f;              // cc/mcts/constants.h:6
constexpr float kMaxQ = 1.0f + kDefaultScoreWeight;
constexpr float kMinQ = -1.0f - kDefaultScoreWeight;
constexpr int kNumVBuckets = 51;
constexpr float kBucketRange = 2.0f / kNumVBuckets;
constexpr float kDefaultFPU = 0.2f;                      // cc/mcts/search_policy.h:17

struct TreeNode;
struct ChildEdge {
  int16_t action;
  int visits;
  TreeNode* node;
};

struct TreeNode {                                        // cc/mcts/tree.h:21-91
  std::atomic<bool> evaluated{false};   // TreeNodeState::kNnEvaluated (read by concurrent descents)
  bool is_terminal = false;
  Color color_to_move = kEmpty;
  int n = 0;
  float w = 0, v = 0, v_var = 0;
  double v_m3 = 0;
  uint32_t v_categorical[kNumVBuckets] = {};
  float w_outcome = 0, v_outcome = 0, v_outcome_var = 0;
  double v_outcome_m3 = 0;
  float score = 0, v_err = 0;
  int max_child_n = 0;
  std::vector<ChildEdge> children;
  float move_logits[kNumMoves], move_probs[kNumMoves], opt_probs[kNumMoves];
  float init_outcome_est = 0, init_score_est = 0, init_score_var = 0, init_util_est = 0, init_err_est = 0;
  uint32_t mark = 0;  // NodePool::Reap
  // descents of the current round through this node, and the running sum of the values it had
  // when each descent arrived (tree.h:90-92; BuUct's O statistic) — atomics: the threaded search
  // (threaded_search.h) updates them from several workers at once
  std::atomic<int> n_in_flight{0}, sum_n_in_flights{0};
  std::mutex mu;        // node lock of the threaded search: child choice + child creation, n / visits in backup
  BiasNodeState bias;   // bias_cache_entry / last_obs_bias_term / last_weight_term, tree.h:83-86
  // graph-search identity (McgsNodeTable's key, node_table.h:77-118)
  uint64_t board_hash = 0;
  bool in_table = false;

  // back to a freshly constructed node (nodes are recycled by NodePool)
  void Reset() {
    bias.Release();
    evaluated.store(false, std::memory_order_relaxed);
    is_terminal = false;
    color_to_move = kEmpty;
    n = 0;
    w = v = v_var = 0;
    v_m3 = 0;
    std::fill(std::begin(v_categorical), std::end(v_categorical), 0u);
    w_outcome = v_outcome = v_outcome_var = 0;
    v_outcome_m3 = 0;
    score = v_err = 0;
    max_child_n = 0;
    children.clear();
    init_outcome_est = init_score_est = init_score_var = init_util_est = init_err_est = 0;
    mark = 0;
    n_in_flight.store(0, std::memory_order_relaxed);
    sum_n_in_flights.store(0, std::memory_order_relaxed);
    board_hash = 0;
    in_table = false;
  }

  ChildEdge* edge(int a) {
    for (auto& e : children)
      if (e.action == a) return &e;
    return nullptr;
  }
  const ChildEdge* edge(int a) const { return const_cast<TreeNode*>(this)->edge(a); }
  TreeNode* child(int a) const { const ChildEdge* e = edge(a); return e ? e->node : nullptr; }
  int child_visits(int a) const { const ChildEdge* e = edge(a); return e ? e->visits : 0; }
};

inline float N(const TreeNode* n) { return n ? (float)n->n : 0; }
inline float V(const TreeNode* n) { return n ? n->v : kMinQ; }
inline float VOutcome(const TreeNode* n) { return n ? n->v_outcome : -1.0f; }
inline float Q(const TreeNode* n, int a) { const TreeNode* c = n->child(a); return c ? -c->v : kMinQ; }
inline float QOutcome(const TreeNode* n, int a) { const TreeNode* c = n->child(a); return c ? -c->v_outcome : -1.0f; }
inline float SumChildrenN(const TreeNode* n) { return n ? (float)(n->n - 1) : 0; }
inline float MaxN(const TreeNode* n) { return n ? (float)n->max_child_n : 0; }

// MctsNodeTable / McgsNodeTable (cc/mcts/node_table.h:40-118, node_table.cc:12-68).  Tree mode
// hands out a fresh node per request; graph mode (Monte-Carlo graph search) returns the one node
// of a (board hash, colour to move, terminal) triple, so transpositions share statistics.  Reap
// keeps what is reachable from the new root.  GetOrCreateGuarded serialises concurrent callers.
class NodePool {
 public:
  explicit NodePool(bool graph = false) : graph_(graph) {}
  bool is_graph() const { return graph_; }
  void set_graph(bool graph) { graph_ = graph; }   // only while the pool is empty
  TreeNode* Create() {
    if (!free_.empty()) {
      TreeNode* n = free_.back();
      free_.pop_back();
      n->Reset();
      live_.push_back(n);
      return n;
    }
    owned_.emplace_back(new TreeNode());
    live_.push_back(owned_.back().get());
    return live_.back();
  }
  TreeNode* GetOrCreate(uint64_t board_hash, Color color_to_move, bool is_terminal) {
    if (!graph_) return Create();
    const Key key{board_hash, color_to_move, is_terminal};
    auto it = table_.find(key);
    if (it != table_.end()) return it->second;
    TreeNode* n = Create();
    n->board_hash = board_hash;
    n->color_to_move = color_to_move;
    n->is_terminal = is_terminal;
    n->in_table = true;
    table_.emplace(key, n);
    return n;
  }
  TreeNode* GetOrCreateGuarded(uint64_t board_hash, Color color_to_move, bool is_terminal) {
    std::lock_guard<std::mutex> l(mu_);
    return GetOrCreate(board_hash, color_to_move, is_terminal);
  }
  int Reap(TreeNode* new_root) {
    std::lock_guard<std::mutex> l(mu_);
    ++epoch_;
    std::vector<TreeNode*> work;
    if (new_root) work.push_back(new_root);
    while (!work.empty()) {
      TreeNode* n = work.back();
      work.pop_back();
      if (n->mark == epoch_) continue;
      n->mark = epoch_;
      for (auto& e : n->children)
        if (e.node) work.push_back(e.node);
    }
    int reaped = 0;
    size_t keep = 0;
    for (TreeNode* n : live_) {
      if (n->mark == epoch_) { live_[keep++] = n; continue; }
      if (n->in_table) table_.erase(Key{n->board_hash, n->color_to_move, n->is_terminal});
      n->bias.Release();   // ~TreeNode, tree.h:23-32
      n->in_table = false;
      free_.push_back(n);
      ++reaped;
    }
    live_.resize(keep);
    return reaped;
  }
  void Clear() { Reap(nullptr); }
  size_t Size() const { return live_.size(); }
  ~NodePool() { for (TreeNode* n : live_) n->bias.Release(); }

 private:
  struct Key {
    uint64_t hash;
    Color color;
    bool terminal;
    bool operator==(const Key& o) const { return hash == o.hash && color == o.color && terminal == o.terminal; }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const {
      return (size_t)(k.hash ^ ((uint64_t)(uint8_t)k.color << 1) ^ (uint64_t)k.terminal) * 0x9e3779b97f4a7c15ull;
    }
  };
  bool graph_;
  std::vector<std::unique_ptr<TreeNode>> owned_;
  std::vector<TreeNode*> live_, free_;
  std::unordered_map<Key, TreeNode*, KeyHash> table_;
  std::mutex mu_;
  uint32_t epoch_ = 0;
};

inline void SoftmaxN(const float* in, float* out, int n) {   // core::SoftmaxV, cc/core/vmath.h
  float m = in[0];
  for (int i = 1; i < n; ++i) m = std::max(m, in[i]);
  float s = 0;
  for (int i = 0; i < n; ++i) { out[i] = std::exp(in[i] - m); s += out[i]; }
  for (int i = 0; i < n; ++i) out[i] /= s;
}

// ---- leaf evaluation (leaf_evaluator.cc) -------------------------------------------
inline float ScoreTransform(float c_score, float score_est, float root_score_est) {   // :78-81
  return c_score * (float)M_2_PI * std::atan((score_est - root_score_est) / kBoardLen);
}

// ScoreUtilityParams / ScoreUtilityMode (leaf_evaluator.h:12-17): `direct` is the transform above,
// `integral` its expectation under a normal score distribution, read from a table over (score mean,
// score stddev) with bilinear interpolation (leaf_evaluator.cc:12-76; the same float / double
// arithmetic, including the z loop's accumulated float steps).
enum class ScoreUtilityMode : uint8_t { kDirect = 0, kIntegral = 1 };
struct ScoreUtilityParams {
  float score_weight = kDefaultScoreWeight;
  ScoreUtilityMode mode = ScoreUtilityMode::kDirect;
};
constexpr int kNumScoreMeans = P3HIP_NUM_SCORE_LOGITS, kNumScoreStddevs = P3HIP_NUM_SCORE_LOGITS / 2;
constexpr int kScoreInflectionPoint = 400;   // constants::kScoreInflectionPoint
inline const std::vector<float>& ScoreTransformTable() {   // kScoreTransformTable, :20-46
  static const std::vector<float> table = [] {
    constexpr float kZStep = 0.1f, kZBound = 5.0f;
    std::vector<float> t((size_t)kNumScoreMeans * kNumScoreStddevs);
    for (int score_idx = 0; score_idx < kNumScoreMeans; ++score_idx) {
      const float score_mean = score_idx - kScoreInflectionPoint + 0.5f;
      for (int stddev = 0; stddev < kNumScoreStddevs; ++stddev) {
        float total_pdf_mass = 0.0f, integral_unnormalized = 0.0f;
        for (float z = -kZBound; z <= kZBound; z += kZStep) {
          const float pdf_scaled = std::exp(-0.5 * z * z);
          const float score_transform = M_2_PI * std::atan((score_mean + z * stddev) / kBoardLen);
          total_pdf_mass += pdf_scaled;
          integral_unnormalized += score_transform * pdf_scaled;
        }
        t[(size_t)score_idx * kNumScoreStddevs + stddev] = integral_unnormalized / total_pdf_mass;
      }
    }
    return t;
  }();
  return table;
}
inline float ScoreTransformIntegral(float c_score, float score_est, float score_stddev, float root_score_est) {   // :48-76
  const float root_score_normalized = 0.75f * root_score_est;
  const float score_mean = score_est - root_score_normalized;
  const int score_floored = std::floor(score_mean - 0.5f);
  const int stddev_floored = std::floor(score_stddev);
  const int score_idx = std::clamp(score_floored + kScoreInflectionPoint, 0, kNumScoreMeans - 2);
  const int stddev_idx = std::clamp(stddev_floored, 0, kNumScoreStddevs - 2);
  const float mean_delta = (score_mean - 0.5f) - float(score_floored);
  const float stddev_delta = score_stddev - float(stddev_floored);
  const std::vector<float>& t = ScoreTransformTable();
  auto at = [&](int x, int y) { return t[(size_t)x * kNumScoreStddevs + y]; };
  const float a00 = at(score_idx, stddev_idx), a01 = at(score_idx, stddev_idx + 1);
  const float a10 = at(score_idx + 1, stddev_idx), a11 = at(score_idx + 1, stddev_idx + 1);
  const float b0 = a00 + stddev_delta * (a01 - a00);
  const float b1 = a10 + stddev_delta * (a11 - a10);
  return c_score * (b0 + mean_delta * (b1 - b0));
}
inline float ScoreUtility(const ScoreUtilityParams& sp, float score_est, float score_stddev, float root_score_est) {   // :125-132
  if (sp.mode == ScoreUtilityMode::kIntegral) return ScoreTransformIntegral(sp.score_weight, score_est, score_stddev, root_score_est);
  return ScoreTransform(sp.score_weight, score_est, root_score_est);
}

inline void InitFields(const p3hip_result& r, TreeNode* node, Color color_to_move) {   // :83-112
  std::memcpy(node->move_logits, r.move_logits, sizeof node->move_logits);
  std::memcpy(node->move_probs, r.move_probs, sizeof node->move_probs);
  std::memcpy(node->opt_probs, r.opt_move_probs, sizeof node->opt_probs);
  float value_est = r.value_probs[0] * -1 + r.value_probs[1] * 1;
  float score_est = 0, score_sq = 0;
  for (int i = 0; i < P3HIP_NUM_SCORE_LOGITS; ++i) {
    float s = i - 400 + .5f, p = r.score_probs[i];
    score_est += p * s;
    score_sq += p * s * s;
  }
  node->color_to_move = color_to_move;
  node->init_outcome_est = value_est;
  node->init_score_est = score_est;
  node->init_score_var = score_sq - score_est * score_est;
  node->init_err_est = std::sqrt(r.err2_outcome);
}

inline void EvaluateRoot(const p3hip_result& r, TreeNode* node, Color c) {   // :131-150
  InitFields(r, node, c);
  node->init_util_est = node->init_outcome_est;
  node->n = 1;
  node->w = node->v = node->init_util_est;
  node->w_outcome = node->v_outcome = node->init_outcome_est;
  node->v_err = node->init_err_est;
  int b = std::clamp((int)((node->init_util_est + 1.0f) / kBucketRange), 0, kNumVBuckets - 1);
  node->v_categorical[b] += 1;
  node->evaluated.store(true, std::memory_order_release);   // last: concurrent descents read the fields above
}

inline void EvaluateLeaf(const p3hip_result& r, TreeNode* node, Color c, Color root_color,
                         float root_score_est, const ScoreUtilityParams& sp = ScoreUtilityParams{}) {   // :152-162
  InitFields(r, node, c);
  root_score_est *= c == root_color ? 1.0f : -1.0f;
  node->init_util_est = node->init_outcome_est +
                        ScoreUtility(sp, node->init_score_est, std::sqrt(node->init_score_var), root_score_est);
  node->evaluated.store(true, std::memory_order_release);
}

inline void EvaluateTerminal(const Scores& s, TreeNode* node, Color c, Color root_color,
                             float root_score_est, const ScoreUtilityParams& sp = ScoreUtilityParams{}) {   // :164-186
  float ps = c == kBlack ? s.black_score : s.white_score;
  float os = c == kBlack ? s.white_score : s.black_score;
  float final_score = ps - os;
  root_score_est *= c == root_color ? 1.0f : -1.0f;
  float su = ScoreUtility(sp, final_score, 0.0f, root_score_est);
  node->color_to_move = c;
  node->is_terminal = true;
  node->init_util_est = (ps > os ? 1.0f : -1.0f) + su;
  node->init_outcome_est = ps > os ? 1.0f : -1.0f;
  node->init_score_est = final_score;
}

// ---- idempotent node statistics (tree.h:175-241) and the bias-cache hooks ---------------------
// Recomputes every statistic of `node` from its own estimate (less `obs_bias`) and its
// children's current statistics: used by the graph / parallel searches and, whenever a bias
// cache is active, by the serial search (gumbel.cc:446-448,608-610).
inline void RecomputeNodeStats(TreeNode* node, float obs_bias = 0.0f) {
  const float adj_init_util_est = node->init_util_est - obs_bias;
  float w = adj_init_util_est, w_outcome = node->init_outcome_est, total_score = node->init_score_est,
        w_err = node->init_err_est;
  int max_child_n = 0;
  for (const ChildEdge& e : node->children) {
    if (!e.node) continue;
    w -= e.visits * e.node->v;
    w_outcome -= e.visits * e.node->v_outcome;
    total_score -= e.visits * e.node->score;
    max_child_n = std::max(max_child_n, e.visits);
    w_err += e.visits * e.node->v_err;
  }
  const float v = w / node->n, v_outcome = w_outcome / node->n;
  const float m = adj_init_util_est - v, m_outcome = node->init_outcome_est - v_outcome;
  float m2 = m * m, m2_outcome = m_outcome * m_outcome;
  double m3 = m2 * m, m3_outcome = m2_outcome * m_outcome;
  for (const ChildEdge& e : node->children) {
    if (e.visits == 0 || !e.node) continue;
    const float dv = -e.node->v - v, dvo = -e.node->v_outcome - v_outcome;
    m2 += e.visits * (e.node->v_var + dv * dv);
    m2_outcome += e.visits * (e.node->v_outcome_var + dvo * dvo);
    m3 += e.visits * (-e.node->v_m3 + 3 * e.node->v_var * dv + dv * dv * dv);
    m3_outcome += e.visits * (-e.node->v_outcome_m3 + 3 * e.node->v_outcome_var * dvo + dvo * dvo * dvo);
  }
  node->w = w; node->w_outcome = w_outcome; node->v = v; node->v_outcome = v_outcome;
  node->score = total_score / node->n;
  node->max_child_n = max_child_n;
  node->v_var = m2 / node->n; node->v_outcome_var = m2_outcome / node->n;
  node->v_m3 = m3 / node->n; node->v_outcome_m3 = m3_outcome / node->n;
  node->v_err = w_err / node->n;
}

// GumbelEvaluator::AssignBiasCacheEntry, gumbel.cc:729-736 (search.cc:39-46)
inline void AssignBiasCacheEntry(BiasCache* cache, const Position& pos, TreeNode* node) {
  if (!cache) return;
  const std::optional<LocalPattern> pattern = LocalPattern::FromCurrentPosition(pos);
  if (!pattern) return;
  node->bias.bias_cache_entry = cache->GetOrCreate(*pattern);
}
// BiasCache::UpdateAndFetch(node), bias_cache.h:156-187 (FetchObsBias, search.cc:48-52): 0 without
// a cache or an entry
inline float UpdateAndFetchObsBias(BiasCache* cache, TreeNode* node) {
  if (!cache || !node->bias.bias_cache_entry) return 0.0f;
  float weighted_child_utility = 0;
  for (const ChildEdge& e : node->children)
    if (e.visits > 0) weighted_child_utility += e.visits * -(e.node->v);
  return cache->UpdateAndFetch(node->bias, node->init_util_est, weighted_child_utility, node->n - 1);
}

// ---- non-root PUCT (search_policy.h:159-368, IdentityQ / IdentityN) -----------------
// ---- confidence bounds: Student-t quantiles (tree.cc:14-40 uses boost students_t) -------------
inline double BetaCf(double a, double b, double x) {   // continued fraction of I_x(a, b)
  const double tiny = 1e-300;
  double qab = a + b, qap = a + 1, qam = a - 1, c = 1, d = 1 - qab * x / qap;
  if (std::abs(d) < tiny) d = tiny;
  d = 1 / d;
  double h = d;
  for (int m = 1; m <= 500; ++m) {
    const int m2 = 2 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1 + aa * d; if (std::abs(d) < tiny) d = tiny;
    c = 1 + aa / c; if (std::abs(c) < tiny) c = tiny;
    d = 1 / d; h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1 + aa * d; if (std::abs(d) < tiny) d = tiny;
    c = 1 + aa / c; if (std::abs(c) < tiny) c = tiny;
    d = 1 / d;
    const double del = d * c;
    h *= del;
    if (std::abs(del - 1) < 1e-15) break;
  }
  return h;
}
inline double RegIncBeta(double a, double b, double x) {
  if (x <= 0) return 0;
  if (x >= 1) return 1;
  const double bt = std::exp(std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b) + a * std::log(x) + b * std::log(1 - x));
  return x < (a + 1) / (a + b + 2) ? bt * BetaCf(a, b, x) / a : 1 - bt * BetaCf(b, a, 1 - x) / b;
}
// upper-tail quantile: t with P(T_nu > t) = p  (0 < p < 0.5)
inline double StudentTUpperQuantile(double nu, double p) {
  auto upper = [&](double t) { return 0.5 * RegIncBeta(nu / 2, 0.5, nu / (nu + t * t)); };
  double lo = 0, hi = 1;
  while (upper(hi) > p) hi *= 2;
  for (int i = 0; i < 200; ++i) {
    const double mid = 0.5 * (lo + hi);
    (upper(mid) > p ? lo : hi) = mid;
  }
  return 0.5 * (lo + hi);
}
constexpr float kLcbAlpha = 0.05f;   // tree.cc:15
inline float CachedTQuantile(int v) {   // tree.cc:16-33: two-sided alpha = 0.05, dof 1..1000
  static const std::array<float, 1000> table = [] {
    std::array<float, 1000> t;
    for (int i = 1; i <= 1000; ++i) t[i - 1] = (float)StudentTUpperQuantile(i, kLcbAlpha / 2);
    return t;
  }();
  if (v < 1) return table[0];
  if (v < 1000) return table[v - 1];
  return table.back();
}
inline float VVar(const TreeNode* n) { return !n || n->n < 3 ? kMaxQ : n->v_var; }   // tree.h:110-112
inline std::pair<float, float> ConfidenceInterval(const TreeNode* node, int a) {   // tree.cc:42-55
  const float n = (float)node->child_visits(a);
  const TreeNode* ch = node->child(a);
  if (!ch || n < 2) return {-1e6f + n, 1e6f - n};
  const float stddev = std::sqrt(VVar(ch) / n);
  const float z = CachedTQuantile((int)n - 1);
  return {Q(node, a) - z * stddev, Q(node, a) + z * stddev};
}
inline float Lcb(const TreeNode* node, int a) { return ConfidenceInterval(node, a).first; }
inline float Ucb(const TreeNode* node, int a) { return ConfidenceInterval(node, a).second; }
// the same interval at another significance level (tree.cc:34-50: ZScore = upper alpha/2 quantile)
inline std::pair<float, float> ConfidenceIntervalAlpha(const TreeNode* node, int a, float alpha) {
  if (std::abs(alpha - kLcbAlpha) < 1e-5f) return ConfidenceInterval(node, a);
  const float n = (float)node->child_visits(a);
  const TreeNode* ch = node->child(a);
  if (!ch || n < 2) return {-1e6f + n, 1e6f - n};
  const float stddev = std::sqrt(VVar(ch) / n);
  const float z = (float)StudentTUpperQuantile((double)((int)n - 1), alpha / 2);
  return {Q(node, a) - z * stddev, Q(node, a) + z * stddev};
}

// Public ComputeImprovedPolicy(node, n) / ComputeKLD of the reference (gumbel.cc:172-204),
// used by the self-play loop for its pre-/post-search KL statistics.
inline float VMixedOf(const TreeNode* node) {   // gumbel.cc:68-87
  if (SumChildrenN(node) == 0) return node->init_util_est;
  double wq = 0, vp = 0;
  for (const ChildEdge& e : node->children)
    if (e.visits > 0) { wq += node->move_probs[e.action] * -e.node->v; vp += node->move_probs[e.action]; }
  const double iq = wq * SumChildrenN(node) / vp + node->init_util_est;
  return (float)(iq / (1 + SumChildrenN(node)));
}
inline void ComputeImprovedPolicyN(const TreeNode* node, int n, float* out) {
  auto q_norm = [](float q) { return (q + 1.5f) / 3.0f; };
  const float v_mix = q_norm(VMixedOf(node));
  const float scale = n <= 0 ? 0.0f : 2 * std::log((float)n);
  float logits[kNumMoves];
  for (int a = 0; a < kNumMoves; ++a) {
    const float q = node->child_visits(a) > 0 ? q_norm(Q(node, a)) : v_mix;
    logits[a] = node->move_logits[a] + (50 + scale) * q;   // QTransform, kVisit = 50, kValueScale = 1
  }
  SoftmaxN(logits, out, kNumMoves);
}
inline float ComputeKLD(const float* target, const float* prior) {
  double kld = 0;
  for (int i = 0; i < kNumMoves; ++i)
    if (target[i] != 0.0f) kld += target[i] * std::log(target[i] / (prior[i] + 1e-10));
  return (float)kld;
}

enum class PuctRootSelection { kVisitCount = 0, kLcb = 1, kVisitCountSample = 2 };   // search_policy.h:18-22
struct PuctParams {   // search_policy.h:24-44
  PuctRootSelection kind = PuctRootSelection::kVisitCountSample;   // self-play's fast moves
  float c_puct = 1.0f, c_puct_visit_scaling = 0.45f;
  float c_puct_v_2 = 3.0f;
  bool use_puct_v = false;          // PUCT-V exploration term instead of the PUCT one
  bool enable_var_scaling = false;
  int var_scale_prior_visits = 0;
  float tau = 1.0f;
  bool enable_m3_bonus = false;     // skewness bonus (third central moment of a child's values)
  int m3_prior_visits = 20;
  float p_opt_weight = 0.0f;        // prior = (1 - w) * move_probs + w * opt_probs
  float root_fpu = kDefaultFPU;
};

// ---- PUCT scores (PuctScorer::ComputeScores, search_policy.h:159-316) with the virtual-loss Q / N
// functions of the parallel search (identity in the serial searches) --------------------------------
enum class QFn : uint8_t { kIdentity = 0, kVirtualLoss = 1, kVirtualLossSoft = 2 };   // search_policy.h:400-446
enum class NFn : uint8_t { kIdentity = 0, kVirtualVisit = 1 };                          // :406-459
struct VirtualFns {
  QFn q = QFn::kIdentity;
  NFn n = NFn::kIdentity;
  float vl_delta = -1.5f;
  float Q(float q_, int n_, int in_flight) const {
    if (q == QFn::kVirtualLoss) return q_ + in_flight * vl_delta;
    if (q == QFn::kVirtualLossSoft) return in_flight == 0 ? q_ : (q_ * n_ + in_flight * vl_delta) / (float)(n_ + in_flight);
    return q_;
  }
  float N(int n_, int in_flight) const { return n == NFn::kVirtualVisit ? (float)(n_ + in_flight) : (float)n_; }
};

inline float ScaleCPuct(float c_puct, float c_puct_visit_scaling, int n) {   // :152-157
  return c_puct + c_puct_visit_scaling * std::log((n + 500.0f) / 500.0f);
}

inline void PuctScoresAll(const TreeNode* node, const PuctParams& pp, bool is_root, float* scores,
                          const VirtualFns& vf = VirtualFns{}) {
  // A node has 362 actions and a handful of children.  The reference's scorer walks all 362 several times
  // (search_policy.h:159-368); here every sum runs over the children only — in ascending action order where the
  // reference's order matters (float sums: the skipped terms are exact zeros) — and the actions without a child get
  // their score from one vectorised pass.  Same operations in the same order on every path (the host is built
  // with -ffp-contract=off); a third of a self-play host thread's time was in the 362-wide version.
  const int n = node->n;
  const float v = node->v, v_var = node->v_var;
  // the prior: the policy, the optimistic policy, or a blend (:171-185)
  float mp_blend[kNumMoves];
  const float* mp = node->move_probs;
  if (pp.p_opt_weight == 1.0f) mp = node->opt_probs;
  else if (pp.p_opt_weight != 0.0f) {
    for (int a = 0; a < kNumMoves; ++a)
      mp_blend[a] = node->move_probs[a] + pp.p_opt_weight * (node->opt_probs[a] - node->move_probs[a]);
    mp = mp_blend;
  }
  float q_std_weighted = 0;
  double q_m3_std_weighted = 0;
  for (const ChildEdge& e : node->children) {
    if (e.visits >= 3) {
      q_std_weighted += std::sqrt(e.node->v_var) * e.visits;
      q_m3_std_weighted += std::cbrt(-e.node->v_m3) * e.visits;
    }
  }
  const float q_std_mean = q_std_weighted / n;
  const double q_m3_std_mean = q_m3_std_weighted / n;
  // children in ascending action order (they sit in order of first visit)
  const int nc = (int)node->children.size();
  constexpr int kInline = 64;
  int order_inline[kInline];
  std::vector<int> order_heap;
  int* order = order_inline;
  if (nc > kInline) { order_heap.resize(nc); order = order_heap.data(); }
  for (int i = 0; i < nc; ++i) {
    int k = i;
    const int act = node->children[i].action;
    while (k > 0 && node->children[order[k - 1]].action > act) { order[k] = order[k - 1]; --k; }
    order[k] = i;
  }
  auto inflight_of = [](const ChildEdge& e) -> int { return e.node ? (int)e.node->n_in_flight : 0; };
  float p_explored = 0;
  for (int i = 0; i < nc; ++i) {
    const ChildEdge& e = node->children[order[i]];
    if (e.visits + inflight_of(e) > 0) p_explored += mp[e.action];
  }
  const float v_fpu = v - (is_root ? pp.root_fpu : kDefaultFPU) * std::sqrt(p_explored);
  const float c_puct = ScaleCPuct(pp.c_puct, pp.c_puct_visit_scaling, n);
  const float c_puct_v_2 = ScaleCPuct(pp.c_puct_v_2, pp.c_puct_visit_scaling, n);
  float total_n = 1;   // the visit to the node itself (:216-224)
  for (int i = 0; i < nc; ++i) {
    const ChildEdge& e = node->children[order[i]];
    total_n += vf.N(e.visits, inflight_of(e));
  }
  const float sqrt_n = std::sqrt(total_n);
  // the score of one action as the reference writes it (:240-294)
  auto score_of = [&](int a, int cv, int inflight, const TreeNode* child) -> float {
    float scale = 1.0f;   // c_puct_var_child_scale_factor (:240-251)
    if (pp.enable_var_scaling && cv >= 3 && q_std_mean != 0) {
      const float pw = (float)pp.var_scale_prior_visits;
      scale = (pw + cv * (std::sqrt(child->v_var) / q_std_mean)) / (pw + cv);
    }
    const float child_n = vf.N(cv, inflight);
    const float q = vf.Q(cv > 0 ? -child->v : v_fpu, cv, inflight);
    double m3_bonus = 0.0;   // compute_m3_bonus (:262-276), as written there
    if (pp.enable_m3_bonus && cv >= 3) {
      const float pw = (float)pp.m3_prior_visits;
      const double abs_bonus = std::cbrt(-child->v_m3) - q_m3_std_mean;
      m3_bonus = (pw + abs_bonus) / double(pw + cv);
    }
    float explore;
    if (pp.use_puct_v) {   // compute_puct_v_explore_term (:279-288)
      const float var = cv < 3 ? (n < 3 ? 1.0f : v_var) : child->v_var;
      const float stddev = std::sqrt(var);
      const float var_scale_term = mp[a] * stddev * (sqrt_n / (1 + child_n));
      const float n_scale_term = mp[a] * std::log(total_n) / (1 + child_n);
      explore = c_puct * var_scale_term + c_puct_v_2 * n_scale_term;
    } else {               // compute_puct_explore_term (:291-294)
      explore = c_puct * scale * mp[a] * (sqrt_n / (1 + child_n));
    }
    return (float)(explore + q + m3_bonus);
  };
  // actions without a child (no visit, nothing in flight): scale = 1, child_n = 0, q = v_fpu, no bonus
  if (!pp.use_puct_v) {
    const float k0 = c_puct * 1.0f, k1 = sqrt_n / (1 + 0.0f);
    int a = 0;
#ifdef __AVX2__
    const __m256 vk0 = _mm256_set1_ps(k0), vk1 = _mm256_set1_ps(k1), vq = _mm256_set1_ps(v_fpu);
    for (; a + 8 <= kNumMoves; a += 8)
      _mm256_storeu_ps(scores + a, _mm256_add_ps(_mm256_mul_ps(_mm256_mul_ps(vk0, _mm256_loadu_ps(mp + a)), vk1), vq));
#endif
    for (; a < kNumMoves; ++a) scores[a] = k0 * mp[a] * k1 + v_fpu;
  } else {
    for (int a = 0; a < kNumMoves; ++a) scores[a] = score_of(a, 0, 0, nullptr);
  }
  for (const ChildEdge& e : node->children) scores[e.action] = score_of(e.action, e.visits, inflight_of(e), e.node);
}

// PuctScorer::TopMove (search_policy.h:353-368): the best-scoring legal move.
inline int PuctTopMove(const TreeNode* node, const Board& board, Color color, const PuctParams& pp,
                       bool is_root = false) {
  float scores[kNumMoves];
  PuctScoresAll(node, pp, is_root, scores);
  float best = -1e6f;
  int best_a = -1;
  for (int a = 0; a < kNumMoves; ++a) {
    if (scores[a] > best) {
      Loc mv = a == kPassEncoding ? kPassLoc : AsLoc(a);
      if (board.IsValidMove(mv, color)) {
        best = scores[a];
        best_a = a;
      }
    }
  }
  return best_a;
}

// ---- Gumbel root search ----------------------------------------------------------------
struct GumbelParams {   // GumbelSearchParams, cc/mcts/gumbel.h:41-57
  int n = 32, k = 4;
  float noise_scaling = 1.0f;
  bool disable_pass = false;
  float tau = 0.0f;
  int nonroot_var_scale_prior_visits = 10;
  bool early_stopping_enabled = false;   // gumbel.h:46 (--early_stopping_enabled, off by default)
};

struct GumbelResult {   // cc/mcts/gumbel.h:31-39 (subset)
  Loc nn_move = kNoopLoc, mcts_move = kNoopLoc;
  float pi_improved[kNumMoves];
  float kld = 0;
  uint32_t visits = 0;
};

inline Loc MoveLoc(int a) { return a == kPassEncoding ? kPassLoc : AsLoc(a); }
inline int MoveIdx(Loc l) { return l == kPassLoc ? kPassEncoding : Idx(l); }

class GumbelSearch {
 public:
  enum class Status { kNeedEval, kDone };
  // IssueNext: an evaluation was requested (eval_game()/eval_color() name it; it stays in flight until Deliver or
  // ResolveBack), nothing can start before a result in flight arrives, or the search is complete
  enum class Issue { kNeedEval, kBlocked, kDone };
  static constexpr int kMaxInflight = 4;   // (6 measured no better than 4, profiles/r04_lanes_c3_more_lanes.txt; each costs a Position per game)

  // `game`, `pool`, `prob` must outlive the search.  `root` must belong to `pool`.
  void Begin(Game* game, NodePool* pool, TreeNode* root, Color color, const GumbelParams& p,
             Probability* prob) {
    game_ = game; pool_ = pool; root_ = root; color_ = color; p_ = p; prob_ = prob;
    puct_root_ = false;
    root_pos_ = Position(*game);
    state_ = root->evaluated ? State::kPrepare : State::kRootEval;
    res_ = GumbelResult();
    head_ = count_ = 0;
    have_pending_ = false;
  }

  // SearchRootPuct (gumbel.cc:563-666) with PuctRootSelectionPolicy::kVisitCountSample as
  // self-play uses it for fast moves (self_play_thread.cc:600-611): n PUCT playouts from the
  // root, improved policy = normalised new visit counts, move = argmax of them when tau > 0
  // (the branch is inverted in the reference).
  void BeginPuct(Game* game, NodePool* pool, TreeNode* root, Color color, int n, const PuctParams& pp,
                 float tau, Probability* prob) {
    GumbelParams gp;
    gp.n = n; gp.tau = tau;
    Begin(game, pool, root, color, gp, prob);
    puct_root_ = true;
    puct_pp_ = pp;
  }

  // Runs until an evaluation is needed (then eval_game()/eval_color() name the position) or
  // the search is complete (result()): one evaluation in flight, the reference's order of work.
  Status Step() {
    if (have_pending_) { have_pending_ = false; Deliver(pending_); }
    return IssueNext(1) == Issue::kNeedEval ? Status::kNeedEval : Status::kDone;
  }
  void Resume(const p3hip_result& r) { pending_ = r; have_pending_ = true; }

  // Several evaluations in flight (round 4).  Within one sequential-halving round the playouts of different
  // considered actions touch disjoint subtrees — a playout starts at its action's child of the root
  // (gumbel.cc:412-452) and only that child's subtree is read by its descent and written by its backup — and the
  // root's own statistics are read only when a round closes or an early-stopping check runs.  So a playout may
  // start while others wait for their evaluations if its action differs from all of theirs and no such read lies
  // between: the tree, the order of evaluation requests and every result are the ones the one-at-a-time search
  // produces.  Off (one in flight) for a PUCT root, in graph mode and with a bias cache, where playouts share state.
  // Results come back in request order (Deliver) or, for the request just made, at once (ResolveBack: a cache hit).
  Issue IssueNext(int max_inflight) {
    if (puct_root_ || bias_cache_ || pool_->is_graph()) max_inflight = 1;
    max_inflight = std::min(max_inflight, kMaxInflight);
    for (;;) {
      switch (state_) {
        case State::kRootEval: {
          Visit& v = ring_[(head_ + count_) % kMaxInflight];   // nothing else is in flight
          v.is_root = true;
          ++count_;
          eval_game_ = &root_pos_;
          eval_color_ = color_;
          state_ = State::kRootEvalWait;
          return Issue::kNeedEval;
        }
        case State::kRootEvalWait:   // Deliver() moves on
          return Issue::kBlocked;
        case State::kPrepare:
          if (puct_root_) {
            PreparePuct();
          } else if (Prepare()) {
            state_ = State::kDone;
            return Issue::kDone;
          }
          state_ = State::kNextVisit;
          break;
        case State::kNextVisit: {
          if (count_ >= max_inflight) return Issue::kBlocked;
          if (puct_root_) {
            if (visits_spent_ >= (uint32_t)p_.n) { FinishPuct(); state_ = State::kDone; return Issue::kDone; }
            if (StartVisitPuct()) return Issue::kNeedEval;
            break;
          }
          const Loop lp = AdvanceLoop(count_ > 0);
          if (lp == Loop::kWait) return Issue::kBlocked;
          if (lp == Loop::kStop) { Finish(); state_ = State::kDone; return Issue::kDone; }
          for (int i = 0; i < count_; ++i)   // same action as a playout in flight: its subtree is not settled
            if (ring_[(head_ + i) % kMaxInflight].a0 == gm_[cand_].enc) return Issue::kBlocked;
          if (StartVisit()) return Issue::kNeedEval;   // else the playout completed synchronously
          break;
        }
        case State::kDone:
          return Issue::kDone;
      }
    }
  }
  // the result of the OLDEST evaluation in flight
  void Deliver(const p3hip_result& r) {
    Visit& v = ring_[head_];
    head_ = (head_ + 1) % kMaxInflight;
    --count_;
    FinishVisit(v, r);
  }
  // the result of the evaluation IssueNext just requested (served from a cache): completes at once, like a terminal leaf
  void ResolveBack(const p3hip_result& r) {
    --count_;
    FinishVisit(ring_[(head_ + count_) % kMaxInflight], r);
  }
  int inflight() const { return count_; }
  // GumbelEvaluator's bias_cache constructor argument (gumbel.cc:247-254); nullptr = off
  void set_score_utility(const ScoreUtilityParams& sp) { score_util_ = sp; }
  void set_bias_cache(BiasCache* cache) { bias_cache_ = cache; }
  const Position* eval_game() const { return eval_game_; }
  Color eval_color() const { return eval_color_; }
  const GumbelResult& result() const { return res_; }

 private:
  enum class State { kRootEval, kRootEvalWait, kPrepare, kNextVisit, kDone };
  enum class Loop { kGo, kStop, kWait };
  struct MoveInfo { float prob = 0, logit = 0, noise = 0, qtransform = 0; int enc = -1; };
  struct PathEntry { int action; TreeNode* node; };
  // one playout: the position it reached, its path, the considered action it started from (-1: a PUCT
  // playout from the root, or the root's own evaluation)
  struct Visit {
    bool is_root = false;
    int a0 = -1;
    Position game;
    std::vector<PathEntry> path;
    Color leaf_color = kBlack;
  };
  void FinishVisit(Visit& v, const p3hip_result& r) {
    if (v.is_root) {
      v.is_root = false;
      EvaluateRoot(r, root_, color_);
      AssignBiasCacheEntry(bias_cache_, root_pos_, root_);   // gumbel.cc:275-278
      state_ = State::kPrepare;
      return;
    }
    EvaluateLeaf(r, v.path.back().node, v.leaf_color, color_, root_->init_score_est, score_util_);
    CompleteVisit(v);
  }
  static constexpr float kSmallLogit = -10000;
  static constexpr int kVisit = 50;

  static int ilog2(int x) { int i = 1; while ((x >> i) > 0) ++i; return i - 1; }
  static bool Greater(const MoveInfo& x, const MoveInfo& y) {
    return x.logit + x.noise + x.qtransform > y.logit + y.noise + y.qtransform;
  }

  // gumbel.cc:268-330: legal-move mask, Gumbel noise, top-k.  Returns true when n == 1.
  bool Prepare() {
    k_ = p_.k;
    num_rounds_ = std::max(ilog2(k_), 1);
    int k_valid = 0;
    for (int i = 0; i < kNumMoves; ++i) {
      gm_[i] = MoveInfo();
      if ((p_.disable_pass && i == kPassEncoding) || !root_pos_.IsValidMove(MoveLoc(i), color_)) {
        masked_logits_[i] = kSmallLogit;
        gm_[i].logit = kSmallLogit;
        continue;
      }
      masked_logits_[i] = root_->move_logits[i];
      gm_[i].prob = root_->move_probs[i];
      gm_[i].logit = root_->move_logits[i];
      gm_[i].noise = p_.noise_scaling * prob_->GumbelSample();
      gm_[i].enc = i;
      ++k_valid;
    }
    k_ = std::min(k_valid, k_);
    std::sort(gm_, gm_ + kNumMoves, Greater);
    int amax = 0;
    for (int i = 1; i < kNumMoves; ++i)
      if (root_->move_logits[i] > root_->move_logits[amax]) amax = i;
    res_.nn_move = MoveLoc(amax);
    if (p_.n == 1) {
      res_.mcts_move = MoveLoc(gm_[0].enc);
      std::memcpy(res_.pi_improved, root_->move_probs, sizeof res_.pi_improved);
      return true;
    }
    top_k_.assign(gm_, gm_ + k_);
    theoretical_winner_visits_ = 0;
    for (int kt = k_; kt > 1; kt /= 2)
      theoretical_winner_visits_ += (int)std::round(float(p_.n) / float(num_rounds_ * kt));
    m_ = k_;
    visits_spent_ = 0;
    round_open_ = false;
    return false;
  }

  // advances (round, visit, candidate) to the next candidate to visit; kStop when k <= 1.  `busy`: playouts are in
  // flight, so anything that reads the root children's statistics (closing a round, an early-stopping check) must
  // wait for them: kWait, with nothing changed.
  Loop AdvanceLoop(bool busy) {
    for (;;) {
      if (!round_open_) {
        if (k_ <= 1) return Loop::kStop;
        visits_per_action_ = (int)std::round(float(p_.n) / float(num_rounds_ * k_));
        visit_num_ = 0;
        cand_ = 0;
        round_open_ = true;
        if (visits_per_action_ <= 0) { CloseRound(); continue; }   // (a round opens with nothing in flight)
      }
      if (cand_ >= k_) {
        // a sweep over the round's candidates is complete: early-stopping check every
        // ceil(v / 4) sweeps (gumbel.cc:396-407,459-466)
        if (p_.early_stopping_enabled) {
          const int interval = (visits_per_action_ + 3) / 4;
          if (interval > 0 && visit_num_ % interval == interval - 1 && visit_num_ >= interval - 1) {
            if (busy) return Loop::kWait;
            for (int i = 0; i < k_; ++i)
              if (gm_[i].enc >= 0) gm_[i].qtransform = (kVisit + MaxN(root_)) * -V(root_->child(gm_[i].enc));
            std::sort(gm_, gm_ + k_, Greater);
            if (CanStopEarly()) { CloseRound(); continue; }
          }
        }
        cand_ = 0;
        ++visit_num_;
      }
      if (visit_num_ >= visits_per_action_) {
        if (busy) return Loop::kWait;
        CloseRound();
        continue;
      }
      if (gm_[cand_].enc < 0) { ++cand_; continue; }
      return Loop::kGo;
    }
  }
  // can_stop_early (gumbel.cc:323-351): with the candidates sorted, no move of the half about to
  // be dropped can still be the best one — its upper bound is below the best lower bound of the
  // half that stays — at significance lambda / ceil(k / 2), lambda = 0.95^(1 / rounds) (sic);
  // every candidate needs kMinEarlyStoppingVisits = 10 visits first.
  bool CanStopEarly() const {
    const float lambda = std::pow(0.95f, 1.0f / num_rounds_);
    const int kb = k_ / 2 + k_ % 2;
    float top_lcb = -2, bot_ucb = -2;
    for (int i = 0; i < k_; ++i) {
      const int a = gm_[i].enc;
      if (a < 0) continue;
      if (!root_->child(a) || root_->child_visits(a) < 10) return false;
      const auto ci = ConfidenceIntervalAlpha(root_, a, lambda / kb);
      if (i < k_ / 2) top_lcb = std::max(top_lcb, ci.first);
      else bot_ucb = std::max(bot_ucb, ci.second);
    }
    return bot_ucb <= top_lcb;
  }
  void CloseRound() {   // gumbel.cc:470-473
    for (int i = 0; i < k_; ++i)
      if (gm_[i].enc >= 0) gm_[i].qtransform = (kVisit + MaxN(root_)) * -V(root_->child(gm_[i].enc));
    std::sort(gm_, gm_ + k_, Greater);
    k_ /= 2;
    round_open_ = false;
  }

  // `pos` = the position after action `a`, `next` = its side to move (graph search shares the node
  // of a transposition: NodeTable::GetOrCreate, gumbel.cc:700-705)
  TreeNode* GetOrCreateChild(TreeNode* parent, int a, const Position& pos, Color next) {
    ChildEdge* e = parent->edge(a);
    if (!e) {
      parent->children.push_back(ChildEdge{(int16_t)a, 0, pool_->GetOrCreate(pos.board.hash(), next, pos.IsGameOver())});
      e = &parent->children.back();
    }
    return e->node;
  }

  // one playout for candidate cand_ (gumbel.cc:412-452 + Search :674-727).  Returns true if
  // it stopped at an unevaluated leaf (evaluation requested).
  bool StartVisit() {
    Visit& v = ring_[(head_ + count_) % kMaxInflight];
    const int a0 = gm_[cand_].enc;
    ++cand_;   // the loop's cursor moves when a playout is issued; its visit is counted when it completes
    v.is_root = false;
    v.a0 = a0;
    v.game = root_pos_;
    v.game.PlayMove(MoveLoc(a0), color_);
    TreeNode* child = GetOrCreateChild(root_, a0, v.game, Opp(color_));
    v.path.clear();
    v.path.push_back(PathEntry{-1, child});
    Color c = Opp(color_);
    PuctParams pp;
    pp.enable_var_scaling = p_.nonroot_var_scale_prior_visits >= 0;
    pp.var_scale_prior_visits = pp.enable_var_scaling ? p_.nonroot_var_scale_prior_visits : 0;
    while (v.path.back().node->evaluated && !v.path.back().node->is_terminal && !v.game.IsGameOver()) {
      TreeNode* node = v.path.back().node;
      int a = PuctTopMove(node, v.game.board, c, pp);
      if (a < 0) a = kPassEncoding;
      v.game.PlayMove(MoveLoc(a), c);
      TreeNode* nx = GetOrCreateChild(node, a, v.game, Opp(c));
      v.path.back().action = a;
      v.path.push_back(PathEntry{-1, nx});
      c = Opp(c);
    }
    return RequestLeaf(v, c);
  }
  // the descent stopped at v.path.back(): true if that leaf needs the network (the playout stays in flight)
  bool RequestLeaf(Visit& v, Color c) {
    v.leaf_color = c;
    TreeNode* leaf = v.path.back().node;
    if (!leaf->evaluated && !v.game.IsGameOver()) {
      eval_game_ = &v.game;
      eval_color_ = c;
      ++count_;
      return true;
    }
    if (!leaf->evaluated) leaf->evaluated = true;   // terminal leaves carry no policy
    CompleteVisit(v);
    return false;
  }

  // ---- PUCT at the root ---------------------------------------------------------------
  void PreparePuct() {
    pre_visits_.clear();
    for (const ChildEdge& e : root_->children) pre_visits_.push_back({e.action, e.visits});
    visits_spent_ = 0;
    int amax = 0;
    for (int i = 1; i < kNumMoves; ++i)
      if (root_->move_logits[i] > root_->move_logits[amax]) amax = i;
    res_.nn_move = MoveLoc(amax);
  }
  bool StartVisitPuct() {
    Visit& v = ring_[(head_ + count_) % kMaxInflight];
    v.is_root = false;
    v.a0 = -1;
    v.game = root_pos_;
    v.path.clear();
    v.path.push_back(PathEntry{-1, root_});
    Color c = color_;
    bool first = true;
    while (v.path.back().node->evaluated && !v.path.back().node->is_terminal && !v.game.IsGameOver()) {
      TreeNode* node = v.path.back().node;
      int a = PuctTopMove(node, v.game.board, c, puct_pp_, first);
      first = false;
      if (a < 0) a = kPassEncoding;
      v.game.PlayMove(MoveLoc(a), c);
      TreeNode* nx = GetOrCreateChild(node, a, v.game, Opp(c));
      v.path.back().action = a;
      v.path.push_back(PathEntry{-1, nx});
      c = Opp(c);
    }
    return RequestLeaf(v, c);
  }
  void FinishPuct() {
    float counts[kNumMoves] = {};
    float total = 0;
    for (const ChildEdge& e : root_->children) {
      int pre = 0;
      for (auto& pv : pre_visits_)
        if (pv.first == e.action) pre = pv.second;
      counts[e.action] = (float)(e.visits - pre);
      total += counts[e.action];
    }
    int amax = 0;
    for (int a = 0; a < kNumMoves; ++a) {
      res_.pi_improved[a] = total > 0 ? counts[a] / total : 0.0f;
      if (counts[a] > counts[amax]) amax = a;
    }
    int mv = amax;   // gumbel.cc:628-643
    if (puct_pp_.kind == PuctRootSelection::kLcb) {
      std::array<std::pair<int, float>, kNumMoves> lcbs;
      for (int a = 0; a < kNumMoves; ++a) lcbs[a] = {a, Lcb(root_, a)};
      std::stable_sort(lcbs.begin(), lcbs.end(), [](const auto& x, const auto& y) { return y.second < x.second; });
      mv = kPassEncoding;
      for (const auto& al : lcbs)
        if (root_pos_.board.IsValidMove(MoveLoc(al.first), color_)) { mv = al.first; break; }
    } else if (puct_pp_.kind == PuctRootSelection::kVisitCountSample) {
      mv = p_.tau > 0.0f ? amax : SampleFromPolicy(res_.pi_improved, p_.tau);   // (sic) the reference's inverted test
    }
    res_.mcts_move = MoveLoc(mv);
    res_.visits = visits_spent_;
    res_.kld = 0;
  }

  void CompleteVisit(Visit& v) {
    TreeNode* leaf = v.path.back().node;
    if (v.game.IsGameOver() && !leaf->is_terminal) {
      Scores s = v.game.GetScores();
      EvaluateTerminal(s, leaf, v.leaf_color, color_, root_->init_score_est, score_util_);
      leaf->evaluated = true;
    }
    AssignBiasCacheEntry(bias_cache_, v.game, leaf);   // gumbel.cc:683,724
    Backward(v);
    ++visits_spent_;
    if (puct_root_) return;   // the root is part of the path and was updated by Backward
    root_->edge(v.a0)->visits += 1;
  }

  void Backward(const Visit& v) {   // gumbel.cc:738-754
    TreeNode* leaf = v.path.back().node;
    const float lq = leaf->init_util_est, lqo = leaf->init_outcome_est, ls = leaf->init_score_est;
    for (int i = (int)v.path.size() - 1; i >= 0; --i) {
      TreeNode* parent = v.path[i].node;
      const float mult = leaf->color_to_move == parent->color_to_move ? 1.0f : -1.0f;
      SingleBackup(parent, v.path[i].action, i == (int)v.path.size() - 1, mult * lq, mult * lqo, mult * ls);
    }
  }

  void SingleBackup(TreeNode* node, int a, bool is_leaf, float leaf_q, float leaf_qo, float leaf_score) {
    if (is_leaf) {   // gumbel.cc:760-767
      node->n += 1;
      node->w = node->v = node->init_util_est;
      node->w_outcome = node->v_outcome = node->init_outcome_est;
      return;
    }
    const float v_old = node->v, vo_old = node->v_outcome, n_old = (float)node->n;
    node->n += 1;
    ChildEdge* e = node->edge(a);
    e->visits += 1;
    if (bias_cache_ || pool_->is_graph()) {
      // use_idempotent_updates (gumbel.cc:446-448,608-610,770-777): every statistic is recomputed
      // from the children with the cache's observed bias taken off the node's own estimate; the
      // value histogram stays incremental as in the reference
      RecomputeNodeStats(node, UpdateAndFetchObsBias(bias_cache_, node));
      const int b = std::clamp((int)((leaf_qo + 1.0f) / kBucketRange), 0, kNumVBuckets - 1);
      node->v_categorical[b] += 1;
      return;
    }
    node->w += leaf_q;
    node->w_outcome += leaf_qo;
    node->v = node->w / node->n;
    node->v_outcome = node->w_outcome / node->n;
    const float nn = (float)node->n;
    node->score = leaf_score * (1.0f / nn) + node->score * ((nn - 1.0f) / nn);
    auto m3f = [](double m3, double m2, double d, double n) {
      const double d3 = d * d * d;
      return m3 + ((n * n - 1) * d3 / (n * n)) - (3 * d * m2 / n);
    };
    const float m3 = (float)m3f(node->v_m3 * n_old, node->v_var * n_old, leaf_q - node->v, n_old);
    const float m3o = (float)m3f(node->v_outcome_m3 * n_old, node->v_outcome_var * n_old, leaf_qo - node->v_outcome, n_old);
    node->v_m3 = m3 / node->n;
    node->v_outcome_m3 = m3o / node->n;
    node->v_var = (n_old * node->v_var + (leaf_q - v_old) * (leaf_q - node->v)) / node->n;
    node->v_outcome_var = (n_old * node->v_outcome_var + (leaf_qo - vo_old) * (leaf_qo - node->v_outcome)) / node->n;
    if (e->visits > node->max_child_n) node->max_child_n = e->visits;
    int b = std::clamp((int)((leaf_qo + 1.0f) / kBucketRange), 0, kNumVBuckets - 1);
    node->v_categorical[b] += 1;
  }

  float VMixed(const TreeNode* node) const {   // gumbel.cc:68-87
    if (SumChildrenN(node) == 0) return node->init_util_est;
    double wq = 0, vp = 0;
    for (const ChildEdge& e : node->children)
      if (e.visits > 0) { wq += node->move_probs[e.action] * -e.node->v; vp += node->move_probs[e.action]; }
    double iq = wq * SumChildrenN(node) / vp + node->init_util_est;
    return (float)(iq / (1 + SumChildrenN(node)));
  }

  void Finish() {   // gumbel.cc:476-559
    const float max_n = 2 * std::log((float)(theoretical_winner_visits_ + 1));
    auto q_norm = [](float q) { return (q + 1.1f) / 2.2f; };
    const float v_mix = q_norm(VMixed(root_));
    float logits[kNumMoves];
    for (int a = 0; a < kNumMoves; ++a) {
      bool visited = false;
      for (const MoveInfo& t : top_k_) visited |= t.enc == a;
      float q = visited ? q_norm(Q(root_, a)) : v_mix;
      logits[a] = masked_logits_[a] + (kVisit + max_n) * q;
    }
    SoftmaxN(logits, res_.pi_improved, kNumMoves);
    int mcts = gm_[0].enc;
    if (p_.tau > 0.0f) mcts = SampleFromPolicy(res_.pi_improved, p_.tau);
    res_.mcts_move = MoveLoc(mcts);
    // root bookkeeping: all child visits, q only from the chosen move
    int total = 0;
    for (int i = 0; i < m_; ++i) {
      const int a = gm_[i].enc;
      if (a < 0) continue;
      const int cn = root_->child_visits(a);
      total += cn;
      if (a == mcts && cn > 0) {
        const float cq = Q(root_, a), cqz = QOutcome(root_, a);
        root_->w += cn * cq;
        root_->w_outcome += cn * cqz;
        const int tv = root_->n + cn;
        const float rr = root_->n / (float)tv, cr = cn / (float)tv;
        root_->v = rr * root_->v + cr * cq;
        root_->v_outcome = rr * root_->v_outcome + cr * cqz;
        const TreeNode* ch = root_->child(a);
        if (ch && root_->n + cn - 2 > 0) {
          root_->v_outcome_var = ((root_->n - 1) * root_->v_outcome_var + (cn - 1) * ch->v_outcome_var) / (root_->n + cn - 2);
          root_->v_var = ((root_->n - 1) * root_->v_var + (cn - 1) * ch->v_var) / (root_->n + cn - 2);
        }
        if (ch)
          for (int b = 0; b < kNumVBuckets; ++b) root_->v_categorical[b] += ch->v_categorical[kNumVBuckets - b - 1];
      }
    }
    root_->n = total;
    double kld = 0;
    for (int i = 0; i < kNumMoves; ++i)
      if (res_.pi_improved[i] != 0.0f) kld += res_.pi_improved[i] * std::log(res_.pi_improved[i] / (root_->move_probs[i] + 1e-10));
    res_.kld = (float)kld;
    res_.visits = visits_spent_;
  }

  int SampleFromPolicy(const float* policy, float tau) {   // gumbel.cc:103-154
    float tempered[kNumMoves], total = 0;
    for (int a = 0; a < kNumMoves; ++a) { tempered[a] = std::pow(policy[a], 1.0f / tau); total += tempered[a]; }
    const float p = prob_->Uniform();
    float mass = 0;
    int last = -1;
    if (std::isfinite(total) && total > 0)
      for (int a = 0; a < kNumMoves; ++a) {
        if (tempered[a] == 0.0f || !std::isfinite(tempered[a])) continue;
        last = a;
        const float pr = tempered[a] / total;
        if (p >= mass && p < mass + pr) return a;
        mass += pr;
      }
    return last >= 0 ? last : kPassEncoding;
  }

  Game* game_ = nullptr;
  NodePool* pool_ = nullptr;
  TreeNode* root_ = nullptr;
  Color color_ = kBlack;
  GumbelParams p_;
  Probability* prob_ = nullptr;
  State state_ = State::kDone;
  GumbelResult res_;
  MoveInfo gm_[kNumMoves];
  std::vector<MoveInfo> top_k_;
  float masked_logits_[kNumMoves];
  int k_ = 0, m_ = 0, num_rounds_ = 1, visits_per_action_ = 0, visit_num_ = 0, cand_ = 0;
  int theoretical_winner_visits_ = 0;
  uint32_t visits_spent_ = 0;
  bool round_open_ = false;
  Position root_pos_;
  Visit ring_[kMaxInflight];   // playouts in flight, oldest at head_
  int head_ = 0, count_ = 0;
  bool have_pending_ = false;
  bool puct_root_ = false;
  PuctParams puct_pp_;
  ScoreUtilityParams score_util_;   // GumbelEvaluator's ScoreUtilityParams (gumbel.h ctor; eval.cc:170,178)
  std::vector<std::pair<int, int>> pre_visits_;
  const Position* eval_game_ = nullptr;
  Color eval_color_ = kBlack;
  p3hip_result pending_;
  BiasCache* bias_cache_ = nullptr;
};

}  // namespace p3
