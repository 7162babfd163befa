// bias_cache.h — observed-error correction of the network's value estimate, keyed by the local
// pattern around the last move (mcts::BiasCache / LocalPattern, cc/mcts/bias_cache.h:51-206;
// node hooks cc/mcts/tree.h:23-32,83-86; used by GumbelEvaluator at gumbel.cc:277,683,724,
// 729-736,774-777 and by selfplay::Run at self_play_thread.cc:407-412,639-641,730-733).
// On in BASELINE config C1: config/v4.json sets bias_cache_lambda 0.3, alpha 0.8.
//
// A search node whose position has a last move on the board shares one entry (sum of weighted
// observed errors, sum of weights) with every other node of the same local situation: same
// player and point of the last move, same point two moves ago, same 5x5 neighbourhood (stones /
// off-board), same atari map and same illegal-empty-point ("ko") map.  Each backup through a
// node re-derives that node's observed error  init_util_est - (visit-weighted mean of -child.v)
// with weight (child visits)^alpha, replaces its previous contribution to the entry, and
// the node's statistics are recomputed with  lambda * (entry error / entry weight)  subtracted
// from its own value estimate.  A node that dies takes 0.8 of its last contribution with it.
//
// The pattern hashes come from a fixed seed here (the reference seeds its table from the clock,
// bias_cache.h:21 — SURVEY.md section 9): they only have to be consistent within a process.
#pragma once
#include <array>
#include <cmath>
#include <memory>
#include <mutex>
#include <optional>
#include <unordered_map>

#include "board.h"
#include "rng.h"

namespace p3 {

constexpr int kPatternLen = 5;          // bias_cache.h:16
constexpr int kOffBoard = 2;            // grid state beside kBlack (1) / kWhite (-1) / kEmpty (0)

struct BiasEntry {                      // BiasCache::Entry: (weighted L1 error, weighted visits)
  float err = 0, weight = 0;
};

class LocalPatternZobrist {             // bias_cache.h:18-49
 public:
  LocalPatternZobrist() {
    PRng prng(0x6c6f63616c706174ull, 0x7465726e7a6f6272ull, 0x6973745f70336163ull, 0x6879676f5f616d64ull);
    for (int i = 0; i < kPatternLen; ++i)
      for (int j = 0; j < kPatternLen; ++j)
        for (int k = 1; k < 4; ++k) table_[i][j][k] = prng.next64();   // EMPTY stays 0
  }
  // state: 0 empty, 1 black (or a set flag), 2 off board, -1 white
  uint64_t hash_at(int i, int j, int state) const {
    if (state == 0) return 0;
    return table_[i][j][state == -1 ? 3 : state];
  }
  static const LocalPatternZobrist& get() {
    static const LocalPatternZobrist z;
    return z;
  }

 private:
  uint64_t table_[kPatternLen][kPatternLen][4] = {};
};

struct LocalPattern {                   // bias_cache.h:51-124
  uint64_t grid_hash = 0, atari_hash = 0, ko_hash = 0;
  std::array<int8_t, kPatternLen * kPatternLen> grid{}, atari{}, ko{};
  Move last_move{kEmpty, kNoopLoc}, two_moves_ago{kEmpty, kNoopLoc};

  // nullopt unless the last move is a stone on the board and a move exists two plies back
  static std::optional<LocalPattern> FromCurrentPosition(const Position& pos) {
    constexpr int kOff = kPatternLen / 2;
    LocalPattern p;
    p.last_move = pos.last[4];
    p.two_moves_ago = pos.last[3];
    const Loc last = p.last_move.loc;
    if (p.two_moves_ago.loc == kNoopLoc || last == kNoopLoc || last == kPassLoc) return std::nullopt;
    const Grid& stones = pos.board.position();
    for (int gi = 0; gi < kPatternLen; ++gi)
      for (int gj = 0; gj < kPatternLen; ++gj) {
        const int i = last.i + gi - kOff, j = last.j + gj - kOff, k = gi * kPatternLen + gj;
        if (i < 0 || i >= kBoardLen || j < 0 || j >= kBoardLen) {
          p.grid[k] = kOffBoard;
          continue;
        }
        const int idx = i * kBoardLen + j;
        p.grid[k] = stones[idx];
        if (stones[idx] == kEmpty && !pos.board.IsValidMove(Loc{i, j}, Opp(p.last_move.color))) p.ko[k] = 1;
        if (pos.board.LibertiesAt(idx) == 1) p.atari[k] = 1;   // Board::IsInAtari
      }
    const LocalPatternZobrist& z = LocalPatternZobrist::get();
    for (int gi = 0; gi < kPatternLen; ++gi)
      for (int gj = 0; gj < kPatternLen; ++gj) {
        const int k = gi * kPatternLen + gj;
        p.grid_hash ^= z.hash_at(gi, gj, p.grid[k]);
        p.atari_hash ^= z.hash_at(gi, gj, p.atari[k]);
        p.ko_hash ^= z.hash_at(gi, gj, p.ko[k]);
      }
    return p;
  }
};

// The slice of a search node the cache reads and writes (TreeNode carries these fields).
struct BiasNodeState {
  std::shared_ptr<BiasEntry> bias_cache_entry;   // tree.h:83-86
  float last_obs_bias_term = 0, last_weight_term = 0;
  // TreeNode::~TreeNode, tree.h:23-32: a dying node takes 0.8 of its last contribution with it
  void Release() {
    constexpr float kBiasCacheForgetWeight = 0.8f;
    if (bias_cache_entry) {
      bias_cache_entry->err -= kBiasCacheForgetWeight * last_obs_bias_term;
      bias_cache_entry->weight -= kBiasCacheForgetWeight * last_weight_term;
      bias_cache_entry.reset();
    }
    last_obs_bias_term = last_weight_term = 0;
  }
};

class BiasCache {                       // bias_cache.h:129-206
 public:
  explicit BiasCache(float alpha = 0.8f, float lambda = 0.4f) : alpha_(alpha), lambda_(lambda) {}
  BiasCache(const BiasCache&) = delete;
  BiasCache& operator=(const BiasCache&) = delete;

  std::shared_ptr<BiasEntry> GetOrCreate(const LocalPattern& p) {
    std::lock_guard<std::mutex> l(mu_);
    const Key key{p.last_move.color, p.last_move.loc, p.two_moves_ago.loc, p.grid_hash, p.atari_hash, p.ko_hash};
    auto it = cache_.find(key);
    if (it != cache_.end()) return it->second;
    auto e = std::make_shared<BiasEntry>();
    cache_.emplace(key, e);
    return e;
  }

  // Replaces the node's contribution to its entry by its current observed error and returns the
  // entry's weighted bias.  `node` is fully updated (n and child visits already incremented);
  // `weighted_child_utility` = sum over visited children of visits * -(child v), `child_visits`
  // = n - 1.
  float UpdateAndFetch(BiasNodeState& st, float init_util_est, float weighted_child_utility, int child_visits) {
    std::lock_guard<std::mutex> l(mu_);
    const float obs_err = init_util_est - weighted_child_utility / child_visits;
    const float weight_term = std::pow((float)child_visits, alpha_);
    const float obs_bias_term = obs_err * weight_term;
    BiasEntry& e = *st.bias_cache_entry;
    e.err += obs_bias_term - st.last_obs_bias_term;
    e.weight += weight_term - st.last_weight_term;
    st.last_weight_term = weight_term;
    st.last_obs_bias_term = obs_bias_term;
    return lambda_ * (e.err / e.weight);
  }

  float Fetch(const BiasNodeState& st) const {   // read-only
    std::lock_guard<std::mutex> l(mu_);
    const auto& e = st.bias_cache_entry;
    if (!e || e->weight == 0.0f) return 0.0f;
    return lambda_ * (e->err / e->weight);
  }

  // drops the entries no live node refers to; returns how many
  uint32_t PruneUnused() {
    std::lock_guard<std::mutex> l(mu_);
    uint32_t n = 0;
    for (auto it = cache_.begin(); it != cache_.end();)
      if (it->second.use_count() <= 1) { it = cache_.erase(it); ++n; }
      else ++it;
    return n;
  }
  size_t size() const {
    std::lock_guard<std::mutex> l(mu_);
    return cache_.size();
  }
  float alpha() const { return alpha_; }
  float lambda() const { return lambda_; }

 private:
  struct Key {
    Color color;
    Loc last, two_ago;
    uint64_t grid, atari, ko;
    bool operator==(const Key& o) const {
      return color == o.color && last == o.last && two_ago == o.two_ago && grid == o.grid && atari == o.atari && ko == o.ko;
    }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const {
      uint64_t h = k.grid ^ (k.atari * 0x9e3779b97f4a7c15ull) ^ (k.ko * 0xc2b2ae3d27d4eb4full);
      h ^= (uint64_t)(uint32_t)(k.last.i * 32 + k.last.j + 64) * 0xff51afd7ed558ccdull;
      h ^= (uint64_t)(uint32_t)(k.two_ago.i * 32 + k.two_ago.j + 64) * 0xc4ceb9fe1a85ec53ull;
      h ^= (uint64_t)(uint8_t)k.color << 56;
      return (size_t)(h ^ (h >> 31));
    }
  };
  std::unordered_map<Key, std::shared_ptr<BiasEntry>, KeyHash> cache_;
  const float alpha_, lambda_;
  mutable std::mutex mu_;
};

}  // namespace p3
