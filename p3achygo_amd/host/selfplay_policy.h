// selfplay_policy.h — the per-game policies of the self-play loop that sit around the search:
// initial-state sampling (GoExploit reuse buffer, handicap games, komi noise), the training
// move selection multiplier and the fork manager that feeds the reuse buffer.
//
// Restates cc/selfplay/{reuse_buffer.h, move_sel_manager.h, fork_manager.h} and
// GetInitState (self_play_thread.cc:202-252).  One structural difference: the reference's
// ForkManager evaluates candidate positions with blocking SearchRoot(n=1) calls
// (fork_manager.h:497-510); here a game never blocks (thousands of games share a handful of
// host threads), so the manager is a resumable task that asks for one network evaluation at
// a time (NextEval / Deliver) and the game loop feeds it through the same batch as the
// search leaves.
#pragma once
#include <algorithm>
#include <cmath>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <vector>

#include "board.h"
#include "rng.h"
#include "search.h"

namespace p3 {

constexpr int kGoExploitBufferSize = 8192;   // constants.h:84
constexpr Move kNoopMove{kEmpty, kNoopLoc};

enum class FirstMoveBehavior : uint8_t { kSample = 0, kPlay = 1, kForceFullSearch = 2 };   // reuse_buffer.h:19-25
enum class ForkKind : uint8_t { kEarly, kLate, kSampleT1, kSampleT2, kSampleUniform, kRegret, kUniform };

struct InitState {   // reuse_buffer.h:27-42
  enum class Kind : uint8_t { kEmpty = 0, kBook = 1, kHandicap = 2, kGoExploit = 3, kRegret = 4 };
  Board board;
  Move last_moves[5] = {kNoopMove, kNoopMove, kNoopMove, kNoopMove, kNoopMove};
  Color color_to_move = kBlack;
  int move_num = 0;
  FirstMoveBehavior first_move_behavior = FirstMoveBehavior::kSample;
  Kind kind = Kind::kEmpty;
  int fork_kind = -1;   // ForkKind, or -1
};

// GoExploitReuseBuffer (reuse_buffer.h:56-76) over core::RingBuffer (ring_buffer.h:16-61):
// append overwrites the oldest entry when full; PopRandom swaps a uniformly drawn entry to
// the front and pops it.  The reference seeds the buffer's PRng from the clock; here it is
// seeded from the run's root seed.
class ReuseBuffer {
 public:
  explicit ReuseBuffer(uint64_t seed = 0x676f6578ull) : prng_(seed) {}
  void Add(const InitState& s) {
    std::lock_guard<std::mutex> l(mu_);
    const int idx = (start_ + size_) % kGoExploitBufferSize;
    buf_[idx] = std::make_unique<InitState>(s);
    if (size_ == kGoExploitBufferSize) start_ = (start_ + 1) % kGoExploitBufferSize;
    else ++size_;
    ++added_;
  }
  std::optional<InitState> Get() {
    std::lock_guard<std::mutex> l(mu_);
    if (size_ == 0 || !buf_[start_]) return std::nullopt;
    const int idx = (start_ + RandRange(prng_, 0, size_)) % kGoExploitBufferSize;
    std::swap(buf_[start_], buf_[idx]);
    InitState e = *buf_[start_];
    buf_[start_].reset();
    start_ = (start_ + 1) % kGoExploitBufferSize;
    --size_;
    return e;
  }
  int size() const { return size_; }
  long added() const { return added_; }

 private:
  std::mutex mu_;
  std::unique_ptr<InitState> buf_[kGoExploitBufferSize];
  int start_ = 0, size_ = 0;
  long added_ = 0;
  PRng prng_;
};

constexpr float kPlayFromBookProb = 0.0f;   // self_play_thread.cc:50 (the opening book is never used)
constexpr float kHandicapGameProb = 0.05f;  // :53

// The opening book (cc/selfplay/book.h: six four-move openings; data).  Line 3 repeats a point:
// the second stone there is an illegal move that the reference plays unchecked (the board
// rejects it, the move list keeps it) — kept as is.
constexpr int kOpeningBook[6][4][2] = {
    {{3, 3}, {15, 15}, {15, 4}, {4, 15}}, {{3, 3}, {15, 15}, {16, 4}, {4, 15}}, {{3, 3}, {15, 4}, {15, 16}, {15, 4}},
    {{3, 3}, {15, 4}, {15, 15}, {4, 15}}, {{3, 3}, {15, 15}, {2, 15}, {15, 15}}, {{3, 3}, {15, 15}, {2, 15}, {16, 15}}};

// A start from the book (self_play_thread.cc:216-233): a random line, a random prefix of 0..4 moves
// (round of a uniform draw), played from the empty board at the sampled komi.
inline InitState BookInitState(Probability& prob, InitState s0) {
  const int index = (int)(prob.Uniform() * 6);
  const int num_moves = (int)std::round(prob.Uniform() * 4);
  s0.kind = InitState::Kind::kBook;
  for (int i = 0; i < 5; ++i) s0.last_moves[i] = kNoopMove;
  for (int i = 0; i < num_moves; ++i) {
    const Loc loc{kOpeningBook[index][i][0], kOpeningBook[index][i][1]};
    (void)s0.board.PlayMove(loc, s0.color_to_move);
    s0.last_moves[5 - num_moves + i] = Move{s0.color_to_move, loc};
    s0.color_to_move = Opp(s0.color_to_move);
  }
  return s0;
}

// self_play_thread.cc:202-252
inline InitState GetInitState(Probability& prob, ReuseBuffer* buffer, float use_seen_state_prob) {
  const float komi = std::round(7.0f + std::clamp(prob.Gaussian(), -3.0f, 3.0f)) + (prob.Uniform() < 0.5f ? -0.5f : 0.5f);
  InitState s0;
  s0.board = Board(komi, true);
  const float p = prob.Uniform();
  if (p <= kPlayFromBookProb) {
    // kPlayFromBookProb = 0: taken only when the uniform draw is exactly 0 (probability 2^-23)
    return BookInitState(prob, s0);
  } else if (p <= kPlayFromBookProb + kHandicapGameProb) {
    const int handicap = (int)std::floor(prob.Uniform() * 3 + 2);
    const float hkomi = (handicap - 2) * 14 + 20.5f;
    InitState s;
    s.board = Board(handicap, hkomi);
    s.color_to_move = kWhite;
    s.kind = InitState::Kind::kHandicap;
    return s;
  } else if (p <= kPlayFromBookProb + kHandicapGameProb + use_seen_state_prob) {
    std::optional<InitState> seen = buffer ? buffer->Get() : std::nullopt;
    if (!seen) return s0;
    seen->kind = InitState::Kind::kGoExploit;
    return *seen;
  }
  return s0;
}

// ---- training move selection multiplier (move_sel_manager.h) --------------------------------
struct SelMultCalibration {   // self_play_thread.h:23-36
  std::map<std::string, float> v_outcome_stddev, v_outcome_stddev_adj, pre_kld, nn_mcts_diff;
  std::map<int, float> expected_std_by_n;
  static float get(const std::map<std::string, float>& m, const std::string& pct, float def) {
    auto it = m.find(pct);
    return it != m.end() ? it->second : def;
  }
};

// ParseCalibrationFile (selfplay/main.cc:71-118): "field.percentile=value" lines written by the RL loop from
// the .stats files (fields v_outcome_stddev, v_outcome_stddev_adj, pre_kld, nn_mcts_diff, and
// expected_std.n<visit bin>); '#' comments, lines without '=' or '.', and unknown fields are skipped; a
// missing file leaves the defaults.
inline SelMultCalibration ParseCalibrationFile(const std::string& path) {
  SelMultCalibration calib;
  if (path.empty()) return calib;
  std::ifstream f(path);
  if (!f.is_open()) return calib;
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#') continue;
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    const std::string key = line.substr(0, eq);
    float val;
    try { val = std::stof(line.substr(eq + 1)); } catch (...) { continue; }
    const size_t dot = key.find('.');
    if (dot == std::string::npos) continue;
    const std::string field = key.substr(0, dot), pct = key.substr(dot + 1);
    if (field == "v_outcome_stddev") calib.v_outcome_stddev[pct] = val;
    else if (field == "v_outcome_stddev_adj") calib.v_outcome_stddev_adj[pct] = val;
    else if (field == "pre_kld") calib.pre_kld[pct] = val;
    else if (field == "nn_mcts_diff") calib.nn_mcts_diff[pct] = val;
    else if (field == "expected_std" && pct.size() > 1 && pct[0] == 'n') {
      try { calib.expected_std_by_n[std::stoi(pct.substr(1))] = val; } catch (...) {}
    }
  }
  return calib;
}

enum MoveSelFlags : uint32_t { kStddevBonus = 1, kStddevPenalty = 2, kKldBonus = 4, kKldPenalty = 8, kNnMctsBonus = 16 };

struct MoveSelResult {
  float modifier, modifier_unscaled, sel_bonus, sel_penalty, sel_std_bonus, sel_std_penalty, sel_kld_bonus,
      sel_kld_penalty, sel_nn_mcts_bonus, sel_q_adjust, std_adj, std_adj_att;
};

class MoveSelManager {
 public:
  MoveSelManager(uint32_t flags, const SelMultCalibration& c) : flags_(flags), c_(c) {}
  MoveSelResult Compute(int n_pre, float std_dev, float pre_kld, float nn_mcts_diff, float q_canonical,
                        float scale_factor) const {   // move_sel_manager.h:41-77
    const float std_adj = StdAdj(n_pre, std_dev);
    const float std_adj_att = StdAdjAtt(n_pre, std_adj);
    const float q_adj = SelQAdjust(q_canonical);
    const float sb = StdBonus(std_adj_att), sp = StdPenalty(std_adj_att), kb = KldBonus(pre_kld),
                kp = KldPenalty(pre_kld), nb = NnMctsBonus(nn_mcts_diff);
    const float raw_bonus = std::min(std::max({flags_ & kStddevBonus ? sb : 1.0f, flags_ & kKldBonus ? kb : 1.0f,
                                               flags_ & kNnMctsBonus ? nb : 1.0f}), 2.5f);
    const float raw_penalty = std::min(flags_ & kStddevPenalty ? sp : 1.0f, flags_ & kKldPenalty ? kp : 1.0f);
    const float bonus = 1.0f + q_adj * (raw_bonus - 1.0f), penalty = 1.0f + q_adj * (raw_penalty - 1.0f);
    const float unscaled = bonus * penalty;
    return MoveSelResult{1.0f + scale_factor * (unscaled - 1.0f), unscaled, bonus, penalty, sb, sp, kb, kp, nb, q_adj,
                         std_adj, std_adj_att};
  }

 private:
  float StdAdj(int n_pre, float std_dev) const {   // :80-111
    if (std_dev == 0.0f || c_.expected_std_by_n.empty()) return 0.0f;
    const int query = std::min((n_pre / 5) * 5, 200);
    std::vector<std::pair<int, float>> nb;
    for (const auto& kv : c_.expected_std_by_n)
      if (kv.second > 0.0f) nb.push_back(kv);
    std::stable_sort(nb.begin(), nb.end(), [&](const auto& a, const auto& b) {
      return std::abs(a.first - query) < std::abs(b.first - query);
    });
    const int k = std::min(4, (int)nb.size());
    float sw = 0, swv = 0;
    for (int i = 0; i < k; ++i) {
      const float w = 1.0f / (std::abs(nb[i].first - query) + 5.0f);
      sw += w;
      swv += w * nb[i].second;
    }
    const float expected = swv / sw;
    return expected > 0.0f ? std_dev / expected : 0.0f;
  }
  float StdAdjAtt(int n_pre, float sa) const {   // :116-121
    if (sa == 0.0f) return 0.0f;
    const float att = std::min(1.0f, 0.2f + 0.8f * std::pow(n_pre / 40.0f, 0.54f));
    return 1.0f + (sa - 1.0f) * att;
  }
  float SelQAdjust(float q) const {   // :125-129
    return std::pow(1.0f - std::clamp((std::abs(q) - 0.5f) / 0.4f, 0.0f, 1.0f), 0.4f);
  }
  float StdBonus(float sa) const {   // :131-139
    if (sa == 0.0f) return 1.0f;
    const float lb = c_.get(c_.v_outcome_stddev_adj, "p80", 1.52f), ub = c_.get(c_.v_outcome_stddev_adj, "p99", 4.96f);
    if (sa <= lb || ub <= lb) return 1.0f;
    return 1.0f + 0.5f * (sa - lb) / (ub - lb);
  }
  float StdPenalty(float sa) const {   // :141-151
    if (sa == 0.0f) return 1.0f;
    const float lb = c_.get(c_.v_outcome_stddev_adj, "p01", 0.02f), ub = c_.get(c_.v_outcome_stddev_adj, "p50", 0.64f);
    if (sa >= ub) return 1.0f;
    if (sa <= lb || ub <= lb) return 0.3f;
    return 1.0f - 0.7f * (ub - sa) / (ub - lb);
  }
  float KldBonus(float k) const {   // :153-158
    const float lb = c_.get(c_.pre_kld, "p70", 0.310f), ub = c_.get(c_.pre_kld, "p95", 1.166f);
    if (k == 0.0f || k <= lb || ub <= lb) return 1.0f;
    return std::min(1.5f, 1.0f + 0.5f * (k - lb) / (ub - lb));
  }
  float KldPenalty(float k) const {   // :160-167
    const float lb = c_.get(c_.pre_kld, "p05", 0.0001f), ub = 0.06f;
    if (k == 0.0f || k >= ub) return 1.0f;
    if (k <= lb || ub <= lb) return 0.3f;
    return 1.0f - 0.7f * (ub - k) / (ub - lb);
  }
  float NnMctsBonus(float d) const {   // :171-179
    if (d == 0.0f) return 1.0f;
    const float lb = c_.get(c_.nn_mcts_diff, "p70", 0.1463f), ub = c_.get(c_.nn_mcts_diff, "p99", 0.6500f);
    if (d <= lb || ub <= lb) return 1.0f;
    return 1.0f + 0.60f * (d - lb) / (ub - lb);
  }
  uint32_t flags_;
  SelMultCalibration c_;
};

// ---- fork manager (fork_manager.h) --------------------------------------------------------------
struct ForkParams {   // ForkManager::Params, fork_manager.h:41-92
  static constexpr float kBase[5] = {0.0f, 0.09f, 0.0f, 0.0f, 0.01f};
  float early = kBase[0], late = kBase[1], t1 = kBase[2], t2 = kBase[3], random = kBase[4], regret = 0.0f,
        uniform = 1.0f - (kBase[0] + kBase[1] + kBase[2] + kBase[3] + kBase[4]);
  float force_full_search_prob = 0.25f, double_sample_prob = 0.5f;
  static ForkParams ForReuse(float reuse_prob) {
    const float scale = reuse_prob == 0 ? 0 : 0.2f / reuse_prob;
    ForkParams p;
    p.early = kBase[0] * scale; p.late = kBase[1] * scale; p.t1 = kBase[2] * scale; p.t2 = kBase[3] * scale;
    p.random = kBase[4] * scale;
    float sum = p.early + p.late + p.t1 + p.t2 + p.random;
    if (sum >= 1.0f) {
      const float d = 0.9f / sum;
      p.early *= d; p.late *= d; p.t1 *= d; p.t2 *= d; p.random *= d;
      sum = 0.9f;
    }
    p.regret = 0.0f;
    p.uniform = 1.0f - sum;
    return p;
  }
};

class ForkManager {
 public:
  struct MoveData {   // fork_manager.h:95-103
    const Board* board;   // before the move
    Color color;
    Loc move;
    float nn_value, mcts_value, mcts_score;
    bool is_eligible;
  };

  ForkManager(const ForkParams& params, ReuseBuffer* buffer, Probability& prob, bool started_from_forced_search)
      : p_(params), buffer_(buffer), forced_(started_from_forced_search) {   // :105-167
    auto trapezoid = [&]() {
      constexpr int kFlatStart = 10, kFlatEnd = 100, kMax = 250;
      constexpr float kFlatMass = 0.6f, kFlatDensity = kFlatMass / (kFlatEnd - kFlatStart);
      constexpr float kTail0 = 2.0f * (1.0f - kFlatMass) / (kMax - kFlatEnd), kSlope = kTail0 / (kMax - kFlatEnd);
      const float u = prob.Uniform();
      float cum = 0;
      for (int mv = kFlatStart; mv < kMax; ++mv) {
        cum += mv < kFlatEnd ? kFlatDensity : kTail0 - kSlope * (mv - kFlatEnd);
        if (u <= cum) return mv;
      }
      return kMax;
    };
    const float u = prob.Uniform();
    float cum = p_.early;
    if (u < cum) { kind_ = ForkKind::kEarly; fork_mv_ = (int)std::round(prob.Exponential() * 9); }
    else if (u < (cum += p_.late)) { kind_ = ForkKind::kLate; fork_mv_ = trapezoid(); }
    else if (u < (cum += p_.t1)) { kind_ = ForkKind::kSampleT1; fork_mv_ = trapezoid(); }
    else if (u < (cum += p_.t2)) { kind_ = ForkKind::kSampleT2; fork_mv_ = trapezoid(); }
    else if (u < (cum += p_.random)) { kind_ = ForkKind::kSampleUniform; fork_mv_ = trapezoid(); }
    else if (u < (cum += p_.regret)) { kind_ = ForkKind::kRegret; fork_mv_ = -1; }
    else { kind_ = ForkKind::kUniform; fork_mv_ = trapezoid(); }
  }

  ForkKind kind() const { return kind_; }
  int fork_move_num() const { return fork_mv_; }

  // Called on every move before it is played (fork_manager.h:169-385).  Returns true when the
  // manager has started a fork that needs network evaluations: drive it with NextEval/Deliver
  // until NextEval returns false.
  bool MaybeFork(const Game& game, const MoveData& d, Probability& prob) {
    if (forced_ || game.IsGameOver()) return false;
    const int move_num = game.num_moves();
    if (kind_ == ForkKind::kUniform) {
      const float atten = 1.0f - std::clamp((std::abs(d.mcts_value) - 0.5f) / 0.4f, 0.0f, 1.0f);
      if (prob.Uniform() > 0.05f * atten) return false;
      InitState s;
      s.board = *d.board;
      BuildLastMoves(game, move_num, nullptr, 0, s.last_moves);
      const float komi_delta = KomiDelta(d.mcts_score, d.color);
      const float p_adjust = std::atan(std::abs(d.mcts_score) / 3.0) * M_2_PI;
      if (prob.Uniform() < p_adjust) s.board.SetKomi(d.board->komi() + komi_delta);
      s.color_to_move = d.color;
      s.move_num = move_num;
      s.first_move_behavior = FirstMoveBehavior::kSample;
      s.kind = InitState::Kind::kGoExploit;
      s.fork_kind = (int)kind_;
      sampled_.push_back(s);
      return false;
    }
    if (kind_ == ForkKind::kRegret) {
      regret_.push_back(RegretEntry{d.color, *d.board, d.move, d.nn_value, d.mcts_value, d.is_eligible});
      return false;
    }
    if (did_fork_ || move_num != fork_mv_) return false;
    did_fork_ = true;
    num_candidates_ = kind_ == ForkKind::kEarly ? RandRange(prob.prng(), 3, 13)
                      : kind_ == ForkKind::kLate ? RandRange(prob.prng(), 5, 37) : 0;
    prob_ = &prob;
    color_ = d.color;
    origin_ = *d.board;
    move_num_ = move_num;
    for (int i = 0; i < 5; ++i) hist_[i] = game.moves()[game.moves().size() - 5 + i];   // game.move(move_num - 5 + i)
    BuildLastMoves(game, move_num, nullptr, 0, cur_last_);
    fork_board_ = origin_;
    stage_ = Stage::kAlt1;
    BeginSample(origin_, color_, cur_last_);
    Advance();
    return stage_ != Stage::kIdle;
  }

  bool NextEval(Position* pos, Color* color) {
    if (stage_ == Stage::kIdle || !want_eval_) return false;
    pos->board = eval_board_;
    for (int i = 0; i < 5; ++i) pos->last[i] = eval_last_[i];
    pos->num_moves = move_num_;
    *color = eval_color_;
    return true;
  }

  void Deliver(const p3hip_result& r) {
    want_eval_ = false;
    // EvalBoard (fork_manager.h:497-510): root evaluation only — utility = outcome estimate
    TreeNode tmp;
    EvaluateRoot(r, &tmp, eval_color_);
    if (stage_ == Stage::kAdjKomi) {
      const float fork_score = adj_same_side_ ? tmp.init_score_est : -tmp.init_score_est;
      const float adj = origin_.komi() + KomiDelta(fork_score, color_);
      if (adj_always_ || prob_->Uniform() < 0.5f) fork_board_.SetKomi(adj);
      Emit();
      stage_ = Stage::kIdle;
      return;
    }
    // sampler evaluation
    if (mode_ == Mode::kBestOfN) {
      if (tmp.init_util_est < best_util_) { best_util_ = tmp.init_util_est; best_ = MoveLoc(cands_[ci_]); }
      ++ci_;
    } else {   // policy sampling
      std::vector<float> w;
      float sum = 0;
      for (int a : cands_) {
        const float x = mode_ == Mode::kPolicySqrt ? std::sqrt(tmp.move_probs[a]) : tmp.move_probs[a];
        w.push_back(x);
        sum += x;
      }
      if (sum <= 0.0f) {
        best_ = MoveLoc(cands_[RandRange(prob_->prng(), 0, (int)cands_.size())]);
      } else {
        const float target = prob_->Uniform() * sum;
        float cum = 0;
        best_ = MoveLoc(cands_.back());
        for (size_t i = 0; i < w.size(); ++i) {
          cum += w[i];
          if (target <= cum) { best_ = MoveLoc(cands_[i]); break; }
        }
      }
      sample_done_ = true;
    }
    Advance();
  }

  // fork_manager.h:387-483
  void FinalizeGame(const Game& game, Probability& prob) {
    if (kind_ == ForkKind::kUniform) {
      if (sampled_.empty()) return;
      buffer_->Add(sampled_[RandRange(prob.prng(), 0, (int)sampled_.size())]);
      sampled_.clear();
    }
    if (kind_ != ForkKind::kRegret || forced_ || regret_.empty()) return;
    if ((int)regret_.size() != game.num_moves()) return;
    constexpr float kDecay = 0.94f;
    constexpr int kHorizon = 50;
    float best_score = -1;
    int best = -1;
    const int n = (int)regret_.size();
    for (int m = 0; m < n; ++m) {
      const RegretEntry& e = regret_[m];
      if (!e.eligible) continue;
      const float outcome = game.result().winner == e.color ? 1.5f : -1.5f;
      float ema = 0, w = 1, ws = 0;
      for (int k = 1; k < kHorizon && m + k < n; ++k) {
        const RegretEntry& f = regret_[m + k];
        w *= kDecay;
        if (!f.eligible) continue;
        ema += w * (f.color == e.color ? f.mcts_value : -f.mcts_value);
        ws += w;
      }
      if (ws > 0) ema /= ws;
      const float smoothed = (e.mcts_value + ema * kDecay) / (1.0f + kDecay);
      const float nn_mis = std::abs(e.nn_value - smoothed), drift = std::abs(e.mcts_value - ema);
      const float verr = std::max(smoothed - outcome - std::abs(outcome), 0.0f);
      const float score = nn_mis * nn_mis + drift * drift + verr * verr;
      const float av = std::abs(e.mcts_value);
      const float wr_w = av > 0.9f ? 0.0f : (av <= 0.5f ? 1.0f : (0.9f - av) / 0.4f);
      const float off = (float)std::clamp(game.init_mv_num() + m - 100, 0, 100);
      const float mv_w = (float)std::clamp(std::pow(1.0f - off / 100, 1.2), 0.0, 1.0);
      if (prob.Uniform() >= wr_w * mv_w) continue;
      if (score > best_score) { best_score = score; best = m; }   // top of the max-heap
    }
    if (best < 0) return;
    InitState s;
    s.board = regret_[best].board;
    for (int off = 5; off > 0; --off) s.last_moves[5 - off] = game.moves()[best - off + Game::kMoveOffset];
    s.color_to_move = regret_[best].color;
    s.move_num = best;
    s.first_move_behavior = prob.Uniform() < p_.force_full_search_prob ? FirstMoveBehavior::kForceFullSearch
                                                                      : FirstMoveBehavior::kSample;
    s.kind = InitState::Kind::kGoExploit;
    s.fork_kind = (int)kind_;
    buffer_->Add(s);
  }

  static float KomiDelta(float fork_score, Color color) { return std::round(color == kBlack ? fork_score : -fork_score); }

 private:
  enum class Stage { kIdle, kAlt1, kAlt2, kAdjKomi };
  enum class Mode { kBestOfN, kPolicy, kPolicySqrt, kUniform };
  struct RegretEntry { Color color; Board board; Loc move; float nn_value, mcts_value; bool eligible; };

  // the (5 - n_extra) most recent game moves ending at move_num, then `extra`
  static void BuildLastMoves(const Game& game, int move_num, const Move* extra, int n_extra, Move out[5]) {
    int o = 0;
    for (int off = 5 - n_extra; off > 0; --off) out[o++] = game.moves()[move_num - off + Game::kMoveOffset];
    for (int i = 0; i < n_extra; ++i) out[o++] = extra[i];
  }
  void Shifted(const Move last[5], Move m, Move out[5]) {
    for (int i = 0; i < 4; ++i) out[i] = last[i + 1];
    out[4] = m;
  }

  // starts sampling an alternative move for `color` on `board` (fork_manager.h:228-337)
  void BeginSample(const Board& board, Color color, const Move last[5]) {
    sample_board_ = board;
    sample_color_ = color;
    for (int i = 0; i < 5; ++i) sample_last_[i] = last[i];
    cands_.clear();
    for (int a = 0; a < kNumMoves; ++a)
      if (board.IsValidMove(MoveLoc(a), color)) cands_.push_back(a);
    best_ = kNoopLoc;
    sample_done_ = false;
    ci_ = 0;
    switch (kind_) {
      case ForkKind::kEarly: case ForkKind::kLate: mode_ = Mode::kBestOfN; break;
      case ForkKind::kSampleT1: mode_ = Mode::kPolicy; break;
      case ForkKind::kSampleT2: mode_ = Mode::kPolicySqrt; break;
      default: mode_ = Mode::kUniform; break;
    }
    if (cands_.empty()) { sample_done_ = true; return; }
    if (mode_ == Mode::kBestOfN) {
      take_ = std::min(num_candidates_, (int)cands_.size());
      for (int i = 0; i < take_; ++i) {   // partial Fisher-Yates
        const int j = i + RandRange(prob_->prng(), 0, (int)cands_.size() - i);
        std::swap(cands_[i], cands_[j]);
      }
      best_util_ = std::numeric_limits<float>::max();
    } else if (mode_ == Mode::kUniform) {
      best_ = MoveLoc(cands_[RandRange(prob_->prng(), 0, (int)cands_.size())]);
      sample_done_ = true;
    }
  }

  // sets up the next evaluation of the running sampler; false when the sampler is finished
  bool SampleWantsEval() {
    if (sample_done_) return false;
    if (mode_ == Mode::kBestOfN) {
      if (ci_ >= take_) { sample_done_ = true; return false; }
      const Loc cand = MoveLoc(cands_[ci_]);
      eval_board_ = sample_board_;
      eval_board_.PlayMove(cand, sample_color_);
      Shifted(sample_last_, Move{sample_color_, cand}, eval_last_);
      eval_color_ = Opp(sample_color_);
    } else {
      eval_board_ = sample_board_;
      for (int i = 0; i < 5; ++i) eval_last_[i] = sample_last_[i];
      eval_color_ = sample_color_;
    }
    want_eval_ = true;
    return true;
  }

  void StartAdjKomi(Color fork_color, bool always) {   // ComputeAdjKomi, fork_manager.h:520-533
    stage_ = Stage::kAdjKomi;
    eval_board_ = fork_board_;
    for (int i = 0; i < 5; ++i) eval_last_[i] = cur_last_[i];
    eval_color_ = fork_color;
    adj_same_side_ = fork_color == color_;
    adj_always_ = always;
    want_eval_ = true;
  }

  void Advance() {
    for (;;) {
      if (stage_ == Stage::kAlt1) {
        if (SampleWantsEval()) return;
        if (best_ == kNoopLoc) { stage_ = Stage::kIdle; return; }
        alt1_ = best_;
        fork_board_.PlayMove(alt1_, color_);
        fmb_ = (kind_ == ForkKind::kSampleUniform || prob_->Uniform() < p_.force_full_search_prob)
                   ? FirstMoveBehavior::kForceFullSearch : FirstMoveBehavior::kPlay;
        Move m1{color_, alt1_};
        for (int i = 0; i < 4; ++i) cur_last_[i] = hist_[i + 1];
        cur_last_[4] = m1;
        n_extra_ = 1;
        if (prob_->Uniform() < p_.double_sample_prob) {
          stage_ = Stage::kAlt2;
          BeginSample(fork_board_, Opp(color_), cur_last_);
          continue;
        }
        StartAdjKomi(Opp(color_), /*always=*/true);
        return;
      }
      if (stage_ == Stage::kAlt2) {
        if (SampleWantsEval()) return;
        if (best_ != kNoopLoc) {
          alt2_ = best_;
          fork_board_.PlayMove(alt2_, Opp(color_));
          Move m2{Opp(color_), alt2_};
          Move prev[5];
          for (int i = 0; i < 5; ++i) prev[i] = cur_last_[i];
          Shifted(prev, m2, cur_last_);
          n_extra_ = 2;
          StartAdjKomi(color_, /*always=*/false);
          return;
        }
        StartAdjKomi(Opp(color_), /*always=*/true);   // single-sample fallback: add P'
        return;
      }
      return;
    }
  }

  void Emit() {
    InitState s;
    s.board = fork_board_;
    for (int i = 0; i < 5; ++i) s.last_moves[i] = cur_last_[i];
    s.color_to_move = n_extra_ == 2 ? color_ : Opp(color_);
    s.move_num = move_num_ + n_extra_;
    s.first_move_behavior = fmb_;
    s.kind = InitState::Kind::kGoExploit;
    s.fork_kind = (int)kind_;
    buffer_->Add(s);
  }

  ForkParams p_;
  ReuseBuffer* buffer_;
  ForkKind kind_ = ForkKind::kUniform;
  bool did_fork_ = false, forced_ = false;
  int fork_mv_ = -1;
  std::vector<InitState> sampled_;
  std::vector<RegretEntry> regret_;
  // running fork task
  Stage stage_ = Stage::kIdle;
  Mode mode_ = Mode::kUniform;
  Probability* prob_ = nullptr;
  Color color_ = kBlack;
  Board origin_, fork_board_, sample_board_, eval_board_;
  Move hist_[5], cur_last_[5], sample_last_[5], eval_last_[5];
  Color sample_color_ = kBlack, eval_color_ = kBlack;
  std::vector<int> cands_;
  int num_candidates_ = 0, take_ = 0, ci_ = 0, move_num_ = 0, n_extra_ = 0;
  float best_util_ = 0;
  Loc best_ = kNoopLoc, alt1_ = kNoopLoc, alt2_ = kNoopLoc;
  bool sample_done_ = true, want_eval_ = false, adj_same_side_ = false, adj_always_ = false;
  FirstMoveBehavior fmb_ = FirstMoveBehavior::kPlay;
};

}  // namespace p3
