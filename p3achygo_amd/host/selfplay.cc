// selfplay.cc — the self-play driver: thousands of concurrent games, their search leaves
// coalesced into large inference batches for the engine behind include/p3hip.h.
//
// Game loop restated from selfplay::Run (cc/selfplay/self_play_thread.cc:309-920), core part:
// raw-policy opening moves, playout-cap randomisation (full search with probability 0.25,
// otherwise a fast search with per-game k and noise scaling), temperature schedule, Gumbel
// root search, tree reuse + Reap, pass-alive refresh at moves 200/250/.../400, max_moves,
// final scoring, init-state sampling (handicap games, GoExploit restarts), the fork manager,
// down-bad visit annealing, sel_mult, the per-game bias cache (--bias_cache_lambda/alpha, on in
// config/v4.json) and the recorders.  Not restated: the opening book
// (probability 0 in the reference).
//
// Scheduling is new (the reference runs one OS thread per game and a 400 us batching
// timeout, nn_interface.cc:279-404): games are resumable state machines (search.h) split in
// N groups, each group bound to its own engine instance; while the GPU evaluates the leaves
// of one group, a pool of host threads consumes the results of the others, advances those
// games to their next leaf and loads the next batch — so every batch holds one leaf of
// (nearly) every game of its group and the GPU never waits for a timeout.  The reference's
// blocking thread-per-game NNInterface is kept too (nn_interface.h) for callers written
// against it.
#include <dlfcn.h>

#include <algorithm>
#include <functional>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

#include "evaluator.h"
#include "features.h"
#include "recorder.h"
#include "search.h"
#include "selfplay_policy.h"
#include "tf_recorder.h"

namespace p3 {

// ---- one game ----------------------------------------------------------------------------
// Buffers finished games for both recorders (recorder::GameRecorder, game_recorder.cc:68-175):
// SGF lines only for games that started on an empty board, training examples for all.
class GameRecorder {
 public:
  GameRecorder(const std::string& path, int gen, const std::string& worker_id, int flush_interval)
      : sgf_(path + "/sgf", gen, worker_id), tf_(path + "/chunks", gen, worker_id), flush_interval_(flush_interval) {
    io_ = std::thread([this] { IoLoop(); });   // game_recorder.cc:83: flushes run on an IO thread
  }
  ~GameRecorder() {
    {
      std::lock_guard<std::mutex> l(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    io_.join();
  }
  void RecordGame(const Board& init_board, const Game& game, std::vector<MoveSearchRecord> infos) {
    if (init_board.IsEmpty()) sgf_.RecordGame(SgfGameString(game, "p3achygo", "p3achygo"));   // game_recorder.cc:20,101-108
    std::lock_guard<std::mutex> l(mu_);
    tf_.RecordGame(init_board, game, std::move(infos));
    if (++buffered_ >= flush_interval_ && !flushing_) {   // should_flush_, game_recorder.cc:113-115
      want_flush_ = true;
      cv_.notify_all();
    }
  }
  // Final flush (the reference flushes on exit): waits for the IO thread, then writes the rest.
  void Flush() {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [this] { return !flushing_ && !want_flush_; });
    auto recs = tf_.TakeRecords();
    buffered_ = 0;
    l.unlock();
    sgf_.Flush();
    const int n = tf_.FlushRecords(std::move(recs));
    l.lock();
    examples_ += n;
  }
  long examples() {
    std::lock_guard<std::mutex> l(mu_);
    return examples_;
  }

 private:
  // Unlike the reference (which holds every game thread's mutex while it replays the games,
  // game_recorder.cc:158-175) the buffered games are detached under the lock and written
  // outside it: no game waits for a flush.
  void IoLoop() {
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      cv_.wait(l, [this] { return want_flush_ || quit_; });
      if (quit_) return;
      want_flush_ = false;
      flushing_ = true;
      auto recs = tf_.TakeRecords();
      buffered_ = 0;
      l.unlock();
      sgf_.Flush();
      const int n = tf_.FlushRecords(std::move(recs));
      l.lock();
      examples_ += n;
      flushing_ = false;
      cv_.notify_all();
    }
  }

  SgfRecorder sgf_;
  TfRecorder tf_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::thread io_;
  int flush_interval_, buffered_ = 0;
  bool want_flush_ = false, flushing_ = false, quit_ = false;
  long examples_ = 0;
};

struct SelfPlayConfig {       // SPConfig, cc/selfplay/self_play_thread.h:38-61
  int selected_n = 128, selected_k = 8;   // --gumbel_selected_{n,k}  selfplay/main.cc:40-43
  int default_n = 32, default_k = 5;      // --gumbel_default_{n,k}   selfplay/main.cc:44-47
  int max_moves = 600;                    // --max_moves
  int nonroot_var_scale_prior_visits = 10;
  float bias_cache_lambda = 0.0f, bias_cache_alpha = 0.8f;   // --bias_cache_{lambda,alpha} main.cc:58-61 (0 = off)
  bool early_stopping_enabled = false;                       // --early_stopping_enabled main.cc:68
  float use_seen_state_prob = 0.5f;       // --use_seen_state_prob     main.cc:48-50
  float sel_mult_base = 0.0f, sel_mult_scale_factor = 1.0f;   // main.cc:51-57
  ForkParams fork_params = ForkParams::ForReuse(0.5f);        // main.cc:191-193
  SelMultCalibration calibration;
  // false: every game starts from the empty board at `komi`, no forks (plumbing tests)
  bool init_state_sampling = true;
  float komi = 7.5f;
  bool raw_policy_opening = true;
  int cache_entries_per_game = 64;   // 0 disables the evaluation cache
  bool enable_puct_fast_search = true;
  ReuseBuffer* reuse = nullptr;      // shared by the games of one process (main.cc:186)
  GameRecorder* recorder = nullptr;  // optional
};

constexpr int kMaxNumRawPolicyMoves = 30;              // self_play_thread.cc:45
constexpr float kMoveSelectedForTrainingProb = 0.25f;  // :62
constexpr int kComputePAMoveNums[] = {200, 250, 300, 350, 400};   // :56
constexpr float kOverSearchNodeProb = 0.15f;           // :65
constexpr float kDownBadThreshold = -0.90f;            // :68
constexpr int kNumDownBadMovesThreshold = 5;           // :71
constexpr float kPuctFastSearchProb = 0.25f;           // :74

struct GameStats {
  long moves = 0, games = 0, evals = 0, black_wins = 0, cache_hits = 0;
  long bias_entries_pruned = 0;   // BiasCache::PruneUnused, self_play_thread.cc:730-733
  double bias_adj_abs_sum = 0;    // sum over moves of |obs_bias| of the root (self_play_thread.cc:639-641)
};

// Per-game evaluation cache: the reference keeps one LRU per game thread keyed by
// NNKey = (colour, board hash, last `num_cache_last_moves` moves, komi)
// (cc/nn/nn_interface.h:206-228, nn_interface.cc:92-118; self-play sets 1 last move,
// selfplay/main.cc:177).  Results are stored un-symmetrised, as the reference does.
class EvalCache {
 public:
  // num_last_moves: how many of the last moves are part of the key (NNInterface::
  // SetNumCacheLastMoves; self-play 1, the default used by cc/eval 5)
  explicit EvalCache(int capacity, int num_last_moves = 1) : cap_(capacity), nlast_(num_last_moves) {}
  struct Key {
    uint64_t hash;
    Loc last[5];
    float komi;
    Color color;
    bool operator==(const Key& o) const {
      if (hash != o.hash || komi != o.komi || color != o.color) return false;
      for (int i = 0; i < 5; ++i)
        if (last[i] != o.last[i]) return false;
      return true;
    }
  };
  // 128-bit digest of a key for the engine's HBM table (two independent mixes of the same fields)
  static void Digest(const Key& k, uint64_t* lo, uint64_t* hi) {
    uint32_t kb;
    std::memcpy(&kb, &k.komi, 4);
    uint64_t a = k.hash ^ (uint64_t(uint8_t(k.color)) * 0x9e3779b97f4a7c15ull);
    uint64_t b = (k.hash * 0xd6e8feb86659fd93ull) ^ (uint64_t(kb) << 8) ^ uint64_t(uint8_t(k.color));
    for (const Loc& l : k.last) {
      const uint64_t m = uint64_t(uint32_t(l.i * 32 + l.j + 64));
      a = (a ^ m) * 0xff51afd7ed558ccdull;
      b = ((b << 7) | (b >> 57)) ^ (m * 0xc2b2ae3d27d4eb4full);
    }
    a = (a ^ kb) * 0xc4ceb9fe1a85ec53ull;
    b = (b ^ (b >> 31)) * 0x94d049bb133111ebull;
    *lo = a ^ (a >> 29);
    *hi = b ^ (b >> 32);
    if ((*lo | *hi) == 0) *lo = 1;
  }
  Key MakeKey(const Position& pos, Color c) const {   // NNInterface::MakeKey, nn_interface.cc:92-106
    Key k{pos.board.hash(), {kNoopLoc, kNoopLoc, kNoopLoc, kNoopLoc, kNoopLoc}, pos.komi(), c};
    for (int i = 5 - nlast_; i < 5; ++i) k.last[i] = pos.last[i].loc;
    return k;
  }
  const p3hip_result* Find(const Key& k) {
    for (auto& e : entries_)
      if (e.key == k) { e.stamp = ++clock_; return &e.result; }
    return nullptr;
  }
  void Insert(const Key& k, const p3hip_result& r) {
    if (Entry* e = Victim()) { e->key = k; e->stamp = ++clock_; e->id = 0; e->pending = false; e->result = r; }
  }
  // Several evaluations of one game in flight (round 4): a request that misses takes its entry AT ONCE
  // (InsertPending) and the result is written into it when it arrives (Fill), so every later lookup sees the cache
  // exactly as the one-at-a-time order Find -> evaluate -> Insert -> Find ... leaves it — the same stamps, the
  // same entry evicted.  A lookup that finds a reserved entry (kPending) waits for that evaluation; an entry
  // evicted before its result arrived is simply not filled (the one-at-a-time order would have filled it and
  // then evicted it).
  enum class Lookup { kMiss, kHit, kPending };
  Lookup Probe(const Key& k, const p3hip_result** hit, uint64_t* pending_id) {
    for (auto& e : entries_)
      if (e.key == k) {
        e.stamp = ++clock_;
        if (e.pending) { *pending_id = e.id; return Lookup::kPending; }
        *hit = &e.result;
        return Lookup::kHit;
      }
    return Lookup::kMiss;
  }
  uint64_t InsertPending(const Key& k) {   // 0: no cache
    Entry* e = Victim();
    if (!e) return 0;
    e->key = k; e->stamp = ++clock_; e->id = ++next_id_; e->pending = true;
    return e->id;
  }
  void Fill(uint64_t id, const p3hip_result& r) {
    if (!id) return;
    for (auto& e : entries_)
      if (e.id == id && e.pending) { e.result = r; e.pending = false; return; }
  }
  void Clear() { entries_.clear(); }

 private:
  struct Entry { Key key; uint64_t stamp = 0, id = 0; bool pending = false; p3hip_result result; };
  Entry* Victim() {   // the entry an insertion takes: a fresh one below capacity, else the least recently used
    if (cap_ <= 0) return nullptr;
    if ((int)entries_.size() < cap_) { entries_.emplace_back(); return &entries_.back(); }
    size_t lru = 0;
    for (size_t i = 1; i < entries_.size(); ++i)
      if (entries_[i].stamp < entries_[lru].stamp) lru = i;
    return &entries_[lru];
  }
  int cap_, nlast_;
  uint64_t clock_ = 0, next_id_ = 0;
  std::vector<Entry> entries_;
};

class GameRunner {
 public:
  GameRunner(const SelfPlayConfig& cfg, uint64_t seed)
      : cfg_(cfg), prob_(seed), seed_(seed), cache_(cfg.cache_entries_per_game),
        move_sel_(kNnMctsBonus | kKldPenalty, cfg.calibration) {   // self_play_thread.cc:315-316
    NewGame();
  }

  // Advances this game until it needs a network evaluation; writes the features of the
  // position to evaluate into *f.  One evaluation in flight (the reference's order of work).
  void AdvanceToEval(p3hip_features* f) {
    if (!TryAdvance(f, 1)) {
      std::fprintf(stderr, "GameRunner::AdvanceToEval: blocked with %zu requests in flight\n", fifo_.size());
      std::abort();
    }
  }
  // Advances this game until it has a position for the engine (true: *f holds it, the request joins the
  // in-flight queue) or until nothing more can start before a result in flight arrives (false).  Up to
  // `max_inflight` playouts of one search wait for results at the same time (GumbelSearch::IssueNext); the
  // game's moves, its random draws, its cache and every result are those of max_inflight = 1.
  bool TryAdvance(p3hip_features* f, int max_inflight) {
    for (;;) {
      if (forking_) {   // the fork manager's candidate evaluations ride in the same batches, one at a time
        if (!fifo_.empty()) return false;
        Color c;
        if (!fork_->NextEval(&side_pos_, &c)) {
          forking_ = false;
          PlayMove();
          continue;
        }
        if (RequestEval(side_pos_, c, f, true)) return true;
        continue;
      }
      const GumbelSearch::Issue s = search_.IssueNext(max_inflight);
      if (s == GumbelSearch::Issue::kNeedEval) {
        if (RequestEval(*search_.eval_game(), search_.eval_color(), f, false)) return true;
        continue;
      }
      if (s == GumbelSearch::Issue::kBlocked) return false;
      FinishMove();
    }
  }
  // the engine's result for the OLDEST request that went to the engine (requests complete in order)
  void DeliverResult(p3hip_result& r) {
    if (fifo_.empty() || fifo_.front().alias_of) {   // a result nobody asked for: a scheduling bug, not something to search on
      std::fprintf(stderr, "GameRunner::DeliverResult: no engine request at the head of the queue (%zu in flight)\n", fifo_.size());
      std::abort();
    }
    Request e = std::move(fifo_.front());
    fifo_.pop_front();
    UnapplySymmetry(e.sym, &r);   // nn_interface.h:263-288
    cache_.Fill(e.cache_id, r);   // nn_interface.cc:130 (the entry was reserved when the request missed)
    for (Request& w : fifo_)      // requests for the same key made while this one was in flight
      if (w.alias_of == e.cache_id && w.alias_of != 0 && !w.res) w.res.reset(new p3hip_result(r));
    Complete(e, r);
    while (!fifo_.empty() && fifo_.front().res) {
      Request a = std::move(fifo_.front());
      fifo_.pop_front();
      Complete(a, *a.res);
    }
  }
  // where the scheduler put the request TryAdvance just returned, and the slot of the oldest request if it went
  // to `lane` (-1 otherwise)
  void SetBackSlot(int lane, int slot) { fifo_.back().lane = lane; fifo_.back().slot = slot; unloaded_ = false; }
  // the request TryAdvance just returned could not be loaded (the batch had gone): it waits, in the scheduler's feature
  // buffer of this game, for the next host phase; nothing else may be requested before it is loaded
  void MarkUnloaded() { unloaded_ = true; }
  bool unloaded() const { return unloaded_; }
  int FrontSlot(int lane) const {
    return !fifo_.empty() && !fifo_.front().alias_of && fifo_.front().lane == lane ? fifo_.front().slot : -1;
  }
  size_t inflight() const { return fifo_.size(); }
  const GameStats& stats() const { return stats_; }
  const Game& game() const { return *game_; }
  // false while the game still samples its opening from the raw policy (one evaluation per move,
  // self_play_thread.cc:44,363-366): the scheduler's advance phase plays every game past it
  bool past_opening() const { return game_->num_moves() >= num_moves_raw_policy_; }
  const std::vector<Move>& last_moves() const { return last_moves_; }
  uint64_t first_game_digest() const { return first_game_digest_; }   // 0 until the first game has finished
  const Game::Result& last_result() const { return last_result_; }

 private:
  struct Request {
    Symmetry sym = kIdentity;
    uint64_t cache_id = 0;    // the cache entry reserved for the result (0: none)
    uint64_t alias_of = 0;    // != 0: no engine request of its own, waits for the request that owns this entry
    bool side = false;        // a fork candidate's evaluation (ForkManager), not the search's
    int lane = -1, slot = -1;
    std::unique_ptr<p3hip_result> res;   // an alias's result, once its owner's has arrived
  };
  // true: *f holds a position for the engine; false: served from the cache (now, or when the evaluation of the
  // same key that is already in flight arrives)
  bool RequestEval(const Position& pos, Color c, p3hip_features* f, bool side) {
    const EvalCache::Key key = cache_.MakeKey(pos, c);
    const p3hip_result* hit = nullptr;
    uint64_t owner = 0;
    switch (cache_.Probe(key, &hit, &owner)) {   // nn_interface.cc:112-118
      case EvalCache::Lookup::kHit:
        ++stats_.cache_hits;
        if (side) fork_->Deliver(*hit);
        else search_.ResolveBack(*hit);
        return false;
      case EvalCache::Lookup::kPending: {
        ++stats_.cache_hits;
        Request w;
        w.alias_of = owner;
        w.side = side;
        fifo_.push_back(std::move(w));
        return false;
      }
      case EvalCache::Lookup::kMiss:
        break;
    }
    Request e;
    e.sym = RandomSymmetry(prob_.prng());   // nn_interface.cc:123
    FillFeatures(pos, c, e.sym, f);
    ++stats_.evals;
    e.cache_id = cache_.InsertPending(key);
    e.side = side;
    fifo_.push_back(std::move(e));
    return true;
  }
  void Complete(const Request& e, const p3hip_result& r) {
    if (e.side) fork_->Deliver(r);
    else search_.Deliver(r);
  }

  void NewGame() {   // self_play_thread.cc:319-426
    InitState init;
    if (cfg_.init_state_sampling) {
      init = GetInitState(prob_, cfg_.reuse, cfg_.use_seen_state_prob);
    } else {
      init.board = Board(cfg_.komi, true);
    }
    force_full_search_first_move_ = init.first_move_behavior == FirstMoveBehavior::kForceFullSearch;
    const bool disable_sampling = init.first_move_behavior != FirstMoveBehavior::kSample;
    const bool is_fresh_game = init.kind == InitState::Kind::kEmpty || init.kind == InitState::Kind::kBook ||
                               init.kind == InitState::Kind::kHandicap;
    init_board_ = init.board;
    game_.reset(new Game(init.board, init.last_moves, init.move_num));
    color_ = init.color_to_move;
    pool_.Clear();
    // one bias cache per game (self_play_thread.cc:407-412): the old game's nodes are gone
    bias_cache_.reset(cfg_.bias_cache_lambda > 0.0f ? new BiasCache(cfg_.bias_cache_alpha, cfg_.bias_cache_lambda)
                                                    : nullptr);
    search_.set_bias_cache(bias_cache_.get());
    root_ = pool_.Create();
    move_infos_.clear();
    num_consecutive_down_bad_moves_ = 0;
    (void)prob_.Uniform();   // log_mcts_trees draw, kLogFullTreeProb = 0 (:353)
    const int max_raw = (int)std::round(kMaxNumRawPolicyMoves * std::pow(0.5f, init.move_num / 40.0f));
    num_moves_raw_policy_ = 0;
    if (cfg_.raw_policy_opening && !disable_sampling && max_raw > 0 && prob_.Uniform() < 1.0f)   // kOpeningExploreProb
      num_moves_raw_policy_ = RandRange(prob_.prng(), 0, max_raw);
    use_puct_fast_search_ = prob_.Uniform() < kPuctFastSearchProb && cfg_.enable_puct_fast_search;
    fast_move_noise_scaling_ = prob_.Uniform() / 1.4f;
    {
      int num_rounds = (int)std::log2((double)cfg_.default_k);
      int min_k = 1 << num_rounds;
      fast_move_gumbel_k_ = RandRange(prob_.prng(), min_k, cfg_.default_k + 1);
    }
    fast_move_root_fpu_ = prob_.Uniform() * 0.1f;   // :424
    fork_.reset();
    if (cfg_.init_state_sampling && cfg_.reuse)
      fork_.reset(new ForkManager(cfg_.fork_params, cfg_.reuse, prob_, /*started_from_forced_search=*/!is_fresh_game));
    forking_ = false;
    fifo_.clear();
    unloaded_ = false;
    cache_.Clear();
    if (game_->IsGameOver() || game_->num_moves() >= cfg_.max_moves) {
      // A restart state can already be finished (a fork whose alternative move was the second
      // pass): the reference's game loop (self_play_thread.cc:427-428) is then skipped and the
      // game is scored and recorded as it stands.
      FinishGame();
      return;
    }
    BeginSearch();
  }

  void FinishGame() {   // self_play_thread.cc:900-912
    game_->WriteResult();
    ++stats_.games;
    if (game_->result().winner == kBlack) ++stats_.black_wins;
    last_result_ = game_->result();
    last_moves_ = game_->moves();
    if (stats_.games == 1) {   // a digest of this runner's first game: schedules are compared by it (tests)
      uint64_t d = 0xcbf29ce484222325ull;
      for (const Move& m : last_moves_) d = (d ^ (uint64_t)(MoveIdx(m.loc) * 4 + (int)m.color + 2)) * 0x100000001b3ull;
      uint32_t bs, ws;
      std::memcpy(&bs, &last_result_.bscore, 4);
      std::memcpy(&ws, &last_result_.wscore, 4);
      first_game_digest_ = ((d ^ bs) * 0x100000001b3ull ^ ws) * 0x100000001b3ull | 1;
    }
    if (fork_) fork_->FinalizeGame(*game_, prob_);
    if (cfg_.recorder) cfg_.recorder->RecordGame(init_board_, *game_, std::move(move_infos_));
    NewGame();
  }

  void BeginSearch() {   // self_play_thread.cc:429-611
    const bool sampling_raw_policy = game_->num_moves() < num_moves_raw_policy_;
    const bool is_either_down_bad = num_consecutive_down_bad_moves_ >= kNumDownBadMovesThreshold;
    float down_bad_coeff = 1.0f;
    {
      const float root_v = VOutcome(root_);
      if (!(root_v > kDownBadThreshold && root_v < -kDownBadThreshold))
        down_bad_coeff = (1.0f - std::abs(root_v)) / (1.0f - std::abs(kDownBadThreshold));
    }
    // pre-search statistics (from tree reuse)
    pre_.sampling_raw_policy = sampling_raw_policy;
    const float qz_nn = root_->init_outcome_est;
    pre_.n_pre = root_->n;
    pre_.q_pre = V(root_);
    const float qz_pre = VOutcome(root_);
    pre_.var_pre = pre_.n_pre < 3 ? 0.0f : root_->v_outcome_var;
    pre_.pre_kld = 0.0f;
    if (root_->n >= 1) {
      float pi[kNumMoves];
      ComputeImprovedPolicyN(root_, 0, pi);
      pre_.pre_kld = ComputeKLD(pi, root_->move_probs);
    }
    const float q_canonical = qz_pre == 0.0f ? qz_nn : qz_pre;
    pre_.nn_mcts_diff_pre = pre_.n_pre > 0 ? std::abs(qz_nn - pre_.q_pre) : 0.0f;
    const MoveSelResult sel = move_sel_.Compute(pre_.n_pre, std::sqrt(pre_.var_pre), pre_.pre_kld, pre_.nn_mcts_diff_pre,
                                                q_canonical, cfg_.sel_mult_scale_factor);
    pre_.sel_mult_modifier = sampling_raw_policy ? 1.0f : sel.modifier;
    const float sel_mult = cfg_.sel_mult_base > 0.0f ? cfg_.sel_mult_base * pre_.sel_mult_modifier : 1.0f;
    float select_move_prob = 0.0f;
    pre_.select_move_prob_base = 0.0f;
    if (!sampling_raw_policy) {
      pre_.select_move_prob_base = is_either_down_bad ? down_bad_coeff * down_bad_coeff * kMoveSelectedForTrainingProb
                                                      : kMoveSelectedForTrainingProb;
      select_move_prob = pre_.select_move_prob_base * sel_mult;
    }
    int train_n = cfg_.selected_n, train_k = cfg_.selected_k;   // trainable_gumbel_params, :526-537
    if (is_either_down_bad) {
      train_n = (int)((1.0f - down_bad_coeff) * cfg_.default_n + down_bad_coeff * cfg_.selected_n);
      train_k = cfg_.default_k;
    }
    const bool force_first_move = force_full_search_first_move_ && game_->num_moves() == 0;
    const bool selected = force_first_move || prob_.Uniform() < select_move_prob;   // :541-542
    (void)prob_.Uniform();                                  // over-search draw (`&& false`, :543-545)
    pre_.selected = selected;
    GumbelParams p;
    p.nonroot_var_scale_prior_visits = cfg_.nonroot_var_scale_prior_visits;
    p.early_stopping_enabled = cfg_.early_stopping_enabled;   // (!is_move_over_search: over-search is dead code, :544-548)
    if (force_first_move) {
      p.n = cfg_.selected_n; p.k = cfg_.selected_k;
    } else if (sampling_raw_policy) {
      p.n = 1; p.k = 1;
    } else if (selected) {
      p.n = train_n; p.k = train_k;
    } else {
      p.n = cfg_.default_n; p.k = fast_move_gumbel_k_; p.noise_scaling = fast_move_noise_scaling_;
    }
    {   // tau schedule, self_play_thread.cc:570-582
      const int non_sample = game_->num_moves() - num_moves_raw_policy_;
      const float lambda = std::log(2.0f) / 19;
      p.tau = std::min(std::max(0.8f * std::exp(-lambda * non_sample), 0.2f), 0.8f);
    }
    if (!selected && !sampling_raw_policy && use_puct_fast_search_) {   // self_play_thread.cc:585-611
      PuctParams pp;
      pp.c_puct = 1.05f;
      pp.c_puct_visit_scaling = 0.28f;
      pp.root_fpu = fast_move_root_fpu_;
      search_.BeginPuct(game_.get(), &pool_, root_, color_, p.n, pp, p.tau, &prob_);
      return;
    }
    search_.Begin(game_.get(), &pool_, root_, color_, p, &prob_);
  }

  void FinishMove() {   // self_play_thread.cc:614-690
    const GumbelResult& res = search_.result();
    move_ = res.mcts_move;
    const float root_q_outcome = VOutcome(root_);
    // bias-cache adjustment of this root, read-only (self_play_thread.cc:638-641; the reference logs it)
    if (bias_cache_) stats_.bias_adj_abs_sum += std::abs(bias_cache_->Fetch(root_->bias));
    if (cfg_.recorder) {
      MoveSearchRecord mi;
      std::memcpy(mi.mcts_pi, res.pi_improved, sizeof mi.mcts_pi);
      mi.move_trainable = pre_.selected;
      mi.root_q_outcome = root_q_outcome;
      mi.root_score = root_->score;
      mi.kld = res.kld;
      std::memcpy(mi.mcts_value_dist, root_->v_categorical, sizeof mi.mcts_value_dist);
      MoveSearchStats& st = mi.move_stats;   // :654-669 (visit_count is never set there: stays 0)
      st.sampled_raw_policy = pre_.sampling_raw_policy;
      st.nn_q = root_->init_util_est;
      st.mcts_q = pre_.q_pre;
      st.nn_mcts_diff = pre_.nn_mcts_diff_pre;
      st.v_outcome_stddev = std::sqrt(pre_.var_pre);
      float ent = 0;
      for (int a = 0; a < kNumMoves; ++a)
        if (root_->move_probs[a] > 0.0f) ent -= root_->move_probs[a] * std::log(root_->move_probs[a]);
      st.prior_entropy = ent;
      st.nn_uncertainty = root_->v_err;
      st.kld = 0.0f;
      if (root_->n >= 1) {
        float pi[kNumMoves];
        ComputeImprovedPolicyN(root_, 0, pi);
        st.kld = ComputeKLD(pi, root_->move_probs);   // post_kld
      }
      st.pre_kld = pre_.pre_kld;
      st.sel_mult_modifier = pre_.sel_mult_modifier;
      st.sel_mult_modifier_weight = pre_.select_move_prob_base / kMoveSelectedForTrainingProb;
      st.visit_count_pre = (float)pre_.n_pre;
      move_infos_.push_back(mi);
    }
    if (-std::abs(root_q_outcome) < kDownBadThreshold) ++num_consecutive_down_bad_moves_;
    else num_consecutive_down_bad_moves_ = 0;
    if (fork_) {   // fork before playing the move (:681-688)
      ForkManager::MoveData md{&game_->board(), color_, move_, root_->init_util_est, V(root_), root_->score,
                               root_->child_visits(MoveIdx(move_)) != 0};
      if (fork_->MaybeFork(*game_, md, prob_)) {
        forking_ = true;
        return;
      }
    }
    PlayMove();
  }

  void PlayMove() {   // self_play_thread.cc:690-722, 900-912
    const Loc move = move_;
    game_->PlayMove(move, color_);
    ++stats_.moves;
    for (int m : kComputePAMoveNums)
      if (game_->num_moves() == m) game_->mutable_board().CalculatePassAliveRegions();
    color_ = Opp(color_);
    TreeNode* next = root_->child(MoveIdx(move));
    if (!next) next = pool_.Create();
    pool_.Reap(next);   // self_play_thread.cc:711-722
    root_ = next;
    if (bias_cache_) stats_.bias_entries_pruned += bias_cache_->PruneUnused();   // :730-733
    if (game_->IsGameOver() || game_->num_moves() >= cfg_.max_moves) {
      FinishGame();
      return;
    }
    if (root_->is_terminal) {   // cannot happen while the game is not over; defensive reset
      root_ = pool_.Create();
      pool_.Reap(root_);
    }
    BeginSearch();
  }

  struct PreSearch {
    bool sampling_raw_policy = false, selected = false;
    int n_pre = 0;
    float q_pre = 0, var_pre = 0, pre_kld = 0, nn_mcts_diff_pre = 0, sel_mult_modifier = 1, select_move_prob_base = 0;
  };

  SelfPlayConfig cfg_;
  Probability prob_;
  uint64_t seed_;
  std::unique_ptr<Game> game_;
  Board init_board_;
  std::unique_ptr<BiasCache> bias_cache_;   // declared before the pool: nodes release their entries first
  NodePool pool_;
  TreeNode* root_ = nullptr;
  Color color_ = kBlack;
  Loc move_ = kNoopLoc;
  GumbelSearch search_;
  int num_moves_raw_policy_ = 0, fast_move_gumbel_k_ = 4, num_consecutive_down_bad_moves_ = 0;
  float fast_move_noise_scaling_ = 1.0f, fast_move_root_fpu_ = 0.0f;
  bool use_puct_fast_search_ = false, force_full_search_first_move_ = false;
  bool forking_ = false;
  std::deque<Request> fifo_;   // evaluation requests in flight, oldest first
  bool unloaded_ = false;
  std::unique_ptr<ForkManager> fork_;
  Position side_pos_;
  PreSearch pre_;
  std::vector<MoveSearchRecord> move_infos_;
  EvalCache cache_;
  MoveSelManager move_sel_;
  GameStats stats_;
  Game::Result last_result_;
  std::vector<Move> last_moves_;
  uint64_t first_game_digest_ = 0;
};

// ---- scheduler -----------------------------------------------------------------------------
// A pool of host threads that runs several ParallelFor jobs at once: every game group submits its
// own job from its own driver thread, the workers always take the oldest job that still has
// items, so a game that is slow to advance (a long ladder read-out) holds up its own group only.
class WorkerPool {
 public:
  explicit WorkerPool(int n) {
    for (int i = 0; i < n; ++i) threads_.emplace_back([this] { Loop(); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  struct Job {
    std::function<void(int)> fn;
    int total = 0, next = 0, unfinished = 0;   // guarded by the pool's mutex
    std::condition_variable done_cv;
  };
  using JobRef = std::shared_ptr<Job>;
  // runs fn(i) for i in [0, n) on the pool and returns when all are done; callable from several
  // threads at the same time
  template <class F>
  void ParallelFor(int n, F&& fn) {
    if (n <= 0) return;
    Wait(Submit(n, std::forward<F>(fn)));
  }
  // the same in two steps: the caller may stop waiting (WaitFor) and come back for the rest later (Wait); whatever
  // fn refers to must then live until that Wait returns
  template <class F>
  JobRef Submit(int n, F&& fn) {
    JobRef job = std::make_shared<Job>();
    job->fn = std::forward<F>(fn);
    job->total = n;
    job->unfinished = n;
    if (n > 0) {
      {
        std::lock_guard<std::mutex> l(mu_);
        jobs_.push_back(job);
      }
      cv_.notify_all();
    }
    return job;
  }
  void Wait(const JobRef& job) {
    std::unique_lock<std::mutex> l(mu_);
    job->done_cv.wait(l, [&] { return job->unfinished == 0; });
  }
  // waits at most `us` microseconds; the items not finished yet (0: done) and whether every item has been handed out.
  // (Polls under the mutex between short sleeps: no timed condition-variable wait, whose pthread_cond_clockwait the
  // ThreadSanitizer of this toolchain does not model.)
  int WaitFor(const JobRef& job, long us, bool* all_started) {
    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(us);
    for (;;) {
      {
        std::lock_guard<std::mutex> l(mu_);
        if (job->unfinished == 0 || std::chrono::steady_clock::now() >= until) {
          if (all_started) *all_started = job->next == job->total;
          return job->unfinished;
        }
      }
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
  }

 private:
  void Loop() {
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      JobRef job;
      cv_.wait(l, [&] {
        if (stop_) return true;
        for (const JobRef& j : jobs_)
          if (j->next < j->total) { job = j; return true; }
        return false;
      });
      if (stop_) return;
      const int i = job->next++;
      if (job->next == job->total) jobs_.erase(std::find(jobs_.begin(), jobs_.end(), job));   // no items left to hand out
      l.unlock();
      job->fn(i);
      l.lock();
      if (--job->unfinished == 0) job->done_cv.notify_all();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<JobRef> jobs_;   // oldest first
  bool stop_ = false;
};

// One game group: its games and its LANES.  A lane is an engine instance with its own stream and a driver
// thread that alternates "evaluate the lane's batch" (Evaluator::Run: H2D, forward pass, D2H) and the lane's host
// phase "hand its results to the games, advance every game to its next leaf, load the leaves" (ParallelFors on the
// shared pool).  With one lane (the default) host and GPU alternate within the group and it is the other groups
// that keep the GPU busy.  With two, the host phases of the lanes alternate — while one lane's batch is on the GPU
// the other lane's results are consumed and its next batch is filled — which needs games that can have leaves in
// both batches at once: GameRunner::TryAdvance with several playouts in flight.
struct Lane {
  std::unique_ptr<Evaluator> eval;
  std::thread driver;
  std::atomic<int> count{0};   // slots loaded for the next run
  // written in the lane's host phase, read by its accounting after the run that evaluated that batch
  GameStats delta;             // game counters added by the host phase that loaded the batch
  double host_seconds = 0;
  bool in_opening = false;     // some game still samples its opening from the raw policy
  long past_opening = 0;
};
struct Half {
  std::vector<std::unique_ptr<Lane>> lanes;
  std::vector<std::unique_ptr<GameRunner>> games;
  std::vector<p3hip_features> feats;
  // whose host phase is next (the lanes' host phases alternate strictly, so a game's results arrive in request order)
  std::mutex turn_mu;
  std::condition_variable turn_cv;
  int turn = 0;
  bool quit = false;
  GameStats prev;              // game counters at the end of the last host phase (host phases only)
  // Stragglers (two or more lanes): a host phase that is waiting for a last few slow games while the GPU has run dry hands
  // them over — its batch leaves without them, they finish on the pool (`late`), and the group's next host phase starts by
  // waiting for them and loads what they asked for into ITS batch.  `settled[g]` = game g's task of the current phase has
  // ended (its counters may be read); `snap` = every game's counters as last read settled.
  WorkerPool::JobRef late;
  std::unique_ptr<std::atomic<uint8_t>[]> settled;
  std::vector<GameStats> snap;
  std::vector<uint8_t> snap_past_opening;
  std::atomic<int> waiting{0};   // lane drivers waiting for their turn (their runs are done: the GPU is about to idle)
  long handed_over_phases = 0, handed_over_games = 0, phase_no = 0;
  // guarded by the job's clock mutex; read by the caller after the threads have been joined
  bool ok = true, ready = false;
  long warm = 0;
  int adv_left = 0;
  double gpu_seconds = 0, host_seconds = 0;   // inside Run / inside the host phases, measured region only
  long measured_batches = 0;
  GameStats counted;                           // game counters summed over the measured batches
  long advance_batches = 0, past_opening_at_start = 0;
};

}  // namespace p3

using namespace p3;

namespace {
std::string g_rec_dir, g_rec_worker = "0";
int g_rec_gen = 0, g_rec_flush_interval = 128;   // --flush_interval, selfplay/main.cc:35
bool g_init_state_sampling = true;
float g_use_seen_state_prob = 0.5f, g_sel_mult_base = 0.0f, g_sel_mult_scale = 1.0f;
float g_bias_cache_lambda = 0.0f, g_bias_cache_alpha = 0.8f;
bool g_early_stopping = false;
std::string g_calibration_file;
long g_last_bias_pruned = 0;
double g_last_bias_adj = 0;
int g_num_groups = 2;
long g_step_limit = 0;   // > 0: the measured region ends after this many engine batches
int g_num_lanes = 1, g_max_inflight = 1;
long g_test_slow_us = 0;   // tests: some games sleep this long in some host phases (a stand-in for a slow ladder read-out)
long g_last_handed_over_phases = 0, g_last_handed_over_games = 0;
std::vector<uint64_t> g_first_game_digests;   // of the last p3host_selfplay_run, one per game runner
long g_step_rounds = 0;  // > 0: the measured region is this many ROUNDS (one batch of every group), anchored on one group
int g_advance_limit = 0;   // > 0: untimed batches per group, at most, to play every game past its raw-policy opening
long g_last_reuse_added = 0, g_last_examples = 0;
}

extern "C" {

// Game-loop policy of subsequent p3host_selfplay_run calls: init_state_sampling = 0 plays
// every game from the empty board at komi 7.5 without forks (plumbing tests); otherwise the
// reference defaults (selfplay/main.cc:48-57) unless overridden here.
void p3host_selfplay_set_policy(int init_state_sampling, float use_seen_state_prob, float sel_mult_base,
                                float sel_mult_scale_factor) {
  g_init_state_sampling = init_state_sampling != 0;
  g_use_seen_state_prob = use_seen_state_prob;
  g_sel_mult_base = sel_mult_base;
  g_sel_mult_scale = sel_mult_scale_factor;
}
// --bias_cache_lambda / --bias_cache_alpha of subsequent runs (selfplay/main.cc:58-61,257-258;
// lambda 0 = off, the reference default; config/v4.json: 0.3 / 0.8).
void p3host_selfplay_set_bias_cache(float lambda, float alpha) {
  g_bias_cache_lambda = lambda;
  g_bias_cache_alpha = alpha;
}
// --sel_mult_calibration_file of subsequent runs (selfplay/main.cc:64-67,224); empty = built-in thresholds
void p3host_selfplay_set_calibration_file(const char* path) { g_calibration_file = path ? path : ""; }
// --early_stopping_enabled of subsequent runs (selfplay/main.cc:68,260; off by default)
void p3host_selfplay_set_early_stopping(int enabled) { g_early_stopping = enabled != 0; }
// bias-cache entries pruned / sum of |root adjustment| over the moves of the last run or game
long p3host_selfplay_last_bias_pruned() { return g_last_bias_pruned; }
double p3host_selfplay_last_bias_adj() { return g_last_bias_adj; }
// Number of game groups of subsequent p3host_selfplay_run calls (1..8).  Each group has its own
// engine instance and is either being advanced on the host or evaluated on the GPU; with G
// groups up to G - 1 forward passes are in flight while one group is on the host (one group:
// host and GPU alternate, BASELINE configs[2] as written).
void p3host_selfplay_set_groups(int n) { g_num_groups = n < 1 ? 1 : (n > 8 ? 8 : n); }
// Lanes per game group (engine instances whose batches the group's games fill in turn) and how many playouts of one
// search may wait for results at once.  1 / 1 (the default): host and GPU alternate within a group.  2 / up to 4:
// one group overlaps its host work with its own forward passes (BASELINE configs[2] as stated: 1024 games, batch 1024).
void p3host_selfplay_set_lanes(int lanes, int max_inflight) {
  g_num_lanes = lanes < 1 ? 1 : (lanes > 4 ? 4 : lanes);
  g_max_inflight = max_inflight < 1 ? 1 : (max_inflight > GumbelSearch::kMaxInflight ? GumbelSearch::kMaxInflight : max_inflight);
}
// > 0: subsequent p3host_selfplay_run calls measure exactly `batches` engine batches (bench.py's
// --steps), whichever groups they fall in, instead of running for `seconds`; 0 restores the time limit.
void p3host_selfplay_set_step_limit(long batches) { g_step_limit = batches > 0 ? batches : 0; }
// > 0 (takes precedence over the step limit): subsequent runs measure `rounds` ROUNDS.  The window opens at the
// completion of the warm-up batch of the group that finishes its warm-up last (the anchor) and closes at the
// completion of the anchor's `rounds`-th batch after that: both ends sit at the same phase of the groups' cycle on
// the GPU (the groups' forward passes complete in bursts, so a window between two arbitrary completions is off by
// up to a burst), and every batch of any group that completes inside (t0, t1] is counted — in steady state one per
// group and round.  bench.py's --steps; 0 = off.
void p3host_selfplay_set_step_rounds(long rounds) { g_step_rounds = rounds > 0 ? rounds : 0; }
// > 0: before the warm-up batches every group runs untimed batches until all its games have left
// their raw-policy opening (up to 30 moves of one evaluation each, self_play_thread.cc:44,363-366),
// at most `max_batches` of them, so that a short measured region is steady-state search; 0 = off.
void p3host_selfplay_set_advance_limit(int max_batches) { g_advance_limit = max_batches > 0 ? max_batches : 0; }
// reuse-buffer insertions and training examples written by the last p3host_selfplay_run
long p3host_selfplay_last_reuse_added() { return g_last_reuse_added; }
// tests: every 61st (game, host phase) pair sleeps `us` microseconds before it advances; 0 = off
void p3host_selfplay_set_test_slow_games(long us) { g_test_slow_us = us > 0 ? us : 0; }
// of the last p3host_selfplay_run: host phases that let their batch leave without a last few slow games, and those games
void p3host_selfplay_last_handed_over(long* phases, long* games) {
  if (phases) *phases = g_last_handed_over_phases;
  if (games) *games = g_last_handed_over_games;
}
// per game runner of the last p3host_selfplay_run (group by group): a digest of its first finished game, 0 if none
int p3host_selfplay_last_first_game_digests(uint64_t* out, int cap) {
  int n = 0;
  for (uint64_t d : g_first_game_digests)
    if (n < cap) out[n++] = d;
  return (int)g_first_game_digests.size();
}
long p3host_selfplay_last_examples() { return g_last_examples; }

// Enables game recording for subsequent p3host_selfplay_run calls (dir == "" disables):
// <dir>/sgf and <dir>/chunks as in selfplay/main.cc:157-158.
void p3host_selfplay_set_recorder(const char* dir, int gen, const char* worker_id, int flush_interval) {
  g_rec_dir = dir ? dir : "";
  g_rec_gen = gen;
  g_rec_worker = worker_id ? worker_id : "0";
  g_rec_flush_interval = flush_interval > 0 ? flush_interval : 128;
}

// Serializes a move list (encoded as in p3host_selfplay_one_game) the way the recorder does.
int p3host_sgf_from_moves(const int* moves, int n, float komi, int write_result, const char* b_name,
                          const char* w_name, char* out, int cap) {
  Game g(komi, true);
  for (int i = 0; i < n; ++i) {
    Color c = moves[i] > 0 ? kBlack : kWhite;
    int idx = (moves[i] > 0 ? moves[i] : -moves[i]) - 1;
    g.PlayMove(MoveLoc(idx), c);
  }
  if (write_result) g.WriteResult();
  std::string s = SgfGameString(g, b_name, w_name);
  if ((int)s.size() + 1 > cap) return -1;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

struct p3host_selfplay_stats {
  double seconds;            // wall time of the measured region
  long positions;            // network evaluations (engine slots loaded)
  long moves, games, black_wins;
  long batches;              // engine runs
  double gpu_seconds;        // time spent inside Evaluator::Run, summed over both halves
  double host_seconds;       // time spent advancing games (both halves, wall)
  long cache_hits;           // evaluations served by the per-game cache (not in `positions`)
  long advance_batches;      // untimed batches of the advance phase, all groups (p3host_selfplay_set_advance_limit)
  long games_past_opening;   // games past their raw-policy opening when the measured region began
  // batches x the least-squares slope of (batch index, completion instant) over the window: the same K steps, but
  // every completion weighs in instead of the first and the last (the groups' kernels interleave on the GPU, so
  // completions come in bursts and a short window's end points are +-1 batch); `seconds` is the window itself
  double seconds_fit;
  long rounds;               // p3host_selfplay_set_step_rounds: the anchor group's batches inside the window (0 otherwise)
};

// Runs self-play for about `seconds` (after `warmup_batches` unmeasured batches per half).
// engine_lib: path of libp3hip.so, or NULL/"" for the NullEvaluator (uniform policy).
// Returns 0 on success; `err` (256 bytes) receives a message otherwise.
int p3host_selfplay_run(const char* engine_lib, const char* weights, int device, int num_games,
                        int num_threads, int default_n, int default_k, int selected_n,
                        int selected_k, int max_moves, double seconds, int warmup_batches,
                        uint64_t seed, p3host_selfplay_stats* out, char* err) {
  SelfPlayConfig cfg;
  cfg.default_n = default_n; cfg.default_k = default_k;
  cfg.selected_n = selected_n; cfg.selected_k = selected_k;
  cfg.max_moves = max_moves;
  cfg.init_state_sampling = g_init_state_sampling;
  cfg.use_seen_state_prob = g_use_seen_state_prob;
  cfg.sel_mult_base = g_sel_mult_base;
  cfg.sel_mult_scale_factor = g_sel_mult_scale;
  cfg.bias_cache_lambda = g_bias_cache_lambda;
  cfg.bias_cache_alpha = g_bias_cache_alpha;
  cfg.early_stopping_enabled = g_early_stopping;
  cfg.calibration = ParseCalibrationFile(g_calibration_file);
  cfg.fork_params = ForkParams::ForReuse(g_use_seen_state_prob);
  auto reuse = std::make_unique<ReuseBuffer>(seed ^ 0x676f6578706c6f69ull);
  cfg.reuse = reuse.get();
  std::unique_ptr<GameRecorder> recorder;
  if (!g_rec_dir.empty()) {
    ::mkdir((g_rec_dir + "/sgf").c_str(), 0755);
    ::mkdir((g_rec_dir + "/chunks").c_str(), 0755);
    recorder.reset(new GameRecorder(g_rec_dir, g_rec_gen, g_rec_worker, g_rec_flush_interval));
    cfg.recorder = recorder.get();
  }
  const int NG = g_num_groups;
  const int NL = g_num_lanes;
  const int depth = NL > 1 ? g_max_inflight : 1;   // playouts of one search in flight
  if (num_games < NG) num_games = NG;
  std::vector<Half> halves(NG);
  const bool use_null = !engine_lib || !engine_lib[0];
  for (int h = 0; h < NG; ++h) {
    const int ng = num_games / NG + (h < num_games % NG ? 1 : 0);
    for (int l = 0; l < NL; ++l) {
      halves[h].lanes.emplace_back(new Lane());
      if (use_null) {
        halves[h].lanes[l]->eval.reset(new NullEvaluator());
      } else if (!std::strcmp(engine_lib, "hash")) {   // tests: position-dependent results without a network
        halves[h].lanes[l]->eval.reset(new HashEvaluator(ng));
      } else {
        auto* e = new HipEvaluator();
        halves[h].lanes[l]->eval.reset(e);
        if (!e->Open(engine_lib, weights, ng, device, P3HIP_FLAG_SHARED_DEVICE)) {   // one engine per lane, all on this GPU
          if (err) snprintf(err, 256, "%s", e->err.c_str());
          return 1;
        }
      }
    }
    for (int g = 0; g < ng; ++g) {
      // documented per-game seed (replaces absl::HashOf(worker_id, thread_id), main.cc:244)
      uint64_t s = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(h * 1000003 + g) * 0xBF58476D1CE4E5B9ull;
      halves[h].games.emplace_back(new GameRunner(cfg, s));
    }
  }
  WorkerPool pool(num_threads > 0 ? num_threads : 1);
  for (int h = 0; h < NG; ++h) {
    Half& H = halves[h];
    const size_t n = H.games.size();
    H.feats.resize(n);
    H.settled.reset(new std::atomic<uint8_t>[n]);
    H.snap.resize(n);
    H.snap_past_opening.assign(n, 0);
    for (size_t g = 0; g < n; ++g) {
      H.settled[g].store(1);
      H.snap[g] = H.games[g]->stats();
    }
  }

  // The host phase of lane `l` of group `h` (the caller holds the group's turn): every game takes its results of
  // the lane's last run, in request order, and advances to its next leaf, which goes into the lane's next batch.
  // With several playouts in flight a game may have nothing to start (its next playout waits for a result of the
  // OTHER lane's batch); the rows such games leave empty go, in a second pass, to games that can start another
  // playout, so the batch stays full.
  struct PhaseCtl { std::atomic<bool> open{true}; std::atomic<int> loading{0}, delivered{0}; };
  const long slow_us = g_test_slow_us;
  auto host_phase = [&](int h, int l) {
    Half& H = halves[h];
    Lane& L = *H.lanes[l];
    const int cap = (int)H.games.size();
    // the previous phase's stragglers first: from here on nothing but this phase touches the group's games
    if (H.late) { pool.Wait(H.late); H.late.reset(); }
    L.count.store(0);
    auto ctl = std::make_shared<PhaseCtl>();
    for (int g = 0; g < cap; ++g) H.settled[g].store(0, std::memory_order_relaxed);
    const long phase_no = H.phase_no++;
    WorkerPool::JobRef job = pool.Submit(cap, [&H, &L, l, depth, ctl, slow_us, phase_no](int g) {
      GameRunner& G = *H.games[g];
      // the request in feats[g] goes into this lane's batch — while the phase is open; after that it waits for the next one
      auto load = [&]() -> bool {
        ctl->loading.fetch_add(1);
        const bool open = ctl->open.load();
        if (open) {
          const int slot = L.count.fetch_add(1);
          L.eval->Load(slot, H.feats[g]);
          G.SetBackSlot(l, slot);
        } else {
          G.MarkUnloaded();
        }
        ctl->loading.fetch_sub(1);
        return open;
      };
      for (int slot; (slot = G.FrontSlot(l)) >= 0;) {
        p3hip_result r;
        L.eval->Get(slot, r);
        G.DeliverResult(r);
      }
      ctl->delivered.fetch_add(1);     // this game no longer needs the lane's last results: the lane may run again
      // a request made by a straggler of the previous phase (after the deliveries: it must not pass for one of this
      // lane's last batch)
      const bool ok = !G.unloaded() || load();
      if (slow_us > 0 && ((uint64_t)g * 2654435761ull + (uint64_t)phase_no * 40503ull) % 61 == 0)   // tests: a slow game
        std::this_thread::sleep_for(std::chrono::microseconds(slow_us));
      if (ok && G.TryAdvance(&H.feats[g], depth)) load();
      H.settled[g].store(1, std::memory_order_release);
    });
    if (H.lanes.size() > 1) {
      // Wait for the games — but not for a last few slow ones (an exact ladder read-out can take tens of milliseconds)
      // once another lane's run has come back, i.e. the GPU has nothing left to do: the batch then leaves without them.
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        bool all_started = false;
        const int left = pool.WaitFor(job, 100, &all_started);
        if (left == 0) break;
        if (left * 32 <= cap && all_started && ctl->delivered.load() == cap && H.waiting.load() > 0 &&
            std::chrono::steady_clock::now() - t0 >= std::chrono::milliseconds(1)) {
          ctl->open.store(false);
          while (ctl->loading.load() > 0) std::this_thread::yield();   // a load that saw the phase open completes
          H.late = job;
          ++H.handed_over_phases;
          H.handed_over_games += left;
          break;
        }
      }
    } else {
      pool.Wait(job);
    }
    if (depth > 1 && L.count.load() < cap) {
      // rows left empty (games that wait for a result of the other lane's batch, stragglers handed over) go to games that
      // can start another playout
      std::atomic<int> spare{cap - L.count.load()};
      pool.ParallelFor(cap, [&H, &L, &spare, l, depth](int g) {
        if (!H.settled[g].load(std::memory_order_acquire)) return;   // a straggler: still at work on the pool
        GameRunner& G = *H.games[g];
        if (G.unloaded()) return;   // a straggler that has just ended with a request: feats[g] holds it for the next phase
        while (spare.load(std::memory_order_relaxed) > 0) {
          if (spare.fetch_sub(1) <= 0) { spare.fetch_add(1); break; }
          if (!G.TryAdvance(&H.feats[g], depth)) { spare.fetch_add(1); break; }
          const int slot = L.count.fetch_add(1);
          L.eval->Load(slot, H.feats[g]);
          G.SetBackSlot(l, slot);
        }
      });
    }
    // counters of the games whose tasks have ended (a straggler's are read when it has)
    for (int g = 0; g < cap; ++g)
      if (H.settled[g].load(std::memory_order_acquire)) {
        H.snap[g] = H.games[g]->stats();
        H.snap_past_opening[g] = H.games[g]->past_opening();
      }
  };
  auto group_totals = [](const Half& H) {
    GameStats t;
    for (const GameStats& g : H.snap) {
      t.moves += g.moves; t.games += g.games;
      t.evals += g.evals; t.black_wins += g.black_wins;
      t.cache_hits += g.cache_hits;
      t.bias_entries_pruned += g.bias_entries_pruned;
      t.bias_adj_abs_sum += g.bias_adj_abs_sum;
    }
    return t;
  };

  // Every group runs on its own: advance batches (until its games are past their openings), warm-up
  // batches, then measured batches.  A batch = advance all games of the group to their next leaf
  // (host pool), then one engine run.  The measured region is one window for the whole job: it opens
  // when the last group finishes its warm-up (at that run's completion) and closes at the completion
  // of the `g_step_limit`-th run after that (or the first completion `seconds` later), whichever
  // groups those runs fall in; every run that completes inside the window is counted, with the game
  // counters of the host advance that loaded it.  The groups keep the GPU under the same load for the
  // whole window: a group leaves only when it sees the window closed.
  const long round_limit = g_step_rounds;
  const bool by_rounds = round_limit > 0;
  const bool by_steps = !by_rounds && g_step_limit > 0;
  const long step_limit = g_step_limit;
  std::mutex clock_mu;
  int phase = 0, groups_ready = 0;   // 0 advance + warm-up, 1 measuring, 2 over (guarded by clock_mu)
  int anchor = -1, anchor_lane = 0;  // by_rounds: the group (and lane) whose completions open and close the window
  long counted = 0, anchor_rounds = 0;
  std::chrono::steady_clock::time_point t0{}, t1{};
  // completion instants, seconds since t0 (guarded by clock_mu): of the counted batches, or (by_rounds) of the anchor's
  std::vector<double> done_at;
  std::atomic<bool> failed{false};
  std::atomic<long> phase_hist_store[16];
  for (auto& c : phase_hist_store) c.store(0);
  std::atomic<long>* phase_hist = getenv("P3HOST_PHASE_STATS") ? phase_hist_store : nullptr;
  for (int h = 0; h < NG; ++h) {
    halves[h].adv_left = g_advance_limit * NL;   // (a game in its opening has one evaluation in flight: every NL-th batch)
    halves[h].prev = group_totals(halves[h]);
  }
  for (int h = 0; h < NG; ++h)
    for (int l = 0; l < NL; ++l) {
      halves[h].lanes[l]->driver = std::thread([&, h, l] {
        Half& H = halves[h];
        Lane& L = *H.lanes[l];
        for (;;) {
          {
            std::unique_lock<std::mutex> tl(H.turn_mu);
            H.waiting.fetch_add(1);
            H.turn_cv.wait(tl, [&] { return H.turn == l || H.quit; });
            H.waiting.fetch_sub(1);
            if (H.quit) break;
          }
          const auto a0 = std::chrono::steady_clock::now();
          host_phase(h, l);
          {   // the counters as the host phase left them (games handed over as stragglers: as of their last settled phase)
            const GameStats now = group_totals(H);
            L.delta.moves = now.moves - H.prev.moves; L.delta.games = now.games - H.prev.games;
            L.delta.evals = now.evals - H.prev.evals; L.delta.black_wins = now.black_wins - H.prev.black_wins;
            L.delta.cache_hits = now.cache_hits - H.prev.cache_hits;
            L.delta.bias_entries_pruned = now.bias_entries_pruned - H.prev.bias_entries_pruned;
            L.delta.bias_adj_abs_sum = now.bias_adj_abs_sum - H.prev.bias_adj_abs_sum;
            H.prev = now;
            L.past_opening = 0;
            for (uint8_t po : H.snap_past_opening) L.past_opening += po;
            L.in_opening = L.past_opening < (long)H.games.size();
          }
          L.host_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count();
          if (phase_hist) {   // P3HOST_PHASE_STATS=1: how long the host phases take (a slow game holds up its whole group)
            const int b = std::min(15, (int)(L.host_seconds * 1e3));
            ++phase_hist[b];
          }
          {
            std::lock_guard<std::mutex> tl(H.turn_mu);
            H.turn = (l + 1) % NL;
          }
          H.turn_cv.notify_all();
          const auto r0 = std::chrono::steady_clock::now();
          const bool ok = L.eval->Run();
          const auto r1 = std::chrono::steady_clock::now();
          if (!ok) failed.store(true);
          bool over = false;
          {
            std::lock_guard<std::mutex> cl(clock_mu);
            if (!ok) H.ok = false;
            // a run belongs to the window by its completion instant, not by when this thread got the lock: r1 is taken
            // before the lock, so a run that completed before the window opened (or after it closed) can arrive here
            // with the phase already changed
            const bool inside = r1 > t0 && (phase == 1 || (phase == 2 && by_rounds && r1 <= t1));
            if (inside && phase != 0) {
              ++H.measured_batches;
              ++counted;
              H.counted.moves += L.delta.moves; H.counted.games += L.delta.games;
              H.counted.evals += L.delta.evals; H.counted.black_wins += L.delta.black_wins;
              H.counted.cache_hits += L.delta.cache_hits;
              H.counted.bias_entries_pruned += L.delta.bias_entries_pruned;
              H.counted.bias_adj_abs_sum += L.delta.bias_adj_abs_sum;
              H.gpu_seconds += std::chrono::duration<double>(r1 - r0).count();
              H.host_seconds += L.host_seconds;
              if (phase == 1) {
                bool enough;
                if (by_rounds) {
                  // a round = one batch of every lane of every group; the anchor is ONE lane's completions
                  const bool mine = h == anchor && l == anchor_lane;
                  if (mine) { ++anchor_rounds; done_at.push_back(std::chrono::duration<double>(r1 - t0).count()); }
                  enough = mine && anchor_rounds >= round_limit;
                } else {
                  done_at.push_back(std::chrono::duration<double>(r1 - t0).count());
                  enough = by_steps ? counted >= step_limit : std::chrono::duration<double>(r1 - t0).count() >= seconds;
                }
                if (enough) { phase = 2; t1 = r1; }
              }
            } else if (phase == 0 && !H.ready) {
              const bool in_opening = H.adv_left > 0 && L.in_opening;
              if (in_opening) { --H.adv_left; ++H.advance_batches; }
              else { H.adv_left = 0; ++H.warm; }
              if (H.warm >= warmup_batches || failed.load()) {
                H.ready = true;
                H.past_opening_at_start = L.past_opening;
                if (++groups_ready == NG) { phase = failed.load() ? 2 : 1; t0 = t1 = r1; anchor = h; anchor_lane = l; }
              }
            }
            if (phase == 1 && failed.load()) { phase = 2; t1 = r1; }
            over = phase == 2;
          }
          if (over) {
            {
              std::lock_guard<std::mutex> tl(H.turn_mu);
              H.quit = true;
            }
            H.turn_cv.notify_all();
            break;
          }
        }
      });
    }
  for (auto& H : halves)
    for (auto& L : H.lanes) L->driver.join();
  g_last_handed_over_phases = g_last_handed_over_games = 0;
  for (auto& H : halves) {
    if (H.late) { pool.Wait(H.late); H.late.reset(); }   // before the games go away
    g_last_handed_over_phases += H.handed_over_phases;
    g_last_handed_over_games += H.handed_over_games;
  }
  if (phase_hist) {
    std::fprintf(stderr, "host phases by duration (ms, last bin >= 15):");
    for (int b = 0; b < 16; ++b) std::fprintf(stderr, " %ld", phase_hist[b].load());
    std::fprintf(stderr, "\n");
  }
  int rc = 0;
  for (auto& H : halves)
    if (!H.ok) {
      rc = 2;
      if (err) snprintf(err, 256, "engine run failed");
    }
  if (recorder) recorder->Flush();
  g_first_game_digests.clear();
  for (auto& H : halves)
    for (auto& g : H.games) g_first_game_digests.push_back(g->first_game_digest());
  g_last_reuse_added = reuse->added();
  g_last_examples = recorder ? recorder->examples() : 0;
  if (out) {
    std::memset(out, 0, sizeof *out);
    g_last_bias_pruned = 0;
    g_last_bias_adj = 0;
    for (auto& H : halves) {
      out->positions += H.counted.evals;
      out->moves += H.counted.moves;
      out->games += H.counted.games;
      out->black_wins += H.counted.black_wins;
      out->cache_hits += H.counted.cache_hits;
      out->batches += H.measured_batches;
      out->gpu_seconds += H.gpu_seconds;
      out->host_seconds += H.host_seconds;
      out->advance_batches += H.advance_batches;
      out->games_past_opening += H.past_opening_at_start;
      g_last_bias_pruned += H.counted.bias_entries_pruned;
      g_last_bias_adj += H.counted.bias_adj_abs_sum;
    }
    out->seconds = std::chrono::duration<double>(t1 - t0).count();
    out->seconds_fit = out->seconds;
    out->rounds = anchor_rounds;
    {
      // completion k (1-based) at done_at[k - 1], the window's opening = completion 0 at 0
      // (by rounds: the anchor group's completions, one per round)
      const size_t n = done_at.size() + 1;
      if (n >= 4) {
        double sx = 0, sy = 0, sxx = 0, sxy = 0;
        for (size_t k = 0; k < n; ++k) {
          const double x = (double)k, y = k == 0 ? 0.0 : done_at[k - 1];
          sx += x; sy += y; sxx += x * x; sxy += x * y;
        }
        const double slope = (n * sxy - sx * sy) / (n * sxx - sx * sx);
        if (slope > 0) out->seconds_fit = slope * (double)(n - 1);
      }
    }
  }
  return rc;
}

// The reference's engine micro-benchmark, nn::Benchmark (cc/nn/engine/benchmark_engine.cc:77-108), over
// any library exporting the C ABI: 100 warm-up RunInference calls, then at most 1001 rounds (the
// reference's `num_inferences > 1000` bound) of LoadBatch x B -> RunInference -> GetBatch x B on ONE
// thread, the clock around RunInference only (its `elapsed_us`); `loop_seconds` is the whole timed loop,
// loads and gets included.  The reference walks a dataset that is not in its repository; here the
// rounds cycle through `n_feats` caller-supplied positions.  One difference, by design of the engine:
// this engine evaluates loaded slots only (the TRT engine always runs its static batch), so the batch is
// loaded once before the warm-up runs and they do the same work as the timed ones.
struct p3host_engine_benchmark_stats {
  long rounds, positions;
  double avg_run_us;        // mean of the per-round RunInference time (DefaultStats "avg_us")
  double loop_seconds;      // wall time of the timed loop: loads + runs + gets
  double checksum;          // sum over every fetched result of move_probs[argmax] (keeps the gets honest)
};
int p3host_engine_benchmark(const char* engine_lib, const char* weights, int device, int batch,
                            const p3hip_features* feats, int n_feats, int warmup_runs, int max_rounds,
                            p3host_engine_benchmark_stats* out, char* err) {
  if (!engine_lib || !engine_lib[0] || !feats || n_feats <= 0 || batch <= 0) {
    if (err) snprintf(err, 256, "engine_benchmark: engine library, positions and batch size are required");
    return 1;
  }
  HipEvaluator e;
  if (!e.Open(engine_lib, weights, batch, device)) {
    if (err) snprintf(err, 256, "%s", e.err.c_str());
    return 1;
  }
  for (int b = 0; b < batch; ++b) e.Load(b, feats[b % n_feats]);
  for (int i = 0; i < warmup_runs; ++i)
    if (!e.Run()) { if (err) snprintf(err, 256, "%s", e.err.c_str()); return 2; }
  p3hip_result r;
  double run_us = 0, checksum = 0;
  long rounds = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (; rounds < max_rounds; ++rounds) {
    const long off = rounds * (long)batch;
    for (int b = 0; b < batch; ++b) e.Load(b, feats[(off + b) % n_feats]);
    const auto s = std::chrono::steady_clock::now();
    if (!e.Run()) { if (err) snprintf(err, 256, "%s", e.err.c_str()); return 2; }
    run_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s).count();
    for (int b = 0; b < batch; ++b) {
      e.Get(b, r);
      float best = r.move_probs[0];
      for (int i = 1; i < kNumMoves; ++i) best = std::max(best, r.move_probs[i]);
      checksum += best;
    }
  }
  if (out) {
    out->rounds = rounds;
    out->positions = rounds * (long)batch;
    out->avg_run_us = rounds ? run_us / rounds : 0;
    out->loop_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out->checksum = checksum;
  }
  return 0;
}

// Plays ONE game to the end with the NullEvaluator or the HIP engine on a single thread and
// returns its move list (deterministic given the seed): plumbing test of BASELINE configs[0].
int p3host_selfplay_one_game(const char* engine_lib, const char* weights, int default_n, int default_k,
                             int max_moves, uint64_t seed, int* moves_out, int max_out, float* bscore,
                             float* wscore, long* evals, char* err) {
  SelfPlayConfig cfg;
  cfg.default_n = default_n; cfg.default_k = default_k;
  cfg.selected_n = default_n; cfg.selected_k = default_k;
  cfg.max_moves = max_moves;
  cfg.init_state_sampling = false;   // one empty-board game at komi 7.5
  cfg.bias_cache_lambda = g_bias_cache_lambda;
  cfg.bias_cache_alpha = g_bias_cache_alpha;
  std::unique_ptr<Evaluator> ev;
  if (!engine_lib || !engine_lib[0]) {
    ev.reset(new NullEvaluator());
  } else {
    auto* e = new HipEvaluator();
    ev.reset(e);
    if (!e->Open(engine_lib, weights, 1, 0)) {
      if (err) snprintf(err, 256, "%s", e->err.c_str());
      return -1;
    }
  }
  GameRunner g(cfg, seed);
  p3hip_features f;
  p3hip_result r;
  while (g.stats().games == 0) {
    g.AdvanceToEval(&f);
    if (g.stats().games > 0) break;
    ev->Load(0, f);
    if (!ev->Run()) return -2;
    ev->Get(0, r);
    g.DeliverResult(r);
  }
  int n_out = 0;
  const std::vector<Move>& rec = g.last_moves();
  for (size_t i = Game::kMoveOffset; i < rec.size() && n_out < max_out; ++i)
    moves_out[n_out++] = (MoveIdx(rec[i].loc) + 1) * (rec[i].color == kBlack ? 1 : -1);
  if (bscore) *bscore = g.last_result().bscore;
  if (wscore) *wscore = g.last_result().wscore;
  if (evals) *evals = g.stats().evals;
  g_last_bias_pruned = g.stats().bias_entries_pruned;
  g_last_bias_adj = g.stats().bias_adj_abs_sum;
  return n_out;
}

// Test hook for the evaluation cache's reserve-then-fill protocol: `nreq` lookups over `nkeys` distinct keys,
// once in the one-at-a-time order (Find, on a miss Insert before the next lookup) and once with up to `depth`
// results outstanding (Probe, on a miss InsertPending at once and Fill `depth` - 1 lookups later).  Returns the
// number of lookups the two orders classify differently (hit / miss; a lookup that finds a reserved entry counts
// as the hit it would have been) or answer with a different result, plus 1000000 if the final contents differ.
long p3host_test_eval_cache_pipeline(int capacity, int nkeys, int nreq, int depth, uint64_t seed) {
  EvalCache seq(capacity), pipe(capacity);
  auto key_of = [](int k) {
    EvalCache::Key key{0x1234567ull * (uint64_t)(k + 1), {kNoopLoc, kNoopLoc, kNoopLoc, kNoopLoc, kNoopLoc}, 7.5f, kBlack};
    return key;
  };
  auto result_of = [](int k) {
    p3hip_result r;
    std::memset(&r, 0, sizeof r);
    r.err2_outcome = (float)k;
    return r;
  };
  struct Outstanding { uint64_t id; int key; int due; };
  std::deque<Outstanding> out;
  long diff = 0;
  uint64_t x = seed | 1;
  for (int i = 0; i < nreq; ++i) {
    while (!out.empty() && out.front().due <= i) {
      pipe.Fill(out.front().id, result_of(out.front().key));
      out.pop_front();
    }
    x = HashEvaluator::Mix(x);
    const int k = (int)(x % (uint64_t)nkeys);
    const p3hip_result* a = seq.Find(key_of(k));
    if (!a) seq.Insert(key_of(k), result_of(k));
    const p3hip_result* b = nullptr;
    uint64_t owner = 0;
    const EvalCache::Lookup lk = pipe.Probe(key_of(k), &b, &owner);
    if (lk == EvalCache::Lookup::kMiss) {
      const uint64_t id = pipe.InsertPending(key_of(k));
      out.push_back({id, k, i + 1 + (int)((x >> 20) % (uint64_t)depth)});
      std::stable_sort(out.begin(), out.end(), [](const Outstanding& p, const Outstanding& q) { return p.due < q.due; });
    }
    const bool hit_seq = a != nullptr, hit_pipe = lk != EvalCache::Lookup::kMiss;
    if (hit_seq != hit_pipe) ++diff;
    else if (a && a->err2_outcome != (float)k) ++diff;
    else if (lk == EvalCache::Lookup::kHit && b->err2_outcome != (float)k) ++diff;
    else if (lk == EvalCache::Lookup::kPending) {
      bool owned = false;
      for (const Outstanding& o : out) owned |= o.id == owner && o.key == k;
      if (!owned) ++diff;
    }
  }
  for (const Outstanding& o : out) pipe.Fill(o.id, result_of(o.key));
  for (int k = 0; k < nkeys; ++k) {   // same contents (these lookups stamp both caches alike)
    const p3hip_result* a = seq.Find(key_of(k));
    const p3hip_result* b = pipe.Find(key_of(k));
    if ((a != nullptr) != (b != nullptr) || (a && a->err2_outcome != b->err2_outcome)) return diff + 1000000;
  }
  return diff;
}

// Test hook for several playouts in flight (GameRunner::TryAdvance): plays `num_games` consecutive games of ONE
// game runner over the hash evaluator, with up to `depth` evaluation requests outstanding and the results handed
// back late — a result is delivered when nothing more can start, or earlier on a coin flip of `sched_seed`.
// out[0] = a digest of every finished game's moves and result, out[1] = evaluations, out[2] = cache hits,
// out[3] = moves, out[4] = the largest number of requests that were outstanding at once.
// depth 1 is the one-at-a-time order; every depth and every schedule must return the same out[0..3].
int p3host_test_game_inflight(int default_n, int default_k, int selected_n, int selected_k, int max_moves, uint64_t seed,
                              int depth, int num_games, int cache_entries, int init_state_sampling, int early_stopping,
                              uint64_t sched_seed, uint64_t* out) {
  SelfPlayConfig cfg;
  cfg.default_n = default_n; cfg.default_k = default_k;
  cfg.selected_n = selected_n; cfg.selected_k = selected_k;
  cfg.max_moves = max_moves;
  cfg.cache_entries_per_game = cache_entries;
  cfg.init_state_sampling = init_state_sampling != 0;
  cfg.early_stopping_enabled = early_stopping != 0;
  cfg.bias_cache_lambda = g_bias_cache_lambda;
  cfg.bias_cache_alpha = g_bias_cache_alpha;
  cfg.sel_mult_base = g_sel_mult_base;
  cfg.fork_params = ForkParams::ForReuse(g_use_seen_state_prob);
  auto reuse = std::make_unique<ReuseBuffer>(seed ^ 0x676f6578706c6f69ull);
  cfg.reuse = reuse.get();
  GameRunner g(cfg, seed);
  std::deque<p3hip_result> queue;   // results of the outstanding requests, oldest first
  uint64_t digest = 0xcbf29ce484222325ull, sched = sched_seed | 1, max_out = 0;
  long folded = 0;
  GameStats at_fold;   // counters when the last game finished (the next game's first request included, at every depth)
  auto fold = [&] {
    while (folded < g.stats().games) {   // (a finished game's record is replaced when the next one finishes)
      for (const Move& m : g.last_moves())
        digest = (digest ^ (uint64_t)(MoveIdx(m.loc) * 4 + (int)m.color + 2)) * 0x100000001b3ull;
      uint32_t bs, ws;
      const float b = g.last_result().bscore, w = g.last_result().wscore;
      std::memcpy(&bs, &b, 4);
      std::memcpy(&ws, &w, 4);
      digest = (digest ^ bs) * 0x100000001b3ull;
      digest = (digest ^ ws) * 0x100000001b3ull;
      ++folded;
      at_fold = g.stats();
    }
  };
  p3hip_features f;
  std::memset(&f, 0, sizeof f);   // FillFeatures writes every field; the hash also covers the padding
  while (g.stats().games < num_games) {
    sched = HashEvaluator::Mix(sched);
    const bool try_issue = queue.empty() || (sched & 3) != 0;
    if (try_issue && g.TryAdvance(&f, depth)) {
      queue.emplace_back();
      HashEvaluator::Evaluate(f, queue.back());
      max_out = std::max<uint64_t>(max_out, queue.size());
      fold();
      continue;
    }
    fold();
    if (queue.empty()) return 1;   // blocked with nothing outstanding: a scheduling bug
    g.DeliverResult(queue.front());
    queue.pop_front();
    fold();
  }
  out[0] = digest;
  out[1] = (uint64_t)at_fold.evals;
  out[2] = (uint64_t)at_fold.cache_hits;
  out[3] = (uint64_t)at_fold.moves;
  out[4] = max_out;
  return 0;
}

}  // extern "C"

// ---- evaluation matches (cc/eval) ------------------------------------------------------------
#include "eval_match.h"
#include "threaded_search.h"

extern "C" {

struct p3host_eval_stats {
  int games, cur_wins, cand_wins, draws, resignations;
  long moves, visits, collisions, positions, batches;
  double seconds;
  long cache_hits;                        // evaluations served by the NN cache (not in `positions`)
  float winrate, c95, rel_elo, elo_c95;   // cand's win rate +- 95 %, relative Elo +- (eval/main.cc:459-471)
};

static int g_eval_mode = 0, g_eval_qfn = 2, g_eval_nfn = 1, g_eval_collision = 0, g_eval_detector = 0;
static int g_eval_descent = 0, g_eval_mcgs = 0, g_eval_cache_per_game = 256;
static float g_eval_max_o = 1.0f, g_eval_bias_lambda = 0.0f, g_eval_bias_alpha = 0.8f;
static std::string g_eval_cur_cfg, g_eval_cand_cfg, g_eval_recorder_dir, g_eval_res_path;
std::string g_eval_cur_cli, g_eval_cand_cli;   // --cur_* / --cand_* flags as "key: value" lines
// Parallel-search knobs of subsequent p3host_eval_match calls (defaults = player_config.h:76-108:
// concurrent rounds, virtual_loss_soft, virtual_visit, abort, noop).
void p3host_eval_set_search(int mode, int q_fn, int n_fn, int collision, int detector) {
  g_eval_mode = mode; g_eval_qfn = q_fn; g_eval_nfn = n_fn; g_eval_collision = collision; g_eval_detector = detector;
}
// descent policy (0 deterministic, 1 bu_uct + max_o_ratio), graph search, bias cache (lambda 0 = off)
// and NN-cache entries per game and player (--cache_size; 0 = off) of subsequent matches
void p3host_eval_set_search_ex(int descent, float max_o_ratio, int use_mcgs, float bias_lambda, float bias_alpha,
                               int cache_entries_per_game) {
  g_eval_descent = descent; g_eval_max_o = max_o_ratio; g_eval_mcgs = use_mcgs;
  g_eval_bias_lambda = bias_lambda; g_eval_bias_alpha = bias_alpha;
  g_eval_cache_per_game = cache_entries_per_game;
}
// --cur_config / --cand_config player files ("" = none), --recorder_path (SGFs of the games under
// <dir>/sgf; "" = none), --res_write_path (relative Elo; "" = none) of subsequent matches
void p3host_eval_set_paths(const char* cur_config, const char* cand_config, const char* recorder_dir,
                           const char* res_write_path) {
  g_eval_cur_cfg = cur_config ? cur_config : "";
  g_eval_cand_cfg = cand_config ? cand_config : "";
  g_eval_recorder_dir = recorder_dir ? recorder_dir : "";
  g_eval_res_path = res_write_path ? res_write_path : "";
}

extern "C" void p3host_eval_set_player_flags(const char* cur, const char* cand) {
  g_eval_cur_cli = cur ? cur : "";
  g_eval_cand_cli = cand ? cand : "";
}

static bool MakeEvalPlayerConfigs(int visits_per_move, int leaves_per_round, EvalPlayerConfig pc[2], char* err) {
  EvalPlayerConfig base;
  base.n = visits_per_move;
  base.num_threads_per_game = leaves_per_round;
  base.search_mode = (SearchMode)g_eval_mode;
  base.q_fn = (QFn)g_eval_qfn;
  base.n_fn = (NFn)g_eval_nfn;
  base.collision_policy = (CollisionPolicy)g_eval_collision;
  base.collision_detector = (CollisionDetector)g_eval_detector;
  base.descent_policy = (DescentPolicy)g_eval_descent;
  base.max_o_ratio = g_eval_max_o;
  base.use_mcgs = g_eval_mcgs != 0;
  base.use_bias_cache = g_eval_bias_lambda > 0.0f;
  base.bias_cache_lambda = g_eval_bias_lambda;
  base.bias_cache_alpha = g_eval_bias_alpha;
  pc[0] = pc[1] = base;
  std::string e;
  if (!g_eval_cur_cfg.empty() && !ParsePlayerConfig(g_eval_cur_cfg, &pc[0], &e)) { if (err) snprintf(err, 256, "cur config: %s", e.c_str()); return false; }
  if (!g_eval_cand_cfg.empty() && !ParsePlayerConfig(g_eval_cand_cfg, &pc[1], &e)) { if (err) snprintf(err, 256, "cand config: %s", e.c_str()); return false; }
  // per-player command-line flags win over the file (eval/main.cc:146-246, 299-320)
  if (!g_eval_cur_cli.empty() && !ApplyPlayerConfigText(g_eval_cur_cli, &pc[0], &e)) { if (err) snprintf(err, 256, "cur flags: %s", e.c_str()); return false; }
  if (!g_eval_cand_cli.empty() && !ApplyPlayerConfigText(g_eval_cand_cli, &pc[1], &e)) { if (err) snprintf(err, 256, "cand flags: %s", e.c_str()); return false; }
  return true;
}

// match bookkeeping shared by both drivers: results, Elo, result file, SGF batch
struct EvalGameOutcome { int cur_result; bool resigned; Color winner; std::string sgf; };
static void FinishEvalMatch(const std::vector<EvalGameOutcome>& games, p3host_eval_stats* out) {
  for (const auto& g : games) {
    out->cur_wins += g.cur_result > 0;
    out->cand_wins += g.cur_result < 0;
    out->draws += g.cur_result == 0;
    out->resignations += g.resigned;
  }
  out->games = (int)games.size();
  const MatchSummary m = SummarizeMatch(out->cand_wins, out->games);
  out->winrate = m.winrate; out->c95 = m.c95; out->rel_elo = m.rel_elo; out->elo_c95 = m.elo_c95;
  if (!g_eval_res_path.empty()) WriteMatchResult(g_eval_res_path, m.rel_elo);
  if (!g_eval_recorder_dir.empty()) {   // GameRecorder::Create(recorder_path, ..., "EVAL_<cur>_<cand>"), eval/main.cc:418-422
    ::mkdir(g_eval_recorder_dir.c_str(), 0755);
    ::mkdir((g_eval_recorder_dir + "/sgf").c_str(), 0755);
    SgfRecorder rec(g_eval_recorder_dir + "/sgf", 0, "EVAL_cur_cand");
    for (const auto& g : games) rec.RecordGame(g.sgf);
    rec.Flush();
  }
}
static std::string EvalGameSgf(const Game& game, bool cur_is_black, bool resigned, Color winner) {
  Game::Result r = game.result();
  if (resigned) { r = Game::Result(); r.winner = winner; r.by_resign = true; }
  return SgfGameString(game.komi(), r, game.moves(), Game::kMoveOffset, cur_is_black ? "cur" : "cand",
                       cur_is_black ? "cand" : "cur");
}

// Evaluation-match drivers: keep the NN cache in the engine's HBM table (p3hip_cache_*, 2^log2_entries entries per
// engine: one table for all games and workers of a player) — instead of NNInterface's per-thread host LRUs in the
// thread-per-game driver, behind the per-game host caches in the batching one; 0 = host caches only (the reference's
// arrangement).
static std::atomic<int> g_device_nn_cache_log2{0};
static std::atomic<long> g_device_nn_cache_hits{0}, g_device_nn_cache_lookups{0};
void p3host_set_device_nn_cache(int log2_entries) { g_device_nn_cache_log2.store(log2_entries < 0 ? 0 : log2_entries); }
long p3host_device_nn_cache_hits() { return g_device_nn_cache_hits.load(); }   // of the last thread-per-game match
long p3host_device_nn_cache_lookups() { return g_device_nn_cache_lookups.load(); }

// Plays `num_games` evaluation games between two networks with the batch parallel search
// (eval.cc:103-518).  engine_lib NULL/"" = NullEvaluator for both players.  One engine
// instance per player, batch = num_games * leaves_per_round slots; every game keeps an NN cache
// per player (NNKey with five last moves, nn_interface.cc:92-132), so a position the search
// reaches again — in a later move's search, or through the other player's tree — costs no slot.
int p3host_eval_match(const char* engine_lib, const char* cur_weights, const char* cand_weights, int device,
                      int num_games, int visits_per_move, int leaves_per_round, int max_moves, int num_threads,
                      uint64_t seed, p3host_eval_stats* out, char* err) {
  EvalPlayerConfig pc[2];
  if (!MakeEvalPlayerConfigs(visits_per_move, leaves_per_round, pc, err)) return 3;
  const int slots = num_games * std::max(pc[0].num_threads_per_game, pc[1].num_threads_per_game);
  std::unique_ptr<Evaluator> ev[2];
  const bool use_null = !engine_lib || !engine_lib[0];
  for (int e = 0; e < 2; ++e) {
    if (use_null) {
      ev[e].reset(new NullEvaluator());
    } else {
      auto* h = new HipEvaluator();
      ev[e].reset(h);
      if (!h->Open(engine_lib, e == 0 ? cur_weights : cand_weights, slots, device, P3HIP_FLAG_SHARED_DEVICE)) {   // the two players' passes run concurrently
        if (err) snprintf(err, 256, "%s", h->err.c_str());
        return 1;
      }
    }
  }
  const int dc_log2 = g_device_nn_cache_log2.load();
  bool device_cache = false;
  if (dc_log2 > 0 && !use_null) {
    device_cache = ev[0]->EnableDeviceCache(dc_log2) && ev[1]->EnableDeviceCache(dc_log2);
    if (!device_cache) {
      if (err) snprintf(err, 256, "the engine has no on-device NN cache (p3hip_cache_enable)");
      return 1;
    }
  }
  std::atomic<long> device_lookups{0}, device_hits{0};
  std::vector<std::unique_ptr<EvalGame>> games;
  for (int g = 0; g < num_games; ++g)
    games.emplace_back(new EvalGame(g, pc[0], pc[1], max_moves, seed * 0x9E3779B97F4A7C15ull + (uint64_t)g * 0xBF58476D1CE4E5B9ull));
  std::vector<EvalCache> caches;   // [game * 2 + engine]
  for (int g = 0; g < 2 * num_games; ++g) caches.emplace_back(g_eval_cache_per_game, 5);
  WorkerPool pool(num_threads > 0 ? num_threads : 1);
  std::vector<int> want(num_games, 0), base(num_games, 0), need(num_games, 0);
  std::vector<std::vector<int>> slot_of(num_games);          // evaluation i of game g -> engine slot offset, -1 = cached
  std::vector<std::vector<EvalCache::Key>> key_of(num_games);
  // copies of the cached results found in advance(): the deliveries below interleave with Insert(),
  // which may evict the very entry a later evaluation of the same round hit (cache smaller than a round)
  std::vector<std::vector<p3hip_result>> hit_of(num_games);
  std::vector<p3hip_features> feats(2 * (size_t)slots);
  long positions = 0, batches = 0;
  std::atomic<long> cache_hits{0};
  const auto t0 = std::chrono::steady_clock::now();
  // Advances game g until it wants evaluations the cache cannot answer (or has finished).
  auto advance = [&](int g) {
    for (;;) {
      want[g] = games[g]->Step();
      need[g] = 0;
      if (want[g] <= 0) return;
      const int e = games[g]->active_engine();
      EvalCache& cache = caches[2 * g + e];
      slot_of[g].assign(want[g], -1);
      key_of[g].resize(want[g]);
      if ((int)hit_of[g].size() < want[g]) hit_of[g].resize(want[g]);
      for (int i = 0; i < want[g]; ++i) {
        key_of[g][i] = cache.MakeKey(games[g]->eval_position(i), games[g]->eval_color_of(i));
        if (const p3hip_result* hit = cache.Find(key_of[g][i])) hit_of[g][i] = *hit;
        else slot_of[g][i] = need[g]++;
      }
      if (need[g] > 0) return;
      for (int i = 0; i < want[g]; ++i) games[g]->DeliverCached(i, hit_of[g][i]);   // all cached
      cache_hits.fetch_add(want[g], std::memory_order_relaxed);
    }
  };
  pool.ParallelFor(num_games, [&](int g) { advance(g); });
  std::vector<int> eng_of(num_games, -1);
  for (;;) {
    // Every game wants leaves from the engine of its side to move: fill both engines' batches,
    // run the two forward passes CONCURRENTLY (own streams on the same device), hand back.
    int total[2] = {0, 0};
    for (int g = 0; g < num_games; ++g) {
      base[g] = -1;
      eng_of[g] = -1;
      if (want[g] <= 0) continue;
      const int e = games[g]->active_engine();
      eng_of[g] = e;
      base[g] = total[e];
      total[e] += need[g];
    }
    if (total[0] + total[1] == 0) break;
    pool.ParallelFor(num_games, [&](int g) {
      if (base[g] < 0) return;
      const int e = eng_of[g];
      p3hip_features* f = &feats[(size_t)e * slots + base[g]];
      for (int i = 0; i < want[g]; ++i) {
        if (slot_of[g][i] < 0) continue;
        games[g]->FillEval(i, f + slot_of[g][i]);
        if (device_cache) {
          uint64_t lo, hi;
          EvalCache::Digest(key_of[g][i], &lo, &hi);
          ev[e]->LoadKeyed(base[g] + slot_of[g][i], f[slot_of[g][i]], lo, hi, (int)games[g]->eval_symmetry(i));
        } else {
          ev[e]->Load(base[g] + slot_of[g][i], f[slot_of[g][i]]);
        }
      }
    });
    bool ok[2] = {true, true};
    std::thread second;
    if (total[1] > 0) second = std::thread([&] { ok[1] = ev[1]->Run(); });
    if (total[0] > 0) ok[0] = ev[0]->Run();
    if (second.joinable()) second.join();
    if (!ok[0] || !ok[1]) {
      if (err) snprintf(err, 256, "engine run failed");
      return 2;
    }
    positions += total[0] + total[1];
    batches += (total[0] > 0) + (total[1] > 0);
    pool.ParallelFor(num_games, [&](int g) {
      if (base[g] < 0) return;
      const int e = eng_of[g];
      EvalCache& cache = caches[2 * g + e];
      p3hip_result r;
      for (int i = 0; i < want[g]; ++i) {
        if (slot_of[g][i] < 0) {
          games[g]->DeliverCached(i, hit_of[g][i]);
          cache_hits.fetch_add(1, std::memory_order_relaxed);
          continue;
        }
        if (device_cache) {
          int sym = (int)games[g]->eval_symmetry(i);
          bool hit = false;
          ev[e]->GetKeyed(base[g] + slot_of[g][i], r, &sym, &hit);
          device_lookups.fetch_add(1, std::memory_order_relaxed);
          if (hit) device_hits.fetch_add(1, std::memory_order_relaxed);
          games[g]->DeliverUnder(i, r, (Symmetry)sym);   // un-symmetrises by the symmetry of the stored result
        } else {
          ev[e]->Get(base[g] + slot_of[g][i], r);
          games[g]->Deliver(i, r);   // un-symmetrises r in place
        }
        cache.Insert(key_of[g][i], r);
      }
      advance(g);
    });
  }
  if (out) {
    std::memset(out, 0, sizeof *out);
    std::vector<EvalGameOutcome> outcomes;
    for (auto& g : games) {
      outcomes.push_back({g->cur_result(), g->resigned(), g->winner(),
                          EvalGameSgf(g->game(), g->cur_is_black(), g->resigned(), g->winner())});
      out->moves += g->num_moves();
      out->visits += g->visits();
      out->collisions += g->collisions();
    }
    FinishEvalMatch(outcomes, out);
    out->positions = positions;
    out->batches = batches;
    out->cache_hits = cache_hits.load();
    g_device_nn_cache_hits.store(device_hits.load());
    g_device_nn_cache_lookups.store(device_lookups.load());
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return 0;
}

// The reference's own shape of the match (eval/main.cc:380-452, eval.cc:103-518): one OS thread
// per game, each player's engine behind its own NNInterface with kExplicit signalling,
// num_shared_search_tasks = num_games, batch = num_games * threads_per_game slots and the
// per-thread NN cache on (cache_size entries in total per interface), the side to move running
// the threaded mcts::Search (threaded_search.h) on slot range [game * T, (game + 1) * T).
int p3host_eval_match_threads(const char* engine_lib, const char* cur_weights, const char* cand_weights, int device,
                              int num_games, int visits_per_move, int threads_per_game, int max_moves,
                              long cache_size, uint64_t seed, p3host_eval_stats* out, char* err) {
  EvalPlayerConfig pc[2];
  if (!MakeEvalPlayerConfigs(visits_per_move, threads_per_game, pc, err)) return 3;
  const bool use_null = !engine_lib || !engine_lib[0];
  std::unique_ptr<NNInterface> nn[2];
  for (int e = 0; e < 2; ++e) {
    const int batch = num_games * pc[e].num_threads_per_game;
    std::unique_ptr<Evaluator> ev;
    if (use_null) {
      ev.reset(new NullEvaluator());
    } else {
      auto* h = new HipEvaluator();
      ev.reset(h);
      if (!h->Open(engine_lib, e == 0 ? cur_weights : cand_weights, batch, device, P3HIP_FLAG_SHARED_DEVICE)) {
        if (err) snprintf(err, 256, "%s", h->err.c_str());
        return 1;
      }
    }
    nn[e].reset(new NNInterface(batch, NNInterface::kTimeoutUs, (size_t)cache_size, std::move(ev),
                                NNInterface::SignalKind::kExplicit, num_games));
    const int dc = g_device_nn_cache_log2.load();
    if (dc > 0 && !use_null && !nn[e]->EnableDeviceCache(dc)) {
      if (err) snprintf(err, 256, "the engine has no on-device NN cache (p3hip_cache_enable)");
      return 1;
    }
  }
  std::vector<EvalGameOutcome> outcomes(num_games);
  std::vector<long> moves(num_games, 0), visits(num_games, 0), collisions(num_games, 0);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> threads;
  for (int g = 0; g < num_games; ++g)
    threads.emplace_back([&, g] {   // PlayEvalGame, eval.cc:103-518
      Probability prob(seed * 0x9E3779B97F4A7C15ull + (uint64_t)g * 0xBF58476D1CE4E5B9ull);
      const bool cur_is_black = g % 2 == 0;
      const int player_of[2] = {cur_is_black ? 0 : 1, cur_is_black ? 1 : 0};   // colour index (0 black) -> player
      Game game(7.5f, true);
      std::unique_ptr<BiasCache> bias[2];
      NodePool pool[2];
      TreeNode* tree[2];
      std::unique_ptr<ThreadedSearch> search[2];
      for (int s = 0; s < 2; ++s) {
        const EvalPlayerConfig& c = pc[player_of[s]];
        pool[s].set_graph(c.use_mcgs);
        if (c.use_bias_cache) bias[s].reset(new BiasCache(c.bias_cache_alpha, c.bias_cache_lambda));
        tree[s] = pool[s].GetOrCreate(game.board().hash(), kBlack, false);
        search[s].reset(new ThreadedSearch(nn[player_of[s]]->MakeSlot(g * c.num_threads_per_game), bias[s].get()));
      }
      Color color = kBlack, resigned = kEmpty;
      while (!game.IsGameOver() && game.num_moves() < max_moves) {
        const int s = color == kBlack ? 0 : 1;
        const EvalPlayerConfig& c = pc[player_of[s]];
        ThreadedSearch::Params p;
        p.num_threads = c.num_threads_per_game;
        p.total_visit_budget = c.time_ms > 0 ? (1 << 30) : c.n;
        p.total_visit_time_ms = c.time_ms;
        p.puct = MakeSearchPuctParams(c);
        p.score_util = MakeScoreUtility(c);
        p.fns = VirtualFns{c.q_fn, c.n_fn, c.vl_delta};
        p.descent = c.descent_policy;
        p.collision = c.collision_policy;
        p.detector = c.collision_detector;
        p.max_collision_retries = c.max_collision_retries;
        p.max_o_ratio = c.max_o_ratio;
        const ThreadedSearch::Result r = search[s]->Run(prob, game, &pool[s], tree[s], color, p);
        visits[g] += r.num_visits;
        collisions[g] += r.num_collisions;
        if (VOutcome(tree[s]) < kResignThreshold) { resigned = color; break; }   // eval.cc:277-282
        game.PlayMove(r.move, color);
        color = Opp(color);
        for (int t = 0; t < 2; ++t) {   // both trees follow the move, eval.cc:318-352
          TreeNode* next = tree[t]->child(MoveIdx(r.move));
          if (!next) next = pool[t].GetOrCreate(game.board().hash(), color, game.IsGameOver());
          pool[t].Reap(next);
          tree[t] = next;
          if (bias[t]) bias[t]->PruneUnused();
        }
      }
      // a finished game leaves both interfaces' signalling quorum (eval.cc:505-507)
      for (int s = 0; s < 2; ++s) nn[s]->MakeSlot(0).UnregisterSearchTask();
      Color winner;
      if (resigned != kEmpty) {
        winner = Opp(resigned);
      } else {
        game.WriteResult();
        winner = game.result().winner;
      }
      const int cur_result = winner == kEmpty ? 0 : ((winner == kBlack) == cur_is_black ? 1 : -1);
      outcomes[g] = {cur_result, resigned != kEmpty, winner, EvalGameSgf(game, cur_is_black, resigned != kEmpty, winner)};
      moves[g] = game.num_moves();
    });
  for (auto& t : threads) t.join();
  g_device_nn_cache_hits.store(nn[0]->device_cache_hits() + nn[1]->device_cache_hits());
  g_device_nn_cache_lookups.store(nn[0]->device_cache_lookups() + nn[1]->device_cache_lookups());
  if (out) {
    std::memset(out, 0, sizeof *out);
    FinishEvalMatch(outcomes, out);
    for (int g = 0; g < num_games; ++g) { out->moves += moves[g]; out->visits += visits[g]; out->collisions += collisions[g]; }
    out->batches = nn[0]->num_inferences() + nn[1]->num_inferences();
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return 0;
}

// core::RelativeElo / the match summary, for the tests
void p3host_match_summary(int num_cand_won, int num_games, float* out4) {
  const MatchSummary m = SummarizeMatch(num_cand_won, num_games);
  out4[0] = m.winrate; out4[1] = m.c95; out4[2] = m.rel_elo; out4[3] = m.elo_c95;
}
// parses a player config file; returns 0 and fills out[0..39] with its fields (order below), or 1 with a message
int p3host_parse_player_config(const char* path, float* out, char* err) {
  EvalPlayerConfig c;
  std::string e;
  if (!ParsePlayerConfig(path, &c, &e)) { if (err) snprintf(err, 256, "%s", e.c_str()); return 1; }
  const PuctParams sp = MakeSearchPuctParams(c), rp = MakeRootPuctParams(c);
  const ScoreUtilityParams su = MakeScoreUtility(c);
  const float v[40] = {(float)c.n, (float)c.num_threads_per_game, c.c_puct, c.c_puct_visit_scaling, c.root_fpu,
                       (float)c.var_scale_cpuct, (float)(int)c.q_fn, (float)(int)c.n_fn, (float)(int)c.collision_policy,
                       (float)(int)c.collision_detector, (float)(int)c.search_mode, (float)(int)c.descent_policy,
                       c.max_o_ratio, (float)c.use_mcgs, (float)c.use_bias_cache, (float)c.time_ms,
                       // 16..
                       (float)c.k, c.noise_scaling, (float)c.early_stopping_for_gumbel, (float)c.use_puct,
                       (float)c.use_puct_v, c.c_puct_v_2, c.tau, (float)c.use_lcb, c.score_weight, (float)(int)su.mode,
                       // 26..
                       (float)c.enable_m3_bonus, (float)c.var_scale_prior_visits, (float)c.m3_prior_visits, c.p_opt_weight,
                       (float)c.enable_pondering, (float)c.time_control_flags, c.vl_delta, (float)c.max_collision_retries,
                       // 34..: derived parameters
                       (float)(int)sp.kind, (float)(int)rp.kind, rp.c_puct_visit_scaling, rp.tau, (float)UsesParallelSearch(c),
                       c.bias_cache_alpha};
  std::memcpy(out, v, sizeof v);
  return 0;
}

// PuctScorer::ComputeScores on a hand-built node (tests): children given as (action, visits, v, v_var, v_m3,
// n_in_flight); pp = {c_puct, c_puct_visit_scaling, c_puct_v_2, use_puct_v, enable_var_scaling,
// var_scale_prior_visits, enable_m3_bonus, m3_prior_visits, p_opt_weight, root_fpu}; fns = {q_fn, n_fn, vl_delta}
void p3host_test_puct_scores(int node_n, float node_v, float node_v_var, const float* move_probs, const float* opt_probs,
                             int n_children, const int* actions, const int* visits, const float* child_v,
                             const float* child_v_var, const double* child_v_m3, const int* in_flight,
                             const float* ppv, const float* fns, int is_root, float* scores) {
  NodePool pool;
  TreeNode* node = pool.Create();
  node->n = node_n; node->v = node_v; node->v_var = node_v_var;
  std::memcpy(node->move_probs, move_probs, sizeof node->move_probs);
  std::memcpy(node->opt_probs, opt_probs, sizeof node->opt_probs);
  for (int i = 0; i < n_children; ++i) {
    TreeNode* ch = pool.Create();
    ch->v = child_v[i]; ch->v_var = child_v_var[i]; ch->v_m3 = child_v_m3[i];
    ch->n_in_flight = in_flight[i];
    node->children.push_back(ChildEdge{(int16_t)actions[i], visits[i], ch});
  }
  PuctParams pp;
  pp.c_puct = ppv[0]; pp.c_puct_visit_scaling = ppv[1]; pp.c_puct_v_2 = ppv[2]; pp.use_puct_v = ppv[3] != 0;
  pp.enable_var_scaling = ppv[4] != 0; pp.var_scale_prior_visits = (int)ppv[5]; pp.enable_m3_bonus = ppv[6] != 0;
  pp.m3_prior_visits = (int)ppv[7]; pp.p_opt_weight = ppv[8]; pp.root_fpu = ppv[9];
  const VirtualFns vf{(QFn)(int)fns[0], (NFn)(int)fns[1], fns[2]};
  PuctScoresAll(node, pp, is_root != 0, scores, vf);
}

// LeafEvaluator::EvaluateLeaf on a neutral evaluation (value 0, expected score 0), the situation of
// cc/mcts/__tests__/leaf_evaluator_test.cc: returns init_util_est
float p3host_test_evaluate_leaf(int color_to_move, int root_color, float root_score_est, int integral, float score_weight) {
  p3hip_result r;
  std::memset(&r, 0, sizeof r);
  r.value_probs[0] = r.value_probs[1] = 0.5f;
  r.score_probs[399] = r.score_probs[400] = 0.5f;   // E[score] = 0
  NodePool pool;
  TreeNode* node = pool.Create();
  EvaluateLeaf(r, node, (Color)color_to_move, (Color)root_color, root_score_est,
               ScoreUtilityParams{score_weight, integral ? ScoreUtilityMode::kIntegral : ScoreUtilityMode::kDirect});
  return node->init_util_est;
}

// LeafEvaluator::ScoreUtility (leaf_evaluator.cc:125-132)
float p3host_test_score_utility(int integral, float score_weight, float score_est, float score_stddev, float root_score_est) {
  return ScoreUtility(ScoreUtilityParams{score_weight, integral ? ScoreUtilityMode::kIntegral : ScoreUtilityMode::kDirect},
                      score_est, score_stddev, root_score_est);
}

}  // extern "C"
