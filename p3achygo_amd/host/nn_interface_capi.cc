// nn_interface_capi.cc — C entry points around NNInterface (nn_interface.h): a handle API so
// a thread-per-game caller can be driven from ctypes, and the stress tests restated from
// cc/nn/__tests__/nn_interface_sync_test.cc.
#include <chrono>
#include <random>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../csrc/slot_state.h"
#include "nn_interface.h"
#include "threaded_search.h"

using namespace p3;

namespace {

// ---- the reference test's CountingEngine (nn_interface_sync_test.cc:73-175) ---------------
// RunInference writes f(slot) into every element of every slot, yielding between slots;
// GetBatch reads its slot element by element, yielding in between, and checks
//   race:   all elements equal (no RunInference during the read),
//   stale:  the result's generation is newer than the generation at LoadBatch,
//   slot:   the value is f(slot).
constexpr int kSlotElems = 32;
constexpr int kPrime = (1 << 19) - 1;
inline int SlotFn(int t) { return (t + t) % kPrime; }

struct CountingEvaluator final : Evaluator {
  explicit CountingEvaluator(int n) : n_(n), buf_(n * kSlotElems), result_gen_(n), load_gen_(n) {
    for (auto& b : buf_) b.store(0);
    for (auto& g : result_gen_) g.store(0);
    for (auto& g : load_gen_) g.store(0);
  }
  void Load(int t, const p3hip_features&) override {
    load_gen_[t].store(gen_.load(std::memory_order_acquire), std::memory_order_release);
    ++loads_;
  }
  bool Run() override {
    const int gen = gen_.fetch_add(1, std::memory_order_relaxed) + 1;
    for (int t = 0; t < n_; ++t) {
      const int v = SlotFn(t);
      for (int i = 0; i < kSlotElems; ++i) buf_[t * kSlotElems + i].store(v, std::memory_order_relaxed);
      result_gen_[t].store(gen, std::memory_order_relaxed);
      std::this_thread::yield();
    }
    return true;
  }
  void Get(int t, p3hip_result& r) override {
    int vals[kSlotElems];
    for (int i = 0; i < kSlotElems; ++i) {
      vals[i] = buf_[t * kSlotElems + i].load(std::memory_order_relaxed);
      std::this_thread::yield();
    }
    for (int i = 1; i < kSlotElems; ++i)
      if (vals[i] != vals[0]) race_ = true;
    if (result_gen_[t].load(std::memory_order_relaxed) <= load_gen_[t].load(std::memory_order_acquire)) stale_ = true;
    if (vals[0] != SlotFn(t)) wrong_slot_ = true;
    std::memset(&r, 0, sizeof r);
    for (float& x : r.move_logits) x = (float)vals[0];
  }
  int failures() const { return (race_ ? 1 : 0) | (stale_ ? 2 : 0) | (wrong_slot_ ? 4 : 0); }

  const int n_;
  std::atomic<int> gen_{0};
  std::vector<std::atomic<int>> buf_, result_gen_, load_gen_;
  std::atomic<bool> race_{false}, stale_{false}, wrong_slot_{false};
  std::atomic<long> loads_{0};
};

// ---- an evaluator with the HIP engine's hand-over rules ----------------------------------------
// Like p3hip_run it evaluates only the dirty slots, compacted into dense rows (the engine's own
// SlotStates, csrc/slot_state.h), and its Run() takes a while with the interface lock held, so
// worker loads land while it is gathering and running — the interleaving of
// cc/nn/nn_interface.cc:351-361, in which a load is picked up by run N but only counted as loaded
// for run N+1.  Get() checks that the slot was evaluated by the last run (bit 16), from the
// features of its latest load (stale, bit 2), into the right row (wrong slot, bit 4).
struct CompactingEvaluator final : Evaluator {
  explicit CompactingEvaluator(int n, int run_us)
      : n_(n), run_us_(run_us), slots_(n), load_seq_(n), row_seq_(n), row_val_(n) {
    for (auto& a : load_seq_) a.store(0);
  }
  void Load(int t, const p3hip_features&) override {
    load_seq_[t].fetch_add(1, std::memory_order_relaxed);   // "writes the features"
    slots_.loaded(t);
  }
  bool Run() override {
    const int n = slots_.gather(false, [&](int s, int row) {
      if (slots_.state(s) == SlotStates::kEvaluated) ++reevaluated_;
      row_seq_[row] = load_seq_[s].load(std::memory_order_relaxed);   // "copies the features"
      row_val_[row] = -1;
      if ((s & 7) == 0) std::this_thread::yield();   // stretch the gather: loads land inside it
    });
    std::this_thread::sleep_for(std::chrono::microseconds(run_us_));   // the forward pass
    for (int s = 0; s < n_; ++s)
      if (slots_.row(s) >= 0) row_val_[slots_.row(s)] = SlotFn(s);
    rows_ += n;
    ++runs_;
    return true;
  }
  void Get(int t, p3hip_result& r) override {
    std::memset(&r, 0, sizeof r);
    const int row = slots_.row(t);
    if (row < 0) {
      unevaluated_ = true;
      return;
    }
    if (row_seq_[row] != load_seq_[t].load(std::memory_order_relaxed)) stale_ = true;
    if (row_val_[row] != SlotFn(t)) wrong_slot_ = true;
    for (float& x : r.move_logits) x = (float)row_val_[row];
    slots_.fetched(t);
  }
  int failures() const { return (stale_ ? 2 : 0) | (wrong_slot_ ? 4 : 0) | (unevaluated_ ? 16 : 0); }

  const int n_, run_us_;
  SlotStates slots_;
  std::vector<std::atomic<long>> load_seq_;
  std::vector<long> row_seq_;
  std::vector<int> row_val_;
  std::atomic<bool> stale_{false}, wrong_slot_{false}, unevaluated_{false};
  long rows_ = 0, runs_ = 0, reevaluated_ = 0;
};

// One worker of the reference's stress loop: jitter, occasional long sleeps (every 8th
// thread misses several batches), then a blocking evaluation whose value must be f(slot).
void StressWorker(int tid, NNInterface* black_nn, NNInterface* white_nn, std::atomic<bool>* stop,
                  std::atomic<bool>* error, std::atomic<long>* calls) {
  Game game;
  Probability prob((uint64_t)tid);
  std::mt19937 rng((uint32_t)tid * 2654435761u);
  std::uniform_int_distribution<int> jitter_us(100, 1000), slow_ms(5, 50);
  const bool slow = tid % 8 == 0;
  int ply = 0;
  while (!stop->load(std::memory_order_relaxed) && !error->load(std::memory_order_relaxed)) {
    std::this_thread::sleep_for(std::chrono::microseconds(jitter_us(rng)));
    if (slow) std::this_thread::sleep_for(std::chrono::milliseconds(slow_ms(rng)));
    NNInterface* nn = ply % 2 == 0 ? black_nn : white_nn;
    const p3hip_result r = nn->LoadAndGetInference(tid, game, kBlack, prob);
    if ((int)r.move_logits[0] != SlotFn(tid)) {
      error->store(true);
      return;
    }
    calls->fetch_add(1, std::memory_order_relaxed);
    ++ply;
  }
}

// ---- an evaluator with the HIP engine's cache rules (include/p3hip.h p3hip_cache_*) -----------------
// "Evaluates" a position by copying the stones of the (rotated) features into move_logits — so a result,
// un-rotated by the symmetry it was computed under, must read as the stones of the game itself — and keeps
// a table of (key -> result, symmetry): a keyed load whose key is in the table is served from it.
struct KeyedTableEvaluator final : Evaluator {
  explicit KeyedTableEvaluator(int n) : feats_(n), keys_(n), out_(n), out_sym_(n), out_hit_(n), loaded_(n, 0) {}
  bool EnableDeviceCache(int) override { return true; }
  void Load(int t, const p3hip_features& f) override { LoadKeyed(t, f, 0, 0, 0); }
  void LoadKeyed(int t, const p3hip_features& f, uint64_t lo, uint64_t hi, int sym) override {
    feats_[t] = f;
    keys_[t] = {lo, hi, sym};
    loaded_[t] = 1;
  }
  bool Run() override {
    for (size_t t = 0; t < feats_.size(); ++t) {
      if (!loaded_[t]) continue;
      loaded_[t] = 0;
      const K& k = keys_[t];
      const bool keyed = (k.lo | k.hi) != 0;
      ++lookups_;
      auto it = keyed ? table_.find({k.lo, k.hi}) : table_.end();
      if (it != table_.end()) {
        out_[t] = it->second.first;
        out_sym_[t] = it->second.second;
        out_hit_[t] = true;
        ++hits_;
        continue;
      }
      p3hip_result r;
      std::memset(&r, 0, sizeof r);
      for (int i = 0; i < kNumLocs; ++i) r.move_logits[i] = r.move_probs[i] = r.opt_move_probs[i] = (float)feats_[t].board[i];
      r.value_probs[1] = 1.0f;
      out_[t] = r;
      out_sym_[t] = k.sym;
      out_hit_[t] = false;
      ++evaluated_;
      if (keyed) table_[{k.lo, k.hi}] = {r, k.sym};
    }
    return true;
  }
  void Get(int t, p3hip_result& r) override { r = out_[t]; }
  void GetKeyed(int t, p3hip_result& r, int* sym, bool* hit) override {
    r = out_[t];
    if (sym) *sym = out_sym_[t];
    if (hit) *hit = out_hit_[t];
  }
  struct K { uint64_t lo, hi; int sym; };
  struct PH { size_t operator()(const std::pair<uint64_t, uint64_t>& p) const { return (size_t)(p.first ^ (p.second * 0x9e3779b97f4a7c15ull)); } };
  std::vector<p3hip_features> feats_;
  std::vector<K> keys_;
  std::vector<p3hip_result> out_;
  std::vector<int> out_sym_;
  std::vector<char> out_hit_, loaded_;
  std::unordered_map<std::pair<uint64_t, uint64_t>, std::pair<p3hip_result, int>, PH> table_;
  long lookups_ = 0, hits_ = 0, evaluated_ = 0;
};

}  // namespace

extern "C" {
// NNInterface over an engine-side cache (EnableDeviceCache): `rounds` passes over `positions` random-playout
// positions through LoadAndGetInference (a fresh random symmetry per call) and, every other round, through the
// async LoadEntry / FetchEntry pair.  Every result, un-rotated by the interface, must show the game's own
// stones.  out: {mismatching results, engine evaluations, engine hits, interface-counted hits, distinct keys}.
// host_cache_entries: the interface's own LRU in front of the table (0: every repeat reaches the engine's table;
// large: a thread's own repeats are answered at once, without a slot or a run, as in the reference).
int p3host_test_nn_device_cache(int positions, int rounds, uint64_t seed, int host_cache_entries, long out[5]) {
  auto* ev = new KeyedTableEvaluator(4);
  NNInterface nn(1, NNInterface::kTimeoutUs, (size_t)host_cache_entries, std::unique_ptr<Evaluator>(ev));
  if (!nn.EnableDeviceCache(10) || !nn.device_cache()) return 1;
  Probability prob(seed);
  std::vector<Game> games;
  std::vector<Color> to_move;
  for (int g = 0; g < positions; ++g) {
    Game game(7.5f, true);
    Color c = kBlack;
    const int plies = 5 + (int)(RandRange(prob.prng(), 0, 60));
    for (int m = 0; m < plies && !game.IsGameOver(); ++m) {
      Loc mv = kPassLoc;
      for (int tries = 0; tries < 40; ++tries) {
        const int idx = RandRange(prob.prng(), 0, kNumLocs);
        if (game.IsValidMove(AsLoc(idx), c)) { mv = AsLoc(idx); break; }
      }
      game.PlayMove(mv, c);
      c = Opp(c);
    }
    games.push_back(game);
    to_move.push_back(c);
  }
  long bad = 0;
  auto check = [&](const Game& game, const p3hip_result& r) {
    for (int i = 0; i < kNumLocs; ++i)
      if (r.move_logits[i] != (float)game.board().at(i) || r.move_probs[i] != r.move_logits[i] || r.opt_move_probs[i] != r.move_logits[i]) { ++bad; return; }
  };
  for (int round = 0; round < rounds; ++round)
    for (int g = 0; g < positions; ++g) {
      if (round % 2 == 0) {
        check(games[g], nn.LoadAndGetInference(0, games[g], to_move[g], prob));
      } else {
        nn.LoadEntry(0, 0, games[g], to_move[g], prob);
        nn.SignalReadyForInference();
        check(games[g], nn.FetchEntry(0, 0, games[g], to_move[g]));
      }
    }
  out[0] = bad; out[1] = ev->evaluated_; out[2] = ev->hits_; out[3] = nn.device_cache_hits(); out[4] = (long)ev->table_.size();
  return 0;
}
}

extern "C" {

// RunSyncTest / RunDualInterfaceTest (nn_interface_sync_test.cc:178-355).  strategy: 0
// kMutex, 1 kGenCounter.  Returns a failure mask: 1 race, 2 stale, 4 wrong slot, 8 worker
// saw a wrong value (second interface's bits shifted by 4); *calls_out = evaluations done.
int p3host_test_nn_sync(int strategy, int num_threads, int millis, int cache_size, int dual, long* calls_out) {
  auto* ea = new CountingEvaluator(num_threads);
  auto* eb = dual ? new CountingEvaluator(num_threads) : nullptr;
  std::atomic<bool> stop{false}, error{false};
  std::atomic<long> calls{0};
  int mask = 0;
  {
    NNInterface nn_a(num_threads, 200, (size_t)cache_size, std::unique_ptr<Evaluator>(ea),
                     (NNInterface::WakeStrategy)strategy);
    std::unique_ptr<NNInterface> nn_b;
    if (dual)
      nn_b.reset(new NNInterface(num_threads, 200, (size_t)cache_size, std::unique_ptr<Evaluator>(eb),
                                 (NNInterface::WakeStrategy)strategy));
    std::vector<std::thread> workers;
    for (int t = 0; t < num_threads; ++t) {
      NNInterface* black = &nn_a;
      NNInterface* white = &nn_a;
      if (dual) {   // cur plays black in even games, white in odd ones
        black = t % 2 == 0 ? &nn_a : nn_b.get();
        white = t % 2 == 0 ? nn_b.get() : &nn_a;
      }
      workers.emplace_back(StressWorker, t, black, white, &stop, &error, &calls);
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(millis));
    stop.store(true);
    for (auto& w : workers) w.join();
    mask = ea->failures() | (error.load() ? 8 : 0);
    if (dual) mask |= eb->failures() << 4;
  }
  if (calls_out) *calls_out = calls.load();
  return mask;
}

// The same stress loop over the compacting evaluator (the engine's slot rules).  Failure mask:
// 2 stale, 4 wrong slot, 8 worker saw a wrong value, 16 a result was requested for a slot the
// last run had not evaluated.  *rows_out / *runs_out = rows evaluated / runs.
int p3host_test_nn_compacting(int strategy, int num_threads, int millis, int timeout_us, int run_us, long* calls_out,
                              long* rows_out, long* runs_out) {
  auto* ev = new CompactingEvaluator(num_threads, run_us);
  std::atomic<bool> stop{false}, error{false};
  std::atomic<long> calls{0};
  int mask = 0;
  {
    NNInterface nn(num_threads, timeout_us, 0, std::unique_ptr<Evaluator>(ev), (NNInterface::WakeStrategy)strategy);
    std::vector<std::thread> workers;
    for (int t = 0; t < num_threads; ++t) workers.emplace_back(StressWorker, t, &nn, &nn, &stop, &error, &calls);
    std::this_thread::sleep_for(std::chrono::milliseconds(millis));
    stop.store(true);
    for (auto& w : workers) w.join();
    mask = ev->failures() | (error.load() ? 8 : 0);
    if (rows_out) *rows_out = ev->rows_;
    if (runs_out) *runs_out = ev->runs_;
  }
  if (calls_out) *calls_out = calls.load();
  return mask;
}

// Deterministic replay of the hand-over sequences on the engine's SlotStates.  Returns 0, or the
// number of the first step that went wrong.
int p3host_test_slot_states() {
  SlotStates st(8);
  std::vector<int> rows;
  auto run = [&](int load_during = -1, int when = -1) {
    rows.clear();
    return st.gather(false, [&](int s, int) {
      rows.push_back(s);
      if (s == when) st.loaded(load_during);   // a LoadBatch landing while the run gathers
    });
  };
  st.loaded(2);
  st.loaded(5);
  if (run() != 2 || rows != std::vector<int>{2, 5} || st.row(2) != 0 || st.row(5) != 1 || st.row(3) != -1) return 1;
  st.fetched(2);
  // 5 was evaluated but its caller has not been handed the result yet (NNInterface counted its
  // load for the next run): the next run must evaluate it again; 2 was fetched and is out
  if (run() != 1 || rows != std::vector<int>{5} || st.row(5) != 0 || st.row(2) != -1) return 2;
  st.fetched(5);
  if (run() != 0 || st.row(5) != -1) return 3;
  // a load that lands BEFORE the gather reaches its slot is evaluated by this run and stays
  // dirty for the next one (the sequence of nn_interface.cc:351-361)
  st.loaded(1);
  if (run(6, 1) != 2 || rows != std::vector<int>{1, 6}) return 4;
  st.fetched(1);
  if (run() != 1 || rows != std::vector<int>{6}) return 5;
  st.fetched(6);
  // a load that lands AFTER the gather passed its slot waits for the next run
  st.loaded(4);
  if (run(0, 4) != 1 || rows != std::vector<int>{4} || st.row(0) != -1) return 6;
  st.fetched(4);
  if (run() != 1 || rows != std::vector<int>{0}) return 7;
  // a reload of an evaluated-but-unfetched slot is a fresh load
  st.loaded(0);
  if (run() != 1 || rows != std::vector<int>{0}) return 8;
  st.fetched(0);
  // RUN_ALL_SLOTS evaluates everything and leaves the states alone
  if (st.gather(true, [](int, int) {}) != 8 || st.row(7) != 7) return 9;
  if (run() != 0) return 10;
  return 0;
}

// The async path parallel search uses (nn_interface.h:176-198, search.cc's SearchTask):
// `tasks` search tasks share one interface in kExplicit mode; each owns `workers` slots,
// loads them all, signals once, then fetches them all.  Returns the same failure mask.
int p3host_test_nn_async(int strategy, int tasks, int workers, int rounds, long* inferences_out) {
  const int n = tasks * workers;
  auto* eng = new CountingEvaluator(n);
  std::atomic<bool> error{false};
  int mask = 0;
  {
    NNInterface nn(n, NNInterface::kTimeoutUs, 0, std::unique_ptr<Evaluator>(eng), NNInterface::SignalKind::kExplicit,
                   tasks, (NNInterface::WakeStrategy)strategy);
    std::vector<std::thread> threads;
    for (int t = 0; t < tasks; ++t)
      threads.emplace_back([&, t] {
        NNInterface::Slot slot = nn.MakeSlot(t * workers);
        Game game;
        Probability prob((uint64_t)t);
        std::mt19937 rng(t + 1);
        for (int r = 0; r < rounds && !error.load(); ++r) {
          if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 700));
          for (int w = 0; w < workers; ++w) slot.LoadEntry(w, game, kBlack, prob);
          slot.SignalReadyForInference();
          for (int w = 0; w < workers; ++w) {
            const p3hip_result res = slot.FetchEntry(w, game, kBlack);
            if ((int)res.move_logits[0] != SlotFn(t * workers + w)) error.store(true);
          }
        }
        slot.UnregisterSearchTask();
      });
    for (auto& th : threads) th.join();
    mask = eng->failures() | (error.load() ? 8 : 0);
    if (inferences_out) *inferences_out = nn.num_inferences();
  }
  return mask;
}

// ---- the threaded concurrent search (threaded_search.h) over a kExplicit interface ------------
// Plays `num_moves` moves from the empty board, each chosen by ThreadedSearch::Run with
// `num_threads` worker threads and the given policies over the uniform NullEvaluator (graph != 0:
// McgsNodeTable semantics; time_ms > 0: time control instead of the visit budget), re-using the
// tree between moves, and checks after every search the invariants of search_test.cc:130-222:
//   1  a node is still marked in flight,           2  n != 1 + sum of child visits on an inner node,
//   4  the visit count is outside [budget, budget + threads),   8  the move is illegal,
//   16 a child edge has visits but its node was never visited.
// stats: [0] visits, [1] aborted, [2] collisions, [3] nodes alive at the end, [4] searches.
int p3host_test_threaded_search(int num_threads, int visit_budget, int q_fn, int n_fn, int collision, int detector,
                                int descent, int graph, int time_ms, int num_moves, uint64_t seed, long* stats) {
  int mask = 0;
  for (int i = 0; i < 5; ++i) stats[i] = 0;
  NNInterface nn(num_threads, NNInterface::kTimeoutUs, 1 << 12, std::unique_ptr<Evaluator>(new NullEvaluator()),
                 NNInterface::SignalKind::kExplicit, /*num_shared_search_tasks=*/1);
  BiasCache bias(0.8f, 0.3f);
  ThreadedSearch search(nn.MakeSlot(0), &bias);
  NodePool pool(graph != 0);
  Game game(7.5f, true);
  Probability prob(seed);
  TreeNode* root = pool.Create();
  Color c = kBlack;
  ThreadedSearch::Params p;
  p.num_threads = num_threads;
  p.total_visit_budget = time_ms > 0 ? (1 << 30) : visit_budget;
  p.total_visit_time_ms = time_ms;
  p.fns = VirtualFns{(QFn)q_fn, (NFn)n_fn, -1.5f};
  p.collision = (CollisionPolicy)collision;
  p.detector = (CollisionDetector)detector;
  p.descent = (DescentPolicy)descent;
  for (int m = 0; m < num_moves && !game.IsGameOver(); ++m) {
    const int n_before = root->n;
    const ThreadedSearch::Result r = search.Run(prob, game, &pool, root, c, p);
    ++stats[4];
    stats[0] += r.num_visits; stats[1] += r.num_aborted; stats[2] += r.num_collisions;
    if (time_ms <= 0 && (r.num_visits < visit_budget || r.num_visits >= visit_budget + num_threads)) mask |= 4;
    if (time_ms > 0 && (r.num_visits < 1 || r.time_ms > 20 * time_ms + 2000)) mask |= 4;
    if (!game.IsValidMove(r.move, c)) mask |= 8;
    // walk the tree from the root
    std::vector<TreeNode*> work{root};
    std::unordered_set<TreeNode*> seen;
    while (!work.empty()) {
      TreeNode* nd = work.back();
      work.pop_back();
      if (!seen.insert(nd).second) continue;
      if (nd->n_in_flight.load() != 0) mask |= 1;
      int cv = 0;
      for (const ChildEdge& e : nd->children) {
        cv += e.visits;
        if (e.visits > 0 && e.node->n == 0) mask |= 16;
        work.push_back(e.node);
      }
      if (nd->evaluated && !nd->is_terminal && nd->n > 0 && nd->n != 1 + cv) mask |= 2;
    }
    (void)n_before;
    game.PlayMove(r.move, c);
    c = Opp(c);
    TreeNode* next = root->child(MoveIdx(r.move));
    if (!next) next = pool.Create();
    pool.Reap(next);
    root = next;
  }
  stats[3] = (long)pool.Size();
  nn.MakeSlot(0).UnregisterSearchTask();
  return mask;
}

// ---- handle API ----------------------------------------------------------------------------
// engine_kind: 0 the uniform NullEvaluator, 1 the HIP engine (lib_path, weights, device).
void* p3host_nn_new(int engine_kind, const char* lib_path, const char* weights, int device, int num_threads,
                    long timeout_us, long cache_size, int strategy, char* err, int err_len) {
  std::unique_ptr<Evaluator> ev;
  if (engine_kind == 0) {
    ev.reset(new NullEvaluator());
  } else {
    auto* h = new HipEvaluator();
    ev.reset(h);
    if (!h->Open(lib_path, weights, num_threads, device)) {
      if (err && err_len > 0) snprintf(err, err_len, "%s", h->err.c_str());
      return nullptr;
    }
  }
  return new NNInterface(num_threads, timeout_us, (size_t)cache_size, std::move(ev), (NNInterface::WakeStrategy)strategy);
}
void p3host_nn_free(void* nn) { delete (NNInterface*)nn; }
void p3host_nn_set_num_cache_last_moves(void* nn, int n) { ((NNInterface*)nn)->SetNumCacheLastMoves(n); }
long p3host_nn_num_inferences(void* nn) { return ((NNInterface*)nn)->num_inferences(); }
void p3host_nn_register_thread(void* nn, int tid) { ((NNInterface*)nn)->RegisterThread(tid); }
void p3host_nn_unregister_thread(void* nn, int tid) { ((NNInterface*)nn)->UnregisterThread(tid); }
// game: a p3host_game_new handle; prob: a p3host_prob_new handle
void p3host_nn_load_and_get_inference(void* nn, int tid, void* game, int color, void* prob, p3hip_result* out) {
  *out = ((NNInterface*)nn)->LoadAndGetInference(tid, *(Game*)game, (Color)color, *(Probability*)prob);
}
void p3host_nn_load_and_get_ownership(void* nn, int tid, void* game, int color, float* out) {
  const auto own = ((NNInterface*)nn)->LoadAndGetOwnership(tid, *(Game*)game, (Color)color);
  std::memcpy(out, own.data(), sizeof(float) * kNumLocs);
}

// Thread-per-game driver over the handle: thread t plays `moves_per_thread` uniformly random
// legal moves from the empty board (seed t), evaluating every position through
// LoadAndGetInference; out[t * stride ...] receives the results of thread t in order.
// Used to compare the threaded interface with a one-thread interface on the same draws.
// sequential != 0 runs the same thread bodies one after another (every batch then holds one
// position and is released by the timeout).
void p3host_nn_play_threads_ex(void* nn_v, int num_threads, int moves_per_thread, uint64_t seed_base, int sequential,
                               p3hip_result* out) {
  NNInterface* nn = (NNInterface*)nn_v;
  std::vector<std::thread> threads;
  for (int t = 0; t < num_threads; ++t) {
    threads.emplace_back([=] {
      Game game(7.5f, true);
      Probability prob(seed_base + (uint64_t)t);
      Color c = kBlack;
      for (int m = 0; m < moves_per_thread; ++m) {
        out[(size_t)t * moves_per_thread + m] = nn->LoadAndGetInference(t, game, c, prob);
        Loc mv = kPassLoc;
        for (int tries = 0; tries < 64; ++tries) {
          const int idx = RandRange(prob.prng(), 0, kNumLocs);
          const Loc l{idx / kBoardLen, idx % kBoardLen};
          if (game.IsValidMove(l, c)) { mv = l; break; }
        }
        game.PlayMove(mv, c);
        c = Opp(c);
      }
      nn->UnregisterThread(t);   // a finished game thread leaves the batch (self_play_thread.cc)
    });
    if (sequential) threads.back().join();
  }
  if (!sequential)
    for (auto& th : threads) th.join();
}
void p3host_nn_play_threads(void* nn_v, int num_threads, int moves_per_thread, uint64_t seed_base, p3hip_result* out) {
  p3host_nn_play_threads_ex(nn_v, num_threads, moves_per_thread, seed_base, 0, out);
}

}  // extern "C"
