// nn_interface.h — the reference's thread-per-game bridge between games and the engine,
// kept API-compatible for callers written against it (gtp, experiments, the eval workers):
// nn::NNInterface, cc/nn/nn_interface.h:85-202 + nn_interface.cc:36-404.
//
// Same public surface: the five constructors, SignalKind, WakeStrategy, Slot/MakeSlot,
// LoadAndGetInference, LoadAndGetOwnership, Register/UnregisterThread, the async trio
// LoadEntry / FetchEntry / SignalReadyForInference, UnregisterSearchTask,
// SetNumCacheLastMoves; kTimeoutUs = 400, default cache 2^20 entries split per thread.
// Same synchronisation contract: LoadBatch/GetBatch of different slots run
// concurrently without a lock; RunInference runs with the lock held, never while a result is
// unread, and only hands results to slots that were loaded before it started.
//
// Built on std::mutex / condition variables / a futex rather than absl::Mutex: conditions
// are re-evaluated on explicit notifications instead of on every unlock.  The self-play
// driver (selfplay.cc) does not go through this class — it schedules resumable games
// directly — but both feed the engine through the same Evaluator boundary.
#pragma once
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <memory>
#include <mutex>
#include <optional>
#include <thread>
#include <unordered_map>
#include <vector>

#include "evaluator.h"
#include "features.h"

namespace p3 {

constexpr size_t kDefaultNNCacheSize = size_t(1) << 20;   // constants.h:81

// core::LRUCache (cc/core/lru_cache.h:17-64): most-recently-used at the back; an insert
// beyond capacity evicts the front.  Capacity 0 keeps nothing.
template <class K, class V, class H>
class LruCache {
 public:
  explicit LruCache(size_t cap = 0) : cap_(cap) {}
  bool Contains(const K& k) const { return map_.find(k) != map_.end(); }
  std::optional<V> Get(const K& k) {
    auto it = map_.find(k);
    if (it == map_.end()) return std::nullopt;
    order_.splice(order_.end(), order_, it->second.second);
    return it->second.first;
  }
  void Insert(const K& k, const V& v) {
    auto it = map_.find(k);
    if (it != map_.end()) {
      it->second.first = v;
      order_.splice(order_.end(), order_, it->second.second);
      return;
    }
    order_.push_back(k);
    map_.emplace(k, std::make_pair(v, std::prev(order_.end())));
    if (map_.size() > cap_) {
      map_.erase(order_.front());
      order_.pop_front();
    }
  }
  size_t size() const { return map_.size(); }

 private:
  size_t cap_;
  std::list<K> order_;
  std::unordered_map<K, std::pair<V, typename std::list<K>::iterator>, H> map_;
};

class NNInterface final {
 public:
  enum class SignalKind : uint8_t { kAuto = 0, kExplicit = 1 };
  // kMutex: waiters sleep on a condition variable tied to the interface lock.
  // kGenCounter: waiters sleep on a generation counter; one FUTEX_WAKE releases them all.
  enum class WakeStrategy : uint8_t { kMutex = 0, kGenCounter = 1 };
  static constexpr int64_t kTimeoutUs = 400;   // nn_interface.h:205

  // View of a contiguous run of slots (nn_interface.h:107-145).
  struct Slot {
    Slot(NNInterface* nn, int task_offset) : nn_(nn), off_(task_offset) {}
    SignalKind signal_kind() const { return nn_->signal_kind(); }
    void LoadEntry(int thread_id, const Game& game, Color c, Probability& prob) {
      nn_->LoadEntry(thread_id, off_, game, c, prob);
    }
    p3hip_result FetchEntry(int thread_id, const Game& game, Color c) { return nn_->FetchEntry(thread_id, off_, game, c); }
    p3hip_result LoadAndGetInference(int thread_id, const Game& game, Color c, Probability& prob) {
      return nn_->LoadAndGetInference(thread_id + off_, game, c, prob);
    }
    void SignalReadyForInference() { nn_->SignalReadyForInference(); }
    void UnregisterSearchTask() { nn_->UnregisterSearchTask(); }

   private:
    NNInterface* nn_;
    int off_;
  };
  Slot MakeSlot(int task_offset) { return Slot(this, task_offset); }

  NNInterface(int num_threads, std::unique_ptr<Evaluator> engine)
      : NNInterface(num_threads, kTimeoutUs, kDefaultNNCacheSize, std::move(engine), SignalKind::kAuto, -1) {}
  NNInterface(int num_threads, int64_t timeout, size_t cache_size, std::unique_ptr<Evaluator> engine)
      : NNInterface(num_threads, timeout, cache_size, std::move(engine), SignalKind::kAuto, -1) {}
  NNInterface(int num_threads, int64_t timeout, size_t cache_size, std::unique_ptr<Evaluator> engine,
              WakeStrategy wake)
      : NNInterface(num_threads, timeout, cache_size, std::move(engine), SignalKind::kAuto, -1, wake) {}
  NNInterface(int num_threads, std::unique_ptr<Evaluator> engine, SignalKind kind, int num_shared_search_tasks)
      : NNInterface(num_threads, kTimeoutUs, kDefaultNNCacheSize, std::move(engine), kind, num_shared_search_tasks) {}
  NNInterface(int num_threads, int64_t timeout, size_t cache_size, std::unique_ptr<Evaluator> engine,
              SignalKind kind, int num_shared_search_tasks, WakeStrategy wake = WakeStrategy::kGenCounter)
      : num_registered_(num_threads), num_threads_(num_threads), info_(num_threads), timeout_us_(timeout),
        engine_(std::move(engine)), syms_(num_threads, kIdentity), signal_kind_(kind),
        num_shared_tasks_(num_shared_search_tasks), wake_(wake) {
    caches_.reserve(num_threads);   // nn_interface.cc:237-242
    for (int t = 0; t < num_threads; ++t) caches_.emplace_back(cache_size / (size_t)num_threads);
    if (num_threads_ > 1) infer_thread_ = std::thread([this] { InferLoop(); });
  }
  ~NNInterface() {
    {
      std::lock_guard<std::mutex> l(mu_);
      running_ = false;
    }
    infer_cv_.notify_all();
    if (infer_thread_.joinable()) infer_thread_.join();
  }
  NNInterface(const NNInterface&) = delete;
  NNInterface& operator=(const NNInterface&) = delete;

  SignalKind signal_kind() const { return signal_kind_; }
  void SetNumCacheLastMoves(int n) { num_cache_last_moves_ = n; }   // [0, 5], default 5
  Evaluator* engine() { return engine_.get(); }

  // Adds the engine's HBM table (p3hip_cache_*) behind this object's per-thread LRUs: one table for every worker
  // and game, looked up and filled by the run itself.  A position a thread has seen itself comes from its own LRU
  // at once, as in the reference (nn_interface.cc:112-118: no slot, no run); one that only another thread or game
  // has seen takes part in a run and costs a table probe instead of a forward pass.  Same contract towards the
  // search either way: the result stored at the first evaluation, un-rotated.  False when the engine has no table.
  bool EnableDeviceCache(int log2_entries) {
    device_cache_ = engine_->EnableDeviceCache(log2_entries);
    return device_cache_;
  }
  bool device_cache() const { return device_cache_; }
  long device_cache_hits() const { return device_hits_.load(std::memory_order_relaxed); }
  long device_cache_lookups() const { return device_lookups_.load(std::memory_order_relaxed); }

  // Blocks until the result is ready (nn_interface.cc:108-133).
  p3hip_result LoadAndGetInference(int thread_id, const Game& game, Color color_to_move, Probability& prob) {
    const Key key = MakeKey(game, color_to_move);
    if (caches_[thread_id].Contains(key)) {
      MarkCached(thread_id, true);
      return *caches_[thread_id].Get(key);
    }
    const Symmetry sym = RandomSymmetry(prob.prng());
    if (device_cache_) LoadBatchKeyed(thread_id, game, color_to_move, sym);
    else LoadBatch(thread_id, game, color_to_move, sym);
    SignalLoadedAndBlockUntilReady(thread_id);
    p3hip_result r = device_cache_ ? GetBatchKeyed(thread_id, sym) : GetBatch(thread_id, sym);
    caches_[thread_id].Insert(key, r);
    return r;
  }

  // nn_interface.cc:135-145
  std::array<float, kNumLocs> LoadAndGetOwnership(int thread_id, const Game& game, Color color_to_move) {
    LoadBatch(thread_id, game, color_to_move, kIdentity);
    SignalLoadedAndBlockUntilReady(thread_id);
    std::array<float, kNumLocs> own;
    engine_->GetOwnership(thread_id, own.data());
    // the result of this slot is consumed: let the next inference run
    info_[thread_id].res_ready.store(false, std::memory_order_release);
    NotifyInfer();
    return own;
  }

  void RegisterThread(int thread_id) {   // nn_interface.cc:147-158
    {
      std::lock_guard<std::mutex> l(mu_);
      ThreadInfo& t = info_[thread_id];
      if (t.registered) return;
      t.registered = true;
      t.loaded = false;
      ++num_registered_;
    }
    infer_cv_.notify_all();
  }
  void UnregisterThread(int thread_id) {   // nn_interface.cc:160-170
    {
      std::lock_guard<std::mutex> l(mu_);
      ThreadInfo& t = info_[thread_id];
      if (!t.registered) return;
      --num_registered_;
      t.registered = false;
    }
    infer_cv_.notify_all();
  }

  // async API (nn_interface.cc:172-230)
  void LoadEntry(int thread_id, int offset, const Game& game, Color color_to_move, Probability& prob) {
    const int tid = thread_id + offset;
    const Key key = MakeKey(game, color_to_move);
    if (caches_[tid].Contains(key)) {
      MarkCached(tid, true);
      return;
    }
    const Symmetry sym = RandomSymmetry(prob.prng());
    syms_[tid] = sym;
    if (device_cache_) LoadBatchKeyed(tid, game, color_to_move, sym);
    else LoadBatch(tid, game, color_to_move, sym);
    {
      std::lock_guard<std::mutex> l(mu_);
      ThreadInfo& t = info_[tid];
      t.loaded = true;
      t.res_ready.store(false, std::memory_order_relaxed);
      t.res_cached = false;
    }
    infer_cv_.notify_all();
  }
  p3hip_result FetchEntry(int thread_id, int offset, const Game& game, Color color_to_move) {
    const int tid = thread_id + offset;
    const Key key = MakeKey(game, color_to_move);
    if (caches_[tid].Contains(key)) {
      MarkCached(tid, false);
      return *caches_[tid].Get(key);
    }
    Wait(tid);
    p3hip_result r = device_cache_ ? GetBatchKeyed(tid, syms_[tid]) : GetBatch(tid, syms_[tid]);
    caches_[tid].Insert(key, r);
    return r;
  }
  void SignalReadyForInference() {   // nn_interface.h:184-192
    {
      std::lock_guard<std::mutex> l(mu_);
      ++num_signaled_tasks_;
    }
    if (num_threads_ == 1) Infer();   // no infer thread: run synchronously
    else infer_cv_.notify_all();
  }
  void UnregisterSearchTask() {
    {
      std::lock_guard<std::mutex> l(mu_);
      ++num_exited_tasks_;
    }
    infer_cv_.notify_all();
  }

  long num_inferences() const { return num_inferences_.load(std::memory_order_relaxed); }

 private:
  struct Key {   // NNKey, nn_interface.h:206-228
    Color color;
    uint64_t board_hash;
    std::array<Loc, 5> last_moves;   // oldest..newest; unused leading entries are noop
    float komi;
    bool operator==(const Key& o) const {
      if (color != o.color || board_hash != o.board_hash || komi != o.komi) return false;
      for (int i = 0; i < 5; ++i)
        if (last_moves[i] != o.last_moves[i]) return false;
      return true;
    }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const {
      uint64_t h = k.board_hash ^ (uint64_t(uint8_t(k.color)) * 0x9e3779b97f4a7c15ull);
      for (const Loc& l : k.last_moves) h = (h ^ uint64_t(uint32_t(l.i * 32 + l.j + 64))) * 0xff51afd7ed558ccdull;
      uint32_t kb;
      std::memcpy(&kb, &k.komi, 4);
      h = (h ^ kb) * 0xc4ceb9fe1a85ec53ull;
      return size_t(h ^ (h >> 29));
    }
  };
  struct ThreadInfo {   // nn_interface.h:230-242
    bool registered = true;
    bool loaded = false;                 // loaded_for_inference
    std::atomic<bool> res_ready{false};
    bool res_cached = false;
  };

  Key MakeKey(const Game& game, Color color_to_move) const {   // nn_interface.cc:92-106
    const int n = game.num_moves();
    Key k{color_to_move, game.board().hash(), {}, game.komi()};
    k.last_moves.fill(kNoopLoc);
    for (int i = 0; i < num_cache_last_moves_; ++i) {
      const int off = n - num_cache_last_moves_ + i;
      if (off >= 0) k.last_moves[5 - num_cache_last_moves_ + i] = game.move(off).loc;
    }
    return k;
  }

  void MarkCached(int tid, bool v) {
    {
      std::lock_guard<std::mutex> l(mu_);
      info_[tid].res_cached = v;
    }
    if (v) infer_cv_.notify_all();
  }

  void LoadBatch(int tid, const Game& game, Color color_to_move, Symmetry sym) {   // nn_interface.cc:245-277
    p3hip_features f;
    FillFeatures(game, color_to_move, sym, &f);
    engine_->Load(tid, f);
  }
  // 128-bit digest of the NNKey for the engine's table (two independent mixes of the same fields)
  static void KeyDigest(const Key& k, uint64_t* lo, uint64_t* hi) {
    uint32_t kb;
    std::memcpy(&kb, &k.komi, 4);
    uint64_t a = k.board_hash ^ (uint64_t(uint8_t(k.color)) * 0x9e3779b97f4a7c15ull);
    uint64_t b = (k.board_hash * 0xd6e8feb86659fd93ull) ^ (uint64_t(kb) << 8) ^ uint64_t(uint8_t(k.color));
    for (const Loc& l : k.last_moves) {
      const uint64_t m = uint64_t(uint32_t(l.i * 32 + l.j + 64));
      a = (a ^ m) * 0xff51afd7ed558ccdull;
      b = ((b << 7) | (b >> 57)) ^ (m * 0xc2b2ae3d27d4eb4full);
    }
    a = (a ^ kb) * 0xc4ceb9fe1a85ec53ull;
    b = (b ^ (b >> 31)) * 0x94d049bb133111ebull;
    *lo = a ^ (a >> 29);
    *hi = b ^ (b >> 32);
    if ((*lo | *hi) == 0) *lo = 1;   // 0/0 means "no key" to the engine
  }
  void LoadBatchKeyed(int tid, const Game& game, Color color_to_move, Symmetry sym) {
    p3hip_features f;
    FillFeatures(game, color_to_move, sym, &f);
    uint64_t lo, hi;
    KeyDigest(MakeKey(game, color_to_move), &lo, &hi);
    engine_->LoadKeyed(tid, f, lo, hi, (int)sym);
  }
  p3hip_result GetBatchKeyed(int tid, Symmetry loaded_sym) {
    p3hip_result r;
    int sym = (int)loaded_sym;
    bool hit = false;
    engine_->GetKeyed(tid, r, &sym, &hit);
    device_lookups_.fetch_add(1, std::memory_order_relaxed);
    if (hit) device_hits_.fetch_add(1, std::memory_order_relaxed);
    info_[tid].res_ready.store(false, std::memory_order_release);
    NotifyInfer();
    UnapplySymmetry((Symmetry)sym, &r);
    return r;
  }
  p3hip_result GetBatch(int tid, Symmetry sym) {   // nn_interface.h:254-292
    p3hip_result r;
    engine_->Get(tid, r);
    // cleared without the lock: Infer() reads it with acquire before RunInference()
    info_[tid].res_ready.store(false, std::memory_order_release);
    NotifyInfer();
    UnapplySymmetry(sym, &r);
    return r;
  }

  void SignalLoadedAndBlockUntilReady(int tid) {   // nn_interface.h:295-312
    if (num_threads_ == 1) {
      RunEngine();
      return;
    }
    {
      std::lock_guard<std::mutex> l(mu_);
      ThreadInfo& t = info_[tid];
      t.loaded = true;
      t.res_ready.store(false, std::memory_order_relaxed);
      t.res_cached = false;
    }
    infer_cv_.notify_all();
    Wait(tid);
  }

  // ---- wake-up of workers ------------------------------------------------------------------
  void Wait(int tid) {
    std::atomic<bool>& ready = info_[tid].res_ready;
    if (wake_ == WakeStrategy::kMutex) {
      std::unique_lock<std::mutex> l(mu_);
      done_cv_.wait(l, [&] { return ready.load(std::memory_order_acquire); });
      return;
    }
    uint32_t gen = gen_.load(std::memory_order_acquire);
    while (!ready.load(std::memory_order_acquire)) {
      syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen_), FUTEX_WAIT_PRIVATE, gen, nullptr, nullptr, 0);
      gen = gen_.load(std::memory_order_acquire);
    }
  }
  void NotifyAllWorkers() {   // called with mu_ held
    if (wake_ == WakeStrategy::kMutex) {
      done_cv_.notify_all();
      return;
    }
    gen_.fetch_add(1, std::memory_order_release);
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen_), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
  }
  // A consumed result can unblock the infer thread; it re-checks under the lock.
  void NotifyInfer() {
    if (num_threads_ > 1) infer_cv_.notify_all();
  }

  // A failed RunInference is fatal, as in the reference (CUDA_OK / CHECK abort the process,
  // trt_engine.cc:27-35): no worker may be handed a result of a run that did not happen.
  void RunEngine() {
    if (!engine_->Run()) {
      std::fprintf(stderr, "NNInterface: engine RunInference failed\n");
      std::abort();
    }
    num_inferences_.fetch_add(1, std::memory_order_relaxed);
  }

  // ---- inference loop (nn_interface.cc:279-404) --------------------------------------------
  void InferLoop() {
    while (running_.load(std::memory_order_acquire)) Infer();
  }
  bool ShouldInfer() const {   // nn_interface.cc:379-404; mu_ held
    if (!running_.load(std::memory_order_acquire)) return true;
    if (signal_kind_ == SignalKind::kExplicit) {
      const int remaining = num_shared_tasks_ - num_exited_tasks_;
      if (remaining <= 0) return false;
      return num_signaled_tasks_ == remaining;
    }
    bool pending = false;
    for (const ThreadInfo& t : info_) {
      if (!t.registered) continue;
      if (!t.res_cached && !t.loaded) return false;
      if (!t.res_cached) pending = true;
    }
    return pending;
  }
  void Infer() {
    std::unique_lock<std::mutex> l(mu_);
    // Always bounded by the timeout when one is set (both signal kinds): inference then
    // runs on whatever is loaded (nn_interface.cc:293-318).
    if (timeout_us_ > 0) infer_cv_.wait_for(l, std::chrono::microseconds(timeout_us_), [this] { return ShouldInfer(); });
    else infer_cv_.wait(l, [this] { return ShouldInfer(); });
    struct Reset {
      int& n;
      ~Reset() { n = 0; }
    } reset{num_signaled_tasks_};
    if (num_registered_ == 0) return;
    // an unread result would be overwritten by RunInference
    for (const ThreadInfo& t : info_)
      if (t.res_ready.load(std::memory_order_acquire)) return;
    bool any_loaded = false;
    for (const ThreadInfo& t : info_) any_loaded |= t.loaded;
    if (!any_loaded) return;

    RunEngine();
    for (ThreadInfo& t : info_) {
      if (t.registered && t.loaded) {   // slots loaded after Run() started wait for the next cycle
        t.res_ready.store(true, std::memory_order_release);
        t.loaded = false;
      }
      t.res_cached = false;
    }
    NotifyAllWorkers();
  }

  std::mutex mu_;
  std::condition_variable infer_cv_, done_cv_;
  int num_registered_;
  const int num_threads_;
  std::vector<ThreadInfo> info_;
  std::atomic<bool> running_{true};
  const int64_t timeout_us_;
  std::unique_ptr<Evaluator> engine_;
  std::vector<Symmetry> syms_;
  std::vector<LruCache<Key, p3hip_result, KeyHash>> caches_;
  std::thread infer_thread_;
  int num_cache_last_moves_ = 5;
  bool device_cache_ = false;
  std::atomic<long> device_hits_{0}, device_lookups_{0};
  const SignalKind signal_kind_;
  const int num_shared_tasks_;
  int num_signaled_tasks_ = 0, num_exited_tasks_ = 0;
  const WakeStrategy wake_;
  std::atomic<uint32_t> gen_{0};
  std::atomic<long> num_inferences_{0};
};

}  // namespace p3
