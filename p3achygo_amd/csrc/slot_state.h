// slot_state.h — which slots of the static batch the next run evaluates, and which dense row
// of the last run holds each slot's result.  Shared by the engine (engine.cpp) and by the
// host's compacting test evaluator (host/nn_interface_capi.cc), so the NNInterface stress
// test exercises the engine's real hand-over rules.
//
// A slot is DIRTY from p3hip_load_slot until its result has been fetched (p3hip_get_slot /
// p3hip_get_ownership), not merely until a run has picked it up.  The reference's infer thread
// calls RunInference() with its lock held while workers call LoadBatch() outside the lock
// (cc/nn/nn_interface.cc:276,354): a load can land while a run is gathering, be evaluated by
// that run, and only be counted as "loaded" by NNInterface for the NEXT run (the sequence
// nn_interface.cc:351-361 describes; harmless for an engine that always runs every slot).  If
// the run consumed the flag, the next run would skip the slot and hand its caller nothing.
// Keeping it dirty until fetched makes that next run evaluate it again — the only cost is one
// redundant row in that rare interleaving; in the ordinary flow every result is fetched
// before the next run starts.
#pragma once
#include <atomic>
#include <cstdint>
#include <vector>

namespace p3 {

class SlotStates {
 public:
  enum : uint8_t { kIdle = 0, kLoaded = 1, kEvaluated = 2 };
  explicit SlotStates(int n = 0) : st_(n), row_(n, -1) {
    for (auto& s : st_) s.store(kIdle, std::memory_order_relaxed);
  }
  int size() const { return (int)st_.size(); }
  uint8_t state(int slot) const { return st_[slot].load(std::memory_order_acquire); }
  // LoadBatch: the features of `slot` have been written (release pairs with gather's acquire).
  void loaded(int slot) { st_[slot].store(kLoaded, std::memory_order_release); }
  // Start of a run: calls take(slot, row) for every dirty slot (every slot when `all`), in slot
  // order, rows dense from 0; returns the number of rows.  Slots that are not part of this
  // run lose their row (their old result is about to be overwritten).
  template <class F>
  int gather(bool all, F&& take) {
    int n = 0;
    for (int s = 0; s < size(); ++s) {
      const uint8_t v = st_[s].load(std::memory_order_acquire);
      if (v != kIdle || all) {
        take(s, n);
        uint8_t expect = kLoaded;   // a load landing after this point keeps the slot kLoaded
        st_[s].compare_exchange_strong(expect, kEvaluated, std::memory_order_acq_rel);
        row_[s] = n++;
      } else {
        row_[s] = -1;
      }
    }
    return n;
  }
  // Dense row of `slot` in the last run, -1 if it was not evaluated.
  int row(int slot) const { return row_[slot]; }
  // GetBatch: the result has been handed over; the slot stays out of later runs until reloaded.
  void fetched(int slot) {
    uint8_t expect = kEvaluated;
    st_[slot].compare_exchange_strong(expect, kIdle, std::memory_order_acq_rel);
  }

 private:
  std::vector<std::atomic<uint8_t>> st_;
  std::vector<int> row_;
};

}  // namespace p3
