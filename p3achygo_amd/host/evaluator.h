// evaluator.h — the engine boundary as the host sees it (mirrors nn::Engine,
// cc/nn/engine/engine.h:22-43), with the two implementations the host links: the HIP engine
// bound through its C ABI, and the reference tests' NullEngine.
#pragma once
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/p3hip.h"
#include "board.h"

namespace p3 {

// ---- evaluator boundary (mirrors nn::Engine, cc/nn/engine/engine.h:22-43) -------------
struct Evaluator {
  virtual ~Evaluator() = default;
  virtual void Load(int slot, const p3hip_features& f) = 0;
  virtual bool Run() = 0;
  virtual void Get(int slot, p3hip_result& r) = 0;
  virtual void GetOwnership(int slot, float out[P3HIP_NUM_LOCS]) { std::memset(out, 0, sizeof(float) * P3HIP_NUM_LOCS); }
  // On-device NN cache (include/p3hip.h p3hip_cache_*): an engine without one says no and the host keeps its own.
  // LoadKeyed carries the position's 128-bit key and the symmetry of the features; GetKeyed returns the symmetry
  // of the result that comes back (the stored one on a hit) for the caller to undo.
  virtual bool EnableDeviceCache(int /*log2_entries*/) { return false; }
  virtual void LoadKeyed(int slot, const p3hip_features& f, uint64_t /*key_lo*/, uint64_t /*key_hi*/, int /*symmetry*/) { Load(slot, f); }
  virtual void GetKeyed(int slot, p3hip_result& r, int* /*symmetry: left as given*/, bool* from_cache) {
    Get(slot, r);
    if (from_cache) *from_cache = false;
  }
};

// Uniform policy, even outcome, zero score: the reference's NullEngine
// (cc/mcts/__tests__/search_test.cc:49-65).  Lets the host be tested without a GPU.
struct NullEvaluator final : Evaluator {
  void Load(int, const p3hip_features&) override {}
  bool Run() override { return true; }
  void Get(int, p3hip_result& r) override {
    for (int i = 0; i < kNumMoves; ++i) {
      r.move_logits[i] = 0.0f;
      r.move_probs[i] = 1.0f / kNumMoves;
      r.opt_move_probs[i] = 1.0f / kNumMoves;
    }
    r.value_probs[0] = r.value_probs[1] = 0.5f;
    for (int i = 0; i < P3HIP_NUM_SCORE_LOGITS; ++i) r.score_probs[i] = 0.0f;
    r.score_probs[400] = 1.0f;
    r.err2_outcome = 0.0f;
  }
};

// A deterministic stand-in for a trained network (test aid): every output is a hash of the feature record, so
// different positions get different policies, outcomes and scores, the same position always the same ones,
// whatever batch or slot it is evaluated in.  Lets tests compare whole searches and games across schedules.
struct HashEvaluator final : Evaluator {
  std::vector<p3hip_features> slots;
  std::vector<p3hip_result> results;   // of the last run: a slot may be reloaded before its old result is fetched
  std::vector<char> dirty;
  explicit HashEvaluator(int n) : slots((size_t)n), results((size_t)n), dirty((size_t)n, 0) {}
  void Load(int slot, const p3hip_features& f) override { slots[(size_t)slot] = f; dirty[(size_t)slot] = 1; }
  bool Run() override {
    for (size_t i = 0; i < slots.size(); ++i)
      if (dirty[i]) { Evaluate(slots[i], results[i]); dirty[i] = 0; }
    return true;
  }
  static uint64_t Mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
  }
  static void Evaluate(const p3hip_features& f, p3hip_result& r) {
    uint64_t h = 0xcbf29ce484222325ull;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&f);
    for (size_t i = 0; i < sizeof f; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    auto unit = [h](uint64_t i) { return float(Mix(h + i * 0x9e3779b97f4a7c15ull) >> 40) * (1.0f / 16777216.0f); };
    float mx = -1e30f, sum = 0;
    for (int i = 0; i < kNumMoves; ++i) { r.move_logits[i] = 6.0f * unit(i) - 3.0f; mx = std::max(mx, r.move_logits[i]); }
    for (int i = 0; i < kNumMoves; ++i) { r.move_probs[i] = std::exp(r.move_logits[i] - mx); sum += r.move_probs[i]; }
    for (int i = 0; i < kNumMoves; ++i) { r.move_probs[i] /= sum; r.opt_move_probs[i] = r.move_probs[i]; }
    const float win = 0.05f + 0.9f * unit(1000);
    r.value_probs[0] = 1.0f - win;
    r.value_probs[1] = win;
    for (int i = 0; i < P3HIP_NUM_SCORE_LOGITS; ++i) r.score_probs[i] = 0.0f;
    const int centre = 400 + int(40.0f * unit(1001)) - 20;
    r.score_probs[centre] = 0.6f;
    r.score_probs[centre + 3] = 0.4f;
    r.err2_outcome = 0.1f * unit(1002);
  }
  void Get(int slot, p3hip_result& r) override { r = results[(size_t)slot]; }
};

// The HIP engine, bound through its C ABI exactly as a foreign host would bind it.
struct HipEvaluator final : Evaluator {
  void* lib = nullptr;
  p3hip_engine* eng = nullptr;
  decltype(&p3hip_create) create = nullptr;
  decltype(&p3hip_destroy) destroy = nullptr;
  decltype(&p3hip_load_slot) load = nullptr;
  decltype(&p3hip_run) run = nullptr;
  decltype(&p3hip_get_slot) get = nullptr;
  decltype(&p3hip_get_ownership) get_own = nullptr;
  decltype(&p3hip_last_error) last_error = nullptr;
  decltype(&p3hip_create_error) create_error = nullptr;
  decltype(&p3hip_cache_enable) cache_enable = nullptr;
  decltype(&p3hip_load_slot_keyed) load_keyed = nullptr;
  decltype(&p3hip_get_slot_keyed) get_keyed = nullptr;
  decltype(&p3hip_cache_stats) cache_stats = nullptr;
  std::string err;

  bool Open(const char* lib_path, const char* weights, int batch, int device, uint32_t flags = 0) {
    lib = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { err = dlerror(); return false; }
    create = (decltype(create))dlsym(lib, "p3hip_create");
    destroy = (decltype(destroy))dlsym(lib, "p3hip_destroy");
    load = (decltype(load))dlsym(lib, "p3hip_load_slot");
    run = (decltype(run))dlsym(lib, "p3hip_run");
    get = (decltype(get))dlsym(lib, "p3hip_get_slot");
    get_own = (decltype(get_own))dlsym(lib, "p3hip_get_ownership");
    last_error = (decltype(last_error))dlsym(lib, "p3hip_last_error");
    create_error = (decltype(create_error))dlsym(lib, "p3hip_create_error");
    cache_enable = (decltype(cache_enable))dlsym(lib, "p3hip_cache_enable");
    load_keyed = (decltype(load_keyed))dlsym(lib, "p3hip_load_slot_keyed");
    get_keyed = (decltype(get_keyed))dlsym(lib, "p3hip_get_slot_keyed");
    cache_stats = (decltype(cache_stats))dlsym(lib, "p3hip_cache_stats");
    if (!create || !destroy || !load || !run || !get) { err = "missing p3hip symbols"; return false; }
    eng = create(weights, batch, 1, device, flags);
    if (!eng) { err = create_error ? create_error() : "p3hip_create failed"; return false; }
    return true;
  }
  ~HipEvaluator() override {
    if (eng) destroy(eng);
    if (lib) dlclose(lib);
  }
  // The reference's engines abort on any failure (CUDA_OK / CHECK, trt_engine.cc:27-35); the
  // adapter over the status-returning C ABI does the same, loudly, instead of handing the
  // search an uninitialised result.
  [[noreturn]] void Fatal(const char* what, int slot, int rc) {
    std::fprintf(stderr, "p3hip %s(slot %d) failed with status %d: %s\n", what, slot, rc,
                 (eng && last_error) ? last_error(eng) : "");
    std::abort();
  }
  void Load(int slot, const p3hip_features& f) override {
    if (int rc = load(eng, slot, &f)) Fatal("load_slot", slot, rc);
  }
  bool Run() override {
    if (run(eng) != 0) { err = last_error(eng); return false; }
    return true;
  }
  void Get(int slot, p3hip_result& r) override {
    if (int rc = get(eng, slot, &r)) Fatal("get_slot", slot, rc);
  }
  bool EnableDeviceCache(int log2_entries) override {
    if (!cache_enable || !load_keyed || !get_keyed) return false;
    if (cache_enable(eng, log2_entries) != 0) { err = last_error(eng); return false; }
    return true;
  }
  void LoadKeyed(int slot, const p3hip_features& f, uint64_t lo, uint64_t hi, int symmetry) override {
    if (int rc = load_keyed(eng, slot, &f, lo, hi, symmetry)) Fatal("load_slot_keyed", slot, rc);
  }
  void GetKeyed(int slot, p3hip_result& r, int* symmetry, bool* from_cache) override {
    int sym = 0, hit = 0;
    if (int rc = get_keyed(eng, slot, &r, &sym, &hit)) Fatal("get_slot_keyed", slot, rc);
    if (symmetry) *symmetry = sym;
    if (from_cache) *from_cache = hit != 0;
  }
  void GetOwnership(int slot, float out[P3HIP_NUM_LOCS]) override {
    if (!get_own) Fatal("get_ownership (symbol missing)", slot, -1);
    if (int rc = get_own(eng, slot, out)) Fatal("get_ownership", slot, rc);
  }
};

}  // namespace p3
