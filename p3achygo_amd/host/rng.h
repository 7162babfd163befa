// rng.h — PRng / RandRange / Probability with the reference's exact bit behaviour.
//   PRng        : PCG-XSH-RR 64/32, four streams      cc/core/rand.cc:7-98
//   RandRange   : mask-and-reject, [lo, hi)            cc/core/rand.cc:100-121
//   Probability : Uniform via mantissa trick (rand>>9), Gumbel = -ln(-ln u)
//                                                      cc/core/probability.cc:12-52
// Unlike the reference there is no time-seeded default constructor: every generator in
// the host is seeded from the run's root seed (SURVEY.md §9 "non-reproducibility sources").
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace p3 {

class PRng {
 public:
  explicit PRng(uint64_t s0 = 0, uint64_t s1 = 0, uint64_t s2 = 0, uint64_t s3 = 0) { seed(s0, s1, s2, s3); }
  void seed(uint64_t s0, uint64_t s1 = 0, uint64_t s2 = 0, uint64_t s3 = 0) {
    st_[0] = s0 + kInc[0]; st_[1] = s1 + kInc[1]; st_[2] = s2 + kInc[2]; st_[3] = s3 + kInc[3];
  }
  uint32_t next() { return step(0); }
  uint64_t next64() {
    uint64_t hi = step(0), lo = step(1);
    return (hi << 32) | lo;
  }
  // 128-bit draw as (high64, low64); same stream use as PRng::next128 (rand.cc:83-97)
  void next128(uint64_t& hi, uint64_t& lo) {
    uint64_t r0 = step(0), r1 = step(1), r2 = step(2), r3 = step(3);
    hi = (r0 << 32) | r1;
    lo = (r2 << 32) | r3;
  }

 private:
  static constexpr uint64_t kMult = 6364136223846793005ull;
  static constexpr uint64_t kInc[4] = {1442695040888963407ull, 6364136223846793007ull,
                                       1865811235122147685ull, 7664345821815920749ull};
  uint32_t step(int k) {
    uint64_t x = st_[k];
    unsigned rot = (unsigned)(x >> 59);
    st_[k] = x * kMult + kInc[k];
    x ^= x >> 18;
    uint32_t v = (uint32_t)(x >> 27);
    return v >> rot | v << (-rot & 31);
  }
  uint64_t st_[4];
};

inline int RandRange(PRng& rng, int lo, int hi) {
  if (lo == hi) return lo;
  uint32_t width = uint32_t(hi) - uint32_t(lo);
  uint32_t mask = 0, shift = 1;
  while (width >> shift) { mask = mask << 1 | 1u; ++shift; }
  mask = mask << 1 | 1u;
  uint32_t r = rng.next();
  while ((r & mask) >= width) r = rng.next();
  return (int)(r & mask) + lo;
}

class Probability {
 public:
  explicit Probability(uint64_t seed = 0) : rng_(seed) {}
  PRng& prng() { return rng_; }
  float Uniform() {
    uint32_t x = (127u << 23) | (rng_.next() >> 9);
    float f;
    std::memcpy(&f, &x, 4);
    return f - 1.0f;
  }
  float GumbelSample() { return -logf(-logf(Uniform())); }
  float Exponential() {
    float u = 0;
    while (u == 0) u = Uniform();
    return -std::log(u);
  }
  float Gaussian() {
    auto nz = [&]() { float u = 0.0f; while (u == 0.0f) u = Uniform(); return u; };
    const float u0 = nz(), u1 = nz();
    return std::sqrt(-2 * std::log(u0)) * std::sin(2 * M_PI * u1);
  }

 private:
  PRng rng_;
};

}  // namespace p3
