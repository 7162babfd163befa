// symmetry.h — the D4 index maps of cc/game/symmetry.{h,cc}:20-80 (same enum order).
#pragma once
#include "rng.h"

namespace p3 {

enum Symmetry : uint8_t { kIdentity = 0, kRot90, kRot180, kRot270, kFlip, kFlipRot90, kFlipRot180, kFlipRot270 };
constexpr int kNumSymmetries = 8;

namespace symm {
inline int inv(int i, int n) { return n - i - 1; }
inline int flip(int idx, int n) { return (idx / n) * n + inv(idx % n, n); }
inline int rot(int idx, int n, int quarter) {  // quarter 1,2,3 = 90,180,270
  int i = idx / n, j = idx % n;
  if (quarter == 1) return j * n + inv(i, n);
  if (quarter == 2) return inv(i, n) * n + inv(j, n);
  return inv(j, n) * n + i;
}
}  // namespace symm

inline int TransformIndex(Symmetry s, int idx, int n) {
  switch (s) {
    case kIdentity: return idx;
    case kRot90: return symm::rot(idx, n, 1);
    case kRot180: return symm::rot(idx, n, 2);
    case kRot270: return symm::rot(idx, n, 3);
    case kFlip: return symm::flip(idx, n);
    case kFlipRot90: return symm::rot(symm::flip(idx, n), n, 1);
    case kFlipRot180: return symm::rot(symm::flip(idx, n), n, 2);
    case kFlipRot270: return symm::rot(symm::flip(idx, n), n, 3);
  }
  return idx;
}

inline int TransformInv(Symmetry s, int idx, int n) {
  switch (s) {
    case kIdentity: return idx;
    case kRot90: return symm::rot(idx, n, 3);
    case kRot180: return symm::rot(idx, n, 2);
    case kRot270: return symm::rot(idx, n, 1);
    case kFlip: return symm::flip(idx, n);
    case kFlipRot90: return symm::flip(symm::rot(idx, n, 3), n);
    case kFlipRot180: return symm::flip(symm::rot(idx, n, 2), n);
    case kFlipRot270: return symm::flip(symm::rot(idx, n, 1), n);
  }
  return idx;
}

// GetRandomSymmetry, cc/game/symmetry.h:33-35
inline Symmetry RandomSymmetry(PRng& rng) { return (Symmetry)RandRange(rng, 0, kNumSymmetries); }

// The 19 x 19 maps as tables (every leaf transforms five feature grids forward and three output grids back:
// per-element div / mod chains were 9 % of the self-play host's time).  Other sizes keep the arithmetic.
struct SymmetryTables {
  static constexpr int kN = 19;
  uint16_t fwd[kNumSymmetries][kN * kN], inv[kNumSymmetries][kN * kN];
  SymmetryTables() {
    for (int s = 0; s < kNumSymmetries; ++s)
      for (int i = 0; i < kN * kN; ++i) {
        fwd[s][i] = (uint16_t)TransformIndex((Symmetry)s, i, kN);
        inv[s][i] = (uint16_t)TransformInv((Symmetry)s, i, kN);
      }
  }
};
inline const SymmetryTables& symmetry_tables() {
  static const SymmetryTables t;
  return t;
}

template <class T>
inline void ApplySymmetry(Symmetry s, const T* in, T* out, int n) {
  if (n == SymmetryTables::kN) {
    const uint16_t* map = symmetry_tables().fwd[s];
    for (int i = 0; i < n * n; ++i) out[map[i]] = in[i];
    return;
  }
  for (int i = 0; i < n * n; ++i) out[TransformIndex(s, i, n)] = in[i];
}
template <class T>
inline void ApplyInverse(Symmetry s, const T* in, T* out, int n) {
  if (n == SymmetryTables::kN) {
    const uint16_t* map = symmetry_tables().inv[s];
    for (int i = 0; i < n * n; ++i) out[map[i]] = in[i];
    return;
  }
  for (int i = 0; i < n * n; ++i) out[TransformInv(s, i, n)] = in[i];
}

}  // namespace p3
