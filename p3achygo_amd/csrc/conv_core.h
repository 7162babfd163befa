// conv_core.h — CDNA4 (gfx950) building blocks for the per-position fused trunk kernels.
//
// Design: one 512-thread workgroup (8 wave64, two per SIMD)
// owns NPOS whole 19x19 positions.  The activated input of the current conv layer lives
// in LDS ("act buffer"): one slot of NCH 16-byte chunks (8 fp16 channels each) per board
// point, laid out on a column-padded grid (row stride S = 19 + pad) with zero pad slots,
// so a KxK tap is a constant slot shift and needs no bounds checks.  Chunks are
// XOR-swizzled by slot so that ds_read_b128 fragment reads are bank-conflict free.
//
// Convolutions are implicit GEMMs on v_mfma_f32_32x32x16_f16 in the orientation
//   D'[cout][loc] = sum_k W[cout][k] * act[k][loc]
// (A operand = weights, B operand = activations) so that each lane ends up holding, for
// ONE board point, groups of 4 consecutive output channels: the epilogue packs them to
// fp16 and writes 8-byte pieces straight into the act buffer (next layer's operand) or
// into the channel-blocked global layout [C/8][361][8].
//
// Weights never touch registers on the way in: the host pre-packs, per residual block,
// one contiguous stream of 8 KiB "macro-steps" in exactly the LDS image order; the eight
// waves copy it with global_load_lds_dwordx4 (one 1 KiB piece per wave per macro-step)
// into an R-slot LDS ring, D macro-steps ahead, with counted vmcnt and one raw s_barrier
// per macro-step.  The stream is circular, so the prefetch runs across layer and
// position boundaries without draining.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace p3 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWG = 512;          // threads per workgroup (8 waves)
constexpr int kNLoc = 361;
constexpr int kBL = 19;
constexpr int kRingSlotBytes = 8192;
constexpr int kRingSlots = 5;     // R
constexpr int kRingDepth = 3;     // D = R - 2 (see ring_acquire)
constexpr int kRingBytes = kRingSlots * kRingSlotBytes;

// ---------------------------------------------------------------------------------------
// Geometry of one conv "space": NPOS positions, NT_TOTAL 32-wide location tiles each.
template <int NPOS_, int CB_, int KW_>
struct Geo {
  static constexpr int NPOS = NPOS_;
  static constexpr int CB = CB_;                 // channels resident in the act buffer
  static constexpr int NCH = CB / 8;             // 16-byte chunks per slot
  static constexpr int SLOTB = CB * 2;           // bytes per slot
  static constexpr int PAD = KW_ / 2;
  static constexpr int S = kBL + PAD;            // padded row stride
  static constexpr int NROWS = (kBL - 1) * S + kBL;             // last valid row + 1
  static constexpr int NT_POS = (NROWS + 31) / 32;              // location tiles per position
  static constexpr int PADTOP = PAD * S + PAD;                  // slots above row 0
  static constexpr int PSLOTS = PADTOP + NT_POS * 32 + PADTOP;  // slots per position
  static constexpr int ACT_BYTES = NPOS * PSLOTS * SLOTB;
  static constexpr int NT_TOTAL = NPOS * NT_POS;
};

template <int NCH>
__device__ __forceinline__ int swz(int slot) {
  static_assert(NCH == 2 || NCH == 4 || NCH == 8 || NCH == 16, "NCH");
  return (slot / (16 / NCH)) & (NCH - 1);
}

// row (column-padded index) -> validity and plain location
template <int S>
__device__ __forceinline__ bool row_valid(int r, int& loc) {
  int y = (r * (65536 / S + 1)) >> 16;  // r / S for r < 1024
  int x = r - y * S;
  loc = y * kBL + x;
  return (x < kBL) && (y < kBL);
}

// ---------------------------------------------------------------------------------------
// Fast mish: x * tanh(softplus(x)) = x * w / (w + 2), w = e^x (e^x + 2).
__device__ __forceinline__ float mish_f(float x) {
  float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
  float w = e * (e + 2.0f);
  return x * w * __builtin_amdgcn_rcpf(w + 2.0f);
}

// ---------------------------------------------------------------------------------------
// Weight ring.
struct Ring {
  const char* gbase;   // packed stream (global), nms_total macro-steps of 8 KiB
  int nms_total;
  uint32_t lds_base;   // byte offset of the ring inside the dynamic LDS array
  int pf;              // next stream macro-step to prefetch (circular)
  int pf_slot;
  int slot;            // ring slot of the next macro-step to consume
};

__device__ __forceinline__ void ring_issue(Ring& r, char* smem) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const char* gp = r.gbase + (size_t)r.pf * kRingSlotBytes + wid * 1024 + lane * 16;
  char* lp = smem + r.lds_base + r.pf_slot * kRingSlotBytes + wid * 1024;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                   (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
  r.pf = (r.pf + 1 == r.nms_total) ? 0 : r.pf + 1;
  r.pf_slot = (r.pf_slot + 1 == kRingSlots) ? 0 : r.pf_slot + 1;
}

__device__ __forceinline__ void ring_init(Ring& r, char* smem, const void* gbase, int nms_total,
                                          uint32_t lds_base) {
  r.gbase = (const char*)gbase;
  r.nms_total = nms_total;
  r.lds_base = lds_base;
  r.pf = 0;
  r.pf_slot = 0;
  r.slot = 0;
  for (int i = 0; i < kRingDepth; ++i) ring_issue(r, smem);
}

// Makes the next macro-step readable and returns its LDS byte offset.
// Invariant on entry: this wave has exactly D glds in flight (for macro-steps m..m+D-1),
// possibly followed by younger ordinary loads/stores (which only make the wait stricter).
// vmcnt(D-1) => piece m of this wave landed; the barrier => every wave's piece landed, and
// every wave has consumed macro-step m-2 completely, whose slot ((m+D) mod R with R = D+2)
// is the one refilled below.  lgkmcnt(0) also retires this wave's LDS writes, so the
// barrier doubles as the "epilogue written" barrier between layers.
__device__ __forceinline__ uint32_t ring_acquire(Ring& r, char* smem) {
  static_assert(kRingDepth == 3, "vmcnt immediate below assumes D == 3");
  asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ring_issue(r, smem);
  uint32_t off = r.lds_base + r.slot * kRingSlotBytes;
  r.slot = (r.slot + 1 == kRingSlots) ? 0 : r.slot + 1;
  return off;
}

__device__ __forceinline__ void ring_drain() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------
// Wave tiling: 8 waves = CG cout-groups x LG location-groups; each wave owns MT(=2) cout
// tiles of 32 and NT location tiles of 32 (tiles lg, lg+LG, ...).
template <class G, int COUT_PASS>
struct Tiling {
  static constexpr int CG = COUT_PASS / 64;
  static constexpr int LG = 8 / CG;
  static constexpr int MT = 2;
  static constexpr int NT = (G::NT_TOTAL + LG - 1) / LG;
  static constexpr int KMS = kRingSlotBytes / (COUT_PASS * 32);  // k16-steps per macro-step
  static_assert(CG * LG == 8 && (CG == 1 || CG == 2), "tiling");
};

// slot index of row 0 of location tile t
template <class G>
__device__ __forceinline__ int tile_slot0(int t) {
  int p = t / G::NT_POS;
  int tt = t - p * G::NT_POS;
  return p * G::PSLOTS + G::PADTOP + tt * 32;
}

// One conv segment: acc[mt][j] += W_seg x act over NK16 k16-steps taken in the order
// (tap major, channel-chunk-pair minor).  KW = kernel width (1, 3 or 5); NTAPS_PAD is the
// number of taps in the packed stream (>= KW*KW, zero weights beyond).
template <class G, int COUT_PASS, int KW, int NTAPS_PAD, int NTn>
__device__ __forceinline__ void conv_segment(Ring& ring, char* smem, f32x16 (&acc)[2][NTn]) {
  using T = Tiling<G, COUT_PASS>;
  static_assert(NTn == T::NT, "accumulator shape");
  constexpr int NQ = G::CB / 16;                 // k16-steps per tap
  constexpr int NK16 = NTAPS_PAD * NQ;
  static_assert(NK16 % T::KMS == 0, "segment must be whole macro-steps");
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wid % T::CG, lg = wid / T::CG;
  const int lr = lane & 31, h = lane >> 5;

  int slot0[T::NT];
#pragma unroll
  for (int j = 0; j < T::NT; ++j) {
    int t = lg + j * T::LG;
    if (t >= G::NT_TOTAL) t = lg;  // padded tile: recompute a valid one, never stored
    slot0[j] = tile_slot0<G>(t) + lr;
  }
  // A fragment offset inside a k16 block: [h][cout][8] fp16
  const uint32_t a_off = (uint32_t)((h * COUT_PASS + cg * 64 + lr) * 16);

  uint32_t wslot = 0;
  h8 a_cur[2], b_cur[T::NT];
  auto load_frags = [&](int g, h8 (&a)[2], h8 (&b)[T::NT]) {
    const int tap = g / NQ, q = g - tap * NQ;
    int shift = 0;
    if (KW > 1) {
      int ky = tap / KW, kx = tap - ky * KW;
      shift = (tap < KW * KW) ? (ky - KW / 2) * G::S + (kx - KW / 2) : 0;
    }
    const uint32_t wk = wslot + (uint32_t)((g % T::KMS) * (COUT_PASS * 32));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) a[mt] = *(const h8*)(smem + wk + a_off + mt * 512);
#pragma unroll
    for (int j = 0; j < T::NT; ++j) {
      int s = slot0[j] + shift;
      int ch = (2 * q + h) ^ swz<G::NCH>(s);
      b[j] = *(const h8*)(smem + s * G::SLOTB + ch * 16);
    }
  };

  wslot = ring_acquire(ring, smem);
  load_frags(0, a_cur, b_cur);
#pragma unroll 2
  for (int g = 0; g < NK16; ++g) {
    h8 a_nxt[2], b_nxt[T::NT];
    if (g + 1 < NK16) {
      if (((g + 1) % T::KMS) == 0) wslot = ring_acquire(ring, smem);
      load_frags(g + 1, a_nxt, b_nxt);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < T::NT; ++j)
        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[mt], b_cur[j], acc[mt][j], 0, 0, 0);
    if (g + 1 < NK16) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) a_cur[mt] = a_nxt[mt];
#pragma unroll
      for (int j = 0; j < T::NT; ++j) b_cur[j] = b_nxt[j];
    }
  }
}

template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[2][NTn]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < NTn; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.0f;
}

// Channel (within the pass) of accumulator register quad g4 (0..3) of cout tile mt.
template <class G, int COUT_PASS>
__device__ __forceinline__ int acc_chan(int mt, int g4) {
  using T = Tiling<G, COUT_PASS>;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wid % T::CG;
  const int h = (threadIdx.x & 63) >> 5;
  return cg * 64 + mt * 32 + g4 * 8 + h * 4;
}

// Epilogue A: v = mish(acc * scale[c] + shift[c]) -> fp16 -> act buffer (chunk index
// cbase/8 + ...), only for valid rows.  `cofs` = channel offset of this cout pass inside
// the act buffer's channel space.
template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void epilogue_to_act(char* smem, f32x16 (&acc)[2][NTn],
                                                const float* __restrict__ scale,
                                                const float* __restrict__ shift, int cofs) {
  using T = Tiling<G, COUT_PASS>;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = wid / T::CG;
  const int lr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int c = cofs + acc_chan<G, COUT_PASS>(mt, g4);
      const f32x4 sc = *(const f32x4*)(scale + c);
      const f32x4 sh = *(const f32x4*)(shift + c);
#pragma unroll
      for (int j = 0; j < T::NT; ++j) {
        const int t = lg + j * T::LG;
        if (t >= G::NT_TOTAL) continue;
        const int p = t / G::NT_POS, tt = t - p * G::NT_POS;
        const int r = tt * 32 + lr;
        int loc;
        const bool ok = row_valid<G::S>(r, loc);
        h4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (_Float16)mish_f(acc[mt][j][g4 * 4 + i] * sc[i] + sh[i]);
        const int s = p * G::PSLOTS + G::PADTOP + r;
        const int ch = (c >> 3) ^ swz<G::NCH>(s);
        if (ok) *(h4*)(smem + s * G::SLOTB + ch * 16 + h * 8) = o;
      }
    }
  }
}

// Stage a CB-channel slice (channel blocks cblk0 .. cblk0+NCH-1) of the residual stream
// x[pos][C/8][361][8] (fp16) into the act buffer, applying y = mish(x*scale+shift)
// when PRE is set.  Only valid locations are written; pad slots stay zero.
template <class G, bool PRE>
__device__ __forceinline__ void stage_in(char* smem, const _Float16* __restrict__ x, int C,
                                         int pos0, int npos, int cblk0,
                                         const float* __restrict__ scale,
                                         const float* __restrict__ shift) {
  constexpr int ITEMS = G::NPOS * G::NCH * kNLoc;
  for (int it = threadIdx.x; it < ITEMS; it += kWG) {
    const int p = it / (G::NCH * kNLoc);
    const int rem = it - p * (G::NCH * kNLoc);
    const int kc = rem / kNLoc;
    const int loc = rem - kc * kNLoc;
    int pos = pos0 + p;
    if (pos >= npos) pos = npos - 1;
    const h8 v = *(const h8*)(x + ((size_t)pos * (C / 8) + cblk0 + kc) * (kNLoc * 8) + loc * 8);
    h8 o;
    if (PRE) {
      const int c = (cblk0 + kc) * 8;
      const f32x4 s0 = *(const f32x4*)(scale + c), s1 = *(const f32x4*)(scale + c + 4);
      const f32x4 t0 = *(const f32x4*)(shift + c), t1 = *(const f32x4*)(shift + c + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = (_Float16)mish_f((float)v[i] * s0[i] + t0[i]);
        o[i + 4] = (_Float16)mish_f((float)v[i + 4] * s1[i] + t1[i]);
      }
    } else {
      o = v;
    }
    const int y = (loc * 3450) >> 16;  // loc / 19 for loc < 361
    const int xx = loc - y * kBL;
    const int s = p * G::PSLOTS + G::PADTOP + y * G::S + xx;
    const int ch = kc ^ swz<G::NCH>(s);
    *(h8*)(smem + s * G::SLOTB + ch * 16) = o;
  }
}

template <class G>
__device__ __forceinline__ void act_zero(char* smem) {
  for (int i = threadIdx.x * 16; i < G::ACT_BYTES; i += kWG * 16) *(f32x4*)(smem + i) = f32x4{0, 0, 0, 0};
}

// Epilogue B: out[c][loc] = acc (+ residual read from the same place) -> fp16 global, in
// the channel-blocked layout.  cofs = first channel of this cout pass.
template <class G, int COUT_PASS, bool RESIDUAL, int NTn>
__device__ __forceinline__ void epilogue_to_global(f32x16 (&acc)[2][NTn],
                                                   _Float16* __restrict__ x, int C, int pos0,
                                                   int npos, int cofs) {
  using T = Tiling<G, COUT_PASS>;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = wid / T::CG;
  const int lr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < T::NT; ++j) {
    const int t = lg + j * T::LG;
    if (t >= G::NT_TOTAL) continue;
    const int p = t / G::NT_POS, tt = t - p * G::NT_POS;
    const int r = tt * 32 + lr;
    int loc;
    const bool ok = row_valid<G::S>(r, loc) && (pos0 + p < npos);
    if (!ok) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c = cofs + acc_chan<G, COUT_PASS>(mt, g4);
        _Float16* px = x + ((size_t)(pos0 + p) * (C / 8) + (c >> 3)) * (kNLoc * 8) + loc * 8 + h * 4;
        h4 o;
        if (RESIDUAL) {
          const h4 rv = *(const h4*)px;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (_Float16)(acc[mt][j][g4 * 4 + i] + (float)rv[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (_Float16)acc[mt][j][g4 * 4 + i];
        }
        *(h4*)px = o;
      }
    }
  }
}

}  // namespace p3
