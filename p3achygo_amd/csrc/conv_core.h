// conv_core.h — CDNA4 (gfx950) building blocks for the per-position fused trunk kernels.
//
// Design: one 512-thread workgroup (8 wave64, two per SIMD)
// owns NPOS whole 19x19 positions.  The activated input of the current conv layer lives
// in LDS ("act buffer"): one slot of NCH 16-byte chunks (8 fp16 channels each) per board
// point, laid out on a column-padded grid (row stride S = 19 + pad) with zero pad slots,
// so a KxK tap is a constant slot shift and needs no bounds checks.  Slots are
// padded by 16 bytes so that ds_read_b128 fragment reads are bank-conflict free.
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
constexpr int kKMS = 4;           // k16-steps (K = 64) per ring macro-step
constexpr int kRingSlots = 3;     // R
constexpr int kRingDepth = 2;     // D = R - 1 (see ring_acquire)
#ifndef P3_RING_DEPTH4
#define P3_RING_DEPTH4 2
#endif
constexpr int kRingDepth4 = P3_RING_DEPTH4;   // ring depth of the 4-wave block kernel (K = 32 steps): 3 and 4 measure the same as 2 (gpurun_out/ringdepth_ab.log)
// bytes of one macro-step for a COUT_PASS-wide weight panel: `kms` blocks of [2][CP][8] fp16
constexpr int ring_slot_bytes(int cout_pass, int kms = kKMS) { return kms * cout_pass * 32; }
constexpr int ring_bytes(int cout_pass, int kms = kKMS, int depth = kRingDepth) { return (depth + 1) * ring_slot_bytes(cout_pass, kms); }

// ---------------------------------------------------------------------------------------
// Geometry of one conv "space": NPOS positions, NT_TOTAL 32-wide location tiles each.
// NW = waves per workgroup (8; 4 for the two-workgroups-per-CU form of the C = 128 block kernel),
// KMS = k16 steps per ring macro-step, RD = macro-steps the weight ring prefetches ahead (RD + 1 slots).
template <int NPOS_, int CB_, int KW_, int NW_ = 8, int KMS_ = kKMS, int RD_ = kRingDepth>
struct Geo {
  static constexpr int NPOS = NPOS_;
  static constexpr int NW = NW_;
  static constexpr int KMS = KMS_;
  static constexpr int RD = RD_;
  static constexpr int CB = CB_;                 // channels resident in the act buffer
  static constexpr int NCH = CB / 8;             // 16-byte chunks per slot
  // bytes per slot: the channels plus one 16-byte pad.  The slot stride is then an odd
  // number of 16-byte bank groups (17, 9 or 3), so 16 consecutive slots read at the same
  // chunk hit 16 distinct bank groups: ds_read_b128 fragment reads are conflict-free with
  // plain (base + immediate) addressing, no swizzle arithmetic in the K loop.
  static constexpr int SLOTB = CB * 2 + 16;
  static constexpr int PAD = KW_ / 2;
  static constexpr int S = kBL + PAD;            // padded row stride
  static constexpr int NROWS = (kBL - 1) * S + kBL;             // last valid row + 1
  static constexpr int NT_POS = (NROWS + 31) / 32;              // location tiles per position
  static constexpr int PADTOP = PAD * S + PAD;                  // slots above row 0
  // slots per position.  Rows >= NROWS of the last location tile are never valid; their
  // (discarded) fragment reads run a few slots past this position into the next one or
  // into the ring, which always follows the act buffer in the same LDS allocation.
  static constexpr int PSLOTS = PADTOP + NROWS + PADTOP;
  static constexpr int ACT_BYTES = (NPOS * PSLOTS * SLOTB + 15) / 16 * 16;
  static constexpr int NT_TOTAL = NPOS * NT_POS;
};

// Returns v unchanged but opaque to the optimiser: address arithmetic derived from it is
// recomputed where it is used instead of being hoisted out of the position loop and
// spilled to scratch (a handful of VALU ops vs a scratch round trip per use).
__device__ __forceinline__ int launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
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
// Fast mish: x * tanh(softplus(x)) = x * (1 - 2 / (s*s + 1)),  s = e^x + 1.
// No clamp needed: e^x -> inf gives s*s+1 = inf, rcp = 0, result x; e^x -> 0 gives 0.
__device__ __forceinline__ float mish_f(float x) {
  const float s = __builtin_amdgcn_exp2f(x * 1.4426950408889634f) + 1.0f;
  const float r = __builtin_amdgcn_rcpf(__builtin_fmaf(s, s, 1.0f));
  return __builtin_fmaf(-2.0f * x, r, x);
}

// Packed form for the hot epilogues: BN + mish of four values, two per v_pk_* instruction
// (the compiler's own vectoriser leaves about half of these as scalar ops); exp2 and rcp stay
// scalar, there is no packed transcendental.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 mish_f2(f32x2 x) {
  const f32x2 t = x * 1.4426950408889634f;
  const f32x2 s = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
  const f32x2 d = __builtin_elementwise_fma(s, s, f32x2{1.0f, 1.0f});
  const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return __builtin_elementwise_fma(x * -2.0f, r, x);
}

// The hot epilogues take their BN parameters pre-multiplied by log2(e) (scale_log2e below, once per epilogue):
//   t = log2(e) * (sc * v + sh) = log2(e) * y,   e = 2^t = e^y,
//   mish(y) = y * (1 - 2 / ((e + 1)^2 + 1)) = t * (ln2 - 2 ln2 / (e * (e + 2) + 2)).
// Per pair of values: fma, exp2 x2, add, fma, rcp x2, fma, mul — five packed operations instead of six.
// e -> inf: d = inf, r = 0, result t * ln2 = y; e -> 0: r = 1/2, the last fma gives exactly 0.
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
__device__ __forceinline__ f32x4 scale_log2e(f32x4 v) { return v * kLog2e; }
__device__ __forceinline__ f32x2 mish_t2(f32x2 t) {
  const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  const f32x2 d = __builtin_elementwise_fma(e, e + 2.0f, f32x2{2.0f, 2.0f});
  const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return t * __builtin_elementwise_fma(r, f32x2{-2.0f * kLn2, -2.0f * kLn2}, f32x2{kLn2, kLn2});
}
// scl, shl: folded BN scale / shift times log2(e)
__device__ __forceinline__ h4 bn_mish4_l2(f32x4 v, f32x4 scl, f32x4 shl) {
  const f32x2 a = mish_t2(__builtin_elementwise_fma(f32x2{v[0], v[1]}, f32x2{scl[0], scl[1]}, f32x2{shl[0], shl[1]}));
  const f32x2 b = mish_t2(__builtin_elementwise_fma(f32x2{v[2], v[3]}, f32x2{scl[2], scl[3]}, f32x2{shl[2], shl[3]}));
  return h4{(_Float16)a[0], (_Float16)a[1], (_Float16)b[0], (_Float16)b[1]};
}
// The same arithmetic on EIGHT values stage by stage (4 packed operations or 8 transcendentals per stage, the
// stages pinned in this order): bn_mish4_l2 is one dependent chain per pair of values; eight independent values keep
// the VALU's issue slots filled (a wave with its SIMD's VALU to itself runs the chains at their latency).
// Bit-identical to bn_mish4_l2.
__device__ __forceinline__ void bn_mish8_l2(f32x4 v0, f32x4 v1, f32x4 sc0, f32x4 sh0, f32x4 sc1, f32x4 sh1, h4& o0, h4& o1) {
  f32x2 t[4], w[4];
  t[0] = __builtin_elementwise_fma(f32x2{v0[0], v0[1]}, f32x2{sc0[0], sc0[1]}, f32x2{sh0[0], sh0[1]});
  t[1] = __builtin_elementwise_fma(f32x2{v0[2], v0[3]}, f32x2{sc0[2], sc0[3]}, f32x2{sh0[2], sh0[3]});
  t[2] = __builtin_elementwise_fma(f32x2{v1[0], v1[1]}, f32x2{sc1[0], sc1[1]}, f32x2{sh1[0], sh1[1]});
  t[3] = __builtin_elementwise_fma(f32x2{v1[2], v1[3]}, f32x2{sc1[2], sc1[3]}, f32x2{sh1[2], sh1[3]});
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = f32x2{__builtin_amdgcn_exp2f(t[k][0]), __builtin_amdgcn_exp2f(t[k][1])};
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = __builtin_elementwise_fma(w[k], w[k] + 2.0f, f32x2{2.0f, 2.0f});
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = f32x2{__builtin_amdgcn_rcpf(w[k][0]), __builtin_amdgcn_rcpf(w[k][1])};
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = t[k] * __builtin_elementwise_fma(w[k], f32x2{-2.0f * kLn2, -2.0f * kLn2}, f32x2{kLn2, kLn2});
  __builtin_amdgcn_sched_barrier(0);
  o0 = h4{(_Float16)t[0][0], (_Float16)t[0][1], (_Float16)t[1][0], (_Float16)t[1][1]};
  o1 = h4{(_Float16)t[2][0], (_Float16)t[2][1], (_Float16)t[3][0], (_Float16)t[3][1]};
}

// ---------------------------------------------------------------------------------------
// Weight ring.  RS = bytes per macro-step (16 KiB for 128-wide panels, 8 KiB for 64-wide, 4 KiB
// for the 4-wave kernel's K = 32 steps); each of the NW waves copies RS/NW bytes (G = RS/NW/1024
// glds of 1 KiB) per macro-step.
template <int RS, int NW = 8, int D = kRingDepth>
struct Ring {
  static constexpr int DEPTH = D, SLOTS = D + 1;
  static constexpr int G = RS / (NW * 1024);
  static constexpr int PIECE = RS / NW;
  static_assert((G == 1 || G == 2) && G * NW * 1024 == RS, "ring slot size");
  // Scalar state only, kept as running pointers and rotating slot addresses so that one
  // acquire costs a handful of SALU instructions (an index-based ring recomputed two 64-bit
  // stream addresses, three LDS addresses and three wrap-arounds per acquire: ~100 scalar
  // instructions at the head of every unrolled K-loop body, stalling both waves of a SIMD
  // at the same time).
  const char* gcur;    // this wave's piece of the next macro-step to prefetch (lane offset excluded)
  const char* gbeg;    // ... of macro-step 0
  const char* gend;    // ... one past the last macro-step (the stream is circular)
  uint32_t fill[D + 1];  // LDS address of this wave's piece in the slot the next, next+1, ... prefetch fills
  uint32_t use[D + 1];   // LDS offset of the slot the next, next+1, ... acquire returns
  uint32_t lds_base;   // byte offset of the ring inside the dynamic LDS array
  int tol;             // acquires left that must tolerate `extra` younger register loads/stores
  int extra;           // 12, 24, 36 or 48 (see ring_note_inflight)
};

// Register prefetch of the next activation slice: every thread issues exactly kXLoads
// 16-byte global loads (see stage_load) while the ring keeps streaming.
constexpr int kXLoads = 12;

template <int N>
__device__ __forceinline__ void rotate_left(uint32_t (&v)[N]) {
  const uint32_t t = v[0];
#pragma unroll
  for (int i = 0; i + 1 < N; ++i) v[i] = v[i + 1];
  v[N - 1] = t;
}

template <int RS, int NW, int D>
__device__ __forceinline__ void ring_issue(Ring<RS, NW, D>& r, char* smem) {
  const uint32_t lane16 = (threadIdx.x & 63) * 16;
  const char* gp = r.gcur + lane16;      // uniform base + 32-bit lane offset
  char* lp = smem + r.fill[0];
  // (P3_PROBE_*: timing probes, garbage results.  Without the ring's barriers k_block needs 5 % fewer cycles; without
  // its LDS-DMA 2 % fewer — and 15-19 % less wall time, but only because the ring then holds no weights and the chip
  // clocks 14 % higher on the lighter MFMA operands, not because of the DMA: profiles/r03_sq_pmc_ring_probes.txt.
  // Every re-arrangement of the DMA tried in round 3 — issued by one wave per SIMD for both, spread over the
  // macro-step instead of right behind the barrier — measured slower than this one: profiles/r03_ring_probes.txt)
#ifndef P3_PROBE_NO_RING_DMA
#pragma unroll
  for (int i = 0; i < Ring<RS, NW, D>::G; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + i * 1024),
                                     (__attribute__((address_space(3))) void*)(lp + i * 1024), 16, 0, 0);
#endif
  r.gcur += RS;
  if (r.gcur == r.gend) r.gcur = r.gbeg;
  rotate_left(r.fill);
}

template <int RS, int NW, int D>
__device__ __forceinline__ void ring_init(Ring<RS, NW, D>& r, char* smem, const void* gbase,
                                          int nms_total, uint32_t lds_base) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  r.gbeg = (const char*)gbase + wid * (RS / NW);
  r.gend = r.gbeg + (size_t)nms_total * RS;
  r.gcur = r.gbeg;
  r.lds_base = lds_base;
#pragma unroll
  for (int i = 0; i < D + 1; ++i) {
    r.use[i] = lds_base + i * RS;
    r.fill[i] = lds_base + i * RS + wid * (RS / NW);
  }
  r.tol = 0;
  r.extra = 0;
  for (int i = 0; i < D; ++i) ring_issue(r, smem);
}

// Makes the next macro-step readable and returns its LDS byte offset.
// Invariant on entry: this wave has exactly D*G glds in flight (macro-steps m..m+D-1),
// possibly followed by younger ordinary loads/stores (which only make the wait stricter).
// vmcnt((D-1)*G) => this wave's pieces of m landed; the barrier => every wave's pieces
// landed.  lgkmcnt(0) retires this wave's LDS reads and writes: every fragment read of
// macro-step m-1 has completed in every wave once the barrier is passed (the last k16 of
// m-1 is fetched into registers before this call), so its slot ((m+D) mod R, R = D+1) can
// be refilled below; the same wait makes the barrier double as the "epilogue written"
// barrier between layers.
// While ring.tol > 0 the wave also has kXLoads register loads in flight that were issued
// right after the D*G glds (ring_note_xloads): for the next D acquires the pieces that must
// have landed are still older than all of them, so the count to leave outstanding is
// (D-1)*G + kXLoads; the (D+1)-th acquire waits with vmcnt((D-1)*G) again, which retires them.
// LGKM = LDS operations of this wave that may stay outstanding across the acquire: 0 for
// the first acquire of a segment (it doubles as the "previous layer written" barrier); a
// K loop that knows how many reads it issued after the last fragment read of the macro-step
// being recycled passes that count instead and does not stall on its own recent reads.
template <int LGKM = 0, int RS = 0, int NW = 8, int D = kRingDepth>
__device__ __forceinline__ uint32_t ring_acquire(Ring<RS, NW, D>& r, char* smem) {
  static_assert(kXLoads == 12, "vmcnt immediates below");
  constexpr int B = (D - 1) * Ring<RS, NW, D>::G;   // glds of the younger macro-steps that may stay in flight
  static_assert(B + 48 <= 63, "vmcnt is a 6-bit count");
  if (r.tol > 0) {
    r.tol--;
    if (r.extra == 12) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"(B + 12), "n"(LGKM) : "memory");
    else if (r.extra == 24) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"(B + 24), "n"(LGKM) : "memory");
    else if (r.extra == 36) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"(B + 36), "n"(LGKM) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"(B + 48), "n"(LGKM) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"(B), "n"(LGKM) : "memory");
  }
#ifndef P3_PROBE_NO_RING_BARRIER
  __builtin_amdgcn_s_barrier();
#endif
  asm volatile("" ::: "memory");
  ring_issue(r, smem);
  const uint32_t off = r.use[0];
  rotate_left(r.use);
  return off;
}

// Call right after issuing `extra` (12, 24, 36 or 48) ordinary vector-memory operations whose
// completion should not be forced by the next D ring acquires (and nothing else since the
// last ring_issue): the glds those acquires wait for are older than all of them.
template <int RS, int NW, int D>
__device__ __forceinline__ void ring_note_inflight(Ring<RS, NW, D>& r, int extra) {
  r.tol = D;
  r.extra = extra;
}
template <int RS, int NW, int D>
__device__ __forceinline__ void ring_note_xloads(Ring<RS, NW, D>& r) { ring_note_inflight(r, kXLoads); }

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
  static constexpr int LG = G::NW / CG;
  static constexpr int MT = 2;
  static constexpr int NT = (G::NT_TOTAL + LG - 1) / LG;
  static constexpr int KMS = G::KMS;                // k16-steps per macro-step
  static constexpr int RS = ring_slot_bytes(COUT_PASS, G::KMS);
  static_assert(CG * LG == G::NW && (CG == 1 || CG == 2), "tiling");
};

// slot index of row 0 of location tile t
template <class G>
__device__ __forceinline__ int tile_slot0(int t) {
  int p = t / G::NT_POS;
  int tt = t - p * G::NT_POS;
  return p * G::PSLOTS + G::PADTOP + tt * 32;
}

// 16-byte LDS read hidden from hipcc's waitcnt bookkeeping (cdna_hip_programming.md §5.7
// form (iii)): the caller counts lgkmcnt itself and fences consumers with sched_barrier.
template <int OFF>
__device__ __forceinline__ h8 lds_read128(uint32_t addr) {
  h8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return r;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
  __builtin_amdgcn_sched_barrier(0);
}

// wait_lgkm for a count that is static only after unrolling
__device__ __forceinline__ void wait_lgkm_n(int n) {
  switch (n) {
    case 0: wait_lgkm<0>(); break;
    case 1: wait_lgkm<1>(); break;
    case 2: wait_lgkm<2>(); break;
    case 3: wait_lgkm<3>(); break;
    case 4: wait_lgkm<4>(); break;
    case 5: wait_lgkm<5>(); break;
    case 6: wait_lgkm<6>(); break;
    default: wait_lgkm<7>(); break;
  }
}

// One conv segment: acc[mt][j] += W_seg x act over NK16 k16-steps taken in the order
// (tap major, channel-chunk-pair minor).  KW = kernel width (1, 3 or 5); NTAPS_PAD is the
// number of taps in the packed stream (>= KW*KW, zero weights beyond).
//
// Schedule: every wave's instruction stream must be MFMA-dense on its own, because the two
// waves of a SIMD share one matrix pipe and the older wave wins every arbitration (a wave
// that stalls on LDS leaves the pipe idle while its partner is blocked behind its own
// in-order MFMAs).  Fragments are therefore fetched TWO k16-steps ahead into a 3-deep
// register ring with hand-issued ds_read_b128, and each MFMA burst waits with a counted
// lgkmcnt for exactly its own fragments (the 2+NT reads of the following step stay in
// flight).  Body = U fully unrolled steps (U % 3 == 0 or the whole segment), runtime outer
// loop; a new ring slot is acquired whenever the step being FETCHED enters a macro-step.
// SWAP: issue mfma(act fragment, weight fragment) instead, i.e. D[act row][weight row]: a lane
// then holds 4 consecutive ACT rows for one weight row (k_bdense: rows are channels).
template <class G, int COUT_PASS, int KW, int NTAPS_PAD, bool SWAP = false, int NTn = 0>
__device__ __forceinline__ void conv_segment(Ring<ring_slot_bytes(COUT_PASS, G::KMS), G::NW, G::RD>& ring, char* smem,
                                             f32x16 (&acc)[2][NTn]) {
  using T = Tiling<G, COUT_PASS>;
  static_assert(NTn == T::NT, "accumulator shape");
  constexpr int NQ = G::CB / 16;                 // k16-steps per tap
  constexpr int NK16 = NTAPS_PAD * NQ;
  constexpr int U = (NK16 <= 28) ? NK16 : 3 * NQ;  // unrolled body length
  constexpr int NOUT = NK16 / U;
  constexpr int NLD = 2 + T::NT;                 // LDS reads per step
  static_assert(NK16 % U == 0 && U % T::KMS == 0 && (NOUT == 1 || (U % 3 == 0 && U % NQ == 0)) &&
                NK16 >= 3, "segment shape");
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
  uint32_t a_addr = 0;       // LDS byte address of this lane's A fragment in the current slot
  uint32_t b_base[T::NT];    // per-tap: byte address of the shifted slot (+ lane half)
  h8 fa[3][2], fb[3][T::NT];

  // Issues piece `pc` of the LDS reads of step v of body o (v is static; v may run past U
  // into the next body, the caller guarantees the step exists).  Piece 0 = ring slot /
  // tap addressing + the two A reads, piece 1+j = B read of location tile j.  The pieces
  // are interleaved one per MFMA so that their issue hides in the MFMA shadows.
  auto fetch_piece = [&](int o, int v, int pc) {
    const int vv = v % U;                      // static step inside its body
    const int ob = o + v / U;                  // runtime body index
    const int q = vv % NQ;
    const int buf = v % 3;                     // == (global step) % 3 since U % 3 == 0 or NOUT == 1
    constexpr int KB = COUT_PASS * 32;         // bytes per k16 block
    const int kk = vv % T::KMS;
    if (pc == 0) {
      if (kk == 0) a_addr = ring_acquire(ring, smem) + a_off;
      if (vv % NQ == 0) {
        const int tap = ob * (U / NQ) + vv / NQ;
        int shift = 0;
        if (KW > 1) {
          const int ky = tap / KW, kx = tap - ky * KW;
          shift = (tap < KW * KW) ? (ky - KW / 2) * G::S + (kx - KW / 2) : 0;
        }
#pragma unroll
        for (int j = 0; j < T::NT; ++j) {
          const int sl = slot0[j] + shift;
          b_base[j] = (uint32_t)(sl * G::SLOTB + h * 16);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        // offset immediate must be a literal: dispatch on the (static) k16 index in the slot
        if (kk == 0) fa[buf][mt] = mt ? lds_read128<512>(a_addr) : lds_read128<0>(a_addr);
        else if (kk == 1) fa[buf][mt] = mt ? lds_read128<KB + 512>(a_addr) : lds_read128<KB>(a_addr);
        else if (kk == 2) fa[buf][mt] = mt ? lds_read128<2 * KB + 512>(a_addr) : lds_read128<2 * KB>(a_addr);
        else fa[buf][mt] = mt ? lds_read128<3 * KB + 512>(a_addr) : lds_read128<3 * KB>(a_addr);
      }
    } else {
      const int j = pc - 1;
      // chunk 2q+h of the slot: the k16 index q is static, so it is an immediate offset
      switch (q) {   // up to 24 k16 steps per tap (k_bdense: K = 384)
        case 0: fb[buf][j] = lds_read128<0>(b_base[j]); break;
        case 1: fb[buf][j] = lds_read128<32>(b_base[j]); break;
        case 2: fb[buf][j] = lds_read128<64>(b_base[j]); break;
        case 3: fb[buf][j] = lds_read128<96>(b_base[j]); break;
        case 4: fb[buf][j] = lds_read128<128>(b_base[j]); break;
        case 5: fb[buf][j] = lds_read128<160>(b_base[j]); break;
        case 6: fb[buf][j] = lds_read128<192>(b_base[j]); break;
        case 7: fb[buf][j] = lds_read128<224>(b_base[j]); break;
        case 8: fb[buf][j] = lds_read128<256>(b_base[j]); break;
        case 9: fb[buf][j] = lds_read128<288>(b_base[j]); break;
        case 10: fb[buf][j] = lds_read128<320>(b_base[j]); break;
        case 11: fb[buf][j] = lds_read128<352>(b_base[j]); break;
        case 12: fb[buf][j] = lds_read128<384>(b_base[j]); break;
        case 13: fb[buf][j] = lds_read128<416>(b_base[j]); break;
        case 14: fb[buf][j] = lds_read128<448>(b_base[j]); break;
        case 15: fb[buf][j] = lds_read128<480>(b_base[j]); break;
        case 16: fb[buf][j] = lds_read128<512>(b_base[j]); break;
        case 17: fb[buf][j] = lds_read128<544>(b_base[j]); break;
        case 18: fb[buf][j] = lds_read128<576>(b_base[j]); break;
        case 19: fb[buf][j] = lds_read128<608>(b_base[j]); break;
        case 20: fb[buf][j] = lds_read128<640>(b_base[j]); break;
        case 21: fb[buf][j] = lds_read128<672>(b_base[j]); break;
        case 22: fb[buf][j] = lds_read128<704>(b_base[j]); break;
        default: fb[buf][j] = lds_read128<736>(b_base[j]); break;
      }
    }
  };
  auto fetch = [&](int o, int v) {
#pragma unroll
    for (int pc = 0; pc < 1 + T::NT; ++pc) fetch_piece(o, v, pc);
  };

  fetch(0, 0);
  fetch(0, 1);
#pragma unroll 1
  for (int o = 0; o < NOUT; ++o) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool last_body = (NOUT == 1) || (o + 1 == NOUT);
      // step u's fragments are complete once at most the following step's reads remain
      if (u + 1 < U || !last_body) wait_lgkm<NLD>();
      else wait_lgkm<0>();
      const bool do_fetch = (u + 2 < U) || !last_body;
#pragma unroll
      for (int m = 0; m < 2 * T::NT; ++m) {
        const int mt = m / T::NT, j = m % T::NT;
        acc[mt][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[u % 3][j], fa[u % 3][mt], acc[mt][j], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u % 3][mt], fb[u % 3][j], acc[mt][j], 0, 0, 0);
        if (m < 1 + T::NT) {
          __builtin_amdgcn_sched_barrier(0);
          if (do_fetch) fetch_piece(o, u + 2, m);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
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

template <class G, int COUT_PASS>
__device__ __forceinline__ int cg_of() {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  return wid % Tiling<G, COUT_PASS>::CG;
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
// The 2 x 8 parameter quads of a layer (folded BN scale / shift of this lane's 32
// channels).  Loaded with epi_params BEFORE the barrier that precedes the epilogue so the
// L2 latency hides under the wait for the other waves.
struct EpiParams { f32x4 sc[8], sh[8]; };

template <class G, int COUT_PASS>
__device__ __forceinline__ void epi_params(EpiParams& ep, const float* __restrict__ scale,
                                           const float* __restrict__ shift, int cofs) {
  const int h = launder(threadIdx.x & 63) >> 5;
  const int c0 = cofs + cg_of<G, COUT_PASS>() * 64 + h * 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // k = mt*4 + g4 -> channel c0 + 8k
    ep.sc[k] = *(const f32x4*)(scale + c0 + 8 * k);
    ep.sh[k] = *(const f32x4*)(shift + c0 + 8 * k);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // times log2(e): bn_mish8_l2
    ep.sc[k] = scale_log2e(ep.sc[k]);
    ep.sh[k] = scale_log2e(ep.sh[k]);
  }
}

// The epilogue is split at the barrier: epilogue_math (BN + mish + fp16 pack, registers
// only) runs BEFORE the barrier that retires the readers of the act buffer, so the wave of
// a SIMD pair that finishes its K loop first transforms its tile under the other wave's
// MFMAs; epilogue_write (LDS stores only) runs after it.
template <int NTn>
struct EpiOut { h4 o[NTn][8]; };

template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void epilogue_math(EpiOut<NTn>& eo, f32x16 (&acc)[2][NTn], const EpiParams& ep) {
#pragma unroll
  for (int j = 0; j < NTn; ++j)
#pragma unroll
    for (int k = 0; k < 8; k += 2) {   // eight values in flight per stage
      const int mt = k >> 2, g4 = k & 3;
      const f32x4 v0 = {acc[mt][j][g4 * 4], acc[mt][j][g4 * 4 + 1], acc[mt][j][g4 * 4 + 2], acc[mt][j][g4 * 4 + 3]};
      const f32x4 v1 = {acc[mt][j][g4 * 4 + 4], acc[mt][j][g4 * 4 + 5], acc[mt][j][g4 * 4 + 6], acc[mt][j][g4 * 4 + 7]};
      bn_mish8_l2(v0, v1, ep.sc[k], ep.sh[k], ep.sc[k + 1], ep.sh[k + 1], eo.o[j][k], eo.o[j][k + 1]);
    }
}

template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void epilogue_write(char* smem, const EpiOut<NTn>& eo, int cofs) {
  using T = Tiling<G, COUT_PASS>;
  const int lane = launder(threadIdx.x & 63);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = wid / T::CG;
  const int lr = lane & 31, h = lane >> 5;
  const int cblk0 = (cofs >> 3) + cg_of<G, COUT_PASS>() * 8;
#pragma unroll
  for (int j = 0; j < T::NT; ++j) {
    const int t = lg + j * T::LG;
    if (t >= G::NT_TOTAL) continue;
    const int p = t / G::NT_POS, tt = t - p * G::NT_POS;
    const int r = tt * 32 + lr;
    int loc;
    const bool ok = row_valid<G::S>(r, loc);
    const uint32_t dst = (uint32_t)((p * G::PSLOTS + G::PADTOP + r) * G::SLOTB + cblk0 * 16 + h * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (ok) *(h4*)(smem + dst + k * 16) = eo.o[j][k];
  }
}

template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void epilogue_to_act(char* smem, f32x16 (&acc)[2][NTn],
                                                const EpiParams& ep, int cofs) {
  EpiOut<NTn> eo;
  epilogue_math<G, COUT_PASS, NTn>(eo, acc, ep);
  epilogue_write<G, COUT_PASS, NTn>(smem, eo, cofs);
}

// Full layer transition: parameters, math, barrier, stores.
template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void epilogue_layer(char* smem, f32x16 (&acc)[2][NTn],
                                               const float* __restrict__ scale,
                                               const float* __restrict__ shift) {
  EpiOut<NTn> eo;
  {
    EpiParams ep;
    epi_params<G, COUT_PASS>(ep, scale, shift, 0);
    epilogue_math<G, COUT_PASS, NTn>(eo, acc, ep);
  }
  lds_barrier();
  epilogue_write<G, COUT_PASS, NTn>(smem, eo, 0);
}

// Staging of a CB-channel slice (channel blocks cblk0 .. cblk0+NCH-1) of the residual
// stream x[pos][C/8][361][8] (fp16) into the act buffer, split in two so that the HBM/L2
// latency hides under MFMA work: stage_load issues this thread's kXLoads 16-byte loads
// into registers, stage_store applies y = mish(x*scale+shift) (when PRE) and writes the
// swizzled LDS image.  Only valid locations are written; pad slots stay zero.
template <class G>
struct XRegs { h8 v[kXLoads]; };

// Thread t owns (position, chunk) combo t/32 and board points (t%32) + 32*i, i < 12
// (12*32 = 384 >= 361), so its BN parameters are loaded once and addressing is trivial.
template <class G>
__device__ __forceinline__ void stage_load(XRegs<G>& xr, const _Float16* __restrict__ x, int C,
                                           int pos0, int npos, int cblk0) {
  static_assert(G::NPOS * G::NCH * 32 == G::NW * 64 && kXLoads * 32 >= kNLoc, "staging map: one (position, chunk) per 32 threads");
  const int tid = launder(threadIdx.x);
  const int combo = tid >> 5, l32 = tid & 31;
  const int p = combo / G::NCH, kc = combo - p * G::NCH;
  int pos = pos0 + p;
  if (pos >= npos) pos = npos - 1;
  const _Float16* src = x + ((size_t)pos * (C / 8) + cblk0 + kc) * (kNLoc * 8);
#pragma unroll
  for (int i = 0; i < kXLoads; ++i) {
    int loc = l32 + 32 * i;
    if (loc >= kNLoc) loc = kNLoc - 1;  // tail lanes re-read a valid item (not stored)
    xr.v[i] = *(const h8*)(src + loc * 8);
  }
}

template <class G, bool PRE>
__device__ __forceinline__ void stage_store(char* smem, const XRegs<G>& xr, int cblk0,
                                            const float* __restrict__ scale,
                                            const float* __restrict__ shift) {
  const int tid = launder(threadIdx.x);
  const int combo = tid >> 5, l32 = tid & 31;
  const int p = combo / G::NCH, kc = combo - p * G::NCH;
  f32x4 s0, s1, t0, t1;
  if (PRE) {
    const int c = (cblk0 + kc) * 8;
    s0 = scale_log2e(*(const f32x4*)(scale + c)); s1 = scale_log2e(*(const f32x4*)(scale + c + 4));
    t0 = scale_log2e(*(const f32x4*)(shift + c)); t1 = scale_log2e(*(const f32x4*)(shift + c + 4));
  }
#pragma unroll
  for (int i = 0; i < kXLoads; ++i) {
    const int loc = l32 + 32 * i;
    if (loc >= kNLoc) continue;
    const h8 v = xr.v[i];
    h8 o;
    if (PRE) {
      h4 lo, hi;
      bn_mish8_l2(f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]},
                  s0, t0, s1, t1, lo, hi);
      o = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else {
      o = v;
    }
    const int y = (loc * 3450) >> 16;  // loc / 19 for loc < 361
    const int xx = loc - y * kBL;
    const int sl = p * G::PSLOTS + G::PADTOP + y * G::S + xx;
    *(h8*)(smem + sl * G::SLOTB + kc * 16) = o;
    if (i & 1) __builtin_amdgcn_sched_barrier(0);  // bound the live range: 2 items in flight
  }
}

// BN + mish of a fetched slice in place (registers only), so that it can run before the
// barrier that frees the act buffer; pair with stage_store<G, false>.
template <class G>
__device__ __forceinline__ void stage_math(XRegs<G>& xr, int cblk0, const float* __restrict__ scale,
                                           const float* __restrict__ shift) {
  const int combo = launder(threadIdx.x) >> 5;
  const int kc = combo % G::NCH;
  const int c = (cblk0 + kc) * 8;
  const f32x4 s0 = scale_log2e(*(const f32x4*)(scale + c)), s1 = scale_log2e(*(const f32x4*)(scale + c + 4));
  const f32x4 t0 = scale_log2e(*(const f32x4*)(shift + c)), t1 = scale_log2e(*(const f32x4*)(shift + c + 4));
#pragma unroll
  for (int i = 0; i < kXLoads; ++i) {
    const h8 v = xr.v[i];
    h4 lo, hi;
    bn_mish8_l2(f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]},
                s0, t0, s1, t1, lo, hi);
    xr.v[i] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}

template <class G, bool PRE>
__device__ __forceinline__ void stage_in(char* smem, const _Float16* __restrict__ x, int C,
                                         int pos0, int npos, int cblk0,
                                         const float* __restrict__ scale,
                                         const float* __restrict__ shift) {
  XRegs<G> xr;
  stage_load<G>(xr, x, C, pos0, npos, cblk0);
  stage_store<G, PRE>(smem, xr, cblk0, scale, shift);
}

template <class G>
__device__ __forceinline__ void act_zero(char* smem) {
  // the zero is made here (opaque to the optimiser): a hoisted zero vector would be carried — spilled —
  // across the whole kernel for a use inside the position loop
  const float z = __builtin_bit_cast(float, launder(0));
  for (int i = threadIdx.x * 16; i < G::ACT_BYTES; i += G::NW * 64 * 16) *(f32x4*)(smem + i) = f32x4{z, z, z, z};
}

// Epilogue B: out[c][loc] = acc + residual (read from the same place) -> fp16 global, in
// the channel-blocked layout; cofs = first channel of this cout pass.  Split in two so the
// Residual loads can be issued before the conv segment whose result they are added to
// (their latency hides under its MFMAs): residual_load, then epilogue_store.
//
// A lane's accumulator quad (mt, g4) is one 8-byte HALF of the 16-byte [8 channels] piece of
// its board row; the other half sits in the partner lane 32 lanes away.  One
// v_permlane32_swap per dword turns "my half of channel blocks k and k+1" into "the whole piece
// of ONE block": block k in lanes 0-31, block k+1 in lanes 32-63.  Loads and stores then move
// 16 bytes per lane (half the instructions; the store tail is issue-bound) and a wave
// instruction covers two contiguous 512-byte runs.
template <class G, int COUT_PASS, int NTn>
struct ResRegs {
  h8 rv[NTn][4];       // piece of this lane's row, channel block 2*kp + (lane >> 5)
  uint32_t base[NTn];  // element offset of that piece for kp = 0
  bool ok[NTn];
};

template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void residual_addr(ResRegs<G, COUT_PASS, NTn>& rr, int C, int pos0,
                                              int npos, int cofs) {
  using T = Tiling<G, COUT_PASS>;
  const int lane = launder(threadIdx.x & 63);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = wid / T::CG;
  const int lr = lane & 31, h = lane >> 5;
  const int cblk = ((cofs + cg_of<G, COUT_PASS>() * 64) >> 3) + h;   // this lane's block of pair 0
#pragma unroll
  for (int j = 0; j < NTn; ++j) {
    const int t = lg + j * T::LG;
    const int tv = t < G::NT_TOTAL ? t : lg;
    const int p = tv / G::NT_POS, tt = tv - p * G::NT_POS;
    int loc;
    rr.ok[j] = row_valid<G::S>(tt * 32 + lr, loc) && (pos0 + p < npos) && (t < G::NT_TOTAL);
    if (!rr.ok[j]) loc = 0;
    const int pp = (pos0 + p < npos) ? pos0 + p : npos - 1;
    rr.base[j] = (uint32_t)((pp * (C / 8) + cblk) * (kNLoc * 8) + loc * 8);
  }
}

// Issues exactly NTn*4 sixteen-byte loads per lane (invalid rows read a valid dummy address).
template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void residual_load(ResRegs<G, COUT_PASS, NTn>& rr,
                                              const _Float16* __restrict__ x) {
#pragma unroll
  for (int j = 0; j < NTn; ++j)
#pragma unroll
    for (int kp = 0; kp < 4; ++kp)   // channel blocks 2*kp, 2*kp + 1
      rr.rv[j][kp] = *(const h8*)((const char*)x + (size_t)kp * (2 * kNLoc * 8 * 2) + (uint32_t)(rr.base[j] * 2u));
}

// (x = my half of block k, y = my half of block k+1)  <->  (low, high half of my own block's piece)
__device__ __forceinline__ void half_swap32(h4& x, h4& y) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  u2 xu = __builtin_bit_cast(u2, x), yu = __builtin_bit_cast(u2, y);
  auto s0 = __builtin_amdgcn_permlane32_swap(xu[0], yu[0], false, false);
  auto s1 = __builtin_amdgcn_permlane32_swap(xu[1], yu[1], false, false);
  xu[0] = s0[0]; yu[0] = s0[1];
  xu[1] = s1[0]; yu[1] = s1[1];
  x = __builtin_bit_cast(h4, xu);
  y = __builtin_bit_cast(h4, yu);
}

// this lane's residual values for channel blocks k = 2*kp (r0) and 2*kp + 1 (r1) of tile j
template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void residual_unpack(const ResRegs<G, COUT_PASS, NTn>& rr, int j, int kp, h4& r0, h4& r1) {
  const h8 v = rr.rv[j][kp];
  r0 = h4{v[0], v[1], v[2], v[3]};
  r1 = h4{v[4], v[5], v[6], v[7]};
  half_swap32(r0, r1);
}

template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void residual_add(f32x16 (&acc)[2][NTn], const ResRegs<G, COUT_PASS, NTn>& rr) {
#pragma unroll
  for (int j = 0; j < NTn; ++j)
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      h4 r0, r1;
      residual_unpack<G, COUT_PASS, NTn>(rr, j, kp, r0, r1);
      const int k0 = 2 * kp, k1 = 2 * kp + 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[k0 >> 2][j][(k0 & 3) * 4 + i] += (float)r0[i];
        acc[k1 >> 2][j][(k1 & 3) * 4 + i] += (float)r1[i];
      }
    }
}

template <class G, int COUT_PASS, bool RESIDUAL, int NTn>
__device__ __forceinline__ void epilogue_store(f32x16 (&acc)[2][NTn],
                                               const ResRegs<G, COUT_PASS, NTn>& rr,
                                               _Float16* __restrict__ x) {
#pragma unroll
  for (int j = 0; j < NTn; ++j)
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      const int k0 = 2 * kp, k1 = 2 * kp + 1;
      h4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0};
      if (RESIDUAL) residual_unpack<G, COUT_PASS, NTn>(rr, j, kp, r0, r1);
      h4 o0, o1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v0 = acc[k0 >> 2][j][(k0 & 3) * 4 + i], v1 = acc[k1 >> 2][j][(k1 & 3) * 4 + i];
        if (RESIDUAL) { v0 += (float)r0[i]; v1 += (float)r1[i]; }
        o0[i] = (_Float16)v0;
        o1[i] = (_Float16)v1;
      }
      half_swap32(o0, o1);   // every lane takes part
      const h8 piece = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
      if (rr.ok[j]) *(h8*)((char*)x + (size_t)kp * (2 * kNLoc * 8 * 2) + (uint32_t)(rr.base[j] * 2u)) = piece;
    }
}

template <class G, int COUT_PASS, bool RESIDUAL, int NTn>
__device__ __forceinline__ void epilogue_to_global(f32x16 (&acc)[2][NTn],
                                                   _Float16* __restrict__ x, int C, int pos0,
                                                   int npos, int cofs) {
  ResRegs<G, COUT_PASS, NTn> rr;
  residual_addr<G, COUT_PASS, NTn>(rr, C, pos0, npos, cofs);
  if (RESIDUAL) residual_load<G, COUT_PASS, NTn>(rr, x);
  epilogue_store<G, COUT_PASS, RESIDUAL, NTn>(acc, rr, x);
}

}  // namespace p3
