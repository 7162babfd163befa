// conv16.h — the conv segment and its epilogues on v_mfma_f32_16x16x32_f16.
//
// Same ownership as conv_core.h's 32x32x16 path (a wave owns 64 output channels x 96
// padded board rows, weights = A operand, activations = B operand, same LDS images and the
// same packed weight stream: one k32 step is two consecutive k16 blocks), but the wave's
// tile is cut into 4 x 6 tiles of 16 x 16 with K = 32 per instruction.  Per k32 step a wave
// issues 24 MFMAs (16 cycles each) and 10 ds_read_b128 — the same LDS traffic per FLOP as
// the 32x32x16 form — and the chip sustains a higher clock on this instruction mix
// (MI355X_MICROARCH.md, "16x16x32 ... 1.12-1.15x the FLOP/s at equal cycles per FLOP").
//
// Lane l = (n = l & 15, q = l >> 4) of tile (ct, j) holds
//   A: weights  [cout = ct*16 + n][k = 8q .. 8q+7]      (chunk q of the k32 block)
//   B: act      [k = 8q .. 8q+7][row = row0(j) + 2n]    (chunk 4*q32 + q of that row's slot)
//   D: acc[ct][j][i] = out[cout = ct*16 + 4q + i][row = row0(j) + 2n]
// The two 16-row tiles of a 32-row pair are INTERLEAVED (even tile = even rows, odd tile =
// odd rows, row0(j) = pair base + (j & 1)).  ds_read_b128 is serviced in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): a group mixes
// columns n of chunk q with the complementary columns of chunk q+1.  With 16 consecutive
// rows per tile and the 17-chunk slot stride, bank group (row + chunk) mod 16 then collides
// once per lane group (measured: SQ_LDS_BANK_CONFLICT = 38 % of the LDS cycles).  With rows
// 2n the even chunk's lanes take the even bank groups and the odd chunk's lanes the odd
// ones: conflict-free.
// i.e. again 4 consecutive output channels of ONE board point per accumulator quad, so
// epilogues keep writing 8-byte fp16 pieces to LDS (16-byte pieces to HBM, see the expand epilogue).
#pragma once
#include <type_traits>
#include "conv_core.h"

namespace p3 {

template <class G, int COUT_PASS>
struct Tiling16 {
  using T = Tiling<G, COUT_PASS>;
  static constexpr int CG = T::CG, LG = T::LG;
  static constexpr int MT = 4;           // cout tiles of 16 per wave
  static constexpr int NT = 2 * T::NT;   // location tiles of 16 per wave
  static constexpr int RS = T::RS;
  static_assert(NT >= MT, "A reads are spread over the first MT location groups");
};

// Padded-grid row of lane column n of location tile j (16 rows) of wave group lg;
// *valid_tile is false for the padding tiles of an uneven split (recomputed, never stored).
template <class G, int COUT_PASS>
__device__ __forceinline__ int tile16_slot0(int lg, int j, bool* valid_tile = nullptr) {
  using T = Tiling16<G, COUT_PASS>;
  int t = lg + (j >> 1) * T::LG;
  const bool ok = t < G::NT_TOTAL;
  if (!ok) t = lg;
  if (valid_tile) *valid_tile = ok;
  const int p = t / G::NT_POS, tt = t - p * G::NT_POS;
  return p * G::PSLOTS + G::PADTOP + tt * 32 + (j & 1);
}

template <int NTn>
__device__ __forceinline__ void acc16_zero(f32x4 (&acc)[4][NTn]) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int j = 0; j < NTn; ++j) acc[ct][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// One conv segment, acc[ct][j] += W_seg x act over NK k32-steps (tap major).  Fragments
// are fetched ONE step ahead: the four A fragments are double-buffered, the B fragment of
// location tile j is reloaded in place right after its four MFMAs have issued, so the read
// has a full step (>= 20 MFMAs) to land.  Issue order per step: after location group 0:
// B_0', A_0', A_1'; after group 1: B_1', A_2', A_3'; after group j >= 2: B_j'.  The counted
// lgkmcnt waits below follow from that order.
//
// Addressing: one VGPR per 32-row tile pair (at kernel row ky, leftmost tap);
// kernel column kx, the odd 16-row tile and the k32 index inside the tap are immediates.
// Body = one kernel row (KW taps), runtime loop over ky.
// SWAP: mfma(act fragment, weight fragment), i.e. the tile transposed — lane (n, q) of tile (ct, j) then holds
//   D: acc[ct][j][i] = out[cout = ct*16 + n][row = row0(j) + 2*(4q + i)]
// four board rows of ONE channel (epilogue_tt16: the channel-major image the broadcast dense contracts over).
template <class G, int COUT_PASS, int KW, int NTAPS_PAD, bool SWAP = false, int NTn = 0>
__device__ __forceinline__ void conv_segment16(Ring<ring_slot_bytes(COUT_PASS, G::KMS), G::NW, G::RD>& ring, char* smem,
                                               f32x4 (&acc)[4][NTn]) {
  using T = Tiling16<G, COUT_PASS>;
  static_assert(NTn == T::NT, "accumulator shape");
  static_assert(G::CB % 32 == 0 && NTAPS_PAD == KW * KW, "k32 steps, unpadded taps");
  constexpr int NT = T::NT, NB = NT / 2;
  constexpr int NQ = G::CB / 32;                   // k32-steps per tap
  constexpr int U = KW * NQ;                       // unrolled body: one kernel row
  constexpr int NOUT = KW;
  constexpr int KM32 = G::KMS / 2;                 // k32 steps per ring macro-step (2, or 1 in the 4-wave kernel)
  static_assert(U % 2 == 0 && U % KM32 == 0 && (KM32 == 1 || KM32 == 2), "segment shape");
  constexpr int KB = COUT_PASS * 32;               // bytes per k16 block of the weight panel
  const int lane = launder(threadIdx.x & 63);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wid % T::CG, lg = wid / T::CG;
  const int n = lane & 15, q = lane >> 4;

  const uint32_t a_off = (uint32_t)((q >> 1) * KB + ((q & 1) * COUT_PASS + cg * 64 + n) * 16);
  uint32_t a_addr = 0;
  uint32_t b_row[NB];                              // byte address for the kernel row being fetched
  h8 fa[2][4], fb[NT];

  // LDS reads of static step V (may be U = first step of the next body) issued after
  // location group J of the step before it; o = body of step 0.
  auto fetch_piece = [&](int o, auto V, auto J) {
    constexpr int v = decltype(V)::value, j = decltype(J)::value;
    constexpr int vv = v % U;
    constexpr int kk = vv % KM32;                  // k32 index inside the ring macro-step
    constexpr int q32 = vv % NQ;                   // k32 index inside the tap
    constexpr int kx = vv / NQ;
    constexpr int nxt = v % 2;                     // U is even: parity of the global step
    if constexpr (j == 0) {
      // mid-segment acquires leave this wave's B_2'..B_(NT-1)' reloads of the previous
      // step in flight: every A read of the macro-step being recycled is older than them
      if constexpr (kk == 0) {
        if constexpr (v == 0) a_addr = ring_acquire<0>(ring, smem) + a_off;
        else a_addr = ring_acquire<NT - 2>(ring, smem) + a_off;
      }
      if constexpr (vv == 0) {
        const int ky = o + v / U;
        const int rowshift = (ky - KW / 2) * G::S - KW / 2;   // tap (ky, kx = 0)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          b_row[b] = (uint32_t)((tile16_slot0<G, COUT_PASS>(lg, 2 * b) + 2 * n + rowshift) * G::SLOTB + q * 16);
      }
    }
    constexpr int BOFF = kx * G::SLOTB + (j & 1) * G::SLOTB + q32 * 64;
    fb[j] = lds_read128<BOFF>(b_row[j >> 1]);
    if constexpr (j < 2) {
      constexpr int AOFF = kk * 2 * KB + j * 512;
      fa[nxt][2 * j] = lds_read128<AOFF>(a_addr);
      fa[nxt][2 * j + 1] = lds_read128<AOFF + 256>(a_addr);
    }
  };

  static_for<0, NT>([&](auto J) { fetch_piece(0, std::integral_constant<int, 0>{}, J); });
#pragma unroll 1
  for (int o = 0; o < NOUT; ++o) {
    const bool last_body = (NOUT == 1) || (o + 1 == NOUT);
    static_for<0, U>([&](auto UU) {
      constexpr int u = decltype(UU)::value;
      const bool do_fetch = (u + 1 < U) || !last_body;
      static_for<0, NT>([&](auto J) {
        constexpr int j = decltype(J)::value;
        // group 0 needs B_0 and the four A fragments (A_3' is the youngest): only
        // B_2'.. of the previous step's fetch may still be outstanding; that covers group 1.
        // Group j >= 2 needs B_j': younger are B_(j+1)'.. plus this step's reads so far
        // (groups 0, 1: three each, groups 2..j-1: one each) = NT + 3, or NT - 1 - j
        // when this step fetches nothing.
        if constexpr (j == 0) wait_lgkm<NT - 2>();
        else if constexpr (j >= 2) {
          if (do_fetch) wait_lgkm<NT + 3>();
          else wait_lgkm<NT - 1 - j>();
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[ct][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[u % 2][ct], acc[ct][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[u % 2][ct], fb[j], acc[ct][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (do_fetch) fetch_piece(o, std::integral_constant<int, u + 1>{}, J);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
}

// The 3x3 segment with the activation fragments of a kernel row's three taps shared.
//
// Rows of a tile pair are interleaved (even tile = rows base + 2n, odd tile = base + 2n + 1), so for
// kernel row ky and k32 index q32 the six (tile, kx) operands of a pair are only FOUR different
// fragments R[s] = slots base + 2n + s, s = 0..3 (s = kx + parity): the even tile's fragment at
// kx + 1 is the odd tile's at kx.  Steps therefore run (ky, q32, kx) instead of (ky, kx, q32) —
// the fused blocks' 3x3 weights are packed in that order (engine.cpp pack_segment_3x3) — and a
// group of three steps (72 MFMAs) loads 12 activation + 12 weight fragments instead of 18 + 12:
// a fifth fewer LDS reads per MFMA.  The kernel is power-limited: what this
// buys is energy, i.e. clock.
//
// Schedule.  Fragments of group g + 1 are loaded during group g, each into the register its
// predecessor just vacated: R[.][0] is dead after step kx = 0 (even tile), R[.][1] after kx = 1,
// R[.][2] and R[.][3] after kx = 2; every reload has two or more steps to land.  The weight
// fragments of step u + 1 are loaded during step u after location groups 0 and 1 (double
// buffer), as in conv_segment16.  One counted wait per step, at its start: everything but the
// activation reloads issued after the previous step's last weight read must have landed — the
// fragments a step uses are all older than that.
template <class G, int COUT_PASS, int NTn = 0>
__device__ __forceinline__ void conv_segment16_3x3(Ring<ring_slot_bytes(COUT_PASS, G::KMS), G::NW, G::RD>& ring, char* smem,
                                                   f32x4 (&acc)[4][NTn]) {
  using T = Tiling16<G, COUT_PASS>;
  static_assert(NTn == T::NT && T::NT == 6 && G::CB % 32 == 0, "shape");
  constexpr int NB = 3;
  constexpr int NQ = G::CB / 32;                   // k32 steps per tap
  constexpr int U = 3 * NQ;                        // one kernel row: NQ groups of three steps
  constexpr int NOUT = 3;
  constexpr int KM32 = G::KMS / 2;
  static_assert(U % 2 == 0 && U % KM32 == 0, "segment shape");
  constexpr int KB = COUT_PASS * 32;
  const int lane = launder(threadIdx.x & 63);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wid % T::CG, lg = wid / T::CG;
  const int n = lane & 15, q = lane >> 4;
  const uint32_t a_off = (uint32_t)((q >> 1) * KB + ((q & 1) * COUT_PASS + cg * 64 + n) * 16);
  uint32_t a_addr = 0;
  uint32_t b_row[NB];
  h8 fa[2][4], fr[NB][4];

  auto set_rows = [&](int ky) {   // leftmost tap of kernel row ky, even tile of each pair
    const int rowshift = (ky - 1) * G::S - 1;
#pragma unroll
    for (int b = 0; b < NB; ++b)
      b_row[b] = (uint32_t)((tile16_slot0<G, COUT_PASS>(lg, 2 * b) + 2 * n + rowshift) * G::SLOTB + q * 16);
  };
  // weight fragments 2*JJ, 2*JJ+1 of static step V (V may be U = step 0 of the next body)
  auto a_fetch = [&](auto V, auto JJ) {
    constexpr int v = decltype(V)::value, jj = decltype(JJ)::value;
    constexpr int kk = (v % U) % KM32;
    constexpr int nxt = v % 2;
    // every weight read of the slot being recycled landed before this step's MFMAs began (the wait
    // at the step's start); up to five younger activation reloads may stay in flight
    if constexpr (jj == 0 && kk == 0) a_addr = ring_acquire<5>(ring, smem) + a_off;
    constexpr int AOFF = kk * 2 * KB + jj * 512;
    fa[nxt][2 * jj] = lds_read128<AOFF>(a_addr);
    fa[nxt][2 * jj + 1] = lds_read128<AOFF + 256>(a_addr);
  };
  // activation fragment R[B][S] of the group with k32 index Q32 (rows set by set_rows)
  auto b_fetch = [&](auto B, auto S, auto Q32) {
    constexpr int b = decltype(B)::value, sidx = decltype(S)::value, q32 = decltype(Q32)::value;
    fr[b][sidx] = lds_read128<sidx * G::SLOTB + q32 * 64>(b_row[b]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // prologue: the first acquire is also the "previous layer written" barrier — reads come after it
  set_rows(0);
  a_addr = ring_acquire<0>(ring, smem) + a_off;
  {
    constexpr int AOFF = 0;
    fa[0][0] = lds_read128<AOFF>(a_addr);
    fa[0][1] = lds_read128<AOFF + 256>(a_addr);
    fa[0][2] = lds_read128<AOFF + 512>(a_addr);
    fa[0][3] = lds_read128<AOFF + 768>(a_addr);
  }
  static_for<0, NB>([&](auto B) { static_for<0, 4>([&](auto S) { b_fetch(B, S, I0{}); }); });

#pragma unroll 1
  for (int o = 0; o < NOUT; ++o) {
    const bool last_body = o + 1 == NOUT;
    static_for<0, U>([&](auto UU) {
      constexpr int u = decltype(UU)::value;
      constexpr int q32 = u / 3, kx = u % 3;
      constexpr bool last_group = q32 == NQ - 1;
      const bool a_next = (u + 1 < U) || !last_body;            // a step follows: load its weight fragments
      const bool b_next = !(last_group && last_body);           // a group follows: reload this group's fragments
      // what may still be in flight: the reloads the previous step issued after its last weight read
      if constexpr (u == 0) {
        if (o == 0) wait_lgkm<0>(); else wait_lgkm<4>();
      } else {
        constexpr bool prev_last_group = (u - 1) / 3 == NQ - 1;
        constexpr int n_prev = ((u - 1) % 3 == 2) ? 4 : 2;
        if (prev_last_group && last_body) wait_lgkm<0>(); else wait_lgkm<n_prev>();
      }
      // the reloads of a body's last group belong to the next kernel row
      if constexpr (last_group && kx == 0) { if (b_next) set_rows(o + 1); }
      using QN = std::integral_constant<int, (q32 + 1) % NQ>;
      static_for<0, 6>([&](auto J) {
        constexpr int j = decltype(J)::value;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[u % 2][ct], fr[j >> 1][kx + (j & 1)], acc[ct][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        using B = std::integral_constant<int, j / 2>;
        if constexpr (j % 2 == 0) {              // the even tile was the last user of R[b][kx]
          if constexpr (j == 0) {
            if (b_next) b_fetch(B{}, std::integral_constant<int, kx>{}, QN{});
            if (a_next) a_fetch(std::integral_constant<int, u + 1>{}, I0{});
          } else {
            if (b_next) b_fetch(B{}, std::integral_constant<int, kx>{}, QN{});
          }
        } else {
          if constexpr (kx == 2) { if (b_next) b_fetch(B{}, std::integral_constant<int, 3>{}, QN{}); }   // the odd tile of the last step: R[b][3]
          if constexpr (j == 1) { if (a_next) a_fetch(std::integral_constant<int, u + 1>{}, I1{}); }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
}

// ---- epilogues ---------------------------------------------------------------------------
struct EpiParams16 { f32x4 sc[4], sh[4]; };

template <class G, int COUT_PASS>
__device__ __forceinline__ void epi_params16(EpiParams16& ep, const float* __restrict__ scale,
                                             const float* __restrict__ shift, int cofs) {
  const int q = launder(threadIdx.x & 63) >> 4;
  const int c0 = cofs + cg_of<G, COUT_PASS>() * 64 + q * 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    ep.sc[ct] = *(const f32x4*)(scale + c0 + 16 * ct);
    ep.sh[ct] = *(const f32x4*)(shift + c0 + 16 * ct);
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {   // times log2(e): bn_mish4_l2 / bn_mish8_l2
    ep.sc[ct] = scale_log2e(ep.sc[ct]);
    ep.sh[ct] = scale_log2e(ep.sh[ct]);
  }
}

template <int NTn>
struct EpiOut16 { h4 o[NTn][4]; };

// BN + mish + fp16 pack in registers (may run before the barrier that frees the act buffer).
template <int NTn>
__device__ __forceinline__ void epilogue_math16(EpiOut16<NTn>& eo, f32x4 (&acc)[4][NTn],
                                                const EpiParams16& ep) {
#ifdef P3_EPI_CHAINS   // A/B: one dependent chain per pair of values
#pragma unroll
  for (int j = 0; j < NTn; ++j) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) eo.o[j][ct] = bn_mish4_l2(acc[ct][j], ep.sc[ct], ep.sh[ct]);
    __builtin_amdgcn_sched_barrier(0);   // one tile at a time: bounds the live temporaries
  }
#else
#pragma unroll
  for (int j = 0; j < NTn; ++j) {
#pragma unroll
    for (int ct = 0; ct < 4; ct += 2)   // eight values in flight per stage
      bn_mish8_l2(acc[ct][j], acc[ct + 1][j], ep.sc[ct], ep.sh[ct], ep.sc[ct + 1], ep.sh[ct + 1], eo.o[j][ct], eo.o[j][ct + 1]);
  }
#endif
}

__device__ __forceinline__ void half_swap(h4& x, h4& y);   // defined with the HBM epilogues below

// 16-byte LDS stores: ds_write_b64 is banked over 32 banks in groups of 16 consecutive lanes, and a tile's lanes
// sit 2 rows = 136 dwords apart, so the 8-byte pieces of one instruction fall on 4 banks 4 ways (13.3 LDS cycles
// per wave-instruction against 5.4 conflict-free, tools/probe/lds_bank_probe.hip; SQ_LDS_BANK_CONFLICT 29 % of the
// kernel's LDS cycles before).  One v_permlane16_swap per dword (half_swap, as on the way to HBM) turns "my half of
// two tiles' pieces" into "the whole piece of one row": half as many stores, ds_write_b128 in groups of 8 lanes, 2-way.
// PIECES: eo already holds (low half, high half) of this lane's row's piece — what a 16-byte global load
// of the piece layout delivers (stash16) — and goes out as it is.
template <class G, int COUT_PASS, int NTn, bool PIECES = false>
__device__ __forceinline__ void epilogue_write16(char* smem, const EpiOut16<NTn>& eo, int cofs) {
  using T = Tiling16<G, COUT_PASS>;
  const int lane = launder(threadIdx.x & 63);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = wid / T::CG;
  const int n = lane & 15, q = lane >> 4;
  const uint32_t coff = (uint32_t)(((cofs >> 3) + cg_of<G, COUT_PASS>() * 8 + (q >> 1)) * 16);
#pragma unroll
  for (int b = 0; b < NTn / 2; ++b) {
    const int t = lg + b * T::LG;
    const int tv = t < G::NT_TOTAL ? t : lg;
    const int p = tv / G::NT_POS, tt = tv - p * G::NT_POS;
    const int r = tt * 32 + 2 * n + (q & 1);        // this lane's row after the swap: the even tile's (q even) or the odd tile's
    int loc;
    const bool ok = row_valid<G::S>(r, loc) && (t < G::NT_TOTAL);
    const uint32_t dst = (uint32_t)((p * G::PSLOTS + G::PADTOP + r) * G::SLOTB) + coff;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      h4 lo = eo.o[2 * b][ct], hi = eo.o[2 * b + 1][ct];
      if (!PIECES) half_swap(lo, hi);   // every lane takes part: partners of invalid rows may be valid
      if (ok) *(h8*)(smem + dst + ct * 32) = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  }
}

// Full layer transition: parameters, math, barrier, stores.
template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void epilogue_layer16(char* smem, f32x4 (&acc)[4][NTn],
                                                 const float* __restrict__ scale,
                                                 const float* __restrict__ shift) {
  EpiOut16<NTn> eo;
  {
    EpiParams16 ep;
    epi_params16<G, COUT_PASS>(ep, scale, shift, 0);
    epilogue_math16<NTn>(eo, acc, ep);
  }
  // the lane pairing of the 16-byte stores before the barrier, under the other waves' last K steps
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) half_swap(eo.o[2 * b][ct], eo.o[2 * b + 1][ct]);
  lds_barrier();
  epilogue_write16<G, COUT_PASS, NTn, true>(smem, eo, 0);
}

// ---- expand epilogue: out = acc + residual -> fp16 global [pos][C/8][361][8] ---------------
// A lane's accumulator quad is one 8-byte HALF of a 16-byte [8 channels] piece; the other half
// of the same board point sits in the partner lane 16 lanes away (q ^ 1).  Lanes of even q keep
// the low halves of tiles 2b and 2b+1, lanes of odd q the high halves, so one
// v_permlane16_swap per dword turns "my half of two tiles" into "the whole piece of ONE tile":
// tile 2b (rows base + 2n) in the even-q lanes, tile 2b+1 (rows base + 2n + 1) in the odd-q
// lanes.  Residual loads and output stores then move 16 bytes per lane, 12 instead of 24
// instructions per pass, and a wave instruction covers two contiguous 512-byte runs.
template <int NTn>
struct ResRegs16 {
  static constexpr int NB = NTn / 2;
  h8 rv[NB][4];         // the 16-byte piece of this lane's own row of tile pair b, cout tile ct
  uint32_t base[NB];    // element offset of that piece for ct = 0
  bool ok[NB];
};

template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void residual_addr16(ResRegs16<NTn>& rr, int C, int pos0, int npos, int cofs) {
  using T = Tiling16<G, COUT_PASS>;
  const int lane = launder(threadIdx.x & 63);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = wid / T::CG;
  const int n = lane & 15, q = lane >> 4;
  const int cblk = ((cofs + cg_of<G, COUT_PASS>() * 64) >> 3) + (q >> 1);   // channel block of ct = 0
#pragma unroll
  for (int b = 0; b < NTn / 2; ++b) {
    const int t = lg + b * T::LG;
    const int tv = t < G::NT_TOTAL ? t : lg;
    const int p = tv / G::NT_POS, tt = tv - p * G::NT_POS;
    int loc;
    rr.ok[b] = row_valid<G::S>(tt * 32 + 2 * n + (q & 1), loc) && (pos0 + p < npos) && (t < G::NT_TOTAL);
    if (!rr.ok[b]) loc = 0;
    const int pp = (pos0 + p < npos) ? pos0 + p : npos - 1;
    rr.base[b] = (uint32_t)((pp * (C / 8) + cblk) * (kNLoc * 8) + loc * 8);
  }
}

// Issues exactly NTn*2 = 12 sixteen-byte loads per lane (invalid rows read a valid dummy
// address).  Addresses are (uniform base of the cout tile's channel block) + (32-bit lane
// offset), the SGPR-base form of global_load: three offset registers.
constexpr int kResLoads = 12;
template <int NTn>
__device__ __forceinline__ void residual_load16(ResRegs16<NTn>& rr, const _Float16* __restrict__ x) {
  static_assert(NTn * 2 == kResLoads, "vmcnt bookkeeping of the callers");
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const char* xc = (const char*)x + (size_t)ct * (2 * kNLoc * 8 * 2);   // channel block +2 per cout tile
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      rr.rv[b][ct] = *(const h8*)(xc + (uint32_t)(rr.base[b] * 2u));
    }
  }
}

// (x = my value for tile 2b, y = my value for tile 2b+1)  <->  (low half, high half) of my row's
// piece; the same swap converts either way.
__device__ __forceinline__ void half_swap(h4& x, h4& y) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  u2 xu = __builtin_bit_cast(u2, x), yu = __builtin_bit_cast(u2, y);
  auto s0 = __builtin_amdgcn_permlane16_swap(xu[0], yu[0], false, false);
  auto s1 = __builtin_amdgcn_permlane16_swap(xu[1], yu[1], false, false);
  xu[0] = s0[0]; yu[0] = s0[1];
  xu[1] = s1[0]; yu[1] = s1[1];
  x = __builtin_bit_cast(h4, xu);
  y = __builtin_bit_cast(h4, yu);
}

// this lane's residual values for tiles 2b (r0) and 2b+1 (r1) of cout tile ct
template <int NTn>
__device__ __forceinline__ void residual_unpack16(const ResRegs16<NTn>& rr, int b, int ct, h4& r0, h4& r1) {
  const h8 v = rr.rv[b][ct];
  r0 = h4{v[0], v[1], v[2], v[3]};
  r1 = h4{v[4], v[5], v[6], v[7]};
  half_swap(r0, r1);
}

template <int NTn>
__device__ __forceinline__ void residual_add16(f32x4 (&acc)[4][NTn], const ResRegs16<NTn>& rr) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      h4 r0, r1;
      residual_unpack16<NTn>(rr, b, ct, r0, r1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[ct][2 * b][i] += (float)r0[i];
        acc[ct][2 * b + 1][i] += (float)r1[i];
      }
    }
}

// Reduce input of a block from x fetched in the accumulator-quad layout (residual_load16 of
// channel half `cofs`): A = mish(bn0(x)) as packed fp16, ready for epilogue_write16.
template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void activate_loaded16(EpiOut16<NTn>& A, const ResRegs16<NTn>& xin,
                                                  const float* __restrict__ scale,
                                                  const float* __restrict__ shift, int cofs) {
  const int q = launder(threadIdx.x & 63) >> 4;
  const int c0 = cofs + cg_of<G, COUT_PASS>() * 64 + q * 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const f32x4 sc = scale_log2e(*(const f32x4*)(scale + c0 + 16 * ct)), sh = scale_log2e(*(const f32x4*)(shift + c0 + 16 * ct));
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      h4 r0, r1;
      residual_unpack16<NTn>(xin, b, ct, r0, r1);
      bn_mish8_l2(f32x4{(float)r0[0], (float)r0[1], (float)r0[2], (float)r0[3]}, f32x4{(float)r1[0], (float)r1[1], (float)r1[2], (float)r1[3]},
                  sc, sh, sc, sh, A.o[2 * b][ct], A.o[2 * b + 1][ct]);
    }
    __builtin_amdgcn_sched_barrier(0);   // one cout tile at a time: bounds the live temporaries
  }
}

// The same in two steps for a half that must wait in registers across a K loop: stash16 parks
// the fetched pieces in A's registers as they are (a renaming: A and the pieces are never live
// together, so the block kernel carries ONE 48-register value across its first K slice whether
// it came from HBM or from the previous block's expand), activate_stashed16 turns them into A.
template <int NTn>
__device__ __forceinline__ void stash16(EpiOut16<NTn>& A, const ResRegs16<NTn>& xin) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      const h8 v = xin.rv[b][ct];
      A.o[2 * b][ct] = h4{v[0], v[1], v[2], v[3]};
      A.o[2 * b + 1][ct] = h4{v[4], v[5], v[6], v[7]};
    }
}
template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void activate_stashed16(EpiOut16<NTn>& A, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, int cofs) {
  const int q = launder(threadIdx.x & 63) >> 4;
  const int c0 = cofs + cg_of<G, COUT_PASS>() * 64 + q * 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const f32x4 sc = scale_log2e(*(const f32x4*)(scale + c0 + 16 * ct)), sh = scale_log2e(*(const f32x4*)(shift + c0 + 16 * ct));
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      h4 r0 = A.o[2 * b][ct], r1 = A.o[2 * b + 1][ct];
      half_swap(r0, r1);
      bn_mish8_l2(f32x4{(float)r0[0], (float)r0[1], (float)r0[2], (float)r0[3]}, f32x4{(float)r1[0], (float)r1[1], (float)r1[2], (float)r1[3]},
                  sc, sh, sc, sh, A.o[2 * b][ct], A.o[2 * b + 1][ct]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// A half fetched in the piece layout that goes into the act buffer as it is (the input of a
// fused broadcast conv_last is already activated by k_bdense): piece halves -> accumulator quads.
template <int NTn>
__device__ __forceinline__ void unstash16(EpiOut16<NTn>& A) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) half_swap(A.o[2 * b][ct], A.o[2 * b + 1][ct]);
}

// Expand epilogue of the position-major block kernel: x' = acc + residual -> fp16 -> HBM, and,
// when `act`, A = mish(bn0_next(x')) of the same fp16 values for the next block's reduce (cofs =
// channel half of this pass; scale/shift = the next block's folded bn0).
template <class G, int COUT_PASS, int NTn>
__device__ __forceinline__ void epilogue_store16_act(f32x4 (&acc)[4][NTn], const ResRegs16<NTn>& rr,
                                                     _Float16* __restrict__ x, EpiOut16<NTn>& A, bool act,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int cofs) {
  const int q = launder(threadIdx.x & 63) >> 4;
  const int c0 = cofs + cg_of<G, COUT_PASS>() * 64 + q * 4;
  // the BN parameters of cout tile ct + 1 are requested while tile ct is worked on: the scheduling fence at the
  // end of an iteration otherwise leaves each tile's two L2 loads exposed
  // (4-wave kernels only: in the 8-wave C = 256 kernels the eight extra registers spill)
  constexpr bool PF = G::NW == 4;
  f32x4 sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0}, nsc = {0, 0, 0, 0}, nsh = {0, 0, 0, 0};
  if (PF && act) {
    sc = *(const f32x4*)(scale + c0);
    sh = *(const f32x4*)(shift + c0);
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    char* xc = (char*)x + (size_t)ct * (2 * kNLoc * 8 * 2);
    if (PF) {
      if (act && ct + 1 < 4) {
        nsc = *(const f32x4*)(scale + c0 + 16 * (ct + 1));
        nsh = *(const f32x4*)(shift + c0 + 16 * (ct + 1));
      }
    } else if (act) {
      sc = *(const f32x4*)(scale + c0 + 16 * ct);
      sh = *(const f32x4*)(shift + c0 + 16 * ct);
    }
    const f32x4 scl = scale_log2e(sc), shl = scale_log2e(sh);   // bn_mish8_l2 takes them times log2(e)
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      h4 r0, r1;
      residual_unpack16<NTn>(rr, b, ct, r0, r1);
      h4 o0, o1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o0[i] = (_Float16)(acc[ct][2 * b][i] + (float)r0[i]);
        o1[i] = (_Float16)(acc[ct][2 * b + 1][i] + (float)r1[i]);
      }
      if (act) {
        bn_mish8_l2(f32x4{(float)o0[0], (float)o0[1], (float)o0[2], (float)o0[3]}, f32x4{(float)o1[0], (float)o1[1], (float)o1[2], (float)o1[3]},
                    scl, shl, scl, shl, A.o[2 * b][ct], A.o[2 * b + 1][ct]);
      } else {   // defined on every path, or the previous block's A stays live through this one
        A.o[2 * b][ct] = h4{0, 0, 0, 0};
        A.o[2 * b + 1][ct] = h4{0, 0, 0, 0};
      }
      half_swap(o0, o1);   // every lane takes part: partners of invalid rows may be valid
      const h8 piece = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
      if (rr.ok[b]) *(h8*)(xc + (uint32_t)(rr.base[b] * 2u)) = piece;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PF) {
      sc = nsc;
      sh = nsh;
    }
  }
}

// out = mish(acc) -> fp16 global (the broadcast block's conv_first + its activation, fused into
// the tail of a block launch; same arithmetic as k_conv1x1's EPI 0).
template <int NTn>
__device__ __forceinline__ void epilogue_store16_mish(f32x4 (&acc)[4][NTn], const ResRegs16<NTn>& rr,
                                                      _Float16* __restrict__ out) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    char* oc = (char*)out + (size_t)ct * (2 * kNLoc * 8 * 2);
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      const f32x2 a0 = mish_f2(f32x2{acc[ct][2 * b][0], acc[ct][2 * b][1]}), a1 = mish_f2(f32x2{acc[ct][2 * b][2], acc[ct][2 * b][3]});
      const f32x2 b0 = mish_f2(f32x2{acc[ct][2 * b + 1][0], acc[ct][2 * b + 1][1]}), b1 = mish_f2(f32x2{acc[ct][2 * b + 1][2], acc[ct][2 * b + 1][3]});
      h4 o0 = {(_Float16)a0[0], (_Float16)a0[1], (_Float16)a1[0], (_Float16)a1[1]};
      h4 o1 = {(_Float16)b0[0], (_Float16)b0[1], (_Float16)b1[0], (_Float16)b1[1]};
      half_swap(o0, o1);   // every lane takes part
      const h8 piece = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
      if (rr.ok[b]) *(h8*)(oc + (uint32_t)(rr.base[b] * 2u)) = piece;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same conv_first, taken with SWAP accumulators, when the broadcast dense follows in the same launch:
// t = mish(acc) -> fp16 -> Tt[channel of this pass][padded board row], TT_STRIDE bytes per channel (k_bdense's LDS
// image, K = the 384 padded rows of the act buffer instead of 361 board points + 23 zeros; the dense matrix is
// packed to match, engine.cpp).  A lane's two tiles of a pair interleave to 8 consecutive rows: one 16-byte store.
// Rows off the board come out as mish(0) = 0: the act buffer's halo slots are zero and the conv is 1x1.
template <class G, int COUT_PASS, int TT_STRIDE, int NTn>
__device__ __forceinline__ void epilogue_tt16(char* smem, f32x4 (&acc)[4][NTn]) {
  using T = Tiling16<G, COUT_PASS>;
  static_assert(G::NPOS == 1 && G::NT_TOTAL == T::LG * (NTn / 2), "one position, every tile pair real");
  const int lane = launder(threadIdx.x & 63);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wid % T::CG, lg = wid / T::CG;
  const int n = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      const f32x2 a0 = mish_f2(f32x2{acc[ct][2 * b][0], acc[ct][2 * b][1]}), a1 = mish_f2(f32x2{acc[ct][2 * b][2], acc[ct][2 * b][3]});
      const f32x2 b0 = mish_f2(f32x2{acc[ct][2 * b + 1][0], acc[ct][2 * b + 1][1]}), b1 = mish_f2(f32x2{acc[ct][2 * b + 1][2], acc[ct][2 * b + 1][3]});
      const h8 v = {(_Float16)a0[0], (_Float16)b0[0], (_Float16)a0[1], (_Float16)b0[1],
                    (_Float16)a1[0], (_Float16)b1[0], (_Float16)a1[1], (_Float16)b1[1]};
      const int t = lg + b * T::LG;
      *(h8*)(smem + (uint32_t)((cg * 64 + ct * 16 + n) * TT_STRIDE + (t * 32 + 8 * q) * 2)) = v;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool RESIDUAL, int NTn>
__device__ __forceinline__ void epilogue_store16(f32x4 (&acc)[4][NTn], const ResRegs16<NTn>& rr,
                                                 _Float16* __restrict__ x) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    char* xc = (char*)x + (size_t)ct * (2 * kNLoc * 8 * 2);
#pragma unroll
    for (int b = 0; b < NTn / 2; ++b) {
      h4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0};
      if (RESIDUAL) residual_unpack16<NTn>(rr, b, ct, r0, r1);
      h4 o0, o1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v0 = acc[ct][2 * b][i], v1 = acc[ct][2 * b + 1][i];
        if (RESIDUAL) { v0 += (float)r0[i]; v1 += (float)r1[i]; }
        o0[i] = (_Float16)v0;
        o1[i] = (_Float16)v1;
      }
      half_swap(o0, o1);   // every lane takes part: partners of invalid rows may be valid
      const h8 piece = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
      if (rr.ok[b]) *(h8*)(xc + (uint32_t)(rr.base[b] * 2u)) = piece;
    }
  }
}

}  // namespace p3
