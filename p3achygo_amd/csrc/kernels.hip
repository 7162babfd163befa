// kernels.hip — gfx950 kernels of the policy/value net forward pass.
//
// Arithmetic spec: reference python/model.py (cited per kernel).  Layouts:
//   residual stream / activations in HBM : x[pos][C/8][361][8]  fp16 (channel-blocked)
//   head pre-activations                  : hp[pos][24][361][4]  fp32 (channel quads)
//   outputs                               : out[pos][kOutStride] fp32 (see kernels.h)
// All trunk convs run through conv_core.h (LDS-resident activations, MFMA implicit GEMM,
// glds weight ring).  One workgroup = 512 threads; grid-stride loop over positions.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <atomic>

#include "conv_core.h"
#include "conv16.h"

namespace p3 {

// =======================================================================================
// Fused residual blocks.  KIND 0 = bottleneck (btl), 1 = nested bottleneck (nbt).
//   btl: BottleneckResidualConvBlock, model.py:372-425 —
//        x + conv1x1_{L+1}( ... conv3x3_j( ... conv1x1_0(x)))   each conv = conv(mish(bn(.)))
//   nbt: NbtResidualBlock, model.py:430-486 —
//        t = conv1x1_0(x); t += conv3(conv3(t)); t += conv3(conv3(t)); x + conv1x1_5(t)
// Weight stream order (must match pack_block_stream in engine.cpp):
//   reduce: for each CB-slice of C input channels; inner 3x3 convs in order; expand: for
//   each CB-slice of C output channels.
// =======================================================================================
// Loop order: POSITION major.  A workgroup takes a position (two at C = 128) through every block
// of the launch before it moves on, so between two blocks of a launch the residual stream never
// comes back from HBM as a conv operand: the expand epilogue already holds x' = x + conv(..) in
// the accumulator-quad layout the act buffer is written in (4 consecutive channels of one board
// point per lane), applies the NEXT block's bn0 + mish right there and keeps the two activated
// 128-channel halves as packed fp16 (A0, A1) until the act buffer is free.  x' still goes to HBM
// once per block — it is the next block's residual (read back by the lane that stored it, an L2
// hit) and the launch's output.  The first block of a position takes the same path with A0/A1
// made from x fetched in that same layout.  The blocks' weight streams are packed back to back
// (engine.cpp), so the ring walks one circular stream per launch and never drains.
//
// The values are the ones the block-major order produced: bn0 + mish is applied to the fp16
// value that is stored, exactly what a re-load would return (bit-identical to one launch per
// block, tests/test_engine_gpu.py::test_fused_block_launches_equal_one_launch_per_block).
//
// NW = 8: one 512-thread workgroup per CU (C = 256: one position; C = 128: two positions side by
// side in the act buffer).  NW = 4 (C = 128 only): a 256-thread workgroup per position with
// K = 32 ring steps — 60.6 KB of activations + 12 KB of ring, so TWO workgroups share a CU and one
// workgroup's BN + mish / store phases run under the other's MFMA phases.
//
// BC form: the 1x1 convs of the broadcast blocks on either side of the run are taken into the launch
// (BroadcastResidualBlock, model.py:570-606: x + conv_last(mish(bn1(dense(mish(conv_first(mish(bn0(x)))))))),
// k_bdense stays its own launch between two block launches):
//   head: x += W_last . z, z = k_bdense's output (already activated).  A C -> C conv is two output
//         passes of the expand's shape over two K slices staged from HBM in the accumulator-quad
//         layout; pass 1's epilogue leaves A1 = mish(bn0(x')) for the first block, A0 is re-made from
//         the half pass 0 just stored (holding it across pass 1 would not fit the registers).
//   tail: t = mish(W_first . mish(bn0_b(x'))): two output passes of the reduce's shape.  Pass 0 takes
//         A0 / A1 from the last block's expand epilogue like any next block; pass 1 re-reads x' as a
//         position's first block does.
// Each removes a launch that moved x through HBM twice (k_conv1x1) and the first block's own read.
// ---- broadcast dense, shared by k_bdense and the fused tail of k_block -------------------------------
constexpr int kTtStride = 784;  // bytes per channel row in LDS: 384 fp16 + 16 B pad
constexpr int kTtChannels = 128;                          // channels resident per pass
constexpr uint32_t kTtBytes = kTtChannels * kTtStride;    // 100,352

// Tt seen as a conv act buffer: one slot per channel (128 per pass), K = 384 board points.
struct GeoTt {
  static constexpr int NW = 8, KMS = kKMS, RD = kRingDepth;
  static constexpr int NPOS = 1, CB = 384, NCH = 48, SLOTB = kTtStride, PAD = 0, S = 1, NROWS = 128,
                       NT_POS = 4, PADTOP = 0, PSLOTS = 128, ACT_BYTES = 128 * kTtStride, NT_TOTAL = 4;
};

// u[c][j] = mish(bn1(sum_i Tt[c][i] W[i][j] + b[j])) for the 128 channels in Tt (channel half `half` of position
// `pos`), three passes of 128 dense columns j streamed through the ring, -> HBM in the piece layout.
// p_bias [384], p_scale / p_shift [C]: LDS.  nch = channels of this pass that are real.
template <int C>
__device__ __forceinline__ void bdense_passes(Ring<16384>& ring, char* smem, const float* p_bias, const float* p_scale,
                                              const float* p_shift, _Float16* __restrict__ u, int pos, int half, int nch) {
  constexpr int CH = kTtChannels;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int jg = wid & 1;        // which 64 of the 128 j rows in this pass
  const int ct = wid >> 1;       // channel tile (32 channels) 0..3
  const int lr = lane & 31, h = lane >> 5;
  const bool ct_active = ct * 32 < nch;
#pragma unroll 1
  for (int jp = 0; jp < 3; ++jp) {
    // D[c][j] = sum_i Tt[c][i] * W[i][j]: the conv K loop with its operands swapped —
    // "act buffer" = Tt (slot = channel, 48 chunks of 8 board points), "weights" = the
    // 128 dense columns of this pass streamed through the ring; fragments are prefetched
    // two k16 steps ahead exactly as in the conv kernels.
    f32x16 acc2[2][1];
    acc_zero<GeoTt, 128>(acc2);
    conv_segment<GeoTt, 128, 1, 1, true>(ring, smem, acc2);
    if (!ct_active) continue;
    const f32x16 acc[2] = {acc2[0][0], acc2[1][0]};
    // epilogue: rows = channel (regs), cols = j (lanes)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int j = jp * 128 + jg * 64 + mt * 32 + lr;
      if (j >= kNLoc) continue;
      const float bj = p_bias[j];
      // channel blocks g4 = 2*gp and 2*gp + 1: each lane's quad is one 8-byte half of a
      // block's piece; after the swap lanes 0-31 hold the whole piece of block 2*gp, lanes
      // 32-63 that of block 2*gp + 1 (see epilogue_store in conv_core.h)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        h4 o[2];
        {
          const int g4 = 2 * gp;
          const int c = half * CH + ct * 32 + g4 * 8 + h * 4;
          const f32x4 sc0 = scale_log2e(*(const f32x4*)(p_scale + c)), sh0 = scale_log2e(*(const f32x4*)(p_shift + c));
          const f32x4 sc1 = scale_log2e(*(const f32x4*)(p_scale + c + 8)), sh1 = scale_log2e(*(const f32x4*)(p_shift + c + 8));
          const f32x4 v0 = {acc[mt][g4 * 4] + bj, acc[mt][g4 * 4 + 1] + bj, acc[mt][g4 * 4 + 2] + bj, acc[mt][g4 * 4 + 3] + bj};
          const f32x4 v1 = {acc[mt][g4 * 4 + 4] + bj, acc[mt][g4 * 4 + 5] + bj, acc[mt][g4 * 4 + 6] + bj, acc[mt][g4 * 4 + 7] + bj};
          bn_mish8_l2(v0, v1, sc0, sh0, sc1, sh1, o[0], o[1]);
        }
        half_swap32(o[0], o[1]);
        const h8 piece = {o[0][0], o[0][1], o[0][2], o[0][3], o[1][0], o[1][1], o[1][2], o[1][3]};
        const int cb = (half * CH + ct * 32) / 8 + 2 * gp + h;   // this lane's channel block
        *(h8*)(u + ((size_t)pos * (C / 8) + cb) * (kNLoc * 8) + j * 8) = piece;
      }
    }
  }
}

// dense bias per board point and folded bn1 per channel into LDS at `dst` ([384] + [C] + [C] floats)
template <int C>
__device__ __forceinline__ void bdense_stage_params(float* dst, const float* __restrict__ bias,
                                                    const float* __restrict__ scale, const float* __restrict__ shift) {
  for (int i = threadIdx.x; i < 384; i += kWG) dst[i] = i < kNLoc ? bias[i] : 0.0f;
  for (int i = threadIdx.x; i < C; i += kWG) { dst[384 + i] = scale[i]; dst[384 + C + i] = shift[i]; }
}

#ifdef P3_DIAG
#define P3_STAMP(section, k)                                                                              \
  do {                                                                                                    \
    if (a.stamps && blockIdx.x < kStampWgs && npos_done == 1 && run == a.stamp_run && (section) < kStampSections && (threadIdx.x & 63) == 0) \
      a.stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * kStampSections + (section)) * kStampSlots + (k)] = \
          __builtin_amdgcn_s_memtime();                                                                   \
  } while (0)
#define P3_SPAN(k)                                                                        \
  do {                                                                                    \
    if (a.spans && blockIdx.x < kSpanWgs && threadIdx.x == 0 && (k) < 8) {                \
      a.spans[blockIdx.x * kSpanSlots + (k)] = __builtin_amdgcn_s_memtime();              \
      a.spans[blockIdx.x * kSpanSlots + 8 + (k)] = __builtin_amdgcn_s_memrealtime();      \
    }                                                                                     \
  } while (0)
#else
#define P3_STAMP(section, k) do {} while (0)
#define P3_SPAN(k) do {} while (0)
#endif

template <int C, int CB, int KIND, int L, int NW = 8, bool BC = false>
__global__ void __launch_bounds__(NW * 64, 2) k_block(BlockArgs a) {
  static_assert(NW == 8 || (NW == 4 && CB == 64), "4-wave workgroups: one 64-channel position each");
  constexpr int NPOS = NW == 8 ? 128 / CB : 1;
  using G = Geo<NPOS, CB, 3, NW, NW == 8 ? kKMS : 2, NW == 8 ? kRingDepth : kRingDepth4>;
  using T = Tiling16<G, CB>;
  constexpr int NT = T::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr uint32_t kRingOff = G::ACT_BYTES;
  static_assert(C / CB == 2, "two input slices / two output passes");

  P3_SPAN(0);
  act_zero<G>(smem);
  if (BC && a.stagger > 0) {
    const int key = (blockIdx.x >> 3) & 7;   // consecutive block ids go to different XCDs: this is the CU slot inside one
    const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)key * a.stagger;
    while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(16);
  }
  Ring<T::RS, NW, G::RD> ring;
  ring_init(ring, smem, a.wstream, a.nms_total, kRingOff);
  lds_barrier();
  P3_SPAN(1);

  int npos_done = 0;
  for (int pos0 = blockIdx.x * NPOS; pos0 < a.npos; pos0 += gridDim.x * NPOS, ++npos_done) {
    EpiOut16<NT> A1;   // activated reduce input, channel half 1, made by the previous block's expand
    if (NW == 4 && a.pair_turns) {   // whose turn at the higher priority (kernels.h)
      if ((blockIdx.x >= gridDim.x / 2) != (bool)(npos_done & 1)) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    // Runs: a launch is one run of blocks with an optional head / tail (a.head, a.tail), or — joined launches,
    // a.nruns > 1 — several runs with the broadcast blocks between them inside: run r > 0 begins with the head
    // (conv_last) of the broadcast block whose tail (conv_first + dense) ended run r - 1 for this position.
    const int nruns = BC ? a.nruns : 1;
    const bool launch_tail = BC && a.tail;
    int blk0 = 0;   // index of the run's first block in a.blk (an index, not a pointer: a.blk stays in the kernarg segment)
#pragma unroll 1
    for (int run = 0; run < nruns; ++run) {
    const int nblk = (BC && nruns > 1) ? a.run_nblk[run] : a.nblk;
    const bool head = BC && (run > 0 || a.head), tail = BC && (run + 1 < nruns || a.tail);
    // joined launches alternate between two u buffers (the tail of run r writes ubuf[r & 1], the head of run r + 1 reads
    // it): no address is written twice within a launch, so nothing this CU read earlier in the launch can be stale
    const _Float16* zin = BC ? ((run & 1) ? a.zin : a.zin2) : nullptr;
    _Float16* uout = BC ? ((run & 1) ? a.uout2 : a.uout) : nullptr;
    if (head) {
      // ---- conv_last of the broadcast block before the run: x' = x + W . z ----------------------
      // Both K slices of z are requested together (one exposed latency).  Pass 0 runs slice 0 then
      // slice 1, pass 1 starts on slice 1 — still in the act buffer — while slice 0 comes back from
      // L2 under it (the fused stream packs pass 1's K slices in that order, engine.cpp).
      f32x4 acc[4][NT];
      ResRegs16<NT> rr;
      EpiOut16<NT> S;
      P3_STAMP(6, 0);
      {
        ResRegs16<NT> z0;
        residual_addr16<G, CB, NT>(z0, C, pos0, a.npos, 0);
        residual_load16<NT>(z0, zin);
        residual_addr16<G, CB, NT>(rr, C, pos0, a.npos, CB);
        residual_load16<NT>(rr, zin);
        stash16<NT>(S, z0);
        lds_barrier();   // the act buffer is free: every wave is past the previous position's last segment
        epilogue_write16<G, CB, NT, true>(smem, S, 0);
        stash16<NT>(S, rr);   // slice 1 waits raw across the first segment
      }
      ring_note_inflight(ring, 12);
      acc16_zero<NT>(acc);
      P3_STAMP(6, 1);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(6, 2);
      lds_barrier();
      epilogue_write16<G, CB, NT, true>(smem, S, 0);
      residual_addr16<G, CB, NT>(rr, C, pos0, a.npos, 0);
      residual_load16<NT>(rr, a.x);
      ring_note_inflight(ring, 12);
      P3_STAMP(6, 4);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(6, 5);
      epilogue_store16_act<G, CB, NT>(acc, rr, a.x, A1, false, a.blk[blk0].scale[0], a.blk[blk0].shift[0], 0);
      P3_STAMP(6, 6);
      residual_addr16<G, CB, NT>(rr, C, pos0, a.npos, 0);
      residual_load16<NT>(rr, zin);
      stash16<NT>(S, rr);
      ring_note_inflight(ring, 12);
      acc16_zero<NT>(acc);
      P3_STAMP(6, 7);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(6, 8);
      lds_barrier();
      epilogue_write16<G, CB, NT, true>(smem, S, 0);
      residual_addr16<G, CB, NT>(rr, C, pos0, a.npos, CB);
      residual_load16<NT>(rr, a.x);
      ring_note_inflight(ring, 12);
      P3_STAMP(6, 10);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(6, 11);
      epilogue_store16_act<G, CB, NT>(acc, rr, a.x, A1, true, a.blk[blk0].scale[0], a.blk[blk0].shift[0], CB);
      P3_STAMP(6, 12);
      {
        // half 0 of x' again (the lanes that stored it read it back), activated for the first block
        ResRegs16<NT> xin;
        EpiOut16<NT> A0;
        residual_addr16<G, CB, NT>(xin, C, pos0, a.npos, 0);
        residual_load16<NT>(xin, a.x);
        activate_loaded16<G, CB, NT>(A0, xin, a.blk[blk0].scale[0], a.blk[blk0].shift[0], 0);
        lds_barrier();
        epilogue_write16<G, CB, NT>(smem, A0, 0);
      }
      P3_STAMP(6, 13);
    }
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
      const BlockParams& bp = a.blk[blk0 + blk];
      const bool from_hbm = blk == 0 && !head;
      const bool last = blk + 1 == nblk;
      f32x4 acc[4][NT];
      P3_STAMP(blk, 0);
      // ---- reduce 1x1 (C -> CB): the act buffer is free here (barrier at the end of the
      // previous block / position), half 0 goes in, half 1 follows under the barrier after the
      // first K slice ------------------------------------------------------------------------
      // x of a position's first block arrives in the accumulator-quad layout (12 sixteen-byte
      // loads per half), half 1 under the first K slice
      if (from_hbm) {
        ResRegs16<NT> xin;
        EpiOut16<NT> A0;
        residual_addr16<G, CB, NT>(xin, C, pos0, a.npos, 0);
        residual_load16<NT>(xin, a.x);
        activate_loaded16<G, CB, NT>(A0, xin, bp.scale[0], bp.shift[0], 0);
        residual_addr16<G, CB, NT>(xin, C, pos0, a.npos, CB);
        residual_load16<NT>(xin, a.x);
        stash16<NT>(A1, xin);   // half 1 waits in A1's registers, still raw
        epilogue_write16<G, CB, NT>(smem, A0, 0);
      }   // otherwise half 0 was written at the end of the previous block
      // ordinary vector-memory operations issued since the ring's last prefetch, all younger than
      // the glds the next two acquires wait for: the 12 residual loads and 12 stores of the
      // previous block's last pass (any block but the launch's very first), and the two halves
      // of x (24 loads) in a position's first block
      // (after a fused tail the previous position ended with 12 stores only: 36)
      // (a fused tail with the dense ends on its last pass's stores, a wave-dependent few: count the loads only)
      ring_note_inflight(ring, from_hbm ? (npos_done == 0 ? 24 : (launch_tail ? (a.tail_dense ? 24 : 36) : 48)) : 24);
      acc16_zero<NT>(acc);
      P3_STAMP(blk, 1);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(blk, 2);
      if (from_hbm) activate_stashed16<G, CB, NT>(A1, bp.scale[0], bp.shift[0], CB);
      lds_barrier();
      epilogue_write16<G, CB, NT>(smem, A1, 0);
      P3_STAMP(blk, 3);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(blk, 4);
      if (KIND == 0) {
#pragma unroll
        for (int j = 1; j <= L; ++j) {
          epilogue_layer16<G, CB, NT>(smem, acc, bp.scale[j], bp.shift[j]);
          acc16_zero<NT>(acc);
          P3_STAMP(blk, 3 + 2 * j);
          conv_segment16_3x3<G, CB>(ring, smem, acc);
          P3_STAMP(blk, 4 + 2 * j);
        }
        epilogue_layer16<G, CB, NT>(smem, acc, bp.scale[L + 1], bp.shift[L + 1]);
      } else {
        // nbt: the raw inner residual stream t is parked in HBM scratch (fp16, as the reference's
        // fp16 engine keeps it) instead of 96 fp32 registers per lane: it is written once after
        // the reduce conv, read back after the second conv of each pair, and t' = t + conv(...) is
        // written back once.
        ResRegs16<NT> tr;
        residual_addr16<G, CB, NT>(tr, CB, pos0, a.npos, 0);
        epilogue_store16<false, NT>(acc, tr, a.t);
        epilogue_layer16<G, CB, NT>(smem, acc, bp.scale[1], bp.shift[1]);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          acc16_zero<NT>(acc);
          conv_segment16_3x3<G, CB>(ring, smem, acc);
          epilogue_layer16<G, CB, NT>(smem, acc, bp.scale[2 + 2 * r], bp.shift[2 + 2 * r]);
          acc16_zero<NT>(acc);
          conv_segment16_3x3<G, CB>(ring, smem, acc);
          // t is fetched after the K loop: holding it across the 3x3 loop costs more (spills of
          // the loaded values, i.e. the same exposed latency plus scratch traffic)
          residual_addr16<G, CB, NT>(tr, CB, pos0, a.npos, 0);
          residual_load16<NT>(tr, a.t);
          residual_add16<NT>(acc, tr);
          if (r == 0) epilogue_store16<false, NT>(acc, tr, a.t);
          epilogue_layer16<G, CB, NT>(smem, acc, bp.scale[3 + 2 * r], bp.shift[3 + 2 * r]);
        }
      }
      // ---- expand 1x1 (CB -> C) + residual -> HBM, and (unless this is the launch's last
      // block) the next block's activated reduce input.  Pass 0's residual is requested before
      // its K loop; pass 1's only after its K loop — A0 occupies those registers meanwhile —
      // and lands under the barrier that follows -------------------------------------------------
      P3_STAMP(blk, 11);
      const bool act_next = !last || tail;
      const float* nsc = last ? (tail ? a.tail_scale[run] : bp.scale[0]) : a.blk[blk0 + blk + 1].scale[0];
      const float* nsh = last ? (tail ? a.tail_shift[run] : bp.shift[0]) : a.blk[blk0 + blk + 1].shift[0];
      {
        ResRegs16<NT> rr;
        EpiOut16<NT> A0;
        residual_addr16<G, CB, NT>(rr, C, pos0, a.npos, 0);
        residual_load16<NT>(rr, a.x);
        ring_note_inflight(ring, 12);
        acc16_zero<NT>(acc);
        P3_STAMP(blk, 12);
        conv_segment16<G, CB, 1, 1>(ring, smem, acc);
        P3_STAMP(blk, 13);
        epilogue_store16_act<G, CB, NT>(acc, rr, a.x, A0, act_next, nsc, nsh, 0);
        ring_note_inflight(ring, 12);   // pass 0's stores
        acc16_zero<NT>(acc);
        P3_STAMP(blk, 14);
        conv_segment16<G, CB, 1, 1>(ring, smem, acc);
        P3_STAMP(blk, 15);
        residual_addr16<G, CB, NT>(rr, C, pos0, a.npos, CB);
        residual_load16<NT>(rr, a.x);
        lds_barrier();   // every wave is done with the act buffer
        if (act_next) epilogue_write16<G, CB, NT>(smem, A0, 0);   // the next block's reduce input, half 0
        P3_STAMP(blk, 16);
        epilogue_store16_act<G, CB, NT>(acc, rr, a.x, A1, act_next, nsc, nsh, CB);
        P3_STAMP(blk, 17);
      }
    }
    if constexpr (BC && C == 256 && CB == 128 && NW == 8 && KIND == 0) {   // (nbt: the allocator parks A1 in scratch inside the block loop)
      if (tail && a.tail_dense) {
        // ---- conv_first AND the dense of the broadcast block after the run -----------------------------
        //   u = mish(bn1(Dense(t))), t = mish(W . mish(bn0_b(x'))), a 128-channel half at a time: the conv's
        //   output pass is taken transposed (SWAP) and written channel-major over the act buffer — the dense
        //   contracts over board points — then three passes of 128 dense columns run off the same ring
        //   (bdense_passes, k_bdense's K loop and epilogue).  t never goes to HBM and the dense is not its own
        //   launch (two launches of 148 us and 185 KB written + read per position and broadcast block less).
        //   Both passes take their K slices as (half 0, half 1): nothing survives the dense in the act buffer.
        //   The act buffer's zero halo is restored after each half (the next conv reads it).
        f32x4 acc[4][NT];
        float* prm = (float*)(smem + kTtBytes);   // behind Tt, inside the act buffer's bytes
        static_assert(kTtBytes + (384 + 2 * C) * 4 <= (uint32_t)G::ACT_BYTES, "dense parameters fit behind Tt");
        ring_note_inflight(ring, 24);   // the last expand pass's 12 residual loads and 12 stores
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
          P3_STAMP(7, 16 * half);
          if (half == 1) {
            // x' again (L2): half 0 activated and written now, half 1 parked raw across the first K slice
            ResRegs16<NT> xin;
            EpiOut16<NT> A0;
            residual_addr16<G, CB, NT>(xin, C, pos0, a.npos, 0);
            residual_load16<NT>(xin, a.x);
            activate_loaded16<G, CB, NT>(A0, xin, a.tail_scale[run], a.tail_shift[run], 0);
            residual_addr16<G, CB, NT>(xin, C, pos0, a.npos, CB);
            residual_load16<NT>(xin, a.x);
            stash16<NT>(A1, xin);
            epilogue_write16<G, CB, NT>(smem, A0, 0);
            ring_note_inflight(ring, 12);
          }
          acc16_zero<NT>(acc);
          P3_STAMP(7, 16 * half + 1);
          conv_segment16<G, CB, 1, 1, true>(ring, smem, acc);
          P3_STAMP(7, 16 * half + 2);
          if (half == 1) activate_stashed16<G, CB, NT>(A1, a.tail_scale[run], a.tail_shift[run], CB);
          lds_barrier();
          epilogue_write16<G, CB, NT>(smem, A1, 0);
          P3_STAMP(7, 16 * half + 3);
          conv_segment16<G, CB, 1, 1, true>(ring, smem, acc);
          P3_STAMP(7, 16 * half + 4);
          lds_barrier();   // every wave is done with the activations
          epilogue_tt16<G, CB, kTtStride, NT>(smem, acc);
          bdense_stage_params<C>(prm, a.dense_bias[run], a.dense_scale[run], a.dense_shift[run]);
          P3_STAMP(7, 16 * half + 5);
          // (the dense's first ring acquire is the barrier behind these writes)
          bdense_passes<C>(ring, smem, prm, prm + 384, prm + 384 + C, uout, pos0, half, kTtChannels);
          P3_STAMP(7, 16 * half + 6);
          // joined launches: the head of the next run reads this position's u back (other lanes of this workgroup):
          // the stores are acknowledged by L2 before the barrier, the loads come after it
          if (half == 1 && run + 1 < nruns) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          lds_barrier();   // every wave is done with Tt
          // (zeroing only the halo slots, or anything else that changes this tail's code, is to be checked against
          // the block loop's register allocation: tests/test_kernel_resources_cpu.py)
          act_zero<G>(smem);
          lds_barrier();
          P3_STAMP(7, 16 * half + 7);
        }
        // (workgroup scope: the workgroup's waves share the CU's write-through vector cache, so the wait and the barrier
        // above are the whole release / acquire.  An agent-scope acquire here — buffer_inv sc1 — also drops the XCD's
        // L2 lines, the weight stream among them, and cost 3 % of the launch.)
        blk0 += nblk;
        continue;   // the next run
      }
    }
    if (tail) {
      // ---- conv_first of the broadcast block after the run: t = mish(W . mish(bn0_b(x'))) --------
      // pass 0: half 0 is in the act buffer, half 1 in A1, as for any next block.  Pass 1 starts on
      // half 1 — still in the act buffer — while half 0 of x' is re-fetched (L2) and activated again.
      f32x4 acc[4][NT];
      ResRegs16<NT> tr;
      P3_STAMP(7, 0);
      ring_note_inflight(ring, 24);   // the last expand pass's 12 residual loads and 12 stores
      acc16_zero<NT>(acc);
      P3_STAMP(7, 1);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(7, 2);
      lds_barrier();
      epilogue_write16<G, CB, NT>(smem, A1, 0);
      P3_STAMP(7, 4);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(7, 5);
      residual_addr16<G, CB, NT>(tr, C, pos0, a.npos, 0);
      epilogue_store16_mish<NT>(acc, tr, a.tout);
      P3_STAMP(7, 6);
      {
        ResRegs16<NT> xin;
        residual_addr16<G, CB, NT>(xin, C, pos0, a.npos, 0);
        residual_load16<NT>(xin, a.x);
        stash16<NT>(A1, xin);
      }
      ring_note_inflight(ring, 12);
      acc16_zero<NT>(acc);
      P3_STAMP(7, 7);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(7, 8);
      activate_stashed16<G, CB, NT>(A1, a.tail_scale[run], a.tail_shift[run], 0);
      lds_barrier();
      epilogue_write16<G, CB, NT>(smem, A1, 0);
      P3_STAMP(7, 10);
      conv_segment16<G, CB, 1, 1>(ring, smem, acc);
      P3_STAMP(7, 11);
      lds_barrier();   // the act buffer is free for the next position
      residual_addr16<G, CB, NT>(tr, C, pos0, a.npos, CB);
      epilogue_store16_mish<NT>(acc, tr, a.tout);
      P3_STAMP(7, 12);
    }
    blk0 += nblk;
    }   // run
    P3_SPAN(2 + npos_done < 7 ? 2 + npos_done : 6);
  }
  ring_drain();
  P3_SPAN(7);
}

// =======================================================================================
// Initial 5x5 conv over the 15 binary input planes + game-state dense, model.py:1230-1237.
// Input planes are expanded on the fly from the packed GoFeatures bytes (restating
// LoadPlanes/LoadFeatures, cc/nn/engine/go_features.cc:10-61): nothing but the 1.9 KB POD
// crosses PCIe.  Channel 15 is a zero pad.
// =======================================================================================
struct FeatOff {  // byte offsets inside p3hip_features (include/p3hip.h)
  static constexpr int color = 4, komi = 8, board = 12, last = 376, atari = 416, two = 777,
                       three = 1138, ladder = 1499, size = 1860;
};

template <int C, int CP = 128>
__global__ void __launch_bounds__(kWG, 2) k_init(InitArgs a) {
  using G = Geo<1, 16, 5>;
  static_assert(C % CP == 0, "output passes");
  using T = Tiling<G, CP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr uint32_t kRingOff = G::ACT_BYTES;
  act_zero<G>(smem);
  Ring<T::RS> ring;
  ring_init(ring, smem, a.wstream, a.nms_total, kRingOff);
  lds_barrier();
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = wid / T::CG;
  const int lr = lane & 31, h = lane >> 5;

  for (int pos = blockIdx.x; pos < a.npos; pos += gridDim.x) {
    const unsigned char* f = (const unsigned char*)a.feats + (size_t)pos * FeatOff::size;
    const int color = (signed char)f[FeatOff::color];
    // ---- planes -> LDS (one thread per board point) ---------------------------------
    for (int loc = threadIdx.x; loc < kNLoc; loc += kWG) {
      h8 lo = {0, 0, 0, 0, 0, 0, 0, 0}, hi = {0, 0, 0, 0, 0, 0, 0, 0};
      auto our = [&](int off) {
        return (_Float16)((signed char)f[off + loc] == color ? 1.0f : 0.0f);
      };
      auto opp = [&](int off) {
        return (_Float16)((signed char)f[off + loc] == -color ? 1.0f : 0.0f);
      };
      lo[0] = our(FeatOff::board); lo[1] = opp(FeatOff::board);
      lo[7] = our(FeatOff::atari); hi[0] = opp(FeatOff::atari);
      hi[1] = our(FeatOff::two); hi[2] = opp(FeatOff::two);
      hi[3] = our(FeatOff::three); hi[4] = opp(FeatOff::three);
      hi[5] = our(FeatOff::ladder); hi[6] = opp(FeatOff::ladder);
      const int y = (loc * 3450) >> 16, xx = loc - y * kBL;
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const int* lm = (const int*)(f + FeatOff::last + m * 8);
        if (lm[0] == y && lm[1] == xx) lo[2 + m] = (_Float16)1.0f;  // pass {19,0}/noop never match
      }
      const int s = G::PADTOP + y * G::S + xx;
      *(h8*)(smem + s * G::SLOTB) = lo;
      *(h8*)(smem + s * G::SLOTB + 16) = hi;
    }
    // ---- game-state scalars (LoadFeatures) ------------------------------------------
    float gsv[8];
    gsv[0] = color == 1 ? 1.0f : 0.0f;
    gsv[1] = color == 1 ? 0.0f : 1.0f;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const int* lm = (const int*)(f + FeatOff::last + m * 8);
      gsv[2 + m] = (lm[0] == 19 && lm[1] == 0) ? 1.0f : 0.0f;
    }
    gsv[7] = (color == 1 ? -1.0f : 1.0f) * (*(const float*)(f + FeatOff::komi)) / 15.0f;

    // game-state dense once per position: thread c computes (gs . Wg + b)[c] into LDS (the
    // area behind the weight ring); the first ring acquire of the K loop below is the barrier
    // that publishes it.  (Each lane used to fetch its 32 channels' 8 x 4 weights from L2 in
    // the epilogue of every output pass.)
    float* bias_lds = (float*)(smem + kRingOff + ring_bytes(CP));
    if (threadIdx.x < C) {
      float b = a.game_b[threadIdx.x];
#pragma unroll
      for (int k = 0; k < 8; ++k) b += a.game_w[k * C + threadIdx.x] * gsv[k];
      bias_lds[threadIdx.x] = b;
    }

#pragma unroll 1
    for (int cp = 0; cp < C / CP; ++cp) {
      f32x16 acc[2][T::NT];
      acc_zero<G, CP>(acc);
      conv_segment<G, CP, 5, 28>(ring, smem, acc);
      // epilogue: + (gs . Wg + b)[c]  -> x
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int c = cp * CP + acc_chan<G, CP>(mt, g4);
          const f32x4 bias = *(const f32x4*)(bias_lds + c);
#pragma unroll
          for (int j = 0; j < T::NT; ++j) {
            const int t = lg + j * T::LG;
            if (t >= G::NT_TOTAL) continue;
            const int r = t * 32 + lr;
            int loc;
            if (!row_valid<G::S>(r, loc)) continue;
            h4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (_Float16)(acc[mt][j][g4 * 4 + i] + bias[i]);
            *(h4*)(a.x + ((size_t)pos * (C / 8) + (c >> 3)) * (kNLoc * 8) + loc * 8 + h * 4) = o;
          }
        }
    }
    lds_barrier();
    // clear the 5 last-move/stone planes for the next position: every valid slot is
    // rewritten in full by the staging loop above, so nothing to do.
  }
  ring_drain();
}

// =======================================================================================
// Generic 1x1 conv kernel over the channel-blocked stream.
//   PRE  : apply mish(bn(.)) while staging (ConvPreActivation prologue)
//   EPI 0: out = mish(acc)               -> y (fp16)   [broadcast conv_first + BroadcastPreAct act,
//                                                       model.py:556-560,590-596]
//   EPI 1: x  += acc                      (residual)   [broadcast conv_last, model.py:600-606]
//   EPI 2: hp  = acc (fp32 [pos][COUT][361])            [policy conv_p/conv_g, value conv;
//                                                       model.py:783-786,889]
// =======================================================================================
template <int CIN, int COUT, bool PRE, int EPI>
__global__ void __launch_bounds__(kWG, 2) k_conv1x1(Conv1x1Args a) {
  constexpr int CB = CIN >= 256 ? 128 : 64;
  constexpr int NPOS = 128 / CB;
  constexpr int CP = (COUT >= 128 && CB == 128) ? 128 : 64;  // cout pass
  using G = Geo<NPOS, CB, 1>;
  using T = Tiling<G, CP>;
  constexpr int NCP = (COUT + CP - 1) / CP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr uint32_t kRingOff = G::ACT_BYTES;
  act_zero<G>(smem);
  // Output passes are split ACROSS workgroups: workgroup b owns pass cp = (b / 8) % NCP of the
  // position groups pg0, pg0 + gridDim/NCP, ...  The NCP workgroups of one position group are
  // 8 block ids apart, i.e. on the same XCD and dispatched together, so the input slices they
  // all stage come from HBM once and from that XCD's L2 afterwards (each workgroup restaging
  // every pass itself re-read them from HBM: the working set of an XCD's 32 workgroups is
  // larger than its L2).  gridDim.x is a multiple of 8 * NCP (conv_split_grid).
  const int cp = (blockIdx.x >> 3) % NCP;
  const int pg0 = (blockIdx.x / (8 * NCP)) * 8 + (blockIdx.x & 7);
  const int pg_stride = gridDim.x / NCP;
  Ring<T::RS> ring;
  ring_init(ring, smem, (const char*)a.wstream + (size_t)cp * (a.nms_total / NCP) * T::RS, a.nms_total / NCP, kRingOff);
  lds_barrier();
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = wid / T::CG;
  const int lr = lane & 31;

  // Staging is software-pipelined: the slice after the current one (next K slice or next
  // position) is fetched into registers while the current segment's MFMAs run, its BN+mish
  // is applied in registers before the barrier that frees the act buffer.
  constexpr int NIP = CIN / CB;
  XRegs<G> xr;
  stage_load<G>(xr, a.in, CIN, pg0 * NPOS, a.npos, 0);
  int pending_stores = 0;   // vector-memory ops of the previous epilogue that may still be in flight
  for (int pos0 = pg0 * NPOS; pos0 < a.npos; pos0 += pg_stride * NPOS) {
    {
      f32x16 acc[2][T::NT];
      acc_zero<G, CP>(acc);
#pragma unroll 1
      for (int ip = 0; ip < NIP; ++ip) {
        if (PRE) stage_math<G>(xr, ip * G::NCH, a.scale, a.shift);
        lds_barrier();
        stage_store<G, false>(smem, xr, ip * G::NCH, nullptr, nullptr);
        int nip = ip + 1, npos0 = pos0;
        if (nip == NIP) {
          nip = 0;
          npos0 = pos0 + pg_stride * NPOS;   // past the end: clamped to a valid position
        }
        stage_load<G>(xr, a.in, CIN, npos0, a.npos, nip * G::NCH);
        ring_note_inflight(ring, pending_stores + kXLoads);
        pending_stores = 0;
        conv_segment<G, CP, 1, 1>(ring, smem, acc);
      }
      static_assert(EPI == 2 || T::NT == 3, "vmcnt bookkeeping: 12 sixteen-byte stores per pass");
      pending_stores = (EPI == 2) ? 0 : 12;
      if (EPI == 1) {
        epilogue_to_global<G, CP, true>(acc, a.out16, COUT, pos0, a.npos, cp * CP);
      } else if (EPI == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < T::NT; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][j][i] = mish_f(acc[mt][j][i]);
        epilogue_to_global<G, CP, false>(acc, a.out16, COUT, pos0, a.npos, cp * CP);
      } else {
        const int h = lane >> 5;
#pragma unroll
        for (int j = 0; j < T::NT; ++j) {
          const int t = lg + j * T::LG;
          if (t >= G::NT_TOTAL) continue;
          const int p = t / G::NT_POS, tt = t - p * G::NT_POS;
          const int loc = tt * 32 + lr;  // S == 19: row == loc
          if (loc >= kNLoc || pos0 + p >= a.npos) continue;
          // head activations go out as the accumulator quads they are: hp[pos][c / 4][loc][4] fp32, one 16-byte
          // store per quad (a wave instruction covers two contiguous 512-byte runs); k_heads reads quads
          static_assert(COUT % 4 == 0, "channel quads");
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int wid_cg = wid % T::CG;
              const int c0 = cp * CP + wid_cg * 64 + mt * 32 + 8 * g4 + 4 * h;   // first channel of the quad
              if (c0 < COUT)
                *(f32x4*)(a.out32 + (((size_t)(pos0 + p) * (COUT / 4) + (c0 >> 2)) * kNLoc + loc) * 4) =
                    f32x4{acc[mt][j][4 * g4], acc[mt][j][4 * g4 + 1], acc[mt][j][4 * g4 + 2], acc[mt][j][4 * g4 + 3]};
            }
        }
      }
    }
  }
  lds_barrier();
  ring_drain();
}

// =======================================================================================
// Layer-wise conv kernel for trunks whose bottleneck does not fit the fused block kernel's
// LDS plan (C = 384, C_b = 192: one position of 192 channels on the padded grid is 168 KB).
// Each conv of a block is its own launch; activations round-trip HBM in fp16.  A workgroup
// owns two positions, input channels are staged in 64-channel slices (K split, accumulators
// stay in registers across slices), outputs are produced in 64-channel passes.
//   pre : stage mish(bn_in(.))          (ConvPreActivation prologue, model.py:203-292)
//   act : store mish(bn_out(acc))       (the NEXT layer's prologue applied by the producer,
//                                        so inner layers stage without VALU work)
//   res : out += acc                    (residual, in place)
// =======================================================================================
// NW = 4 (the shipped form; NW = 8 = two positions per 512-thread workgroup, one per CU, P3HIP_LCONV_WG8): a
// 256-thread workgroup per position — 1x1: 52 KB of activations + 24 KB of ring, 3x3: 60.6 KB + 12 KB (K = 32 ring
// steps) — so that TWO workgroups share a CU and one's loads, stores and BN + mish run under the other's K loop;
// they take turns at the higher wave priority, one position group each (see BlockArgs::pair_turns).
template <int KW, int CIN, int COUT, bool PRE, bool ACT, bool RES, bool DUAL = false, int NW = 8>
__global__ void __launch_bounds__(NW * 64, 2) k_lconv(LConvArgs a) {
  static_assert(!(ACT && DUAL), "act stores the activated tensor only, dual stores both");
  constexpr int CB = 64, NPOS = NW == 8 ? 2 : 1, CP = 64;
  // 3x3 layers in the 4-wave form: K = 32 ring steps (60.6 KB of activations + 12 KB of ring), as in k_block
  using G = Geo<NPOS, CB, KW, NW, (NW == 4 && KW == 3) ? 2 : kKMS>;
  using T = Tiling<G, CP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  act_zero<G>(smem);
  constexpr int NIP = CIN / CB, NCP = COUT / CP;
  // output passes split across workgroups, as in k_conv1x1
  const int cp = (blockIdx.x >> 3) % NCP;
  const int pg0 = (blockIdx.x / (8 * NCP)) * 8 + (blockIdx.x & 7);
  const int pg_stride = gridDim.x / NCP;
  Ring<T::RS, G::NW, G::RD> ring;
  ring_init(ring, smem, (const char*)a.wstream + (size_t)cp * (a.nms_total / NCP) * T::RS, a.nms_total / NCP, G::ACT_BYTES);
  lds_barrier();
  XRegs<G> xr;   // software-pipelined staging, as in k_conv1x1
  stage_load<G>(xr, a.in, CIN, pg0 * NPOS, a.npos, 0);
  int pending_stores = 0;
  int turn = 0;
  for (int pos0 = pg0 * NPOS; pos0 < a.npos; pos0 += pg_stride * NPOS, ++turn) {
    if (NW == 4 && a.pair_split > 0) {
      if (((int)blockIdx.x >= a.pair_split) != (bool)(turn & 1)) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    {
      f32x16 acc[2][T::NT];
      acc_zero<G, CP>(acc);
#pragma unroll 1
      for (int ip = 0; ip < NIP; ++ip) {
        if (PRE) stage_math<G>(xr, ip * G::NCH, a.scale_in, a.shift_in);
        lds_barrier();
        stage_store<G, false>(smem, xr, ip * G::NCH, nullptr, nullptr);
        int nip = ip + 1, npos0 = pos0;
        if (nip == NIP) {
          nip = 0;
          npos0 = pos0 + pg_stride * NPOS;
        }
        stage_load<G>(xr, a.in, CIN, npos0, a.npos, nip * G::NCH);
        ring_note_inflight(ring, pending_stores + kXLoads);
        pending_stores = 0;
        conv_segment<G, CP, KW, KW * KW>(ring, smem, acc);
      }
      // BN + mish of the output in place; parameters are fetched one channel quad at a time so
      // the prefetched slice stays in registers
      auto activate = [&]() {
        const int c0 = cp * CP + cg_of<G, CP>() * 64 + (launder(threadIdx.x & 63) >> 5) * 4;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const f32x4 sc = *(const f32x4*)(a.scale_out + c0 + 8 * k), sh = *(const f32x4*)(a.shift_out + c0 + 8 * k);
#pragma unroll
          for (int j = 0; j < T::NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              acc[k >> 2][j][(k & 3) * 4 + i] = mish_f(acc[k >> 2][j][(k & 3) * 4 + i] * sc[i] + sh[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (DUAL) {
        ResRegs<G, CP, T::NT> rr;
        residual_addr<G, CP, T::NT>(rr, COUT, pos0, a.npos, cp * CP);
        if (RES) {
          residual_load<G, CP, T::NT>(rr, a.out);
          residual_add<G, CP, T::NT>(acc, rr);
        }
        static_assert(T::NT == 3, "vmcnt bookkeeping: 12 sixteen-byte stores per output");
        epilogue_store<G, CP, false, T::NT>(acc, rr, a.out);    // raw y
        activate();
        epilogue_store<G, CP, false, T::NT>(acc, rr, a.out2);   // the consumer's input, activated once here
        pending_stores = 24;
      } else {
        pending_stores = 12;
        if (ACT) activate();
        epilogue_to_global<G, CP, RES>(acc, a.out, COUT, pos0, a.npos, cp * CP);
      }
    }
  }
  lds_barrier();
  ring_drain();
}

// =======================================================================================
// Broadcast dense: per channel c, u[c][j] = sum_i t[c][i] W[i][j] + b[j]  (Dense(361) over
// the flattened board, weights shared by all channels; BroadcastPreAct.call, model.py:
// 556-567; `t` already carries the mish).  Then the following ConvPreActivation prologue
// mish(bn1(u)) is applied here so that conv_last runs with PRE = false.
//   MFMA orientation: D[c][j] = sum_i Tt[c][i] * Wt[j][i]  (A = activations transposed in
//   LDS to [c][i], B = dense matrix rows streamed through the ring).
// =======================================================================================
template <int C>
__global__ void __launch_bounds__(kWG, 2) k_bdense(BDenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = kTtChannels;
  constexpr uint32_t kRingOff = kTtBytes;
  for (int i = threadIdx.x * 16; i < (int)kTtBytes; i += kWG * 16) *(f32x4*)(smem + i) = f32x4{0, 0, 0, 0};
  Ring<16384> ring;
  ring_init(ring, smem, a.wstream, a.nms_total, kRingOff);
  // epilogue parameters (dense bias per board point, folded BN per channel) live in LDS behind
  // the ring: fetched once per workgroup instead of from L2 after every K loop
  float* p_bias = (float*)(smem + kRingOff + ring_bytes(128));
  float* p_scale = p_bias + 384;
  float* p_shift = p_scale + C;
  bdense_stage_params<C>(p_bias, a.bias, a.scale, a.shift);
  lds_barrier();

  // Staging is software-pipelined like the conv kernels': the 12 16-byte loads of the next
  // 128-channel pass (next half or next position) are issued before the current pass's K loop
  // and scattered (transposed) into LDS after the barrier that ends it.
  // Thread (combo = channel block of the pass, l32) owns the six PAIRS of adjacent board points
  // 2*(l32 + 32*i), +1: two adjacent 16-byte loads per pair (still kXLoads = 12 per thread, the
  // ring's vmcnt bookkeeping is unchanged) and one ds_write_b32 per channel and pair when the
  // slice is transposed into Tt[c][i] — half the LDS store instructions of a per-point scatter.
  using GS = Geo<1, 128, 1>;
  constexpr int NHALF = (C + CH - 1) / CH;
  XRegs<GS> xr;
  auto pair_load = [&](int pos_, int cblk) {
    static_assert(kXLoads == 12, "six pairs");
    int pp = pos_ < a.npos ? pos_ : a.npos - 1;
    const _Float16* src = a.t + ((size_t)pp * (C / 8) + cblk + (threadIdx.x >> 5)) * (kNLoc * 8);
    const int l32 = threadIdx.x & 31;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      int l0 = 2 * (l32 + 32 * i), l1 = l0 + 1;
      if (l0 >= kNLoc) l0 = kNLoc - 1;   // tail lanes re-read a valid item (not stored)
      if (l1 >= kNLoc) l1 = kNLoc - 1;
      xr.v[2 * i] = *(const h8*)(src + l0 * 8);
      xr.v[2 * i + 1] = *(const h8*)(src + l1 * 8);
    }
  };
  pair_load(blockIdx.x, 0);
  for (int pos = blockIdx.x; pos < a.npos; pos += gridDim.x) {
#pragma unroll 1
    for (int half = 0; half < NHALF; ++half) {
      // channels of this pass (the last pass of C = 192 has 64: its upper channel tiles idle
      // but still take part in the ring's barriers)
      const int nch = (C - half * CH) < CH ? (C - half * CH) : CH;
      lds_barrier();
      // ---- transpose-stage t[pos][cblk][loc][8] -> Tt[c][i] --------------------------
      {
        const int combo = threadIdx.x >> 5, l32 = threadIdx.x & 31;   // combo = channel block of this pass
        if (combo * 8 < nch) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const int loc = 2 * (l32 + 32 * i);
            if (loc >= kNLoc) continue;
            const h8 v0 = xr.v[2 * i];
            h8 v1 = xr.v[2 * i + 1];
            if (loc + 1 >= kNLoc) v1 = h8{0, 0, 0, 0, 0, 0, 0, 0};   // board point 361 is padding (K = 384)
#pragma unroll
            for (int e = 0; e < 8; ++e) *(h2*)(smem + (combo * 8 + e) * kTtStride + loc * 2) = h2{v0[e], v1[e]};
          }
        }
      }
      {
        int nhalf = half + 1, npos = pos;
        if (nhalf == NHALF) { nhalf = 0; npos = pos + gridDim.x; }
        // channel blocks past C (second pass of C = 192) are clamped to a valid block and ignored
        int cblk0 = nhalf * (CH / 8);
        pair_load(npos, cblk0 + ((threadIdx.x >> 5) * 8 < C - nhalf * CH ? 0 : -(int)(threadIdx.x >> 5)));
        ring_note_xloads(ring);
      }
      bdense_passes<C>(ring, smem, p_bias, p_scale, p_shift, a.u, pos, half, nch);
    }
  }
  lds_barrier();
  ring_drain();
}

// =======================================================================================
// Heads tail: everything after the three 1x1 head convs (hp = [p(32) | g(32) | v(32)]).
// PolicyHead.call model.py:783-812, GlobalPoolBias.call :696-706, ValueHead.call :887-979,
// output softmaxes :1265-1267; optimistic-policy softmax is the reference's host-side
// core::Softmax in TrtEngineImpl::GetBatch (cc/nn/engine/trt_engine.cc:347-348) moved on
// device.  fp32 VALU throughout; one 256-thread workgroup per position.
// =======================================================================================
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// LDS plan of k_heads (float offsets): every small weight matrix of the heads staged once
// per workgroup, then one scratch area per 256-thread position group.
template <int H, int V>
struct HeadsLds {
  static constexpr int gd_w = 0;                               // [2H][H]
  static constexpr int oq_embed_w = gd_w + 2 * H * H;          // [2H][V]
  static constexpr int gamma_pre_w = oq_embed_w + 2 * H * V;   // [2H][V]
  static constexpr int score_pre_w = gamma_pre_w + 2 * H * V;  // [2H+1][V]
  static constexpr int oq_out_w = score_pre_w + (2 * H + 1) * V;  // [V][14]
  static constexpr int gamma_out_w = oq_out_w + V * 14;        // [V]
  static constexpr int score_out_w = gamma_out_w + V;          // [V]
  static constexpr int pass_w = score_out_w + V;               // [2H][2]
  static constexpr int opt_pass_w = pass_w + 4 * H;            // [2H]
  static constexpr int moves_w = opt_pass_w + 2 * H;           // [H][2]
  static constexpr int opt_moves_w = moves_w + 2 * H;          // [H]
  static constexpr int own_w = opt_moves_w + H;                // [H]
  static constexpr int gbn_scale = own_w + H;                  // [H]
  static constexpr int gbn_shift = gbn_scale + H;              // [H]
  static constexpr int gd_b = gbn_shift + H;                   // [H]
  static constexpr int oq_embed_b = gd_b + H;                  // [V]
  static constexpr int gamma_pre_b = oq_embed_b + V;           // [V]
  static constexpr int score_pre_b = gamma_pre_b + V;          // [V]
  static constexpr int oq_out_b = score_pre_b + V;             // [14] (+2 pad)
  static constexpr int n_weights = oq_out_b + 16;
  // per position group
  static constexpr int gp = 0, vp = gp + 2 * H, gbias = vp + 2 * H, emb = gbias + H, gpre = emb + V,
                       base = gpre + V, red = base + V, misc = red + 16, logits = misc + 4,
                       pi = logits + 800, opt = pi + 364, n_scratch = opt + 364;
  static constexpr int kGroups = 4;
  static constexpr size_t bytes = (size_t)(n_weights + kGroups * n_scratch) * 4;
};

// One 1024-thread workgroup evaluates the heads of four positions at a time, one per
// 256-thread group; all groups run the same phases in lockstep (block-wide barriers).
// Every weight is read from LDS; the only global traffic inside the phases is the head
// conv output `hp` (each value once or twice, coalesced) and the result row.
template <int H, int V>
__global__ void __launch_bounds__(1024) k_heads(HeadsArgs a) {
  static_assert(H == 32, "head channels");
  using L = HeadsLds<H, V>;
  extern __shared__ __attribute__((aligned(16))) float hl[];
  const int tid = threadIdx.x, g = tid >> 8, t = tid & 255;
  const int wid = t >> 6, lane = t & 63;
  auto stage = [&](int off, const float* __restrict__ src, int n) {
    for (int i = tid; i < n; i += 1024) hl[off + i] = src[i];
  };
  stage(L::gd_w, a.gd_w, 2 * H * H);
  stage(L::oq_embed_w, a.oq_embed_w, 2 * H * V);
  stage(L::gamma_pre_w, a.gamma_pre_w, 2 * H * V);
  stage(L::score_pre_w, a.score_pre_w, (2 * H + 1) * V);
  stage(L::oq_out_w, a.oq_out_w, V * 14);
  stage(L::gamma_out_w, a.gamma_out_w, V);
  stage(L::score_out_w, a.score_out_w, V);
  stage(L::pass_w, a.pass_w, 4 * H);
  stage(L::opt_pass_w, a.opt_pass_w, 2 * H);
  stage(L::moves_w, a.moves_w, 2 * H);
  stage(L::opt_moves_w, a.opt_moves_w, H);
  stage(L::own_w, a.own_w, H);
  stage(L::gbn_scale, a.gbn_scale, H);
  stage(L::gbn_shift, a.gbn_shift, H);
  stage(L::gd_b, a.gd_b, H);
  stage(L::oq_embed_b, a.oq_embed_b, V);
  stage(L::gamma_pre_b, a.gamma_pre_b, V);
  stage(L::score_pre_b, a.score_pre_b, V);
  stage(L::oq_out_b, a.oq_out_b, 14);
  const float pass_b0 = a.pass_b[0], opt_pass_b = a.opt_pass_b[0], gamma_out_b = a.gamma_out_b[0],
              score_out_b = a.score_out_b[0];
  float* sc = hl + L::n_weights + g * L::n_scratch;
  __syncthreads();

  for (int pos4 = blockIdx.x * L::kGroups; pos4 < a.npos; pos4 += gridDim.x * L::kGroups) {
    const bool live = pos4 + g < a.npos;
    const int pos = live ? pos4 + g : a.npos - 1;   // idle groups recompute the last position
    // head activations as channel quads: hp4[(c / 4) * 361 + loc] = channels c .. c + 3 of board point loc
    // (k_conv1x1's EPI 2); quads 0 .. H/4-1 = p, H/4 .. 2H/4-1 = g, 2H/4 .. 3H/4-1 = v
    const f32x4* __restrict__ hp4 = (const f32x4*)(a.hp + (size_t)pos * 3 * H * kNLoc);
    float* __restrict__ out = a.out + (size_t)pos * kOutStride;
    float* __restrict__ res = a.res ? a.res + (size_t)pos * kResultFloats : nullptr;
    // ---- pooled g (after bn+mish) and pooled v (raw): one wave per channel quad, six 16-byte loads in
    // flight -------------------------------------------------------------------------------------------
    static_assert(H % 4 == 0, "channel quads");
    for (int qd = wid; qd < 2 * H / 4; qd += 4) {   // quad of the g / v channels
      const int c = 4 * qd;                          // g channels are c < H, v channels c >= H
      const f32x4* src = hp4 + (size_t)(H / 4 + qd) * kNLoc;
      f32x4 v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int i = lane + 64 * k;
        v[k] = src[i < kNLoc ? i : kNLoc - 1];
      }
      const bool is_g = c < H;
      const int ch = is_g ? c : c - H;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bsc = is_g ? hl[L::gbn_scale + ch + e] : 1.0f, bsh = is_g ? hl[L::gbn_shift + ch + e] : 0.0f;
        float s = 0.0f, m = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          float x = v[k][e];
          if (is_g) x = mish_f(x * bsc + bsh);
          if (lane + 64 * k < kNLoc) {
            s += x;
            m = fmaxf(m, x);
          }
        }
        s = wave_sum(s);
        m = wave_max(m);
        if (lane == 0) {
          float* dst = sc + (is_g ? L::gp : L::vp);
          dst[ch + e] = s * (1.0f / kNLoc);
          dst[H + ch + e] = m;
        }
      }
    }
    __syncthreads();
    // ---- small dense layers ----------------------------------------------------------
    if (t < H) {
      float s = hl[L::gd_b + t];
#pragma unroll 8
      for (int k = 0; k < 2 * H; ++k) s += sc[L::gp + k] * hl[L::gd_w + k * H + t];
      sc[L::gbias + t] = s;
    } else if (t >= 64 && t < 64 + V) {
      const int o = t - 64;
      float s = hl[L::oq_embed_b + o], gm = hl[L::gamma_pre_b + o], b = hl[L::score_pre_b + o];
#pragma unroll 8
      for (int k = 0; k < 2 * H; ++k) {
        const float x = sc[L::vp + k];
        s += x * hl[L::oq_embed_w + k * V + o];
        gm += x * hl[L::gamma_pre_w + k * V + o];
        b += x * hl[L::score_pre_w + k * V + o];
      }
      sc[L::emb + o] = mish_f(s);
      sc[L::gpre + o] = mish_f(gm);
      sc[L::base + o] = b;
    } else if (t == 192) {
      float s0 = pass_b0, so = opt_pass_b;
#pragma unroll 8
      for (int k = 0; k < 2 * H; ++k) {
        s0 += sc[L::gp + k] * hl[L::pass_w + k * 2];
        so += sc[L::gp + k] * hl[L::opt_pass_w + k];
      }
      sc[L::pi + 361] = s0 - 3.0f;
      sc[L::opt + 361] = so - 3.0f;
    }
    __syncthreads();
    if (t < 14) {
      float s = hl[L::oq_out_b + t];
#pragma unroll 8
      for (int k = 0; k < V; ++k) s += sc[L::emb + k] * hl[L::oq_out_w + k * 14 + t];
      if (t < 2) sc[L::misc + t] = s;
      if (live) {
        if (t < 2) out[kOffOutcomeLogits + t] = s;
        if (t == 5) {
          out[kOffErr2] = 4.0f / (1.0f + __expf(-s));
          if (res) res[kOffErr2] = out[kOffErr2];
        }
      }
    } else if (t == 64) {
      float s = gamma_out_b;
#pragma unroll 8
      for (int k = 0; k < V; ++k) s += sc[L::gpre + k] * hl[L::gamma_out_w + k];
      if (live) out[kOffGamma] = s;
      const float sp = s > 20.0f ? s : log1pf(__expf(s));
      sc[L::misc + 2] = fminf(sp, 10.0f);
    }
    // ---- per-location policy logits and ownership (needs gbias only) ------------------
    for (int i = t; i < kNLoc; i += 256) {
      float pi = 0.0f, po = 0.0f, ow = 0.0f;
      const int z = launder(0);   // keeps the LDS weight reads inside the location loop
#pragma unroll
      for (int c0 = 0; c0 < H; c0 += 8) {   // four 16-byte loads in flight per round
        float pv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 p4 = hp4[(size_t)(c0 / 4 + u) * kNLoc + i];
          const f32x4 v4 = hp4[(size_t)(2 * H / 4 + c0 / 4 + u) * kNLoc + i];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pv[4 * u + e] = p4[e];
            vv[4 * u + e] = v4[e];
          }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float p = mish_f(pv[c] + sc[L::gbias + z + c0 + c]);
          pi += p * hl[L::moves_w + z + (c0 + c) * 2];
          po += p * hl[L::opt_moves_w + z + c0 + c];
          ow += vv[c] * hl[L::own_w + z + c0 + c];
        }
      }
      sc[L::pi + i] = pi;
      sc[L::opt + i] = po;
      if (live) out[kOffOwnership + i] = tanhf(ow);
    }
    __syncthreads();   // misc[2] (gamma) ready
    // ---- score logits: 800 bins x V ---------------------------------------------------
    {
      const float gam = sc[L::misc + 2];
      for (int sidx = t; sidx < 800; sidx += 256) {
        const float sv = 0.05f * (float)(sidx - 400) + 0.025f;
        float s = score_out_b;
        const int z = launder(0);   // keeps the LDS reads inside the bin loop
#pragma unroll 8
        for (int k = 0; k < V; ++k)
          s += mish_f(sc[L::base + z + k] + sv * hl[L::score_pre_w + (2 * H) * V + z + k]) * hl[L::score_out_w + z + k];
        sc[L::logits + sidx] = gam * s;
      }
    }
    __syncthreads();
    // ---- raw logits out + the three softmaxes, reductions fused ------------------------
    float m3[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = t; i < 362; i += 256) {
      const float p = sc[L::pi + i], o = sc[L::opt + i];
      if (live) {
        out[kOffMoveLogits + i] = p;
        out[kOffOptLogits + i] = o;
        if (res) res[kOffMoveLogits + i] = p;
      }
      m3[0] = fmaxf(m3[0], p);
      m3[1] = fmaxf(m3[1], o);
    }
    for (int i = t; i < 800; i += 256) {
      const float l = sc[L::logits + i];
      if (live) out[kOffScoreLogits + i] = l;
      m3[2] = fmaxf(m3[2], l);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) m3[r] = wave_max(m3[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) sc[L::red + r * 4 + wid] = m3[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r)
      m3[r] = fmaxf(fmaxf(sc[L::red + r * 4], sc[L::red + r * 4 + 1]),
                    fmaxf(sc[L::red + r * 4 + 2], sc[L::red + r * 4 + 3]));
    __syncthreads();   // red is rewritten below
    float s3[3] = {0.0f, 0.0f, 0.0f};
    for (int i = t; i < 362; i += 256) {
      const float e0 = __expf(sc[L::pi + i] - m3[0]), e1 = __expf(sc[L::opt + i] - m3[1]);
      sc[L::pi + i] = e0;
      sc[L::opt + i] = e1;
      s3[0] += e0;
      s3[1] += e1;
    }
    for (int i = t; i < 800; i += 256) {
      const float e = __expf(sc[L::logits + i] - m3[2]);
      sc[L::logits + i] = e;
      s3[2] += e;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) s3[r] = wave_sum(s3[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) sc[L::red + r * 4 + wid] = s3[r];
    }
    __syncthreads();
    if (live) {
      const float i0 = 1.0f / (sc[L::red + 0] + sc[L::red + 1] + sc[L::red + 2] + sc[L::red + 3]);
      const float i1 = 1.0f / (sc[L::red + 4] + sc[L::red + 5] + sc[L::red + 6] + sc[L::red + 7]);
      const float i2 = 1.0f / (sc[L::red + 8] + sc[L::red + 9] + sc[L::red + 10] + sc[L::red + 11]);
      for (int i = t; i < 362; i += 256) {
        out[kOffMoveProbs + i] = sc[L::pi + i] * i0;
        out[kOffOptProbs + i] = sc[L::opt + i] * i1;
        if (res) {
          res[kOffMoveProbs + i] = sc[L::pi + i] * i0;
          res[kOffOptProbs + i] = sc[L::opt + i] * i1;
        }
      }
      for (int i = t; i < 800; i += 256) {
        out[kOffScoreProbs + i] = sc[L::logits + i] * i2;
        if (res) res[kOffScoreProbs + i] = sc[L::logits + i] * i2;
      }
      if (t == 0) {
        const float v0 = sc[L::misc], v1 = sc[L::misc + 1];
        const float m = fmaxf(v0, v1);
        const float e0 = __expf(v0 - m), e1 = __expf(v1 - m);
        out[kOffValueProbs] = e0 / (e0 + e1);
        out[kOffValueProbs + 1] = e1 / (e0 + e1);
        if (res) {
          res[kOffValueProbs] = e0 / (e0 + e1);
          res[kOffValueProbs + 1] = e1 / (e0 + e1);
        }
      }
    }
    __syncthreads();
  }
}

// =======================================================================================
// Heads with their 1x1 convs inside (C <= 256): k_headsx.  The three head convs (conv_p | conv_g | value.conv,
// C -> 96, raw x in: PolicyHead.call model.py:783-812, ValueHead.call :887-979) are 18 MFLOP per position, but as
// their own launch they read x (185 KB per position at C = 256) and write 139 KB of fp32 head activations that
// k_heads then reads 1.5 times: 544 MB per 1024 positions, three quarters of it the hand-over.  Here a position's
// conv runs on MFMA 16x16x32 with the activation fragments taken STRAIGHT from global memory — in x's channel-blocked
// layout [C/8][361][8] a lane's 8 consecutive channels of one board point are 16 contiguous bytes, exactly a B
// fragment — the 96 x C weights resident in LDS as A fragments, and everything k_heads did with the head activations
// done on the accumulators: g is pooled (mean / max of mish(bn(g))) and v pooled raw as the tiles come, ownership is
// finished per tile, p waits in registers (48 per lane) for the pooled bias and becomes the policy logits after the
// small dense layers.  x is read once, nothing else moves: 189 MB per 1024 positions.
// One 512-thread workgroup = two positions side by side (four waves each, 256 VGPRs a wave: p and the prefetched
// fragments live in registers), one workgroup per CU.  The phases after the pooling are k_heads' own, from LDS.
// Lane (n = lane & 15, q = lane >> 4) of location tile t: acc[ct][i] = head channel ct*16 + 4q + i at board point 16 t + n.
// =======================================================================================
template <int C, int H, int V>
struct HeadsxLds {
  using L = HeadsLds<H, V>;
  static constexpr int kGroups = 2, NS = C / 32;
  static constexpr int part = L::n_weights + kGroups * L::n_scratch;       // [group][wave][4 kinds][H] pooling partials
  static constexpr int n_part = kGroups * 4 * 4 * H;
  static constexpr size_t conv_off = ((size_t)(part + n_part) * 4 + 15) / 16 * 16;   // A fragments [6][NS][64 lanes][8] fp16
  static constexpr size_t bytes = conv_off + (size_t)6 * NS * 1024;
};

template <int C, int H, int V>
__global__ void __launch_bounds__(512, 2) k_headsx(HeadsArgs a) {
  static_assert(H == 32 && C % 32 == 0, "head channels");
  using L = HeadsLds<H, V>;
  using X = HeadsxLds<C, H, V>;
  constexpr int NS = X::NS;
  extern __shared__ __attribute__((aligned(16))) float hl[];
  const int tid = threadIdx.x, g = tid >> 8, t = tid & 255;
  const int wid = t >> 6, lane = t & 63;
  const int n = lane & 15, q = lane >> 4;
  // the small head tensors arrive as one image in L's order (engine.cpp): one round of 16-byte loads instead of
  // nineteen dependent little loops (k_heads' staging: a round trip to L2 each, ~30 us of a 110 us kernel)
  static_assert(L::n_weights == heads_image_floats(H, V) && L::n_weights % 4 == 0, "image = HeadsLds");
  {
    const f32x4* src = (const f32x4*)a.image;
    f32x4* dst = (f32x4*)hl;
    for (int i = tid; i < L::n_weights / 4; i += 512) dst[i] = src[i];
  }
  {
    const f32x4* src = (const f32x4*)a.conv_a;
    f32x4* dst = (f32x4*)((char*)hl + X::conv_off);
    for (int i = tid; i < 6 * NS * 64; i += 512) dst[i] = src[i];
  }
  const float pass_b0 = a.pass_b[0], opt_pass_b = a.opt_pass_b[0], gamma_out_b = a.gamma_out_b[0],
              score_out_b = a.score_out_b[0];
  float* sc = hl + L::n_weights + g * L::n_scratch;
  float* part = hl + X::part + g * (4 * 4 * H);
  const char* wlds = (const char*)hl + X::conv_off + lane * 16;
  __syncthreads();

  constexpr int NTILE = (kNLoc + 15) / 16;          // 23 location tiles
  constexpr int TPW = (NTILE + 3) / 4;               // tiles per wave (6; the last wave has 5)
  for (int pos2 = blockIdx.x * X::kGroups; pos2 < a.npos; pos2 += gridDim.x * X::kGroups) {
    const bool live = pos2 + g < a.npos;
    const int pos = live ? pos2 + g : a.npos - 1;   // an idle group recomputes the last position
    float* __restrict__ out = a.out + (size_t)pos * kOutStride;
    float* __restrict__ res = a.res ? a.res + (size_t)pos * kResultFloats : nullptr;
    const _Float16* __restrict__ xp = a.x + (size_t)pos * C * kNLoc;
    // per-lane constants of this lane's 8 channels (cout tiles 2u and 2u + 1 of a head: channels 16u' + 4q + i)
    float bsc[2][4], bsh[2][4], ownw[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bsc[u][i] = hl[L::gbn_scale + 16 * u + 4 * q + i];
        bsh[u][i] = hl[L::gbn_shift + 16 * u + 4 * q + i];
        ownw[u][i] = hl[L::own_w + 16 * u + 4 * q + i];
      }
    float gs[2][4], gm[2][4], vs[2][4], vm[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) { gs[u][i] = 0.0f; gm[u][i] = -3.0e38f; vs[u][i] = 0.0f; vm[u][i] = -3.0e38f; }
    f32x4 P[TPW][2];   // the p channels of this wave's tiles, until the pooled bias is known
    // activation fragments of one tile: k32 step s_ = channel blocks 4 s_ + q of board point 16 tile + n
    // (two tiles per pass over the weights — half the LDS fragment reads — measured slower: the second accumulator
    // set and fragment buffer spill; gpurun_out/hx_ab2.log)
    auto xload = [&](h8 (&xb)[NS], int tile) {
      int loc = tile * 16 + n;
      if (loc >= kNLoc) loc = kNLoc - 1;   // pad columns of the last tile: computed, never used
      const _Float16* src = xp + ((size_t)q * kNLoc + loc) * 8;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) xb[s_] = *(const h8*)(src + (size_t)s_ * 4 * kNLoc * 8);
    };
    h8 xb[2][NS];
    xload(xb[0], wid);
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int tile = wid + 4 * k;
      if (k + 1 < TPW) xload(xb[(k + 1) & 1], (tile + 4 < NTILE) ? tile + 4 : NTILE - 1);
      f32x4 acc[6];
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const h8*)(wlds + (ct * NS + s_) * 1024), xb[k & 1][s_], acc[ct], 0, 0, 0);
      const int loc = tile * 16 + n;
      const bool ok = tile < NTILE && loc < kNLoc;
      P[k][0] = acc[0];
      P[k][1] = acc[1];
      float ow = 0.0f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float y = mish_f(acc[2 + u][i] * bsc[u][i] + bsh[u][i]);
          const float v = acc[4 + u][i];
          if (ok) {
            gs[u][i] += y;
            gm[u][i] = fmaxf(gm[u][i], y);
            vs[u][i] += v;
            vm[u][i] = fmaxf(vm[u][i], v);
          }
          ow += v * ownw[u][i];
        }
      ow += __shfl_xor(ow, 16);
      ow += __shfl_xor(ow, 32);
      if (ok && live && q == 0) out[kOffOwnership + loc] = tanhf(ow);
    }
    // pooled g / v: over the 16 board points of a lane row, then over the waves through LDS
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          gs[u][i] += __shfl_xor(gs[u][i], m);
          gm[u][i] = fmaxf(gm[u][i], __shfl_xor(gm[u][i], m));
          vs[u][i] += __shfl_xor(vs[u][i], m);
          vm[u][i] = fmaxf(vm[u][i], __shfl_xor(vm[u][i], m));
        }
        if (n == 0) {
          const int ch = 16 * u + 4 * q + i;
          float* pw = part + wid * (4 * H);
          pw[ch] = gs[u][i];
          pw[H + ch] = gm[u][i];
          pw[2 * H + ch] = vs[u][i];
          pw[3 * H + ch] = vm[u][i];
        }
      }
    __syncthreads();
    if (t < 4 * H) {
      const int kind = t / H, ch = t - kind * H;
      const float p0 = part[kind * H + ch], p1 = part[4 * H + kind * H + ch], p2 = part[8 * H + kind * H + ch],
                  p3 = part[12 * H + kind * H + ch];
      float* dst = sc + ((kind < 2) ? L::gp : L::vp);
      if ((kind & 1) == 0) dst[ch] = (p0 + p1 + p2 + p3) * (1.0f / kNLoc);
      else dst[H + ch] = fmaxf(fmaxf(p0, p1), fmaxf(p2, p3));
    }
    __syncthreads();
    // ---- small dense layers (k_heads) ------------------------------------------------
    if (t < H) {
      float s_ = hl[L::gd_b + t];
#pragma unroll 8
      for (int k = 0; k < 2 * H; ++k) s_ += sc[L::gp + k] * hl[L::gd_w + k * H + t];
      sc[L::gbias + t] = s_;
    } else if (t >= 64 && t < 64 + V) {
      const int o = t - 64;
      float s_ = hl[L::oq_embed_b + o], gmm = hl[L::gamma_pre_b + o], b = hl[L::score_pre_b + o];
#pragma unroll 8
      for (int k = 0; k < 2 * H; ++k) {
        const float x = sc[L::vp + k];
        s_ += x * hl[L::oq_embed_w + k * V + o];
        gmm += x * hl[L::gamma_pre_w + k * V + o];
        b += x * hl[L::score_pre_w + k * V + o];
      }
      sc[L::emb + o] = mish_f(s_);
      sc[L::gpre + o] = mish_f(gmm);
      sc[L::base + o] = b;
    } else if (t == 192) {
      float s0 = pass_b0, so = opt_pass_b;
#pragma unroll 8
      for (int k = 0; k < 2 * H; ++k) {
        s0 += sc[L::gp + k] * hl[L::pass_w + k * 2];
        so += sc[L::gp + k] * hl[L::opt_pass_w + k];
      }
      sc[L::pi + 361] = s0 - 3.0f;
      sc[L::opt + 361] = so - 3.0f;
    }
    __syncthreads();
    if (t < 14) {
      float s_ = hl[L::oq_out_b + t];
#pragma unroll 8
      for (int k = 0; k < V; ++k) s_ += sc[L::emb + k] * hl[L::oq_out_w + k * 14 + t];
      if (t < 2) sc[L::misc + t] = s_;
      if (live) {
        if (t < 2) out[kOffOutcomeLogits + t] = s_;
        if (t == 5) {
          out[kOffErr2] = 4.0f / (1.0f + __expf(-s_));
          if (res) res[kOffErr2] = out[kOffErr2];
        }
      }
    } else if (t == 64) {
      float s_ = gamma_out_b;
#pragma unroll 8
      for (int k = 0; k < V; ++k) s_ += sc[L::gpre + k] * hl[L::gamma_out_w + k];
      if (live) out[kOffGamma] = s_;
      const float sp = s_ > 20.0f ? s_ : log1pf(__expf(s_));
      sc[L::misc + 2] = fminf(sp, 10.0f);
    }
    // ---- policy logits from the p channels kept in registers (needs gbias only) ------------
    {
      float gb[2][4], mw[2][4], omw[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = 16 * u + 4 * q + i;
          gb[u][i] = sc[L::gbias + c];
          mw[u][i] = hl[L::moves_w + c * 2];
          omw[u][i] = hl[L::opt_moves_w + c];
        }
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const int tile = wid + 4 * k;
        float pi = 0.0f, po = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pm = mish_f(P[k][u][i] + gb[u][i]);
            pi += pm * mw[u][i];
            po += pm * omw[u][i];
          }
        pi += __shfl_xor(pi, 16);
        pi += __shfl_xor(pi, 32);
        po += __shfl_xor(po, 16);
        po += __shfl_xor(po, 32);
        const int loc = tile * 16 + n;
        if (q == 0 && tile < NTILE && loc < kNLoc) {
          sc[L::pi + loc] = pi;
          sc[L::opt + loc] = po;
        }
      }
    }
    __syncthreads();   // misc[2] (gamma), pi, opt ready
    // ---- score logits: 800 bins x V ---------------------------------------------------
    {
      // two bins per lane and iteration, packed: 102 k mish per workgroup pass made this phase a quarter of the
      // kernel (VALU-bound; the exp2 / rcp stay scalar, everything around them is v_pk_*)
      const float gam = sc[L::misc + 2];
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int s0 = t + 512 * pr, s1 = s0 + 256;
        const f32x2 sv = {0.05f * (float)(s0 - 400) + 0.025f, 0.05f * (float)(s1 - 400) + 0.025f};
        f32x2 acc = {score_out_b, score_out_b};
        const int z = launder(0);   // keeps the LDS reads inside the bin loop
#pragma unroll 8
        for (int k = 0; k < V; ++k) {
          const float b = sc[L::base + z + k], w = hl[L::score_pre_w + (2 * H) * V + z + k], o = hl[L::score_out_w + z + k];
          acc = __builtin_elementwise_fma(mish_f2(__builtin_elementwise_fma(sv, f32x2{w, w}, f32x2{b, b})), f32x2{o, o}, acc);
        }
        sc[L::logits + s0] = gam * acc[0];
        if (s1 < 800) sc[L::logits + s1] = gam * acc[1];
      }
    }
    __syncthreads();
    // ---- raw logits out + the three softmaxes, reductions fused ------------------------
    float m3[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = t; i < 362; i += 256) {
      const float p = sc[L::pi + i], o = sc[L::opt + i];
      if (live) {
        out[kOffMoveLogits + i] = p;
        out[kOffOptLogits + i] = o;
        if (res) res[kOffMoveLogits + i] = p;
      }
      m3[0] = fmaxf(m3[0], p);
      m3[1] = fmaxf(m3[1], o);
    }
    for (int i = t; i < 800; i += 256) {
      const float l = sc[L::logits + i];
      if (live) out[kOffScoreLogits + i] = l;
      m3[2] = fmaxf(m3[2], l);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) m3[r] = wave_max(m3[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) sc[L::red + r * 4 + wid] = m3[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r)
      m3[r] = fmaxf(fmaxf(sc[L::red + r * 4], sc[L::red + r * 4 + 1]),
                    fmaxf(sc[L::red + r * 4 + 2], sc[L::red + r * 4 + 3]));
    __syncthreads();   // red is rewritten below
    float s3[3] = {0.0f, 0.0f, 0.0f};
    for (int i = t; i < 362; i += 256) {
      const float e0 = __expf(sc[L::pi + i] - m3[0]), e1 = __expf(sc[L::opt + i] - m3[1]);
      sc[L::pi + i] = e0;
      sc[L::opt + i] = e1;
      s3[0] += e0;
      s3[1] += e1;
    }
    for (int i = t; i < 800; i += 256) {
      const float e = __expf(sc[L::logits + i] - m3[2]);
      sc[L::logits + i] = e;
      s3[2] += e;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) s3[r] = wave_sum(s3[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) sc[L::red + r * 4 + wid] = s3[r];
    }
    __syncthreads();
    if (live) {
      const float i0 = 1.0f / (sc[L::red + 0] + sc[L::red + 1] + sc[L::red + 2] + sc[L::red + 3]);
      const float i1 = 1.0f / (sc[L::red + 4] + sc[L::red + 5] + sc[L::red + 6] + sc[L::red + 7]);
      const float i2 = 1.0f / (sc[L::red + 8] + sc[L::red + 9] + sc[L::red + 10] + sc[L::red + 11]);
      for (int i = t; i < 362; i += 256) {
        out[kOffMoveProbs + i] = sc[L::pi + i] * i0;
        out[kOffOptProbs + i] = sc[L::opt + i] * i1;
        if (res) {
          res[kOffMoveProbs + i] = sc[L::pi + i] * i0;
          res[kOffOptProbs + i] = sc[L::opt + i] * i1;
        }
      }
      for (int i = t; i < 800; i += 256) {
        out[kOffScoreProbs + i] = sc[L::logits + i] * i2;
        if (res) res[kOffScoreProbs + i] = sc[L::logits + i] * i2;
      }
      if (t == 0) {
        const float v0 = sc[L::misc], v1 = sc[L::misc + 1];
        const float m = fmaxf(v0, v1);
        const float e0 = __expf(v0 - m), e1 = __expf(v1 - m);
        out[kOffValueProbs] = e0 / (e0 + e1);
        out[kOffValueProbs + 1] = e1 / (e0 + e1);
        if (res) {
          res[kOffValueProbs] = e0 / (e0 + e1);
          res[kOffValueProbs + 1] = e1 / (e0 + e1);
        }
      }
    }
    __syncthreads();
  }
}

// =======================================================================================
// Host-side launchers
// =======================================================================================
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE, and the launchers are called from many host threads
// (bench.py: one driver thread per game group; an evaluation match: one engine per player, possibly on two devices):
// one flag per (kernel instantiation, device ordinal), set after the attribute call succeeded.  Racing first calls
// set the same value twice, which is harmless; nobody launches before the attribute is set on ITS device.
struct AttrOnce { std::atomic<bool> done[32]; };
template <class K>
static hipError_t ensure_lds(AttrOnce& once, K kernel, size_t lds) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32)
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (once.done[dev].load(std::memory_order_acquire)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) once.done[dev].store(true, std::memory_order_release);
  return e;
}
template <int C, int CB, int KIND, int L, int NW, bool BC>
static hipError_t launch_block_bc(const BlockArgs& a, int n_cu, hipStream_t s) {
  constexpr int NPOS = NW == 8 ? 128 / CB : 1;
  using G = Geo<NPOS, CB, 3, NW, NW == 8 ? kKMS : 2, NW == 8 ? kRingDepth : kRingDepth4>;
  constexpr size_t lds = G::ACT_BYTES + ring_bytes(CB, G::KMS, G::RD);
  static AttrOnce once;
  if (hipError_t e = ensure_lds(once, k_block<C, CB, KIND, L, NW, BC>, lds); e != hipSuccess) return e;
  const int groups = (a.npos + NPOS - 1) / NPOS, cap = NW == 8 ? n_cu : 2 * n_cu;   // NW = 4: two workgroups per CU
  hipLaunchKernelGGL((k_block<C, CB, KIND, L, NW, BC>), dim3(groups < cap ? groups : cap), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}
template <int C, int CB, int KIND, int L, int NW>
static hipError_t launch_block_t(const BlockArgs& a, int n_cu, hipStream_t s) {
  if (a.head || a.tail || a.nruns > 1) return launch_block_bc<C, CB, KIND, L, NW, true>(a, n_cu, s);
  return launch_block_bc<C, CB, KIND, L, NW, false>(a, n_cu, s);
}

// bytes of one ring macro-step of the block kernel that launch_block picks for width C
int block_macro_step_bytes(int C, bool wg8) { return C == 256 ? 128 * 32 * kKMS : (wg8 ? 64 * 32 * kKMS : 64 * 32 * 2); }

hipError_t launch_block(int C, int kind, int L, bool wg8, const BlockArgs& a, int n_cu, hipStream_t s) {
  if (C == 256 && kind == 0 && L == 3) return launch_block_t<256, 128, 0, 3, 8>(a, n_cu, s);
  if (C == 256 && kind == 0 && L == 2) return launch_block_t<256, 128, 0, 2, 8>(a, n_cu, s);
  if (C == 256 && kind == 0 && L == 1) return launch_block_t<256, 128, 0, 1, 8>(a, n_cu, s);
  if (C == 256 && kind == 1) return launch_block_t<256, 128, 1, 2, 8>(a, n_cu, s);
  if (C == 128 && wg8) {   // P3HIP_C128_WG8: the one-workgroup-per-CU form, kept for A/B timing
    if (kind == 0 && L == 3) return launch_block_t<128, 64, 0, 3, 8>(a, n_cu, s);
    if (kind == 0 && L == 2) return launch_block_t<128, 64, 0, 2, 8>(a, n_cu, s);
    if (kind == 0 && L == 1) return launch_block_t<128, 64, 0, 1, 8>(a, n_cu, s);
    if (kind == 1) return launch_block_t<128, 64, 1, 2, 8>(a, n_cu, s);
  }
  if (C == 128 && kind == 0 && L == 3) return launch_block_t<128, 64, 0, 3, 4>(a, n_cu, s);
  if (C == 128 && kind == 0 && L == 2) return launch_block_t<128, 64, 0, 2, 4>(a, n_cu, s);
  if (C == 128 && kind == 0 && L == 1) return launch_block_t<128, 64, 0, 1, 4>(a, n_cu, s);
  if (C == 128 && kind == 1) return launch_block_t<128, 64, 1, 2, 4>(a, n_cu, s);
  return hipErrorInvalidValue;
}

template <class K>
static hipError_t set_lds(K kernel, size_t lds) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t launch_init(int C, const InitArgs& a, int grid, hipStream_t s) {
  using G = Geo<1, 16, 5>;
  constexpr size_t lds = G::ACT_BYTES + ring_bytes(128) + 384 * 4;   // + the game-state bias vector
  if (C == 192) {
    constexpr size_t lds64 = G::ACT_BYTES + ring_bytes(64) + 192 * 4;
    hipLaunchKernelGGL((k_init<192, 64>), dim3(grid), dim3(kWG), lds64, s, a);
  } else if (C == 384) {
    hipLaunchKernelGGL((k_init<384>), dim3(grid), dim3(kWG), lds, s, a);
  } else if (C == 256) {
    hipLaunchKernelGGL((k_init<256>), dim3(grid), dim3(kWG), lds, s, a);
  } else if (C == 128) {
    hipLaunchKernelGGL((k_init<128>), dim3(grid), dim3(kWG), lds, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Grid of the kernels that split their NCP output passes across workgroups: a multiple of
// 8 * NCP, at most n_cu, enough for every (position group, pass) pair.
static int conv_split_grid(int npos, int npos_per_wg, int ncp, int n_cu) {
  const int unit = 8 * ncp;
  const int groups = (npos + npos_per_wg - 1) / npos_per_wg;
  int want = ((groups + 7) / 8) * unit;                 // pairs, padded to whole units
  int cap = (n_cu / unit) * unit;
  if (cap < unit) cap = unit;
  return want < cap ? want : cap;
}

template <int CIN, int COUT, bool PRE, int EPI>
static hipError_t launch_conv1x1_t(const Conv1x1Args& a, int n_cu, hipStream_t s) {
  constexpr int CB = CIN >= 256 ? 128 : 64;
  using G = Geo<128 / CB, CB, 1>;
  constexpr int CP = (COUT >= 128 && CB == 128) ? 128 : 64;
  const int grid = conv_split_grid(a.npos, 128 / CB, (COUT + CP - 1) / CP, n_cu);
  constexpr size_t lds = G::ACT_BYTES + ring_bytes(CP);
  static AttrOnce once;
  if (hipError_t e = ensure_lds(once, k_conv1x1<CIN, COUT, PRE, EPI>, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((k_conv1x1<CIN, COUT, PRE, EPI>), dim3(grid), dim3(kWG), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_conv1x1(int C, int which, const Conv1x1Args& a, int grid, hipStream_t s) {   // grid = CU count
  if (C == 192) {
    if (which == 0) return launch_conv1x1_t<192, 192, true, 0>(a, grid, s);
    if (which == 1) return launch_conv1x1_t<192, 192, false, 1>(a, grid, s);
    if (which == 2) return launch_conv1x1_t<192, 96, false, 2>(a, grid, s);
  } else if (C == 384) {
    if (which == 0) return launch_conv1x1_t<384, 384, true, 0>(a, grid, s);
    if (which == 1) return launch_conv1x1_t<384, 384, false, 1>(a, grid, s);
    if (which == 2) return launch_conv1x1_t<384, 96, false, 2>(a, grid, s);
  } else if (C == 256) {
    if (which == 0) return launch_conv1x1_t<256, 256, true, 0>(a, grid, s);
    if (which == 1) return launch_conv1x1_t<256, 256, false, 1>(a, grid, s);
    if (which == 2) return launch_conv1x1_t<256, 96, false, 2>(a, grid, s);
  } else if (C == 128) {
    if (which == 0) return launch_conv1x1_t<128, 128, true, 0>(a, grid, s);
    if (which == 1) return launch_conv1x1_t<128, 128, false, 1>(a, grid, s);
    if (which == 2) return launch_conv1x1_t<128, 96, false, 2>(a, grid, s);
  }
  return hipErrorInvalidValue;
}

template <int KW, int CIN, int COUT, bool PRE, bool ACT, bool RES, bool DUAL, int NW>
static hipError_t launch_lconv_nw(const LConvArgs& a, int n_cu, hipStream_t s) {
  constexpr int NPOS = NW == 8 ? 2 : 1;
  using G = Geo<NPOS, 64, KW, NW, (NW == 4 && KW == 3) ? 2 : kKMS>;
  const int grid = conv_split_grid(a.npos, NPOS, COUT / 64, NW == 8 ? n_cu : 2 * n_cu);   // NW = 4: two workgroups per CU
  constexpr size_t lds = G::ACT_BYTES + ring_bytes(64, G::KMS);
  static_assert(NW == 8 || 2 * lds <= 160 * 1024, "two workgroups per CU");
  static AttrOnce once;
  if (hipError_t e = ensure_lds(once, k_lconv<KW, CIN, COUT, PRE, ACT, RES, DUAL, NW>, lds); e != hipSuccess) return e;
  LConvArgs b = a;
  b.nms_total = a.nms_total * (kKMS / G::KMS);   // the host counts macro-steps of kKMS k16-steps
  static const bool turns = getenv("P3HIP_NO_PAIR_TURNS") == nullptr;
  b.pair_split = (NW == 4 && turns && grid > n_cu) ? n_cu : 0;
  hipLaunchKernelGGL((k_lconv<KW, CIN, COUT, PRE, ACT, RES, DUAL, NW>), dim3(grid), dim3(NW * 64), lds, s, b);
  return hipGetLastError();
}
template <int KW, int CIN, int COUT, bool PRE, bool ACT, bool RES, bool DUAL>
static hipError_t launch_lconv_t(const LConvArgs& a, int n_cu, hipStream_t s) {
  static const bool wg8 = getenv("P3HIP_LCONV_WG8") != nullptr;   // 8-wave workgroups, one per CU (A/B timing)
  if (wg8) return launch_lconv_nw<KW, CIN, COUT, PRE, ACT, RES, DUAL, 8>(a, n_cu, s);
  return launch_lconv_nw<KW, CIN, COUT, PRE, ACT, RES, DUAL, 4>(a, n_cu, s);
}

// the layer shapes of the C=384 / C_b=192 btl and nbt blocks and of the C=192 classic blocks
// (engine.cpp build_plan); f = pre, act, res, dual
hipError_t launch_lconv(int kw, int cin, int cout, const LConvArgs& a, int grid, hipStream_t s) {   // grid = CU count
  const int f = (a.pre ? 8 : 0) | (a.act ? 4 : 0) | (a.res ? 2 : 0) | (a.dual ? 1 : 0);
#define P3_LCONV(KW, CIN, COUT, PRE, ACT, RES, DUAL) \
  if (kw == KW && cin == CIN && cout == COUT && f == ((PRE ? 8 : 0) | (ACT ? 4 : 0) | (RES ? 2 : 0) | (DUAL ? 1 : 0))) \
    return launch_lconv_t<KW, CIN, COUT, PRE, ACT, RES, DUAL>(a, grid, s);
  P3_LCONV(1, 384, 192, true, true, false, false)     // btl reduce from the raw stream
  P3_LCONV(1, 384, 192, false, true, false, false)    // btl reduce from the activated copy
  P3_LCONV(1, 384, 192, true, false, false, true)     // nbt reduce (t raw + act1(t))
  P3_LCONV(1, 384, 192, false, false, false, true)
  P3_LCONV(3, 192, 192, false, true, false, false)    // inner conv, activated output
  P3_LCONV(3, 192, 192, true, true, false, false)     // classic conv0 from the raw stream
  P3_LCONV(3, 192, 192, false, false, true, false)    // classic conv1 (+x), last block
  P3_LCONV(3, 192, 192, false, false, true, true)     // nbt conv2 / conv4 (t' raw + act(t')), classic conv1 (+x, + next act)
  P3_LCONV(1, 192, 384, false, false, true, false)    // expand + x
  P3_LCONV(1, 192, 384, false, false, true, true)     // expand + x, + the next block's activated input
#undef P3_LCONV
  return hipErrorInvalidValue;
}

hipError_t launch_bdense(int C, const BDenseArgs& a, int grid, hipStream_t s) {
  constexpr size_t lds = 128 * kTtStride + ring_bytes(128) + (384 + 2 * 384) * 4;   // + epilogue parameters
  static AttrOnce once[4];
  {
    hipError_t e = C == 256 ? ensure_lds(once[0], k_bdense<256>, lds) : C == 128 ? ensure_lds(once[1], k_bdense<128>, lds)
                 : C == 384 ? ensure_lds(once[2], k_bdense<384>, lds) : C == 192 ? ensure_lds(once[3], k_bdense<192>, lds)
                 : hipErrorInvalidValue;
    if (e != hipSuccess) return e;
  }
  if (C == 192) {
    hipLaunchKernelGGL((k_bdense<192>), dim3(grid), dim3(kWG), lds, s, a);
  } else if (C == 384) {
    hipLaunchKernelGGL((k_bdense<384>), dim3(grid), dim3(kWG), lds, s, a);
  } else if (C == 256) {
    hipLaunchKernelGGL((k_bdense<256>), dim3(grid), dim3(kWG), lds, s, a);
  } else if (C == 128) {
    hipLaunchKernelGGL((k_bdense<128>), dim3(grid), dim3(kWG), lds, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int V>
static hipError_t launch_heads_t(const HeadsArgs& a, hipStream_t s) {
  using L = HeadsLds<32, V>;
  static AttrOnce once;
  if (hipError_t e = ensure_lds(once, k_heads<32, V>, L::bytes); e != hipSuccess) return e;
  const int grid = (a.npos + L::kGroups - 1) / L::kGroups;
  hipLaunchKernelGGL((k_heads<32, V>), dim3(grid), dim3(1024), L::bytes, s, a);
  return hipGetLastError();
}

template <int C, int V>
static hipError_t launch_headsx_t(const HeadsArgs& a, int n_cu, hipStream_t s) {
  using X = HeadsxLds<C, 32, V>;
  static_assert(X::bytes <= 160 * 1024, "LDS");
  static AttrOnce once;
  if (hipError_t e = ensure_lds(once, k_headsx<C, 32, V>, X::bytes); e != hipSuccess) return e;
  const int groups = (a.npos + X::kGroups - 1) / X::kGroups;
  hipLaunchKernelGGL((k_headsx<C, 32, V>), dim3(groups < n_cu ? groups : n_cu), dim3(512), X::bytes, s, a);
  return hipGetLastError();
}

// heads with their convs inside (a.x, a.conv_a set): C in {128, 256}, V in {32, 48, 64}
bool heads_fusable(int C, int V) { return (C == 128 || C == 256) && (V == 32 || V == 48 || V == 64); }
hipError_t launch_headsx(int C, const HeadsArgs& a, int n_cu, hipStream_t s) {
  if (C == 256 && a.V == 64) return launch_headsx_t<256, 64>(a, n_cu, s);
  if (C == 256 && a.V == 48) return launch_headsx_t<256, 48>(a, n_cu, s);
  if (C == 256 && a.V == 32) return launch_headsx_t<256, 32>(a, n_cu, s);
  if (C == 128 && a.V == 64) return launch_headsx_t<128, 64>(a, n_cu, s);
  if (C == 128 && a.V == 48) return launch_headsx_t<128, 48>(a, n_cu, s);
  if (C == 128 && a.V == 32) return launch_headsx_t<128, 32>(a, n_cu, s);
  return hipErrorInvalidValue;
}

hipError_t launch_heads(const HeadsArgs& a, int grid, hipStream_t s) {
  (void)grid;
  if (a.V == 64) return launch_heads_t<64>(a, s);
  if (a.V == 80) return launch_heads_t<80>(a, s);
  if (a.V == 48) return launch_heads_t<48>(a, s);
  if (a.V == 32) return launch_heads_t<32>(a, s);
  return hipErrorInvalidValue;
}

// =======================================================================================
// On-device NN cache (the reference caches NNInferResults per thread on the
// host, cc/nn/nn_interface.cc:107-132, cc/core/lru_cache.h).  Plain HBM byte work: a probe of
// kCacheWays consecutive entries per key, record copies of kOutStride floats.
// =======================================================================================
__global__ void __launch_bounds__(256) k_cache_probe(CacheArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n) return;
  const unsigned long long lo = a.keys[r].lo, hi = a.keys[r].hi;
  int hit = -1, victim = -1;
  if (lo | hi) {
    unsigned best_age = 0;
    bool have = false;
    const unsigned h = (unsigned)(lo ^ (lo >> 32));
#pragma unroll 1
    for (int w = 0; w < kCacheWays; ++w) {
      const unsigned i = (h + (unsigned)w) & a.mask;
      const unsigned long long tl = a.tkeys[2 * (size_t)i], th = a.tkeys[2 * (size_t)i + 1];
      if (tl == lo && th == hi) { hit = (int)i; break; }
      // replacement: an empty entry first, else the least recently used of the window
      const unsigned age = (tl | th) ? (a.run - (a.tmeta[i] >> 3)) & 0x1fffffffu : 0xffffffffu;
      if (!have || age > best_age) { have = true; best_age = age; victim = (int)i; }
    }
    if (hit >= 0) victim = -1;
  }
  a.hit[r] = hit;
  a.victim[r] = victim;
}

// feats_out[j] = feats_in[rows[j]] (1,860 bytes = 465 dwords per position)
__global__ void __launch_bounds__(256) k_cache_gather(CacheArgs a) {
  const int j = blockIdx.x;
  const unsigned* src = (const unsigned*)(a.feats_in + (size_t)a.rows[j] * FeatOff::size);
  unsigned* dst = (unsigned*)(a.feats_out + (size_t)j * FeatOff::size);
  for (int i = threadIdx.x; i < FeatOff::size / 4; i += 256) dst[i] = src[i];
}

// out[out_row0 + j] = tvals[idx[j]]; the entry is marked used in this run
__global__ void __launch_bounds__(256) k_cache_fill(CacheArgs a) {
  const int j = blockIdx.x;
  const int e = a.idx[j];
  const f32x4* src = (const f32x4*)(a.tvals + (size_t)e * kOutStride);
  f32x4* dst = (f32x4*)(a.out + (size_t)(a.out_row0 + j) * kOutStride);
  static_assert(kOutStride % 4 == 0, "records are copied in 16-byte pieces");
  for (int i = threadIdx.x; i < kOutStride / 4; i += 256) dst[i] = src[i];
  if (threadIdx.x == 0) {
    const unsigned meta = a.tmeta[e];
    a.out_sym[a.out_row0 + j] = meta & 7u;
    a.tmeta[e] = (meta & 7u) | (a.run << 3);
  }
}

// tvals[idx[j]] = out[rows[j]], key and symmetry of row rows[j]; the host lists every entry at most once
__global__ void __launch_bounds__(256) k_cache_insert(CacheArgs a) {
  const int j = blockIdx.x;
  const int e = a.idx[j], r = a.rows[j];
  const f32x4* src = (const f32x4*)(a.out + (size_t)a.src[j] * kOutStride);
  f32x4* dst = (f32x4*)(a.tvals + (size_t)e * kOutStride);
  for (int i = threadIdx.x; i < kOutStride / 4; i += 256) dst[i] = src[i];
  if (threadIdx.x == 0) {
    a.tkeys[2 * (size_t)e] = a.keys[r].lo;
    a.tkeys[2 * (size_t)e + 1] = a.keys[r].hi;
    a.tmeta[e] = ((unsigned)a.keys[r].sym & 7u) | (a.run << 3);
  }
}

hipError_t launch_cache_probe(const CacheArgs& a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cache_probe, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_cache_gather(const CacheArgs& a, hipStream_t s) {
  if (a.m <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cache_gather, dim3(a.m), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_cache_fill(const CacheArgs& a, hipStream_t s) {
  if (a.m <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cache_fill, dim3(a.m), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_cache_insert(const CacheArgs& a, hipStream_t s) {
  if (a.m <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_cache_insert, dim3(a.m), dim3(256), 0, s, a);
  return hipGetLastError();
}

const char* block_kernel_name(int C, int kind, int L) {
  (void)L;
  if (kind == 0) return C == 256 ? "k_block<256,128,btl>" : "k_block<128,64,btl>";
  return C == 256 ? "k_block<256,128,nbt>" : "k_block<128,64,nbt>";
}

}  // namespace p3
