// kernels.h — argument structs and launchers shared by kernels.hip and engine.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace p3 {

// Per-position fp32 output record written by k_heads.
//   [result region, copied to the host every run: 7,556 B]
constexpr int kOffMoveLogits = 0;      // 362  -> NNInferResult::move_logits
constexpr int kOffMoveProbs = 362;     // 362  -> move_probs
constexpr int kOffValueProbs = 724;    // 2    -> value_probs
constexpr int kOffScoreProbs = 726;    // 800  -> score_probs
constexpr int kOffOptProbs = 1526;     // 362  -> opt_move_probs
constexpr int kOffErr2 = 1888;         // 1    -> err2_outcome
constexpr int kResultFloats = 1889;
//   [debug / optional region, copied on demand]
constexpr int kOffOptLogits = 1889;      // 362
constexpr int kOffOutcomeLogits = 2251;  // 2
constexpr int kOffScoreLogits = 2253;    // 800
constexpr int kOffOwnership = 3053;      // 361
constexpr int kOffGamma = 3414;          // 1
constexpr int kOutStride = 3416;

constexpr int kMaxBlockLayers = 8;

// One launch runs `nblk` consecutive residual blocks of the same shape: positions are private
// to a workgroup, so block b+1 of a position only needs block b of the SAME workgroup — no
// grid-wide dependency, just the launch boundaries (ramp, tail, cold ring) saved.
constexpr int kMaxFuse = 6;
// Joined launches (C = 256 btl, broadcast dense fused): up to kMaxRuns runs of blocks with the broadcast blocks
// between them inside ONE launch — a position goes through the whole trunk in its workgroup.
constexpr int kMaxRuns = 4, kMaxLaunchBlocks = 12;
struct BlockParams {
  const float* scale[kMaxBlockLayers];  // folded BN of conv j's prologue
  const float* shift[kMaxBlockLayers];
};
struct BlockArgs {
  _Float16* x;          // residual stream, updated in place
  _Float16* t;          // nbt: scratch for the inner residual stream [pos][CB/8][361][8]
  int npos;
  int nblk;
  const void* wstream;  // the packed weight streams of the launch's blocks, back to back
  int nms_total;        // their total length in macro-steps (the ring walks them circularly)
  BlockParams blk[kMaxLaunchBlocks];   // all runs' blocks, in order
  int nruns;            // >= 1; > 1: joined launch, run r has run_nblk[r] blocks (nblk is unused)
  int run_nblk[kMaxRuns];
  // The 1x1 convs of the broadcast blocks next to the run ride in the same launch (k_block's BC
  // form): `head` = conv_last of the broadcast block BEFORE the run (x += W . zin, zin = k_bdense's
  // output, its stream first in wstream), `tail` = conv_first of the broadcast block AFTER it
  // (tout = mish(W . mish(bn0(x))), bn0 = tail_scale/shift, its stream last in wstream).
  int head, tail;
  const _Float16* zin;    // read by the head of an odd run (and of run 0: one launch per run reads zin, writes uout)
  const _Float16* zin2;   // ... of an even run > 0 (joined launches: what the tail of the odd run before it wrote)
  _Float16* tout;
  const float* tail_scale[kMaxRuns];   // [r]: of the broadcast block that ends run r
  const float* tail_shift[kMaxRuns];
  // tail_dense (C = 256 only): the broadcast block's dense and its bn1 + mish run in the tail as well — u goes to
  // uout, nothing to tout; the tail's stream is [conv_first pass 0][dense][conv_first pass 1][dense], the dense
  // matrix packed over the act buffer's 384 padded board rows (engine.cpp)
  int tail_dense;
  _Float16* uout;    // written by the tail of an even run
  _Float16* uout2;   // ... of an odd run (joined launches alternate)
  const float* dense_bias[kMaxRuns];    // [361]
  const float* dense_scale[kMaxRuns];   // folded bn1 [C]
  const float* dense_shift[kMaxRuns];
  // start-up stagger (shader-clock cycles per step, 0 = none): workgroup b begins (b / 8) % 8 steps late,
  // so the CUs of an XCD are not all in their HBM-bound phases (head / tail / residual traffic) at once
  int stagger;
  // 4-wave form, two workgroups per CU (grid = 2 x CUs: blocks b and b + grid/2 share a CU): the SIMD arbiter
  // serves the older workgroup's waves first, so at equal priority the second workgroup of every CU falls behind
  // and finishes its positions alone on the CU.  With pair_turns the two take turns at the higher wave priority,
  // one position each (the second workgroup first), and finish together.
  int pair_turns;
#ifdef P3_DIAG
  // diagnostic build only (make diag): lane 0 of every wave of workgroups 0..7 stores s_memtime at
  // the phase boundaries of its second position: stamps[((wg * 8 + wave) * 8 + section) * 32 + k],
  // section = block index (0..5), 6 = head, 7 = tail.  Never read by the kernel.
  unsigned long long* stamps;
  int stamp_run;        // joined launches: the run whose phases are stamped
  // spans[wg * 16 + k], every workgroup (up to 512), wave 0: k = 0 kernel entry, 1 ring ready, 2 + p end of
  // the workgroup's position p (p < 5), 7 exit in shader-clock ticks (s_memtime; not comparable across XCDs);
  // [8 + k] the same instants on the 100 MHz device-wide counter (s_memrealtime)
  unsigned long long* spans;
#endif
};
#ifdef P3_DIAG
constexpr int kStampWgs = 8, kStampSections = 8, kStampSlots = 32, kSpanWgs = 512, kSpanSlots = 16;
#endif

struct InitArgs {
  const void* feats;  // npos x p3hip_features (1860 B each)
  _Float16* x;
  int npos;
  const void* wstream;
  int nms_total;
  const float* game_w;  // [8][C]
  const float* game_b;  // [C]
};

struct Conv1x1Args {
  const _Float16* in;
  _Float16* out16;
  float* out32;
  int npos;
  const void* wstream;
  int nms_total;
  const float* scale;
  const float* shift;
};

// One conv layer of the layer-wise path (k_lconv): y = conv(in') [+ y], in' = mish(bn_in(in))
// when pre != 0; act: mish(bn_out(y)) stored instead of y; dual: y stored to `out` AND
// mish(bn_out(y)) to `out2` (the consumer's prologue applied once by the producer).
struct LConvArgs {
  const _Float16* in;   // [npos][CIN/8][361][8]
  _Float16* out;        // [npos][COUT/8][361][8]; read as the residual when res != 0
  _Float16* out2;       // dual != 0: mish(bn_out(y)) goes here, the raw y (after the residual add) to `out`
  int npos;
  const void* wstream;
  int nms_total;
  int pre, act, res, dual;
  const float *scale_in, *shift_in;    // folded BN of the prologue   [CIN]
  const float *scale_out, *shift_out;  // folded BN of the epilogue   [COUT]
  int pair_split;   // 4-wave form: blocks >= pair_split are the second workgroup of their CU (0: no priority turns)
};

struct BDenseArgs {
  const _Float16* t;
  _Float16* u;
  int npos;
  const void* wstream;
  int nms_total;
  const float* bias;   // dense bias [361]
  const float* scale;  // folded bn1 [C]
  const float* shift;
};

// k_headsx stages the small head tensors as ONE image, in HeadsLds' order (kernels.hip): gd_w [2H][H], oq_embed_w
// [2H][V], gamma_pre_w [2H][V], score_pre_w [2H+1][V], oq_out_w [V][14], gamma_out_w [V], score_out_w [V], pass_w
// [2H][2], opt_pass_w [2H], moves_w [H][2], opt_moves_w [H], own_w [H], gbn_scale [H], gbn_shift [H], gd_b [H],
// oq_embed_b [V], gamma_pre_b [V], score_pre_b [V], oq_out_b [14 + 2 pad]
constexpr int heads_image_floats(int H, int V) {
  return 2 * H * H + 2 * H * V + 2 * H * V + (2 * H + 1) * V + V * 14 + V + V + 4 * H + 2 * H + 2 * H + H + H + H + H + H + V + V + V + 16;
}

struct HeadsArgs {
  const float* hp;  // [npos][96 / 4][361][4]: channel quads, as k_conv1x1 (EPI 2) stores its accumulators
  // k_headsx (the head convs inside): the residual stream and the 96 x C conv weights as MFMA A fragments
  const _Float16* x;        // [npos][C/8][361][8]
  const void* conv_a;       // [6 cout tiles][C/32 k32 steps][64 lanes][8] fp16
  const float* image;       // heads_image_floats(32, V) floats, 16-byte aligned
  float* out;       // [npos][kOutStride]
  // p3hip_run's result records (the first kResultFloats of a row) written a second time into a dense buffer, so that the
  // D2H copy of TrtEngineImpl::RunInference (trt_engine.cc:283-297) is one contiguous transfer; null: d_out only
  float* res;       // [npos][kResultFloats]
  int npos;
  int V;
  const float *gbn_scale, *gbn_shift;      // policy.gpool_bn folded [32]
  const float *gd_w, *gd_b;                // policy.gpool_dense [64][32], [32]
  const float *moves_w;                    // policy.out_moves [32][2]
  const float *pass_w, *pass_b;            // policy.out_pass [64][2], [2]
  const float *opt_moves_w;                // [32]
  const float *opt_pass_w, *opt_pass_b;    // [64], [1]
  const float *oq_embed_w, *oq_embed_b;    // [64][V], [V]
  const float *oq_out_w, *oq_out_b;        // [V][14], [14]
  const float *own_w;                      // [32]
  const float *gamma_pre_w, *gamma_pre_b;  // [64][V], [V]
  const float *gamma_out_w, *gamma_out_b;  // [V], [1]
  const float *score_pre_w, *score_pre_b;  // [65][V], [V]
  const float *score_out_w, *score_out_b;  // [V], [1]
};

// picks its own grid (one workgroup per CU, or two 4-wave workgroups per CU at C = 128 unless wg8)
hipError_t launch_block(int C, int kind, int L, bool wg8, const BlockArgs& a, int n_cu, hipStream_t s);
int block_macro_step_bytes(int C, bool wg8);
hipError_t launch_init(int C, const InitArgs& a, int grid, hipStream_t s);
// which: 0 = broadcast conv_first (bn+mish prologue, mish epilogue), 1 = broadcast
// conv_last (+residual), 2 = head convs (fp32 out, COUT = 96)
hipError_t launch_conv1x1(int C, int which, const Conv1x1Args& a, int n_cu, hipStream_t s);   // picks its own grid
hipError_t launch_bdense(int C, const BDenseArgs& a, int grid, hipStream_t s);
// layer-wise conv: (kw, cin, cout) in {(1,384,192), (3,192,192), (1,192,384)}; two positions per workgroup
hipError_t launch_lconv(int kw, int cin, int cout, const LConvArgs& a, int n_cu, hipStream_t s);
hipError_t launch_heads(const HeadsArgs& a, int grid, hipStream_t s);
bool heads_fusable(int C, int V);
hipError_t launch_headsx(int C, const HeadsArgs& a, int n_cu, hipStream_t s);

// ---- on-device NN cache (engine.cpp p3hip_cache_*) -------------------------------------------------
// Open-addressed table in HBM: 128-bit keys, kCacheWays consecutive entries probed per key, one result
// record (kOutStride floats, the layout k_heads writes) per entry.  meta = symmetry (3 bits) | last-use
// run counter << 3.  A zero key is "no key": never looked up, never stored.
constexpr int kCacheWays = 8;
struct CacheKey { unsigned long long lo, hi, sym; };
struct CacheArgs {
  const CacheKey* keys;            // [n] keys of this run's rows
  int n;
  unsigned long long* tkeys;       // [cap][2]
  unsigned* tmeta;                 // [cap]
  float* tvals;                    // [cap][kOutStride]
  unsigned mask;                   // cap - 1
  unsigned run;                    // run counter (for replacement)
  int* hit;                        // [n] entry index, -1 = miss
  int* victim;                     // [n] entry a miss may take (-1: none, key 0)
  // fill / insert / gather lists (device copies of what the host decided after the probe)
  const int* rows;                 // [m] row of this run (gather: its features; insert: its key)
  const int* src;                  // [m] insert: row of `out` that holds the result
  const int* idx;                  // [m] table entry
  int m;
  int out_row0;                    // fill: first destination row of d_out
  float* out;                      // [batch][kOutStride]
  unsigned* out_sym;               // fill: [n] symmetry of the stored result, by destination row
  const unsigned char* feats_in;   // gather
  unsigned char* feats_out;
};
hipError_t launch_cache_probe(const CacheArgs& a, hipStream_t s);
hipError_t launch_cache_gather(const CacheArgs& a, hipStream_t s);
hipError_t launch_cache_fill(const CacheArgs& a, hipStream_t s);
hipError_t launch_cache_insert(const CacheArgs& a, hipStream_t s);
const char* block_kernel_name(int C, int kind, int L);

}  // namespace p3
