// engine.cpp — the p3hip C ABI (include/p3hip.h): weight loading / repacking, pinned
// staging, slot compaction, launch sequence.  Compiled with hipcc into libp3hip.so.
//
// Replaces TrtEngineImpl (cc/nn/engine/trt_engine.cc:85-351) behind nn::Engine
// (cc/nn/engine/engine.h:22-43).  Differences by design:
//   * the GoFeatures POD itself (1,860 B) is what crosses PCIe; planes are expanded on
//     the device (go_features.cc:10-61 restated in k_init) instead of 21,692 B of fp32;
//   * only slots loaded since the previous run are uploaded and evaluated, compacted
//     into a dense batch (the TRT engine always runs the full static batch);
//   * only the consumed outputs (7,556 B / position) come back every run; ownership and
//     raw logits stay on the device until asked for.
#include <hip/hip_runtime.h>
#include <chrono>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <unordered_set>
#include <string>
#include <vector>

#include "../../include/p3hip.h"
#include "kernels.h"
#include "slot_state.h"

namespace {

constexpr float kBnEps = 1e-3f;  // model.py:231
constexpr int kNLoc = 361;
constexpr size_t kFeatBytes = sizeof(p3hip_features);
static_assert(sizeof(p3hip_features) == 1860, "p3hip_features layout");
static_assert(sizeof(p3hip_result) == 4 * 1892, "p3hip_result layout");

thread_local std::string g_create_error;

struct Tensor {
  std::vector<int> dims;
  const float* data;
  size_t size() const {
    size_t n = 1;
    for (int d : dims) n *= d;
    return n;
  }
};

struct WeightFile {
  int version = 0, nblocks = 0, C = 0, Cb = 0, H = 0, V = 0, bint = 0, inner = 0, btype = 0;
  std::vector<float> data;
  std::map<std::string, Tensor> tensors;

  bool load(const char* path, std::string& err) {
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    char magic[4];
    int hdr[10];
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "P3W1", 4) != 0 || fread(hdr, 4, 10, f) != 10) {
      err = "not a .p3w file"; fclose(f); return false;
    }
    version = hdr[0]; nblocks = hdr[1]; C = hdr[2]; Cb = hdr[3]; H = hdr[4]; V = hdr[5];
    bint = hdr[6]; inner = hdr[7]; btype = hdr[8];
    int nt = hdr[9];
    if (nt < 1 || nt > 8192 || nblocks < 1 || nblocks > 256 || bint < 1) {
      err = "implausible .p3w header"; fclose(f); return false;
    }
    struct Ent { char name[48]; int ndim; int dims[4]; long long off; };
    std::vector<Ent> ents(nt);
    long long total = 0;
    for (auto& e : ents) {
      if (fread(e.name, 1, 48, f) != 48 || fread(&e.ndim, 4, 1, f) != 1 ||
          fread(e.dims, 4, 4, f) != 4 || fread(&e.off, 8, 1, f) != 1) {
        err = "truncated tensor table"; fclose(f); return false;
      }
      long long sz = 1;
      bool ok = e.ndim >= 0 && e.ndim <= 4 && e.off >= 0 && e.off < (1ll << 31);
      for (int d = 0; ok && d < e.ndim; ++d) {
        ok = e.dims[d] > 0 && e.dims[d] < (1 << 24);
        sz *= e.dims[d];
        ok = ok && sz < (1ll << 31);
      }
      if (!ok) { err = "corrupt tensor table"; fclose(f); return false; }
      if (e.off + sz > total) total = e.off + sz;
    }
    long pos = ftell(f);
    pos += (64 - pos % 64) % 64;
    fseek(f, pos, SEEK_SET);
    data.resize(total);
    if (fread(data.data(), 4, total, f) != (size_t)total) { err = "truncated data"; fclose(f); return false; }
    fclose(f);
    for (auto& e : ents) {
      Tensor t;
      t.dims.assign(e.dims, e.dims + e.ndim);
      t.data = data.data() + e.off;
      tensors[std::string(e.name, strnlen(e.name, sizeof e.name))] = t;
    }
    return true;
  }
  // A missing or mis-shaped tensor (truncated / foreign file) is recorded and answered with a
  // zero tensor of the expected size; build_plan checks `missing` once at the end and
  // p3hip_create fails with the list — the library never aborts the host process.
  mutable std::string missing;
  mutable std::vector<std::vector<float>> zeros;
  mutable std::map<std::string, Tensor> stand_ins;
  const Tensor& get(const std::string& n, size_t expect = 0) const {
    auto it = tensors.find(n);
    if (it != tensors.end() && (expect == 0 || it->second.size() == expect)) return it->second;
    if (missing.size() < 400) missing += (missing.empty() ? "" : ", ") + n + (it == tensors.end() ? "" : " (wrong size)");
    auto st = stand_ins.find(n);
    if (st != stand_ins.end()) return st->second;
    zeros.emplace_back(expect ? expect : 1, 0.0f);
    Tensor t;
    t.dims = {(int)zeros.back().size()};
    t.data = zeros.back().data();
    return stand_ins[n] = t;
  }
  bool is_broadcast(int i) const { return i % bint == bint - 1; }  // model.py:1002
};

// ---- device arena -----------------------------------------------------------------
struct Arena {
  std::vector<unsigned char> host;
  bool bad_stream = false;
  size_t add(const void* p, size_t bytes) {
    size_t off = (host.size() + 255) & ~size_t(255);
    host.resize(off + bytes);
    memcpy(host.data() + off, p, bytes);
    return off;
  }
};

// k16 blocks [h(2)][CP couts][8] fp16 in (tap major, channel-pair minor) order; see
// conv_segment in conv_core.h.  W is HWIO flattened as [taps][cin_total][cout_total].
void pack_segment(std::vector<_Float16>& dst, const float* W, int taps, int ntaps_pad,
                  int cin_total, int cout_total, int cin0, int CB, int cout0, int CP) {
  for (int tap = 0; tap < ntaps_pad; ++tap)
    for (int q = 0; q < CB / 16; ++q)
      for (int h = 0; h < 2; ++h)
        for (int co = 0; co < CP; ++co)
          for (int e = 0; e < 8; ++e) {
            int ci = cin0 + q * 16 + h * 8 + e, c = cout0 + co;
            float v = 0.0f;
            if (tap < taps && ci < cin_total && c < cout_total)
              v = W[((size_t)tap * cin_total + ci) * cout_total + c];
            dst.push_back((_Float16)v);
          }
}

// 3x3 weights of the fused block kernel: k16 blocks in (kernel row, k32 index, kernel column)
// order — the three taps of a kernel row share their activation fragments (conv16.h
// conv_segment16_3x3), so a step advances the column before the channel slice.
void pack_segment_3x3(std::vector<_Float16>& dst, const float* W, int cin_total, int cout_total, int CB, int CP) {
  for (int ky = 0; ky < 3; ++ky)
    for (int q32 = 0; q32 < CB / 32; ++q32)
      for (int kx = 0; kx < 3; ++kx)
        for (int q = 2 * q32; q < 2 * q32 + 2; ++q)
          for (int h = 0; h < 2; ++h)
            for (int co = 0; co < CP; ++co)
              for (int e = 0; e < 8; ++e) {
                const int ci = q * 16 + h * 8 + e, tap = ky * 3 + kx;
                float v = 0.0f;
                if (ci < cin_total && co < cout_total) v = W[((size_t)tap * cin_total + ci) * cout_total + co];
                dst.push_back((_Float16)v);
              }
}

// k_blockw's weight granule (csrc/asm/blockw_gen.py): 64 output channels x 32 input channels of one tap as four
// MFMA 32x32x16 A fragments, [k16 half j][cout tile c][h][n][8] = W[tap][k0 + 16 j + 8 h + e][cout0 + 32 c + n]: lane
// (n, h) of fragment (j, c) reads its 16 bytes at (2 j + c) * 1024 + lane * 16.  scale (may be null): the folded BN scale
// of the layer that FOLLOWS the conv, times log2(e), per output channel — multiplied in before the one fp16 rounding.
void pack_granule(std::vector<_Float16>& dst, const float* W, int cin_total, int cout_total, int tap, int k0, int cout0,
                  const float* scale) {
  for (int j = 0; j < 2; ++j)
    for (int c = 0; c < 2; ++c)
      for (int h = 0; h < 2; ++h)
        for (int n = 0; n < 32; ++n)
          for (int el = 0; el < 8; ++el) {
            const int co = cout0 + 32 * c + n;
            float v = W[((size_t)tap * cin_total + k0 + 16 * j + 8 * h + el) * cout_total + co];
            if (scale) v *= scale[co];
            dst.push_back((_Float16)v);
          }
}

struct FoldedBN { size_t scale_off, shift_off; };

// One conv launch of a layer-wise block (kind 4).  Regions: 0 = x; 1, 2 = the two C_b-channel
// halves of the scratch buffer t; 3, 4 = those of u (3 also names u as a whole C-channel buffer).
struct LayerPlan {
  int kw, cin, cout;
  bool pre, act, res, dual;   // see p3::LConvArgs
  FoldedBN pre_bn, out_bn;    // prologue / epilogue BN (epilogue: of act or of dual's second output)
  int in_buf, out_buf, out2_buf;
  size_t stream_off = 0;
  int nms = 0;
};

struct BlockPlan {
  int kind;  // 0 btl, 1 nbt, 3 broadcast, 4 layer-wise (btl/nbt at widths the fused kernel lacks)
  std::vector<LayerPlan> layers;
  size_t stream_off = 0;
  int nms = 0;
  size_t stream_bytes = 0;   // fused blocks: the launch picks the macro-step size (kernels.h block_macro_step_bytes)
  FoldedBN bn[p3::kMaxBlockLayers];
  // broadcast extras
  size_t stream2_off = 0, stream3_off = 0;
  int nms2 = 0, nms3 = 0;
  size_t dense_bias_off = 0;
  // Broadcast 1x1 convs taken into the neighbouring block launches (k_block's BC form).
  //   on a broadcast block: its conv_first runs at the tail of the launch before it / its conv_last
  //   at the head of the launch after it;
  //   on a fused block: the run's first block carries the head conv (head_of = that broadcast block,
  //   its stream lies right before stream_off), the run's last block the tail conv (right after).
  bool first_fused = false, last_fused = false;
  bool dense_fused = false;   // C = 256: the dense runs in that tail as well (no k_bdense launch, t never stored)
  int head_of = -1, tail_of = -1;
  size_t head_bytes = 0, tail_bytes = 0;
};

}  // namespace

struct p3hip_engine {
  std::string path, err;
  int batch = 0, device = 0;
  uint32_t flags = 0;
  WeightFile wf;
  int n_cu = 256;
  bool c128_wg8 = false;   // P3HIP_C128_WG8: C = 128 blocks as one 8-wave workgroup per CU (A/B timing)
  bool runs_contiguous = false;   // build_plan laid every run's streams back to back (joined launches possible)
  bool bcast_fuse = true;  // P3HIP_NO_BFUSE clears it: broadcast 1x1 convs as their own launches (A/B, tests)
  // k_blockw (csrc/asm/blockw_gen.py): the runs of C = 256 btl blocks by the hand-scheduled one-wave-per-SIMD kernel
  bool blockw = false;
  struct BlockwRun { size_t first; int nblk; size_t stream_off, prm_off; };
  std::vector<BlockwRun> bw_runs;
  hipModule_t bw_mod = nullptr;
  hipFunction_t bw_fn = nullptr;
  unsigned long long* d_bw_stamps = nullptr;   // P3HIP_BLOCKW_DIAG: s_memtime stamps of the _diag kernel
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // P3HIP_FLAG_LAUNCH_GRAPH: the forward pass over the full static batch, captured once (trt_engine.cc:260-303)
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  bool graph_failed = false, graph_warm = false;
  const unsigned char* graph_feats = nullptr;   // the feature buffer the captured k_init reads (kernel arguments are baked in)
  bool graph_direct = false;                    // ... and whether the captured heads kernel writes the host result buffer
  // p3hip_time_trunk_kernel: event pairs around every fused-block launch of a forward pass
  std::vector<hipEvent_t> blk_ev;
  bool time_blocks = false;
  int timed_blocks = 0;

  unsigned char* d_arena = nullptr;
  std::vector<BlockPlan> blocks;
  size_t init_stream_off = 0; int init_nms = 0;
  size_t game_w_off = 0, game_b_off = 0;
  size_t heads_stream_off = 0; int heads_nms = 0;
  size_t heads_conv_a_off = 0, heads_image_off = 0;
  bool heads_fused = false;   // k_headsx: the head convs inside the heads kernel (C <= 256; P3HIP_NO_HFUSE clears it)
  std::map<std::string, size_t> head_off;

  // buffers
  unsigned char* h_feats = nullptr;       // pinned [batch] slots as loaded
  unsigned char* h_feats_compact = nullptr;  // pinned, dense
  unsigned char* d_feats = nullptr;
  _Float16 *d_x = nullptr, *d_t = nullptr, *d_u = nullptr;
#ifdef P3_DIAG
  unsigned long long* d_stamps = nullptr;   // diagnostic build: k_block phase stamps of one launch (P3DIAG_LAUNCH)
  unsigned long long* d_spans = nullptr;    // and every workgroup's entry / per-position / exit times of that launch
  int launch_index = 0;
#endif
  _Float16* d_s = nullptr;   // nbt trunks: the block kernel's inner-stream scratch (t and u carry the broadcast blocks' tensors)
  float* d_hp = nullptr;
  float* d_out = nullptr;
  float* h_out = nullptr;  // pinned [batch][kResultFloats]
  float* d_res = nullptr;  // [batch][kResultFloats] dense: the heads kernel writes the result records a second time there
                           // (run_direct), so that the D2H copy is ONE contiguous transfer instead of a strided one
  bool run_direct = false; // this run's heads kernel fills d_res
  double t_h2d = 0, t_fwd = 0, t_d2h = 0;   // P3HIP_TIME_RUN
  long t_runs = 0;
  bool feats_identity = false;   // gather_loaded: every slot was dirty, row == slot: the upload comes straight from h_feats
  p3::SlotStates slots;   // dirty flags + slot -> dense row of the last run (slot_state.h)
  int last_n = 0;
  std::vector<unsigned char> slot_sym, row_sym;   // symmetry given with a keyed load, by slot / by row of the last run

  // on-device NN cache (p3hip_cache_enable): the table, the per-slot keys as loaded, and the per-run lists
  struct DeviceCache {
    bool on = false;
    unsigned mask = 0, run = 0;
    unsigned long long* d_tkeys = nullptr;
    unsigned* d_tmeta = nullptr;
    float* d_tvals = nullptr;
    p3::CacheKey* h_slot_keys = nullptr;   // [batch] by slot (plain memory, written by load_slot_keyed)
    p3::CacheKey *h_keys = nullptr, *d_keys = nullptr;   // [batch] by row of the run (pinned / device)
    int *h_hit = nullptr, *d_hit = nullptr, *h_victim = nullptr, *d_victim = nullptr;
    int *h_lists = nullptr, *d_lists = nullptr;          // [5][batch]: miss rows, hit entries, insert rows, insert src, insert entries
    unsigned *h_sym = nullptr, *d_sym = nullptr;         // [batch] symmetry of the result in out row r
    unsigned char* d_feats2 = nullptr;                   // features of the misses, dense
    std::vector<int> out_row;                            // row of the run -> row of d_out / h_out
    std::vector<unsigned char> was_hit;                  // by row of the run
    unsigned long long lookups = 0, hits = 0, inserts = 0;
  } cache;
  // row of d_out / h_out that holds `slot`'s result (-1: not evaluated by the last run)
  int out_row_of(int slot) const {
    const int row = slots.row(slot);
    return (row >= 0 && cache.on) ? cache.out_row[row] : row;
  }

  bool check(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
  }
  template <class T>
  const T* dev(size_t off) const { return reinterpret_cast<const T*>(d_arena + off); }
  // HIP's current device is per host thread, and the ABI is called from whatever thread the
  // host likes (the infer thread, one GPU thread per game group, a rank's main thread): every
  // entry point that touches HIP binds the engine's device first.
  bool bind() { return check(hipSetDevice(device), "hipSetDevice"); }
};

namespace {

FoldedBN fold_bn(Arena& ar, const WeightFile& wf, const std::string& prefix, size_t n) {
  const Tensor& g = wf.get(prefix + ".gamma", n);
  const Tensor& b = wf.get(prefix + ".beta", n);
  const Tensor& m = wf.get(prefix + ".mean", n);
  const Tensor& v = wf.get(prefix + ".var", n);
  std::vector<float> sc(n), sh(n);
  for (size_t i = 0; i < n; ++i) {
    sc[i] = g.data[i] / std::sqrt(v.data[i] + kBnEps);
    sh[i] = b.data[i] - m.data[i] * sc[i];
  }
  FoldedBN f;
  f.scale_off = ar.add(sc.data(), n * 4);
  f.shift_off = ar.add(sh.data(), n * 4);
  return f;
}

// A stream is a whole number of macro-steps of 4 k16 blocks = cout_pass * 128 bytes
// (conv_core.h ring_slot_bytes).
// A stream that is not a whole number of macro-steps is a packing bug: it is reported through
// Arena::bad_stream and fails p3hip_create (never abort() inside the library).
size_t add_stream(Arena& ar, const std::vector<_Float16>& s, int& nms, int cout_pass) {
  const size_t ms = (size_t)cout_pass * 128;
  if (s.size() * 2 % ms != 0) ar.bad_stream = true;
  nms = (int)(s.size() * 2 / ms);
  return ar.add(s.data(), s.size() * 2);
}

bool build_plan(p3hip_engine* e, Arena& ar) {
  const WeightFile& wf = e->wf;
  const int C = wf.C, Cb = wf.Cb;
  const bool fused = (C == 256 && Cb == 128) || (C == 128 && Cb == 64);
  const bool classic = wf.btype == 2 && C == 192 && wf.inner == 2;   // b15c192_classic
  const bool bottleneck_ok = wf.btype == 1 || (wf.btype == 0 && wf.inner >= 1 && wf.inner <= 3);
  const bool layerwise = (C == 384 && Cb == 192 && bottleneck_ok) || classic;
  const bool v_ok = wf.V == 32 || wf.V == 48 || wf.V == 64 || wf.V == 80;
  if (!((fused && bottleneck_ok) || layerwise) || wf.H != 32 || !v_ok) {
    e->err = "unsupported architecture for the HIP engine (need (C, Cb) in {(128,64), (256,128), (384,192)} with "
             "btl (1-3 inner layers) or nbt blocks, or C=192 classic blocks of two convs; H=32, V in {32,48,64,80})";
    return false;
  }
  // slice width of the per-position kernels that stage C channels (k_conv1x1 family)
  const int CB = classic ? 64 : (layerwise ? 128 : Cb);
  const int CPI = classic ? 64 : 128;   // output pass width of the init conv
  // init conv
  {
    std::vector<_Float16> s;
    const Tensor& w = wf.get("init_conv.w", (size_t)25 * 15 * C);  // [5][5][15][C]
    for (int cp = 0; cp < C / CPI; ++cp)
      pack_segment(s, w.data, 25, 28, 15, C, 0, 16, cp * CPI, CPI);
    e->init_stream_off = add_stream(ar, s, e->init_nms, CPI);
    e->game_w_off = ar.add(wf.get("init_game.w", (size_t)8 * C).data, 8 * C * 4);
    e->game_b_off = ar.add(wf.get("init_game.b", (size_t)C).data, C * 4);
  }
  // k_blockw serves C = 256 / C_b = 128 btl trunks; the broadcast blocks then run as their own launches
  {
    static const bool want_blockw = getenv("P3HIP_BLOCKW") != nullptr && atoi(getenv("P3HIP_BLOCKW")) != 0;
    e->blockw = want_blockw && C == 256 && Cb == 128 && wf.btype == 0 && wf.inner >= 1 && wf.inner <= 3;
    if (e->blockw) e->bcast_fuse = false;
  }
  bool have_xa = false;   // layer-wise path: u holds mish(bn0(x)) of the next block
  // The weight streams of consecutive fused blocks go into the arena back to back, after the
  // run's other tensors: one k_block launch walks the streams of all its blocks as ONE circular
  // stream (position-major order, kernels.hip), so a run must be contiguous.
  // With the broadcast 1x1 convs fused in, a run's stream is [conv_last of the broadcast block before
  // it] [its blocks] [conv_first of the broadcast block after it].
  std::vector<std::pair<size_t, std::vector<_Float16>>> run_streams;   // (block index, stream)
  std::vector<_Float16> head_stream, tail_stream;   // of the run being collected
  int head_of = -1, tail_of = -1;
  // The runs' streams are laid out after the loop, all runs back to back ([head r][blocks r][tail r][head r + 1] ...):
  // with everything of the broadcast blocks between them fused, one k_block launch walks them all (joined_launch).
  std::vector<std::function<void()>> layout;
  auto flush_run = [&]() {
    if (!run_streams.empty()) {
      auto rs_all = std::make_shared<std::vector<std::pair<size_t, std::vector<_Float16>>>>(std::move(run_streams));
      auto hs = std::make_shared<std::vector<_Float16>>(std::move(head_stream));
      auto ts = std::make_shared<std::vector<_Float16>>(std::move(tail_stream));
      const int h_of = head_of, t_of = tail_of;
      layout.push_back([&ar, e, rs_all, hs, ts, h_of, t_of, Cb]() {
        int nms_unused = 0;
        if (h_of >= 0) {
          add_stream(ar, *hs, nms_unused, Cb);
          BlockPlan& b = e->blocks[rs_all->front().first];
          b.head_of = h_of;
          b.head_bytes = hs->size() * 2;
          e->blocks[h_of].last_fused = true;
        }
        for (auto& rs : *rs_all) {
          BlockPlan& b = e->blocks[rs.first];
          b.stream_off = add_stream(ar, rs.second, b.nms, Cb);
          b.stream_bytes = rs.second.size() * 2;
        }
        if (t_of >= 0) {
          add_stream(ar, *ts, nms_unused, Cb);
          BlockPlan& b = e->blocks[rs_all->back().first];
          b.tail_of = t_of;
          b.tail_bytes = ts->size() * 2;
        }
      });
    }
    run_streams.clear();
    head_stream.clear();
    tail_stream.clear();
    head_of = tail_of = -1;
  };
  for (int i = 0; i < wf.nblocks; ++i) {
    BlockPlan bp;
    if (layerwise) flush_run();
    const std::string p = "blocks." + std::to_string(i);
    // conv j of this block, checked against the [k][k][cin][cout] size the packer will read
    auto W = [&](int j, int kw, int cin, int cout) {
      return wf.get(p + ".conv" + std::to_string(j) + ".w", (size_t)kw * kw * cin * cout).data;
    };
    if (wf.is_broadcast(i)) {
      bp.kind = 3;
      have_xa = false;
      bp.bn[0] = fold_bn(ar, wf, p + ".bn0", C);
      bp.bn[1] = fold_bn(ar, wf, p + ".bn1", C);
      const int CPb = (CB == 128) ? 128 : 64;
      std::vector<_Float16> s0, s1, s2;
      for (int cp = 0; cp < C / CPb; ++cp)
        for (int ip = 0; ip < C / CB; ++ip) {
          pack_segment(s0, W(0, 1, C, C), 1, 1, C, C, ip * CB, CB, cp * CPb, CPb);
          pack_segment(s2, W(1, 1, C, C), 1, 1, C, C, ip * CB, CB, cp * CPb, CPb);
        }
      const Tensor& dw = wf.get(p + ".dense.w", (size_t)kNLoc * kNLoc);  // [361 i][361 j]
      for (int jp = 0; jp < 3; ++jp)
        for (int q = 0; q < 24; ++q)
          for (int h = 0; h < 2; ++h)
            for (int jj = 0; jj < 128; ++jj)
              for (int el = 0; el < 8; ++el) {
                int ii = q * 16 + h * 8 + el, j = jp * 128 + jj;
                float v = (ii < kNLoc && j < kNLoc) ? dw.data[(size_t)ii * kNLoc + j] : 0.0f;
                s1.push_back((_Float16)v);
              }
      // The run before this block takes conv_first as its tail, the run after it conv_last as its
      // head (the stand-alone streams below serve P3HIP_NO_BFUSE and broadcast blocks without a
      // fused neighbour).
      const bool fused_neighbours = fused && e->bcast_fuse;
      // fused copies: output pass 1 takes its K slices in the order (1, 0) — slice 1 is the one
      // still in the act buffer when pass 0 ends (kernels.hip k_block, BC form)
      std::vector<_Float16> f0, f2;
      if (fused_neighbours)
        for (int cp = 0; cp < 2; ++cp)
          for (int k = 0; k < 2; ++k) {
            const int ip = cp == 0 ? k : 1 - k;
            pack_segment(f0, W(0, 1, C, C), 1, 1, C, C, ip * Cb, Cb, cp * Cb, Cb);
            pack_segment(f2, W(1, 1, C, C), 1, 1, C, C, ip * Cb, Cb, cp * Cb, Cb);
          }
      if (fused_neighbours && !run_streams.empty()) {
        tail_stream = f0;
        tail_of = i;
        bp.first_fused = true;
        // C = 256: the dense rides in that tail too (k_block, tail_dense).  Its stream there:
        // [conv_first pass 0 (K slices 0, 1)] [dense] [conv_first pass 1 (K slices 0, 1)] [dense], the dense matrix
        // with its K index = the act buffer's padded board row r = 20 y + x (zero rows for x = 19 and r >= 379)
        static const bool no_dfuse = getenv("P3HIP_NO_DFUSE") != nullptr;
        if (C == 256 && Cb == 128 && wf.btype == 0 && !no_dfuse) {   // btl blocks only (k_block's tail_dense)
          bp.dense_fused = true;
          std::vector<_Float16> dpad;
          for (int jp = 0; jp < 3; ++jp)
            for (int q = 0; q < 24; ++q)
              for (int h = 0; h < 2; ++h)
                for (int jj = 0; jj < 128; ++jj)
                  for (int el = 0; el < 8; ++el) {
                    const int r = q * 16 + h * 8 + el, y = r / 20, x = r % 20, j = jp * 128 + jj;
                    const bool on_board = x < 19 && y < 19;
                    float v = (on_board && j < kNLoc) ? dw.data[(size_t)(y * 19 + x) * kNLoc + j] : 0.0f;
                    dpad.push_back((_Float16)v);
                  }
          tail_stream.clear();
          for (int cp = 0; cp < 2; ++cp) {
            for (int ip = 0; ip < 2; ++ip) pack_segment(tail_stream, W(0, 1, C, C), 1, 1, C, C, ip * Cb, Cb, cp * Cb, Cb);
            tail_stream.insert(tail_stream.end(), dpad.begin(), dpad.end());
          }
        }
      }
      flush_run();
      if (fused_neighbours && i + 1 < wf.nblocks && !wf.is_broadcast(i + 1)) {
        head_stream = f2;
        head_of = i;   // last_fused is set when the run is laid out
      }
      bp.stream_off = add_stream(ar, s0, bp.nms, CPb);
      bp.stream2_off = add_stream(ar, s1, bp.nms2, 128);
      bp.stream3_off = add_stream(ar, s2, bp.nms3, CPb);
      bp.dense_bias_off = ar.add(wf.get(p + ".dense.b", (size_t)kNLoc).data, kNLoc * 4);
    } else if (layerwise) {
      bp.kind = 4;
      const int nconv = classic ? 2 : ((wf.btype == 0) ? wf.inner + 2 : 6);
      for (int j = 0; j < nconv; ++j) bp.bn[j] = fold_bn(ar, wf, p + ".bn" + std::to_string(j), (classic || j == 0) ? C : Cb);
      // The consumer's prologue (BN + mish of its input) is applied ONCE by the producer: a
      // layer either stores its output already activated for the next conv (act), or stores it
      // raw and a second, activated copy (dual).  Only the first layer after the init conv or
      // a broadcast block still activates its input while staging (pre) — every workgroup of
      // an output pass would otherwise redo that VALU work.  `xa` = activated copy of x, in u.
      FoldedBN next_bn0{};
      const bool next_layerwise = i + 1 < wf.nblocks && !wf.is_broadcast(i + 1);
      if (next_layerwise) next_bn0 = fold_bn(ar, wf, "blocks." + std::to_string(i + 1) + ".bn0", C);
      const FoldedBN none{};
      auto add_layer = [&](int j, int kw, int cin, int cout, bool pre, const FoldedBN& pre_bn, bool act, bool res,
                           bool dual, const FoldedBN& out_bn, int in_buf, int out_buf, int out2_buf) {
        LayerPlan lp{kw, cin, cout, pre, act, res, dual, pre_bn, out_bn, in_buf, out_buf, out2_buf};
        std::vector<_Float16> s;
        for (int cp = 0; cp < cout / 64; ++cp)
          for (int ip = 0; ip < cin / 64; ++ip) pack_segment(s, W(j, kw, cin, cout), kw * kw, kw * kw, cin, cout, ip * 64, 64, cp * 64, 64);
        lp.stream_off = add_stream(ar, s, lp.nms, 64);
        bp.layers.push_back(lp);
      };
      const bool from_xa = have_xa;          // first layer input: activated copy in u, or raw x with pre
      const int in0 = from_xa ? 3 : 0;
      if (classic) {         // x + conv3(act1(conv3(act0(x)))), model.py:330-368
        add_layer(0, 3, C, C, !from_xa, bp.bn[0], true, false, false, bp.bn[1], in0, 1, -1);
        add_layer(1, 3, C, C, false, none, false, true, next_layerwise, next_bn0, 1, 0, 3);
      } else if (wf.btype == 0) {   // btl
        add_layer(0, 1, C, Cb, !from_xa, bp.bn[0], true, false, false, bp.bn[1], in0, 1, -1);
        int cur = 1;
        for (int j = 1; j <= wf.inner; ++j) {
          add_layer(j, 3, Cb, Cb, false, none, true, false, false, bp.bn[j + 1], cur, 3 - cur, -1);
          cur = 3 - cur;
        }
        add_layer(wf.inner + 1, 1, Cb, C, false, none, false, true, next_layerwise, next_bn0, cur, 0, 3);
      } else {               // nbt: the inner residual stream t stays raw in region 1
        add_layer(0, 1, C, Cb, !from_xa, bp.bn[0], false, false, true, bp.bn[1], in0, 1, 2);   // t, act1(t)
        add_layer(1, 3, Cb, Cb, false, none, true, false, false, bp.bn[2], 2, 3, -1);
        add_layer(2, 3, Cb, Cb, false, none, false, true, true, bp.bn[3], 3, 1, 2);             // t' = t + ., act3(t')
        add_layer(3, 3, Cb, Cb, false, none, true, false, false, bp.bn[4], 2, 3, -1);
        add_layer(4, 3, Cb, Cb, false, none, false, true, true, bp.bn[5], 3, 1, 2);             // t'', act5(t'')
        add_layer(5, 1, Cb, C, false, none, false, true, next_layerwise, next_bn0, 2, 0, 3);
      }
      have_xa = next_layerwise;
    } else {
      bp.kind = wf.btype;
      const int nconv = (wf.btype == 0) ? wf.inner + 2 : 6;
      for (int j = 0; j < nconv; ++j) bp.bn[j] = fold_bn(ar, wf, p + ".bn" + std::to_string(j), j == 0 ? C : Cb);
      std::vector<_Float16> s;
      for (int ip = 0; ip < C / CB; ++ip) pack_segment(s, W(0, 1, C, Cb), 1, 1, C, Cb, ip * CB, CB, 0, CB);
      for (int j = 1; j < nconv - 1; ++j) pack_segment_3x3(s, W(j, 3, Cb, Cb), Cb, Cb, CB, CB);
      for (int cp = 0; cp < C / CB; ++cp) pack_segment(s, W(nconv - 1, 1, Cb, C), 1, 1, Cb, C, 0, CB, cp * CB, CB);
      run_streams.emplace_back(e->blocks.size(), std::move(s));
    }
    e->blocks.push_back(bp);
  }
  flush_run();
  for (auto& f : layout) f();
  e->runs_contiguous = true;
  if (e->blockw) {
    // Per run of consecutive btl blocks: the weight stream in the order k_blockw consumes it and the parameter table.
    // The BN in front of a conv's consumer rides in the conv: its folded scale times log2(e) is multiplied into the fp16
    // weights, its shift times log2(e) is the accumulators' initial value (the kernel's exp2-based mish takes
    // log2(e) * y, as bn_mish8_l2 does).
    //   block stream: reduce: x halves (quarters 0, 1 / 2, 3); in a half set A's four k32 granules, then set B's
    //                 layer j: phases (A, lo) (B, lo) (A, hi) (B, hi), each (ky, q32, kx) over its 64 input channels
    //                 expand: output quarters 0..3, k32 steps 0..3 (unscaled: the residual add follows)
    //   block table:  bn0 scale[256] shift[256] (times log2 e) | conv j = 0 .. L: shift[128] of bn j + 1 (times log2 e)
    const int L = wf.inner;
    const float kLog2e = 1.4426950408889634f;
    for (size_t bi = 0; bi < e->blocks.size();) {
      if (e->blocks[bi].kind != 0) { ++bi; continue; }
      size_t n = 1;
      while (bi + n < e->blocks.size() && e->blocks[bi + n].kind == 0) ++n;
      std::vector<_Float16> ws;
      std::vector<float> prm;
      for (size_t b = bi; b < bi + n; ++b) {
        // block index in the weight file = plan index (one BlockPlan per trunk block)
        const std::string p = "blocks." + std::to_string(b);
        auto W = [&](int j, int kw, int cin, int cout) {
          return wf.get(p + ".conv" + std::to_string(j) + ".w", (size_t)kw * kw * cin * cout).data;
        };
        const BlockPlan& bp = e->blocks[b];
        auto bn_row = [&](int j, bool shift) {
          const float* v = reinterpret_cast<const float*>(ar.host.data() + (shift ? bp.bn[j].shift_off : bp.bn[j].scale_off));
          std::vector<float> r((size_t)(j == 0 ? C : Cb));
          for (size_t c = 0; c < r.size(); ++c) r[c] = v[c] * kLog2e;
          return r;
        };
        const float* w0 = W(0, 1, C, Cb);
        const std::vector<float> s1 = bn_row(1, false);
        for (int half = 0; half < 2; ++half)
          for (int s0 = 0; s0 < 128; s0 += 64)
            for (int st = 0; st < 4; ++st) pack_granule(ws, w0, C, Cb, 0, 128 * half + 32 * st, s0, s1.data());
        for (int j = 1; j <= L; ++j) {
          const float* wj = W(j, 3, Cb, Cb);
          const std::vector<float> sj = bn_row(j + 1, false);
          for (int ph = 0; ph < 4; ++ph) {
            const int s0 = (ph & 1) * 64, half = ph >> 1;
            for (int ky = 0; ky < 3; ++ky)
              for (int q = 0; q < 2; ++q)
                for (int kx = 0; kx < 3; ++kx) pack_granule(ws, wj, Cb, Cb, ky * 3 + kx, 64 * half + 32 * q, s0, sj.data());
          }
        }
        const float* we = W(L + 1, 1, Cb, C);
        for (int qo = 0; qo < 4; ++qo)
          for (int c = 0; c < 4; ++c) pack_granule(ws, we, Cb, C, 0, 32 * c, 64 * qo, nullptr);
        for (int sh = 0; sh < 2; ++sh) {
          const std::vector<float> r = bn_row(0, sh == 1);
          prm.insert(prm.end(), r.begin(), r.end());
        }
        for (int j = 1; j <= L + 1; ++j) {
          const std::vector<float> r = bn_row(j, true);
          prm.insert(prm.end(), r.begin(), r.end());
        }
      }
      p3hip_engine::BlockwRun run;
      run.first = bi;
      run.nblk = (int)n;
      run.stream_off = ar.add(ws.data(), ws.size() * 2);
      run.prm_off = ar.add(prm.data(), prm.size() * 4);
      if (ws.size() * 2 != n * (size_t)(32 + 72 * L) * 4096 || prm.size() != n * (size_t)(512 + 128 * (L + 1))) ar.bad_stream = true;
      e->bw_runs.push_back(run);
      bi += n;
    }
  }
  // heads: conv_p | conv_g | value.conv  -> [C][96]
  {
    std::vector<float> w((size_t)C * 96);
    const float* wp = wf.get("policy.conv_p.w", (size_t)C * 32).data;
    const float* wg = wf.get("policy.conv_g.w", (size_t)C * 32).data;
    const float* wv = wf.get("value.conv.w", (size_t)C * 32).data;
    for (int c = 0; c < C; ++c)
      for (int o = 0; o < 32; ++o) {
        w[(size_t)c * 96 + o] = wp[c * 32 + o];
        w[(size_t)c * 96 + 32 + o] = wg[c * 32 + o];
        w[(size_t)c * 96 + 64 + o] = wv[c * 32 + o];
      }
    std::vector<_Float16> s;
    for (int cp = 0; cp < 2; ++cp)
      for (int ip = 0; ip < C / CB; ++ip) pack_segment(s, w.data(), 1, 1, C, 96, ip * CB, CB, cp * 64, 64);
    e->heads_stream_off = add_stream(ar, s, e->heads_nms, 64);
    // the same weights as MFMA 16x16x32 A fragments for k_headsx (the convs inside the heads kernel):
    // [cout tile ct][k32 step][lane (n = lane & 15, q = lane >> 4)][8] = w[cin = 32 step + 8 q + e][cout = 16 ct + n]
    if (p3::heads_fusable(C, wf.V)) {
      std::vector<_Float16> af;
      for (int ct = 0; ct < 6; ++ct)
        for (int st = 0; st < C / 32; ++st)
          for (int lane = 0; lane < 64; ++lane)
            for (int el = 0; el < 8; ++el)
              af.push_back((_Float16)w[(size_t)(st * 32 + 8 * (lane >> 4) + el) * 96 + ct * 16 + (lane & 15)]);
      e->heads_conv_a_off = ar.add(af.data(), af.size() * 2);
      e->heads_fused = getenv("P3HIP_NO_HFUSE") == nullptr;
    }
    FoldedBN g = fold_bn(ar, wf, "policy.gpool_bn", 32);
    e->head_off["gbn_scale"] = g.scale_off;
    e->head_off["gbn_shift"] = g.shift_off;
    const size_t H = 32, V = (size_t)wf.V;
    const struct { const char* name; size_t n; } heads[] = {   // sizes k_heads reads (kernels.h HeadsArgs)
        {"policy.gpool_dense.w", 2 * H * H}, {"policy.gpool_dense.b", H}, {"policy.out_moves.w", 2 * H},
        {"policy.out_pass.w", 4 * H}, {"policy.out_pass.b", 2}, {"policy.opt_moves.w", H},
        {"policy.opt_pass.w", 2 * H}, {"policy.opt_pass.b", 1}, {"value.oq_embed.w", 2 * H * V},
        {"value.oq_embed.b", V}, {"value.oq_out.w", V * 14}, {"value.oq_out.b", 14}, {"value.own.w", H},
        {"value.gamma_pre.w", 2 * H * V}, {"value.gamma_pre.b", V}, {"value.gamma_out.w", V},
        {"value.gamma_out.b", 1}, {"value.score_pre.w", (2 * H + 1) * V}, {"value.score_pre.b", V},
        {"value.score_out.w", V}, {"value.score_out.b", 1}};
    for (const auto& h : heads) {
      const Tensor& t = wf.get(h.name, h.n);
      e->head_off[h.name] = ar.add(t.data, t.size() * 4);
    }
    // k_headsx takes the same tensors as one image in its LDS order (kernels.h heads_image_floats)
    if (p3::heads_fusable(C, wf.V) && wf.missing.empty()) {
      std::vector<float> img;
      auto put = [&](const char* name, size_t n, size_t pad = 0) {
        const Tensor& t = wf.get(name, n);
        img.insert(img.end(), t.data, t.data + n);
        img.insert(img.end(), pad, 0.0f);
      };
      put("policy.gpool_dense.w", 2 * H * H); put("value.oq_embed.w", 2 * H * V); put("value.gamma_pre.w", 2 * H * V);
      put("value.score_pre.w", (2 * H + 1) * V); put("value.oq_out.w", V * 14); put("value.gamma_out.w", V);
      put("value.score_out.w", V); put("policy.out_pass.w", 4 * H); put("policy.opt_pass.w", 2 * H);
      put("policy.out_moves.w", 2 * H); put("policy.opt_moves.w", H); put("value.own.w", H);
      {
        const float* gs = reinterpret_cast<const float*>(ar.host.data() + g.scale_off);
        const float* gh = reinterpret_cast<const float*>(ar.host.data() + g.shift_off);
        std::vector<float> tmp(gs, gs + H);
        img.insert(img.end(), tmp.begin(), tmp.end());
        tmp.assign(gh, gh + H);
        img.insert(img.end(), tmp.begin(), tmp.end());
      }
      put("policy.gpool_dense.b", H); put("value.oq_embed.b", V); put("value.gamma_pre.b", V); put("value.score_pre.b", V);
      put("value.oq_out.b", 14, 2);
      if ((int)img.size() != p3::heads_image_floats(32, wf.V)) {
        e->err = "internal error: heads image size";
        return false;
      }
      e->heads_image_off = ar.add(img.data(), img.size() * 4);
    }
  }
  if (!wf.missing.empty()) {
    e->err = "weight file lacks tensors of the architecture its header names: " + wf.missing;
    return false;
  }
  if (ar.bad_stream) {
    e->err = "internal error: a packed weight stream is not a whole number of ring macro-steps";
    return false;
  }
  return true;
}

int grid_for(const p3hip_engine* e, int npos, int npos_per_wg) {
  int wgs = (npos + npos_per_wg - 1) / npos_per_wg;
  return wgs < e->n_cu ? wgs : e->n_cu;
}

// Arguments of one k_block launch over the consecutive fused blocks [first, first + count).
p3::BlockArgs block_args(p3hip_engine* e, size_t first, int count, int npos) {
  p3::BlockArgs a{};
  a.x = e->d_x;
  a.t = e->d_s ? e->d_s : e->d_t;
  a.npos = npos;
  a.nblk = count;
  a.nruns = 1;
  // the streams of consecutive fused blocks lie back to back in the arena (build_plan), the fused
  // broadcast convs' right before the run's first and right after its last block
  const BlockPlan& fb = e->blocks[first];
  const BlockPlan& lb = e->blocks[first + count - 1];
  const size_t ms_bytes = (size_t)p3::block_macro_step_bytes(e->wf.C, e->c128_wg8);
  a.wstream = e->d_arena + fb.stream_off;
  if (fb.head_of >= 0) {
    a.head = 1;
    a.zin = a.zin2 = e->d_u;
    a.wstream = e->d_arena + fb.stream_off - fb.head_bytes;
    a.nms_total += (int)(fb.head_bytes / ms_bytes);
  }
  if (lb.tail_of >= 0) {
    const BlockPlan& bb = e->blocks[lb.tail_of];
    a.tail = 1;
    a.tout = e->d_t;
    a.tail_scale[0] = e->dev<float>(bb.bn[0].scale_off);
    a.tail_shift[0] = e->dev<float>(bb.bn[0].shift_off);
    if (bb.dense_fused) {
      a.tail_dense = 1;
      a.uout = a.uout2 = e->d_u;
      a.dense_bias[0] = e->dev<float>(bb.dense_bias_off);
      a.dense_scale[0] = e->dev<float>(bb.bn[1].scale_off);
      a.dense_shift[0] = e->dev<float>(bb.bn[1].shift_off);
    }
    a.nms_total += (int)(lb.tail_bytes / ms_bytes);
  }
  {
    // Workgroups run their positions in lockstep, so the HBM-bound phases of a launch (the fused
    // broadcast convs above all) hit the memory system from every CU at once.  A start-up stagger of
    // 10,000 cycles per step (seven steps across the CU slots of an XCD) measured -2 % on the forward
    // pass at four positions per workgroup (gpurun_out/stagger_ab.log); it costs its own length once
    // per launch, so short launches go without.  P3HIP_STAGGER overrides (0 = off).
    static const int stagger_env = getenv("P3HIP_STAGGER") ? atoi(getenv("P3HIP_STAGGER")) : -1;
    const bool long_launch = npos >= 3 * e->n_cu * (e->wf.C == 256 || e->c128_wg8 ? 1 : 2);
    // (engines sharing the GPU with others: the spread costs its own length and another stream's kernels fill a
    // launch's tail anyway — 0.3-0.6 % of the self-play rate, gpurun_out/stagger_selfplay.log)
    const bool shared = (e->flags & P3HIP_FLAG_SHARED_DEVICE) != 0;
    a.stagger = stagger_env >= 0 ? stagger_env : ((a.head || a.tail) && long_launch && !shared ? 10000 : 0);
    // two 4-wave workgroups per CU and at least two positions each: they take turns at the higher wave
    // priority (kernels.h; b12c128btl3 forward -3.5 %, b8c128nbt -2.5 % at 1024 positions, nothing at 512 and
    // below; profiles/r02_c128_pair_turns.txt).  P3HIP_NO_PAIR_TURNS=1 leaves the priorities alone.
    static const bool no_pair_turns = getenv("P3HIP_NO_PAIR_TURNS") != nullptr;
    a.pair_turns = !no_pair_turns && e->wf.C == 128 && !e->c128_wg8 && npos >= 4 * e->n_cu;
  }
  for (int b = 0; b < count; ++b) {
    const BlockPlan& bp = e->blocks[first + b];
    a.nms_total += (int)(bp.stream_bytes / ms_bytes);
    for (int j = 0; j < p3::kMaxBlockLayers; ++j) {
      a.blk[b].scale[j] = e->dev<float>(bp.bn[j].scale_off);
      a.blk[b].shift[j] = e->dev<float>(bp.bn[j].shift_off);
    }
  }
  return a;
}

// Joined launch (C = 256 btl): the runs of fused blocks from `first` on, with the broadcast blocks between them inside
// ONE k_block launch — possible when every such broadcast block has both its 1x1 convs AND its dense fused into the
// neighbouring runs (then their streams lie back to back in the arena, build_plan).  Returns the number of plan
// blocks covered (0: not joinable) and fills `a`.
int joined_launch(p3hip_engine* e, size_t first, int npos, p3::BlockArgs* out) {
  static const bool no_join = getenv("P3HIP_NO_JOIN") != nullptr || getenv("P3HIP_NO_FUSE") != nullptr;
  if (no_join || !e->runs_contiguous) return 0;
  std::vector<std::pair<size_t, int>> runs;   // (first block, count)
  size_t bi = first;
  int nb = 0;
  while (bi < e->blocks.size() && (int)runs.size() < p3::kMaxRuns) {
    const int kind = e->blocks[bi].kind;
    if (kind != 0) break;
    int n = 1;
    while (bi + n < e->blocks.size() && e->blocks[bi + n].kind == kind) ++n;
    if (n > p3::kMaxFuse || nb + n > p3::kMaxLaunchBlocks) break;
    runs.emplace_back(bi, n);
    nb += n;
    bi += n;
    // a broadcast block with everything fused, followed by another run?
    if (bi + 1 < e->blocks.size() && e->blocks[bi].kind == 3 && e->blocks[bi].first_fused && e->blocks[bi].last_fused &&
        e->blocks[bi].dense_fused && e->blocks[bi + 1].kind == 0) ++bi;
    else break;
  }
  if (runs.size() < 2) return 0;
  // the loop may have stepped over a broadcast block without taking the run behind it (kMaxRuns, block limit)
  const size_t last_run_end = runs.back().first + runs.back().second;
  p3::BlockArgs a = block_args(e, runs[0].first, runs[0].second, npos);   // head of run 0 (if any), stagger, stream start
  a.nruns = (int)runs.size();
  a.nms_total = 0;
  a.tail = 0;
  a.tail_dense = 1;
  // u alternates between two buffers inside the launch (t's buffer is free: the dense is fused) when the launch
  // neither starts with a head fed from outside nor ends with a tail read from outside; otherwise one buffer
  const bool closed = !a.head && !(e->blocks[runs.back().first + runs.back().second - 1].tail_of >= 0);
  _Float16* other = closed ? e->d_t : e->d_u;
  a.zin2 = e->d_u;   // head of an even run: what the odd run before it wrote
  a.uout = other;    // tail of an even run
  a.zin = other;     // head of an odd run
  a.uout2 = e->d_u;  // tail of an odd run
  const size_t ms_bytes = (size_t)p3::block_macro_step_bytes(e->wf.C, e->c128_wg8);
  int k = 0;
  for (size_t r = 0; r < runs.size(); ++r) {
    a.run_nblk[r] = runs[r].second;
    const BlockPlan& fb = e->blocks[runs[r].first];
    const BlockPlan& lb = e->blocks[runs[r].first + runs[r].second - 1];
    if (fb.head_of >= 0) a.nms_total += (int)(fb.head_bytes / ms_bytes);
    for (int b = 0; b < runs[r].second; ++b, ++k) {
      const BlockPlan& bp = e->blocks[runs[r].first + b];
      a.nms_total += (int)(bp.stream_bytes / ms_bytes);
      for (int j = 0; j < p3::kMaxBlockLayers; ++j) {
        a.blk[k].scale[j] = e->dev<float>(bp.bn[j].scale_off);
        a.blk[k].shift[j] = e->dev<float>(bp.bn[j].shift_off);
      }
    }
    if (r + 1 < runs.size() || lb.tail_of >= 0) {
      const BlockPlan& bb = e->blocks[lb.tail_of];
      a.nms_total += (int)(lb.tail_bytes / ms_bytes);
      a.tail_scale[r] = e->dev<float>(bb.bn[0].scale_off);
      a.tail_shift[r] = e->dev<float>(bb.bn[0].shift_off);
      a.dense_bias[r] = e->dev<float>(bb.dense_bias_off);
      a.dense_scale[r] = e->dev<float>(bb.bn[1].scale_off);
      a.dense_shift[r] = e->dev<float>(bb.bn[1].shift_off);
      if (r + 1 == runs.size()) a.tail = 1;   // the last run has a tail too (a broadcast block follows it)
    }
  }
  // every run's streams must lie back to back: [head r][blocks r][tail r][head r+1] ...
  for (size_t r = 0; r + 1 < runs.size(); ++r) {
    const BlockPlan& lb = e->blocks[runs[r].first + runs[r].second - 1];
    const BlockPlan& nf = e->blocks[runs[r + 1].first];
    if (lb.stream_off + lb.stream_bytes + lb.tail_bytes + nf.head_bytes != nf.stream_off) return 0;
  }
  *out = a;
  return (int)(last_run_end - first);
}

// Number of blocks the launch starting at block `first` covers: consecutive blocks of the fused
// kernel's kind, at most kMaxFuse (1 when P3HIP_NO_FUSE is set: one launch per block).
int fused_run(const p3hip_engine* e, size_t first) {
  const int kind = e->blocks[first].kind;
  if (kind != 0 && kind != 1) return 0;
  static const bool no_fuse = getenv("P3HIP_NO_FUSE") != nullptr;
  int n = 1;
  while (!no_fuse && n < p3::kMaxFuse && first + n < e->blocks.size() && e->blocks[first + n].kind == kind) ++n;
  return n;
}

// k_blockw: the code object assembled from csrc/asm/blockw_gen.py's output rides in the library as a blob
// (blockw_blob.S); one module per engine (modules are per device).  Kernel arguments: csrc/asm/blockw_gen.py.
extern "C" const unsigned char p3_blockw_hsaco[];
extern "C" const unsigned char p3_blockw_hsaco_end[];
struct BlockwArgs {
  const void* x; const void* ws; const void* prm;
  int npos, nblk, nwg, pad;
  void* stamps;
  unsigned long long pad2[2];
};
static_assert(sizeof(BlockwArgs) == 64, "kernarg layout of k_blockw");

bool load_blockw(p3hip_engine* e) {
  if (e->bw_fn) return true;
  static const bool diag = getenv("P3HIP_BLOCKW_DIAG") != nullptr;
  if (!e->check(hipModuleLoadData(&e->bw_mod, p3_blockw_hsaco), "hipModuleLoadData k_blockw")) return false;
  const std::string name = "k_blockw_L" + std::to_string(e->wf.inner) + (diag ? "_diag" : "");
  if (!e->check(hipModuleGetFunction(&e->bw_fn, e->bw_mod, name.c_str()), "hipModuleGetFunction k_blockw")) return false;
  if (diag) {
    constexpr size_t bytes = 8 * 16 * 4 * 24 * 8;   // [workgroup 0..7][block][wave][stamp]
    if (!e->check(hipMalloc((void**)&e->d_bw_stamps, bytes), "hipMalloc stamps") ||
        !e->check(hipMemsetAsync(e->d_bw_stamps, 0, bytes, e->stream), "hipMemset stamps")) return false;
  }
  return true;
}

bool launch_blockw(p3hip_engine* e, const p3hip_engine::BlockwRun& run, int npos) {
  if (!load_blockw(e)) return false;
  BlockwArgs a{};
  a.x = e->d_x;
  a.ws = e->d_arena + run.stream_off;
  a.prm = e->d_arena + run.prm_off;
  a.npos = npos;
  a.nblk = run.nblk;
  a.nwg = npos < e->n_cu ? npos : e->n_cu;
  a.stamps = e->d_bw_stamps;
  size_t size = sizeof a;
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return e->check(hipModuleLaunchKernel(e->bw_fn, a.nwg, 1, 1, 256, 1, 1, 0, e->stream, nullptr, cfg), "launch k_blockw");
}

// Enqueues the whole forward pass for `npos` dense positions already in d_feats.
bool enqueue_forward(p3hip_engine* e, int npos) {
  const WeightFile& wf = e->wf;
  const int C = wf.C;
  const int npw = (C >= 256) ? 1 : 2;   // positions per workgroup of the k_conv1x1 family (CB = 64: two)
  hipStream_t s = e->stream;
  {
    p3::InitArgs a{};
    a.feats = e->d_feats; a.x = e->d_x; a.npos = npos;
    a.wstream = e->d_arena + e->init_stream_off; a.nms_total = e->init_nms;
    a.game_w = e->dev<float>(e->game_w_off); a.game_b = e->dev<float>(e->game_b_off);
    if (!e->check(p3::launch_init(C, a, grid_for(e, npos, 1), s), "launch k_init")) return false;
  }
#ifdef P3_DIAG
  e->launch_index = 0;
#endif
  // debugging aid (tools/gpu_blockw_ab.py xdiff): stop the forward pass in front of plan block P3HIP_DEBUG_STOP_BLOCK
  static const int stop_block = getenv("P3HIP_DEBUG_STOP_BLOCK") ? atoi(getenv("P3HIP_DEBUG_STOP_BLOCK")) : -1;
  for (size_t bi = 0; bi < e->blocks.size(); ++bi) {
    if (stop_block >= 0 && (int)bi >= stop_block) return true;
    const BlockPlan& bp = e->blocks[bi];
    if (bp.kind == 3) {
      p3::Conv1x1Args c0{};
      c0.in = e->d_x; c0.out16 = e->d_t; c0.npos = npos;
      c0.wstream = e->d_arena + bp.stream_off; c0.nms_total = bp.nms;
      c0.scale = e->dev<float>(bp.bn[0].scale_off); c0.shift = e->dev<float>(bp.bn[0].shift_off);
      if (!bp.first_fused && !e->check(p3::launch_conv1x1(C, 0, c0, e->n_cu, s), "launch conv_first")) return false;
      p3::BDenseArgs d{};
      d.t = e->d_t; d.u = e->d_u; d.npos = npos;
      d.wstream = e->d_arena + bp.stream2_off; d.nms_total = bp.nms2;
      d.bias = e->dev<float>(bp.dense_bias_off);
      d.scale = e->dev<float>(bp.bn[1].scale_off); d.shift = e->dev<float>(bp.bn[1].shift_off);
      if (!(bp.first_fused && bp.dense_fused) &&
          !e->check(p3::launch_bdense(C, d, grid_for(e, npos, 1), s), "launch bdense")) return false;
      p3::Conv1x1Args c1{};
      c1.in = e->d_u; c1.out16 = e->d_x; c1.npos = npos;
      c1.wstream = e->d_arena + bp.stream3_off; c1.nms_total = bp.nms3;
      if (!bp.last_fused && !e->check(p3::launch_conv1x1(C, 1, c1, e->n_cu, s), "launch conv_last")) return false;
    } else if (bp.kind == 4) {
      const size_t half = (size_t)e->batch * wf.Cb * kNLoc;   // elements of one C_b-channel tensor
      _Float16* bufs[5] = {e->d_x, e->d_t, e->d_t + half, e->d_u, e->d_u + half};
      for (const LayerPlan& lp : bp.layers) {
        p3::LConvArgs a{};
        a.in = bufs[lp.in_buf]; a.out = bufs[lp.out_buf]; a.npos = npos;
        a.out2 = lp.out2_buf >= 0 ? bufs[lp.out2_buf] : nullptr;
        a.wstream = e->d_arena + lp.stream_off; a.nms_total = lp.nms;
        a.pre = lp.pre; a.act = lp.act; a.res = lp.res; a.dual = lp.dual;
        if (a.pre) { a.scale_in = e->dev<float>(lp.pre_bn.scale_off); a.shift_in = e->dev<float>(lp.pre_bn.shift_off); }
        if (a.act || a.dual) { a.scale_out = e->dev<float>(lp.out_bn.scale_off); a.shift_out = e->dev<float>(lp.out_bn.shift_off); }
        const bool timed = e->time_blocks && lp.kw == 3 && 2 * e->timed_blocks + 1 < (int)e->blk_ev.size();
        if (timed) hipEventRecord(e->blk_ev[2 * e->timed_blocks], s);
        if (!e->check(p3::launch_lconv(lp.kw, lp.cin, lp.cout, a, e->n_cu, s), "launch k_lconv")) return false;
        if (timed) hipEventRecord(e->blk_ev[2 * e->timed_blocks++ + 1], s);
      }
    } else if (e->blockw && bp.kind == 0) {
      const p3hip_engine::BlockwRun* run = nullptr;
      for (const auto& r : e->bw_runs) if (r.first == bi) run = &r;
      if (!run) { e->err = "internal error: no k_blockw run starts at this block"; return false; }
      const bool timed = e->time_blocks && 2 * e->timed_blocks + 1 < (int)e->blk_ev.size();
      if (timed) hipEventRecord(e->blk_ev[2 * e->timed_blocks], s);
      if (!launch_blockw(e, *run, npos)) return false;
      if (timed) hipEventRecord(e->blk_ev[2 * e->timed_blocks++ + 1], s);
      bi += run->nblk - 1;
    } else {
      int run = fused_run(e, bi);
      p3::BlockArgs a;
      const int joined = joined_launch(e, bi, npos, &a);
      if (joined > 0) run = joined;
      else a = block_args(e, bi, run, npos);
#ifdef P3_DIAG
      {
        static const int which = getenv("P3DIAG_LAUNCH") ? atoi(getenv("P3DIAG_LAUNCH")) : 1;
        constexpr size_t bytes = (size_t)p3::kStampWgs * 8 * p3::kStampSections * p3::kStampSlots * 8;
        if (!e->d_stamps && hipMalloc((void**)&e->d_stamps, bytes) == hipSuccess) hipMemset(e->d_stamps, 0, bytes);
        constexpr size_t span_bytes = (size_t)p3::kSpanWgs * p3::kSpanSlots * 8;
        if (!e->d_spans && hipMalloc((void**)&e->d_spans, span_bytes) == hipSuccess) hipMemset(e->d_spans, 0, span_bytes);
        // joined launches: ONE block launch per forward pass (index 0); P3DIAG_RUN picks the run whose phases are stamped
        static const int which_run = getenv("P3DIAG_RUN") ? atoi(getenv("P3DIAG_RUN")) : 1;
        const int idx = (a.nruns > 1) ? 0 : which;
        a.stamp_run = (a.nruns > 1) ? which_run : 0;
        a.spans = (e->launch_index == idx) ? e->d_spans : nullptr;
        a.stamps = (e->launch_index++ == idx) ? e->d_stamps : nullptr;
      }
#endif
      const bool timed = e->time_blocks && 2 * e->timed_blocks + 1 < (int)e->blk_ev.size();
      if (timed) hipEventRecord(e->blk_ev[2 * e->timed_blocks], s);
      if (!e->check(p3::launch_block(C, bp.kind, wf.inner, e->c128_wg8, a, e->n_cu, s), "launch k_block")) return false;
      if (timed) hipEventRecord(e->blk_ev[2 * e->timed_blocks++ + 1], s);
      bi += run - 1;
    }
  }
  {
    p3::Conv1x1Args c{};
    c.in = e->d_x; c.out32 = e->d_hp; c.npos = npos;
    c.wstream = e->d_arena + e->heads_stream_off; c.nms_total = e->heads_nms;
    if (!e->heads_fused && !e->check(p3::launch_conv1x1(C, 2, c, e->n_cu, s), "launch head convs")) return false;
    p3::HeadsArgs h{};
    h.x = e->d_x;
    h.conv_a = e->d_arena + e->heads_conv_a_off;
    h.image = e->dev<float>(e->heads_image_off);
    h.hp = e->d_hp; h.out = e->d_out; h.npos = npos; h.V = wf.V;
    h.res = e->run_direct ? e->d_res : nullptr;
    auto F = [&](const char* n) { return e->dev<float>(e->head_off.at(n)); };
    h.gbn_scale = F("gbn_scale"); h.gbn_shift = F("gbn_shift");
    h.gd_w = F("policy.gpool_dense.w"); h.gd_b = F("policy.gpool_dense.b");
    h.moves_w = F("policy.out_moves.w");
    h.pass_w = F("policy.out_pass.w"); h.pass_b = F("policy.out_pass.b");
    h.opt_moves_w = F("policy.opt_moves.w");
    h.opt_pass_w = F("policy.opt_pass.w"); h.opt_pass_b = F("policy.opt_pass.b");
    h.oq_embed_w = F("value.oq_embed.w"); h.oq_embed_b = F("value.oq_embed.b");
    h.oq_out_w = F("value.oq_out.w"); h.oq_out_b = F("value.oq_out.b");
    h.own_w = F("value.own.w");
    h.gamma_pre_w = F("value.gamma_pre.w"); h.gamma_pre_b = F("value.gamma_pre.b");
    h.gamma_out_w = F("value.gamma_out.w"); h.gamma_out_b = F("value.gamma_out.b");
    h.score_pre_w = F("value.score_pre.w"); h.score_pre_b = F("value.score_pre.b");
    h.score_out_w = F("value.score_out.w"); h.score_out_b = F("value.score_out.b");
    if (e->heads_fused) {
      if (!e->check(p3::launch_headsx(C, h, e->n_cu, s), "launch k_headsx")) return false;
    } else if (!e->check(p3::launch_heads(h, npos, s), "launch k_heads")) return false;
  }
  return true;
}

// The forward pass of a run: one captured graph for the full static batch when the engine was created with
// P3HIP_FLAG_LAUNCH_GRAPH (the reference's TensorRT engine replays a captured graph, trt_engine.cc:260-303), the
// kernel-by-kernel launches otherwise and for every other position count.  The first full-batch run goes out
// kernel by kernel (the launchers set their kernels' LDS attributes on first use, which a capture must not see),
// the second is captured, the rest replay.  A capture that fails falls back to the launches for good.
bool run_forward(p3hip_engine* e, int npos) {
  const bool want = (e->flags & P3HIP_FLAG_LAUNCH_GRAPH) && npos == e->batch && !e->time_blocks && !e->graph_failed;
  if (!want) return enqueue_forward(e, npos);
  // The capture bakes every kernel argument in, k_init's feature pointer among them, and run_cached points
  // e->d_feats at the cache's gathered copy around its forward pass: a graph captured for one buffer must never be
  // replayed for the other.  The graph serves the buffer it was captured on; the other goes out kernel by kernel.
  if (e->graph_exec) {
    if (e->d_feats != e->graph_feats || e->run_direct != e->graph_direct) return enqueue_forward(e, npos);
    return e->check(hipGraphLaunch(e->graph_exec, e->stream), "hipGraphLaunch");
  }
  if (!e->graph_warm) {
    e->graph_warm = true;
    return enqueue_forward(e, npos);
  }
  if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    e->graph_failed = true;
    (void)hipGetLastError();
    return enqueue_forward(e, npos);
  }
  const bool ok = enqueue_forward(e, npos);
  hipGraph_t g = nullptr;
  const hipError_t ce = hipStreamEndCapture(e->stream, &g);
  if (!ok || ce != hipSuccess || !g || hipGraphInstantiate(&e->graph_exec, g, nullptr, nullptr, 0) != hipSuccess) {
    if (g) hipGraphDestroy(g);
    e->graph_exec = nullptr;
    e->graph_failed = true;
    (void)hipGetLastError();
    return enqueue_forward(e, npos);   // nothing was executed by the capture
  }
  e->graph = g;
  e->graph_feats = e->d_feats;
  e->graph_direct = e->run_direct;
  return e->check(hipGraphLaunch(e->graph_exec, e->stream), "hipGraphLaunch");
}

}  // namespace

extern "C" {

const char* p3hip_create_error(void) { return g_create_error.c_str(); }
int p3hip_graph_state(const p3hip_engine* e) { return e->graph_failed ? -1 : (e->graph_exec ? 1 : 0); }

p3hip_engine* p3hip_create(const char* weights_path, int batch_size, int version,
                           int device_ordinal, uint32_t flags) {
  g_create_error.clear();
  if (version != 1) { g_create_error = "only model version 1 (15 planes + 8 scalars) is supported"; return nullptr; }
  if (batch_size < 1 || batch_size > (1 << 16)) { g_create_error = "bad batch size"; return nullptr; }
  p3hip_engine* e = new p3hip_engine();
  e->path = weights_path;
  e->batch = batch_size;
  e->device = device_ordinal;
  e->flags = flags;
  e->c128_wg8 = getenv("P3HIP_C128_WG8") != nullptr;
  e->bcast_fuse = getenv("P3HIP_NO_BFUSE") == nullptr;
  auto fail = [&](const std::string& m) {
    g_create_error = m;
    p3hip_destroy(e);
    return (p3hip_engine*)nullptr;
  };
  if (!e->wf.load(weights_path, e->err)) return fail(e->err);
  // host-side plan first (weight repacking; no HIP call): a bad file fails here, GPU or not
  Arena ar;
  if (!build_plan(e, ar)) return fail(e->err);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_ordinal)
    return fail("no HIP device " + std::to_string(device_ordinal) + " (the HIP engine has no CPU fallback)");
  if (!e->check(hipSetDevice(device_ordinal), "hipSetDevice")) return fail(e->err);
  hipDeviceProp_t prop;
  if (!e->check(hipGetDeviceProperties(&prop, device_ordinal), "hipGetDeviceProperties")) return fail(e->err);
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(std::string("device is ") + prop.gcnArchName + ", this engine is built for gfx950 only");
  e->n_cu = prop.multiProcessorCount;
  const int C = e->wf.C;
  const size_t B = batch_size;
  bool ok = e->check(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate") &&
            e->check(hipEventCreate(&e->ev0), "hipEventCreate") &&
            e->check(hipEventCreate(&e->ev1), "hipEventCreate") &&
            e->check(hipMalloc((void**)&e->d_arena, ar.host.size()), "hipMalloc arena") &&
            // on the engine's own stream: it is non-blocking (no implicit ordering with the null stream a plain
            // hipMemcpy / hipMemset runs on), and the first run must find the weights there
            e->check(hipMemcpyAsync(e->d_arena, ar.host.data(), ar.host.size(), hipMemcpyHostToDevice, e->stream), "upload weights") &&
            e->check(hipHostMalloc((void**)&e->h_feats, B * kFeatBytes, hipHostMallocDefault), "hipHostMalloc") &&
            e->check(hipHostMalloc((void**)&e->h_feats_compact, B * kFeatBytes, hipHostMallocDefault), "hipHostMalloc") &&
            e->check(hipHostMalloc((void**)&e->h_out, B * p3::kResultFloats * 4, hipHostMallocDefault), "hipHostMalloc") &&
            e->check(hipMalloc((void**)&e->d_res, B * p3::kResultFloats * 4), "hipMalloc results") &&
            e->check(hipMalloc((void**)&e->d_feats, B * kFeatBytes), "hipMalloc feats") &&
            e->check(hipMalloc((void**)&e->d_x, B * C * kNLoc * 2), "hipMalloc x") &&
            e->check(hipMalloc((void**)&e->d_t, B * C * kNLoc * 2), "hipMalloc t") &&
            e->check(hipMalloc((void**)&e->d_u, B * C * kNLoc * 2), "hipMalloc u") &&
            (e->wf.btype != 1 || e->check(hipMalloc((void**)&e->d_s, B * e->wf.Cb * kNLoc * 2), "hipMalloc s")) &&
            e->check(hipMalloc((void**)&e->d_hp, B * 96 * kNLoc * 4), "hipMalloc hp") &&
            e->check(hipMalloc((void**)&e->d_out, B * p3::kOutStride * 4), "hipMalloc out");
  if (!ok) return fail(e->err);
  memset(e->h_feats, 0, B * kFeatBytes);
  if (!e->check(hipMemsetAsync(e->d_feats, 0, B * kFeatBytes, e->stream), "hipMemset feats") ||
      !e->check(hipMemsetAsync(e->d_out, 0, B * p3::kOutStride * 4, e->stream), "hipMemset out") ||
      !e->check(hipStreamSynchronize(e->stream), "upload sync")) return fail(e->err);
  e->slots = p3::SlotStates((int)B);
  e->slot_sym.assign(B, 0);
  e->row_sym.assign(B, 0);
  return e;
}

void p3hip_destroy(p3hip_engine* e) {
  if (!e) return;
  if (e->stream || e->d_arena) (void)hipSetDevice(e->device);
  if (e->stream) hipStreamSynchronize(e->stream);
  hipFree(e->d_arena); hipFree(e->d_feats); hipFree(e->d_x); hipFree(e->d_t); hipFree(e->d_u); hipFree(e->d_s);
  hipFree(e->d_hp); hipFree(e->d_out); hipFree(e->d_res);
  hipFree(e->d_bw_stamps);
  if (e->bw_mod) hipModuleUnload(e->bw_mod);
  {
    auto& c = e->cache;
    hipFree(c.d_tkeys); hipFree(c.d_tmeta); hipFree(c.d_tvals); hipFree(c.d_keys); hipFree(c.d_hit); hipFree(c.d_victim);
    hipFree(c.d_lists); hipFree(c.d_sym); hipFree(c.d_feats2);
    if (c.h_keys) hipHostFree(c.h_keys);
    if (c.h_hit) hipHostFree(c.h_hit);
    if (c.h_victim) hipHostFree(c.h_victim);
    if (c.h_lists) hipHostFree(c.h_lists);
    if (c.h_sym) hipHostFree(c.h_sym);
    delete[] c.h_slot_keys;
  }
  if (e->h_feats) hipHostFree(e->h_feats);
  if (e->h_feats_compact) hipHostFree(e->h_feats_compact);
  if (e->h_out) hipHostFree(e->h_out);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  for (hipEvent_t ev : e->blk_ev) hipEventDestroy(ev);
  if (e->graph_exec) hipGraphExecDestroy(e->graph_exec);
  if (e->graph) hipGraphDestroy(e->graph);
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
}

int p3hip_kind(const p3hip_engine*) { return P3HIP_KIND_HIP; }
const char* p3hip_path(const p3hip_engine* e) { return e->path.c_str(); }
int p3hip_batch_size(const p3hip_engine* e) { return e->batch; }
const char* p3hip_last_error(const p3hip_engine* e) { return e->err.c_str(); }

int p3hip_load_slot(p3hip_engine* e, int slot, const p3hip_features* f) {
  if (slot < 0 || slot >= e->batch) return 1;
  memcpy(e->h_feats + (size_t)slot * kFeatBytes, f, kFeatBytes);
  e->slot_sym[slot] = 0;
  if (e->cache.on) e->cache.h_slot_keys[slot] = p3::CacheKey{0, 0, 0};   // no key: evaluated, never cached
  e->slots.loaded(slot);
  return 0;
}

int p3hip_load_slot_keyed(p3hip_engine* e, int slot, const p3hip_features* f, uint64_t key_lo, uint64_t key_hi, int symmetry) {
  if (slot < 0 || slot >= e->batch) return 1;
  if (symmetry < 0 || symmetry > 7) return 1;
  memcpy(e->h_feats + (size_t)slot * kFeatBytes, f, kFeatBytes);
  e->slot_sym[slot] = (unsigned char)symmetry;
  if (e->cache.on) e->cache.h_slot_keys[slot] = p3::CacheKey{key_lo, key_hi, (unsigned long long)symmetry};
  e->slots.loaded(slot);
  return 0;
}

int p3hip_cache_enable(p3hip_engine* e, int log2_entries) {
  if (!e->bind()) return 1;
  if (e->cache.on) { e->err = "cache already enabled"; return 1; }
  if (log2_entries < 4 || log2_entries > 26) { e->err = "cache size: 2^4 .. 2^26 entries"; return 1; }
  auto& c = e->cache;
  const size_t cap = (size_t)1 << log2_entries, B = (size_t)e->batch;
  bool ok = e->check(hipMalloc((void**)&c.d_tkeys, cap * 16), "hipMalloc cache keys") &&
            e->check(hipMalloc((void**)&c.d_tmeta, cap * 4), "hipMalloc cache meta") &&
            e->check(hipMalloc((void**)&c.d_tvals, cap * p3::kOutStride * 4), "hipMalloc cache records") &&
            e->check(hipMalloc((void**)&c.d_keys, B * sizeof(p3::CacheKey)), "hipMalloc") &&
            e->check(hipMalloc((void**)&c.d_hit, B * 4), "hipMalloc") && e->check(hipMalloc((void**)&c.d_victim, B * 4), "hipMalloc") &&
            e->check(hipMalloc((void**)&c.d_lists, 5 * B * 4), "hipMalloc") && e->check(hipMalloc((void**)&c.d_sym, B * 4), "hipMalloc") &&
            e->check(hipMalloc((void**)&c.d_feats2, B * kFeatBytes), "hipMalloc") &&
            e->check(hipHostMalloc((void**)&c.h_keys, B * sizeof(p3::CacheKey), hipHostMallocDefault), "hipHostMalloc") &&
            e->check(hipHostMalloc((void**)&c.h_hit, B * 4, hipHostMallocDefault), "hipHostMalloc") &&
            e->check(hipHostMalloc((void**)&c.h_victim, B * 4, hipHostMallocDefault), "hipHostMalloc") &&
            e->check(hipHostMalloc((void**)&c.h_lists, 5 * B * 4, hipHostMallocDefault), "hipHostMalloc") &&
            e->check(hipHostMalloc((void**)&c.h_sym, B * 4, hipHostMallocDefault), "hipHostMalloc") &&
            // (the engine's stream, as in p3hip_create: the table must be empty before the first probe kernel)
            e->check(hipMemsetAsync(c.d_tkeys, 0, cap * 16, e->stream), "hipMemset") &&
            e->check(hipMemsetAsync(c.d_tmeta, 0, cap * 4, e->stream), "hipMemset") &&
            e->check(hipStreamSynchronize(e->stream), "cache table sync");
  if (!ok) {
    // free what was allocated (an over-large table fails at the records): a later, smaller enable starts clean
    const std::string why = e->err;
    hipFree(c.d_tkeys); hipFree(c.d_tmeta); hipFree(c.d_tvals); hipFree(c.d_keys); hipFree(c.d_hit); hipFree(c.d_victim);
    hipFree(c.d_lists); hipFree(c.d_sym); hipFree(c.d_feats2);
    if (c.h_keys) hipHostFree(c.h_keys);
    if (c.h_hit) hipHostFree(c.h_hit);
    if (c.h_victim) hipHostFree(c.h_victim);
    if (c.h_lists) hipHostFree(c.h_lists);
    if (c.h_sym) hipHostFree(c.h_sym);
    c.d_tkeys = nullptr; c.d_tmeta = nullptr; c.d_tvals = nullptr; c.d_keys = nullptr; c.d_hit = nullptr; c.d_victim = nullptr;
    c.d_lists = nullptr; c.d_sym = nullptr; c.d_feats2 = nullptr;
    c.h_keys = nullptr; c.h_hit = nullptr; c.h_victim = nullptr; c.h_lists = nullptr; c.h_sym = nullptr;
    (void)hipGetLastError();   // the failed allocation's sticky error
    e->err = why;
    return 1;
  }
  c.h_slot_keys = new p3::CacheKey[B]();
  c.out_row.assign(B, -1);
  c.was_hit.assign(B, 0);
  c.mask = (unsigned)(cap - 1);
  c.run = 0;
  c.on = true;
  return 0;
}

int p3hip_cache_stats(const p3hip_engine* e, uint64_t out[4]) {
  out[0] = e->cache.lookups; out[1] = e->cache.hits; out[2] = e->cache.inserts; out[3] = e->cache.on ? (uint64_t)e->cache.mask + 1 : 0;
  return e->cache.on ? 0 : 1;
}

// Compacts every dirty slot (loaded and not yet fetched, slot_state.h) into the dense upload.
static int gather_loaded(p3hip_engine* e) {
  const bool all = (e->flags & P3HIP_FLAG_RUN_ALL_SLOTS) != 0;
  // When every slot of the static batch is evaluated (the common case: NNInterface fills the whole batch, the self-play
  // scheduler always does) the dense upload IS h_feats: no second host copy of 1.9 MB per run.
  std::vector<std::pair<int, int>> moved;
  bool identity = true;
  const int n = e->slots.gather(all, [&](int s, int row) {
    if (s != row) identity = false;
    moved.emplace_back(s, row);
    e->row_sym[row] = e->slot_sym[s];
    if (e->cache.on) {
      e->cache.h_keys[row] = e->cache.h_slot_keys[s];
      e->cache.out_row[row] = row;   // p3hip_run re-maps (misses first, then hits)
      e->cache.was_hit[row] = 0;
      e->cache.h_sym[row] = (unsigned)e->cache.h_slot_keys[s].sym;
    }
  });
  e->feats_identity = identity && n == e->batch && !e->cache.on;
  if (!e->feats_identity)
    for (const auto& m : moved)
      memcpy(e->h_feats_compact + (size_t)m.second * kFeatBytes, e->h_feats + (size_t)m.first * kFeatBytes, kFeatBytes);
  e->last_n = n;
  return n;
}

int p3hip_upload(p3hip_engine* e) {
  if (!e->bind()) return 1;
  int n = gather_loaded(e);
  if (n == 0) return 0;
  if (!e->check(hipMemcpyAsync(e->d_feats, e->feats_identity ? e->h_feats : e->h_feats_compact, (size_t)n * kFeatBytes,
                               hipMemcpyHostToDevice, e->stream), "H2D features")) return 1;
  return e->check(hipStreamSynchronize(e->stream), "sync") ? 0 : 1;
}

int p3hip_forward_resident(p3hip_engine* e, int n_positions) {
  if (n_positions < 1 || n_positions > e->batch || !e->bind()) return 1;
  return run_forward(e, n_positions) ? 0 : 1;
}

int p3hip_sync(p3hip_engine* e) { return e->bind() && e->check(hipStreamSynchronize(e->stream), "sync") ? 0 : 1; }

// p3hip_run with the cache on: probe, evaluate the misses only, fill the hits from the table, store the misses.
// d_out / h_out rows: the misses first (in row order), then the hits (in row order); cache.out_row maps.
static int run_cached(p3hip_engine* e, int n) {
  auto& c = e->cache;
  hipStream_t s = e->stream;
  const size_t B = (size_t)e->batch;
  ++c.run;
  if (!e->check(hipMemcpyAsync(e->d_feats, e->h_feats_compact, (size_t)n * kFeatBytes, hipMemcpyHostToDevice, s), "H2D features") ||
      !e->check(hipMemcpyAsync(c.d_keys, c.h_keys, (size_t)n * sizeof(p3::CacheKey), hipMemcpyHostToDevice, s), "H2D keys")) return 1;
  p3::CacheArgs a{};
  a.keys = c.d_keys; a.n = n; a.tkeys = c.d_tkeys; a.tmeta = c.d_tmeta; a.tvals = c.d_tvals; a.mask = c.mask; a.run = c.run;
  a.hit = c.d_hit; a.victim = c.d_victim; a.out = e->d_out; a.out_sym = c.d_sym;
  if (!e->check(p3::launch_cache_probe(a, s), "launch k_cache_probe") ||
      !e->check(hipMemcpyAsync(c.h_hit, c.d_hit, (size_t)n * 4, hipMemcpyDeviceToHost, s), "D2H hits") ||
      !e->check(hipMemcpyAsync(c.h_victim, c.d_victim, (size_t)n * 4, hipMemcpyDeviceToHost, s), "D2H victims") ||
      !e->check(hipStreamSynchronize(s), "sync")) return 1;
  int* miss_rows = c.h_lists;
  int* hit_idx = c.h_lists + B;
  int* ins_rows = c.h_lists + 2 * B;
  int* ins_src = c.h_lists + 3 * B;
  int* ins_idx = c.h_lists + 4 * B;
  int nm = 0, nh = 0, ni = 0;
  for (int r = 0; r < n; ++r) nm += c.h_hit[r] < 0;
  int mi = 0, hi = 0;
  // entries this run reads (hits) or has already given to an insert: one writer per entry, no reader evicted
  std::unordered_set<int> claimed;
  for (int r = 0; r < n; ++r) if (c.h_hit[r] >= 0) claimed.insert(c.h_hit[r]);
  for (int r = 0; r < n; ++r) {
    const bool keyed = (c.h_keys[r].lo | c.h_keys[r].hi) != 0;
    c.lookups += keyed;
    if (c.h_hit[r] >= 0) {
      c.was_hit[r] = 1;
      c.out_row[r] = nm + hi;
      hit_idx[hi++] = c.h_hit[r];
      ++c.hits;
    } else {
      c.was_hit[r] = 0;
      c.out_row[r] = mi;
      c.h_sym[mi] = (unsigned)c.h_keys[r].sym;
      miss_rows[mi] = r;
      const int v = c.h_victim[r];
      if (keyed && v >= 0 && claimed.insert(v).second) {
        ins_rows[ni] = r; ins_src[ni] = mi; ins_idx[ni] = v; ++ni;
      }
      ++mi;
    }
  }
  nh = hi;
  if (!e->check(hipMemcpyAsync(c.d_lists, c.h_lists, 5 * B * 4, hipMemcpyHostToDevice, s), "H2D lists")) return 1;
  if (nm > 0) {
    a.rows = c.d_lists; a.m = nm; a.feats_in = e->d_feats; a.feats_out = c.d_feats2;
    if (!e->check(p3::launch_cache_gather(a, s), "launch k_cache_gather")) return 1;
    unsigned char* keep = e->d_feats;
    e->d_feats = c.d_feats2;
    const bool ok = run_forward(e, nm);
    e->d_feats = keep;
    if (!ok) return 1;
  }
  if (nh > 0) {
    a.idx = c.d_lists + B; a.m = nh; a.out_row0 = nm;
    if (!e->check(p3::launch_cache_fill(a, s), "launch k_cache_fill")) return 1;
  }
  if (ni > 0) {
    a.rows = c.d_lists + 2 * B; a.src = c.d_lists + 3 * B; a.idx = c.d_lists + 4 * B; a.m = ni;
    if (!e->check(p3::launch_cache_insert(a, s), "launch k_cache_insert")) return 1;
    c.inserts += ni;
  }
  if (!e->check(hipMemcpy2DAsync(e->h_out, p3::kResultFloats * 4, e->d_out, p3::kOutStride * 4,
                                 p3::kResultFloats * 4, n, hipMemcpyDeviceToHost, s), "D2H results")) return 1;
  if (nh > 0 && !e->check(hipMemcpyAsync(c.h_sym + nm, c.d_sym + nm, (size_t)nh * 4, hipMemcpyDeviceToHost, s), "D2H symmetries")) return 1;
  return e->check(hipStreamSynchronize(s), "sync") ? 0 : 1;
}

int p3hip_run(p3hip_engine* e) {
  if (!e->bind()) return 1;
  int n = gather_loaded(e);
  if (n == 0) return 0;
  if (e->cache.on) return run_cached(e, n);
  if (!e->check(hipMemcpyAsync(e->d_feats, e->feats_identity ? e->h_feats : e->h_feats_compact, (size_t)n * kFeatBytes,
                               hipMemcpyHostToDevice, e->stream), "H2D features")) return 1;
  // The heads kernel writes the result records (the first kResultFloats of an output row) a second time into a dense
  // device buffer (HeadsArgs::res), so the copy TrtEngineImpl::RunInference queues behind its graph (trt_engine.cc:283-297)
  // is one contiguous 7.7 MB transfer at the link's rate; the strided copy of rounds 1-3 (1024 rows of 7,556 B out of a
  // 13,664 B pitch) took 0.45 ms, and 4-byte stores straight into host memory from the kernel took as long (round 4,
  // gpurun_out/r4d/breakdown.log).  P3HIP_NO_DIRECT_RESULTS=1: the strided copy (A/B, tests).
  static const bool no_direct = getenv("P3HIP_NO_DIRECT_RESULTS") != nullptr;
  // P3HIP_TIME_RUN=1 (tools/gpu_run_breakdown.py): the stream is drained after every stage and the stages' wall times are
  // summed into the engine's error string on request — a measurement aid, never set in production
  static const bool time_run = getenv("P3HIP_TIME_RUN") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_start = now();
  if (time_run) {
    hipStreamSynchronize(e->stream);
    e->t_h2d += std::chrono::duration<double>(now() - t_start).count();
    t_start = now();
  }
  e->run_direct = e->d_res != nullptr && !no_direct;
  const bool ok = run_forward(e, n);
  if (time_run) {
    hipStreamSynchronize(e->stream);
    e->t_fwd += std::chrono::duration<double>(now() - t_start).count();
    t_start = now();
  }
  const bool direct = e->run_direct;
  e->run_direct = false;
  if (!ok) return 1;
  if (direct) {
    if (!e->check(hipMemcpyAsync(e->h_out, e->d_res, (size_t)n * p3::kResultFloats * 4, hipMemcpyDeviceToHost, e->stream), "D2H results")) return 1;
  } else if (!e->check(hipMemcpy2DAsync(e->h_out, p3::kResultFloats * 4, e->d_out, p3::kOutStride * 4,
                                        p3::kResultFloats * 4, n, hipMemcpyDeviceToHost, e->stream), "D2H results")) return 1;
  const bool sync_ok = e->check(hipStreamSynchronize(e->stream), "sync");
  if (time_run) {
    e->t_d2h += std::chrono::duration<double>(now() - t_start).count();
    ++e->t_runs;
    char buf[200];
    snprintf(buf, sizeof buf, "timing: runs %ld  h2d %.1f us  forward %.1f us  d2h %.1f us", e->t_runs, e->t_h2d / e->t_runs * 1e6,
             e->t_fwd / e->t_runs * 1e6, e->t_d2h / e->t_runs * 1e6);
    e->err = buf;
  }
  return sync_ok ? 0 : 1;
}

int p3hip_get_slot(p3hip_engine* e, int slot, p3hip_result* out) {
  if (slot < 0 || slot >= e->batch) return 1;
  int row = e->out_row_of(slot);
  if (row < 0) return 2;
  const float* r = e->h_out + (size_t)row * p3::kResultFloats;
  memcpy(out->move_logits, r + p3::kOffMoveLogits, 362 * 4);
  memcpy(out->move_probs, r + p3::kOffMoveProbs, 362 * 4);
  memcpy(out->value_probs, r + p3::kOffValueProbs, 2 * 4);
  memcpy(out->score_probs, r + p3::kOffScoreProbs, 800 * 4);
  memcpy(out->opt_move_probs, r + p3::kOffOptProbs, 362 * 4);
  out->err2_outcome = r[p3::kOffErr2];
  e->slots.fetched(slot);
  return 0;
}

int p3hip_get_slot_keyed(p3hip_engine* e, int slot, p3hip_result* out, int* symmetry, int* from_cache) {
  if (slot < 0 || slot >= e->batch) return 1;
  const int row = e->slots.row(slot), orow = e->out_row_of(slot);
  if (row < 0) return 2;
  if (symmetry) *symmetry = e->cache.on ? (int)e->cache.h_sym[orow] : (int)e->row_sym[row];
  if (from_cache) *from_cache = e->cache.on ? e->cache.was_hit[row] : 0;
  return p3hip_get_slot(e, slot, out);
}

int p3hip_get_ownership(p3hip_engine* e, int slot, float out[P3HIP_NUM_LOCS]) {
  if (slot < 0 || slot >= e->batch) return 1;
  int row = e->out_row_of(slot);
  if (row < 0) return 2;
  if (!e->bind()) return 1;
  if (!e->check(hipMemcpy(out, e->d_out + (size_t)row * p3::kOutStride + p3::kOffOwnership,
                          kNLoc * 4, hipMemcpyDeviceToHost), "D2H ownership")) return 1;
  e->slots.fetched(slot);
  return 0;
}

int p3hip_get_raw(p3hip_engine* e, int slot, float* out) {
  if (slot < 0 || slot >= e->batch) return 1;
  int row = e->out_row_of(slot);
  if (row < 0) return 2;
  std::vector<float> rec(p3::kOutStride);
  if (!e->bind()) return 1;
  if (!e->check(hipMemcpy(rec.data(), e->d_out + (size_t)row * p3::kOutStride, p3::kOutStride * 4,
                          hipMemcpyDeviceToHost), "D2H raw")) return 1;
  memcpy(out, rec.data() + p3::kOffMoveLogits, 362 * 4);
  memcpy(out + 362, rec.data() + p3::kOffOptLogits, 362 * 4);
  memcpy(out + 724, rec.data() + p3::kOffOutcomeLogits, 2 * 4);
  memcpy(out + 726, rec.data() + p3::kOffScoreLogits, 800 * 4);
  memcpy(out + 1526, rec.data() + p3::kOffOwnership, 361 * 4);
  out[1887] = rec[p3::kOffErr2];
  out[1888] = rec[p3::kOffGamma];
  return 0;
}

void p3hip_flops_per_position(const p3hip_engine* e, double* total, double* conv3x3) {
  const WeightFile& w = e->wf;
  const double C = w.C, Cb = w.Cb, H = w.H, V = w.V, L = kNLoc;
  double mac = L * 25 * 15 * C + 8 * C, mac3 = 0;
  for (int i = 0; i < w.nblocks; ++i) {
    if (w.is_broadcast(i)) mac += L * 2 * C * C + C * L * L;
    else if (w.btype == 0) { mac += L * 2 * C * Cb; mac3 += L * w.inner * 9 * Cb * Cb; }
    else if (w.btype == 1) { mac += L * 2 * C * Cb; mac3 += L * 4 * 9 * Cb * Cb; }
    else { mac3 += L * w.inner * 9 * C * C; }
  }
  mac += L * 3 * C * H + L * H * 5 + 2 * H * H + 2 * H * 4 + 2 * H * V * 2 + V * (14 + 51 + 1) +
         (2 * H + 1) * V + 800 * V;
  if (total) *total = 2.0 * (mac + mac3);
  if (conv3x3) *conv3x3 = 2.0 * mac3;
}

double p3hip_time_trunk_kernel(p3hip_engine* e, int n_positions, int iters,
                               double* flops_per_launch, const char** kernel_name) {
  const WeightFile& wf = e->wf;
  const BlockPlan* bp = nullptr;
  int nfused = 0, n3x3 = 0, c3 = 0;
  for (const BlockPlan& b : e->blocks) {
    if (b.kind == 0 || b.kind == 1) { if (!bp) bp = &b; ++nfused; }   // fused block kernel
    if (b.kind == 4)
      for (const LayerPlan& lp : b.layers)
        if (lp.kw == 3) { ++n3x3; c3 = lp.cin; }
  }
  if (n_positions < 1 || n_positions > e->batch || iters < 1 || !e->bind()) return -1.0;
  if (!bp && n3x3 > 0) {
    // layer-wise trunks (C = 384, classic): the dominant kernel is the 3x3 layer conv, k_lconv<3, ..>
    while ((int)e->blk_ev.size() < 2 * n3x3) {
      hipEvent_t ev;
      if (!e->check(hipEventCreate(&ev), "hipEventCreate")) return -1.0;
      e->blk_ev.push_back(ev);
    }
    if (!enqueue_forward(e, n_positions)) return -1.0;   // warm-up
    double total_ms = 0.0;
    long launches = 0;
    for (int i = 0; i < iters; ++i) {
      e->time_blocks = true;
      e->timed_blocks = 0;
      const bool ok = enqueue_forward(e, n_positions);
      e->time_blocks = false;
      if (!ok || !e->check(hipStreamSynchronize(e->stream), "sync")) return -1.0;
      for (int b = 0; b < e->timed_blocks; ++b) {
        float ms = 0;
        hipEventElapsedTime(&ms, e->blk_ev[2 * b], e->blk_ev[2 * b + 1]);
        total_ms += ms;
        ++launches;
      }
    }
    if (flops_per_launch) *flops_per_launch = 2.0 * n_positions * kNLoc * 9.0 * c3 * c3;
    if (kernel_name) *kernel_name = c3 == 192 ? "k_lconv<3,192,192>" : "k_lconv<3,64,64>";
    return launches ? total_ms / launches : -1.0;
  }
  if (!bp) return -1.0;
  // Time the kernel where it runs: whole forward passes over the resident batch, with a HIP
  // event pair (on the engine's stream) around each fused-block launch.  The average over all
  // launches is what rocprofv3 --kernel-trace --stats reports for the same run.
  while ((int)e->blk_ev.size() < 2 * nfused) {
    hipEvent_t ev;
    if (!e->check(hipEventCreate(&ev), "hipEventCreate")) return -1.0;
    e->blk_ev.push_back(ev);
  }
  if (!enqueue_forward(e, n_positions)) return -1.0;   // warm-up
  double total_ms = 0.0;
  long launches = 0;
  for (int i = 0; i < iters; ++i) {
    e->time_blocks = true;
    e->timed_blocks = 0;
    const bool ok = enqueue_forward(e, n_positions);
    e->time_blocks = false;
    if (!ok || !e->check(hipStreamSynchronize(e->stream), "sync")) return -1.0;
    for (int b = 0; b < e->timed_blocks; ++b) {
      float ms = 0;
      hipEventElapsedTime(&ms, e->blk_ev[2 * b], e->blk_ev[2 * b + 1]);
      total_ms += ms;
      ++launches;
    }
  }
  const double n3 = (wf.btype == 0) ? wf.inner : 4;
  // every conv the block kernel executes: the inner 3x3s plus the 1x1 reduce and expand
  // a launch covers `nfused / launches-per-forward` blocks on average
  const double blocks_per_launch = launches ? (double)nfused * iters / launches : 1.0;
  // plus the broadcast blocks' C -> C convs that ride in the block launches
  int nbconv = 0;
  for (const BlockPlan& b : e->blocks) nbconv += (b.head_of >= 0) + (b.tail_of >= 0);
  const double bconv_per_launch = launches ? (double)nbconv * iters / launches : 0.0;
  // ... and their dense where it rides in the tail (algorithmic 361 x 361 per channel, not the padded K = 384)
  int ndense = 0;
  for (const BlockPlan& b : e->blocks) ndense += b.kind == 3 && b.first_fused && b.dense_fused;
  const double dense_per_launch = launches ? (double)ndense * iters / launches : 0.0;
  if (flops_per_launch)
    *flops_per_launch = 2.0 * n_positions * kNLoc *
                        (blocks_per_launch * (n3 * 9.0 * wf.Cb * wf.Cb + 2.0 * wf.C * wf.Cb) + bconv_per_launch * (double)wf.C * wf.C +
                         dense_per_launch * (double)wf.C * kNLoc);
  if (kernel_name) *kernel_name = e->blockw ? (wf.inner == 3 ? "k_blockw_L3" : wf.inner == 2 ? "k_blockw_L2" : "k_blockw_L1")
                                            : p3::block_kernel_name(wf.C, bp->kind, wf.inner);
  return launches ? total_ms / launches : -1.0;
}

// debugging aid: the residual stream x of the last forward pass, n_positions x C x 361 fp16 in the device layout
// [pos][C / 8][361][8], as floats
int p3hip_debug_x(p3hip_engine* e, float* out, int n_positions) {
  if (!e->bind() || n_positions < 1 || n_positions > e->batch) return 1;
  const size_t n = (size_t)n_positions * e->wf.C * kNLoc;
  std::vector<_Float16> h(n);
  hipStreamSynchronize(e->stream);
  if (hipMemcpy(h.data(), e->d_x, n * 2, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  for (size_t i = 0; i < n; ++i) out[i] = (float)h[i];
  return 0;
}

int p3hip_blockw_stamps(p3hip_engine* e, unsigned long long* out, int n) {
  if (!e->bind() || !e->d_bw_stamps) return 1;
  constexpr int total = 8 * 16 * 4 * 24;
  hipStreamSynchronize(e->stream);
  return hipMemcpy(out, e->d_bw_stamps, (size_t)(n < total ? n : total) * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}

#ifdef P3_DIAG
// diagnostic build only: the phase stamps of the last forward pass's launch P3DIAG_LAUNCH
int p3hip_debug_block_stamps(p3hip_engine* e, unsigned long long* out, int n) {
  if (!e->bind() || !e->d_stamps) return 1;
  constexpr int total = p3::kStampWgs * 8 * p3::kStampSections * p3::kStampSlots;
  hipStreamSynchronize(e->stream);
  return hipMemcpy(out, e->d_stamps, (size_t)(n < total ? n : total) * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
int p3hip_debug_block_spans(p3hip_engine* e, unsigned long long* out, int n) {
  if (!e->bind() || !e->d_spans) return 1;
  constexpr int total = p3::kSpanWgs * p3::kSpanSlots;
  hipStreamSynchronize(e->stream);
  return hipMemcpy(out, e->d_spans, (size_t)(n < total ? n : total) * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif

}  // extern "C"
