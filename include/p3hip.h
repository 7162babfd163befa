/*
 * p3hip.h — C ABI of the MI355X (gfx950) policy/value-net engine.
 *
 * This is the drop-in boundary for the reference's `nn::Engine` virtuals
 * (cc/nn/engine/engine.h:22-43).  Every entry point mirrors one virtual 1:1 so that a
 * ~50-line C++ adapter `HipEngine : nn::Engine` can forward to it.
 * Plain pointers and sizes only; no C++/torch types cross this line.
 *
 *   Engine::LoadBatch(batch_id, GoFeatures)   -> p3hip_load_slot   (engine.h:35)
 *   Engine::RunInference()                    -> p3hip_run         (engine.h:36)
 *   Engine::GetBatch(batch_id, NNInferResult) -> p3hip_get_slot    (engine.h:37)
 *   Engine::GetOwnership(batch_id, own)       -> p3hip_get_ownership (engine.h:38-39)
 *   CreateEngine(kind, path, batch, version)  -> p3hip_create      (engine_factory.cc:56-73)
 *   Engine::~Engine                           -> p3hip_destroy
 *   Engine::kind()/path()                     -> p3hip_kind / p3hip_path (engine.h:32-33)
 *
 * Threading contract (same as the reference): load_slot/get_slot may be
 * called concurrently from many threads, each on its own slot, without locks; p3hip_run is
 * called by one thread at a time and never overlaps get_slot.  A load_slot on a slot that
 * is not part of the running batch may overlap p3hip_run.
 *
 * Which slots a run evaluates: every slot that has been loaded and whose result has not been
 * fetched yet (p3hip_get_slot / p3hip_get_ownership), compacted into a dense batch.  A slot
 * stays in that set until its result is fetched, not merely until a run has picked it up: the
 * reference's infer thread may start a run while a worker's LoadBatch is landing
 * (cc/nn/nn_interface.cc:351-361) and only count that worker as loaded for the NEXT run, which
 * must then still produce its result.  A slot that has not been evaluated by the last run
 * answers p3hip_get_slot with 2.  Every entry point binds the engine's HIP device on the
 * calling thread, so an engine may be driven from any host thread.
 *
 * Error convention: the reference aborts on failure (trt_engine.cc:27-35).  The C ABI
 * returns status codes and keeps a per-engine message (p3hip_last_error); the C++ adapter
 * CHECK-fails on non-zero, reproducing the reference behaviour.
 */
#ifndef P3HIP_H_
#define P3HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3HIP_BOARD_LEN 19
#define P3HIP_NUM_LOCS 361          /* constants::kNumBoardLocs        constants.h:39 */
#define P3HIP_NUM_MOVES 362         /* constants::kMaxMovesPerPosition constants.h:42 */
#define P3HIP_NUM_LAST_MOVES 5      /* constants::kNumLastMoves        constants.h:66 */
#define P3HIP_NUM_VALUE_LOGITS 2    /* constants::kNumValueLogits      constants.h:57 */
#define P3HIP_NUM_SCORE_LOGITS 800  /* constants::kNumScoreLogits      constants.h:60 */
#define P3HIP_NUM_PLANES 15         /* constants::kNumInputFeaturePlanesV1  constants.h:48 */
#define P3HIP_NUM_SCALARS 8         /* constants::kNumInputFeatureScalarsV1 constants.h:54 */

/* Engine::Kind extended with the HIP engine (engine.h:24-30 has 0..4). */
#define P3HIP_KIND_HIP 5

/* game::Loc (cc/game/loc.h:16-22).  pass = {19,0}, noop = {-1,-1} (loc.h:46-47). */
typedef struct p3hip_loc {
  int32_t i;
  int32_t j;
} p3hip_loc;

/* POD mirror of nn::GoFeatures (cc/nn/engine/go_features.h:12-22); fixed 19x19.
 * Colour values: -1 white, 0 empty, +1 black (constants.h:17-27). */
typedef struct p3hip_features {
  int32_t bsize;
  int8_t color;
  float komi;
  int8_t board[P3HIP_NUM_LOCS];
  p3hip_loc last_moves[P3HIP_NUM_LAST_MOVES];
  int8_t stones_atari[P3HIP_NUM_LOCS];
  int8_t stones_two_liberties[P3HIP_NUM_LOCS];
  int8_t stones_three_liberties[P3HIP_NUM_LOCS];
  int8_t stones_laddered[P3HIP_NUM_LOCS];
} p3hip_features;

/* POD mirror of nn::NNInferResult (cc/nn/engine/engine.h:12-20). */
typedef struct p3hip_result {
  float move_logits[P3HIP_NUM_MOVES];
  float move_probs[P3HIP_NUM_MOVES];
  float value_probs[P3HIP_NUM_VALUE_LOGITS]; /* [0]=P(loss) [1]=P(win), side to move */
  float score_probs[P3HIP_NUM_SCORE_LOGITS];
#ifdef __cplusplus
  alignas(16)
#else
  _Alignas(16)
#endif
      float opt_move_probs[P3HIP_NUM_MOVES];
  float err2_outcome;
} p3hip_result;

typedef struct p3hip_engine p3hip_engine;

/* p3hip_create flags */
#define P3HIP_FLAG_NONE 0u
#define P3HIP_FLAG_RUN_ALL_SLOTS 2u /* always run the full static batch (TRT behaviour,
                                       trt_engine.cc:238-304); default compacts to loaded slots */
#define P3HIP_FLAG_SHARED_DEVICE 4u /* several engines keep this GPU busy at once (the self-play host's game groups, the
                                       two players of a match): launches leave out the start-up stagger that only pays
                                       when a launch has the GPU to itself */
#define P3HIP_FLAG_LAUNCH_GRAPH 8u  /* a run over the full static batch replays ONE captured launch graph, as
                                       TrtEngineImpl::RunInference does (trt_engine.cc:260-303); runs over fewer slots
                                       (compaction, cache hits) are launched kernel by kernel.  Same kernels, same results */

/* Creates an engine from a `.p3w` weight file (see p3achygo_amd/netspec.py) for a static
 * batch of `batch_size` slots on HIP device `device_ordinal`.  `version` is the model
 * feature version (engine_factory.cc:37-53; only 1 is supported: 15 planes + 8 scalars).
 * Returns NULL on failure; p3hip_create_error() then holds the reason. */
p3hip_engine* p3hip_create(const char* weights_path, int batch_size, int version,
                           int device_ordinal, uint32_t flags);
const char* p3hip_create_error(void);
void p3hip_destroy(p3hip_engine* e);

int p3hip_kind(const p3hip_engine* e);          /* always P3HIP_KIND_HIP */
const char* p3hip_path(const p3hip_engine* e);  /* path given to p3hip_create */
int p3hip_batch_size(const p3hip_engine* e);

/* LoadBatch: copy the features of one position into pinned staging slot `slot`. */
int p3hip_load_slot(p3hip_engine* e, int slot, const p3hip_features* f);
/* RunInference: H2D of the loaded-and-unfetched slots, one forward pass, D2H of results,
 * stream sync.  Returns 0 on success (also when no slot is pending: nothing is launched). */
int p3hip_run(p3hip_engine* e);
/* GetBatch: copy the results of slot `slot` of the last p3hip_run and mark the slot fetched.
 * Returns 2 if the last run did not evaluate the slot (`out` is left untouched). */
int p3hip_get_slot(p3hip_engine* e, int slot, p3hip_result* out);
/* GetOwnership: tanh ownership map of slot `slot` (the TRT engine leaves this
 * unsupported, trt_engine.cc:353-356; the HIP engine provides it). */
int p3hip_get_ownership(p3hip_engine* e, int slot, float out[P3HIP_NUM_LOCS]);
const char* p3hip_last_error(const p3hip_engine* e);

/* ---- on-device NN cache (extension; the reference caches on the host, above the engine) ----------
 * The reference's NNInterface keeps an LRU cache of NNInferResults per worker thread, keyed by
 * NNKey{color to move, board hash, last moves, komi}; a hit returns the stored result without touching the
 * engine, a miss evaluates under a random symmetry and stores the un-rotated result
 * (cc/nn/nn_interface.cc:93-132, cc/core/lru_cache.h:17-64).  With 288 GB of HBM the table can live beside the
 * engine instead: one table for all workers and games, looked up and filled by the run itself.
 *   p3hip_cache_enable      once after p3hip_create: 2^log2_entries entries of 13.7 KB (key, symmetry, the
 *                           whole result record) in HBM; each key probes 8 consecutive entries, the least
 *                           recently used of them is replaced.
 *   p3hip_load_slot_keyed   LoadBatch with the position's 128-bit key (any digest of the reference's NNKey;
 *                           0/0 = do not cache) and the symmetry (0..7) the features were rotated by.
 *   p3hip_run               evaluates only the keys the table does not hold, serves the rest from HBM, stores
 *                           what it evaluated.  Two slots with the same new key in one run are both evaluated.
 *   p3hip_get_slot_keyed    GetBatch plus the symmetry of the returned result — the stored one on a hit, which
 *                           the caller undoes exactly as the reference's GetBatch(thread, sym) would have when
 *                           the entry was made — and whether it came from the table.
 * Slots loaded with p3hip_load_slot are evaluated and never cached.  Without p3hip_cache_enable the keyed calls
 * behave as the plain ones (every slot is evaluated; the symmetry reported back is the one loaded, from_cache 0).
 * p3hip_cache_stats: lookups, hits, stored entries, table entries. */
int p3hip_cache_enable(p3hip_engine* e, int log2_entries);
int p3hip_load_slot_keyed(p3hip_engine* e, int slot, const p3hip_features* f, uint64_t key_lo, uint64_t key_hi,
                          int symmetry);
int p3hip_get_slot_keyed(p3hip_engine* e, int slot, p3hip_result* out, int* symmetry, int* from_cache);
int p3hip_cache_stats(const p3hip_engine* e, uint64_t out[4]);

/* ---- measurement / test hooks (not part of the reference surface) ------------------ */

/* Device-resident benchmark step: runs the forward pass on whatever is already staged in
 * HBM for `n_positions` slots, no H2D/D2H, no sync.  Used by bench.py's timed region. */
int p3hip_forward_resident(p3hip_engine* e, int n_positions);
/* Upload all currently loaded slots to HBM without running (pairs with the above). */
int p3hip_upload(p3hip_engine* e);
int p3hip_sync(p3hip_engine* e);
/* Raw head outputs of the last run for one slot, for parity tests:
 * out[0..361] pi_logits, [362..723] opt logits, [724..725] outcome logits,
 * [726..1525] score logits, [1526..1886] ownership, [1887] q6_err, [1888] gamma. */
#define P3HIP_RAW_LEN 1889
int p3hip_get_raw(p3hip_engine* e, int slot, float* out);
/* Times the dominant trunk kernel in place: runs `iters` forward passes over the resident
 * batch with a HIP event pair on the engine's stream around every fused-block launch and
 * returns the average milliseconds per launch (<0 on error; also <0 for layer-wise trunks,
 * which have no fused block kernel); writes the
 * algorithmic FLOPs of the convs one launch executes (inner 3x3s + 1x1 reduce/expand,
 * unpadded 361 points). */
double p3hip_time_trunk_kernel(p3hip_engine* e, int n_positions, int iters,
                               double* flops_per_launch, const char** kernel_name);
/* P3HIP_FLAG_LAUNCH_GRAPH: 1 once the full-batch forward pass has been captured and is being replayed,
 * 0 before (or without the flag), -1 when the capture failed and the engine fell back to plain launches. */
int p3hip_graph_state(const p3hip_engine* e);
/* Diagnostics of the hand-scheduled block kernel k_blockw (engines created with P3HIP_BLOCKW_DIAG in the environment
 * run its _diag twin, which stamps s_memtime at its section boundaries): copies up to n 64-bit stamps of the last
 * forward pass, [workgroup 0..7][block 0..15][wave 0..3][24], to out.  Returns 0, or 1 when there are none. */
int p3hip_blockw_stamps(p3hip_engine* e, unsigned long long* out, int n);
/* Debugging aid: the residual stream x after the last forward pass (stopped early by P3HIP_DEBUG_STOP_BLOCK in the
 * environment, if set), n_positions x C x 361 values in the device layout [pos][C / 8][361][8], as floats. */
int p3hip_debug_x(p3hip_engine* e, float* out, int n_positions);
/* Algorithmic FLOPs (2*MAC) of one position: total, and 3x3 trunk convs only. */
void p3hip_flops_per_position(const p3hip_engine* e, double* total, double* conv3x3);

#ifdef __cplusplus
}
#endif
#endif /* P3HIP_H_ */
