/*
 * cpu_engine.c — the CPU fp32 oracle behind the engine's C ABI (include/p3hip.h).
 *
 * TEST / BASELINE INFRASTRUCTURE ONLY.  SURVEY.md section 8(d) asks for the reference's CPU
 * path timed beside the GPU engine: "the build's CPU fp32 Engine behind the same NNInterface
 * and the same self-play host code".  The reference's own TF-CPU engine is stale and
 * unbuildable, so the baseline engine is this wrapper: nn_oracle.c's
 * forward pass (direct fp32 convolution, OpenMP over positions) exported under the very entry
 * points the self-play host binds with dlopen (p3achygo_amd/host/evaluator.h).  bench.py's
 * cpu_baseline leg hands the host THIS library instead of libp3hip.so; nothing under
 * p3achygo_amd/ links, loads or names it.
 *
 * Semantics follow the reference's engines: every slot of the static batch that was loaded
 * since the previous run is evaluated (no compaction needed on a CPU), results stay valid
 * until the next run.  nn::Engine, cc/nn/engine/engine.h:22-43.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/p3hip.h"

typedef struct oracle_net oracle_net;
oracle_net* oracle_load(const char* path);
void oracle_free(oracle_net* n);
void oracle_forward_features(const oracle_net* n, int count, const p3hip_features* f,
                             p3hip_result* res, float* raw, int nthreads);

#define RAW_LEN 1889
#define OFF_OWN 1526

struct p3hip_engine {
  oracle_net* net;
  char path[1024];
  char err[256];
  int batch, nthreads;
  p3hip_features* feats;    /* [batch] as loaded */
  unsigned char* loaded;    /* [batch] */
  int* row;                 /* slot -> row of the last run, -1 if absent */
  p3hip_features* dense;    /* [batch] compacted inputs of the run */
  p3hip_result* res;        /* [batch] rows */
  float* raw;               /* [batch][RAW_LEN] rows */
};

static char g_create_error[256];

const char* p3hip_create_error(void) { return g_create_error; }

p3hip_engine* p3hip_create(const char* weights_path, int batch_size, int version, int device_ordinal,
                           uint32_t flags) {
  (void)device_ordinal;
  (void)flags;
  g_create_error[0] = 0;
  if (version != 1 || batch_size < 1) {
    snprintf(g_create_error, sizeof g_create_error, "cpu engine: bad version or batch size");
    return NULL;
  }
  p3hip_engine* e = (p3hip_engine*)calloc(1, sizeof *e);
  e->net = oracle_load(weights_path);
  if (!e->net) {
    snprintf(g_create_error, sizeof g_create_error, "cpu engine: cannot load %s", weights_path);
    free(e);
    return NULL;
  }
  snprintf(e->path, sizeof e->path, "%s", weights_path);
  e->batch = batch_size;
  const char* nt = getenv("P3CPU_THREADS");   /* cores given to the forward pass */
  e->nthreads = nt ? atoi(nt) : 1;
  if (e->nthreads < 1) e->nthreads = 1;
  e->feats = (p3hip_features*)calloc(batch_size, sizeof *e->feats);
  e->dense = (p3hip_features*)calloc(batch_size, sizeof *e->dense);
  e->loaded = (unsigned char*)calloc(batch_size, 1);
  e->row = (int*)malloc(sizeof(int) * batch_size);
  e->res = (p3hip_result*)calloc(batch_size, sizeof *e->res);
  e->raw = (float*)calloc((size_t)batch_size * RAW_LEN, sizeof(float));
  for (int i = 0; i < batch_size; ++i) e->row[i] = -1;
  return e;
}

void p3hip_destroy(p3hip_engine* e) {
  if (!e) return;
  oracle_free(e->net);
  free(e->feats); free(e->dense); free(e->loaded); free(e->row); free(e->res); free(e->raw);
  free(e);
}

int p3hip_kind(const p3hip_engine* e) { (void)e; return P3HIP_KIND_HIP; }
const char* p3hip_path(const p3hip_engine* e) { return e->path; }
int p3hip_batch_size(const p3hip_engine* e) { return e->batch; }
const char* p3hip_last_error(const p3hip_engine* e) { return e->err; }

int p3hip_load_slot(p3hip_engine* e, int slot, const p3hip_features* f) {
  if (slot < 0 || slot >= e->batch) return 1;
  memcpy(&e->feats[slot], f, sizeof *f);
  __atomic_store_n(&e->loaded[slot], 1, __ATOMIC_RELEASE);
  return 0;
}

int p3hip_run(p3hip_engine* e) {
  int n = 0;
  for (int s = 0; s < e->batch; ++s) {
    if (__atomic_exchange_n(&e->loaded[s], 0, __ATOMIC_ACQUIRE)) {
      e->dense[n] = e->feats[s];
      e->row[s] = n++;
    } else {
      e->row[s] = -1;
    }
  }
  if (n) oracle_forward_features(e->net, n, e->dense, e->res, e->raw, e->nthreads);
  return 0;
}

int p3hip_get_slot(p3hip_engine* e, int slot, p3hip_result* out) {
  if (slot < 0 || slot >= e->batch) return 1;
  if (e->row[slot] < 0) return 2;
  memcpy(out, &e->res[e->row[slot]], sizeof *out);
  return 0;
}

int p3hip_get_ownership(p3hip_engine* e, int slot, float out[P3HIP_NUM_LOCS]) {
  if (slot < 0 || slot >= e->batch) return 1;
  if (e->row[slot] < 0) return 2;
  memcpy(out, e->raw + (size_t)e->row[slot] * RAW_LEN + OFF_OWN, sizeof(float) * P3HIP_NUM_LOCS);
  return 0;
}
