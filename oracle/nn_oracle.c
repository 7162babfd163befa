/*
 * nn_oracle.c — CPU fp32 restatement of the reference policy/value network.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP engine: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (p3achygo_amd/) never links or calls it.
 *
 * PARITY PINNING: the reference has no test that pins network numerics (SURVEY.md §0
 * fact 10; python/test/model_v1_test.py checks shapes and sums only), and neither
 * TensorFlow/Keras nor TensorRT are available to run the reference graph.  The oracle is
 * therefore pinned by (1) an independent float64 PyTorch restatement
 * (oracle/torch_restatement.py) whose outputs are committed under tests/golden/, and
 * (2) the reference's feature-plane known answers (cc/nn/__tests__/nn_board_utils_test.cc
 * :84-112 channel map).  Logit parity against the reference's own TF/TRT execution is
 * "parity unpinned".
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * reference checkout).  Plain C99 + OpenMP, direct convolution, NHWC, fp32 accumulate.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/p3hip.h"

#define NLOC 361
#define BL 19
#define BN_EPS 1e-3f /* python/model.py:231 */

enum { BT_BTL = 0, BT_NBT = 1, BT_CLASSIC = 2 };

typedef struct {
  char name[48];
  int ndim;
  int dims[4];
  long long off;
} tensor_ent;

typedef struct oracle_net {
  int version, nblocks, C, Cb, H, V, bint, inner, btype, ntensors;
  tensor_ent* ents;
  float* data;
} oracle_net;

/* ---------------------------------------------------------------- weight file ------ */

oracle_net* oracle_load(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  char magic[4];
  int hdr[10];
  if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "P3W1", 4) != 0 ||
      fread(hdr, 4, 10, f) != 10) {
    fclose(f);
    return NULL;
  }
  oracle_net* n = (oracle_net*)calloc(1, sizeof(oracle_net));
  n->version = hdr[0]; n->nblocks = hdr[1]; n->C = hdr[2]; n->Cb = hdr[3]; n->H = hdr[4];
  n->V = hdr[5]; n->bint = hdr[6]; n->inner = hdr[7]; n->btype = hdr[8]; n->ntensors = hdr[9];
  n->ents = (tensor_ent*)calloc(n->ntensors, sizeof(tensor_ent));
  long long total = 0;
  for (int i = 0; i < n->ntensors; ++i) {
    tensor_ent* e = &n->ents[i];
    if (fread(e->name, 1, 48, f) != 48 || fread(&e->ndim, 4, 1, f) != 1 ||
        fread(e->dims, 4, 4, f) != 4 || fread(&e->off, 8, 1, f) != 1) {
      fclose(f);
      return NULL;
    }
    long long sz = 1;
    for (int d = 0; d < e->ndim; ++d) sz *= e->dims[d];
    if (e->off + sz > total) total = e->off + sz;
  }
  long pos = ftell(f);
  pos += (64 - pos % 64) % 64;
  fseek(f, pos, SEEK_SET);
  n->data = (float*)malloc(sizeof(float) * total);
  if (fread(n->data, 4, total, f) != (size_t)total) {
    fclose(f);
    return NULL;
  }
  fclose(f);
  return n;
}

void oracle_free(oracle_net* n) {
  if (!n) return;
  free(n->ents);
  free(n->data);
  free(n);
}

void oracle_config(const oracle_net* n, int out[9]) {
  out[0] = n->version; out[1] = n->nblocks; out[2] = n->C; out[3] = n->Cb; out[4] = n->H;
  out[5] = n->V; out[6] = n->bint; out[7] = n->inner; out[8] = n->btype;
}

static const float* T(const oracle_net* n, const char* name) {
  for (int i = 0; i < n->ntensors; ++i)
    if (strcmp(n->ents[i].name, name) == 0) return n->data + n->ents[i].off;
  fprintf(stderr, "oracle: missing tensor %s\n", name);
  abort();
}

static const float* TB(const oracle_net* n, int blk, const char* suffix) {
  char buf[64];
  snprintf(buf, sizeof buf, "blocks.%d.%s", blk, suffix);
  return T(n, buf);
}

/* ------------------------------------------------------------- input features ------ */

/* Restates nn::FillPlanePair (cc/nn/engine/buf_utils.h:60-77): NHWC one-hot of a
 * {-1,0,+1} grid into an (our, opp) channel pair, from the side-to-move's view. */
static void fill_plane_pair(float* planes, int our_ch, int opp_ch, const int8_t* grid,
                            int8_t color) {
  for (int i = 0; i < NLOC; ++i) {
    int8_t c = grid[i];
    if (c == color)
      planes[i * P3HIP_NUM_PLANES + our_ch] = 1.0f;
    else if (c == (int8_t)-color)
      planes[i * P3HIP_NUM_PLANES + opp_ch] = 1.0f;
  }
}

/* Restates nn::LoadPlanes / nn::LoadFeatures for version 1
 * (cc/nn/engine/go_features.cc:10-61) plus the zero-fill TrtEngineImpl::LoadBatch does
 * first (cc/nn/engine/trt_engine.cc:222-236).  planes: [361][15], feats: [8]. */
void oracle_fill_inputs(const p3hip_features* f, float* planes, float* feats) {
  memset(planes, 0, sizeof(float) * NLOC * P3HIP_NUM_PLANES);
  memset(feats, 0, sizeof(float) * P3HIP_NUM_SCALARS);
  fill_plane_pair(planes, 0, 1, f->board, f->color);
  fill_plane_pair(planes, 7, 8, f->stones_atari, f->color);
  fill_plane_pair(planes, 9, 10, f->stones_two_liberties, f->color);
  fill_plane_pair(planes, 11, 12, f->stones_three_liberties, f->color);
  fill_plane_pair(planes, 13, 14, f->stones_laddered, f->color);
  for (int i = 0; i < P3HIP_NUM_LAST_MOVES; ++i) {
    p3hip_loc m = f->last_moves[i];
    int is_noop = (m.i == -1 && m.j == -1), is_pass = (m.i == 19 && m.j == 0);
    if (is_noop || is_pass) continue;
    planes[(m.i * BL + m.j) * P3HIP_NUM_PLANES + (i + 2)] = 1.0f;
  }
  feats[f->color == 1 ? 0 : 1] = 1.0f;
  for (int i = 0; i < P3HIP_NUM_LAST_MOVES; ++i) {
    p3hip_loc m = f->last_moves[i];
    if (m.i == 19 && m.j == 0) feats[i + 2] = 1.0f;
  }
  feats[7] = (f->color == 1 ? -1.0f : 1.0f) * f->komi / 15.0f;
}

/* -------------------------------------------------------------------- math --------- */

static inline float softplusf(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
/* keras.activations.mish: x * tanh(softplus(x)) */
static inline float mishf(float x) { return x * tanhf(softplusf(x)); }
static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

static void softmax(const float* in, float* out, int n) {
  float m = in[0];
  for (int i = 1; i < n; ++i) m = in[i] > m ? in[i] : m;
  double s = 0;
  for (int i = 0; i < n; ++i) {
    out[i] = expf(in[i] - m);
    s += out[i];
  }
  float inv = (float)(1.0 / s);
  for (int i = 0; i < n; ++i) out[i] *= inv;
}

/* SAME-padded KxK convolution, NHWC, kernel HWIO, no bias (model.py:101-117 make_conv). */
static void conv2d(const float* in, int cin, const float* w, int k, int cout, float* out) {
  int r = k / 2;
  for (int y = 0; y < BL; ++y)
    for (int x = 0; x < BL; ++x) {
      float* o = out + (y * BL + x) * cout;
      for (int c = 0; c < cout; ++c) o[c] = 0.0f;
      for (int ky = 0; ky < k; ++ky) {
        int yy = y + ky - r;
        if (yy < 0 || yy >= BL) continue;
        for (int kx = 0; kx < k; ++kx) {
          int xx = x + kx - r;
          if (xx < 0 || xx >= BL) continue;
          const float* ip = in + (yy * BL + xx) * cin;
          const float* wp = w + (size_t)(ky * k + kx) * cin * cout;
          for (int ci = 0; ci < cin; ++ci) {
            float a = ip[ci];
            if (a == 0.0f) continue;
            const float* wr = wp + (size_t)ci * cout;
            for (int c = 0; c < cout; ++c) o[c] += a * wr[c];
          }
        }
      }
    }
}

/* BatchNormalization (inference) followed by mish: the ConvPreActivation prologue
 * (model.py:276-282; BN form y = gamma*(x-mean)/sqrt(var+eps)+beta). */
static void bn_mish(const float* in, int c, const float* g, const float* b, const float* m,
                    const float* v, float* out) {
  for (int i = 0; i < NLOC; ++i)
    for (int ch = 0; ch < c; ++ch) {
      float s = g[ch] / sqrtf(v[ch] + BN_EPS);
      float y = (in[i * c + ch] - m[ch]) * s + b[ch];
      out[i * c + ch] = mishf(y);
    }
}

/* ConvPreActivation.call (model.py:276-282): conv(mish(bn(x))). */
static void preact_conv(const oracle_net* n, int blk, int idx, const float* in, int cin, int k,
                        int cout, float* tmp, float* out) {
  char s[32];
  const float *g, *b, *m, *v, *w;
  snprintf(s, sizeof s, "bn%d.gamma", idx); g = TB(n, blk, s);
  snprintf(s, sizeof s, "bn%d.beta", idx); b = TB(n, blk, s);
  snprintf(s, sizeof s, "bn%d.mean", idx); m = TB(n, blk, s);
  snprintf(s, sizeof s, "bn%d.var", idx); v = TB(n, blk, s);
  snprintf(s, sizeof s, "conv%d.w", idx); w = TB(n, blk, s);
  bn_mish(in, cin, g, b, m, v, tmp);
  conv2d(tmp, cin, w, k, cout, out);
}

/* GlobalPool.call (model.py:641-645): concat(mean over HW, max over HW). */
static void gpool(const float* in, int c, float* out) {
  for (int ch = 0; ch < c; ++ch) {
    double s = 0;
    float mx = in[ch];
    for (int i = 0; i < NLOC; ++i) {
      float x = in[i * c + ch];
      s += x;
      mx = x > mx ? x : mx;
    }
    out[ch] = (float)(s / NLOC);
    out[c + ch] = mx;
  }
}

static void dense(const float* in, int cin, const float* w, const float* b, int cout,
                  float* out) {
  for (int o = 0; o < cout; ++o) {
    double s = b ? b[o] : 0.0;
    for (int i = 0; i < cin; ++i) s += (double)in[i] * w[i * cout + o];
    out[o] = (float)s;
  }
}

/* -------------------------------------------------------------- forward pass ------- */

typedef struct {
  float *x, *t0, *t1, *t2, *tmp;
} scratch;

static int is_broadcast(const oracle_net* n, int i) { /* model.py:1002 */
  return i % n->bint == n->bint - 1;
}

/* One trunk block; restates ResidualBlock.call (model.py:315-321) over the layer lists of
 * BottleneckResidualConvBlock (:372-425), NbtResidualBlock (:430-486),
 * ClassicResidualBlock (:329-368) and BroadcastResidualBlock (:490-631). */
static void run_block(const oracle_net* n, int i, scratch* s) {
  int C = n->C, Cb = n->Cb;
  float* x = s->x;
  if (is_broadcast(n, i)) {
    preact_conv(n, i, 0, x, C, 1, C, s->tmp, s->t0);
    /* BroadcastPreAct.call (model.py:556-567): NHWC->NCHW, mish, Dense(361) over the
     * flattened board of each channel (weights shared by all channels), back to NHWC. */
    const float* dw = TB(n, i, "dense.w");
    const float* db = TB(n, i, "dense.b");
    for (int k = 0; k < NLOC * C; ++k) s->tmp[k] = mishf(s->t0[k]);
    for (int c = 0; c < C; ++c)
      for (int j = 0; j < NLOC; ++j) {
        double a = db[j];
        for (int p = 0; p < NLOC; ++p) a += (double)s->tmp[p * C + c] * dw[p * NLOC + j];
        s->t1[j * C + c] = (float)a;
      }
    preact_conv(n, i, 1, s->t1, C, 1, C, s->tmp, s->t0);
    for (int k = 0; k < NLOC * C; ++k) x[k] += s->t0[k];
  } else if (n->btype == BT_BTL) {
    int L = n->inner;
    preact_conv(n, i, 0, x, C, 1, Cb, s->tmp, s->t0);
    float *a = s->t0, *b = s->t1;
    for (int j = 1; j <= L; ++j) {
      preact_conv(n, i, j, a, Cb, 3, Cb, s->tmp, b);
      float* sw = a; a = b; b = sw;
    }
    preact_conv(n, i, L + 1, a, Cb, 1, C, s->tmp, s->t2);
    for (int k = 0; k < NLOC * C; ++k) x[k] += s->t2[k];
  } else if (n->btype == BT_NBT) {
    preact_conv(n, i, 0, x, C, 1, Cb, s->tmp, s->t0);
    for (int r = 0; r < 2; ++r) { /* nbt_res0, nbt_res1: ClassicResidualBlock of 2 convs */
      preact_conv(n, i, 1 + 2 * r, s->t0, Cb, 3, Cb, s->tmp, s->t1);
      preact_conv(n, i, 2 + 2 * r, s->t1, Cb, 3, Cb, s->tmp, s->t2);
      for (int k = 0; k < NLOC * Cb; ++k) s->t0[k] += s->t2[k];
    }
    preact_conv(n, i, 5, s->t0, Cb, 1, C, s->tmp, s->t2);
    for (int k = 0; k < NLOC * C; ++k) x[k] += s->t2[k];
  } else { /* classic */
    preact_conv(n, i, 0, x, C, 3, C, s->tmp, s->t0);
    preact_conv(n, i, 1, s->t0, C, 3, C, s->tmp, s->t1);
    for (int k = 0; k < NLOC * C; ++k) x[k] += s->t1[k];
  }
}

/* Restates P3achyGoModel.call (model.py:1222-1295), PolicyHead.call (:783-812),
 * GlobalPoolBias.call (:696-706) and ValueHead.call (:887-979) for one position.
 * raw layout: see P3HIP_RAW_LEN in include/p3hip.h. */
static void forward_one(const oracle_net* n, const float* planes, const float* feats,
                        p3hip_result* res, float* raw, float* trunk_out) {
  int C = n->C, H = n->H, V = n->V;
  size_t big = (size_t)NLOC * (C > 64 ? C : 64);
  scratch s;
  s.x = (float*)malloc(sizeof(float) * big);
  s.t0 = (float*)malloc(sizeof(float) * big);
  s.t1 = (float*)malloc(sizeof(float) * big);
  s.t2 = (float*)malloc(sizeof(float) * big);
  s.tmp = (float*)malloc(sizeof(float) * big);

  /* init conv + game-state dense broadcast-add (model.py:1230-1237) */
  conv2d(planes, P3HIP_NUM_PLANES, T(n, "init_conv.w"), 5, C, s.x);
  float* gs = (float*)malloc(sizeof(float) * C);
  dense(feats, P3HIP_NUM_SCALARS, T(n, "init_game.w"), T(n, "init_game.b"), C, gs);
  for (int i = 0; i < NLOC; ++i)
    for (int c = 0; c < C; ++c) s.x[i * C + c] += gs[c];
  free(gs);

  for (int i = 0; i < n->nblocks; ++i) run_block(n, i, &s);
  if (trunk_out) memcpy(trunk_out, s.x, sizeof(float) * NLOC * C);

  /* ---- policy head ---- */
  float* p = s.t0;
  float* g = s.t1;
  conv2d(s.x, C, T(n, "policy.conv_p.w"), 1, H, p);
  conv2d(s.x, C, T(n, "policy.conv_g.w"), 1, H, g);
  bn_mish(g, H, T(n, "policy.gpool_bn.gamma"), T(n, "policy.gpool_bn.beta"),
          T(n, "policy.gpool_bn.mean"), T(n, "policy.gpool_bn.var"), g);
  float gp[256], gb[128];
  gpool(g, H, gp);
  dense(gp, 2 * H, T(n, "policy.gpool_dense.w"), T(n, "policy.gpool_dense.b"), H, gb);
  for (int i = 0; i < NLOC; ++i)
    for (int c = 0; c < H; ++c) p[i * H + c] = mishf(p[i * H + c] + gb[c]);
  float pi_logits[362], opt_logits[362];
  {
    const float* wm = T(n, "policy.out_moves.w"); /* [1,1,H,2] */
    const float* wo = T(n, "policy.opt_moves.w"); /* [1,1,H,1] */
    for (int i = 0; i < NLOC; ++i) {
      double a = 0, o = 0;
      for (int c = 0; c < H; ++c) {
        a += (double)p[i * H + c] * wm[c * 2 + 0];
        o += (double)p[i * H + c] * wo[c];
      }
      pi_logits[i] = (float)a;
      opt_logits[i] = (float)o;
    }
    float pass2[2], pass1[1];
    dense(gp, 2 * H, T(n, "policy.out_pass.w"), T(n, "policy.out_pass.b"), 2, pass2);
    pi_logits[361] = pass2[0] - 3.0f; /* model.py:796 "- 3" */
    dense(gp, 2 * H, T(n, "policy.opt_pass.w"), T(n, "policy.opt_pass.b"), 1, pass1);
    opt_logits[361] = pass1[0] - 3.0f;
  }

  /* ---- value head ---- */
  float* v = s.t1;
  conv2d(s.x, C, T(n, "value.conv.w"), 1, H, v);
  float vp[256];
  gpool(v, H, vp);
  float emb[256], go[14];
  dense(vp, 2 * H, T(n, "value.oq_embed.w"), T(n, "value.oq_embed.b"), V, emb);
  for (int c = 0; c < V; ++c) emb[c] = mishf(emb[c]);
  dense(emb, V, T(n, "value.oq_out.w"), T(n, "value.oq_out.b"), 14, go);
  float own[361];
  {
    const float* wo = T(n, "value.own.w");
    for (int i = 0; i < NLOC; ++i) {
      double a = 0;
      for (int c = 0; c < H; ++c) a += (double)v[i * H + c] * wo[c];
      own[i] = tanhf((float)a);
    }
  }
  float gpre[256], gamma;
  dense(vp, 2 * H, T(n, "value.gamma_pre.w"), T(n, "value.gamma_pre.b"), V, gpre);
  for (int c = 0; c < V; ++c) gpre[c] = mishf(gpre[c]);
  dense(gpre, V, T(n, "value.gamma_out.w"), T(n, "value.gamma_out.b"), 1, &gamma);
  float score_logits[800];
  {
    const float* wp = T(n, "value.score_pre.w"); /* [2H+1, V] */
    const float* bp = T(n, "value.score_pre.b");
    const float* wo = T(n, "value.score_out.w"); /* [V,1] */
    const float* bo = T(n, "value.score_out.b");
    float base[256];
    dense(vp, 2 * H, wp, bp, V, base); /* rows 0..2H-1 of score_pre.w */
    float scale = softplusf(gamma);
    scale = scale < 10.0f ? scale : 10.0f;
    for (int sidx = 0; sidx < 800; ++sidx) {
      float sc = 0.05f * (float)(sidx - 400) + 0.025f; /* model.py:1223-1228 */
      double a = bo[0];
      for (int c = 0; c < V; ++c) a += (double)mishf(base[c] + sc * wp[(2 * H) * V + c]) * wo[c];
      score_logits[sidx] = scale * (float)a;
    }
  }

  if (raw) {
    memcpy(raw, pi_logits, sizeof pi_logits);
    memcpy(raw + 362, opt_logits, sizeof opt_logits);
    raw[724] = go[0];
    raw[725] = go[1];
    memcpy(raw + 726, score_logits, sizeof score_logits);
    memcpy(raw + 1526, own, sizeof own);
    raw[1887] = 4.0f * sigmoidf(go[5]); /* model.py:950 q6_err */
    raw[1888] = gamma;
  }
  if (res) {
    /* Output mapping of the live engine: trt_names.h:15-21, trt_engine.cc:306-351. */
    memcpy(res->move_logits, pi_logits, sizeof pi_logits);
    softmax(pi_logits, res->move_probs, 362);
    softmax(go, res->value_probs, 2);
    softmax(score_logits, res->score_probs, 800);
    softmax(opt_logits, res->opt_move_probs, 362); /* host softmax, trt_engine.cc:347-348 */
    res->err2_outcome = 4.0f * sigmoidf(go[5]);
  }
  free(s.x); free(s.t0); free(s.t1); free(s.t2); free(s.tmp);
}

/* Batch forward from NHWC planes / scalars.  res, raw, trunk may each be NULL. */
void oracle_forward(const oracle_net* n, int count, const float* planes, const float* feats,
                    p3hip_result* res, float* raw, float* trunk, int nthreads) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
  for (int i = 0; i < count; ++i)
    forward_one(n, planes + (size_t)i * NLOC * P3HIP_NUM_PLANES, feats + i * P3HIP_NUM_SCALARS,
                res ? res + i : NULL, raw ? raw + (size_t)i * P3HIP_RAW_LEN : NULL,
                trunk ? trunk + (size_t)i * NLOC * n->C : NULL);
}

/* Batch forward from GoFeatures PODs: LoadBatch + RunInference + GetBatch of the
 * reference engine contract (cc/nn/engine/engine.h:32-39). */
void oracle_forward_features(const oracle_net* n, int count, const p3hip_features* f,
                             p3hip_result* res, float* raw, int nthreads) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
  for (int i = 0; i < count; ++i) {
    float* planes = (float*)malloc(sizeof(float) * NLOC * P3HIP_NUM_PLANES);
    float feats[P3HIP_NUM_SCALARS];
    oracle_fill_inputs(&f[i], planes, feats);
    forward_one(n, planes, feats, res ? res + i : NULL,
                raw ? raw + (size_t)i * P3HIP_RAW_LEN : NULL, NULL);
    free(planes);
  }
}
