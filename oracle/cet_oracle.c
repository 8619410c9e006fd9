/*
 * cet_oracle.c -- CPU restatement of the reference's inference arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY. This file is the checker the HIP kernels are compared against
 * (tests/, the smoke test, and the cpu_baseline leg of bench.py). Nothing in the shipped
 * package links, loads or calls it.
 *
 * Each function cites the reference lines it restates. Parity status:
 *   - Transformer: pinned by the reference's stored input/output pairs
 *     (ExampleCodeTsinghua/data/CR_data_30.mat -> CR_transformer_30.mat and the _36/_60 pairs)
 *     and by vectors captured from the imported reference (tests/golden/make_golden.py).
 *   - GRU / LSTM / RNN: pinned by vectors captured from the imported reference.
 *   - LMMSE: PARITY UNPINNED. The arithmetic lives in Sionna (sionna.ofdm.LMMSEEqualizer, 0.x
 *     API line, no version pinned by the reference, not vendored, not installable here) and
 *     the reference holds no vector for it. The closed form below follows Sionna's published
 *     definition; it is cross-checked only against numpy float64 in tests/.
 *
 * Plain C99 + libm, float32 arithmetic where the reference computes in float32.
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/cet.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------ */
/* small dense helpers                                                                   */

/* y[r][o] = b[o] + sum_k x[r][k] * W[o][k]   (torch.nn.Linear; also the 1x1 Conv1d of
 * encoder.py:30-31, 50-51 and decoder.py:12-13, 33-34) */
static void linear(const float* x, int rows, int K, const float* W, const float* b, int N,
                   float* y) {
  for (int r = 0; r < rows; ++r)
    for (int o = 0; o < N; ++o) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += x[r * K + k] * W[o * K + k];
      y[r * N + o] = acc + (b ? b[o] : 0.f);
    }
}

/* torch.nn.LayerNorm over the last dim, eps 1e-5, affine */
static void layer_norm(float* x, int rows, int D, const float* g, const float* b) {
  for (int r = 0; r < rows; ++r) {
    float* p = x + r * D;
    float mean = 0.f;
    for (int i = 0; i < D; ++i) mean += p[i];
    mean /= (float)D;
    float var = 0.f;
    for (int i = 0; i < D; ++i) var += (p[i] - mean) * (p[i] - mean);
    var /= (float)D;
    float inv = 1.0f / sqrtf(var + 1e-5f);
    for (int i = 0; i < D; ++i) p[i] = (p[i] - mean) * inv * g[i] + b[i];
  }
}

static float act_fn(float v, int activation) {
  if (activation == 0) return v > 0.f ? v : 0.f;                /* F.relu */
  return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); /* F.gelu (erf form) */
}

/* Conv1d(k=3, padding_mode='circular', padding=pad) along time on a [L, Cin] sequence.
 * out length = L + 2*pad - 2. W is [Cout][Cin][3] (PyTorch layout).
 * embed.py:29-38 (pad 1), encoder.py:8-12,18 (pad 2). */
static void conv1d_circ(const float* x, int L, int Cin, const float* W, const float* b,
                        int Cout, int pad, float* y) {
  int Lout = L + 2 * pad - 2;
  for (int t = 0; t < Lout; ++t)
    for (int o = 0; o < Cout; ++o) {
      float acc = 0.f;
      for (int c = 0; c < Cin; ++c)
        for (int j = 0; j < 3; ++j) {
          int src = ((t + j - pad) % L + L) % L;
          acc += x[src * Cin + c] * W[(o * Cin + c) * 3 + j];
        }
      y[t * Cout + o] = acc + b[o];
    }
}

/* ------------------------------------------------------------------------------------ */
/* Transformer predictor                                                                 */

typedef struct {
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;
} attn_w;

static const float* take(const float** p, size_t n) {
  const float* r = *p;
  *p += n;
  return r;
}
/* AttentionLayer (attn.py:121-138): d_keys = d_values = d_model // n_heads; the q/k/v
 * projections are Linear(d_model, d_keys * n_heads), the output projection
 * Linear(d_values * n_heads, d_model) -- narrower than d_model when it is not a multiple of
 * n_heads. */
static void take_attn(const float** p, int D, int H, attn_w* a) {
  int HE = (D / H) * H;
  a->wq = take(p, (size_t)HE * D); a->bq = take(p, HE);
  a->wk = take(p, (size_t)HE * D); a->bk = take(p, HE);
  a->wv = take(p, (size_t)HE * D); a->bv = take(p, HE);
  a->wo = take(p, (size_t)D * HE); a->bo = take(p, D);
}

/* AttentionLayer.forward + FullAttention.forward (attn.py:140-156, 18-36).
 * q_in [L,D], kv_in [S,D] -> out [L,D]; amap (optional) [H,L,S]. */
static void attention(const float* q_in, int L, const float* kv_in, int S, int D, int H,
                      const attn_w* w, int causal, float* out, float* amap) {
  int E = D / H, HE = E * H;
  float* q = (float*)malloc(sizeof(float) * L * HE);
  float* k = (float*)malloc(sizeof(float) * S * HE);
  float* v = (float*)malloc(sizeof(float) * S * HE);
  float* ctx = (float*)malloc(sizeof(float) * L * HE);
  float* sc = (float*)malloc(sizeof(float) * S);
  linear(q_in, L, D, w->wq, w->bq, HE, q);
  linear(kv_in, S, D, w->wk, w->bk, HE, k);
  linear(kv_in, S, D, w->wv, w->bv, HE, v);
  float scale = 1.0f / sqrtf((float)E);
  for (int h = 0; h < H; ++h)
    for (int l = 0; l < L; ++l) {
      float mx = -INFINITY;
      for (int s = 0; s < S; ++s) {
        float acc = 0.f;
        for (int e = 0; e < E; ++e) acc += q[l * HE + h * E + e] * k[s * HE + h * E + e];
        /* masked_fill_(triu(1), -inf) happens before the scaling (attn.py:24-30) */
        if (causal && s > l) acc = -INFINITY;
        sc[s] = scale * acc;
        if (sc[s] > mx) mx = sc[s];
      }
      float sum = 0.f;
      for (int s = 0; s < S; ++s) { sc[s] = expf(sc[s] - mx); sum += sc[s]; }
      for (int s = 0; s < S; ++s) {
        sc[s] /= sum;
        if (amap) amap[((size_t)h * L + l) * S + s] = sc[s];
      }
      for (int e = 0; e < E; ++e) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += sc[s] * v[s * HE + h * E + e];
        ctx[l * HE + h * E + e] = acc;
      }
    }
  linear(ctx, L, HE, w->wo, w->bo, D, out);
  free(q); free(k); free(v); free(ctx); free(sc);
}

/* Debug taps: each per-sample pass records (rows, data) per stage; the caller scatters
 * them into the stage-major [stage][B][rows][D] buffer. */
typedef struct { int rows; float* data; } tap_rec;
typedef struct { tap_rec* recs; int n, cap; int enabled; int D; } tap_list;

static void tap(tap_list* tl, const float* x, int rows) {
  if (!tl->enabled) return;
  if (tl->n == tl->cap) {
    tl->cap = tl->cap ? tl->cap * 2 : 64;
    tl->recs = (tap_rec*)realloc(tl->recs, sizeof(tap_rec) * tl->cap);
  }
  tl->recs[tl->n].rows = rows;
  tl->recs[tl->n].data = (float*)malloc(sizeof(float) * rows * tl->D);
  memcpy(tl->recs[tl->n].data, x, sizeof(float) * rows * tl->D);
  tl->n++;
}

/* EncoderLayer.forward (encoder.py:37-53) in place on x [L,D]. */
static void encoder_layer(float* x, int L, const cet_informer_cfg* c, const float** p,
                          float* amap) {
  int D = c->d_model, F = c->d_ff;
  attn_w aw; take_attn(p, D, c->n_heads, &aw);
  const float* w1 = take(p, (size_t)F * D); const float* b1 = take(p, F);
  const float* w2 = take(p, (size_t)D * F); const float* b2 = take(p, D);
  const float* g1 = take(p, D); const float* be1 = take(p, D);
  const float* g2 = take(p, D); const float* be2 = take(p, D);
  float* a = (float*)malloc(sizeof(float) * L * D);
  float* h = (float*)malloc(sizeof(float) * L * F);
  attention(x, L, x, L, D, c->n_heads, &aw, 0, a, amap);
  for (int i = 0; i < L * D; ++i) x[i] += a[i];
  layer_norm(x, L, D, g1, be1);
  linear(x, L, D, w1, b1, F, h);
  for (int i = 0; i < L * F; ++i) h[i] = act_fn(h[i], c->activation);
  linear(h, L, F, w2, b2, D, a);
  for (int i = 0; i < L * D; ++i) x[i] += a[i];
  layer_norm(x, L, D, g2, be2);
  free(a); free(h);
}

/* ConvLayer.forward (encoder.py:17-23): circular conv (pad 2) -> BatchNorm1d in eval mode
 * -> ELU -> MaxPool1d(3, stride 2, padding 1). x [L,D] -> returns new length, writes y. */
static int distil_layer(const float* x, int L, int D, const float** p, float* y) {
  const float* w = take(p, (size_t)D * D * 3); const float* b = take(p, D);
  const float* g = take(p, D); const float* be = take(p, D);
  const float* rm = take(p, D); const float* rv = take(p, D);
  int Lc = L + 2;
  float* t = (float*)malloc(sizeof(float) * Lc * D);
  conv1d_circ(x, L, D, w, b, D, 2, t);
  for (int o = 0; o < D; ++o) {
    /* ATen's eval-mode batch norm: alpha = weight/sqrt(var+eps), beta = bias - mean*alpha */
    float alpha = g[o] * (1.0f / sqrtf(rv[o] + 1e-5f));
    float beta = be[o] - rm[o] * alpha;
    for (int r = 0; r < Lc; ++r) {
      float v = t[r * D + o] * alpha + beta;
      t[r * D + o] = v > 0.f ? v : (expf(v) - 1.0f); /* nn.ELU(alpha=1) */
    }
  }
  int Lo = (Lc + 2 - 3) / 2 + 1;
  for (int u = 0; u < Lo; ++u)
    for (int o = 0; o < D; ++o) {
      float m = -INFINITY;
      for (int j = -1; j <= 1; ++j) {
        int r = 2 * u + j;
        if (r >= 0 && r < Lc && t[r * D + o] > m) m = t[r * D + o];
      }
      y[u * D + o] = m;
    }
  free(t);
  return Lo;
}

/* Encoder.forward (encoder.py:62-80) on x [L,D] in place (buffer must hold L rows);
 * returns the output length. amaps advances by H*L*L per attention layer. */
static int encoder_fwd(float* x, int L, int n_layers, const cet_informer_cfg* c,
                       const float** p, float** amaps, tap_list* tl) {
  int D = c->d_model, H = c->n_heads;
  int distil = c->distil;
  /* weights are stored: all attention layers, then all conv layers, then the norm */
  const float* pa = *p;
  size_t HE = (size_t)(D / H) * H;
  size_t per_layer = 3 * (HE * D + HE) + ((size_t)D * HE + D) + (size_t)c->d_ff * D + c->d_ff +
                     (size_t)D * c->d_ff + D + 4 * (size_t)D;
  const float* pc = pa + per_layer * n_layers;
  float* y = (float*)malloc(sizeof(float) * (L + 2) * D);
  for (int l = 0; l < n_layers; ++l) {
    float* am = (amaps && *amaps) ? *amaps : NULL;
    encoder_layer(x, L, c, &pa, am);
    if (am) *amaps += (size_t)H * L * L;
    tap(tl, x, L);
    if (distil && l < n_layers - 1) {
      L = distil_layer(x, L, D, &pc, y);
      memcpy(x, y, sizeof(float) * L * D);
      tap(tl, x, L);
    }
  }
  const float* g = take(&pc, D); const float* b = take(&pc, D);
  layer_norm(x, L, D, g, b);
  tap(tl, x, L);
  *p = pc;
  free(y);
  return L;
}

/* DataEmbedding.forward (embed.py:50-53): circular token conv + fixed positional table. */
static void embedding(const float* x, int L, int Cin, int D, const float** p, float* y) {
  const float* w = take(p, (size_t)D * Cin * 3); const float* b = take(p, D);
  const float* pe = take(p, (size_t)L * D);
  conv1d_circ(x, L, Cin, w, b, D, 1, y);
  for (int i = 0; i < L * D; ++i) y[i] += pe[i];
}

/* lengths of the attention maps, in execution order; returns count */
static int attn_lengths(const cet_informer_cfg* c, int* lens, int cap) {
  int n = 0;
  int inp_len = c->seq_len, cur = c->seq_len;
  int el_start = c->e_layers, el_end = c->stack ? 3 : c->e_layers;
  for (int el = el_start; el >= el_end; --el) {
    int L = inp_len < cur ? inp_len : cur;
    for (int l = 0; l < el; ++l) {
      if (n < cap) lens[n] = L;
      n++;
      if (c->distil && l < el - 1) L = (L + 1) / 2 + 1;
    }
    cur = L;
    inp_len /= 2;
  }
  return n;
}

ORC_API size_t orc_informer_attn_size(const cet_informer_cfg* c, int B) {
  int lens[64];
  int n = attn_lengths(c, lens, 64);
  size_t t = 0;
  for (int i = 0; i < n; ++i) t += (size_t)c->n_heads * lens[i] * lens[i];
  return t * (size_t)B;
}

/* InformerStack.forward / Informer.forward (model.py:220-234, 67-81), EncoderStack.forward
 * (encoder.py:87-101, including its re-use of the previous encoder's OUTPUT as the next
 * encoder's input), Decoder.forward / DecoderLayer.forward (decoder.py:44-51, 20-36).
 *
 * attn (optional): per attention layer a [B,H,L,L] block, blocks concatenated.
 * taps (optional): stage-major [stage][B][rows][D] (row counts: orc_informer_tap_rows).
 */
ORC_API int orc_informer_forward(const cet_informer_cfg* c, const float* blob,
                                 const float* x_enc, const float* x_dec, float* out,
                                 float* attn, float* taps, int B) {
  int D = c->d_model, H = c->n_heads;
  int Ld = c->label_len + c->pred_len;
  int lens[64];
  int n_att = attn_lengths(c, lens, 64);
  /* offsets of each attention block in the [layer][B][H][L][L] buffer */
  size_t att_off[65]; att_off[0] = 0;
  for (int i = 0; i < n_att; ++i) att_off[i + 1] = att_off[i] + (size_t)B * H * lens[i] * lens[i];

  tap_list* all = NULL;
  if (taps) all = (tap_list*)calloc(B, sizeof(tap_list));

#pragma omp parallel for schedule(dynamic, 4)
  for (int b = 0; b < B; ++b) {
    const float* p = blob;
    tap_list tl; memset(&tl, 0, sizeof tl); tl.enabled = taps != NULL; tl.D = D;
    float* xe = (float*)malloc(sizeof(float) * (c->seq_len + 2) * D);
    float* mem = (float*)malloc(sizeof(float) * (c->seq_len + 2) * D * 2);
    float* xd = (float*)malloc(sizeof(float) * Ld * D);
    float* tmp = (float*)malloc(sizeof(float) * Ld * D);
    float* amap_tmp = NULL;
    size_t amap_total = 0;
    if (attn) {
      for (int i = 0; i < n_att; ++i) amap_total += (size_t)H * lens[i] * lens[i];
      amap_tmp = (float*)malloc(sizeof(float) * amap_total);
    }
    embedding(x_enc + (size_t)b * c->seq_len * c->enc_in, c->seq_len, c->enc_in, D, &p, xe);
    tap(&tl, xe, c->seq_len);
    /* the decoder embedding's weights come next in the blob; remember them */
    const float* p_dec_emb = p;
    p += (size_t)D * c->dec_in * 3 + D + (size_t)Ld * D;

    /* encoder (stack) */
    int mem_rows = 0;
    {
      float* am = amap_tmp;
      int inp_len = c->seq_len, cur = c->seq_len;
      int el_end = c->stack ? 3 : c->e_layers;
      for (int el = c->e_layers; el >= el_end; --el) {
        int take_rows = inp_len < cur ? inp_len : cur;
        /* x[:, -inp_len:, :] of the running tensor */
        memmove(xe, xe + (size_t)(cur - take_rows) * D, sizeof(float) * take_rows * D);
        cur = encoder_fwd(xe, take_rows, el, c, &p, attn ? &am : NULL, &tl);
        memcpy(mem + (size_t)mem_rows * D, xe, sizeof(float) * cur * D);
        mem_rows += cur;
        inp_len /= 2;
      }
    }
    tap(&tl, mem, mem_rows);

    /* decoder */
    embedding(x_dec + (size_t)b * Ld * c->dec_in, Ld, c->dec_in, D, &p_dec_emb, xd);
    tap(&tl, xd, Ld);
    for (int l = 0; l < c->d_layers; ++l) {
      attn_w sa, ca; take_attn(&p, D, H, &sa); take_attn(&p, D, H, &ca);
      int F = c->d_ff;
      const float* w1 = take(&p, (size_t)F * D); const float* b1 = take(&p, F);
      const float* w2 = take(&p, (size_t)D * F); const float* b2 = take(&p, D);
      const float* g1 = take(&p, D); const float* be1 = take(&p, D);
      const float* g2 = take(&p, D); const float* be2 = take(&p, D);
      const float* g3 = take(&p, D); const float* be3 = take(&p, D);
      attention(xd, Ld, xd, Ld, D, H, &sa, 1, tmp, NULL);
      for (int i = 0; i < Ld * D; ++i) xd[i] += tmp[i];
      layer_norm(xd, Ld, D, g1, be1);
      attention(xd, Ld, mem, mem_rows, D, H, &ca, 0, tmp, NULL);
      for (int i = 0; i < Ld * D; ++i) xd[i] += tmp[i];
      layer_norm(xd, Ld, D, g2, be2);
      float* hbuf = (float*)malloc(sizeof(float) * Ld * F);
      linear(xd, Ld, D, w1, b1, F, hbuf);
      for (int i = 0; i < Ld * F; ++i) hbuf[i] = act_fn(hbuf[i], c->activation);
      linear(hbuf, Ld, F, w2, b2, D, tmp);
      free(hbuf);
      for (int i = 0; i < Ld * D; ++i) xd[i] += tmp[i];
      layer_norm(xd, Ld, D, g3, be3);
      tap(&tl, xd, Ld);
    }
    {
      const float* g = take(&p, D); const float* be = take(&p, D);
      layer_norm(xd, Ld, D, g, be);
      tap(&tl, xd, Ld);
      const float* pw = take(&p, (size_t)c->c_out * D); const float* pb = take(&p, c->c_out);
      float* proj = (float*)malloc(sizeof(float) * Ld * c->c_out);
      linear(xd, Ld, D, pw, pb, c->c_out, proj);
      memcpy(out + (size_t)b * c->pred_len * c->c_out,
             proj + (size_t)(Ld - c->pred_len) * c->c_out,
             sizeof(float) * c->pred_len * c->c_out);
      free(proj);
    }
    if (attn) {
      const float* am = amap_tmp;
      for (int i = 0; i < n_att; ++i) {
        size_t n = (size_t)H * lens[i] * lens[i];
        memcpy(attn + att_off[i] + (size_t)b * n, am, sizeof(float) * n);
        am += n;
      }
      free(amap_tmp);
    }
    if (taps) all[b] = tl;
    free(xe); free(mem); free(xd); free(tmp);
  }
  if (taps) {
    size_t off = 0;
    for (int s = 0; s < all[0].n; ++s) {
      int rows = all[0].recs[s].rows;
      for (int b = 0; b < B; ++b) {
        memcpy(taps + off + (size_t)b * rows * D, all[b].recs[s].data,
               sizeof(float) * rows * D);
        free(all[b].recs[s].data);
      }
      off += (size_t)B * rows * D;
    }
    for (int b = 0; b < B; ++b) free(all[b].recs);
    free(all);
  }
  return 0;
}

/* Number of tap stages and their row counts, so a test can slice the taps buffer. */
ORC_API int orc_informer_tap_rows(const cet_informer_cfg* c, int* rows, int cap) {
  int n = 0;
#define PUSH(r) do { if (n < cap) rows[n] = (r); n++; } while (0)
  int Ld = c->label_len + c->pred_len;
  PUSH(c->seq_len);
  int inp_len = c->seq_len, cur = c->seq_len, mem_rows = 0;
  int el_end = c->stack ? 3 : c->e_layers;
  for (int el = c->e_layers; el >= el_end; --el) {
    int L = inp_len < cur ? inp_len : cur;
    for (int l = 0; l < el; ++l) {
      PUSH(L);
      if (c->distil && l < el - 1) { L = (L + 1) / 2 + 1; PUSH(L); }
    }
    PUSH(L);
    cur = L; mem_rows += L; inp_len /= 2;
  }
  PUSH(mem_rows);
  PUSH(Ld);
  for (int l = 0; l < c->d_layers; ++l) PUSH(Ld);
  PUSH(Ld);
#undef PUSH
  return n;
}

/* ------------------------------------------------------------------------------------ */
/* Recurrent predictors                                                                  */

static float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

/* RNN/GRU/LSTM .test_data (model.py:298-312, 373-388, 449-465) with the Unit forward
 * (model.py:253-270, 327-344, 403-420): encoder Linear -> num_layers cells (one time step)
 * -> decoder Linear; zero initial state; closed loop after seq_len steps.
 * Gate row order follows torch.nn.GRU [r,z,n] and torch.nn.LSTM [i,f,g,o]. */
ORC_API int orc_rnn_rollout(int kind, int F, int I, int Hs, int layers, const float* blob,
                            const float* x, float* out, int B, int seq_len, int pred_len) {
  int G = kind == CET_RNN_TANH ? 1 : (kind == CET_RNN_GRU ? 3 : 4);
  const float* p = blob;
  const float* enc_w = take(&p, (size_t)I * F); const float* enc_b = take(&p, I);
  const float *w_ih[8], *w_hh[8], *b_ih[8], *b_hh[8];
  if (layers > 8) return 1;
  for (int l = 0; l < layers; ++l) {
    int in = l == 0 ? I : Hs;
    w_ih[l] = take(&p, (size_t)G * Hs * in); w_hh[l] = take(&p, (size_t)G * Hs * Hs);
    b_ih[l] = take(&p, (size_t)G * Hs); b_hh[l] = take(&p, (size_t)G * Hs);
  }
  const float* dec_w = take(&p, (size_t)F * Hs); const float* dec_b = take(&p, F);

#pragma omp parallel for schedule(dynamic, 4)
  for (int b = 0; b < B; ++b) {
    float* h = (float*)calloc((size_t)layers * Hs, sizeof(float));
    float* cst = (float*)calloc((size_t)layers * Hs, sizeof(float));
    float* gi = (float*)malloc(sizeof(float) * G * Hs);
    float* gh = (float*)malloc(sizeof(float) * G * Hs);
    float* inp = (float*)malloc(sizeof(float) * (I > Hs ? I : Hs));
    float* hn = (float*)malloc(sizeof(float) * Hs);
    float* o = (float*)malloc(sizeof(float) * F);
    float* xin = (float*)malloc(sizeof(float) * F);
    for (int idx = 0; idx < seq_len + pred_len - 1; ++idx) {
      if (idx < seq_len) memcpy(xin, x + ((size_t)b * seq_len + idx) * F, sizeof(float) * F);
      else memcpy(xin, o, sizeof(float) * F);
      linear(xin, 1, F, enc_w, enc_b, I, inp);
      int in = I;
      for (int l = 0; l < layers; ++l) {
        float* hl = h + (size_t)l * Hs;
        float* cl = cst + (size_t)l * Hs;
        linear(inp, 1, in, w_ih[l], b_ih[l], G * Hs, gi);
        linear(hl, 1, Hs, w_hh[l], b_hh[l], G * Hs, gh);
        for (int j = 0; j < Hs; ++j) {
          if (kind == CET_RNN_TANH) {
            hn[j] = tanhf(gi[j] + gh[j]);
          } else if (kind == CET_RNN_GRU) {
            float r = sigmoidf_(gi[j] + gh[j]);
            float z = sigmoidf_(gi[Hs + j] + gh[Hs + j]);
            float n = tanhf(gi[2 * Hs + j] + r * gh[2 * Hs + j]);
            hn[j] = (1.0f - z) * n + z * hl[j];
          } else {
            float ig = sigmoidf_(gi[j] + gh[j]);
            float fg = sigmoidf_(gi[Hs + j] + gh[Hs + j]);
            float gg = tanhf(gi[2 * Hs + j] + gh[2 * Hs + j]);
            float og = sigmoidf_(gi[3 * Hs + j] + gh[3 * Hs + j]);
            cl[j] = fg * cl[j] + ig * gg;
            hn[j] = og * tanhf(cl[j]);
          }
        }
        memcpy(hl, hn, sizeof(float) * Hs);
        memcpy(inp, hn, sizeof(float) * Hs);
        in = Hs;
      }
      linear(inp, 1, Hs, dec_w, dec_b, F, o);
      if (idx >= seq_len - 1)
        memcpy(out + ((size_t)b * pred_len + (idx - (seq_len - 1))) * F, o, sizeof(float) * F);
    }
    free(h); free(cst); free(gi); free(gh); free(inp); free(hn); free(o); free(xin);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* LMMSE equalizer (PARITY UNPINNED, see the header comment)                              */

/* Per resource element, in float64:
 *   whiten with the diagonal S:  Hw = S^-1/2 H,  yw = S^-1/2 y
 *   A = Hw^H Hw + I ; G = A^-1 Hw^H ; d = diag(G Hw) ; x_hat = (G yw)/d ; no_eff = Re(1/d - 1)
 * which is what Sionna's lmmse_equalizer(y, h, s, whiten_interference=True) defines and what
 * LMMSEEqualizer applies per RE (call sites EndtoEndtest.py:133,191).
 * Inputs/outputs are float32 interleaved complex like the C ABI's. */
ORC_API int orc_lmmse(const float* y, const float* H, int no_mode, float no_scalar,
                      const float* no_arr, float* x_hat, float* no_eff, long N, int Nr,
                      int Ns) {
  if (Nr > 64 || Ns > 16) return 1;
#pragma omp parallel for schedule(static)
  for (long n = 0; n < N; ++n) {
    double hr[64][16], hi[64][16], yr[64], yi[64];
    for (int r = 0; r < Nr; ++r) {
      double s = no_mode == CET_NO_SCALAR ? (double)no_scalar
               : no_mode == CET_NO_PER_RE ? (double)no_arr[n]
                                          : (double)no_arr[n * Nr + r];
      double w = 1.0 / sqrt(s);
      yr[r] = w * y[(n * Nr + r) * 2]; yi[r] = w * y[(n * Nr + r) * 2 + 1];
      for (int c = 0; c < Ns; ++c) {
        hr[r][c] = w * H[((n * Nr + r) * Ns + c) * 2];
        hi[r][c] = w * H[((n * Nr + r) * Ns + c) * 2 + 1];
      }
    }
    /* A = Hw^H Hw + I  and the right-hand sides  R = Hw^H [Hw | yw]  (Ns x (Ns+1)) */
    double ar[16][16], ai[16][16], rr[16][17], ri[16][17];
    for (int a = 0; a < Ns; ++a)
      for (int c = 0; c <= Ns; ++c) {
        double sr = 0, si = 0;
        for (int r = 0; r < Nr; ++r) {
          double br = c < Ns ? hr[r][c] : yr[r], bi = c < Ns ? hi[r][c] : yi[r];
          /* conj(h[r][a]) * b */
          sr += hr[r][a] * br + hi[r][a] * bi;
          si += hr[r][a] * bi - hi[r][a] * br;
        }
        rr[a][c] = sr; ri[a][c] = si;
        if (c < Ns) { ar[a][c] = sr + (a == c ? 1.0 : 0.0); ai[a][c] = si; }
      }
    /* Gauss-Jordan with partial pivoting on [A | R] */
    for (int k = 0; k < Ns; ++k) {
      int piv = k; double best = ar[k][k] * ar[k][k] + ai[k][k] * ai[k][k];
      for (int r = k + 1; r < Ns; ++r) {
        double m = ar[r][k] * ar[r][k] + ai[r][k] * ai[r][k];
        if (m > best) { best = m; piv = r; }
      }
      if (piv != k) {
        for (int c = 0; c < Ns; ++c) {
          double t = ar[k][c]; ar[k][c] = ar[piv][c]; ar[piv][c] = t;
          t = ai[k][c]; ai[k][c] = ai[piv][c]; ai[piv][c] = t;
        }
        for (int c = 0; c <= Ns; ++c) {
          double t = rr[k][c]; rr[k][c] = rr[piv][c]; rr[piv][c] = t;
          t = ri[k][c]; ri[k][c] = ri[piv][c]; ri[piv][c] = t;
        }
      }
      double pr = ar[k][k], pi = ai[k][k], den = pr * pr + pi * pi;
      double ir = pr / den, ii = -pi / den;
      for (int c = 0; c < Ns; ++c) {
        double t = ar[k][c] * ir - ai[k][c] * ii; ai[k][c] = ar[k][c] * ii + ai[k][c] * ir;
        ar[k][c] = t;
      }
      for (int c = 0; c <= Ns; ++c) {
        double t = rr[k][c] * ir - ri[k][c] * ii; ri[k][c] = rr[k][c] * ii + ri[k][c] * ir;
        rr[k][c] = t;
      }
      for (int r = 0; r < Ns; ++r) {
        if (r == k) continue;
        double fr = ar[r][k], fi = ai[r][k];
        for (int c = 0; c < Ns; ++c) {
          ar[r][c] -= fr * ar[k][c] - fi * ai[k][c];
          ai[r][c] -= fr * ai[k][c] + fi * ar[k][c];
        }
        for (int c = 0; c <= Ns; ++c) {
          rr[r][c] -= fr * rr[k][c] - fi * ri[k][c];
          ri[r][c] -= fr * ri[k][c] + fi * rr[k][c];
        }
      }
    }
    /* rr/ri now hold  A^-1 Hw^H [Hw | yw] = [G Hw | G yw] */
    for (int s = 0; s < Ns; ++s) {
      double dr = rr[s][s], di = ri[s][s], den = dr * dr + di * di;
      double zr = rr[s][Ns], zi = ri[s][Ns];
      x_hat[(n * Ns + s) * 2] = (float)((zr * dr + zi * di) / den);
      x_hat[(n * Ns + s) * 2 + 1] = (float)((zi * dr - zr * di) / den);
      no_eff[n * Ns + s] = (float)(dr / den - 1.0);
    }
  }
  return 0;
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
