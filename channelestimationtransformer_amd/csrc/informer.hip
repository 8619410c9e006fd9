// Transformer channel predictor (Informer / InformerStack forward, eval mode) for gfx950.
// Restates LocalSimulations/models/model.py:220-234 with embed.py, encoder.py, decoder.py and
// attn.py as ONE kernel: one wavefront carries one sequence through the whole network.
//
// Data layout ("T layout"). The running activation x[rows<=32][64] never sits in memory: it
// lives in MFMA accumulator registers, transposed. For the 16x16 tile (rt, mt) lane
// l = 16*g + c holds x[row 16*rt + c][feature 16*mt + 4*g + i], i = 0..3, in one f32x4.
// That is exactly the C/D layout of v_mfma_f32_16x16x4_f32 when the product is taken as
// out^T = W * x^T (A operand = weight rows, B operand = activation). Because the k index of
// that instruction is the lane group g, register i of a T tile is directly the B operand of
// k-step (mt, i) of the next product: Linear -> GELU -> Linear -> +residual -> LayerNorm
// chains run register to register, and every weight matrix is read in its PyTorch row-major
// layout with one 16-byte load per lane and k-block (no packing).
// LDS (26 KB per wave) only holds what needs a cross-row view: Q, K, V of the attention at
// hand, and the staging rows of the circular convolutions / max-pool.
// Attention: S^T = K Q^T per head on MFMA (A = K rows, B = Q rows, k = the 8 head features),
// which leaves every query's scores in one lane column: softmax is 8 in-lane values plus two
// cross-lane steps. The normalised probabilities are then the B operand of the P*V product
// as they stand; two heads share one 16-feature output tile by zeroing the other head's
// rows of the V^T operand.
// All arithmetic is fp32 (MFMA f32 = exact fmaf chains); exp/erf/sqrt are the accurate forms.
#include <cstdlib>
#include <new>
#include <vector>

#include "cet_internal.h"

namespace cet {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DM = 64;    // d_model
constexpr int NHEAD = 8;  // heads (head dim 8)
constexpr int LDW = 68;   // LDS row stride in floats (64 + 4: conflict-free fragment reads)
constexpr int MAXROWS = 32;

struct Dims {
  int enc_in, dec_in, c_out, seq_len, label_len, pred_len;
  int n_enc, enc_layers[8], distil, activation, d_layers;
  int n_attn;          // attention maps produced
  int attn_len[64];    // their sequence lengths
  int n_taps;
  int tap_rows[128];
  int mem_rows;        // rows of the concatenated encoder outputs (cross-attention keys)
};

struct KArgs {
  const float* w;  // device blob, execution order
  const float* x_enc;
  const float* x_dec;
  float* out;
  float* attn;  // nullable
  float* taps;  // nullable
  int B;
  int enc_in, dec_in, c_out, seq_len, label_len, pred_len;
  int n_enc, distil, activation, d_layers;
  int enc_layers[8];
  int region_rows;  // rows of each of the three LDS regions (Q | K | V)
};

#define DEV __device__ __forceinline__

DEV f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
DEV f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
DEV f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
DEV void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

struct Lane {
  int c;  // lane & 15 : row inside a tile (B operand / C column)
  int g;  // lane >> 4 : k group / C row block
};

// out^T[rt][mo] = W[64 x 64] * in^T  (+ bias). W row-major [out][in], bias[out].
template <int RT>
DEV void linear64(const float* __restrict__ W, const float* __restrict__ bias, const Lane ln,
                  const f32x4 (&in)[2][4], f32x4 (&out)[2][4]) {
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) {
    f32x4 a[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a[mi] = ld4(W + (16 * mo + ln.c) * DM + 16 * mi + 4 * ln.g);
    const f32x4 bv = ld4(bias + 16 * mo + 4 * ln.g);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 acc = bv;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = mfma(a[mi][i], in[rt][mi][i], acc);
      out[rt][mo] = acc;
    }
  }
}

// Rows [0, nrows) of the tiles go to LDS; rows of the padded tiles beyond that are dropped
// (the regions are sized to the sequence, not to whole tiles).
template <int RT>
DEV void store_rows(float* buf, const Lane ln, const f32x4 (&v)[2][4], int nrows) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if (16 * rt + ln.c < nrows) {
#pragma unroll
      for (int m = 0; m < 4; ++m) st4(buf + (16 * rt + ln.c) * LDW + 16 * m + 4 * ln.g, v[rt][m]);
    }
  }
}

// torch.nn.LayerNorm(64), eps 1e-5, over the feature axis (in-lane 16 values + lane groups).
template <int RT>
DEV void layer_norm(f32x4 (&x)[2][4], const float* __restrict__ gamma,
                    const float* __restrict__ beta, const Lane ln) {
  f32x4 gm[4], bt[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    gm[m] = ld4(gamma + 16 * m + 4 * ln.g);
    bt[m] = ld4(beta + 16 * m + 4 * ln.g);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) s += x[rt][m][i];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / DM);
    float q = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = x[rt][m][i] - mean;
        q = fmaf(d, d, q);
      }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const float inv = 1.0f / sqrtf(q * (1.0f / DM) + 1e-5f);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        x[rt][m][i] = (x[rt][m][i] - mean) * inv * gm[m][i] + bt[m][i];
  }
}

DEV float act(float v, int activation) {
  if (activation == 0) return v > 0.f ? v : 0.f;
  return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

template <int RT>
DEV void add_into(f32x4 (&x)[2][4], const f32x4 (&y)[2][4]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int m = 0; m < 4; ++m) x[rt][m] += y[rt][m];
}

// Copies the first `rows` rows to the debug buffer, stage-major [stage][B][rows][64].
template <int RT>
DEV void tap(float*& taps, int B, int b, int rows, const Lane ln, const f32x4 (&x)[2][4]) {
  if (!taps) return;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = 16 * rt + ln.c;
    if (r < rows) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        st4(taps + ((size_t)b * rows + r) * DM + 16 * m + 4 * ln.g, x[rt][m]);
    }
  }
  taps += (size_t)B * rows * DM;
}

// FullAttention (attn.py:18-36) on the Q/K/V rows in LDS; result in T layout (ctx).
// RQ / RK: 16-row tiles of queries / keys. amap: this sample's [8][Lq][Lk] map or null.
template <int RQ, int RK>
DEV void attention(const float* Qb, const float* Kb, const float* Vb, int Lq, int Lk, bool causal,
                   const Lane ln, f32x4 (&ctx)[2][4], float* amap) {
  const float scale = 0.35355339059327373f;  // 1/sqrt(8), as float(1./sqrt(E))
  const bool upper = (ln.c & 8) != 0;        // lane feeds the second head of a pair
#pragma unroll
  for (int hp = 0; hp < 4; ++hp) {
    f32x4 p[2][RK][RQ];  // [head of the pair][key tile][query tile]
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int col = 8 * (2 * hp + hh) + 2 * ln.g;  // this lane's two head features
      float2 kf[RK], qf[RQ];
#pragma unroll
      for (int kt = 0; kt < RK; ++kt)
        kf[kt] = *reinterpret_cast<const float2*>(Kb + (16 * kt + ln.c) * LDW + col);
#pragma unroll
      for (int qt = 0; qt < RQ; ++qt)
        qf[qt] = *reinterpret_cast<const float2*>(Qb + (16 * qt + ln.c) * LDW + col);
      // scores^T: lane (query 16*qt + c) holds keys 16*kt + 4*g + i
#pragma unroll
      for (int kt = 0; kt < RK; ++kt)
#pragma unroll
        for (int qt = 0; qt < RQ; ++qt) {
          f32x4 s = zero4();
          s = mfma(kf[kt].x, qf[qt].x, s);
          s = mfma(kf[kt].y, qf[qt].y, s);
          p[hh][kt][qt] = s;
        }
      // masked, scaled softmax over the keys of each query
#pragma unroll
      for (int qt = 0; qt < RQ; ++qt) {
        const int query = 16 * qt + ln.c;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < RK; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int key = 16 * kt + 4 * ln.g + i;
            const bool ok = key < Lk && (!causal || key <= query);
            const float t = ok ? scale * p[hh][kt][qt][i] : -INFINITY;
            p[hh][kt][qt][i] = t;
            mx = fmaxf(mx, t);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < RK; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float e = expf(p[hh][kt][qt][i] - mx);
            p[hh][kt][qt][i] = e;
            sum += e;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
#pragma unroll
        for (int kt = 0; kt < RK; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) p[hh][kt][qt][i] = p[hh][kt][qt][i] / sum;
        if (amap && query < Lq) {
          float* row = amap + ((size_t)(2 * hp + hh) * Lq + query) * Lk;
#pragma unroll
          for (int kt = 0; kt < RK; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int key = 16 * kt + 4 * ln.g + i;
              if (key < Lk) row[key] = p[hh][kt][qt][i];
            }
        }
      }
    }
    // context^T tile (features 16*hp .. 16*hp+15): A = V^T with the other head's rows zeroed
    f32x4 acc[RQ];
#pragma unroll
    for (int qt = 0; qt < RQ; ++qt) acc[qt] = zero4();
#pragma unroll
    for (int kt = 0; kt < RK; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int vr = 16 * kt + 4 * ln.g + i;
        vr = vr < Lk ? vr : Lk - 1;  // padded keys carry probability 0; stay inside the region
        const float v = Vb[vr * LDW + 16 * hp + ln.c];
        const float v0 = upper ? 0.f : v;
        const float v1 = upper ? v : 0.f;
#pragma unroll
        for (int qt = 0; qt < RQ; ++qt) {
          acc[qt] = mfma(v0, p[0][kt][qt][i], acc[qt]);
          acc[qt] = mfma(v1, p[1][kt][qt][i], acc[qt]);
        }
      }
#pragma unroll
    for (int qt = 0; qt < RQ; ++qt) ctx[qt][hp] = acc[qt];
  }
}

// Blob block sizes (floats), execution order.
constexpr int SZ_ATTN = 4 * (DM * DM + DM);
constexpr int SZ_FFN = 2 * (DM * DM + DM);

// AttentionLayer (attn.py:140-156): projections -> LDS, attention, output projection.
// xq: query-side input, xkv: key/value-side input (T layout). Result in `o`.
template <int RQ, int RK>
DEV void attention_block(const float* __restrict__ w, float* smem, int R, const Lane ln,
                         const f32x4 (&xq)[2][4], const f32x4 (&xkv)[2][4], int Lq, int Lk,
                         bool causal, float* amap, f32x4 (&o)[2][4]) {
  float* Qb = smem;  // R rows each; a padded tile read may run into the next region (finite data)
  float* Kb = smem + R * LDW;
  float* Vb = smem + 2 * R * LDW;
  f32x4 t[2][4];
  __syncthreads();  // earlier readers of the Q/K/V regions are done
  linear64<RQ>(w, w + DM * DM, ln, xq, t);
  store_rows<RQ>(Qb, ln, t, Lq);
  linear64<RK>(w + (DM * DM + DM), w + (DM * DM + DM) + DM * DM, ln, xkv, t);
  store_rows<RK>(Kb, ln, t, Lk);
  linear64<RK>(w + 2 * (DM * DM + DM), w + 2 * (DM * DM + DM) + DM * DM, ln, xkv, t);
  store_rows<RK>(Vb, ln, t, Lk);
  __syncthreads();
  f32x4 ctx[2][4];
  attention<RQ, RK>(Qb, Kb, Vb, Lq, Lk, causal, ln, ctx, amap);
  linear64<RQ>(w + 3 * (DM * DM + DM), w + 3 * (DM * DM + DM) + DM * DM, ln, ctx, o);
}

// conv1 -> activation -> conv2 of EncoderLayer / DecoderLayer (1x1 convolutions = Linear).
template <int RT>
DEV void ffn(const float* __restrict__ w, const Lane ln, int activation, const f32x4 (&x)[2][4],
             f32x4 (&y)[2][4]) {
  f32x4 h[2][4];
  linear64<RT>(w, w + DM * DM, ln, x, h);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) h[rt][m][i] = act(h[rt][m][i], activation);
  linear64<RT>(w + DM * DM + DM, w + 2 * DM * DM + DM, ln, h, y);
}

// EncoderLayer.forward (encoder.py:37-53). Block: attn | ffn | norm1 g,b | norm2 g,b.
constexpr int SZ_ENC_LAYER = SZ_ATTN + SZ_FFN + 4 * DM;
template <int RT>
DEV void encoder_layer(const float* __restrict__ w, float* smem, int R, const Lane ln, int L,
                       int activation, float* amap, f32x4 (&x)[2][4]) {
  f32x4 y[2][4];
  attention_block<RT, RT>(w, smem, R, ln, x, x, L, L, false, amap, y);
  add_into<RT>(x, y);
  const float* nw = w + SZ_ATTN + SZ_FFN;
  layer_norm<RT>(x, nw, nw + DM, ln);
  ffn<RT>(w + SZ_ATTN, ln, activation, x, y);
  add_into<RT>(x, y);
  layer_norm<RT>(x, nw + 2 * DM, nw + 3 * DM, ln);
}

// DecoderLayer.forward (decoder.py:20-36).
// Block: self attn | cross attn | ffn | norm1 | norm2 | norm3.
constexpr int SZ_DEC_LAYER = 2 * SZ_ATTN + SZ_FFN + 6 * DM;
template <int RD, int RM>
DEV void decoder_layer(const float* __restrict__ w, float* smem, int R, const Lane ln, int Ld,
                       int Lm, int activation, const f32x4 (&mem)[2][4], f32x4 (&x)[2][4]) {
  f32x4 y[2][4];
  const float* nw = w + 2 * SZ_ATTN + SZ_FFN;
  attention_block<RD, RD>(w, smem, R, ln, x, x, Ld, Ld, true, nullptr, y);
  add_into<RD>(x, y);
  layer_norm<RD>(x, nw, nw + DM, ln);
  attention_block<RD, RM>(w + SZ_ATTN, smem, R, ln, x, mem, Ld, Lm, false, nullptr, y);
  add_into<RD>(x, y);
  layer_norm<RD>(x, nw + 2 * DM, nw + 3 * DM, ln);
  ffn<RD>(w + 2 * SZ_ATTN, ln, activation, x, y);
  add_into<RD>(x, y);
  layer_norm<RD>(x, nw + 4 * DM, nw + 5 * DM, ln);
}

// ConvLayer.forward (encoder.py:17-23): Conv1d(64->64, k=3, circular pad 2) -> BatchNorm1d
// (eval; folded to alpha/beta) -> ELU -> MaxPool1d(3, 2, 1). L rows in, returns rows out.
// Block: w[3][64][64] (tap-major) | bias | alpha | beta.
constexpr int SZ_DISTIL = 3 * DM * DM + 3 * DM;
template <int RT>  // tiles covering L + 2 conv outputs
DEV int distil(const float* __restrict__ w, float* smem, int R, const Lane ln, int L,
               f32x4 (&x)[2][4]) {
  float* Xb = smem;            // L input rows
  float* Yb = smem + R * LDW;  // L + 2 activated conv outputs (R + L + 2 <= 3 R)
  __syncthreads();
  store_rows<RT>(Xb, ln, x, L);
  __syncthreads();
  const int Lc = L + 2;
  f32x4 y[2][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int t = 16 * rt + ln.c;
    t = t < Lc ? t : Lc - 1;
    f32x4 bf[3][4];  // shifted input rows as B operands
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int src = t + j - 2;  // circular padding of 2 on both sides
      src = src < 0 ? src + L : src;
      src = src >= L ? src - L : src;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) bf[j][mi] = ld4(Xb + src * LDW + 16 * mi + 4 * ln.g);
    }
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      f32x4 acc = ld4(w + 3 * DM * DM + 16 * mo + 4 * ln.g);  // bias
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const f32x4 a = ld4(w + (j * DM + 16 * mo + ln.c) * DM + 16 * mi + 4 * ln.g);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = mfma(a[i], bf[j][mi][i], acc);
        }
      const f32x4 al = ld4(w + 3 * DM * DM + DM + 16 * mo + 4 * ln.g);
      const f32x4 be = ld4(w + 3 * DM * DM + 2 * DM + 16 * mo + 4 * ln.g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = acc[i] * al[i] + be[i];
        acc[i] = v > 0.f ? v : (expf(v) - 1.0f);
      }
      y[rt][mo] = acc;
    }
  }
  store_rows<RT>(Yb, ln, y, Lc);
  __syncthreads();
  const int Lo = (Lc - 1) / 2 + 1;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    if (rt < RT) {
      int u = 16 * rt + ln.c;
      u = u < Lo ? u : Lo - 1;
      const int r0 = 2 * u - 1 < 0 ? 0 : 2 * u - 1;         // window rows clipped to [0, Lc)
      const int r2 = 2 * u + 1 > Lc - 1 ? Lc - 1 : 2 * u + 1;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 a = ld4(Yb + r0 * LDW + 16 * m + 4 * ln.g);
        const f32x4 b = ld4(Yb + (2 * u) * LDW + 16 * m + 4 * ln.g);
        const f32x4 c = ld4(Yb + r2 * LDW + 16 * m + 4 * ln.g);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[rt][m][i] = fmaxf(fmaxf(a[i], b[i]), c[i]);
      }
    }
  }
  return Lo;
}

// DataEmbedding.forward (embed.py:50-53): circular Conv1d(c_in->64, k=3, pad 1) + positional
// table. Block: w[3][64][c_in] (tap-major) | bias[64] | pe[L][64].  NKB = c_in / 16.
template <int RT, int NKB>
DEV void embed(const float* __restrict__ w, const float* __restrict__ xg, float* smem,
               const Lane ln, int lane, int L, f32x4 (&x)[2][4]) {
  constexpr int CIN = 16 * NKB;
  constexpr int XW = CIN + 4;  // staging row stride
  __syncthreads();
  for (int i = lane; i < L * (CIN / 4); i += 64) {
    const int r = i / (CIN / 4), c4 = i % (CIN / 4);
    st4(smem + r * XW + 4 * c4, ld4(xg + (size_t)r * CIN + 4 * c4));
  }
  __syncthreads();
  const float* bias = w + 3 * DM * CIN;
  const float* pe = bias + DM;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int t = 16 * rt + ln.c;
    t = t < L ? t : L - 1;
    f32x4 bf[3][NKB];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int src = t + j - 1;
      src = src < 0 ? src + L : src;
      src = src >= L ? src - L : src;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) bf[j][kb] = ld4(smem + src * XW + 16 * kb + 4 * ln.g);
    }
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      f32x4 acc = zero4();
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          const f32x4 a = ld4(w + (j * DM + 16 * mo + ln.c) * CIN + 16 * kb + 4 * ln.g);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = mfma(a[i], bf[j][kb][i], acc);
        }
      acc += ld4(bias + 16 * mo + 4 * ln.g);
      acc += ld4(pe + t * DM + 16 * mo + 4 * ln.g);
      x[rt][mo] = acc;
    }
  }
}

// Rows [first, first+n) of x (via LDS) -> rows [dst, dst+n) of `into`; other rows of `into`
// keep their value. Used for x[:, -inp_len:, :] and for the concatenation of the encoder
// outputs (encoder.py:96-99).
DEV void move_rows(float* smem, const Lane ln, const f32x4 (&x)[2][4], int first, int n, int dst,
                   f32x4 (&into)[2][4]) {
  __syncthreads();
  store_rows<2>(smem, ln, x, first + n);
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int r = 16 * rt + ln.c;
    if (r >= dst && r < dst + n) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        into[rt][m] = ld4(smem + (r - dst + first) * LDW + 16 * m + 4 * ln.g);
    }
  }
}

// WPE: waves per SIMD the register allocation must allow (occupancy is then set by LDS).
template <int WPE>
__global__ __launch_bounds__(64, WPE) void informer_kernel(const KArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  const Lane ln{lane & 15, lane >> 4};
  const int b = blockIdx.x;
  const int R = a.region_rows;

  // Rows the padded 16-row tiles may touch must hold finite numbers from the start.
  for (int i = lane; i < 3 * R * LDW / 4; i += 64) st4(smem + 4 * i, zero4());

  const float* w = a.w;
  float* taps = a.taps;
  float* attn = a.attn;
  f32x4 x[2][4], mem[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int m = 0; m < 4; ++m) x[rt][m] = mem[rt][m] = zero4();

  // ---- encoder embedding ------------------------------------------------------------
  int L = a.seq_len;
  {
    const float* xg = a.x_enc + (size_t)b * L * a.enc_in;
    if (a.enc_in == 16) embed<2, 1>(w, xg, smem, ln, lane, L, x);
    else embed<2, 2>(w, xg, smem, ln, lane, L, x);
    w += 3 * DM * a.enc_in + DM + L * DM;
    tap<2>(taps, a.B, b, L, ln, x);
  }
  const float* w_dec_emb = w;
  const int Ld = a.label_len + a.pred_len;
  w += 3 * DM * a.dec_in + DM + Ld * DM;

  // ---- encoder stack (encoder.py:87-101) ---------------------------------------------
  int cur = L, inp_len = L, mem_rows = 0;
  for (int e = 0; e < a.n_enc; ++e) {
    const int el = a.enc_layers[e];
    const int take = inp_len < cur ? inp_len : cur;
    if (take < cur) move_rows(smem, ln, x, cur - take, take, 0, x);
    L = take;
    for (int l = 0; l < el; ++l) {
      float* amap = nullptr;
      if (attn) {
        amap = attn + (size_t)b * NHEAD * L * L;
        attn += (size_t)a.B * NHEAD * L * L;
      }
      if (L > 16) encoder_layer<2>(w, smem, R, ln, L, a.activation, amap, x);
      else encoder_layer<1>(w, smem, R, ln, L, a.activation, amap, x);
      w += SZ_ENC_LAYER;
      tap<2>(taps, a.B, b, L, ln, x);
      if (a.distil && l < el - 1) {
        if (L + 2 > 16) L = distil<2>(w, smem, R, ln, L, x);
        else L = distil<1>(w, smem, R, ln, L, x);
        w += SZ_DISTIL;
        tap<2>(taps, a.B, b, L, ln, x);
      }
    }
    if (L > 16) layer_norm<2>(x, w, w + DM, ln);
    else layer_norm<1>(x, w, w + DM, ln);
    w += 2 * DM;
    tap<2>(taps, a.B, b, L, ln, x);
    move_rows(smem, ln, x, 0, L, mem_rows, mem);
    mem_rows += L;
    cur = L;
    inp_len /= 2;
  }
  tap<2>(taps, a.B, b, mem_rows, ln, mem);

  // ---- decoder ---------------------------------------------------------------------
  {
    const float* xg = a.x_dec + (size_t)b * Ld * a.dec_in;
    if (a.dec_in == 16) embed<2, 1>(w_dec_emb, xg, smem, ln, lane, Ld, x);
    else embed<2, 2>(w_dec_emb, xg, smem, ln, lane, Ld, x);
    tap<2>(taps, a.B, b, Ld, ln, x);
  }
  for (int l = 0; l < a.d_layers; ++l) {
    if (Ld > 16) {
      if (mem_rows > 16) decoder_layer<2, 2>(w, smem, R, ln, Ld, mem_rows, a.activation, mem, x);
      else decoder_layer<2, 1>(w, smem, R, ln, Ld, mem_rows, a.activation, mem, x);
    } else {
      if (mem_rows > 16) decoder_layer<1, 2>(w, smem, R, ln, Ld, mem_rows, a.activation, mem, x);
      else decoder_layer<1, 1>(w, smem, R, ln, Ld, mem_rows, a.activation, mem, x);
    }
    w += SZ_DEC_LAYER;
    tap<2>(taps, a.B, b, Ld, ln, x);
  }
  layer_norm<2>(x, w, w + DM, ln);
  w += 2 * DM;
  tap<2>(taps, a.B, b, Ld, ln, x);

  // ---- projection Linear(64 -> c_out) and the last pred_len rows ------------------------
  // Block: W[ceil16(c_out)][64] zero padded | bias[ceil16(c_out)].
  const int n_mo = (a.c_out + 15) / 16;
  const float* pb = w + n_mo * 16 * DM;
  for (int mo = 0; mo < n_mo; ++mo) {
    f32x4 wa[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) wa[mi] = ld4(w + (16 * mo + ln.c) * DM + 16 * mi + 4 * ln.g);
    const f32x4 bv = ld4(pb + 16 * mo + 4 * ln.g);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      f32x4 acc = bv;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = mfma(wa[mi][i], x[rt][mi][i], acc);
      const int r = 16 * rt + ln.c;
      const int f0 = 16 * mo + 4 * ln.g;
      if (r >= Ld - a.pred_len && r < Ld && f0 < a.c_out)
        st4(a.out + ((size_t)b * a.pred_len + (r - (Ld - a.pred_len))) * a.c_out + f0, acc);
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Host side

struct InformerModel {
  cet_informer_cfg cfg;
  Dims d;
  float* dev_blob = nullptr;
  size_t attn_per_sample = 0;  // floats
  size_t taps_per_sample = 0;
};

static bool supported(const cet_informer_cfg& c) {
  if (c.d_model != DM || c.n_heads != NHEAD || c.d_ff != DM) return false;
  if (c.enc_in != 16 && c.enc_in != 32) return false;
  if (c.dec_in != 16 && c.dec_in != 32) return false;
  if (c.c_out < 4 || c.c_out > 64 || c.c_out % 4) return false;
  if (c.seq_len < 3 || c.seq_len > MAXROWS - (c.distil ? 2 : 0)) return false;
  const int Ld = c.label_len + c.pred_len;
  if (c.pred_len < 1 || c.label_len < 0 || Ld < 3 || Ld > MAXROWS) return false;
  if (c.e_layers < 1 || c.e_layers > 8 || c.d_layers < 1 || c.d_layers > 16) return false;
  if (c.stack && c.e_layers < 3) return false;  // the reference's stack would be empty
  if (c.activation != 0 && c.activation != 1) return false;
  return true;
}

// Fills the execution-order description: encoder depths, lengths of every stage.
static bool plan(const cet_informer_cfg& c, Dims& d) {
  d = Dims{};
  d.enc_in = c.enc_in; d.dec_in = c.dec_in; d.c_out = c.c_out;
  d.seq_len = c.seq_len; d.label_len = c.label_len; d.pred_len = c.pred_len;
  d.distil = c.distil; d.activation = c.activation; d.d_layers = c.d_layers;
  if (c.stack) {
    for (int el = c.e_layers; el > 2; --el) d.enc_layers[d.n_enc++] = el;
  } else {
    d.enc_layers[d.n_enc++] = c.e_layers;
  }
  const int Ld = c.label_len + c.pred_len;
  int cur = c.seq_len, inp_len = c.seq_len, mem_rows = 0;
  d.tap_rows[d.n_taps++] = c.seq_len;
  for (int e = 0; e < d.n_enc; ++e) {
    int L = inp_len < cur ? inp_len : cur;
    if (L < 1) return false;
    for (int l = 0; l < d.enc_layers[e]; ++l) {
      d.attn_len[d.n_attn++] = L;
      d.tap_rows[d.n_taps++] = L;
      if (c.distil && l < d.enc_layers[e] - 1) {
        L = (L + 1) / 2 + 1;
        d.tap_rows[d.n_taps++] = L;
      }
    }
    d.tap_rows[d.n_taps++] = L;
    cur = L;
    mem_rows += L;
    inp_len /= 2;
  }
  if (mem_rows > MAXROWS) return false;
  d.mem_rows = mem_rows;
  d.tap_rows[d.n_taps++] = mem_rows;
  d.tap_rows[d.n_taps++] = Ld;
  for (int l = 0; l < c.d_layers; ++l) d.tap_rows[d.n_taps++] = Ld;
  d.tap_rows[d.n_taps++] = Ld;
  return true;
}

size_t informer_blob_size(const cet_informer_cfg* c) {
  Dims d;
  if (!supported(*c) || !plan(*c, d)) return 0;
  const size_t D = DM, F = c->d_ff;
  const size_t attn = 4 * (D * D + D), ffn = F * D + F + D * F + D;
  size_t n = D * c->enc_in * 3 + D + (size_t)c->seq_len * D;
  n += D * c->dec_in * 3 + D + (size_t)(c->label_len + c->pred_len) * D;
  for (int e = 0; e < d.n_enc; ++e) {
    const size_t el = d.enc_layers[e];
    n += el * (attn + ffn + 4 * D);
    if (c->distil) n += (el - 1) * (3 * D * D + D + 4 * D);
    n += 2 * D;
  }
  n += (size_t)c->d_layers * (2 * attn + ffn + 6 * D);
  n += 2 * D + (size_t)c->c_out * D + c->c_out;
  return n;
}

// [out][in][3] (PyTorch Conv1d) -> [3][out][in]
static void conv_taps_major(const float* src, int cout, int cin, std::vector<float>& dst) {
  for (int j = 0; j < 3; ++j)
    for (int o = 0; o < cout; ++o)
      for (int c = 0; c < cin; ++c) dst.push_back(src[((size_t)o * cin + c) * 3 + j]);
}

int informer_create(const cet_informer_cfg* c, const float* wsrc, size_t n, InformerModel** out) {
  if (!supported(*c)) return CET_ERR_UNSUPPORTED;
  Dims d;
  if (!plan(*c, d)) return CET_ERR_UNSUPPORTED;
  if (n != informer_blob_size(c)) return CET_ERR_BLOB_SIZE;

  // Re-order the caller's (PyTorch-layout) blob into execution order.
  std::vector<float> v;
  v.reserve(n + 4096);
  const float* p = wsrc;
  auto copy = [&](size_t k) { v.insert(v.end(), p, p + k); p += k; };
  const size_t D = DM;
  const int Ld = c->label_len + c->pred_len;
  const int emb_in[2] = {c->enc_in, c->dec_in};
  const int emb_len[2] = {c->seq_len, Ld};
  for (int s = 0; s < 2; ++s) {
    conv_taps_major(p, DM, emb_in[s], v);
    p += D * emb_in[s] * 3;
    copy(D);
    copy((size_t)emb_len[s] * D);
  }
  const size_t layer_sz = 4 * (D * D + D) + 2 * (D * D + D) + 4 * D;
  for (int e = 0; e < d.n_enc; ++e) {
    const int el = d.enc_layers[e];
    const float* layers = p;
    const float* convs = p + layer_sz * el;
    for (int l = 0; l < el; ++l) {
      v.insert(v.end(), layers + layer_sz * l, layers + layer_sz * (l + 1));
      if (c->distil && l < el - 1) {
        const float* cw = convs + (size_t)l * (3 * D * D + 5 * D);
        conv_taps_major(cw, DM, DM, v);
        const float* cb = cw + 3 * D * D;
        v.insert(v.end(), cb, cb + D);  // conv bias
        const float *g = cb + D, *be = cb + 2 * D, *rm = cb + 3 * D, *rv = cb + 4 * D;
        // eval-mode BatchNorm1d as ATen folds it: alpha = w/sqrt(var+eps), beta = b - mean*alpha
        std::vector<float> alpha(D), beta(D);
        for (size_t i = 0; i < D; ++i) {
          alpha[i] = g[i] * (1.0f / sqrtf(rv[i] + 1e-5f));
          beta[i] = be[i] - rm[i] * alpha[i];
        }
        v.insert(v.end(), alpha.begin(), alpha.end());
        v.insert(v.end(), beta.begin(), beta.end());
      }
    }
    p = convs + (c->distil ? (size_t)(el - 1) * (3 * D * D + 5 * D) : 0);
    copy(2 * D);  // encoder norm
  }
  copy((size_t)c->d_layers * (2 * 4 * (D * D + D) + 2 * (D * D + D) + 6 * D));
  copy(2 * D);  // decoder norm
  {
    const int rows = (c->c_out + 15) / 16 * 16;
    std::vector<float> pw((size_t)rows * D, 0.f), pbias(rows, 0.f);
    std::copy(p, p + (size_t)c->c_out * D, pw.begin());
    p += (size_t)c->c_out * D;
    std::copy(p, p + c->c_out, pbias.begin());
    p += c->c_out;
    v.insert(v.end(), pw.begin(), pw.end());
    v.insert(v.end(), pbias.begin(), pbias.end());
  }
  if ((size_t)(p - wsrc) != n) return CET_ERR_BLOB_SIZE;

  InformerModel* m = new (std::nothrow) InformerModel;
  if (!m) return CET_ERR_ALLOC;
  m->cfg = *c;
  m->d = d;
  for (int i = 0; i < d.n_attn; ++i) m->attn_per_sample += (size_t)NHEAD * d.attn_len[i] * d.attn_len[i];
  for (int i = 0; i < d.n_taps; ++i) m->taps_per_sample += (size_t)d.tap_rows[i] * DM;
  hipError_t e = hipMalloc(&m->dev_blob, v.size() * sizeof(float));
  if (e == hipSuccess)
    e = hipMemcpy(m->dev_blob, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (m->dev_blob) (void)hipFree(m->dev_blob);
    delete m;
    return hip_fail(e);
  }
  *out = m;
  return CET_OK;
}

void informer_destroy(InformerModel* m) {
  if (!m) return;
  if (m->dev_blob) (void)hipFree(m->dev_blob);
  delete m;
}

size_t informer_attn_size(const InformerModel* m, int B) { return m->attn_per_sample * (size_t)B; }
size_t informer_taps_size(const InformerModel* m, int B) { return m->taps_per_sample * (size_t)B; }

int informer_forward(const InformerModel* m, const float* x_enc, const float* x_dec, float* out,
                     float* attn, float* taps, int B, hipStream_t stream) {
  KArgs a{};
  a.w = m->dev_blob;
  a.x_enc = x_enc; a.x_dec = x_dec; a.out = out; a.attn = attn; a.taps = taps; a.B = B;
  a.enc_in = m->d.enc_in; a.dec_in = m->d.dec_in; a.c_out = m->d.c_out;
  a.seq_len = m->d.seq_len; a.label_len = m->d.label_len; a.pred_len = m->d.pred_len;
  a.n_enc = m->d.n_enc; a.distil = m->d.distil; a.activation = m->d.activation;
  a.d_layers = m->d.d_layers;
  for (int i = 0; i < 8; ++i) a.enc_layers[i] = m->d.enc_layers[i];
  // LDS: three regions of R rows (Q | K | V; also the conv / max-pool staging, which needs
  // L + (L + 2) rows, and the 32-row staging of move_rows).
  const int Ld = m->d.label_len + m->d.pred_len;
  int R = m->d.seq_len > Ld ? m->d.seq_len : Ld;
  if (R < m->d.mem_rows) R = m->d.mem_rows;
  if (R < 12) R = 12;
  a.region_rows = R;
  const size_t lds = (size_t)3 * R * LDW * sizeof(float);
  static const int wpe = [] {
    const char* e = getenv("CET_INFORMER_WPE");
    return e ? atoi(e) : 2;
  }();
  if (wpe == 1)
    hipLaunchKernelGGL(informer_kernel<1>, dim3((unsigned)B), dim3(64), lds, stream, a);
  else
    hipLaunchKernelGGL(informer_kernel<2>, dim3((unsigned)B), dim3(64), lds, stream, a);
  CET_HIP(hipGetLastError());
  return CET_OK;
}

}  // namespace cet
