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
// LDS only holds what needs a cross-row view: K and V (and, in the one-wave kernel, Q) of the
// attention at hand, and the staging rows of the circular convolutions / max-pool.
// Attention: S^T = K Q^T per head on MFMA (A = K rows, B = Q rows, k = the 8 head features),
// which leaves every query's scores in one lane column: softmax is 8 in-lane values plus two
// cross-lane steps. The normalised probabilities are then the B operand of the P*V product
// as they stand; two heads share one 16-feature output tile by zeroing the other head's
// rows of the V^T operand.
// Two launch shapes of this code (informer_kernel<.., WG>): one wave per workgroup with the
// weights read from global memory, and workgroups of 8 waves (8 sequences) that walk the
// network in step, share every weight block through LDS ("weight staging") and, once the
// sequences are short, carry two sequences in one tile ("two sequences in one tile").
// All arithmetic is fp32 (MFMA f32 = exact fmaf chains). exp is 2^(x log2 e) on the hardware
// exp2 (1 ulp), reciprocals and 1/sqrt are the 1-ulp hardware forms, erf is a 1.5e-7 fit.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "cet_internal.h"

namespace cet {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // for 4-byte aligned vector stores

constexpr int DM = 64;    // d_model
constexpr int HDIM = 8;   // features per head as the kernels lay them out (d_keys <= 8, zero padded)
#ifndef CET_LDW
#define CET_LDW 72
#endif
// LDS row stride in floats. A ds_read_b128 is served in four groups of 16 lanes; lane (c, g) of a
// fragment read addresses row c, floats 4 g .. 4 g + 3, i.e. bank group (stride / 4 * c + g) mod 16:
// with 64 + 8 (18 c + g) every group of 16 lanes covers all 64 banks once, with 64 + 4 (17 c + g)
// two lanes of every group collide.
constexpr int LDW = CET_LDW;
constexpr int MAXT = 3;   // row tiles of 16 an activation can span (sequences of up to 48 rows)
constexpr int MAXQ = 6;   // 16-feature tiles of the widest activation (Q / K / V of 12 heads)
constexpr int MAXROWS = 16 * MAXT;
// An activation in T layout: [row tile][feature tile]. Only the tiles a kernel instantiation
// touches ever exist (the arrays are fully scalarised), so the bounds cost nothing.
typedef f32x4 Act[MAXT][MAXQ];
// Heads come in pairs (one 16-feature tile each): NHP = 4 serves up to 8 heads, NHP = 6 up to 12.
// Row stride of the Q / K / V rows in LDS for NHP head pairs.
template <int NHP> constexpr int ldq() { return 16 * NHP + (LDW - 64); }

struct Dims {
  int enc_in, dec_in, c_out, seq_len, label_len, pred_len;
  int n_enc, enc_layers[8], distil, activation, d_layers;
  int n_attn;          // attention maps produced
  int attn_len[64];    // their sequence lengths
  int n_taps;
  int tap_rows[128];
  int mem_rows;        // rows of the concatenated encoder outputs (cross-attention keys)
};

// One step of the encoder stack, in execution order (built on the host).
enum { OP_LAYER = 0, OP_DISTIL = 1, OP_NORM_APPEND = 2, OP_SLICE = 3 };
struct Op {
  unsigned char type;
  unsigned char L;     // rows on entry
  unsigned char a, b;  // NORM_APPEND: a = row offset in the memory; SLICE: a = first row, b = rows kept
};
constexpr int MAX_OPS = 80;

struct KArgs {
  const float* w;  // device blob, execution order
  const float* x_enc;
  const float* x_dec;
  float* out;
  float* attn;  // nullable
  float* taps;  // nullable
  int B;            // sequences of the call (the strides of attn and taps)
  int first, last;  // sequences [first, last) are this launch's
  int enc_in, dec_in, c_out, seq_len, label_len, pred_len;
  int n_enc, distil, activation, d_layers;
  int enc_layers[8];
  int region_rows;  // rows of each of the three LDS regions (Q | K | V)
  int n_ops, n_ops2, n_ops3;  // encoder steps; the first n_ops2 touch > 16 rows, the first n_ops3 > 32
  int mem_rows;
  Op ops[MAX_OPS];
#ifdef CET_EXP
  int ablate;       // experiment build only (CET_INFORMER_ABLATE): skips parts, wrong results
#endif
  const float* wblk;    // workgroup kernel: staged Linear blocks in consumption order
  int n_blk;
  int wave_lds_floats;  // LDS floats private to a wave
  int n_pack0;          // workgroup kernel: first step that runs on packed tiles (n_ops: none)
  float ln_inv_n, ln_pad;  // LayerNorm: 1 / d_model and the number of zero-padded features
  float attn_c;            // log2(e) / sqrt(d_keys)
  int n_heads;             // heads of the model
  unsigned long long* prof;  // profiling build only
};

#define DEV __device__ __forceinline__

DEV f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
DEV f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
DEV f32x4 splat(float v) { return f32x4{v, v, v, v}; }
DEV f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
DEV void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// Pins a wave-uniform pointer in scalar registers: loads through it then take the
// "scalar base + per-lane 32-bit offset" addressing form and need no 64-bit vector adds.
// The result is typed as a global-memory pointer (the laundering would otherwise demote
// the loads to flat ones).
typedef const __attribute__((address_space(1))) float* gptr;
DEV gptr uniform(const float* p) {
  asm("" : "+s"(p));
  return (gptr)p;
}
DEV f32x4 ld4(gptr p) {
  return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(p);
}
// Attention-map stores go through a buffer descriptor: the lanes that have nothing to store get
// an offset beyond the buffer and the hardware drops their part, so a store is one instruction
// with no exec-mask branch around it and no 64-bit address arithmetic (base and per-head offset
// are scalar). Rows of a map are only 4-byte aligned; buffer stores of 16 bytes take that.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned MAP_DROP = 0x80000000u;  // beyond any map buffer, and + 16 kt + i does not wrap
DEV void bst4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
DEV void bst1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
// The first n (wave-uniform, 0..3) floats of v, as one store of 4, 8 or 12 bytes.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
DEV void bst_n(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int n, f32x4 v) {
  const u32x4 u = __builtin_bit_cast(u32x4, v);
  if (n == 1) bst1(r, voff, soff, v[0]);
  else if (n == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2{u[0], u[1]}, r, voff, soff, 0);
  else if (n == 3) __builtin_amdgcn_raw_buffer_store_b96(u32x3{u[0], u[1], u[2]}, r, voff, soff, 0);
}

// Attention maps of a packed tile (two sequences of pack_L <= 8 rows; amap / amap2: their maps of
// this layer, null where a sequence does not exist): descriptor from the first existing map to
// the end of the second, and this lane's byte offsets (or MAP_DROP): lane (c, g) holds keys
// 4 (g & 1) .. + 3 of row c & 7 of sequence c >> 3, and only in the lane groups of its own sequence.
struct PackedMapLanes {
  unsigned all, lo, hi;  // every lane of a row / the lanes holding keys 0..3 / keys 4..7
};
DEV __amdgpu_buffer_rsrc_t packed_map_rsrc(float* amap, float* amap2, int pack_L, int n_heads) {
  float* const base = amap ? amap : amap2;
  const unsigned delta = (amap && amap2) ? (unsigned)(amap2 - amap) : 0u;
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, base ? (int)((delta + (unsigned)(n_heads * pack_L * pack_L)) * 4u) : 0,
                                           0x00020000);
}
DEV PackedMapLanes packed_map_lanes(const float* amap, const float* amap2, int pack_L, const int c, const int g) {
  const int sq = c >> 3, ql = c & 7;
  const unsigned delta = (amap && amap2) ? (unsigned)(amap2 - amap) : 0u;  // the second sequence's maps lie behind the first's
  const bool row = (g >> 1) == sq && ql < pack_L && (sq ? amap2 != nullptr : amap != nullptr);
  const unsigned off = ((unsigned)(ql * pack_L + 4 * (g & 1)) + (sq ? delta : 0u)) * 4u;
  return PackedMapLanes{row ? off : MAP_DROP, row && !(g & 1) ? off : MAP_DROP, row && (g & 1) ? off : MAP_DROP};
}
// One head's tile v of a packed map: whole rows of 8 as one 16-byte store per lane, shorter rows
// as a store of keys 0..3 (or fewer) and one of keys 4..
DEV void packed_map_put(__amdgpu_buffer_rsrc_t r, const PackedMapLanes ml, int head, int pack_L, f32x4 v) {
  const unsigned hoff = (unsigned)(head * pack_L * pack_L) * 4u;  // wave-uniform: the head's map
  if (pack_L == 8) {
    bst4(r, ml.all, hoff, v);
  } else {
    if (pack_L >= 4) bst4(r, ml.lo, hoff, v);
    else bst_n(r, ml.lo, hoff, pack_L, v);
    bst_n(r, ml.hi, hoff, pack_L - 4, v);
  }
}
// Floats of the LDS area through which the primary wave of a packed tile hands the tiles of its
// attention maps (8 heads x 64 lanes x 4 floats) to its partner wave, which is idle and issues the
// stores in its place (a store instruction costs the issuing wave 70-140 clocks).
constexpr int MAP_STAGE_FLOATS = 8 * 64 * 4;

// Sum / max over the four lane groups g (lanes c, c+16, c+32, c+48), result in all four.
// v_permlane16_swap / v_permlane32_swap exchange 16- and 32-lane halves inside the VALU, so
// the reductions of softmax and LayerNorm never take the LDS round trip of a ds_bpermute.
// With vdst = src = v the two results hold v of (own half, other half) in every lane.
// The results are copied into scalars before the bit cast: a __builtin_bit_cast applied
// directly to an element of the returned vector reads element 0 for both (hipcc, ROCm 7.2).
DEV void swap16(float v, float& lo, float& hi) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  lo = __builtin_bit_cast(float, r0);
  hi = __builtin_bit_cast(float, r1);
}
DEV void swap32(float v, float& lo, float& hi) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  lo = __builtin_bit_cast(float, r0);
  hi = __builtin_bit_cast(float, r1);
}
DEV float group_sum(float v) {
  float a, b;
  swap16(v, a, b);
  swap32(a + b, a, b);
  return a + b;
}
DEV float group_max(float v) {
  float a, b;
  swap16(v, a, b);
  swap32(fmaxf(a, b), a, b);
  return fmaxf(a, b);
}

struct Lane {
  int c;  // lane & 15 : row inside a tile (B operand / C column)
  int g;  // lane >> 4 : k group / C row block
  unsigned wo;  // c * 64 + 4 * g : this lane's offset inside a 16-row block of a [.][64] matrix
  float inv_n, pad_n;  // LayerNorm over the real features: 1 / d_model, 64 - d_model
  float attn_c;        // softmax scale times log2(e)
  int n_heads;         // heads of the model (attention maps exist for these only)
#ifdef CET_EXP
  int ablate;  // experiment build only: mask of the parts to skip
#endif
#ifdef CET_PROF
  unsigned long long* prof;  // this wave's stamp trace, or null when the wave is not sampled
  int* pidx;                 // next free entry of the trace
#endif
};

// Timing experiments that skip parts of the arithmetic exist in the experiment build only
// (make libcet_exp.so EXP="-DCET_EXP"); in the product library the tests fold to false.
#ifdef CET_EXP
#define ABLATED(holder, bit) (((holder).ablate & (bit)) != 0)
#else
#define ABLATED(holder, bit) false
#endif

// Phase stamps of the profiling build (libcet_prof.so): a few sampled waves append
// (clock << 4 | category) to a trace in global memory; the host sums the intervals per
// category. Compiles to nothing in the product library.
#ifdef CET_PROF
constexpr int PROF_SAMPLES = 16, PROF_MAXM = 1024;
#define PROF_MARK(ln, cat)                                                              \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                       \
    if ((ln).prof && (threadIdx.x & 63) == 0 && *(ln).pidx < PROF_MAXM)                 \
      (ln).prof[*(ln).pidx] = (now_ << 4) | (unsigned long long)(cat);                  \
    *(ln).pidx += 1;                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
#else
#define PROF_MARK(ln, cat) do { } while (0)
#endif
enum { P_EMBED = 0, P_QKV, P_ATTN, P_OPROJ, P_ADDLN, P_FFN, P_DISTIL, P_MOVE, P_PROJ, P_OTHER, P_N };

// Orders this wave's LDS writes before its later LDS reads for the compiler. The hardware
// executes one wave's LDS instructions in order, and every LDS region below is private to a
// wave, so no barrier is needed.
DEV void wave_sync() { asm volatile("" ::: "memory"); }

// ---- weight staging of the workgroup kernel ------------------------------------------------
// The 8 waves of a workgroup (8 sequences) walk the network in lockstep. The weights of each
// Linear(64,64) are brought into LDS once per workgroup by LDS-DMA, one product ahead of their
// use, and every wave reads its MFMA A fragments from there: 1/8 of the vector-memory
// instructions of the one-wave kernel, whose address path is the shared resource of a CU.
// Staged block (floats): W[64][LDW] | bias[64] | gamma[64] | beta[64] | pad = 19 KiB.
constexpr int WBLK = (DM * LDW + 3 * DM + 255) / 256 * 256;  // whole 1 KiB DMA pieces
constexpr int WBLK_BIAS = DM * LDW;
constexpr int WBLK_NORM = WBLK_BIAS + DM;
constexpr int WG_WAVES = 8;
struct Stage {
  const float* blocks;  // global memory: the blocks in consumption order
  float* buf;           // LDS: two buffers of WBLK floats
  unsigned lds;         // LDS byte address of buf
  int next, total;      // next block to request; number of blocks
  int wave, lane;
  int pub;              // staging barriers passed = blocks published to this wave
  int use;              // next block this wave consumes
  bool lag;             // this wave runs one staging barrier ahead of its consumption (see stage_lag)
  int dma_first, dma_step;  // this wave requests pieces dma_first, dma_first + dma_step, ... of a block
};
DEV unsigned lds_address(const void* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}
// LDS-DMA of 16 B per lane to lds + 16 * lane. Issued through inline asm so that the compiler
// does not order later LDS reads behind it with vmcnt(0); ordered by stage_acquire's wait.
DEV void dma16(const void* src, unsigned lds_addr_wave_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_addr_wave_uniform)
      : "memory");
}
// Requests the next block into the buffer (next & 1): 18 wave instructions of 1 KiB, spread
// over the 8 waves (over waves 4..7 alone while those have nothing else to do, see stage_share).
DEV void stage_request(Stage& st) {
  if (st.next >= st.total) return;
  const char* src = reinterpret_cast<const char*>(st.blocks + (size_t)st.next * WBLK) + 16 * st.lane;
  const unsigned dst = st.lds + (st.next & 1) * (WBLK * 4);
  for (int i = st.dma_first; i < WBLK * 4 / 1024; i += st.dma_step) dma16(src + i * 1024, dst + i * 1024);
  ++st.next;
}
// Staging barrier number `pub` (all 8 waves pass the same sequence of them): every wave's share
// of block `pub` has landed and every wave is done reading block `pub - 1`, whose buffer the
// request for block `pub + 1` then overwrites. The wait also covers this wave's own LDS reads.
DEV void stage_barrier(Stage& st) {
#ifdef CET_EXP_NOBARRIER  // timing experiment: what the workgroup barrier costs (results invalid)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  __builtin_amdgcn_sched_barrier(0);  // nothing that follows is moved in front of the rendezvous
  ++st.pub;
  stage_request(st);
}
// The next block of the consumption order, once it is published to this wave.
DEV const float* stage_acquire(Stage& st) {
  if (st.pub <= st.use) stage_barrier(st);
  return st.buf + (st.use++ & 1) * WBLK;
}
// Waves 4..7 of a workgroup share their SIMDs with waves 0..3. Left alone, both waves of a SIMD
// would wait at the same barrier, read their weight fragments at the same time (8 x 16 KiB
// through the LDS pipe) and only then start their MFMAs. In lag mode a wave passes the NEXT
// staging barrier right after its fragment reads and before its MFMAs: its matrix and vector work
// on block k then runs while the other wave of the SIMD waits for and reads block k + 1, and the
// other way round. The count of barriers is the same for every wave; only their place in the
// instruction stream differs. No barrier exists beyond the last block.
DEV void stage_lag(Stage& st) {
  if (st.lag && st.pub < st.total) stage_barrier(st);
}
// Who requests the pieces of a block: all 8 waves in turn, or (busy_low) waves 4..7 only.
DEV void stage_share(Stage& st, bool busy_low) {
  if (!busy_low) {
    st.dma_first = st.wave;
    st.dma_step = WG_WAVES;
  } else {
    st.dma_first = st.wave >= WG_WAVES / 2 ? st.wave - WG_WAVES / 2 : WBLK;  // WBLK: none
    st.dma_step = WG_WAVES / 2;
  }
}
// Brings every wave to the same number of passed staging barriers (one ahead of consumption)
// before a barrier of another kind; a wave in lag mode is there already.
DEV void stage_align(Stage& st) {
  while (st.pub <= st.use && st.pub < st.total) stage_barrier(st);
}

// out^T[rt][mo] = W[64 x 64] * in^T  (+ bias). W row-major [out][in], bias[out].
// All 16 weight fragments (and the bias) of the product are requested before the first MFMA:
// the wave then pays one load latency per product, and the remaining fragments land while the
// matrix pipe works (left to itself the compiler, which is register-bound at two waves per
// SIMD, sinks every load next to its first use). Addresses are a uniform base plus one
// per-lane offset, so the loads need no vector address arithmetic.
struct NormParams {
  f32x4 gm[4], bt[4];  // gamma / beta of this lane's 16 features
};
DEV void load_norm(NormParams& n, const float* __restrict__ gamma_beta_, const Lane ln) {
  const gptr gamma_beta = uniform(gamma_beta_);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    n.gm[m] = ld4(gamma_beta + 16 * m + 4 * ln.g);
    n.bt[m] = ld4(gamma_beta + DM + 16 * m + 4 * ln.g);
  }
}

// Weight fragments of one 64x64 product (A operands of all 64 MFMAs per row tile) + bias.
struct WFrag {
  f32x4 a[MAXQ][MAXQ], bv[MAXQ];  // [output tile][input tile]
};
// Requests the fragments of W[64][64] (row-major) and bias[64]. The sched_barrier pins the
// requests where they are written: the kernel's layer schedules issue them one product ahead
// of their use, which the register-bound compiler would otherwise undo.
template <int MO = 4, int KI = 4>  // W[16 MO][16 KI] row-major | bias[16 MO]
DEV void wload(WFrag& f, const float* __restrict__ W, const Lane ln, NormParams* norm = nullptr,
               const float* __restrict__ norm_w = nullptr) {
  constexpr int IN = 16 * KI;
  const gptr bias = uniform(W + 16 * MO * IN);
  const unsigned wo = KI == 4 ? ln.wo : (unsigned)(ln.c * IN + 4 * ln.g);
#pragma unroll
  for (int mo = 0; mo < MO; ++mo) {
    const gptr Wm = uniform(W + 16 * mo * IN);
#pragma unroll
    for (int mi = 0; mi < KI; ++mi) f.a[mo][mi] = ld4(Wm + wo + 16 * mi);
    f.bv[mo] = ld4(bias + 16 * mo + 4 * ln.g);
  }
  if (norm) load_norm(*norm, norm_w, ln);
  __builtin_amdgcn_sched_barrier(0);
}
// The same from a staged block in LDS (row stride LDW: conflict-free fragment reads).
// Reads are issued in the order the MFMAs consume them (wmul runs its four output tiles as
// interleaved chains: bias, then input tile 0 of every output tile, then input tile 1, ...), so
// the matrix work starts when the first eight reads have landed, not all twenty.
DEV void wload_lds(WFrag& f, const float* blk, const Lane ln, NormParams* norm = nullptr) {
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) f.bv[mo] = ld4(blk + WBLK_BIAS + 16 * mo + 4 * ln.g);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) f.a[mo][mi] = ld4(blk + (16 * mo + ln.c) * LDW + 16 * mi + 4 * ln.g);
  if (norm) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      norm->gm[m] = ld4(blk + WBLK_NORM + 16 * m + 4 * ln.g);
      norm->bt[m] = ld4(blk + WBLK_NORM + DM + 16 * m + 4 * ln.g);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}
// Fragments of the next Linear of the layer: from global memory (one-wave kernel) or from the
// next staged block (workgroup kernel; the global pointers are then unused).
template <bool WG, int MO = 4, int KI = 4>
DEV void next_weights(WFrag& f, Stage& st, const float* __restrict__ W, const Lane ln,
                      NormParams* norm = nullptr, const float* __restrict__ norm_w = nullptr) {
  if (WG) {
    static_assert(!WG || (MO == 4 && KI == 4), "staged blocks are 64 x 64");
#ifdef CET_PROF_FINE  // profiling build: the wait for the block and the fragment reads as phases of their own
    PROF_MARK(ln, P_PROJ);
    const float* blk_ = stage_acquire(st);
    PROF_MARK(ln, P_OTHER);
    wload_lds(f, blk_, ln, norm);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PROF_MARK(ln, P_MOVE);
    stage_lag(st);
#else
    wload_lds(f, stage_acquire(st), ln, norm);
    stage_lag(st);
#endif
  } else {
    wload<MO, KI>(f, W, ln, norm, norm_w);
  }
}
// out^T[rt][mo] = W * in^T + bias
template <int RT, int MO = 4, int KI = 4>
DEV void wmul(const WFrag& f, const Act& in, Act& out) {
#pragma unroll
  for (int mo = 0; mo < MO; ++mo) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 acc = f.bv[mo];
#pragma unroll
      for (int mi = 0; mi < KI; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = mfma(f.a[mo][mi][i], in[rt][mi][i], acc);
      out[rt][mo] = acc;
    }
  }
}

// Rows [0, nrows) of the tiles go to LDS; rows of the padded tiles beyond that are dropped
// (the regions are sized to the sequence, not to whole tiles).
template <int RT, int NT = 4, int LD = LDW>  // NT feature tiles per row, row stride LD floats
DEV void store_rows(float* buf, const Lane ln, const Act& v, int nrows) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if (16 * rt + ln.c < nrows) {
#pragma unroll
      for (int m = 0; m < NT; ++m) st4(buf + (16 * rt + ln.c) * LD + 16 * m + 4 * ln.g, v[rt][m]);
    }
  }
}

// torch.nn.LayerNorm(64), eps 1e-5, over the feature axis (in-lane 16 values + lane groups).
template <int RT>
DEV void layer_norm(Act& x, const NormParams& n, const Lane ln) {
  if (ABLATED(ln, 16)) return;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    // sums are kept four wide (packed fp32 adds / FMAs) and folded at the end
    const f32x4 s4 = (x[rt][0] + x[rt][1]) + (x[rt][2] + x[rt][3]);
    const float mean = group_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * ln.inv_n;
    f32x4 q4 = zero4();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 d = x[rt][m] - mean;
      q4 = __builtin_elementwise_fma(d, d, q4);
    }
    // zero-padded features each contributed mean^2 to the sum of squares
    const float var = fmaf(-ln.pad_n, mean * mean, group_sum((q4[0] + q4[1]) + (q4[2] + q4[3]))) * ln.inv_n;
    const float inv = __builtin_amdgcn_rsqf(var + 1e-5f);
    const float shift = -mean * inv;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 xn = x[rt][m] * inv + shift;  // (x - mean) / sqrt(var + eps)
      x[rt][m] = __builtin_elementwise_fma(xn, n.gm[m], n.bt[m]);
    }
  }
}
template <int RT>
DEV void layer_norm(Act& x, const float* __restrict__ gamma,
                    const float* __restrict__ beta, const Lane ln) {
  (void)beta;  // stored right behind gamma
  NormParams n;
  load_norm(n, gamma, ln);
  layer_norm<RT>(x, n, ln);
}

// relu, or gelu(v) = v * Phi(v) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7):
// 1 - erf(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p z), z >= 0.
// Written as max(v, 0) - |v| / 2 * (1 - erf(|v| / sqrt 2)), which is branch-free and has no
// cancellation in the negative tail.
DEV f32x4 act(f32x4 v, int activation) {
  const f32x4 pos = __builtin_elementwise_max(v, zero4());
  if (activation == 0) return pos;
  const f32x4 av = __builtin_elementwise_abs(v);
  const f32x4 d = __builtin_elementwise_fma(av, splat(0.3275911f * 0.70710678118654752440f), splat(1.0f));
  f32x4 t, e;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_rcpf(d[i]);
  f32x4 q = __builtin_elementwise_fma(splat(1.061405429f), t, splat(-1.453152027f));
  q = __builtin_elementwise_fma(q, t, splat(1.421413741f));
  q = __builtin_elementwise_fma(q, t, splat(-0.284496736f));
  q = __builtin_elementwise_fma(q, t, splat(0.254829592f));
  const f32x4 z = av * av * (-0.5f * 1.44269504088896340736f);
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(z[i]);
  return __builtin_elementwise_fma(av * -0.5f, q * t * e, pos);
}

template <int RT>
DEV void add_into(Act& x, const Act& y) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int m = 0; m < 4; ++m) x[rt][m] += y[rt][m];
}

// Copies the first `rows` rows to the debug buffer, stage-major [stage][B][rows][64].
template <int RT>
DEV void tap(float*& taps, int B, int b, int rows, const Lane ln, const Act& x) {
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
// Heads are processed NP pairs at a time, each step for all of them before the next step:
// scores (MFMA), softmax (VALU, cross-lane), P*V (MFMA). The steps of different heads are
// independent, so their latencies overlap instead of forming one long dependent chain.
// QREG: the queries are not in LDS but in registers (q, T layout), and the Q and K projections
// were taken with row-permuted weights (qk_row() on the host) so that position 16 t + 4 g + i of
// their outputs holds feature 16 t + 8 (i >> 1) + 2 g + (i & 1): register i of tile t then spans,
// over the four lane groups g (the k index of the MFMA), features of ONE head, and a head's
// scores are two MFMAs of k = 4 (registers 2 hh and 2 hh + 1) with the query registers as B
// operands as they stand and the keys read as the matching 16-byte groups.
// pack_L > 0 (single tiles only): the tile holds two sequences of pack_L <= 8 rows, at rows
// 0.. and 8..; a query then only admits the keys of its own sequence, and amap / amap2 are the
// maps of the first / second sequence (Lq = Lk = 16 for the row bookkeeping).
// HPT = 3 (models of 9 to 12 heads with at most 5 features per head, QREG only): a tile holds
// THREE heads, so that 12 heads fill four tiles like 8 heads of 8 features do, not six. Head j of
// a tile has features 0..3 in register j of the query / key tile (one per lane group) and
// feature 4 in register 3 of lane group j (qk3_pos()): its scores are register j and register 3
// with the keys of the other lane groups zeroed, two MFMAs as before. Values and context have the
// head's features at positions 5 j .. 5 j + 4 of the tile (position 15 is zero).
template <int RQ, int RK, bool QREG, int NHP = 4, int HPT = 2>
DEV void attention(const Act& q, const float* Qb, const float* Kb, const float* Vb, int Lq,
                   int Lk, bool causal, const Lane ln, Act& ctx, float* amap,
                   int pack_L = 0, float* amap2 = nullptr, float* map_stage = nullptr) {
  static_assert(HPT == 2 || (HPT == 3 && QREG), "three heads per tile: queries in registers only");
  constexpr int LDQ = ldq<NHP>();
  // tiles per group (a divisor of NHP)
  constexpr int NP = HPT == 3 ? (RQ * RK == 1 ? 2 : 1)
                              : ((RQ * RK == 1) ? (NHP % 4 == 0 ? 4 : 2) : (RQ * RK == 2 ? 2 : 1));
  // softmax(scale * s) is evaluated as 2^(c s + bias - max) / sum with c = scale * log2(e);
  // bias is 0 for admitted keys and -inf for padded or (causal) future ones.
  const float c = ln.attn_c;  // float(1 / sqrt(d_keys)) * log2 e
  const bool upper = (ln.c & 8) != 0;  // lane feeds the second head of a pair
  f32x4 bias[RK][RQ];
#pragma unroll
  for (int kt = 0; kt < RK; ++kt)
#pragma unroll
    for (int qt = 0; qt < RQ; ++qt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = 16 * kt + 4 * ln.g + i;
        bool ok = key < Lk && (!causal || key <= 16 * qt + ln.c);
        if (pack_L) ok = (key >> 3) == (ln.c >> 3) && (key & 7) < pack_L;
        bias[kt][qt][i] = ok ? 0.f : -INFINITY;
      }
  // Attention-map stores, prepared once per call: a descriptor over this sequence's maps of the
  // layer and, per query tile, this lane's byte offset inside a head's [Lq][Lk] map (its query
  // row, key 4 g of key tile 0), or MAP_DROP where the lane holds nothing of the map.
  float* const mapb = amap ? amap : amap2;
  const int map_hl = pack_L ? pack_L * pack_L : Lq * Lk;  // floats of one head's map
  const __amdgpu_buffer_rsrc_t mrsrc =
      pack_L ? packed_map_rsrc(amap, amap2, pack_L, ln.n_heads)
             : __builtin_amdgcn_make_buffer_rsrc(mapb, 0, mapb ? ln.n_heads * map_hl * 4 : 0, 0x00020000);
  unsigned mfull[RQ][RK];  // 16-byte stores of whole key groups
  unsigned mtail[RQ];      // the ragged end of a row (Lk & 3 floats of one key group)
  PackedMapLanes mpk{MAP_DROP, MAP_DROP, MAP_DROP};
  const int tail_n = Lk & 3, tail_k = Lk & ~3;
  if (mapb) {
    if (pack_L) {
      if (!map_stage) mpk = packed_map_lanes(amap, amap2, pack_L, ln.c, ln.g);
    } else {
#pragma unroll
      for (int qt = 0; qt < RQ; ++qt) {
        const int query = 16 * qt + ln.c;
        const unsigned off = (unsigned)(query * Lk + 4 * ln.g) * 4u;
#pragma unroll
        for (int kt = 0; kt < RK; ++kt)
          mfull[qt][kt] = (query < Lq && 16 * kt + 4 * ln.g + 3 < Lk) ? off + 64u * kt : MAP_DROP;
        mtail[qt] = (query < Lq && tail_n && 4 * ln.g == (tail_k & 15)) ? off + 64u * (tail_k >> 4) : MAP_DROP;
      }
    }
  }
#pragma unroll
  for (int hp0 = 0; hp0 < NHP; hp0 += NP) {
    f32x4 p[NP][HPT][RK][RQ];  // [tile][head of the tile][key tile][query tile]
    // ---- scores^T = K Q^T: lane (query 16*qt + c) holds keys 16*kt + 4*g + i ----
    if (QREG) {
      f32x4 kf[NP][RK];
#pragma unroll
      for (int pi = 0; pi < NP; ++pi)
#pragma unroll
        for (int kt = 0; kt < RK; ++kt)
          kf[pi][kt] = ld4(Kb + (16 * kt + ln.c) * LDQ + 16 * (hp0 + pi) + 4 * ln.g);
      // registers 2 hh and 2 hh + 1 of a tile hold, over the four lane groups, the 8 features
      // of head hh of the pair (the staged Q / K weights are row-permuted for this, qk_row())
#pragma unroll
      for (int pi = 0; pi < NP; ++pi)
#pragma unroll
        for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
          for (int kt = 0; kt < RK; ++kt) {
            // three heads: the head's fifth feature sits in register 3 of lane group hh only
            const float k3 = (HPT == 3 && ln.g == hh) ? kf[pi][kt][3] : 0.f;
#pragma unroll
            for (int qt = 0; qt < RQ; ++qt) {
              f32x4 sc = zero4();
              if (HPT == 3) {
                sc = mfma(kf[pi][kt][hh], q[qt][hp0 + pi][hh], sc);
                sc = mfma(k3, q[qt][hp0 + pi][3], sc);
              } else {
                sc = mfma(kf[pi][kt][2 * hh], q[qt][hp0 + pi][2 * hh], sc);
                sc = mfma(kf[pi][kt][2 * hh + 1], q[qt][hp0 + pi][2 * hh + 1], sc);
              }
              p[pi][hh][kt][qt] = sc;
            }
          }
    } else {
      float2 kf[NP][2][RK], qf[NP][2][RQ];
#pragma unroll
      for (int pi = 0; pi < NP; ++pi)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int col = 8 * (2 * (hp0 + pi) + hh) + 2 * ln.g;  // this lane's two head features
#pragma unroll
          for (int kt = 0; kt < RK; ++kt)
            kf[pi][hh][kt] = *reinterpret_cast<const float2*>(Kb + (16 * kt + ln.c) * LDQ + col);
#pragma unroll
          for (int qt = 0; qt < RQ; ++qt)
            qf[pi][hh][qt] = *reinterpret_cast<const float2*>(Qb + (16 * qt + ln.c) * LDQ + col);
        }
#pragma unroll
      for (int pi = 0; pi < NP; ++pi)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int kt = 0; kt < RK; ++kt)
#pragma unroll
            for (int qt = 0; qt < RQ; ++qt) {
              f32x4 sc = zero4();
              sc = mfma(kf[pi][hh][kt].x, qf[pi][hh][qt].x, sc);
              sc = mfma(kf[pi][hh][kt].y, qf[pi][hh][qt].y, sc);
              p[pi][hh][kt][qt] = sc;
            }
    }
    // ---- masked, scaled softmax over the keys of each query ----
    float mx[NP][HPT][RQ], sum[NP][HPT][RQ];
#pragma unroll
    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
        for (int qt = 0; qt < RQ; ++qt) {
          float m = -INFINITY;
#pragma unroll
          for (int kt = 0; kt < RK; ++kt) {
            const f32x4 t = __builtin_elementwise_fma(p[pi][hh][kt][qt], splat(c), bias[kt][qt]);
            p[pi][hh][kt][qt] = t;
            m = fmaxf(fmaxf(m, fmaxf(t[0], t[1])), fmaxf(t[2], t[3]));
          }
          mx[pi][hh][qt] = m;
        }
#pragma unroll
    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
        for (int qt = 0; qt < RQ; ++qt) mx[pi][hh][qt] = group_max(mx[pi][hh][qt]);
#pragma unroll
    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
        for (int qt = 0; qt < RQ; ++qt) {
          f32x4 sm = zero4();
#pragma unroll
          for (int kt = 0; kt < RK; ++kt) {
            const f32x4 d = p[pi][hh][kt][qt] - mx[pi][hh][qt];
            f32x4 e;
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = ABLATED(ln, 4) ? d[i] : __builtin_amdgcn_exp2f(d[i]);
            p[pi][hh][kt][qt] = e;
            sm += e;
          }
          sum[pi][hh][qt] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        }
#pragma unroll
    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
        for (int qt = 0; qt < RQ; ++qt)
          sum[pi][hh][qt] = __builtin_amdgcn_rcpf(group_sum(sum[pi][hh][qt]));
#pragma unroll
    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
        for (int qt = 0; qt < RQ; ++qt) {
#pragma unroll
          for (int kt = 0; kt < RK; ++kt) p[pi][hh][kt][qt] *= sum[pi][hh][qt];
          const int head = HPT * (hp0 + pi) + hh;
          if (mapb && head < ln.n_heads) {
            const unsigned hoff = (unsigned)(head * map_hl) * 4u;  // wave-uniform: the head's map
            if (pack_L) {
              // the tile goes to the partner wave, which stores it (partner_write_maps()), or out
              if (map_stage) st4(map_stage + (head * 64 + 16 * ln.g + ln.c) * 4, p[pi][hh][0][qt]);
              else packed_map_put(mrsrc, mpk, head, pack_L, p[pi][hh][0][qt]);
            } else {
              // a lane's four consecutive keys go out as one 16-byte store, the ragged end of a
              // row as single floats
#pragma unroll
              for (int kt = 0; kt < RK; ++kt) bst4(mrsrc, mfull[qt][kt], hoff, p[pi][hh][kt][qt]);
              if (tail_n) {
#pragma unroll
                for (int kt = 0; kt < RK; ++kt)
                  if ((tail_k >> 4) == kt) bst_n(mrsrc, mtail[qt], hoff, tail_n, p[pi][hh][kt][qt]);
              }
            }
          }
        }
    // ---- context^T tile (features 16*hp .. 16*hp+15): A = V^T with the other head's rows
    // zeroed, so both heads of a pair accumulate into the same output tile ----
    float vf[NP][RK][4];
#pragma unroll
    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
      for (int kt = 0; kt < RK; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int vr = 16 * kt + 4 * ln.g + i;
          vr = vr < Lk ? vr : Lk - 1;  // padded keys carry probability 0; stay inside the region
          vf[pi][kt][i] = Vb[vr * LDQ + 16 * (hp0 + pi) + ln.c];
        }
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      f32x4 acc[RQ];
#pragma unroll
      for (int qt = 0; qt < RQ; ++qt) acc[qt] = zero4();
#pragma unroll
      for (int kt = 0; kt < RK; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (HPT == 3) {
            const int hsel = (ln.c * 13) >> 6;  // c / 5 for c < 16: the head this lane's feature belongs to
#pragma unroll
            for (int hh = 0; hh < HPT; ++hh) {
              const float vh = hsel == hh ? vf[pi][kt][i] : 0.f;
#pragma unroll
              for (int qt = 0; qt < RQ; ++qt) acc[qt] = mfma(vh, p[pi][hh][kt][qt][i], acc[qt]);
            }
          } else {
            const float v0 = upper ? 0.f : vf[pi][kt][i];
            const float v1 = upper ? vf[pi][kt][i] : 0.f;
#pragma unroll
            for (int qt = 0; qt < RQ; ++qt) {
              acc[qt] = mfma(v0, p[pi][0][kt][qt][i], acc[qt]);
              acc[qt] = mfma(v1, p[pi][1][kt][qt][i], acc[qt]);
            }
          }
        }
#pragma unroll
      for (int qt = 0; qt < RQ; ++qt) ctx[qt][hp0 + pi] = acc[qt];
    }
  }
}

// Blob block sizes (floats), execution order.
constexpr int SZ_LIN = DM * DM + DM;  // one Linear(64,64): weight | bias
constexpr int SZ_FFN = 2 * SZ_LIN;
// Q / K / V projection 64 -> 16 NHP (weight | bias) and output projection 16 NHP -> 64
template <int NHP> constexpr int sz_qkv() { return 16 * NHP * DM + 16 * NHP; }
template <int NHP> constexpr int sz_attn() { return 3 * sz_qkv<NHP>() + DM * 16 * NHP + DM; }
#ifndef CET_WPS
#define CET_WPS 2  // waves per SIMD the register budget is set for
#endif

template <int RT>
DEV void activate(Act& h, int activation) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int m = 0; m < 4; ++m) h[rt][m] = act(h[rt][m], activation);
}

// LDS regions of a wave: Q | K | V (R rows each), or K | V where the queries stay in registers
// (QR: the workgroup kernel and the large one-wave shapes). A padded tile read may run into the
// next region (finite data).
template <bool QR, int NHP>
DEV float* region_q(float* smem, int R) { return smem; }
template <bool QR, int NHP>
DEV float* region_k(float* smem, int R) { return QR ? smem : smem + R * ldq<NHP>(); }
template <bool QR, int NHP>
DEV float* region_v(float* smem, int R) { return QR ? smem + R * ldq<NHP>() : smem + 2 * R * ldq<NHP>(); }

// The three projections of an attention layer: K and V go to LDS, Q to LDS or (QR) stays in
// registers.
template <int RQ, int RK, bool WG, bool QR, int NHP>
DEV void project_qkv(const float* __restrict__ w, Stage& st, float* smem, int R, const Lane ln,
                     const Act& xq, const Act& xkv, int Lq, int Lk,
                     Act& q) {
  constexpr int LDQ = ldq<NHP>();
  WFrag A;
  Act t;
  wave_sync();  // earlier reads of the regions are issued
  next_weights<WG, NHP, 4>(A, st, w, ln);
  wmul<RQ, NHP, 4>(A, xq, q);
  if (!QR) store_rows<RQ, NHP, LDQ>(region_q<QR, NHP>(smem, R), ln, q, Lq);
  next_weights<WG, NHP, 4>(A, st, w + sz_qkv<NHP>(), ln);
  wmul<RK, NHP, 4>(A, xkv, t);
  store_rows<RK, NHP, LDQ>(region_k<QR, NHP>(smem, R), ln, t, Lk);
  next_weights<WG, NHP, 4>(A, st, w + 2 * sz_qkv<NHP>(), ln);
  wmul<RK, NHP, 4>(A, xkv, t);
  store_rows<RK, NHP, LDQ>(region_v<QR, NHP>(smem, R), ln, t, Lk);
  wave_sync();
  PROF_MARK(ln, P_QKV);
}

// EncoderLayer.forward (encoder.py:37-53) with AttentionLayer / FullAttention (attn.py).
// Block: q,k,v,o Linear | conv1 | conv2 | norm1 g,b | norm2 g,b.
template <int NHP> constexpr int sz_enc_layer() { return sz_attn<NHP>() + SZ_FFN + 4 * DM; }
template <int RT, bool WG, bool QR, int NHP>
DEV void encoder_layer(const float* __restrict__ w, Stage& st, float* smem, int R, const Lane ln,
                       int L, int activation, float* amap, Act& x, int pack_L = 0,
                       float* amap2 = nullptr, float* map_stage = nullptr) {
  constexpr int SZ_ATTN = sz_attn<NHP>();
  const float* nw = w + SZ_ATTN + SZ_FFN;
  WFrag A;
  NormParams n;
  Act t, y;
  project_qkv<RT, RT, WG, QR, NHP>(w, st, smem, R, ln, x, x, L, L, t);
  attention<RT, RT, QR, NHP>(t, region_q<QR, NHP>(smem, R), region_k<QR, NHP>(smem, R),
                             region_v<QR, NHP>(smem, R), L, L, false, ln, y, amap, pack_L, amap2, map_stage);
  PROF_MARK(ln, P_ATTN);
  next_weights<WG, 4, NHP>(A, st, w + 3 * sz_qkv<NHP>(), ln, &n, nw);  // output projection, norm1
  wmul<RT, 4, NHP>(A, y, t);
  PROF_MARK(ln, P_OPROJ);
  add_into<RT>(x, t);
  layer_norm<RT>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
  next_weights<WG>(A, st, w + SZ_ATTN, ln);  // conv1
  wmul<RT>(A, x, t);
  activate<RT>(t, activation);
  next_weights<WG>(A, st, w + SZ_ATTN + SZ_LIN, ln, &n, nw + 2 * DM);  // conv2, norm2
  wmul<RT>(A, t, y);
  PROF_MARK(ln, P_FFN);
  add_into<RT>(x, y);
  layer_norm<RT>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
}

// DecoderLayer.forward (decoder.py:20-36).
// Block: self q,k,v,o | cross q,k,v,o | conv1 | conv2 | norm1 | norm2 | norm3.
template <int NHP> constexpr int sz_dec_layer() { return 2 * sz_attn<NHP>() + SZ_FFN + 6 * DM; }
template <int RD, int RM, bool WG, bool QR, int NHP>
DEV void decoder_layer(const float* __restrict__ w, Stage& st, float* smem, int R, const Lane ln,
                       int Ld, int Lm, int activation, const Act& mem, Act& x) {
  constexpr int SZ_ATTN = sz_attn<NHP>();
  float* Qb = region_q<QR, NHP>(smem, R);
  float* Kb = region_k<QR, NHP>(smem, R);
  float* Vb = region_v<QR, NHP>(smem, R);
  const float* wc = w + SZ_ATTN;       // cross attention
  const float* wf = w + 2 * SZ_ATTN;   // conv1 | conv2
  const float* nw = wf + SZ_FFN;
  WFrag A;
  NormParams n;
  Act t, y;
  // causal self attention
  project_qkv<RD, RD, WG, QR, NHP>(w, st, smem, R, ln, x, x, Ld, Ld, t);
  attention<RD, RD, QR, NHP>(t, Qb, Kb, Vb, Ld, Ld, true, ln, y, nullptr);
  PROF_MARK(ln, P_ATTN);
  next_weights<WG, 4, NHP>(A, st, w + 3 * sz_qkv<NHP>(), ln, &n, nw);  // output projection, norm1
  wmul<RD, 4, NHP>(A, y, t);
  PROF_MARK(ln, P_OPROJ);
  add_into<RD>(x, t);
  layer_norm<RD>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
  // cross attention: queries from x, keys / values from the encoder memory
  project_qkv<RD, RM, WG, QR, NHP>(wc, st, smem, R, ln, x, mem, Ld, Lm, t);
  attention<RD, RM, QR, NHP>(t, Qb, Kb, Vb, Ld, Lm, false, ln, y, nullptr);
  PROF_MARK(ln, P_ATTN);
  next_weights<WG, 4, NHP>(A, st, wc + 3 * sz_qkv<NHP>(), ln, &n, nw + 2 * DM);  // output projection, norm2
  wmul<RD, 4, NHP>(A, y, t);
  PROF_MARK(ln, P_OPROJ);
  add_into<RD>(x, t);
  layer_norm<RD>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
  next_weights<WG>(A, st, wf, ln);  // conv1
  wmul<RD>(A, x, t);
  activate<RD>(t, activation);
  next_weights<WG>(A, st, wf + SZ_LIN, ln, &n, nw + 4 * DM);  // conv2, norm3
  wmul<RD>(A, t, y);
  PROF_MARK(ln, P_FFN);
  add_into<RD>(x, y);
  layer_norm<RD>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
}

// The two layer forms once more for the large one-wave shapes, which run one wave per SIMD with
// the register budget that goes with it and have no second wave to cover a load: the weight
// fragments of every product are requested one product ahead, into two fragment sets used in
// turn (A, B), so that they travel from L2 during the MFMAs of the product before. Same products,
// same operand order, same results as encoder_layer / decoder_layer.
template <int RT, int NHP, int HPT>
DEV void encoder_layer_ahead(const float* __restrict__ w, float* smem, int R, const Lane ln, int L,
                             int activation, float* amap, Act& x) {
  constexpr int SZ_ATTN = sz_attn<NHP>(), SZ_QKV = sz_qkv<NHP>(), LDQ = ldq<NHP>();
  const float* nw = w + SZ_ATTN + SZ_FFN;
  WFrag A, B;
  NormParams n;
  Act q, t, y;
  wave_sync();  // earlier reads of the regions are issued
  wload<NHP, 4>(A, w, ln);                         // query projection
  wload<NHP, 4>(B, w + SZ_QKV, ln);                // key projection
  wmul<RT, NHP, 4>(A, x, q);
  wload<NHP, 4>(A, w + 2 * SZ_QKV, ln);            // value projection
  wmul<RT, NHP, 4>(B, x, t);
  store_rows<RT, NHP, LDQ>(region_k<true, NHP>(smem, R), ln, t, L);
  wmul<RT, NHP, 4>(A, x, t);
  store_rows<RT, NHP, LDQ>(region_v<true, NHP>(smem, R), ln, t, L);
  wave_sync();
  PROF_MARK(ln, P_QKV);
  // no fragment set is held across the attention, which needs every register itself
  attention<RT, RT, true, NHP, HPT>(q, nullptr, region_k<true, NHP>(smem, R), region_v<true, NHP>(smem, R), L,
                                    L, false, ln, y, amap);
  PROF_MARK(ln, P_ATTN);
  wload<4, NHP>(B, w + 3 * SZ_QKV, ln);            // output projection
  wload<4, 4>(A, w + SZ_ATTN, ln);                 // conv1
  load_norm(n, nw, ln);
  wmul<RT, 4, NHP>(B, y, t);
  PROF_MARK(ln, P_OPROJ);
  add_into<RT>(x, t);
  wload<4, 4>(B, w + SZ_ATTN + SZ_LIN, ln);        // conv2
  layer_norm<RT>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
  wmul<RT>(A, x, t);
  activate<RT>(t, activation);
  load_norm(n, nw + 2 * DM, ln);
  wmul<RT>(B, t, y);
  PROF_MARK(ln, P_FFN);
  add_into<RT>(x, y);
  layer_norm<RT>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
}

template <int RD, int RM, int NHP, int HPT>
DEV void decoder_layer_ahead(const float* __restrict__ w, float* smem, int R, const Lane ln, int Ld,
                             int Lm, int activation, const Act& mem, Act& x) {
  constexpr int SZ_ATTN = sz_attn<NHP>(), SZ_QKV = sz_qkv<NHP>(), LDQ = ldq<NHP>();
  float* Kb = region_k<true, NHP>(smem, R);
  float* Vb = region_v<true, NHP>(smem, R);
  const float* wc = w + SZ_ATTN;       // cross attention
  const float* wf = w + 2 * SZ_ATTN;   // conv1 | conv2
  const float* nw = wf + SZ_FFN;
  WFrag A, B;
  NormParams n;
  Act q, t, y;
  // causal self attention
  wave_sync();
  wload<NHP, 4>(A, w, ln);
  wload<NHP, 4>(B, w + SZ_QKV, ln);
  wmul<RD, NHP, 4>(A, x, q);
  wload<NHP, 4>(A, w + 2 * SZ_QKV, ln);
  wmul<RD, NHP, 4>(B, x, t);
  store_rows<RD, NHP, LDQ>(Kb, ln, t, Ld);
  wmul<RD, NHP, 4>(A, x, t);
  store_rows<RD, NHP, LDQ>(Vb, ln, t, Ld);
  wave_sync();
  PROF_MARK(ln, P_QKV);
  attention<RD, RD, true, NHP, HPT>(q, nullptr, Kb, Vb, Ld, Ld, true, ln, y, nullptr);
  PROF_MARK(ln, P_ATTN);
  wload<4, NHP>(B, w + 3 * SZ_QKV, ln);
  wload<NHP, 4>(A, wc, ln);                        // cross query projection
  load_norm(n, nw, ln);
  wmul<RD, 4, NHP>(B, y, t);
  PROF_MARK(ln, P_OPROJ);
  add_into<RD>(x, t);
  wload<NHP, 4>(B, wc + SZ_QKV, ln);               // cross key projection
  layer_norm<RD>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
  // cross attention: queries from x, keys / values from the encoder memory
  wave_sync();
  wmul<RD, NHP, 4>(A, x, q);
  wload<NHP, 4>(A, wc + 2 * SZ_QKV, ln);           // cross value projection
  wmul<RM, NHP, 4>(B, mem, t);
  store_rows<RM, NHP, LDQ>(Kb, ln, t, Lm);
  wmul<RM, NHP, 4>(A, mem, t);
  store_rows<RM, NHP, LDQ>(Vb, ln, t, Lm);
  wave_sync();
  PROF_MARK(ln, P_QKV);
  attention<RD, RM, true, NHP, HPT>(q, nullptr, Kb, Vb, Ld, Lm, false, ln, y, nullptr);
  PROF_MARK(ln, P_ATTN);
  wload<4, NHP>(B, wc + 3 * SZ_QKV, ln);           // cross output projection
  wload<4, 4>(A, wf, ln);                          // conv1
  load_norm(n, nw + 2 * DM, ln);
  wmul<RD, 4, NHP>(B, y, t);
  PROF_MARK(ln, P_OPROJ);
  add_into<RD>(x, t);
  wload<4, 4>(B, wf + SZ_LIN, ln);                 // conv2
  layer_norm<RD>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
  wmul<RD>(A, x, t);
  activate<RD>(t, activation);
  load_norm(n, nw + 4 * DM, ln);
  wmul<RD>(B, t, y);
  PROF_MARK(ln, P_FFN);
  add_into<RD>(x, y);
  layer_norm<RD>(x, n, ln);
  PROF_MARK(ln, P_ADDLN);
}

// ConvLayer.forward (encoder.py:17-23): Conv1d(64->64, k=3, circular pad 2) -> BatchNorm1d
// (eval; folded to alpha/beta) -> ELU -> MaxPool1d(3, 2, 1). L rows in, returns rows out.
// Block: w[3][64][64] (tap-major) | bias | alpha | beta.
constexpr int SZ_DISTIL = 3 * DM * DM + 3 * DM;
// Workgroup kernel: the three taps are three staged blocks (tap 0 carries the conv bias and,
// in its gamma | beta slots, alpha | beta); each is one accumulating product over the rows
// shifted by that tap.
template <int RT, bool WG>  // tiles covering L + 2 conv outputs
DEV int distil(const float* __restrict__ w, Stage& st, float* smem, int R, const Lane ln, int L,
               Act& x) {
  float* Xb = smem;            // L input rows
  float* Yb = smem + R * LDW;  // L + 2 activated conv outputs (R + L + 2 rows in all)
  wave_sync();
  store_rows<RT>(Xb, ln, x, L);
  wave_sync();
  const int Lc = L + 2;
  Act y;
  if (WG) {
    NormParams ab;  // gm = alpha, bt = beta
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      WFrag A;
      wload_lds(A, stage_acquire(st), ln, j == 0 ? &ab : nullptr);
      stage_lag(st);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        int t = 16 * rt + ln.c;
        t = t < Lc ? t : Lc - 1;
        int src = t + j - 2;  // circular padding of 2 on both sides
        src = src < 0 ? src + L : src;
        src = src >= L ? src - L : src;
        f32x4 bf[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) bf[mi] = ld4(Xb + src * LDW + 16 * mi + 4 * ln.g);
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
          f32x4 acc = j == 0 ? A.bv[mo] : y[rt][mo];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = mfma(A.a[mo][mi][i], bf[mi][i], acc);
          y[rt][mo] = acc;
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = y[rt][mo][i] * ab.gm[mo][i] + ab.bt[mo][i];
          y[rt][mo][i] = v > 0.f ? v : (__expf(v) - 1.0f);
        }
  } else {
  // Weight fragments of one 16-feature output block (3 taps x 4 k-blocks) plus its bias and
  // folded BatchNorm vectors; requested one block ahead of the MFMAs that use them.
  struct Block {
    f32x4 a[3][4], bias, al, be;
  };
  auto load_block = [&](Block& blk, int mo) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const gptr Wm = uniform(w + (j * DM + 16 * mo) * DM);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) blk.a[j][mi] = ld4(Wm + ln.wo + 16 * mi);
    }
    const gptr vec = uniform(w + 3 * DM * DM + 16 * mo);
    blk.bias = ld4(vec + 4 * ln.g);
    blk.al = ld4(vec + DM + 4 * ln.g);
    blk.be = ld4(vec + 2 * DM + 4 * ln.g);
    __builtin_amdgcn_sched_barrier(0);
  };
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
    auto compute = [&](const Block& blk, int mo) {
      f32x4 acc = blk.bias;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = mfma(blk.a[j][mi][i], bf[j][mi][i], acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = acc[i] * blk.al[i] + blk.be[i];
        acc[i] = v > 0.f ? v : (__expf(v) - 1.0f);
      }
      y[rt][mo] = acc;
    };
    Block b0, b1;
    load_block(b0, 0);
    load_block(b1, 1);
    compute(b0, 0);
    load_block(b0, 2);
    compute(b1, 1);
    load_block(b1, 3);
    compute(b0, 2);
    compute(b1, 3);
  }
  }
  store_rows<RT>(Yb, ln, y, Lc);
  wave_sync();
  const int Lo = (Lc - 1) / 2 + 1;
#pragma unroll
  for (int rt = 0; rt < MAXT; ++rt) {
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
               const Lane ln, int lane, int L, Act& x) {
  constexpr int CIN = 16 * NKB;
  constexpr int XW = CIN + 4;  // staging row stride
  wave_sync();
  for (int i = lane; i < L * (CIN / 4); i += 64) {
    const int r = i / (CIN / 4), c4 = i % (CIN / 4);
    st4(smem + r * XW + 4 * c4, ld4(xg + (size_t)r * CIN + 4 * c4));
  }
  wave_sync();
  const float* bias = w + 3 * DM * CIN;
  const float* pe = bias + DM;
  // all weight fragments (3 taps x NKB k-blocks x 4 output blocks) are requested up front
  f32x4 a[4][3][NKB], bv[4];
  const unsigned wo = ln.c * CIN + 4 * ln.g;
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const gptr Wm = uniform(w + (j * DM + 16 * mo) * CIN);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) a[mo][j][kb] = ld4(Wm + wo + 16 * kb);
    }
    bv[mo] = ld4(uniform(bias + 16 * mo) + 4 * ln.g);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int t = 16 * rt + ln.c;
    t = t < L ? t : L - 1;
    f32x4 bf[3][NKB], pv[4];
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) pv[mo] = ld4(pe + t * DM + 16 * mo + 4 * ln.g);
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
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = mfma(a[mo][j][kb][i], bf[j][kb][i], acc);
      acc += bv[mo];
      acc += pv[mo];
      x[rt][mo] = acc;
    }
  }
}

// Rows [first, first+n) of x (via LDS) -> rows [dst, dst+n) of `into`; other rows of `into`
// keep their value. Used for x[:, -inp_len:, :] and for the concatenation of the encoder
// outputs (encoder.py:96-99). RS / RD: tiles of the source / destination that can be involved.
template <int RS, int RD>
DEV void move_rows(float* smem, const Lane ln, const Act& x, int first, int n, int dst,
                   Act& into) {
  wave_sync();
  store_rows<RS>(smem, ln, x, first + n);
  wave_sync();
#pragma unroll
  for (int rt = 0; rt < RD; ++rt) {
    const int r = 16 * rt + ln.c;
    if (r >= dst && r < dst + n) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        into[rt][m] = ld4(smem + (r - dst + first) * LDW + 16 * m + 4 * ln.g);
    }
  }
}

// ---- two sequences in one tile ------------------------------------------------------------
// In the workgroup kernel waves w and w + 4 run on the same SIMD, one after the other. Once
// the rows per sequence are down to 8 or fewer, wave w (the primary) takes the rows of wave
// w + 4 (the partner) into rows 8.. of its tile and carries both sequences through the rest of
// the encoder stack with the MFMA work of one; the partner only keeps the weight staging going.
// The encoder outputs of the second sequence travel back through the partner's LDS region.

DEV void exchange_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// tap() for a packed tile: rows 0.. belong to sequence b, rows 8.. to sequence b2 (-1: none).
DEV void tap_packed(float*& taps, int B, int b, int b2, int rows, const Lane ln,
                    const Act& x) {
  if (!taps) return;
  const int bs = (ln.c >> 3) ? b2 : b, rl = ln.c & 7;
  if (bs >= 0 && rl < rows) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
      st4(taps + ((size_t)bs * rows + rl) * DM + 16 * m + 4 * ln.g, x[0][m]);
  }
  taps += (size_t)B * rows * DM;
}

// distil() for a packed tile (L <= 8 rows per sequence). NT = 1: the L + 2 <= 8 conv outputs of
// both sequences fit one tile (rows 0.. and 8..); NT = 2: up to 10 outputs per sequence, tile s
// holds those of sequence s. The pooled rows are packed again. Returns rows out.
template <int NT>
DEV int distil_packed(Stage& st, float* smem, const Lane ln, int L, Act& x) {
  float* Xb = smem;             // 16 input rows
  float* Yb = smem + 16 * LDW;  // NT tiles of activated conv outputs
  wave_sync();
  store_rows<1>(Xb, ln, x, 16);
  wave_sync();
  const int Lc = L + 2;
  Act y;
  NormParams ab;  // gm = alpha, bt = beta
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    WFrag A;
    wload_lds(A, stage_acquire(st), ln, j == 0 ? &ab : nullptr);
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const int base = NT == 1 ? (ln.c & 8) : 8 * rt;  // first input row of the lane's sequence
      int t = NT == 1 ? (ln.c & 7) : ln.c;
      t = t < Lc ? t : Lc - 1;
      int src = t + j - 2;  // circular padding of 2 on both sides, inside the own sequence
      src = src < 0 ? src + L : src;
      src = src >= L ? src - L : src;
      f32x4 bf[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) bf[mi] = ld4(Xb + (base + src) * LDW + 16 * mi + 4 * ln.g);
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) {
        f32x4 acc = j == 0 ? A.bv[mo] : y[rt][mo];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = mfma(A.a[mo][mi][i], bf[mi][i], acc);
        y[rt][mo] = acc;
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < NT; ++rt)
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = y[rt][mo][i] * ab.gm[mo][i] + ab.bt[mo][i];
        y[rt][mo][i] = v > 0.f ? v : (__expf(v) - 1.0f);
      }
  store_rows<NT>(Yb, ln, y, 16 * NT);
  wave_sync();
  const int Lo = (Lc - 1) / 2 + 1;
  const int ybase = NT == 1 ? (ln.c & 8) : 2 * (ln.c & 8);  // first output row of the sequence
  int u = ln.c & 7;
  u = u < Lo ? u : Lo - 1;
  const int r0 = ybase + (2 * u - 1 < 0 ? 0 : 2 * u - 1);  // window rows clipped to the sequence
  const int r1 = ybase + 2 * u;
  const int r2 = ybase + (2 * u + 1 > Lc - 1 ? Lc - 1 : 2 * u + 1);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const f32x4 a = ld4(Yb + r0 * LDW + 16 * m + 4 * ln.g);
    const f32x4 b = ld4(Yb + r1 * LDW + 16 * m + 4 * ln.g);
    const f32x4 c = ld4(Yb + r2 * LDW + 16 * m + 4 * ln.g);
#pragma unroll
    for (int i = 0; i < 4; ++i) x[0][m][i] = fmaxf(fmaxf(a[i], b[i]), c[i]);
  }
  return Lo;
}

// State that the steps of the encoder stack advance.
struct Cursor {
  const float* w;  // next weight block
  float* attn;     // next attention map block (null: not wanted)
  float* taps;     // next debug stage (null: not wanted)
};

// One encoder step on RT row tiles (RM: tiles of the memory).
template <int RT, int RM, bool WG, bool QR, int NHP, int HPT>
DEV void run_op(const Op op, const KArgs& a, Cursor& cu, Stage& st, float* smem, int R, const Lane ln,
                int b, Act& x, Act& mem) {
  const int L = op.L;
  if (op.type == OP_LAYER) {
    float* amap = nullptr;
    if (cu.attn) {
      amap = cu.attn + (size_t)b * a.n_heads * L * L;
      cu.attn += (size_t)a.B * a.n_heads * L * L;
    }
    if (!WG && QR) encoder_layer_ahead<RT, NHP, HPT>(cu.w, smem, R, ln, L, a.activation, amap, x);
    else encoder_layer<RT, WG, QR, NHP>(cu.w, st, smem, R, ln, L, a.activation, amap, x);
    if (!ABLATED(a, 8)) cu.w += sz_enc_layer<NHP>();
    tap<RT>(cu.taps, a.B, b, L, ln, x);
  } else if (op.type == OP_DISTIL) {
    const int Lo = distil<RT, WG>(cu.w, st, smem, R, ln, L, x);
    cu.w += SZ_DISTIL;
    tap<RT>(cu.taps, a.B, b, Lo, ln, x);
    PROF_MARK(ln, P_DISTIL);
  } else if (op.type == OP_NORM_APPEND) {
    layer_norm<RT>(x, cu.w, cu.w + DM, ln);
    cu.w += 2 * DM;
    tap<RT>(cu.taps, a.B, b, L, ln, x);
    PROF_MARK(ln, P_ADDLN);
    move_rows<RT, RM>(smem, ln, x, 0, L, op.a, mem);
    PROF_MARK(ln, P_MOVE);
  } else {
    move_rows<RT, RT>(smem, ln, x, op.a, op.b, 0, x);
    PROF_MARK(ln, P_MOVE);
  }
}

// Rows 0..15 of a partner's region carry the rows exchanged between the two waves; the map tiles
// follow (8 KiB of the 32 rows a packed region has beyond them).
constexpr int MAP_STAGE_ROW = 16;
static_assert(MAP_STAGE_FLOATS <= (48 - MAP_STAGE_ROW) * LDW, "map staging area exceeds a packed region");
// The partner's side: stores heads [h0, h1) of the staged map tiles of a packed layer.
DEV void partner_write_maps(const float* stage, float* amap, float* amap2, int L, const Lane ln, int h0, int h1) {
  if (!amap && !amap2) return;
  const __amdgpu_buffer_rsrc_t r = packed_map_rsrc(amap, amap2, L, ln.n_heads);
  const PackedMapLanes ml = packed_map_lanes(amap, amap2, L, ln.c, ln.g);
  if (h1 > ln.n_heads) h1 = ln.n_heads;
  for (int h = h0; h < h1; ++h) packed_map_put(r, ml, h, L, ld4(stage + (h * 64 + 16 * ln.g + ln.c) * 4));
}
// One encoder step of the primary wave on a packed tile (sequences b and b2; b2 = -1: none).
template <int RM>
DEV void run_op_packed(const Op op, const KArgs& a, Cursor& cu, Stage& st, float* smem,
                       float* partner_smem, int R, const Lane ln, int b, int b2, Act& x,
                       Act& mem) {
  const int L = op.L;
  if (op.type == OP_LAYER) {
    float *amap = nullptr, *amap2 = nullptr;
    if (cu.attn) {
      if (b >= 0) amap = cu.attn + (size_t)b * a.n_heads * L * L;
      if (b2 >= 0) amap2 = cu.attn + (size_t)b2 * a.n_heads * L * L;
      cu.attn += (size_t)a.B * a.n_heads * L * L;
    }
    // the tiles of the maps go to rows MAP_STAGE_ROW.. of the partner's region; the partner stores
    // them once the staging barrier of the output projection has published them
    encoder_layer<1, true, true, 4>(cu.w, st, smem, R, ln, 16, a.activation, amap, x, L, amap2,
                                    partner_smem + MAP_STAGE_ROW * LDW);
    cu.w += sz_enc_layer<4>();
    tap_packed(cu.taps, a.B, b, b2, L, ln, x);
  } else if (op.type == OP_DISTIL) {
    const int Lo = L + 2 <= 8 ? distil_packed<1>(st, smem, ln, L, x) : distil_packed<2>(st, smem, ln, L, x);
    cu.w += SZ_DISTIL;
    tap_packed(cu.taps, a.B, b, b2, Lo, ln, x);
  } else {  // OP_NORM_APPEND (the host packs no other step)
    layer_norm<1>(x, cu.w, cu.w + DM, ln);
    cu.w += 2 * DM;
    tap_packed(cu.taps, a.B, b, b2, L, ln, x);
    move_rows<1, RM>(smem, ln, x, 0, L, op.a, mem);
    // the second sequence's rows go back through the partner's region
    store_rows<1>(partner_smem, ln, x, 16);
    exchange_barrier();
  }
}
// The same step seen from the partner wave: it keeps its share of the weight staging and the
// barriers, advances the cursors and collects its encoder outputs.
// attn_all: the maps of the step's layer for all sequences (null: not wanted); b1 / b2: the
// primary's and the own sequence (-1: beyond the batch).
template <int RM>
DEV void run_op_partner(const Op op, const KArgs& a, Cursor& cu, Stage& st, float* smem,
                        const Lane ln, Act& mem, float*& attn_all, int b1, int b2) {
  const int L = op.L;
  if (op.type == OP_LAYER) {
    float *amap = nullptr, *amap2 = nullptr;
    if (attn_all) {
      if (b1 >= 0) amap = attn_all + (size_t)b1 * a.n_heads * L * L;
      if (b2 >= 0) amap2 = attn_all + (size_t)b2 * a.n_heads * L * L;
      attn_all += (size_t)a.B * a.n_heads * L * L;
    }
    // blocks q k v | o conv1 conv2. The staging barrier of the output projection (every acquire
    // but the packed phase's very first passes one) has the primary's map tiles in LDS; they
    // are stored in three parts, so that the acknowledgements the next barrier's wait collects
    // with the weight pieces do not make this wave late for it.
    for (int k = 0; k < 6; ++k) {
      stage_acquire(st);
      if (k == 3) partner_write_maps(smem + MAP_STAGE_ROW * LDW, amap, amap2, L, ln, 0, 3);
      if (k == 4) partner_write_maps(smem + MAP_STAGE_ROW * LDW, amap, amap2, L, ln, 3, 6);
      if (k == 5) partner_write_maps(smem + MAP_STAGE_ROW * LDW, amap, amap2, L, ln, 6, 8);
    }
    if (cu.attn) cu.attn += (size_t)a.B * a.n_heads * L * L;
    cu.w += sz_enc_layer<4>();
    if (cu.taps) cu.taps += (size_t)a.B * L * DM;
  } else if (op.type == OP_DISTIL) {
    for (int k = 0; k < 3; ++k) stage_acquire(st);
    cu.w += SZ_DISTIL;
    const int Lo = (L + 1) / 2 + 1;
    if (cu.taps) cu.taps += (size_t)a.B * Lo * DM;
  } else {
    cu.w += 2 * DM;
    if (cu.taps) cu.taps += (size_t)a.B * L * DM;
    exchange_barrier();
#pragma unroll
    for (int rt = 0; rt < RM; ++rt) {
      const int r = 16 * rt + ln.c;
      if (r >= op.a && r < op.a + L) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          mem[rt][m] = ld4(smem + (8 + r - op.a) * LDW + 16 * m + 4 * ln.g);
      }
    }
  }
}

// One wave = one sequence. RD / RM: row tiles of the decoder sequence / of the encoder memory
// (known on the host). The encoder steps that touch more than 16 rows come first (sequence
// lengths only shrink along the stack) and run on two row tiles; everything after them runs
// on one.
// WG = false: a workgroup is one wave; every wave loads its weights from global memory.
// WG = true: a workgroup is 8 waves that keep in step through the weight staging (Stage);
// waves beyond the batch compute a copy of the last sequence and store nothing.
// RE: row tiles of the longest encoder step (2: sequences of up to 30 slots; 3: up to 46).
// NHP: 16-column tiles of the Q / K / V projections, HPT heads each (4 x 2: up to 8 heads; 6 x 2: up
// to 12 heads of up to 8 features; 4 x 3: up to 12 heads of up to 5). The instantiations with three
// row tiles or more than 8 heads ("large": the one-wave shape only) get the register budget of one
// wave per SIMD.
template <int RE, int RD, int NHP, int HPT>
constexpr bool large_shape() { return RE > 2 || RD > 2 || NHP > 4 || HPT > 2; }
template <int RE, int RD, int RM, bool WG, int NHP, int HPT = 2>
__global__ __launch_bounds__(WG ? 64 * WG_WAVES : 64,
                             (WG || large_shape<RE, RD, NHP, HPT>()) ? 1 : CET_WPS) void informer_kernel(const KArgs a) {
  static_assert(!WG || (RE <= 2 && RD <= 2 && NHP == 4 && HPT == 2), "the workgroup shape serves the 64-wide, 8-head family");
  // queries in registers (and Q / K projections with permuted rows, see attention<.., QREG>)
  constexpr bool QR = WG || large_shape<RE, RD, NHP, HPT>();
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  const int lane = threadIdx.x & 63;
  const int wave = WG ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const int R = a.region_rows;
  const int seq = a.first + (WG ? (int)blockIdx.x * WG_WAVES + wave : (int)blockIdx.x);
  const bool valid = seq < a.last;
  const int b = valid ? seq : a.last - 1;
  float* smem = WG ? smem_all + 2 * WBLK + wave * a.wave_lds_floats : smem_all;
#ifdef CET_PROF
  int prof_idx = 0;
  const int prof_stride = a.B >= PROF_SAMPLES ? (a.B / PROF_SAMPLES) | 4 : 1;  // waves 0 and 4 in turn
  const int prof_slot = seq / prof_stride;
  const bool prof_on = valid && seq % prof_stride == 0 && prof_slot < PROF_SAMPLES;
  Lane ln{lane & 15, lane >> 4, (unsigned)((lane & 15) * DM + 4 * (lane >> 4)), a.ln_inv_n, a.ln_pad,
          a.attn_c, a.n_heads};
  ln.prof = prof_on ? a.prof + (size_t)prof_slot * PROF_MAXM : nullptr;
  ln.pidx = &prof_idx;
  PROF_MARK(ln, P_OTHER);
#else
  Lane ln{lane & 15, lane >> 4, (unsigned)((lane & 15) * DM + 4 * (lane >> 4)), a.ln_inv_n, a.ln_pad,
          a.attn_c, a.n_heads};
#endif
#ifdef CET_EXP
  ln.ablate = a.ablate;
#endif
  Stage st{a.wblk, smem_all, WG ? lds_address(smem_all) : 0u, 0, WG ? a.n_blk : 0, wave, lane, 0, 0,
           WG && wave >= WG_WAVES / 2, wave, WG_WAVES};
  if (WG) stage_request(st);  // block 0 travels while the embedding is computed

  // Rows the padded 16-row tiles may touch must hold finite numbers from the start.
  for (int i = lane; i < a.wave_lds_floats / 4; i += 64) st4(smem + 4 * i, zero4());

  Cursor cu{a.w, valid ? a.attn : nullptr, valid ? a.taps : nullptr};
  Act x, mem;
#pragma unroll
  for (int rt = 0; rt < MAXT; ++rt)
#pragma unroll
    for (int m = 0; m < 4; ++m) x[rt][m] = mem[rt][m] = zero4();

  // ---- encoder embedding ------------------------------------------------------------
  {
    const int L = a.seq_len;
    const float* xg = a.x_enc + (size_t)b * L * a.enc_in;
    if (RE > 2 && L > 32) {
      if (a.enc_in == 16) embed<RE, 1>(cu.w, xg, smem, ln, lane, L, x);
      else embed<RE, 2>(cu.w, xg, smem, ln, lane, L, x);
      tap<RE>(cu.taps, a.B, b, L, ln, x);
    } else if (L > 16) {
      if (a.enc_in == 16) embed<2, 1>(cu.w, xg, smem, ln, lane, L, x);
      else embed<2, 2>(cu.w, xg, smem, ln, lane, L, x);
      tap<2>(cu.taps, a.B, b, L, ln, x);
    } else {
      if (a.enc_in == 16) embed<1, 1>(cu.w, xg, smem, ln, lane, L, x);
      else embed<1, 2>(cu.w, xg, smem, ln, lane, L, x);
      tap<1>(cu.taps, a.B, b, L, ln, x);
    }
    cu.w += 3 * DM * a.enc_in + DM + L * DM;
    PROF_MARK(ln, P_EMBED);
  }
  const float* w_dec_emb = cu.w;
  const int Ld = a.label_len + a.pred_len;
  cu.w += 3 * DM * a.dec_in + DM + Ld * DM;

  // ---- encoder stack (encoder.py:87-101) ---------------------------------------------
  int i = 0;
  if (RE > 2)
    for (; i < a.n_ops3; ++i) run_op<RE, RM, WG, QR, NHP, HPT>(a.ops[i], a, cu, st, smem, R, ln, b, x, mem);
  for (; i < a.n_ops2; ++i) run_op<2, RM, WG, QR, NHP, HPT>(a.ops[i], a, cu, st, smem, R, ln, b, x, mem);
  const int n_plain = WG ? a.n_pack0 : a.n_ops;
  for (; i < n_plain; ++i) run_op<1, RM, WG, QR, NHP, HPT>(a.ops[i], a, cu, st, smem, R, ln, b, x, mem);
  if (WG && i < a.n_ops) {
    // hand the partner's rows to the primary (see "two sequences in one tile")
    const int L0 = a.ops[i].L;
    stage_align(st);
    st.lag = false;  // the packed steps are computed by waves 0..3 alone,
    stage_share(st, true);  // and waves 4..7, idle meanwhile, request the weight blocks
    if (wave >= WG_WAVES / 2) {
      wave_sync();
      store_rows<1>(smem, ln, x, L0);
      exchange_barrier();
      // the maps of the next layer, for all sequences: behind those of the layers done so far
      float* attn_all = a.attn;
      if (attn_all)
        for (int j = 0; j < i; ++j)
          if (a.ops[j].type == OP_LAYER) attn_all += (size_t)a.B * a.n_heads * a.ops[j].L * a.ops[j].L;
      const int seq1 = seq - WG_WAVES / 2;
      for (; i < a.n_ops; ++i)
        run_op_partner<RM>(a.ops[i], a, cu, st, smem, ln, mem, attn_all, seq1 < a.last ? seq1 : -1, valid ? b : -1);
    } else {
      float* partner_smem = smem + (WG_WAVES / 2) * a.wave_lds_floats;
      const int seq2 = seq + WG_WAVES / 2;
      exchange_barrier();
      if (ln.c >= 8 && ln.c < 8 + L0) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          x[0][m] = ld4(partner_smem + (ln.c - 8) * LDW + 16 * m + 4 * ln.g);
      }
      for (; i < a.n_ops; ++i)
        run_op_packed<RM>(a.ops[i], a, cu, st, smem, partner_smem, R, ln, valid ? b : -1,
                          seq2 < a.last ? seq2 : -1, x, mem);
    }
  }
  tap<RM>(cu.taps, a.B, b, a.mem_rows, ln, mem);
  st.lag = WG && wave >= WG_WAVES / 2;
  stage_share(st, false);

  // ---- decoder ---------------------------------------------------------------------
  {
    const float* xg = a.x_dec + (size_t)b * Ld * a.dec_in;
    if (a.dec_in == 16) embed<RD, 1>(w_dec_emb, xg, smem, ln, lane, Ld, x);
    else embed<RD, 2>(w_dec_emb, xg, smem, ln, lane, Ld, x);
    tap<RD>(cu.taps, a.B, b, Ld, ln, x);
    PROF_MARK(ln, P_EMBED);
  }
  for (int l = 0; l < a.d_layers; ++l) {
    if (!WG && QR) decoder_layer_ahead<RD, RM, NHP, HPT>(cu.w, smem, R, ln, Ld, a.mem_rows, a.activation, mem, x);
    else decoder_layer<RD, RM, WG, QR, NHP>(cu.w, st, smem, R, ln, Ld, a.mem_rows, a.activation, mem, x);
    if (!ABLATED(a, 8)) cu.w += sz_dec_layer<NHP>();
    tap<RD>(cu.taps, a.B, b, Ld, ln, x);
  }
  layer_norm<RD>(x, cu.w, cu.w + DM, ln);
  cu.w += 2 * DM;
  tap<RD>(cu.taps, a.B, b, Ld, ln, x);

  // ---- projection Linear(64 -> c_out) and the last pred_len rows ------------------------
  // Block: W[ceil16(c_out)][64] zero padded | bias[ceil16(c_out)].
  PROF_MARK(ln, P_ADDLN);
  const float* w = cu.w;
  const int n_mo = (a.c_out + 15) / 16;
  const float* pb = w + n_mo * 16 * DM;
  for (int mo = 0; mo < n_mo; ++mo) {
    f32x4 wa[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) wa[mi] = ld4(w + (16 * mo + ln.c) * DM + 16 * mi + 4 * ln.g);
    const f32x4 bv = ld4(pb + 16 * mo + 4 * ln.g);
#pragma unroll
    for (int rt = 0; rt < RD; ++rt) {
      f32x4 acc = bv;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) acc = mfma(wa[mi][i2], x[rt][mi][i2], acc);
      const int r = 16 * rt + ln.c;
      const int f0 = 16 * mo + 4 * ln.g;
      if (valid && r >= Ld - a.pred_len && r < Ld && f0 < a.c_out)
        st4(a.out + ((size_t)b * a.pred_len + (r - (Ld - a.pred_len))) * a.c_out + f0, acc);
    }
  }
  PROF_MARK(ln, P_PROJ);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Host side

struct InformerModel {
  cet_informer_cfg cfg;
  Dims d;
  float* dev_blob = nullptr;
  float* dev_blocks = nullptr;  // staged Linear blocks of the workgroup kernel
  int n_blocks = 0;
  size_t attn_per_sample = 0;  // floats
  size_t taps_per_sample = 0;
  int device = 0;  // the HIP device that holds the weights
  int cus = 0;     // its compute units
  // Launch shape: 0 = by batch size, 1 = one wave per sequence, 2 = workgroups of 8 sequences;
  // pack = two sequences per tile in the short encoder stages of the workgroup kernel. Set
  // through cet_informer_select only (the library reads no environment variables).
  int kernel_choice = 0;
  bool pack = true;
  bool large = false;  // large_model(): its query / key projections are stored row-permuted
};

// Head pairs the kernels run a model with: 4 (up to 8 heads) or 6 (up to 12).
// Heads per 16-column tile of the Q / K / V projections and the number of such tiles: two heads of
// up to 8 features (4 tiles for up to 8 heads, 6 for up to 12), or, for 9 to 12 heads of at most 5
// features, three to a tile (4 tiles).
static int heads_per_tile(const cet_informer_cfg& c) { return c.n_heads > 8 && c.d_model / c.n_heads <= 5 ? 3 : 2; }
static int head_pairs(const cet_informer_cfg& c) { return c.n_heads <= 8 || heads_per_tile(c) == 3 ? 4 : 6; }

// Steps of the encoder stack (encoder.py:87-101) into a.ops / n_ops / n_ops2 / n_ops3 / mem_rows.
static void encoder_steps(const Dims& d, KArgs& a) {
  int n = 0, cur = d.seq_len, inp_len = d.seq_len, mem_rows = 0, n2 = 0, n3 = 0;
  auto push = [&](int type, int L, int p, int q, int need) {
    a.ops[n] = Op{(unsigned char)type, (unsigned char)L, (unsigned char)p, (unsigned char)q};
    if (need > 16) n2 = n + 1;
    if (need > 32) n3 = n + 1;
    ++n;
  };
  for (int e = 0; e < d.n_enc; ++e) {
    const int el = d.enc_layers[e];
    const int take = inp_len < cur ? inp_len : cur;
    if (take < cur) push(OP_SLICE, cur, cur - take, take, cur);
    int L = take;
    for (int l = 0; l < el; ++l) {
      push(OP_LAYER, L, 0, 0, L);
      if (d.distil && l < el - 1) {
        push(OP_DISTIL, L, 0, 0, L + 2);
        L = (L + 1) / 2 + 1;
      }
    }
    push(OP_NORM_APPEND, L, mem_rows, 0, L);
    mem_rows += L;
    cur = L;
    inp_len /= 2;
  }
  a.n_ops = n;
  a.n_ops2 = n2;
  a.n_ops3 = n3;
  a.mem_rows = mem_rows;
}

// "Large" models run on the one-wave shape with three row tiles and / or six head pairs: more
// than 8 heads, or more than 32 rows somewhere (the conv of a distil step adds two).
static bool large_model(const cet_informer_cfg& c, const Dims& d) {
  KArgs a{};
  encoder_steps(d, a);
  return c.n_heads > 8 || a.n_ops3 > 0 || d.seq_len > 32 || d.label_len + d.pred_len > 32;
}

static bool supported(const cet_informer_cfg& c) {
  // models run zero-padded to the kernels' 64 features and head layouts (widen())
  if (c.n_heads < 1 || c.n_heads > 12 || c.d_model < c.n_heads || c.d_model > DM) return false;
  if (c.d_model / c.n_heads > HDIM || c.d_ff < 1 || c.d_ff > DM) return false;
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
  if (mem_rows > 32) return false;  // the cross-attention memory spans at most two row tiles
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
  const size_t D = c->d_model, F = c->d_ff, HE = (size_t)(c->d_model / c->n_heads) * c->n_heads;
  const size_t attn = 3 * (HE * D + HE) + D * HE + D, ffn = F * D + F + D * F + D;
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

// Output position p of a staged query / key projection holds feature qk_row(p): inside every
// 16-feature tile, position 4 g + i <- feature 8 (i >> 1) + 2 g + (i & 1).
static int qk_row(int p) {
  const int t = p >> 4, g = (p >> 2) & 3, i = p & 3;
  return 16 * t + 8 * (i >> 1) + 2 * g + (i & 1);
}

// The inverse: feature f of a query / key projection sits at output position qk_pos(f).
static int qk_pos(int f) {
  const int t = f >> 4, r = f & 15;
  return 16 * t + 4 * ((r >> 1) & 3) + 2 * (r >> 3) + (r & 1);
}

// Three heads to a tile: dimension e < 5 of head j < 3 of the tile sits at position 4 e + j
// (register j of lane group e) for e < 4 and at 4 j + 3 (register 3 of lane group j) for e = 4.
static int qk3_pos(int j, int e) { return e < 4 ? 4 * e + j : 4 * j + 3; }

// [out][in][3] (PyTorch Conv1d) -> [3][out][in]
static void conv_taps_major(const float* src, int cout, int cin, std::vector<float>& dst) {
  for (int j = 0; j < 3; ++j)
    for (int o = 0; o < cout; ++o)
      for (int c = 0; c < cin; ++c) dst.push_back(src[((size_t)o * cin + c) * 3 + j]);
}

// Zero-pads a model to the layout the kernels are built for, in the caller's tensor order: 64
// model and FFN features (d and f keep their index), and 8 features per head, dimension e of
// head h at 8 h + e of a Q / K / V row of HQ = 16 * head_pairs features (9 to 12 heads of at most
// 5 features: three heads to a 16-feature tile, 5 positions each, see attention<.., HPT>). Padded rows / columns,
// biases, LayerNorm and BatchNorm scales are zero (BatchNorm variance one), so padded features
// are exactly zero everywhere; LayerNorm and the attention scale take the real sizes from the
// kernel arguments. A 64-wide model with 8 heads of 8 comes out as it went in.
// qk_permuted (large models, whose kernels keep the queries in registers): the output rows of
// every query and key projection are stored in the order of qk_row() (qk3_pos() with three heads
// to a tile).
static std::vector<float> widen(const cet_informer_cfg& c, const Dims& d, const float* p, bool qk_permuted) {
  const int D = c.d_model, F = c.d_ff, E = D / c.n_heads, HE = E * c.n_heads;
  const int HQ = 16 * head_pairs(c);
  std::vector<float> v;
  auto ident = [](int i) { return i; };
  const int HPT = heads_per_tile(c);
  // position of dimension e of head h in a value / context row
  auto head = [E, HPT](int i) {
    const int h = i / E, e = i % E;
    return HPT == 3 ? (h / 3) * 16 + (h % 3) * 5 + e : h * HDIM + e;
  };
  // ... and in a permuted query / key row
  auto qk_head = [E, HPT](int i) {
    const int h = i / E, e = i % E;
    return HPT == 3 ? (h / 3) * 16 + qk3_pos(h % 3, e) : qk_pos(h * HDIM + e);
  };
  // matrix [rows][cols][taps] -> [R][C][taps]
  auto mat = [&](int rows, int cols, int taps, int R, int C, auto rmap, auto cmap) {
    const size_t base = v.size();
    v.resize(base + (size_t)R * C * taps, 0.f);
    for (int r = 0; r < rows; ++r)
      for (int q = 0; q < cols; ++q)
        for (int t = 0; t < taps; ++t)
          v[base + ((size_t)rmap(r) * C + cmap(q)) * taps + t] = p[((size_t)r * cols + q) * taps + t];
    p += (size_t)rows * cols * taps;
  };
  auto vec = [&](int n, auto map, float fill = 0.f, int N = DM) {
    const size_t base = v.size();
    v.resize(base + N, fill);
    for (int i = 0; i < n; ++i) v[base + map(i)] = p[i];
    p += n;
  };
  auto attn = [&]() {
    for (int k = 0; k < 3; ++k) {
      if (qk_permuted && k < 2) {
        mat(HE, D, 1, HQ, DM, qk_head, ident);
        vec(HE, qk_head, 0.f, HQ);
      } else {
        mat(HE, D, 1, HQ, DM, head, ident);
        vec(HE, head, 0.f, HQ);
      }
    }
    mat(D, HE, 1, DM, HQ, ident, head);
    vec(D, ident);
  };
  auto ffn = [&]() {
    mat(F, D, 1, DM, DM, ident, ident);
    vec(F, ident);
    mat(D, F, 1, DM, DM, ident, ident);
    vec(D, ident);
  };
  const int emb_in[2] = {c.enc_in, c.dec_in};
  const int emb_len[2] = {c.seq_len, c.label_len + c.pred_len};
  for (int s = 0; s < 2; ++s) {
    mat(D, emb_in[s], 3, DM, emb_in[s], ident, ident);
    vec(D, ident);
    mat(emb_len[s], D, 1, emb_len[s], DM, ident, ident);  // positional table rows
  }
  for (int e = 0; e < d.n_enc; ++e) {
    const int el = d.enc_layers[e];
    for (int l = 0; l < el; ++l) {
      attn();
      ffn();
      for (int k = 0; k < 4; ++k) vec(D, ident);  // norm1 g,b, norm2 g,b
    }
    if (c.distil)
      for (int l = 0; l < el - 1; ++l) {
        mat(D, D, 3, DM, DM, ident, ident);
        vec(D, ident);                              // conv bias
        vec(D, ident);                              // BatchNorm weight
        vec(D, ident);                              // BatchNorm bias
        vec(D, ident);                              // running mean
        vec(D, ident, 1.f);                         // running variance
      }
    vec(D, ident);
    vec(D, ident);
  }
  for (int l = 0; l < c.d_layers; ++l) {
    attn();
    attn();
    ffn();
    for (int k = 0; k < 6; ++k) vec(D, ident);
  }
  vec(D, ident);
  vec(D, ident);
  mat(c.c_out, D, 1, c.c_out, DM, ident, ident);
  v.insert(v.end(), p, p + c.c_out);
  return v;
}

int informer_create(const cet_informer_cfg* c_in, const float* wsrc_in, size_t n_in,
                    InformerModel** out) {
  if (!supported(*c_in)) return CET_ERR_UNSUPPORTED;
  Dims d;
  if (!plan(*c_in, d)) return CET_ERR_UNSUPPORTED;
  if (n_in != informer_blob_size(c_in)) return CET_ERR_BLOB_SIZE;
  // from here on everything has the kernels' dimensions: 64 features, HQ-wide Q / K / V
  const cet_informer_cfg* c = c_in;
  const int NHP = head_pairs(*c_in);
  const size_t HQ = 16 * NHP;
  const bool large = large_model(*c_in, d);
  const std::vector<float> wide = widen(*c_in, d, wsrc_in, large);
  const float* wsrc = wide.data();
  const size_t n = wide.size();

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
  // Staged blocks of the workgroup kernel, in consumption order: every Linear(64,64) as
  // W[64][LDW] | bias | gamma | beta (the LayerNorm that follows it, if any) | zero padding.
  // qk: the block is a query or key projection, whose output rows the workgroup kernel wants in
  // the order of qk_row() (see attention<.., QREG>).
  std::vector<float> blk;
  const bool staged = !large;  // the workgroup kernel serves the 8-head family of up to 30 slots only
  auto add_block = [&](const float* lin, const float* norm, bool qk = false) {
    if (!staged) return;
    const size_t base = blk.size();
    blk.resize(base + WBLK, 0.f);
    for (int r = 0; r < DM; ++r) {
      const int src = qk ? qk_row(r) : r;
      std::copy(lin + (size_t)src * DM, lin + (size_t)(src + 1) * DM, blk.begin() + base + (size_t)r * LDW);
      blk[base + WBLK_BIAS + r] = lin[D * D + src];
    }
    if (norm) std::copy(norm, norm + 2 * D, blk.begin() + base + WBLK_NORM);
  };
  const size_t lin_sz = D * D + D;                      // Linear(64, 64): conv1 / conv2
  const size_t qkv_sz = HQ * D + HQ;                    // Q / K / V projection
  const size_t attn_sz = 3 * qkv_sz + D * HQ + D;       // + output projection
  const size_t layer_sz = attn_sz + 2 * lin_sz + 4 * D;
  for (int e = 0; e < d.n_enc; ++e) {
    const int el = d.enc_layers[e];
    const float* layers = p;
    const float* convs = p + layer_sz * el;
    for (int l = 0; l < el; ++l) {
      v.insert(v.end(), layers + layer_sz * l, layers + layer_sz * (l + 1));
      {
        const float* lw = layers + layer_sz * l;       // q k v o conv1 conv2 norm1 norm2
        const float* norms = lw + attn_sz + 2 * lin_sz;
        for (int k = 0; k < 3; ++k) add_block(lw + k * qkv_sz, nullptr, k < 2);
        add_block(lw + 3 * qkv_sz, norms);
        add_block(lw + attn_sz, nullptr);
        add_block(lw + attn_sz + lin_sz, norms + 2 * D);
      }
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
        // staged: one block per tap (the taps-major copy just appended to v is [3][out][in])
        {
          const size_t taps_at = v.size() - (3 * D * D + 3 * D);
          std::vector<float> lin(D * D + D), ab(2 * D);
          std::copy(alpha.begin(), alpha.end(), ab.begin());
          std::copy(beta.begin(), beta.end(), ab.begin() + D);
          for (int j = 0; j < 3; ++j) {
            std::copy(v.begin() + taps_at + (size_t)j * D * D, v.begin() + taps_at + (size_t)(j + 1) * D * D, lin.begin());
            for (size_t i = 0; i < D; ++i) lin[D * D + i] = j == 0 ? cb[i] : 0.f;
            add_block(lin.data(), j == 0 ? ab.data() : nullptr);
          }
        }
      }
    }
    p = convs + (c->distil ? (size_t)(el - 1) * (3 * D * D + 5 * D) : 0);
    copy(2 * D);  // encoder norm
  }
  for (int l = 0; l < c->d_layers; ++l) {
    const float* lw = p;  // self q k v o | cross q k v o | conv1 conv2 | norm1 norm2 norm3
    const float* norms = lw + 2 * attn_sz + 2 * lin_sz;
    for (int k = 0; k < 3; ++k) add_block(lw + k * qkv_sz, nullptr, k < 2);
    add_block(lw + 3 * qkv_sz, norms);
    for (int k = 0; k < 3; ++k) add_block(lw + attn_sz + k * qkv_sz, nullptr, k < 2);
    add_block(lw + attn_sz + 3 * qkv_sz, norms + 2 * D);
    add_block(lw + 2 * attn_sz, nullptr);
    add_block(lw + 2 * attn_sz + lin_sz, norms + 4 * D);
    copy(2 * attn_sz + 2 * lin_sz + 6 * D);
  }
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
  m->cfg = *c_in;
  m->d = d;
  m->large = large;
  for (int i = 0; i < d.n_attn; ++i) m->attn_per_sample += (size_t)c_in->n_heads * d.attn_len[i] * d.attn_len[i];
  for (int i = 0; i < d.n_taps; ++i) m->taps_per_sample += (size_t)d.tap_rows[i] * DM;
  m->n_blocks = (int)(blk.size() / WBLK);
  hipError_t e = hipGetDevice(&m->device);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&m->cus, hipDeviceAttributeMultiprocessorCount, m->device);
  // the workgroup kernel's dynamic LDS exceeds the default limit: raised once per device, here
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&informer_kernel<2, 1, 1, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&informer_kernel<2, 1, 2, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&informer_kernel<2, 2, 1, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&informer_kernel<2, 2, 2, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipMalloc(&m->dev_blob, v.size() * sizeof(float));
  if (e == hipSuccess)
    e = hipMemcpy(m->dev_blob, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && !blk.empty()) e = hipMalloc(&m->dev_blocks, blk.size() * sizeof(float));
  if (e == hipSuccess && !blk.empty())
    e = hipMemcpy(m->dev_blocks, blk.data(), blk.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (m->dev_blob) (void)hipFree(m->dev_blob);
    if (m->dev_blocks) (void)hipFree(m->dev_blocks);
    delete m;
    return hip_fail(e);
  }
  *out = m;
  return CET_OK;
}

void informer_destroy(InformerModel* m) {
  if (!m) return;
  if (m->dev_blob) (void)hipFree(m->dev_blob);
  if (m->dev_blocks) (void)hipFree(m->dev_blocks);
  delete m;
}

int informer_device(const InformerModel* m) { return m->device; }
void informer_select(InformerModel* m, int kernel, int pack) {
  if (kernel >= 0) m->kernel_choice = kernel;
  if (pack >= 0) m->pack = pack != 0;
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
  {
    encoder_steps(m->d, a);
    const int n = a.n_ops, n2 = a.n_ops2;
    // Packed tail of the step list: from the first step on after which every layer and distil
    // step has at most 8 rows per sequence and nothing is sliced.
    int p0 = n;
    for (int k = n - 1; k >= n2; --k) {
      const Op& o = a.ops[k];
      const bool ok = (o.type == OP_LAYER && o.L <= 8) || (o.type == OP_DISTIL && o.L <= 8) ||
                      (o.type == OP_NORM_APPEND && o.L <= 8);
      if (!ok) break;
      p0 = k;
    }
    while (p0 < n && a.ops[p0].type != OP_LAYER) ++p0;  // start at a layer
    a.n_pack0 = m->pack ? p0 : n;
  }
  // LDS of a wave. One-wave kernel: three regions of R rows (Q | K | V), which also cover the
  // conv / max-pool staging (L + (L + 2) rows) and the staging of move_rows. Workgroup kernel:
  // K | V (2 R rows) or the conv staging (R + R + 2 rows), next to the two weight buffers.
  const int Ld = m->d.label_len + m->d.pred_len;
  int R = m->d.seq_len > Ld ? m->d.seq_len : Ld;
  if (R < m->d.mem_rows) R = m->d.mem_rows;
  if (R < 12) R = 12;
  // "large": three row tiles somewhere or six head pairs; one-wave shape, one instantiation per
  // head-pair count with three decoder and two memory tiles (padded tiles hold finite rows).
  const int NHP = head_pairs(m->cfg);
  const bool large = m->large;
  if (large && R < 24) R = 24;  // padded key tiles are read up to row 47 of the K | V regions
  a.region_rows = R;
  a.n_heads = m->cfg.n_heads;
  a.ln_inv_n = 1.0f / (float)m->cfg.d_model;
  a.ln_pad = (float)(DM - m->cfg.d_model);
  a.attn_c = (float)(1.0 / sqrt((double)(m->cfg.d_model / m->cfg.n_heads))) * 1.44269504088896340736f;
  a.wblk = m->dev_blocks;
  a.n_blk = m->n_blocks;
#ifdef CET_EXP
  static const int ablate = [] {
    const char* e = getenv("CET_INFORMER_ABLATE");
    return e ? atoi(e) : 0;
  }();
  a.ablate = ablate;
#endif
  const bool force_one = m->kernel_choice == 1, force_wg = m->kernel_choice == 2;
  // Packed tiles are whole 16-row tiles: K | V regions of at least 16 rows, and the packed
  // conv staging of 16 + 32 rows. Where that does not fit, the workgroup kernel runs unpacked.
  int R_wg = R, wg_rows = 2 * R + 2;
  if (a.n_pack0 < a.n_ops) {
    const int Rp = R < 16 ? 16 : R;
    const int rows_p = 2 * Rp + 2 < 48 ? 48 : 2 * Rp + 2;
    if (2 * WBLK * sizeof(float) + WG_WAVES * (size_t)rows_p * LDW * sizeof(float) <= 160 * 1024) {
      R_wg = Rp;
      wg_rows = rows_p;
    } else {
      a.n_pack0 = a.n_ops;
    }
  }
  const size_t wg_wave_bytes = (size_t)wg_rows * LDW * sizeof(float);
  const size_t wg_lds = 2 * WBLK * sizeof(float) + WG_WAVES * wg_wave_bytes;
  // The workgroup kernel fills a CU with 8 sequences and takes about 0.27 ms per round of
  // 8 x CUs sequences; the one-wave kernel spreads single waves over all CUs and finishes sooner
  // as long as every SIMD holds at most one of them (measured: 0.23 vs 0.27 ms at 4 sequences
  // per CU, 0.32 vs 0.28 ms at 5).
  const bool wg = !large && m->n_blocks > 0 && wg_lds <= 160 * 1024 && !force_one && (force_wg || B > 4 * m->cus);
#ifdef CET_PROF
  static unsigned long long* prof_dev = nullptr;
  const size_t prof_bytes = (size_t)PROF_SAMPLES * PROF_MAXM * sizeof(unsigned long long);
  if (!prof_dev) CET_HIP(hipMalloc(&prof_dev, prof_bytes));
  CET_HIP(hipMemsetAsync(prof_dev, 0, prof_bytes, stream));
  a.prof = prof_dev;
#endif
  const bool d2 = Ld > 16, m2 = a.mem_rows > 16;
  const KArgs& a_all = a;
  // Launches sequences [first, last) on one of the two launch shapes.
  auto launch = [&](const bool wg, const int first, const int last) -> int {
  KArgs a = a_all;
  a.first = first;
  a.last = last;
  const int B = last - first;
  if (wg) {
    a.wave_lds_floats = (int)(wg_wave_bytes / sizeof(float));
    a.region_rows = R_wg;
    const dim3 grid((unsigned)((B + WG_WAVES - 1) / WG_WAVES)), block(64 * WG_WAVES);
#define CET_LAUNCH_WG(RD_, RM_) \
  hipLaunchKernelGGL((informer_kernel<2, RD_, RM_, true, 4>), grid, block, wg_lds, stream, a)
    if (!d2 && !m2) CET_LAUNCH_WG(1, 1);
    else if (!d2 && m2) CET_LAUNCH_WG(1, 2);
    else if (d2 && !m2) CET_LAUNCH_WG(2, 1);
    else CET_LAUNCH_WG(2, 2);
#undef CET_LAUNCH_WG
  } else {
    // Q | K | V regions, K | V for the large shapes (queries in registers); either also covers
    // the conv / max-pool staging of 2 R + 2 rows of LDW floats
    a.wave_lds_floats = (large ? 2 : 3) * R * (16 * NHP + (LDW - 64));
    if (a.wave_lds_floats < (2 * R + 2) * LDW) a.wave_lds_floats = (2 * R + 2) * LDW;
    const size_t lds = (size_t)a.wave_lds_floats * sizeof(float);
    if (lds > 64 * 1024) return CET_ERR_UNSUPPORTED;  // cannot happen within supported()'s limits
    const dim3 grid((unsigned)B), block(64);
    // large shapes: decoder and memory row tiles as the model needs them (a 25-row decoder on
    // three tiles would issue 2.25 times the attention work of two)
    const int rd = Ld > 32 ? 3 : (Ld > 16 ? 2 : 1);
#define CET_LAUNCH_LARGE(RD_, RM_, NHP_, HPT_) \
  hipLaunchKernelGGL((informer_kernel<3, RD_, RM_, false, NHP_, HPT_>), grid, block, lds, stream, a)
#define CET_LAUNCH_LARGE_RM(RD_, NHP_, HPT_)            \
  do {                                                  \
    if (m2) CET_LAUNCH_LARGE(RD_, 2, NHP_, HPT_);       \
    else CET_LAUNCH_LARGE(RD_, 1, NHP_, HPT_);          \
  } while (0)
#define CET_LAUNCH_LARGE_RD(NHP_, HPT_)                      \
  do {                                                       \
    if (rd == 3) CET_LAUNCH_LARGE_RM(3, NHP_, HPT_);         \
    else if (rd == 2) CET_LAUNCH_LARGE_RM(2, NHP_, HPT_);    \
    else CET_LAUNCH_LARGE_RM(1, NHP_, HPT_);                 \
  } while (0)
    if (large && NHP == 6) CET_LAUNCH_LARGE_RD(6, 2);
    else if (large && heads_per_tile(m->cfg) == 3) CET_LAUNCH_LARGE_RD(4, 3);  // four tiles of three heads
    else if (large) CET_LAUNCH_LARGE_RD(4, 2);
#undef CET_LAUNCH_LARGE_RD
#undef CET_LAUNCH_LARGE_RM
#undef CET_LAUNCH_LARGE
    else if (!d2 && !m2) hipLaunchKernelGGL((informer_kernel<2, 1, 1, false, 4>), grid, block, lds, stream, a);
    else if (!d2 && m2) hipLaunchKernelGGL((informer_kernel<2, 1, 2, false, 4>), grid, block, lds, stream, a);
    else if (d2 && !m2) hipLaunchKernelGGL((informer_kernel<2, 2, 1, false, 4>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((informer_kernel<2, 2, 2, false, 4>), grid, block, lds, stream, a);
  }
  CET_HIP(hipGetLastError());
  return CET_OK;
  };
  // The automatic choice takes a batch of whole rounds of the workgroup kernel (8 sequences per CU)
  // plus up to 2 per CU in two launches: the rounds, then the rest on the one-wave kernel (0.17 to
  // 0.19 ms instead of another round's 0.25: 2,304 sequences 0.50 -> 0.44 ms, 4,352 0.76 -> 0.69;
  // with 4 per CU left over the gain is gone: 5,120 sequences 0.76 -> 0.80). The launches write
  // disjoint sequences of the same outputs.
  const int round = WG_WAVES * m->cus;
  if (wg && !force_wg && B % round > 0 && B % round <= 2 * m->cus) {
    if (const int rc = launch(true, 0, B / round * round)) return rc;
    if (const int rc = launch(false, B / round * round, B)) return rc;
  } else if (const int rc = launch(wg, 0, B)) {
    return rc;
  }
#ifdef CET_PROF
  {
    static unsigned long long h[PROF_SAMPLES * PROF_MAXM];
    CET_HIP(hipStreamSynchronize(stream));
    CET_HIP(hipMemcpy(h, prof_dev, sizeof h, hipMemcpyDeviceToHost));
    static const char* names[P_N] = {"embed", "qkv", "attn", "oproj", "add+ln", "ffn", "distil",
                                     "move", "proj", "other"};
    double sum[P_N] = {0}, tot = 0, t0min = 0, t1max = 0;
    int waves = 0;
    for (int s = 0; s < PROF_SAMPLES; ++s) {
      const unsigned long long* t = h + (size_t)s * PROF_MAXM;
      if (!t[0]) continue;
      ++waves;
      int n = 1;
      for (; n < PROF_MAXM && t[n]; ++n) sum[t[n] & 15] += (double)((t[n] >> 4) - (t[n - 1] >> 4));
      tot += (double)((t[n - 1] >> 4) - (t[0] >> 4));
      if (waves == 1 || (double)(t[0] >> 4) < t0min) t0min = (double)(t[0] >> 4);
      if ((double)(t[n - 1] >> 4) > t1max) t1max = (double)(t[n - 1] >> 4);
    }
    if (const char* dump = getenv("CET_PROF_DUMP")) {  // raw stamp traces, one line per sampled wave
      if (FILE* f = fopen(dump, "a")) {
        for (int s = 0; s < PROF_SAMPLES; ++s) {
          const unsigned long long* t = h + (size_t)s * PROF_MAXM;
          if (!t[0]) continue;
          fprintf(f, "B=%d wave", B);
          for (int n = 0; n < PROF_MAXM && t[n]; ++n) fprintf(f, " %llu:%d", (t[n] >> 4) - (t[0] >> 4), (int)(t[n] & 15));
          fprintf(f, "\n");
        }
        fclose(f);
      }
    }
    if (waves) {
      fprintf(stderr, "[cet prof] %d sampled waves, clocks per wave:", waves);
      for (int i = 0; i < P_N; ++i) fprintf(stderr, " %s=%.0f", names[i], sum[i] / waves);
      fprintf(stderr, " total=%.0f span=%.0f\n", tot / waves, t1max - t0min);
    }
  }
#endif
  return CET_OK;
}

}  // namespace cet
