// Recurrent channel predictors (RNN / GRU / LSTM `test_data` rollout) for gfx950.
// Restates LocalSimulations/models/model.py:298-312, 373-388, 449-465 (rollout) and the Unit
// forwards at :253-270, :327-344, :403-420 as ONE persistent kernel per call: a workgroup
// takes 32 sequences and runs every time step of every layer without leaving the chip.
//
//   * 4 waves per workgroup; wave w owns hidden units [w*H/4, (w+1)*H/4) of every layer and
//     computes ALL gates of those units, so the cell update is lane-local: the cell state c
//     (LSTM) and the wave's own slice of h stay in registers for the whole rollout.
//   * Gate pre-activations are gates^T[unit][sequence] = W * [x ; h]^T on
//     v_mfma_f32_32x32x2_f32 (A = weight rows, B = activations, fp32 exact). The full h of a
//     layer (all waves need it as B operand) lives in LDS, [sequence][H+4] floats, read as
//     conflict-free ds_read_b128 (4 k-steps per read).
//   * Weights are re-packed once at create time into the exact order a wave consumes them:
//     1 KiB per (k-block of 8, unit tile, gate), one contiguous stream per wave that covers a
//     whole time step, so the dominant traffic (the whole weight set, once per time step per
//     workgroup, from L2) is perfectly linear 16 B/lane loads. A ring of D k-blocks of
//     registers keeps D loads in flight at all times: a slot is refilled right after its
//     MFMAs were issued, across unit tiles, layers, barriers and time steps (the stream wraps),
//     so L2 latency is paid once per launch instead of once per matrix.
//   * The encoder Linear is folded into the first layer's input weights at create time
//     (W_ih0 (We x + be) = (W_ih0 We) x + W_ih0 be, nothing non-linear sits between the two;
//     products and sums in double, rounded once): the first layer takes the step's features
//     straight from registers, and a time step has no encoder stage and no barrier for it.
//   * Biases live in LDS for the whole rollout, b_ih + b_hh summed at create time where the cell
//     adds them anyway (every gate but the GRU's n).
//   * The decoder Linear consumes the wave's fresh h straight from the accumulator registers
//     (k = own units), the four partial sums meet in LDS; the next step consumes that result
//     from registers again. In the closed-loop phase nothing but the predictions ever goes to
//     HBM.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#include "cet_internal.h"

namespace cet {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BT = 32;       // sequences per workgroup
constexpr int FP = 32;       // features / encoder width, padded
constexpr int PSTR = FP + 4; // row stride of the partial-sum / x staging buffers

#define DEV __device__ __forceinline__

// -DCET_PROF: every wave of 16 sampled workgroups accumulates shader-clock ticks per phase of a
// time step and the host prints the averages after the launch (which it then waits for).
#ifdef CET_PROF
constexpr int RP_CATS = 8, RP_SLOTS = 16, RP_WORDS = RP_CATS + 2;
#define RP_MARK(cat)                                              \
  {                                                               \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();   \
    rp_acc[cat] += n_ - rp_t;                                     \
    rp_t = n_;                                                    \
  }
#else
#define RP_MARK(cat)
#endif

DEV f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
DEV f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
DEV void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// Gate non-linearities on the hardware exp2 / rcp (1 ulp each): absolute error ~1e-7, the size of
// the fp32 rounding the recurrence carries anyway. exp2 of a large argument gives inf and
// rcp(inf) = 0, so both saturate correctly.
DEV float sigmoidf_(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
}
DEV float tanhf_(float v) {
  const float t = __builtin_amdgcn_exp2f(-2.0f * 1.44269504088896340736f * fabsf(v));  // e^(-2|v|)
  return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), v);
}

template <int KIND> struct Cell;
template <> struct Cell<CET_RNN_TANH> { static constexpr int G = 1, NACC = 1; };
template <> struct Cell<CET_RNN_GRU> { static constexpr int G = 3, NACC = 4; };   // r, z, n_x, n_h
template <> struct Cell<CET_RNN_LSTM> { static constexpr int G = 4, NACC = 4; };  // i, f, g, o

struct RArgs {
  const float* wstream;  // packed gate weights, [wave][layer][unit tile][x|h][k-block][gate][64][4]
  // [layer][2][G*H]: b_ih + b_hh | unused, but for the GRU's n gate b_in | b_hn; the first layer's
  // b_ih includes the folded encoder bias
  const float* bias;
  const float* dec_w;    // [FP][H] zero padded (row = feature)
  const float* dec_b;    // [FP]
  const float* x;        // [B][seq][F]
  float* out;            // [B][pred][F]
  int B, seq_len, pred_len, F;
  // Conditional launch: when set, the workgroups return at once unless this word is non-zero (the
  // failure flag of the hidden-split rollout launched just before on the same stream).
  const unsigned* only_if;
  unsigned long long* prof;         // profiling build only
  // shared half-tile kernel only: exchange area [tile][layer][parity][sequence][H], arrival counters
  // [tile][layer] (zero at launch; the header's last word is the failure flag), polls before a wait gives up
  float* hx;
  unsigned* cnt;
  int spin_limit;
};

// C/D row of register r for lane half hh (v_mfma_f32_32x32x2_f32): (r&3) + 8*(r>>2) + 4*hh.
// Registers 4q..4q+3 are 4 consecutive rows starting at 8*q + 4*hh.

// One k-block (8 k values = 4 MFMA k-steps) for every gate of one unit tile.
// PART_H: the block belongs to W_hh (selects the accumulator the GRU n gate feeds).
template <int KIND, bool PART_H>
DEV void kblock(f32x16 (&acc)[Cell<KIND>::NACC], const f32x4 (&wt)[Cell<KIND>::G],
                const f32x4 bf) {
  constexpr int G = Cell<KIND>::G;
#pragma unroll
  for (int gt = 0; gt < G; ++gt) {
    const int ai = (KIND == CET_RNN_GRU && gt == 2 && PART_H) ? 3 : gt;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[ai] = mfma32(wt[gt][i], bf[i], acc[ai]);
  }
}

// k-blocks of weight tiles in flight per wave, by gate count (1 tanh cell, 3 GRU, 4 LSTM): the
// MFMAs of D-1 k-blocks (64 * 4 * G cycles each) cover the L2 latency of a refill (~2500 cycles).
#ifndef CET_RNN_D1
#define CET_RNN_D1 16
#endif
#ifndef CET_RNN_D3
#define CET_RNN_D3 4
#endif
#ifndef CET_RNN_D4
#define CET_RNN_D4 4
#endif
constexpr int ring_depth(int kind) {
  return kind == CET_RNN_TANH ? CET_RNN_D1 : (kind == CET_RNN_GRU ? CET_RNN_D3 : CET_RNN_D4);
}

// The wave's weight stream as a ring of D register sets, fetched with buffer loads: the
// descriptor covers the wave's stream of one time step, the lane offset is a constant VGPR, the
// position in the stream a scalar byte offset (`off`, the next k-block to fetch) that wraps at
// the end of the stream, and the G tiles of a k-block sit in the instruction's immediate offset.
template <int G, int D>
struct WRing {
  f32x4 w[D][G];
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned lane_off;    // 16 * lane
  unsigned off, total;  // bytes
};

template <int G, int D>
DEV void ring_refill(WRing<G, D>& r, const int d) {
#pragma unroll
  for (int t = 0; t < G; ++t)
    r.w[d][t] = __builtin_bit_cast(
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, r.lane_off + t * 1024, r.off, 0));
  r.off += G * 1024;
  if (r.off == r.total) r.off = 0;
}

// Keeps the compiler's scheduler from moving anything across this point: the refill of a ring
// slot must stay behind the MFMAs that read the slot, or it needs a second register set.
DEV void pin_order() { __builtin_amdgcn_sched_barrier(0); }

// acc += W_seg * act^T over one segment of the stream of nblk k-blocks (a multiple of D). On
// entry ring slot d holds the segment's block d. `act`: this lane's activation row (+4*hh) in
// LDS, read D k-blocks ahead as well.
template <int KIND, bool PART_H, int D>
DEV void segment(f32x16 (&acc)[Cell<KIND>::NACC], WRing<Cell<KIND>::G, D>& r, const float* act,
                 const int nblk) {
  f32x4 b[D];
#pragma unroll
  for (int d = 0; d < D; ++d) b[d] = ld4(act + 8 * d);
  for (int kb = 0; kb < nblk; kb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      kblock<KIND, PART_H>(acc, r.w[d], b[d]);
      ring_refill(r, d);
      if (kb + D + d < nblk) b[d] = ld4(act + 8 * (kb + D + d));
      pin_order();
    }
  }
}

// The layer-0 input segment: D k-blocks in the stream (so that every segment is a multiple of
// the ring depth) of which the first nvalid <= 4 carry weights; the rest is fetched and dropped.
// b: the step's features as they sit in registers (k-block q <-> features 8 q + 4 hh .. + 3).
template <int KIND, int D>
DEV void segment_input(f32x16 (&acc)[Cell<KIND>::NACC], WRing<Cell<KIND>::G, D>& r,
                       const f32x4 (&b)[4], const int nvalid) {
  constexpr int NB = D < 4 ? D : 4;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < NB) {
      if (d < nvalid) kblock<KIND, false>(acc, r.w[d], b[d < NB ? d : 0]);
    }
    ring_refill(r, d);
    pin_order();
  }
}

template <int KIND, int UT, int NL>
__global__ __launch_bounds__(256) void rnn_rollout_kernel(const RArgs a) {
  constexpr int H = 128 * UT;
  constexpr int G = Cell<KIND>::G, NACC = Cell<KIND>::NACC;
  constexpr int HS = H + 4;        // LDS row stride of an h buffer
  constexpr int TILE = 256;        // floats per packed weight tile (64 lanes x 4)
  constexpr bool LSTM = KIND == CET_RNN_LSTM;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* part = smem;                       // [4][BT][PSTR] decoder partial sums
  float* hbuf = part + 4 * BT * PSTR;       // [NL][BT][HS]
  float* lbias = hbuf + NL * BT * HS;       // [NL][2][G*H]  (see RArgs::bias)
  constexpr int D = ring_depth(KIND);
  static_assert((H / 8) % D == 0, "segment lengths must be multiples of the ring depth");

  if (a.only_if != nullptr &&
      __hip_atomic_load(a.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
    return;  // uniform over the grid: the rollout this launch stands by for has succeeded
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;   // sequence inside the tile (B operand row / C column)
  const int hh = lane >> 5;  // k half / C row block
  const int b0 = blockIdx.x * BT;
  const int bj = (b0 + j < a.B) ? b0 + j : a.B - 1;  // tail tile: replay the last sequence

  // weight ring: blocks per wave and time step = UT * (D + H/8) for layer 0 (input segment
  // padded to D) + UT * 2 * H/8 for every further layer
  WRing<G, D> ring;
  ring.total = (unsigned)(UT * (D + H / 8) + (NL - 1) * UT * 2 * (H / 8)) * G * TILE * 4;
  ring.rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wstream) + (size_t)wave * (ring.total / 4), 0, ring.total, 0x00020000);
  ring.lane_off = lane * 16;
  ring.off = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) ring_refill(ring, d);

  for (int i = tid; i < (BT * PSTR * 4 + NL * BT * HS) / 4; i += 256)
    st4(smem + 4 * i, f32x4{0, 0, 0, 0});
  for (int i = tid; i < NL * 2 * G * H / 4; i += 256) st4(lbias + 4 * i, ld4(a.bias + 4 * i));

  f32x16 cst[LSTM ? NL : 1][LSTM ? UT : 1];  // LSTM cell state of this wave's units
#pragma unroll
  for (int l = 0; l < (LSTM ? NL : 1); ++l)
#pragma unroll
    for (int u = 0; u < (LSTM ? UT : 1); ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) cst[l][u][r] = 0.f;

  f32x4 xin[4];  // this step's input features, T layout: register q <-> features 8q+4hh..+3
#pragma unroll
  for (int q = 0; q < 4; ++q) xin[q] = f32x4{0, 0, 0, 0};
  auto load_x = [&](int idx, f32x4 (&dst)[4]) {
    const float* row = a.x + ((size_t)bj * a.seq_len + idx) * a.F;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (8 * q + 4 * hh < a.F) dst[q] = ld4(row + 8 * q + 4 * hh);
  };
  load_x(0, xin);
  __syncthreads();
#ifdef CET_PROF
  unsigned long long rp_acc[RP_CATS] = {};
  const unsigned long long rp_t0 = __builtin_amdgcn_s_memtime(), rp_w0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long rp_t = rp_t0;
#endif

  const int n_steps = a.seq_len + a.pred_len - 1;
  for (int idx = 0; idx < n_steps; ++idx) {
    // The weights do not change between steps, which invites the compiler to hoist hundreds of
    // loop-invariant loads (all biases, decoder rows) out of the time loop and pin them in
    // registers. Making the base pointers opaque per iteration keeps them where they are used.
    const float *dec_w = a.dec_w, *dec_b = a.dec_b;
    asm volatile("" : "+s"(dec_w), "+s"(dec_b));
    // ---- the next teacher-forced input, requested a step ahead ----
    f32x4 xnext[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xnext[q] = xin[q];
    if (idx + 1 < a.seq_len) load_x(idx + 1, xnext);
    RP_MARK(0)

    const bool decode = idx >= a.seq_len - 1;  // decoder output needed from here on
    f32x16 dec_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dec_acc[r] = 0.f;

    // ---- the layers ---------------------------------------------------------------------
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const float* in_buf = hbuf + (l > 0 ? l - 1 : 0) * BT * HS;  // the layer below (l > 0)
      float* hl = hbuf + l * BT * HS;
      const float* bi = lbias + l * 2 * G * H;
      const float* bh = bi + G * H;

      f32x16 hnew[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int u0 = wave * (32 * UT) + 32 * u;  // first unit of this tile
        f32x16 acc[NACC];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int off = u0 + 8 * q + 4 * hh;
#pragma unroll
          for (int gt = 0; gt < G; ++gt) {
            const f32x4 v1 = ld4(bi + gt * H + off);  // b_ih + b_hh, or b_in
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[gt][4 * q + i] = v1[i];
            if (KIND == CET_RNN_GRU && gt == 2) {  // n gate keeps W_in x + b_in apart from W_hn h + b_hn
              const f32x4 v2 = ld4(bh + gt * H + off);
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[3][4 * q + i] = v2[i];
            }
          }
        }
        RP_MARK(1)
        if (l == 0)  // ceil(features / 8) valid k-blocks in a segment of D
          segment_input<KIND, D>(acc, ring, xin, (a.F + 7) / 8);
        else
          segment<KIND, false, D>(acc, ring, in_buf + j * HS + 4 * hh, H / 8);
        RP_MARK(2)
        segment<KIND, true, D>(acc, ring, hl + j * HS + 4 * hh, H / 8);
        RP_MARK(3)

        // cell update, lane-local
        f32x16 hprev;
        if (KIND == CET_RNN_GRU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = ld4(hl + j * HS + u0 + 8 * q + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) hprev[4 * q + i] = v[i];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float hv;
          if (KIND == CET_RNN_TANH) {
            hv = tanhf_(acc[0][r]);
          } else if (KIND == CET_RNN_GRU) {
            const float rg = sigmoidf_(acc[0][r]);
            const float zg = sigmoidf_(acc[1][r]);
            const float ng = tanhf_(acc[2][r] + rg * acc[3][r]);
            hv = (1.0f - zg) * ng + zg * hprev[r];
          } else {
            const float ig = sigmoidf_(acc[0][r]);
            const float fg = sigmoidf_(acc[1][r]);
            const float gg = tanhf_(acc[2][r]);
            const float og = sigmoidf_(acc[3][r]);
            // two rounded products and a rounded sum, as the elementwise tensor ops of
            // nn.LSTM compute it (no contraction into an fma, whose choice of operand would
            // otherwise be left to the optimiser)
            const float cv = __fadd_rn(__fmul_rn(fg, cst[LSTM ? l : 0][LSTM ? u : 0][r]), __fmul_rn(ig, gg));
            cst[LSTM ? l : 0][LSTM ? u : 0][r] = cv;
            hv = og * tanhf_(cv);
          }
          hnew[u][r] = hv;
        }
        // decoder Linear partial over this tile's units, straight from the registers
        if (l == NL - 1 && decode) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 wv = ld4(dec_w + (size_t)j * H + u0 + 8 * q + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) dec_acc = mfma32(wv[i], hnew[u][4 * q + i], dec_acc);
          }
        }
      }
      RP_MARK(4)
      __syncthreads();  // every wave has finished reading the old h of this layer
      RP_MARK(5)
#pragma unroll
      for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          st4(hl + j * HS + wave * (32 * UT) + 32 * u + 8 * q + 4 * hh,
              f32x4{hnew[u][4 * q], hnew[u][4 * q + 1], hnew[u][4 * q + 2], hnew[u][4 * q + 3]});
      if (l == NL - 1 && decode) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          st4(part + (wave * BT + j) * PSTR + 8 * q + 4 * hh,
              f32x4{dec_acc[4 * q], dec_acc[4 * q + 1], dec_acc[4 * q + 2], dec_acc[4 * q + 3]});
      }
      __syncthreads();
      RP_MARK(6)
    }

    // ---- decoder output = bias + the four waves' partial sums (fixed order) ----------------
    if (decode) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 s = ld4(dec_b + 8 * q + 4 * hh);
#pragma unroll
        for (int w = 0; w < 4; ++w) s += ld4(part + (w * BT + j) * PSTR + 8 * q + 4 * hh);
        xnext[q] = s;  // next step's input, already in registers in every wave
      }
      if (wave == 1 && b0 + j < a.B) {
        float* dst = a.out + ((size_t)(b0 + j) * a.pred_len + (idx - (a.seq_len - 1))) * a.F;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (8 * q + 4 * hh < a.F) st4(dst + 8 * q + 4 * hh, xnext[q]);
      }
      // `part` is written again only after the barriers of the next step
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xin[q] = xnext[q];
    RP_MARK(7)
  }
#ifdef CET_PROF
  const int rp_stride = gridDim.x >= RP_SLOTS ? gridDim.x / RP_SLOTS : 1;
  if (blockIdx.x % rp_stride == 0 && blockIdx.x / rp_stride < RP_SLOTS && lane == 0) {
    unsigned long long* dst = a.prof + ((size_t)(blockIdx.x / rp_stride) * 4 + wave) * RP_WORDS;
    for (int i = 0; i < RP_CATS; ++i) dst[i] = rp_acc[i];
    dst[RP_CATS] = __builtin_amdgcn_s_memtime() - rp_t0;
    dst[RP_CATS + 1] = __builtin_amdgcn_s_memrealtime() - rp_w0;
  }
#endif
}


// ---- hidden-split rollout for small batches -------------------------------------------------
// The kernel above gives a workgroup 32 sequences and ALL hidden units: per time step it streams
// the whole weight set through one CU, so a rollout of 64 sequences (2 workgroups) takes as long
// as one of 8192. Below, the hidden units are split instead: workgroup (tile, ug) owns 16
// sequences and hidden units [16 ug, 16 ug + 16) of every layer, keeps exactly the weight rows of
// those units in REGISTERS for the whole rollout (4 waves split the k range; nothing is streamed
// per step) and exchanges hidden states with the H / 16 - 1 other workgroups of its sequence tile
// through a scratch buffer in global memory, once per layer and time step:
//   * gates^T[16 units][16 sequences] = W * [x ; h]^T on v_mfma_f32_16x16x4_f32, k split over the
//     4 waves, partial sums meet in LDS, wave 0 applies the cell update for the 256 (unit,
//     sequence) pairs and keeps the workgroup's own h / c slice in registers;
//   * publish: wave 0 stores the slice (1 KiB, whole 128-byte lines per store instruction) with
//     agent-scope (write-through) stores, waits for them, and one lane adds 1 to the arrival
//     counter of (tile, layer);
//   * fetch: one lane polls that counter with agent-scope loads until all H / 16 workgroups of
//     the tile have arrived at this step (bounded: a wait that does not end marks the workgroup
//     dead and every later wait returns at once, so the grid always drains), a workgroup barrier
//     follows, then every thread reads the published slices with agent-scope loads into LDS.
//   The slices of a step go to the buffer of the step's parity; a workgroup can be at most one
//   step ahead of the slowest one of its tile, so two buffers suffice.
//   * the encoder Linear is folded into the first layer's input weights (as in the kernel above)
//     and the decoder Linear is small and computed redundantly by every workgroup, so the
//     closed-loop feedback needs no further exchange; a teacher-forced step's input rows are
//     requested one step ahead.
// Only the H / 16 workgroups of ONE sequence tile wait for each other, and the hardware starts a
// grid's workgroups in index order (a tile's are consecutive), so a grid has at most one tile that
// is partly resident; every complete tile runs to its end and frees its CUs. A wait can therefore
// only fail to end while other work holds the CUs that the missing workgroups of a tile need:
//   * other kernels that finish by themselves only delay it;
//   * other hidden-split grids of this process are limited to SPLIT_SLOTS at a time per GPU (the
//     pool of exchange buffers below), i.e. to SPLIT_SLOTS partly resident tiles, far below the CU
//     count, so among them some tile is always complete;
//   * whatever remains (grids of other processes, a co-resident kernel that runs for seconds) ends
//     in the bounded wait: the grid drains, raises its failure flag, and the batch-tiled kernel,
//     launched behind every hidden-split launch on the same stream and returning at once while
//     the flag is down, computes the chunk instead. No path leaves a wrong result.
// The launcher only picks this kernel while a launch's grid fits the device's CU count.
constexpr int ST = 16;        // sequences per workgroup
// Head of the scratch buffer (zeroed before every launch): arrival counters, one per tile and layer
// (at most 32 tiles x 3 layers), each in a 128-byte line of its own (with all of them in one
// 512-byte block every poll and every arrival of the whole grid went to one memory channel), and in
// the last word the grid's failure flag.
constexpr int CNT_STRIDE = 32;                 // words between counters
constexpr size_t SPLIT_CNT_SLOTS = 767;        // counters the header has room for
constexpr size_t SPLIT_CNT_FLOATS = (SPLIT_CNT_SLOTS + 1) * CNT_STRIDE;  // 96 KiB
constexpr int SPIN_LIMIT = 1 << 17;  // polls of about a microsecond each: a tenth of a second or two, then the stand-by launch serves

DEV f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct SArgs {
  // [G*H][in_l], PyTorch layout, device; layer 0 with the encoder Linear folded in: [G*H][F]
  const float* w_ih[3];
  const float* w_hh[3];  // [G*H][H]
  const float* bias;     // [layer][2][G*H]   (b_ih, b_hh; layer 0's b_ih with the folded encoder bias)
  const float* dec_w;    // [FP][H] zero padded (row = feature)
  const float* dec_b;    // [FP]
  const float* x;        // [B][seq][F]
  float* out;            // [B][pred][F]
  float* hx;             // exchange: [tile][layer][parity][unit group][sequence][16 units]
  unsigned* cnt;         // arrival counters [tile][layer], zero at launch
  int B, seq_len, pred_len, F;
  int spin_limit;        // polls after which a wait gives up
  unsigned long long* prof;  // profiling build only: [workgroup][SP_WORDS] phase times (10 ns units)
};
#ifdef CET_PROF
// phases of an exchange, wave 0 lane 0: compute (from the last fetch's end to the publish store),
// store acknowledged, counter wait, loads + LDS copy
constexpr int SP_WORDS = 8;
#define SP_NOW() __builtin_amdgcn_s_memrealtime()
#endif

// UW: hidden units per workgroup (16, 8 or 4), P = H / UW workgroups per sequence tile. The rows of
// a layer's gate product are NACC groups of UW units (tanh cell: h; GRU: r, z, n from the input, n
// from the state; LSTM: i, f, g, o), row = group * UW + unit, in tiles of 16 rows. With 16 units a
// group is a tile; with 4 units the four groups of a GRU or LSTM share one tile, so a workgroup
// issues a quarter of the MFMAs per layer step (128 instead of 384 / 512 at 256 hidden units) and a
// batch of up to 64 sequences spreads over four times as many CUs. The launcher picks the
// smallest UW whose grid fits the device.
// TPW: sequence tiles per workgroup (1 or 2). A workgroup's weights serve any tile, so with two it
// takes a layer step of the first, then of the second, and only then waits for the first one's
// exchange, which the second one's matrix work has covered meanwhile.
template <int KIND, int H, int NL, int UW, int TPW>
__global__ __launch_bounds__(256, (UW < 16 && TPW == 1) ? 2 : 1) void rnn_split_kernel(const SArgs a) {
  constexpr int G = Cell<KIND>::G, NACC = Cell<KIND>::NACC;
  constexpr bool LSTM = KIND == CET_RNN_LSTM, GRU = KIND == CET_RNN_GRU;
  constexpr int P = H / UW;            // workgroups per sequence tile
  // Arrival counters of a (tile, layer): one per 16 workgroups, each in a line of its own (64 arrivals
  // on one word took 0.55 us longer per layer step than 16)
  constexpr int NS = P > 16 ? P / 16 : 1;
  constexpr int UPL = UW / 4;          // units per lane in the cell update: lane (c, g) <-> units g UPL + e
  constexpr int NT = (NACC * UW + 15) / 16;  // row tiles of the gate product
  // 16 units: the GRU's two n groups (input part, state part) have the same rows, so they share a
  // set of weight registers; a block of the input feeds one accumulator, a block of the state the other
  constexpr bool SHARE_N = GRU && UW == 16;
  constexpr int NTW = SHARE_N ? 3 : NT;  // weight register sets per k block
  constexpr int NA = SHARE_N ? 4 : NT;   // accumulators
  constexpr int HB = H / 16;           // k blocks of 16 in a hidden state
  constexpr int HS = H + 4;            // LDS row stride of a hidden state
  constexpr int XS = FP + 4;           // LDS row stride of the input rows
  constexpr int XB = FP / 16;          // k blocks of the (padded) features
  constexpr int JX = (XB + HB + 3) / 4;  // blocks of layer 0 per wave
  constexpr int JH = (2 * HB + 3) / 4;   // blocks of a higher layer per wave
  static_assert(UW == 4 || UW == 8 || UW == 16, "units per workgroup");

  __shared__ __attribute__((aligned(16))) float xcur[TPW][ST * XS];   // this step's input features (layer-0 input)
  __shared__ __attribute__((aligned(16))) float hfull[TPW][NL][ST * HS];
  __shared__ __attribute__((aligned(16))) float part[TPW][4][NA > 2 ? NA : 2][256];
  __shared__ __attribute__((aligned(16))) float pub[ST * UW];    // a slice on its way out (UW < 16)
  __shared__ int dead;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // workgroup (tg, ug) takes tiles tg TPW + ts, ts < TPW, as far as the batch has them
  const int tg = blockIdx.x / P, ug = blockIdx.x % P;
  const int n_tiles = (a.B + ST - 1) / ST;

  // ---- this wave's weights, in registers for the whole rollout --------------------------------
  // Block b of a layer covers k = 16 b .. 16 b + 15 of [input ; hidden]; wave w owns blocks
  // 4 j + w. Lane (c, g) holds, per row tile m, W[row 16 m + c][k = 16 b + 4 g + i], i = 0..3:
  // zero for rows beyond the groups and, where the GRU's n groups have rows of their own, for the
  // input-part group in a block of the state and the other way round.
  auto wload = [&](int l, int m, bool x_block, int k0, int in) {
    const int r = 16 * m + c, grp = r / UW, unit = r % UW;
    int gate = grp;
    bool valid = r < NACC * UW;
    if (GRU) {
      gate = grp < 2 ? grp : 2;
      if (!SHARE_N) valid = valid && !(grp == 2 && !x_block) && !(grp == 3 && x_block);
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int width = x_block ? in : H;  // a multiple of 4
    if (valid && k0 + 4 * g < width) {
      const size_t row = (size_t)gate * H + UW * ug + unit;
      v = ld4((x_block ? a.w_ih[l] + row * in : a.w_hh[l] + row * H) + k0 + 4 * g);
    }
    return v;
  };
  f32x4 w0[JX][NTW], wh[NL > 1 ? NL - 1 : 1][JH][NTW];
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int b = 4 * j + wave;
#pragma unroll
    for (int m = 0; m < NTW; ++m)
      w0[j][m] = b < XB ? wload(0, m, true, 16 * b, a.F)
                        : (b < XB + HB ? wload(0, m, false, 16 * (b - XB), a.F) : f32x4{0.f, 0.f, 0.f, 0.f});
  }
#pragma unroll
  for (int l = 1; l < NL; ++l)
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      const int b = 4 * j + wave;
#pragma unroll
      for (int m = 0; m < NTW; ++m)
        wh[l - 1][j][m] = b < HB ? wload(l, m, true, 16 * b, H)
                                 : (b < 2 * HB ? wload(l, m, false, 16 * (b - HB), H) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
  // decoder rows (features) x this wave's k blocks of the top hidden state
  f32x4 wd[XB][HB / 4];
#pragma unroll
  for (int ft = 0; ft < XB; ++ft)
#pragma unroll
    for (int j = 0; j < HB / 4; ++j)
      wd[ft][j] = ld4(a.dec_w + (size_t)(16 * ft + c) * H + 16 * (4 * j + wave) + 4 * g);
  f32x4 bd[XB];
#pragma unroll
  for (int it = 0; it < XB; ++it) bd[it] = ld4(a.dec_b + 16 * it + 4 * g);
  // biases of the own units (wave 0 applies them): lane (c, g) <-> units UW ug + g UPL + e
  float bsum[NL][NACC][UPL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const float* bi = a.bias + (size_t)l * 2 * G * H;
    const float* bh = bi + G * H;
#pragma unroll
    for (int gt = 0; gt < G; ++gt)
#pragma unroll
      for (int e = 0; e < UPL; ++e) {
        const int u = gt * H + UW * ug + g * UPL + e;
        if (GRU && gt == 2) {
          bsum[l][2][e] = bi[u];  // n gate: W_in x + b_in apart from W_hn h + b_hn
          bsum[l][3][e] = bh[u];
        } else {
          bsum[l][gt][e] = bi[u] + bh[u];
        }
      }
  }

  for (int i = tid; i < TPW * NL * ST * HS; i += 256) (&hfull[0][0][0])[i] = 0.f;
  for (int i = tid; i < TPW * ST * XS; i += 256) (&xcur[0][0])[i] = 0.f;
  if (tid == 0) dead = 0;
  // wave 0: own slices of h (and c), units g UPL + e of sequence c of tile slot ts
  float hown[TPW][NL][UPL], cown[TPW][NL][UPL];
#pragma unroll
  for (int ts = 0; ts < TPW; ++ts)
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int e = 0; e < UPL; ++e) hown[ts][l][e] = cown[ts][l][e] = 0.f;
  __syncthreads();

  // A tile's exchange area [layer][parity][unit group][sequence][UW units] through a buffer
  // descriptor: 16-byte agent-scope (sc1, write-through / L1-bypassing) stores and loads, a quarter
  // of the instructions of 4-byte ones and, for the stores, a sixth of their time per byte.
  constexpr int AGENT_SCOPE = 16;  // sc1
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t hx_rsrc[TPW];
  unsigned* cnt_tile[TPW];  // [layer][shard]
  int b0[TPW];
  bool live[TPW];           // the batch has this tile
#pragma unroll
  for (int ts = 0; ts < TPW; ++ts) {
    const int tile = tg * TPW + ts;
    live[ts] = tile < n_tiles;
    b0[ts] = tile * ST;
    hx_rsrc[ts] = __builtin_amdgcn_make_buffer_rsrc(a.hx + (size_t)tile * NL * 2 * H * ST, 0,
                                                    live[ts] ? NL * 2 * H * ST * 4 : 0, 0x00020000);
    cnt_tile[ts] = a.cnt + (size_t)tile * NL * NS * CNT_STRIDE;
  }
  unsigned* const fail = a.cnt + (SPLIT_CNT_FLOATS - 1);  // set by any workgroup whose wait ran out

#ifdef CET_PROF
  unsigned long long sp[SP_WORDS] = {};
  unsigned long long sp_t = SP_NOW();
#define SP_MARK(i) { const unsigned long long n_ = SP_NOW(); sp[i] += n_ - sp_t; sp_t = n_; }
#else
#define SP_MARK(i)
#endif
  // The arrival at a layer's counter follows the acknowledgement of the slice's stores. Waiting for
  // that right behind the stores would hold wave 0 for half a microsecond, so the arrival is put
  // off (pend[ts]: the layer, wave 0): to the middle of the next matrix product, where the
  // acknowledgement has long come, at the latest to the next publication or to the fetch that needs
  // it. For the top layer of a step whose prediction is not needed nobody waits before then.
  int pend[TPW];
#pragma unroll
  for (int ts = 0; ts < TPW; ++ts) pend[ts] = -1;
  auto arrive_flush = [&](const int only) __attribute__((always_inline)) {  // only: a tile slot, or -1 for all
    if (wave == 0) {
      bool any = false;
#pragma unroll
      for (int ts = 0; ts < TPW; ++ts) any = any || ((only < 0 || only == ts) && pend[ts] >= 0);
      if (any) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ts = 0; ts < TPW; ++ts)
          if ((only < 0 || only == ts) && pend[ts] >= 0) {
            if (lane == 0)
              __hip_atomic_fetch_add(cnt_tile[ts] + (pend[ts] * NS + ug % NS) * CNT_STRIDE, 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
            pend[ts] = -1;
          }
      }
    }
  };
  // Waits until every workgroup of tile slot ts has published layer l of step t, then copies the
  // layer's full hidden state into hfull[ts][l].
  // With l2 >= 0 the same for layer l2 of step t2 in the same round trip (the top layer's state of
  // the step before, whose wait was put off until its first use).
  auto fetch = [&](const int ts, const int l, const int t, const int l2 = -1, const int t2 = 0)
                   __attribute__((always_inline)) {
    arrive_flush(ts);  // the own arrival first: the tile's other workgroups wait for it as this one for theirs
    SP_MARK(2)  // publish -> fetch entry
    if (wave == 0 && !dead) {  // lane s < NS polls shard s; the wave leaves when every shard has arrived
      const unsigned target = (unsigned)(P / NS) * (unsigned)(t + 1), target2 = (unsigned)(P / NS) * (unsigned)(t2 + 1);
      const int sh = lane < NS ? lane : 0;
      int spins = 0;
      for (;;) {
        bool ok = __hip_atomic_load(cnt_tile[ts] + (l * NS + sh) * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
        if (l2 >= 0)
          ok = ok && __hip_atomic_load(cnt_tile[ts] + (l2 * NS + sh) * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target2;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > a.spin_limit) {
          if (lane == 0) {
            dead = 1;
            __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          break;
        }
      }
    }
    __syncthreads();
    SP_MARK(3)  // counter wait + barrier
    // A layer's state is H ST / 4 pieces of 16 bytes; piece p = (unit group, sequence, four units)
    // in the order they lie in the exchange area; thread tid takes pieces tid + 256 r.
    constexpr int NPC = H * ST / 4 / 256;
    const unsigned voff = (unsigned)(tid * 4) * 4u;
    const unsigned so = (unsigned)((l * 2 + (t & 1)) * H * ST) * 4u;
    const unsigned so2 = (unsigned)(((l2 >= 0 ? l2 : 0) * 2 + (t2 & 1)) * H * ST) * 4u;
    u32x4 v[NPC], v2[NPC];  // all loads of the round in flight together: one memory round trip
#pragma unroll
    for (int r = 0; r < NPC; ++r)
      v[r] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc[ts], voff + (unsigned)(r * 4 * 256 * 4), so, AGENT_SCOPE);
    if (l2 >= 0) {
#pragma unroll
      for (int r = 0; r < NPC; ++r)
        v2[r] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc[ts], voff + (unsigned)(r * 4 * 256 * 4), so2, AGENT_SCOPE);
    }
    // another slot's arrival that is still pending: its acknowledgement comes with these loads
    arrive_flush(-1);
    int at[NPC];
#pragma unroll
    for (int r = 0; r < NPC; ++r) {
      const int p = tid + 256 * r, q = p % (4 * UW);
      at[r] = (q / UPL) * HS + UW * (p / (4 * UW)) + 4 * (q % UPL);
    }
#pragma unroll
    for (int r = 0; r < NPC; ++r) st4(&hfull[ts][l][at[r]], __builtin_bit_cast(f32x4, v[r]));
    if (l2 >= 0) {
#pragma unroll
      for (int r = 0; r < NPC; ++r) st4(&hfull[ts][l2][at[r]], __builtin_bit_cast(f32x4, v2[r]));
    }
    __syncthreads();
    SP_MARK(4)  // loads + LDS copy + barrier
  };

  // Teacher-forced input features: wave 0 requests a step's rows one step ahead (ST rows of up to
  // FP floats: two 16-byte pieces per lane), so their memory latency is off the step's critical path.
  constexpr int XPL = ST * (FP / 4) / 64;
  const int f4n = a.F / 4;  // 16-byte pieces per row
  f32x4 xpre[TPW][XPL];
  auto request_x = [&](const int ts, const int t) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < XPL; ++r) {
      const int q = lane + 64 * r;
      if (q < ST * f4n) {
        const int sq = q / f4n, f4 = q % f4n;
        const int bs = b0[ts] + sq < a.B ? b0[ts] + sq : a.B - 1;  // tail tile: replay the last sequence
        xpre[ts][r] = ld4(a.x + ((size_t)bs * a.seq_len + t) * a.F + 4 * f4);
      }
    }
  };
  if (wave == 0) {
#pragma unroll
    for (int ts = 0; ts < TPW; ++ts)
      if (live[ts]) request_x(ts, 0);
  }

  const int n_steps = a.seq_len + a.pred_len - 1;
  bool pending_top[TPW];
#pragma unroll
  for (int ts = 0; ts < TPW; ++ts) pending_top[ts] = false;
  for (int t = 0; t < n_steps; ++t) {
    // ---- input features of a teacher-forced step, wave 0 (a closed-loop step finds the decoder's
    // output of the step before in xcur) -------------------------------------------------------
    if (wave == 0 && t < a.seq_len) {
#pragma unroll
      for (int ts = 0; ts < TPW; ++ts)
        if (live[ts]) {
#pragma unroll
          for (int r = 0; r < XPL; ++r) {
            const int q = lane + 64 * r;
            if (q < ST * f4n) st4(xcur[ts] + (q / f4n) * XS + 4 * (q % f4n), xpre[ts][r]);
          }
          if (t + 1 < a.seq_len) request_x(ts, t + 1);
        }
    }
    __syncthreads();

#pragma unroll
    for (int l = 0; l < NL; ++l) {
#pragma unroll
      for (int ts = 0; ts < TPW; ++ts) {
        if (!live[ts]) continue;
        if (l == NL - 1 && pending_top[ts]) {  // only when no lower layer's fetch took it along (NL == 1)
          fetch(ts, NL - 1, t - 1);
          pending_top[ts] = false;
        }
        // ---- partial gate sums over this wave's k blocks -------------------------------------------
        f32x4 acc[NA];
#pragma unroll
        for (int q = 0; q < NA; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // one k block of 16: B operand bv, weights wt[row tile], is_x: the block belongs to the layer's
        // input; the row tiles take turns so that consecutive MFMAs do not wait for each other
        auto block = [&](const f32x4 bv, const f32x4 (&wt)[NTW], const bool is_x) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m < NTW; ++m) {
              if (SHARE_N && m == 2) {  // n gate: W_in x apart from W_hn h
                if (is_x) acc[2] = mfma16(wt[m][i], bv[i], acc[2]);
                else acc[NA - 1] = mfma16(wt[m][i], bv[i], acc[NA - 1]);
              } else {
                acc[m] = mfma16(wt[m][i], bv[i], acc[m]);
              }
            }
        };
        if (l == 0) {
#pragma unroll
          for (int j = 0; j < JX; ++j) {
            const int b = 4 * j + wave;
            if (b < XB + HB) {
              const bool is_x = b < XB;
              block(is_x ? ld4(xcur[ts] + c * XS + 16 * b + 4 * g) : ld4(hfull[ts][0] + c * HS + 16 * (b - XB) + 4 * g),
                    w0[j], is_x);
            }
            if (j == JX / 2) arrive_flush(-1);
          }
        } else {
#pragma unroll
          for (int j = 0; j < JH; ++j) {
            const int b = 4 * j + wave;
            if (b < 2 * HB) {
              const bool is_x = b < HB;
              block(is_x ? ld4(hfull[ts][l - 1] + c * HS + 16 * b + 4 * g) : ld4(hfull[ts][l] + c * HS + 16 * (b - HB) + 4 * g),
                    wh[l > 0 ? l - 1 : 0][j], is_x);
            }
            if (j == JH / 2) arrive_flush(-1);
          }
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) st4(&part[ts][wave][q][4 * lane], acc[q]);
        __syncthreads();
        // ---- cell update of the own UW x 16 slice and its publication, wave 0 ------------------------
        if (wave == 0) {
          // group q of this lane's units: rows q UW + g UPL + e of the gate product, i.e. row rr of tile
          // m, which the lanes (rr / 4, c) of every wave hold in register rr % 4 of accumulator m
          float s[NACC][UPL];
#pragma unroll
          for (int q = 0; q < NACC; ++q) {
            const int r0 = q * UW + g * UPL, m = r0 / 16, rr = r0 % 16;
            const int off = 4 * (16 * (rr / 4) + c) + rr % 4;
#pragma unroll
            for (int e = 0; e < UPL; ++e)
              s[q][e] = ((part[ts][0][m][off + e] + part[ts][1][m][off + e]) +
                         (part[ts][2][m][off + e] + part[ts][3][m][off + e])) + bsum[l][q][e];
          }
          float hn[UPL];
#pragma unroll
          for (int e = 0; e < UPL; ++e) {
            float hv;
            if (KIND == CET_RNN_TANH) {
              hv = tanhf_(s[0][e]);
            } else if (GRU) {
              const float rg = sigmoidf_(s[0][e]);
              const float zg = sigmoidf_(s[1][e]);
              const float ng = tanhf_(s[2][e] + rg * s[NACC - 1][e]);
              hv = (1.0f - zg) * ng + zg * hown[ts][l][e];
            } else {
              const float ig = sigmoidf_(s[0][e]);
              const float fg = sigmoidf_(s[1][e]);
              const float gg = tanhf_(s[LSTM ? 2 : 0][e]);
              const float og = sigmoidf_(s[LSTM ? 3 : 0][e]);
              const float cv = __fadd_rn(__fmul_rn(fg, cown[ts][l][e]), __fmul_rn(ig, gg));
              cown[ts][l][e] = cv;
              hv = og * tanhf_(cv);
            }
            hn[e] = hv;
            hown[ts][l][e] = hv;
          }
          SP_MARK(0)  // matrix work, partial sums, cell update
          arrive_flush(-1);  // an arrival of the same counter must not overtake an earlier one's
          // the slice [sequence][UW units] goes out in 16-byte pieces, its 64 UW bytes in whole 128-byte
          // lines by one store instruction
          const unsigned slice = (unsigned)((l * 2 + (t & 1)) * H * ST + ug * UW * ST) * 4u;
          if (UW == 16) {
            const f32x4 hv4 = {hn[0], hn[UPL > 1 ? 1 : 0], hn[UPL > 2 ? 2 : 0], hn[UPL > 3 ? 3 : 0]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv4), hx_rsrc[ts],
                                                   (unsigned)(c * 16 + 4 * g) * 4u, slice, AGENT_SCOPE);
          } else {
            // a lane holds fewer than four units: regrouped through LDS (the wave's own accesses stay in order)
#pragma unroll
            for (int e = 0; e < UPL; ++e) pub[c * UW + g * UPL + e] = hn[e];
            if (lane < 4 * UW)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ld4(pub + 4 * lane)), hx_rsrc[ts],
                                                     (unsigned)lane * 16u, slice, AGENT_SCOPE);
          }
          pend[ts] = l;
          SP_MARK(1)  // slice stored
        }
      }
#pragma unroll
      for (int ts = 0; ts < TPW; ++ts) {
        if (!live[ts]) continue;
        if (l < NL - 1) {
          if (l == NL - 2 && pending_top[ts]) {  // the layer above needs both: one wait, one round trip
            fetch(ts, l, t, NL - 1, t - 1);
            pending_top[ts] = false;
          } else {
            fetch(ts, l, t);
          }
        } else {
          pending_top[ts] = true;
        }
      }
    }

    // ---- decoder Linear on the new top hidden state: the prediction and the next input --------------
    if (t >= a.seq_len - 1) {
#pragma unroll
      for (int ts = 0; ts < TPW; ++ts) {
        if (!live[ts]) continue;
        fetch(ts, NL - 1, t);
        pending_top[ts] = false;
        f32x4 dacc[XB];
#pragma unroll
        for (int ft = 0; ft < XB; ++ft) dacc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < HB / 4; ++j) {
          const f32x4 bv = ld4(hfull[ts][NL - 1] + c * HS + 16 * (4 * j + wave) + 4 * g);
#pragma unroll
          for (int ft = 0; ft < XB; ++ft)
#pragma unroll
            for (int i = 0; i < 4; ++i) dacc[ft] = mfma16(wd[ft][j][i], bv[i], dacc[ft]);
        }
#pragma unroll
        for (int ft = 0; ft < XB; ++ft) st4(&part[ts][wave][ft][4 * lane], dacc[ft]);
        __syncthreads();
        if (wave == 0) {
#pragma unroll
          for (int ft = 0; ft < XB; ++ft) {
            const f32x4 y = ((ld4(&part[ts][0][ft][4 * lane]) + ld4(&part[ts][1][ft][4 * lane])) +
                             (ld4(&part[ts][2][ft][4 * lane]) + ld4(&part[ts][3][ft][4 * lane]))) + bd[ft];
            st4(xcur[ts] + c * XS + 16 * ft + 4 * g, y);  // features 16 ft + 4 g + i of sequence c
            const int f0 = 16 * ft + 4 * g;
            if (ug == 0 && b0[ts] + c < a.B && f0 < a.F)
              st4(a.out + ((size_t)(b0[ts] + c) * a.pred_len + (t - (a.seq_len - 1))) * a.F + f0, y);
          }
        }
        // the next step's barrier orders these LDS writes before layer 0 reads them and the reads of
        // `part` before it is written again
      }
    }
  }
  arrive_flush(-1);  // nobody waits for it; the counters are simply complete
#ifdef CET_PROF
  if (tid == 0 && a.prof) {
    SP_MARK(5)
    for (int i = 0; i < SP_WORDS; ++i) a.prof[(size_t)blockIdx.x * SP_WORDS + i] = sp[i];
  }
#endif
  // A wait that ran out anywhere in the grid (the workgroups were not all resident: the GPU was
  // shared with other work) leaves garbage in the exchange. The predictions are then overwritten
  // with NaN, so that such a rollout cannot pass for a result.
  __syncthreads();
  if (ug == 0 && (dead || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
    const int per_seq = a.pred_len * a.F;
#pragma unroll
    for (int ts = 0; ts < TPW; ++ts)
      for (int i = tid; i < ST * per_seq; i += 256) {
        const int sq = i / per_seq;
        if (b0[ts] + sq < a.B) a.out[(size_t)(b0[ts] + sq) * per_seq + (i - sq * per_seq)] = __builtin_nanf("");
      }
  }
}

// ---- half-tile rollout for batches that leave half of the CUs idle ---------------------------
// The batch-tiled kernel takes 32 sequences per workgroup and as long for 2,100 sequences (66 of
// 256 CUs busy) as for 8,192. This form takes 16 (v_mfma_f32_16x16x4_f32: the sequences are the N
// dimension) and is otherwise the same design: wave w owns the hidden units [w H / 4, (w + 1) H / 4)
// of every layer, all gates, in tiles of 16 units; the wave's weights are one linear stream per
// time step through a ring of D k-blocks of registers (a k-block here is 16 k values of 16 units:
// 1 KiB per gate, four MFMAs); the cell update is lane-local; h of a layer lives in LDS for the B
// operand; the decoder Linear takes the fresh h from registers. A workgroup issues half the MFMAs
// per step and streams the same weights, so it needs twice the L2 bandwidth per CU (32 B / clock).
constexpr int MT = 16;  // sequences per workgroup
// A tile's hidden units go on S workgroups only while each keeps a unit tile of 16 per wave.
static bool mid_share_ok(int H, int S) { return S == 1 || (H / S >= 64 && S <= 8); }
// (a divisor of H / 16: 16 for the tanh cell from 256 hidden units on, else 8, 4 at 64 hidden units)
constexpr int mid_ring_depth(int kind, int H) { return H < 128 ? 4 : (kind == CET_RNN_TANH && H >= 256 ? 16 : 8); }

template <int KIND, bool PART_H>
DEV void kblock16(f32x4 (&acc)[Cell<KIND>::NACC], const f32x4 (&wt)[Cell<KIND>::G], const f32x4 bf) {
  constexpr int G = Cell<KIND>::G;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int gt = 0; gt < G; ++gt) {
      const int ai = (KIND == CET_RNN_GRU && gt == 2 && PART_H) ? 3 : gt;
      acc[ai] = mfma16(wt[gt][i], bf[i], acc[ai]);
    }
}

// acc += W_seg * act^T over nblk k-blocks of 16 (a multiple of D); `act`: this lane's activation
// row (+ 4 g) in LDS. On entry ring slot d holds the segment's block d.
template <int KIND, bool PART_H, int D>
DEV void segment16(f32x4 (&acc)[Cell<KIND>::NACC], WRing<Cell<KIND>::G, D>& r, const float* act,
                   const int nblk) {
  f32x4 b[D];
#pragma unroll
  for (int d = 0; d < D; ++d) b[d] = ld4(act + 16 * d);
  for (int kb = 0; kb < nblk; kb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      kblock16<KIND, PART_H>(acc, r.w[d], b[d]);
      ring_refill(r, d);
      if (kb + D + d < nblk) b[d] = ld4(act + 16 * (kb + D + d));
      pin_order();
    }
  }
}

// The layer-0 input segment: D k-blocks in the stream of which the first nvalid <= 2 carry weights.
template <int KIND, int D>
DEV void segment_input16(f32x4 (&acc)[Cell<KIND>::NACC], WRing<Cell<KIND>::G, D>& r,
                         const f32x4 (&b)[2], const int nvalid) {
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < 2) {
      if (d < nvalid) kblock16<KIND, false>(acc, r.w[d], b[d < 2 ? d : 0]);
    }
    ring_refill(r, d);
    pin_order();
  }
}

template <int KIND, int H, int NL>
__global__ __launch_bounds__(256) void rnn_mid_kernel(const RArgs a) {
  constexpr int G = Cell<KIND>::G, NACC = Cell<KIND>::NACC;
  constexpr int UTW = H / 64;      // unit tiles of 16 per wave
  constexpr int HS = H + 4;        // LDS row stride of an h buffer
  constexpr int XB = FP / 16;      // k-blocks of the (padded) features
  constexpr bool LSTM = KIND == CET_RNN_LSTM, GRU = KIND == CET_RNN_GRU;
  constexpr int D = mid_ring_depth(KIND, H);
  static_assert((H / 16) % D == 0, "segment lengths must be multiples of the ring depth");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* part = smem;                       // [4][MT][PSTR] decoder partial sums
  float* hbuf = part + 4 * MT * PSTR;       // [NL][MT][HS]
  float* lbias = hbuf + NL * MT * HS;       // [NL][2][G*H]  (see RArgs::bias)

  if (a.only_if != nullptr &&
      __hip_atomic_load(a.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
    return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;  // sequence inside the tile; k group / unit group
  const int b0 = blockIdx.x * MT;
  const int bc = (b0 + c < a.B) ? b0 + c : a.B - 1;  // tail tile: replay the last sequence

  WRing<G, D> ring;
  ring.total = (unsigned)(UTW * (D + H / 16) + (NL - 1) * UTW * 2 * (H / 16)) * G * 1024;
  ring.rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wstream) + (size_t)wave * (ring.total / 4), 0, ring.total, 0x00020000);
  ring.lane_off = lane * 16;
  ring.off = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) ring_refill(ring, d);

  for (int i = tid; i < (4 * MT * PSTR + NL * MT * HS) / 4; i += 256) st4(smem + 4 * i, f32x4{0, 0, 0, 0});
  for (int i = tid; i < NL * 2 * G * H / 4; i += 256) st4(lbias + 4 * i, ld4(a.bias + 4 * i));

  // decoder rows (features 16 ft + c) x this wave's units, for the whole rollout
  f32x4 wd[UTW][XB];
#pragma unroll
  for (int u = 0; u < UTW; ++u)
#pragma unroll
    for (int ft = 0; ft < XB; ++ft)
      wd[u][ft] = ld4(a.dec_w + (size_t)(16 * ft + c) * H + wave * (16 * UTW) + 16 * u + 4 * g);
  f32x4 bd[XB];
#pragma unroll
  for (int ft = 0; ft < XB; ++ft) bd[ft] = ld4(a.dec_b + 16 * ft + 4 * g);

  f32x4 cst[LSTM ? NL : 1][LSTM ? UTW : 1];  // LSTM cell state of this wave's units
#pragma unroll
  for (int l = 0; l < (LSTM ? NL : 1); ++l)
#pragma unroll
    for (int u = 0; u < (LSTM ? UTW : 1); ++u) cst[l][u] = f32x4{0, 0, 0, 0};

  f32x4 xin[XB];  // this step's input: features 16 q + 4 g .. + 3 of sequence c
#pragma unroll
  for (int q = 0; q < XB; ++q) xin[q] = f32x4{0, 0, 0, 0};
  auto load_x = [&](int idx, f32x4 (&dst)[XB]) {
    const float* row = a.x + ((size_t)bc * a.seq_len + idx) * a.F;
#pragma unroll
    for (int q = 0; q < XB; ++q)
      if (16 * q + 4 * g < a.F) dst[q] = ld4(row + 16 * q + 4 * g);
  };
  load_x(0, xin);
  __syncthreads();

  const int n_steps = a.seq_len + a.pred_len - 1;
  for (int idx = 0; idx < n_steps; ++idx) {
    f32x4 xnext[XB];
#pragma unroll
    for (int q = 0; q < XB; ++q) xnext[q] = xin[q];
    if (idx + 1 < a.seq_len) load_x(idx + 1, xnext);
    const bool decode = idx >= a.seq_len - 1;
    f32x4 dec_acc[XB];
#pragma unroll
    for (int ft = 0; ft < XB; ++ft) dec_acc[ft] = f32x4{0, 0, 0, 0};

#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const float* in_buf = hbuf + (l > 0 ? l - 1 : 0) * MT * HS;  // the layer below (l > 0)
      float* hl = hbuf + l * MT * HS;
      const float* bi = lbias + l * 2 * G * H;
      const float* bh = bi + G * H;
      f32x4 hnew[UTW];
#pragma unroll
      for (int u = 0; u < UTW; ++u) {
        const int u0 = wave * (16 * UTW) + 16 * u;  // first unit of this tile
        f32x4 acc[NACC];
#pragma unroll
        for (int gt = 0; gt < G; ++gt) acc[gt] = ld4(bi + gt * H + u0 + 4 * g);  // b_ih + b_hh, or b_in
        if (GRU) acc[NACC - 1] = ld4(bh + 2 * H + u0 + 4 * g);                  // b_hn
        if (l == 0)
          segment_input16<KIND, D>(acc, ring, xin, (a.F + 15) / 16);
        else
          segment16<KIND, false, D>(acc, ring, in_buf + c * HS + 4 * g, H / 16);
        segment16<KIND, true, D>(acc, ring, hl + c * HS + 4 * g, H / 16);

        f32x4 hprev = {0, 0, 0, 0};
        if (GRU) hprev = ld4(hl + c * HS + u0 + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float hv;
          if (KIND == CET_RNN_TANH) {
            hv = tanhf_(acc[0][i]);
          } else if (GRU) {
            const float rg = sigmoidf_(acc[0][i]);
            const float zg = sigmoidf_(acc[GRU ? 1 : 0][i]);
            const float ng = tanhf_(acc[GRU ? 2 : 0][i] + rg * acc[NACC - 1][i]);
            hv = (1.0f - zg) * ng + zg * hprev[i];
          } else {
            const float ig = sigmoidf_(acc[0][i]);
            const float fg = sigmoidf_(acc[LSTM ? 1 : 0][i]);
            const float gg = tanhf_(acc[LSTM ? 2 : 0][i]);
            const float og = sigmoidf_(acc[LSTM ? 3 : 0][i]);
            const float cv = __fadd_rn(__fmul_rn(fg, cst[LSTM ? l : 0][LSTM ? u : 0][i]), __fmul_rn(ig, gg));
            cst[LSTM ? l : 0][LSTM ? u : 0][i] = cv;
            hv = og * tanhf_(cv);
          }
          hnew[u][i] = hv;
        }
        // decoder Linear partial over this tile's units, straight from the registers
        if (l == NL - 1 && decode) {
#pragma unroll
          for (int ft = 0; ft < XB; ++ft)
#pragma unroll
            for (int i = 0; i < 4; ++i) dec_acc[ft] = mfma16(wd[u][ft][i], hnew[u][i], dec_acc[ft]);
        }
      }
      __syncthreads();  // every wave has finished reading the old h of this layer
#pragma unroll
      for (int u = 0; u < UTW; ++u) st4(hl + c * HS + wave * (16 * UTW) + 16 * u + 4 * g, hnew[u]);
      if (l == NL - 1 && decode) {
#pragma unroll
        for (int ft = 0; ft < XB; ++ft) st4(part + (wave * MT + c) * PSTR + 16 * ft + 4 * g, dec_acc[ft]);
      }
      __syncthreads();
    }

    if (decode) {
#pragma unroll
      for (int q = 0; q < XB; ++q) {
        f32x4 sm = bd[q];
#pragma unroll
        for (int w = 0; w < 4; ++w) sm += ld4(part + (w * MT + c) * PSTR + 16 * q + 4 * g);
        xnext[q] = sm;
      }
      if (wave == 1 && b0 + c < a.B) {
        float* dst = a.out + ((size_t)(b0 + c) * a.pred_len + (idx - (a.seq_len - 1))) * a.F;
#pragma unroll
        for (int q = 0; q < XB; ++q)
          if (16 * q + 4 * g < a.F) st4(dst + 16 * q + 4 * g, xnext[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < XB; ++q) xin[q] = xnext[q];
  }
}

// ---- the half-tile rollout with a tile's hidden units shared by S workgroups --------------------
// Between the hidden-split kernel's launches and one half tile per CU (about 600 to 2,000 sequences
// at 256 hidden units) neither fills the device. Here workgroup (tile, s) takes the hidden units
// [s H / S, (s + 1) H / S) of a 16-sequence tile: the half-tile kernel's streamed weights and
// lane-local cell update on a 1 / S share of the units, and after every layer step the S workgroups
// exchange their slices of h through global memory as the hidden-split kernel's do (agent-scope
// 16-byte stores, every storing wave's wait, a workgroup barrier, one arrival per workgroup on the
// (tile, layer) counter; a bounded poll, a barrier, agent-scope loads of the full state into LDS;
// buffers alternate by step parity; a wait that runs out raises the failure flag and the batch-tiled
// kernel, launched behind on stand-by, recomputes the batch). The decoder Linear needs all hidden
// units, so every workgroup computes it from the fetched state, k split over its waves.
template <int KIND, int H, int NL, int S>
__global__ __launch_bounds__(256) void rnn_mid_shared_kernel(const RArgs a) {
  constexpr int G = Cell<KIND>::G, NACC = Cell<KIND>::NACC;
  constexpr int HU = H / S;        // hidden units of this workgroup
  constexpr int UTW = HU / 64;     // unit tiles of 16 per wave
  constexpr int HS = H + 4;        // LDS row stride of an h buffer
  constexpr int XB = FP / 16;      // k-blocks of the (padded) features
  constexpr int HB = H / 16;       // k-blocks of a hidden state
  constexpr bool LSTM = KIND == CET_RNN_LSTM, GRU = KIND == CET_RNN_GRU;
  constexpr int D = mid_ring_depth(KIND, H);
  static_assert(UTW >= 1 && (H / 16) % D == 0, "16 units per wave at least; segments in multiples of the ring depth");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* part = smem;                       // [4][MT][PSTR] decoder partial sums
  float* hbuf = part + 4 * MT * PSTR;       // [NL][MT][HS]
  float* lbias = hbuf + NL * MT * HS;       // [NL][2][G*H]  (see RArgs::bias)
  __shared__ int dead;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x / S, sh = blockIdx.x % S;
  const int b0 = tile * MT;
  const int bc = (b0 + c < a.B) ? b0 + c : a.B - 1;
  const int unit_w = sh * HU + wave * (16 * UTW);  // first unit of this wave

  WRing<G, D> ring;
  ring.total = (unsigned)(UTW * (D + H / 16) + (NL - 1) * UTW * 2 * (H / 16)) * G * 1024;
  ring.rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wstream) + (size_t)(sh * 4 + wave) * (ring.total / 4), 0, ring.total, 0x00020000);
  ring.lane_off = lane * 16;
  ring.off = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) ring_refill(ring, d);

  for (int i = tid; i < (4 * MT * PSTR + NL * MT * HS) / 4; i += 256) st4(smem + 4 * i, f32x4{0, 0, 0, 0});
  for (int i = tid; i < NL * 2 * G * H / 4; i += 256) st4(lbias + 4 * i, ld4(a.bias + 4 * i));
  if (tid == 0) dead = 0;

  // decoder rows (features 16 ft + c) x this wave's k-blocks 4 j + wave of the top hidden state
  f32x4 wd[XB][HB / 4];
#pragma unroll
  for (int ft = 0; ft < XB; ++ft)
#pragma unroll
    for (int j = 0; j < HB / 4; ++j)
      wd[ft][j] = ld4(a.dec_w + (size_t)(16 * ft + c) * H + 16 * (4 * j + wave) + 4 * g);
  f32x4 bd[XB];
#pragma unroll
  for (int ft = 0; ft < XB; ++ft) bd[ft] = ld4(a.dec_b + 16 * ft + 4 * g);

  f32x4 cst[LSTM ? NL : 1][LSTM ? UTW : 1];
#pragma unroll
  for (int l = 0; l < (LSTM ? NL : 1); ++l)
#pragma unroll
    for (int u = 0; u < (LSTM ? UTW : 1); ++u) cst[l][u] = f32x4{0, 0, 0, 0};

  f32x4 xin[XB];
#pragma unroll
  for (int q = 0; q < XB; ++q) xin[q] = f32x4{0, 0, 0, 0};
  auto load_x = [&](int idx, f32x4 (&dst)[XB]) {
    const float* row = a.x + ((size_t)bc * a.seq_len + idx) * a.F;
#pragma unroll
    for (int q = 0; q < XB; ++q)
      if (16 * q + 4 * g < a.F) dst[q] = ld4(row + 16 * q + 4 * g);
  };
  load_x(0, xin);

  constexpr int AGENT_SCOPE = 16;  // sc1
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.hx + (size_t)tile * NL * 2 * H * MT, 0, NL * 2 * H * MT * 4, 0x00020000);
  unsigned* const cnt_tile = a.cnt + (size_t)tile * NL * CNT_STRIDE;
  unsigned* const fail = a.cnt + (SPLIT_CNT_FLOATS - 1);
  __syncthreads();

  // Waits until all S workgroups of the tile have published layer l of step t (and, with l2 >= 0,
  // layer l2 of step t2), then copies the full state(s) into LDS.
  auto fetch = [&](const int l, const int t, const int l2 = -1, const int t2 = 0) __attribute__((always_inline)) {
    if (tid == 0 && !dead) {
      const unsigned target = (unsigned)S * (unsigned)(t + 1), target2 = (unsigned)S * (unsigned)(t2 + 1);
      int spins = 0;
      while (__hip_atomic_load(cnt_tile + l * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target ||
             (l2 >= 0 &&
              __hip_atomic_load(cnt_tile + l2 * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target2)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > a.spin_limit) {
          dead = 1;
          __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    // a layer's full state: H MT / 4 pieces of 16 bytes, thread tid takes pieces tid + 256 r
    constexpr int NPC = H * MT / 4 / 256;
    const unsigned so = (unsigned)((l * 2 + (t & 1)) * H * MT) * 4u;
    const unsigned so2 = (unsigned)(((l2 >= 0 ? l2 : 0) * 2 + (t2 & 1)) * H * MT) * 4u;
    u32x4 v[NPC], v2[NPC];
#pragma unroll
    for (int r = 0; r < NPC; ++r)
      v[r] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, (unsigned)(tid + 256 * r) * 16u, so, AGENT_SCOPE);
    if (l2 >= 0) {
#pragma unroll
      for (int r = 0; r < NPC; ++r)
        v2[r] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, (unsigned)(tid + 256 * r) * 16u, so2, AGENT_SCOPE);
    }
#pragma unroll
    for (int r = 0; r < NPC; ++r) {
      const int p = tid + 256 * r;
      st4(hbuf + l * MT * HS + (p / (H / 4)) * HS + 4 * (p % (H / 4)), __builtin_bit_cast(f32x4, v[r]));
      if (l2 >= 0)
        st4(hbuf + l2 * MT * HS + (p / (H / 4)) * HS + 4 * (p % (H / 4)), __builtin_bit_cast(f32x4, v2[r]));
    }
    __syncthreads();
  };

  const int n_steps = a.seq_len + a.pred_len - 1;
  bool pending_top = false;
  for (int idx = 0; idx < n_steps; ++idx) {
    f32x4 xnext[XB];
#pragma unroll
    for (int q = 0; q < XB; ++q) xnext[q] = xin[q];
    if (idx + 1 < a.seq_len) load_x(idx + 1, xnext);
    const bool decode = idx >= a.seq_len - 1;

#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const float* in_buf = hbuf + (l > 0 ? l - 1 : 0) * MT * HS;
      float* hl = hbuf + l * MT * HS;
      const float* bi = lbias + l * 2 * G * H;
      const float* bh = bi + G * H;
      f32x4 hnew[UTW];
#pragma unroll
      for (int u = 0; u < UTW; ++u) {
        const int u0 = unit_w + 16 * u;
        f32x4 acc[NACC];
#pragma unroll
        for (int gt = 0; gt < G; ++gt) acc[gt] = ld4(bi + gt * H + u0 + 4 * g);
        if (GRU) acc[NACC - 1] = ld4(bh + 2 * H + u0 + 4 * g);
        if (l == 0)
          segment_input16<KIND, D>(acc, ring, xin, (a.F + 15) / 16);
        else
          segment16<KIND, false, D>(acc, ring, in_buf + c * HS + 4 * g, H / 16);
        segment16<KIND, true, D>(acc, ring, hl + c * HS + 4 * g, H / 16);

        f32x4 hprev = {0, 0, 0, 0};
        if (GRU) hprev = ld4(hl + c * HS + u0 + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float hv;
          if (KIND == CET_RNN_TANH) {
            hv = tanhf_(acc[0][i]);
          } else if (GRU) {
            const float rg = sigmoidf_(acc[0][i]);
            const float zg = sigmoidf_(acc[GRU ? 1 : 0][i]);
            const float ng = tanhf_(acc[GRU ? 2 : 0][i] + rg * acc[NACC - 1][i]);
            hv = (1.0f - zg) * ng + zg * hprev[i];
          } else {
            const float ig = sigmoidf_(acc[0][i]);
            const float fg = sigmoidf_(acc[LSTM ? 1 : 0][i]);
            const float gg = tanhf_(acc[LSTM ? 2 : 0][i]);
            const float og = sigmoidf_(acc[LSTM ? 3 : 0][i]);
            const float cv = __fadd_rn(__fmul_rn(fg, cst[LSTM ? l : 0][LSTM ? u : 0][i]), __fmul_rn(ig, gg));
            cst[LSTM ? l : 0][LSTM ? u : 0][i] = cv;
            hv = og * tanhf_(cv);
          }
          hnew[u][i] = hv;
        }
      }
      // ---- the own slice goes out: [sequence][H] rows of the (layer, parity) area ----
      const unsigned so = (unsigned)((l * 2 + (idx & 1)) * H * MT) * 4u;
#pragma unroll
      for (int u = 0; u < UTW; ++u)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hnew[u]), hx_rsrc,
                                               (unsigned)(c * H + unit_w + 16 * u + 4 * g) * 4u, so, AGENT_SCOPE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier the arrival follows
      __syncthreads();  // ... which also says that every wave has finished reading the old h of this layer
      if (tid == 0)
        __hip_atomic_fetch_add(cnt_tile + l * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // The top layer's state is needed by the decoder (closed-loop steps) or by the next step's top
      // layer: in a teacher-forced step its wait is put off behind the next step's lower layers and
      // shares their round trip (as in the hidden-split kernel).
      if (l < NL - 1) {
        if (l == NL - 2 && pending_top) fetch(l, idx, NL - 1, idx - 1);
        else fetch(l, idx);
        if (l == NL - 2) pending_top = false;
      } else if (decode || NL == 1) {
        fetch(NL - 1, idx);
      } else {
        pending_top = true;
      }
    }

    // ---- decoder Linear on the new top hidden state: the prediction and the next input ----
    if (decode) {
      f32x4 dacc[XB];
#pragma unroll
      for (int ft = 0; ft < XB; ++ft) dacc[ft] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < HB / 4; ++j) {
        const f32x4 bv = ld4(hbuf + (NL - 1) * MT * HS + c * HS + 16 * (4 * j + wave) + 4 * g);
#pragma unroll
        for (int ft = 0; ft < XB; ++ft)
#pragma unroll
          for (int i = 0; i < 4; ++i) dacc[ft] = mfma16(wd[ft][j][i], bv[i], dacc[ft]);
      }
#pragma unroll
      for (int ft = 0; ft < XB; ++ft) st4(part + (wave * MT + c) * PSTR + 16 * ft + 4 * g, dacc[ft]);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < XB; ++q) {
        f32x4 sm = bd[q];
#pragma unroll
        for (int w = 0; w < 4; ++w) sm += ld4(part + (w * MT + c) * PSTR + 16 * q + 4 * g);
        xnext[q] = sm;
      }
      if (sh == 0 && wave == 1 && b0 + c < a.B) {
        float* dst = a.out + ((size_t)(b0 + c) * a.pred_len + (idx - (a.seq_len - 1))) * a.F;
#pragma unroll
        for (int q = 0; q < XB; ++q)
          if (16 * q + 4 * g < a.F) st4(dst + 16 * q + 4 * g, xnext[q]);
      }
      // `part` is written again only behind the barriers of the next step's layers
    }
#pragma unroll
    for (int q = 0; q < XB; ++q) xin[q] = xnext[q];
  }
}

// Scratch of the hidden-split kernel: arrival counters (one per tile and layer, in a block of
// their own that is zeroed before every launch), then the exchange slices
// [tile][layer][parity][unit group][256 floats].
constexpr int SPLIT_MAX_WG = 256;    // workgroups the static buffer below serves (tiles x H / 16)
constexpr int SPLIT_MAX_LAYERS = 3;
constexpr size_t SPLIT_MAX_FLOATS = SPLIT_CNT_FLOATS + (size_t)SPLIT_MAX_WG * SPLIT_MAX_LAYERS * 2 * 256;

// The buffer of the entry point that allocates nothing (cet_rnn_rollout_static): 1.6 MiB per GPU
// the library runs on, held until the process ends. Host code never uses this name as an
// address; rnn_create() asks the runtime for the device address once per handle.
static __device__ __align__(256) float g_split_scratch[SPLIT_MAX_FLOATS];

}  // namespace

// ---------------------------------------------------------------------------------------
// Host side

struct RnnModel {
  int kind, F, I, H, layers;
  bool core = true;      // rnn_core_shape(): every kernel serves it, else the half-tile kernel alone
  float* dev = nullptr;  // one allocation; the pointers below point into it
  // bias: SArgs::bias (hidden-split kernel); bias_tiled: RArgs::bias
  const float *wstream, *wstream_mid, *bias, *bias_tiled, *dec_w, *dec_b;
  const float* wstream_shared[4] = {nullptr, nullptr, nullptr, nullptr};  // shared half-tile kernel, S = 2, 4, 8 at [1], [2], [3]
  // gate weights in their PyTorch layouts (hidden-split kernel); w_ih[0] with the encoder folded in
  const float *w_ih[3], *w_hh[3];
  int device = 0, cus = 0;
  // 0: by batch size, 1: batch-tiled kernel, 2: hidden-split kernel, 3: hidden-split kernel whose
  // exchange gives up at its first unanswered poll (tests: exercises the stand-by launch), 4 / 5 / 6:
  // half-tile kernel with a tile's hidden units on 1 / 2 / 4 workgroups, 7: as 5, giving up like 3
  int variant = 0;
  float* static_scratch = nullptr;  // device address of g_split_scratch on this handle's device
  // workgroups of the hidden-split kernel that a CU holds at once, for 4, 8 and 16 hidden units per
  // workgroup and for two tiles per workgroup (registers and LDS decide; at most SPLIT_MAX_OCC are used)
  int split_occ[4] = {1, 1, 1, 1};
  int split_spill[4] = {0, 0, 0, 0};  // scratch bytes per lane of the same instantiations
  int split_chunk = 0;      // sequences one launch of the hidden-split kernel takes at most
  // estimated microseconds per time step of the batch-tiled kernel (any batch up to 32 per CU)
  double est_tiled_step = 0.0;
};
constexpr int SPLIT_MAX_OCC = 2;

// Exchange buffers of the hidden-split kernel: SPLIT_SLOTS per GPU, shared by all handles on it,
// allocated by the first rollout that needs one and kept until the process ends. A rollout claims
// a slot for its launches and leaves an event behind them: the next claim takes the slot its own
// stream used last, else one whose event has completed, else makes its stream wait for the oldest
// slot's event. The pool's mutex is held from the claim until that event is recorded, so a second
// host thread never sees a claimed slot with the event of its previous use. Rollouts of any
// handles on any streams may therefore overlap; at most SPLIT_SLOTS hidden-split grids are in
// flight per GPU.
constexpr int SPLIT_SLOTS = 4;
constexpr int SPLIT_MAX_DEVICES = 64;
struct SplitSlot {
  float* buf = nullptr;
  hipEvent_t done = nullptr;
  hipStream_t last = nullptr;
  bool used = false;
  int cus = 0;  // CUs the launches behind `done` need for all their workgroups to be resident at once
};
struct SplitPool {
  std::mutex mu;
  SplitSlot slot[SPLIT_SLOTS];
  size_t slot_bytes = 0;
  unsigned next = 0;  // round robin over busy slots
};
static SplitPool g_split_pool[SPLIT_MAX_DEVICES];

static int rnn_prepare(const RnnModel* m);  // per-device kernel attributes, defined with the launchers
static int rnn_split_occupancy(RnnModel* m);  // fills split_occ, defined with the launchers
static void rnn_split_limits(RnnModel* m);    // fills split_chunk from split_occ

static int gates_of(int kind) { return kind == CET_RNN_TANH ? 1 : (kind == CET_RNN_GRU ? 3 : 4); }

// Every kernel serves hidden 128 / 256 with up to 3 layers ("core" shapes); hidden 64 / 512 and 4
// layers run on the half-tile kernel alone, whatever the batch.
static bool rnn_core_shape(int H, int layers) { return (H == 128 || H == 256) && layers <= 3; }
static bool rnn_supported(int kind, int F, int I, int H, int layers) {
  if (kind < CET_RNN_TANH || kind > CET_RNN_LSTM) return false;
  if (H != 64 && H != 128 && H != 256 && H != 512) return false;
  if (layers < 1 || layers > (H == 512 ? 3 : 4)) return false;  // 512 x 4 layers: the h buffers exceed the LDS
  if (F < 4 || F > FP || F % 4) return false;
  if (I < 1 || I > 4096) return false;  // folded into the first layer at create time: any width
  return true;
}

void rnn_destroy(RnnModel* m) {
  if (!m) return;
  if (m->dev) (void)hipFree(m->dev);
  delete m;
}

size_t rnn_blob_size(int kind, int F, int I, int H, int layers) {
  if (kind < CET_RNN_TANH || kind > CET_RNN_LSTM || F < 1 || I < 1 || H < 1 || layers < 1) return 0;
  const size_t G = gates_of(kind);
  size_t n = (size_t)I * F + I;
  for (int l = 0; l < layers; ++l) {
    const size_t in = l == 0 ? I : H;
    n += G * H * in + G * H * H + 2 * G * H;
  }
  return n + (size_t)F * H + F;
}

int rnn_create(int kind, int F, int I, int H, int layers, const float* w, size_t n, RnnModel** out) {
  if (!rnn_supported(kind, F, I, H, layers)) return CET_ERR_UNSUPPORTED;
  if (n != rnn_blob_size(kind, F, I, H, layers)) return CET_ERR_BLOB_SIZE;
  const int G = gates_of(kind);
  const int UT = H / 128;

  const float* p = w;
  const float* enc_w = p; p += (size_t)I * F;
  const float* enc_b = p; p += I;
  const bool core = rnn_core_shape(H, layers);
  const int D = ring_depth(kind);
  if (core && (D < 4 || (H / 8) % D)) return CET_ERR_UNSUPPORTED;  // up to 4 k-blocks of features in a segment of D
  struct LayerW { const float *w_ih, *w_hh; int in; };
  std::vector<LayerW> lw;
  std::vector<float> stream, bias, bias_tiled;
  // The encoder Linear folded into the first layer for the batch-tiled kernel:
  // W_ih0 (We x + be) + b_ih0 = (W_ih0 We) x + (W_ih0 be + b_ih0), accumulated in double.
  std::vector<float> w_fold((size_t)G * H * F), b_fold((size_t)G * H);
  for (int l = 0; l < layers; ++l) {
    const int in = l == 0 ? I : H;          // input width
    const float* w_ih = p; p += (size_t)G * H * in;
    const float* w_hh = p; p += (size_t)G * H * H;
    const float* b_ih = p; p += (size_t)G * H;
    const float* b_hh = p; p += (size_t)G * H;
    if (l == 0) {
      for (int r = 0; r < G * H; ++r) {
        double bs = b_ih[r];
        for (int k = 0; k < I; ++k) bs += (double)w_ih[(size_t)r * I + k] * (double)enc_b[k];
        b_fold[r] = (float)bs;
        for (int c = 0; c < F; ++c) {
          double ws = 0.0;
          for (int k = 0; k < I; ++k) ws += (double)w_ih[(size_t)r * I + k] * (double)enc_w[(size_t)k * F + c];
          w_fold[(size_t)r * F + c] = (float)ws;
        }
      }
    }
    // b_ih + b_hh where the cell adds them anyway; the GRU's n gate (rows 2 H .. 3 H) keeps both
    const float* bi = l == 0 ? b_fold.data() : b_ih;
    bias.insert(bias.end(), bi, bi + (size_t)G * H);
    bias.insert(bias.end(), b_hh, b_hh + (size_t)G * H);
    for (int part = 0; part < 2; ++part)
      for (int r = 0; r < G * H; ++r) {
        const bool apart = kind == CET_RNN_GRU && r >= 2 * H;
        bias_tiled.push_back(part == 0 ? (apart ? bi[r] : bi[r] + b_hh[r]) : (apart ? b_hh[r] : 0.f));
      }
    lw.push_back({w_ih, w_hh, in});
  }
  // one stream per wave in consumption order: [layer][unit tile][input segment | hidden segment]
  // [k-block][gate][64 lanes][4]; the layer-0 input segment is padded to D k-blocks
  for (int wave = 0; core && wave < 4; ++wave)
    for (int l = 0; l < layers; ++l)
      for (int u = 0; u < UT; ++u)
        for (int part = 0; part < 2; ++part) {
          const float* W = part == 0 ? (l == 0 ? w_fold.data() : lw[l].w_ih) : lw[l].w_hh;
          const int K = part == 0 ? (l == 0 ? F : H) : H;
          const int nblk = part == 0 ? (l == 0 ? D : H / 8) : H / 8;
          for (int kb = 0; kb < nblk; ++kb)
            for (int g = 0; g < G; ++g)
              for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                  const int unit = wave * 32 * UT + 32 * u + (lane & 31);
                  const int k = 8 * kb + 4 * (lane >> 5) + i;
                  stream.push_back(k < K ? W[((size_t)g * H + unit) * K + k] : 0.f);
                }
        }
  // the same for the half-tile kernel: unit tiles of 16, k-blocks of 16, [64 lanes][4] with lane
  // (c, g) <-> W[unit0 + c][16 kb + 4 g + i]; the layer-0 input segment is padded to its ring depth
  std::vector<float> stream_mid;
  {
    const int DM_ = mid_ring_depth(kind, H), UTW = H / 64;
    for (int wave = 0; wave < 4; ++wave)
      for (int l = 0; l < layers; ++l)
        for (int u = 0; u < UTW; ++u)
          for (int part = 0; part < 2; ++part) {
            const float* W = part == 0 ? (l == 0 ? w_fold.data() : lw[l].w_ih) : lw[l].w_hh;
            const int K = part == 0 ? (l == 0 ? F : H) : H;
            const int nblk = part == 0 ? (l == 0 ? DM_ : H / 16) : H / 16;
            for (int kb = 0; kb < nblk; ++kb)
              for (int g = 0; g < G; ++g)
                for (int lane = 0; lane < 64; ++lane)
                  for (int i = 0; i < 4; ++i) {
                    const int unit = wave * 16 * UTW + 16 * u + (lane & 15);
                    const int k = 16 * kb + 4 * (lane >> 4) + i;
                    stream_mid.push_back(k < K ? W[((size_t)g * H + unit) * K + k] : 0.f);
                  }
          }
  }
  // ... and for its shared form, one stream per (share of the units, wave): [share][wave][layer]...
  std::vector<float> stream_shared[4];
  for (int si = 1; si < 4; ++si) {
    const int S = 1 << si;
    if (!mid_share_ok(H, S)) continue;
    const int DM_ = mid_ring_depth(kind, H), UTW = H / S / 64;
    for (int sh = 0; sh < S; ++sh)
      for (int wave = 0; wave < 4; ++wave)
        for (int l = 0; l < layers; ++l)
          for (int u = 0; u < UTW; ++u)
            for (int part = 0; part < 2; ++part) {
              const float* W = part == 0 ? (l == 0 ? w_fold.data() : lw[l].w_ih) : lw[l].w_hh;
              const int K = part == 0 ? (l == 0 ? F : H) : H;
              const int nblk = part == 0 ? (l == 0 ? DM_ : H / 16) : H / 16;
              for (int kb = 0; kb < nblk; ++kb)
                for (int g = 0; g < G; ++g)
                  for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i) {
                      const int unit = sh * (H / S) + wave * 16 * UTW + 16 * u + (lane & 15);
                      const int k = 16 * kb + 4 * (lane >> 4) + i;
                      stream_shared[si].push_back(k < K ? W[((size_t)g * H + unit) * K + k] : 0.f);
                    }
            }
  }
  const float* dec_w = p; p += (size_t)F * H;
  const float* dec_b = p; p += F;
  if ((size_t)(p - w) != n) return CET_ERR_BLOB_SIZE;

  std::vector<float> dw((size_t)FP * H, 0.f), db(FP, 0.f);
  for (int r = 0; r < F; ++r) {
    for (int c = 0; c < H; ++c) dw[(size_t)r * H + c] = dec_w[(size_t)r * H + c];
    db[r] = dec_b[r];
  }

  RnnModel* m = new (std::nothrow) RnnModel;
  if (!m) return CET_ERR_ALLOC;
  m->kind = kind; m->F = F; m->I = I; m->H = H; m->layers = layers; m->core = core;
  std::vector<float> all;
  auto place = [&](const std::vector<float>& v) {
    const size_t off = all.size();
    all.insert(all.end(), v.begin(), v.end());
    while (all.size() % 64) all.push_back(0.f);  // keep every block 256-byte aligned
    return off;
  };
  const size_t o_stream = place(stream), o_stream_mid = place(stream_mid), o_bias = place(bias),
               o_bias_t = place(bias_tiled),
               o_dw = place(dw), o_db = place(db);
  size_t o_shared[4] = {0, 0, 0, 0};
  for (int si = 1; si < 4; ++si)
    if (!stream_shared[si].empty()) o_shared[si] = place(stream_shared[si]);
  size_t o_ih[3] = {0, 0, 0}, o_hh[3] = {0, 0, 0};
  for (int l = 0; core && l < layers; ++l) {
    o_ih[l] = place(l == 0 ? w_fold : std::vector<float>(lw[l].w_ih, lw[l].w_ih + (size_t)G * H * lw[l].in));
    o_hh[l] = place(std::vector<float>(lw[l].w_hh, lw[l].w_hh + (size_t)G * H * H));
  }
  hipError_t e = hipMalloc(&m->dev, all.size() * sizeof(float));
  if (e == hipSuccess)
    e = hipMemcpy(m->dev, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (m->dev) (void)hipFree(m->dev);
    delete m;
    return hip_fail(e);
  }
  m->wstream = m->dev + o_stream; m->wstream_mid = m->dev + o_stream_mid; m->bias = m->dev + o_bias;
  m->bias_tiled = m->dev + o_bias_t;
  for (int si = 1; si < 4; ++si)
    m->wstream_shared[si] = stream_shared[si].empty() ? nullptr : m->dev + o_shared[si];
  m->dec_w = m->dev + o_dw; m->dec_b = m->dev + o_db;
  for (int l = 0; l < 3; ++l) {
    m->w_ih[l] = core && l < layers ? m->dev + o_ih[l] : nullptr;
    m->w_hh[l] = l < layers ? m->dev + o_hh[l] : nullptr;
  }
  e = hipGetDevice(&m->device);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&m->cus, hipDeviceAttributeMultiprocessorCount, m->device);
  if (e == hipSuccess) e = hipGetSymbolAddress(reinterpret_cast<void**>(&m->static_scratch), HIP_SYMBOL(g_split_scratch));
  if (e != hipSuccess) {
    rnn_destroy(m);
    return hip_fail(e);
  }
  int rc = rnn_prepare(m);
  if (rc == CET_OK && core) rc = rnn_split_occupancy(m);
  if (rc == CET_OK) rnn_split_limits(m);
  if (rc != CET_OK) {
    rnn_destroy(m);
    return rc;
  }
  *out = m;
  return CET_OK;
}


template <int KIND, int UT, int NL>
static size_t lds_bytes() {
  constexpr int H = 128 * UT;
  return sizeof(float) * (size_t)(BT * PSTR * 4 + NL * BT * (H + 4) + NL * 2 * Cell<KIND>::G * H);
}

// Once per handle, on the handle's device: the kernel's dynamic LDS exceeds the default limit.
template <int KIND, int UT, int NL>
static int prepare() {
  CET_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_rollout_kernel<KIND, UT, NL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes<KIND, UT, NL>()));
  return CET_OK;
}

template <int KIND, int UT, int NL>
static int launch(const RArgs& a, hipStream_t stream) {
  const unsigned grid = (unsigned)((a.B + BT - 1) / BT);
  const size_t lds = lds_bytes<KIND, UT, NL>();
  hipLaunchKernelGGL((rnn_rollout_kernel<KIND, UT, NL>), dim3(grid), dim3(256), lds, stream, a);
  CET_HIP(hipGetLastError());
  return CET_OK;
}

// The instantiation that serves a model: (cell kind, H / 128, layers).
typedef int (*PrepareFn)();
typedef int (*LaunchFn)(const RArgs&, hipStream_t);
template <int KIND>
static bool variant_of_kind(int ut, int layers, PrepareFn* pf, LaunchFn* lf) {
#define CET_RNN_CASE(UT_, NL_)                 \
  if (ut == UT_ && layers == NL_) {            \
    *pf = &prepare<KIND, UT_, NL_>;            \
    *lf = &launch<KIND, UT_, NL_>;             \
    return true;                               \
  }
  CET_RNN_CASE(1, 1) CET_RNN_CASE(1, 2) CET_RNN_CASE(1, 3)
  CET_RNN_CASE(2, 1) CET_RNN_CASE(2, 2) CET_RNN_CASE(2, 3)
#undef CET_RNN_CASE
  return false;
}
static bool variant(int kind, int ut, int layers, PrepareFn* pf, LaunchFn* lf) {
  switch (kind) {
    case CET_RNN_TANH: return variant_of_kind<CET_RNN_TANH>(ut, layers, pf, lf);
    case CET_RNN_GRU: return variant_of_kind<CET_RNN_GRU>(ut, layers, pf, lf);
    case CET_RNN_LSTM: return variant_of_kind<CET_RNN_LSTM>(ut, layers, pf, lf);
    default: return false;
  }
}

// The half-tile kernel's instantiations (S = 1: a tile per workgroup; 2, 4: a tile's hidden units
// shared by S workgroups): prepare (dynamic LDS limit) or launch.
template <int KIND, int H, int NL, int S>
static int mid_do(const RArgs* a, hipStream_t stream) {
  const size_t lds = sizeof(float) * (size_t)(4 * MT * PSTR + NL * MT * (H + 4) + NL * 2 * Cell<KIND>::G * H);
  const void* fn;
  if constexpr (S == 1) fn = reinterpret_cast<const void*>(&rnn_mid_kernel<KIND, H, NL>);
  else fn = reinterpret_cast<const void*>(&rnn_mid_shared_kernel<KIND, H, NL, S>);
  if (!a) {
    if (lds > 160 * 1024 - 64) return CET_ERR_UNSUPPORTED;
    CET_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return CET_OK;
  }
  const unsigned tiles = (unsigned)((a->B + MT - 1) / MT);
  if constexpr (S == 1)
    hipLaunchKernelGGL((rnn_mid_kernel<KIND, H, NL>), dim3(tiles), dim3(256), lds, stream, *a);
  else
    hipLaunchKernelGGL((rnn_mid_shared_kernel<KIND, H, NL, S>), dim3(tiles * S), dim3(256), lds, stream, *a);
  CET_HIP(hipGetLastError());
  return CET_OK;
}
// Workgroups a tile's hidden units may be shared by: a wave takes at least one tile of 16 units
// (S <= H / 64), at most 4 at 256 hidden units and 8 at 512.
template <int KIND>
static int mid_kind(const RnnModel* m, const RArgs* a, int S, hipStream_t s) {
#define CET_MID_S(H_, NL_, S_) \
  if (S == S_) return mid_do<KIND, H_, NL_, (H_ / S_ >= 64 ? S_ : 1)>(a, s);
#define CET_MID_CASE(H_, NL_)                         \
  if (m->H == H_ && m->layers == NL_) {               \
    if (!mid_share_ok(H_, S)) return CET_ERR_UNSUPPORTED; \
    CET_MID_S(H_, NL_, 1) CET_MID_S(H_, NL_, 2) CET_MID_S(H_, NL_, 4) CET_MID_S(H_, NL_, 8) \
    return CET_ERR_UNSUPPORTED;                       \
  }
  CET_MID_CASE(64, 1) CET_MID_CASE(64, 2) CET_MID_CASE(64, 3) CET_MID_CASE(64, 4)
  CET_MID_CASE(128, 1) CET_MID_CASE(128, 2) CET_MID_CASE(128, 3) CET_MID_CASE(128, 4)
  CET_MID_CASE(256, 1) CET_MID_CASE(256, 2) CET_MID_CASE(256, 3) CET_MID_CASE(256, 4)
  CET_MID_CASE(512, 1) CET_MID_CASE(512, 2) CET_MID_CASE(512, 3)
#undef CET_MID_CASE
#undef CET_MID_S
  return CET_ERR_UNSUPPORTED;
}
// a == nullptr: prepare
static int mid_dispatch(const RnnModel* m, const RArgs* a, int S, hipStream_t s) {
  switch (m->kind) {
    case CET_RNN_TANH: return mid_kind<CET_RNN_TANH>(m, a, S, s);
    case CET_RNN_GRU: return mid_kind<CET_RNN_GRU>(m, a, S, s);
    default: return mid_kind<CET_RNN_LSTM>(m, a, S, s);
  }
}

static int rnn_prepare(const RnnModel* m) {
  for (int S = 1; S <= 8; S *= 2)
    if (mid_share_ok(m->H, S))
      if (const int rc = mid_dispatch(m, nullptr, S, nullptr)) return rc;
  if (!m->core) return CET_OK;
  PrepareFn pf;
  LaunchFn lf;
  if (!variant(m->kind, m->H / 128, m->layers, &pf, &lf)) return CET_ERR_UNSUPPORTED;
  return pf();
}

template <int KIND, int H, int NL, int UW, int TPW>
static int launch_split(const SArgs& a, int tiles, hipStream_t stream) {
  const unsigned groups = (unsigned)((tiles + TPW - 1) / TPW);
  hipLaunchKernelGGL((rnn_split_kernel<KIND, H, NL, UW, TPW>), dim3(groups * (H / UW)), dim3(256), 0, stream, a);
  CET_HIP(hipGetLastError());
  return CET_OK;
}
// occ[0]: workgroups of the instantiation a CU holds at once; occ[1]: its scratch bytes per lane
// (registers the compiler had to spill)
template <int KIND, int H, int NL, int UW, int TPW>
static int occupancy_split(int* occ) {
  const void* fn = reinterpret_cast<const void*>(&rnn_split_kernel<KIND, H, NL, UW, TPW>);
  CET_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[0], fn, 256, 0));
  hipFuncAttributes at{};
  CET_HIP(hipFuncGetAttributes(&at, fn));
  occ[1] = (int)at.localSizeBytes;
  return CET_OK;
}
// How a launch of the hidden-split kernel is laid out: hidden units per workgroup (4, 8 or 16; 0: the
// batch does not fit one launch) and sequence tiles per workgroup (2 only with 16 units).
struct SplitPlan {
  int uw, tpw;
};
// occ != nullptr: no launch, the residency and spill figures of that instantiation instead.
template <int KIND>
static int launch_split_kind(const RnnModel* m, const SArgs& a, int tiles, SplitPlan pl, hipStream_t s,
                             int* occ = nullptr) {
#define CET_SPLIT_DO(H_, NL_, UW_, TPW_) \
  return occ ? occupancy_split<KIND, H_, NL_, UW_, TPW_>(occ) : launch_split<KIND, H_, NL_, UW_, TPW_>(a, tiles, s)
#define CET_SPLIT_CASE(H_, NL_)                                                          \
  if (m->H == H_ && m->layers == NL_) {                                                  \
    if (KIND != CET_RNN_TANH && pl.uw == 4) CET_SPLIT_DO(H_, NL_, (KIND == CET_RNN_TANH ? 16 : 4), 1); \
    if (KIND != CET_RNN_TANH && pl.uw == 8) CET_SPLIT_DO(H_, NL_, (KIND == CET_RNN_TANH ? 16 : 8), 1); \
    if (pl.tpw == 2) CET_SPLIT_DO(H_, NL_, 16, 2);                                       \
    CET_SPLIT_DO(H_, NL_, 16, 1);                                                        \
  }
  CET_SPLIT_CASE(128, 1) CET_SPLIT_CASE(128, 2) CET_SPLIT_CASE(128, 3)
  CET_SPLIT_CASE(256, 1) CET_SPLIT_CASE(256, 2) CET_SPLIT_CASE(256, 3)
#undef CET_SPLIT_CASE
#undef CET_SPLIT_DO
  return CET_ERR_UNSUPPORTED;
}
static int split_kind_dispatch(const RnnModel* m, const SArgs& a, int tiles, SplitPlan pl, hipStream_t s, int* occ) {
  switch (m->kind) {
    case CET_RNN_TANH: return launch_split_kind<CET_RNN_TANH>(m, a, tiles, pl, s, occ);
    case CET_RNN_GRU: return launch_split_kind<CET_RNN_GRU>(m, a, tiles, pl, s, occ);
    default: return launch_split_kind<CET_RNN_LSTM>(m, a, tiles, pl, s, occ);
  }
}

// split_occ: [0..2] one tile per workgroup with 4, 8, 16 units, [3] two tiles with 16 units
static int rnn_split_occupancy(RnnModel* m) {
  const SArgs none{};
  const SplitPlan plans[4] = {{4, 1}, {8, 1}, {16, 1}, {16, 2}};
  for (int k = 0; k < 4; ++k) {
    int occ[2] = {1, 0};
    if (const int rc = split_kind_dispatch(m, none, 0, plans[k], nullptr, occ)) return rc;
    m->split_occ[k] = occ[0] < 1 ? 1 : occ[0];
    m->split_spill[k] = occ[1];
  }
  return CET_OK;
}

// Workgroups of a launch laid out as `pl` that are resident at once, with one workgroup per CU or
// with as many as fit (at most SPLIT_MAX_OCC).
static long split_capacity(const RnnModel* m, SplitPlan pl, bool shared_cu) {
  const int occ = m->split_occ[pl.tpw == 2 ? 3 : (pl.uw == 4 ? 0 : (pl.uw == 8 ? 1 : 2))];
  return (long)m->cus * (shared_cu ? (occ < SPLIT_MAX_OCC ? occ : SPLIT_MAX_OCC) : 1);
}
static long split_workgroups(const RnnModel* m, int tiles, SplitPlan pl) {
  return (long)((tiles + pl.tpw - 1) / pl.tpw) * (m->H / pl.uw);
}

// The layout for a launch of `tiles` sequence tiles (uw 0: they do not fit one launch). With a CU to
// itself a workgroup is fastest, so first the finest split that fits one workgroup per CU (a small
// batch spreads over the whole device; the tanh cell has one row group and gains nothing below
// 16). Beyond that (more than 256 sequences at 256 hidden units): for the tanh cell two workgroups
// of 16 units per CU, where they fit, else two tiles per workgroup, which take turns so that one's
// matrix work covers the other's exchange. Two workgroups to a CU of one launch run in step and
// share the MFMA pipes, so they only pay where the matrix work is short (tanh cell: 512 sequences in
// 0.18 ms) or the alternative is a second launch (256 sequences, GRU / LSTM: 16 units and one per
// CU 0.22 / 0.24 ms, 8 units and two per CU 0.25 / 0.25 ms).
static SplitPlan split_plan(const RnnModel* m, int tiles) {
  const bool tanh_cell = m->kind == CET_RNN_TANH;
  for (int uw = tanh_cell ? 16 : 4; uw <= 16; uw *= 2)
    if (split_workgroups(m, tiles, {uw, 1}) <= split_capacity(m, {uw, 1}, false)) {
      // An instantiation the compiler had to spill for runs at about half its speed (LSTM, 3 layers
      // of 256, 8 units: 0.43 ms for 96 sequences): rather the finer split with two workgroups per
      // CU (0.33 ms), or the next coarser one.
      if (uw == 8 && m->split_spill[1] > 256) {
        if (split_workgroups(m, tiles, {4, 1}) <= split_capacity(m, {4, 1}, true)) return {4, 1};
        continue;
      }
      return {uw, 1};
    }
  if (tanh_cell && split_workgroups(m, tiles, {16, 1}) <= split_capacity(m, {16, 1}, true)) return {16, 1};
  // (not where two tiles' state costs the compiler registers it has to spill: LSTM, 3 layers of 256)
  if (m->split_spill[3] <= 256 && split_workgroups(m, tiles, {16, 2}) <= split_capacity(m, {16, 2}, false))
    return {16, 2};
  for (int uw = 16; uw >= (tanh_cell ? 16 : 4); uw /= 2)
    if (split_workgroups(m, tiles, {uw, 1}) <= split_capacity(m, {uw, 1}, true)) return {uw, 1};
  return {0, 1};
}

// The largest launch of the hidden-split kernel and the time estimates behind the automatic choice
// of a kernel (rnn_choose()). Times per time step from a model fitted to measurements (MI355X;
// hidden 128 / 256, 1 to 3 layers, all three cells; tools/rnnshapes.py, tools/midtest.py, within 15 %):
//   batch-tiled: 0.115 us per (k block of 8, gate, unit tile of a wave), whatever the batch up to 32
//   sequences per CU; half-tile kernel: 0.56 of that up to 16 per CU, divided by the workgroups S
//   that share a tile, plus 1.95 us of exchange per layer and 0.6 us per step when S > 1;
//   hidden-split, per launch: 13.3 ns per MFMA a wave issues (32 cycles) + per layer 2.4 us of
//   exchange and fixed work (2.1 us with two workgroups to a CU, 3.6 us for two tiles taking turns)
//   + 0.6 (1.0) us per step.
static void rnn_split_limits(RnnModel* m) {
  int tiles = 0;
  while (m->core && tiles < 4096 && split_plan(m, tiles + 1).uw) ++tiles;
  m->split_chunk = tiles * ST;
  const int G = gates_of(m->kind), H = m->H;
  double tiled = 0.0;
  for (int l = 0; l < m->layers; ++l) tiled += 0.115 * (H / 128) * G * (l == 0 ? 4 + H / 8 : 2 * H / 8);
  m->est_tiled_step = tiled;
}

// Estimated microseconds per time step of one hidden-split launch of `tiles` sequence tiles.
static double split_step_estimate(const RnnModel* m, int tiles) {
  const SplitPlan pl = split_plan(m, tiles);
  if (!pl.uw) return 1e30;
  const int G = gates_of(m->kind), H = m->H;
  const bool shared = split_workgroups(m, tiles, pl) > m->cus;  // two workgroups to a CU share its MFMA pipes
  const int sets = pl.uw == 16 ? G : (pl.uw == 8 ? 2 : 1);      // weight register sets per k block
  double t = pl.tpw == 2 ? 1.0 : 0.6;
  for (int l = 0; l < m->layers; ++l) {
    const int blocks = l == 0 ? 2 + H / 16 : 2 * H / 16;  // k blocks of 16, a quarter per wave
    const int mfma = (blocks + 3) / 4 * sets * 4;
    t += (pl.tpw == 2 ? 3.6 : (shared ? 2.1 : 2.4)) + 0.0133 * mfma * (pl.tpw == 2 || shared ? 2 : 1);
  }
  return t;
}

// The kernel for a batch under the automatic choice: 1 batch-tiled, 2 hidden-split (in chunks), 4 / 5
// / 6 / 8 half-tile with 1 / 2 / 4 / 8 workgroups per tile; the least estimated time. pooled: kernels that
// exchange through a buffer of the pool are allowed (not under stream capture).
static int rnn_choose(const RnnModel* m, int B, bool pooled, double* t_out = nullptr) {
  const long tiles = (B + MT - 1) / MT;
  // the batch-tiled kernel, or for the shapes it does not serve the half-tile kernel in rounds
  int best = m->core ? 1 : 4;
  double t_best = m->core ? m->est_tiled_step * ((B + 32L * m->cus - 1) / (32L * m->cus))
                          : 0.56 * m->est_tiled_step * ((tiles + m->cus - 1) / m->cus);
  for (int S = 1, si = 0; S <= 8; S *= 2, ++si) {
    if (tiles * S > m->cus || (S > 1 && (!pooled || !m->wstream_shared[si]))) continue;
    if (S > 1 && (size_t)tiles * m->layers > SPLIT_CNT_SLOTS) continue;
    const double t = 0.56 * m->est_tiled_step / S + (S > 1 ? 0.6 + 1.95 * m->layers : 0.0);
    if (t < t_best) {
      t_best = t;
      best = S == 1 ? 4 : (S == 2 ? 5 : (S == 4 ? 6 : 8));
    }
  }
  if (pooled && m->core && m->split_chunk > 0) {
    const int full = B / m->split_chunk, rest = B % m->split_chunk;
    double t = full ? full * split_step_estimate(m, m->split_chunk / ST) : 0.0;
    if (rest) t += split_step_estimate(m, (rest + ST - 1) / ST);
    if (t < t_best) {
      t_best = t;
      best = 2;
    }
  }
  if (t_out) *t_out = t_best;
  return best;
}

// Whether the hidden-split kernel takes a batch in one launch: all its workgroups resident at once
// (under some split of the hidden units), counters for its tiles, and an exchange area of at most
// max_wg16 * layers * 2 * 256 floats (a tile's area is that of H / 16 workgroups of 16 units).
static int split_tiles(int B) { return (B + ST - 1) / ST; }
static bool split_fits(const RnnModel* m, int B, long max_wg16) {
  const int tiles = split_tiles(B);
  if ((long)tiles * (m->H / 16) > max_wg16) return false;
  const int uw = split_plan(m, tiles).uw;
  if (!uw) return false;
  const int ns = m->H / uw > 16 ? m->H / uw / 16 : 1;
  return (size_t)tiles * m->layers * ns <= SPLIT_CNT_SLOTS;  // the last slot is the failure flag
}

static RArgs tiled_args(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len) {
  RArgs a{};
  a.wstream = m->wstream; a.bias = m->bias_tiled;
  a.dec_w = m->dec_w; a.dec_b = m->dec_b; a.x = x; a.out = out;
  a.B = B; a.seq_len = seq_len; a.pred_len = pred_len; a.F = m->F;
  return a;
}

// The hidden-split rollout on `scratch` (device, SPLIT_CNT_FLOATS + tiles * layers * 2 * H / 16 *
// 256 floats): zeroes the counters, launches, and launches the batch-tiled kernel behind it on
// condition of the failure flag. Allocates nothing.
static int rollout_split(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len,
                         float* scratch, hipStream_t stream) {
  SArgs a{};
  a.spin_limit = m->variant == 3 ? 0 : SPIN_LIMIT;
  for (int l = 0; l < 3; ++l) {
    a.w_ih[l] = m->w_ih[l];
    a.w_hh[l] = m->w_hh[l];
  }
  a.bias = m->bias; a.dec_w = m->dec_w; a.dec_b = m->dec_b;
  a.x = x; a.out = out; a.B = B; a.seq_len = seq_len; a.pred_len = pred_len; a.F = m->F;
  a.cnt = reinterpret_cast<unsigned*>(scratch);
  a.hx = scratch + SPLIT_CNT_FLOATS;
  CET_HIP(hipMemsetAsync(scratch, 0, SPLIT_CNT_FLOATS * sizeof(float), stream));
  const int tiles = split_tiles(B);
  const SplitPlan plan = split_plan(m, tiles);
  if (!plan.uw) return CET_ERR_TOO_LARGE;
#ifdef CET_PROF
  static unsigned long long* sp_dev = nullptr;
  const int sp_wg = (int)split_workgroups(m, tiles, plan);
  if (!sp_dev) CET_HIP(hipMalloc(&sp_dev, 1024 * SP_WORDS * sizeof(unsigned long long)));
  CET_HIP(hipMemsetAsync(sp_dev, 0, 1024 * SP_WORDS * sizeof(unsigned long long), stream));
  a.prof = sp_wg <= 1024 ? sp_dev : nullptr;
#endif
  const int rc = split_kind_dispatch(m, a, tiles, plan, stream, nullptr);
  if (rc != CET_OK) return rc;
#ifdef CET_PROF
  if (a.prof && getenv("CET_SPLIT_PROF")) {
    static unsigned long long h[1024 * SP_WORDS];
    CET_HIP(hipStreamSynchronize(stream));
    CET_HIP(hipMemcpy(h, sp_dev, sizeof h, hipMemcpyDeviceToHost));
    static const char* names[6] = {"matrix work + cell update", "publish store acknowledged", "publish -> fetch entry",
                                   "counter wait + barrier", "loads + LDS copy + barrier", "rest (decode, input)"};
    const int ex = (seq_len + pred_len - 1) * m->layers;
    for (int i = 0; i < 6; ++i) {
      double sum = 0, mx = 0;
      for (int w = 0; w < sp_wg; ++w) {
        sum += (double)h[w * SP_WORDS + i];
        if ((double)h[w * SP_WORDS + i] > mx) mx = (double)h[w * SP_WORDS + i];
      }
      fprintf(stderr, "[cet split prof] %-28s mean %7.3f us  max %7.3f us per layer step (B %d, %d workgroups)\n",
              names[i], sum / sp_wg / ex * 0.01, mx / ex * 0.01, B, sp_wg);
    }
  }
#endif
  RArgs t = tiled_args(m, x, out, B, seq_len, pred_len);
  t.only_if = a.cnt + (SPLIT_CNT_FLOATS - 1);
  PrepareFn pf;
  LaunchFn lf;
  if (!variant(m->kind, m->H / 128, m->layers, &pf, &lf)) return CET_ERR_UNSUPPORTED;
  return lf(t, stream);
}

// The shared half-tile rollout (S workgroups per 16-sequence tile) on `scratch`: zeroes the counters,
// launches, and launches the unshared half-tile kernel behind it on condition of the failure flag.
static int rollout_shared(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len,
                          int S, float* scratch, hipStream_t stream) {
  const int si = S == 2 ? 1 : (S == 4 ? 2 : 3);
  const long tiles = (B + MT - 1) / MT;
  if (!m->wstream_shared[si] || tiles * S > m->cus || (size_t)tiles * m->layers > SPLIT_CNT_SLOTS)
    return CET_ERR_TOO_LARGE;
  RArgs a = tiled_args(m, x, out, B, seq_len, pred_len);
  a.wstream = m->wstream_shared[si];
  a.cnt = reinterpret_cast<unsigned*>(scratch);
  a.hx = scratch + SPLIT_CNT_FLOATS;
  a.spin_limit = m->variant == 7 ? 0 : SPIN_LIMIT;  // 7: the exchange gives up at its first unanswered poll (tests)
  CET_HIP(hipMemsetAsync(scratch, 0, SPLIT_CNT_FLOATS * sizeof(float), stream));
  if (const int rc = mid_dispatch(m, &a, S, stream)) return rc;
  // on stand-by: the half-tile kernel with a tile per workgroup (every shape has it)
  RArgs t = tiled_args(m, x, out, B, seq_len, pred_len);
  t.wstream = m->wstream_mid;
  t.only_if = a.cnt + (SPLIT_CNT_FLOATS - 1);
  return mid_dispatch(m, &t, 1, stream);
}

// Claims an exchange buffer of this device's pool for launches on `stream` (see SplitPool). The
// caller holds pool.mu and keeps it until it has recorded the slot's event behind its launches.
static int split_claim(const RnnModel* m, SplitPool& pool, hipStream_t stream, int need_cus, int* slot_out) {
  // hidden-split launches: up to SPLIT_MAX_OCC * CUs workgroup-slices of 256 floats per layer and parity;
  // shared half-tile launches: up to CUs / 2 tiles of 16 x 256 floats
  // (up to 512 hidden units and 4 layers: sized for the largest shape, so that a slot never grows)
  const size_t per_lp = (size_t)(m->cus / 2) * MT * 512;
  const size_t need = (SPLIT_CNT_FLOATS + per_lp * 4 * 2) * sizeof(float);
  if (pool.slot_bytes < need) {  // first use on this device (the CU count of a device is fixed)
    for (int i = 0; i < SPLIT_SLOTS; ++i) {
      SplitSlot& sl = pool.slot[i];
      if (!sl.buf) CET_HIP(hipMalloc(&sl.buf, need));
      if (!sl.done) CET_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    pool.slot_bytes = need;
  }
  int pick = -1;
  for (int i = 0; i < SPLIT_SLOTS && pick < 0; ++i)  // the slot this stream used last
    if (pool.slot[i].used && pool.slot[i].last == stream) pick = i;
  for (int i = 0; i < SPLIT_SLOTS && pick < 0; ++i)  // one never used
    if (!pool.slot[i].used) pick = i;
  for (int i = 0; i < SPLIT_SLOTS && pick < 0; ++i) {  // one whose launches have finished
    const hipError_t q = hipEventQuery(pool.slot[i].done);
    if (q == hipSuccess) pick = i;
    else if (q != hipErrorNotReady) return hip_fail(q);
  }
  // all in flight on other streams: queue behind the one whose turn it is
  if (pick < 0) pick = (int)(pool.next++ % SPLIT_SLOTS);
  // Behind the slot's previous launches in every case: free when they ran on this stream or have
  // finished, and it covers a new stream that was given the handle value of a destroyed one whose
  // work is still running.
  if (pool.slot[pick].used) CET_HIP(hipStreamWaitEvent(stream, pool.slot[pick].done, 0));
  // Exchanging grids that are in flight together never ask for more CUs than the device has: a
  // workgroup that waits for the others of its tile holds its CU, and with the XCDs placing their
  // shares of a grid independently several grids could otherwise each hold part of the CUs the
  // others' missing workgroups need (seen as rollouts that ended in the bounded wait and the
  // stand-by launch when three streams launched 256-workgroup grids at once). Beyond that the
  // stream waits, on the device, for earlier launches of other streams.
  int busy = 0;
  bool counts[SPLIT_SLOTS] = {};
  for (int i = 0; i < SPLIT_SLOTS; ++i) {
    const SplitSlot& sl = pool.slot[i];
    if (i == pick || !sl.used || sl.last == stream) continue;  // the stream's own launches run before this one
    const hipError_t q = hipEventQuery(sl.done);
    if (q == hipSuccess) continue;
    if (q != hipErrorNotReady) return hip_fail(q);
    counts[i] = true;
    busy += sl.cus;
  }
  for (int i = 0; i < SPLIT_SLOTS && busy + need_cus > m->cus; ++i)
    if (counts[i]) {
      CET_HIP(hipStreamWaitEvent(stream, pool.slot[i].done, 0));
      busy -= pool.slot[i].cus;
    }
  pool.slot[pick].cus = need_cus;
  pool.slot[pick].used = true;
  pool.slot[pick].last = stream;
  *slot_out = pick;
  return CET_OK;
}

int rnn_device(const RnnModel* m) { return m->device; }
void rnn_select(RnnModel* m, int variant) { m->variant = variant; }

// The rollout on the library's static scratch buffer: no allocation on any path.
int rnn_rollout_static(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len,
                       hipStream_t stream) {
  if (!m->core) return CET_ERR_UNSUPPORTED;  // no hidden-split kernel for this shape
  if (!split_fits(m, B, SPLIT_MAX_WG) || m->layers > SPLIT_MAX_LAYERS) return CET_ERR_TOO_LARGE;
  return rollout_split(m, x, out, B, seq_len, pred_len, m->static_scratch, stream);
}

// Sequences one launch of the hidden-split kernel takes on this handle's device (whole tiles).
static int split_chunk(const RnnModel* m) { return m->split_chunk; }

static int rollout_with(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len,
                        int kernel, hipStream_t stream);

int rnn_rollout(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len,
                hipStream_t stream) {
  if (B <= 0) return CET_OK;
  if (m->variant != 0) return rollout_with(m, x, out, B, seq_len, pred_len, m->variant, stream);
  // The automatic choice, part by part. Whole rounds of the batch-tiled kernel (32 sequences per CU)
  // go in one launch; what is left goes to the kernel with the least estimated time, or, between
  // 16 and 32 sequences per CU, to one round of half tiles followed by a rollout of the rest when
  // the two together beat a round of full tiles (5,000 sequences, GRU: 1.10 + 0.37 ms instead of
  // 2.0). Shapes without the batch-tiled kernel: whole rounds of half tiles, then the rest.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  CET_HIP(hipStreamIsCapturing(stream, &cap));
  const bool pooled = cap == hipStreamCaptureStatusNone;  // see rollout_with()
  const long round_full = 32L * m->cus, round_half = (long)MT * m->cus;
  const size_t xs = (size_t)seq_len * m->F, os = (size_t)pred_len * m->F;
  long done = 0;
  while (done < B) {
    const long rem = B - done;
    long nb = rem;
    int kernel;
    if (m->core && rem >= round_full) {
      nb = rem / round_full * round_full;
      kernel = 1;
    } else if (!m->core && rem > round_half) {
      nb = rem / round_half * round_half;
      kernel = 4;
    } else {
      double t_one = 0.0, t_rest = 0.0;
      kernel = rnn_choose(m, (int)rem, pooled, &t_one);
      if (rem > round_half) {
        rnn_choose(m, (int)(rem - round_half), pooled, &t_rest);
        if (0.56 * m->est_tiled_step + t_rest + 0.35 < t_one) {  // 0.35 us per step: a second launch (10 us per rollout of 29)
          nb = round_half;
          kernel = 4;
        }
      }
    }
    if (const int rc = rollout_with(m, x + (size_t)done * xs, out + (size_t)done * os, (int)nb, seq_len, pred_len,
                                    kernel, stream))
      return rc;
    done += nb;
  }
  return CET_OK;
}

// One rollout on one kernel (1 batch-tiled, 2 / 3 hidden-split, 4 / 5 / 6 / 8 half-tile on 1 / 2 / 4 / 8
// workgroups per tile, 7 as 5 with an exchange that gives up).
static int rollout_with(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len,
                        int kernel, hipStream_t stream) {
  // Small and medium batches: hidden units split over workgroups (a batch-tiled rollout of 64
  // sequences keeps 2 of the CUs busy for as long as one of 8192 keeps all of them busy). One
  // launch takes as many sequences as its workgroups are resident at once, each with one or two
  // sequence tiles (512 at 256 hidden units); larger batches go through it in chunks, one launch
  // after the other on the stream, for as long as that beats the batch-tiled kernel, whose time is
  // the same from 32 to 8192 sequences (rnn_split_limits(); measured per full launch / per
  // batch-tiled rollout, 2 layers of 256: GRU 0.35 / 2.0 ms, LSTM 0.40 / 2.6 ms, tanh 0.18 /
  // 0.70 ms).
  const int chunk = split_chunk(m);
  if (B <= 0) return CET_OK;
  // hidden 64 / 512 or 4 layers: the half-tile kernels only
  if (!m->core && (kernel == 1 || kernel == 2 || kernel == 3)) return CET_ERR_UNSUPPORTED;
  if (kernel == 2 || kernel == 3 || kernel >= 5) {
    // These exchange through a buffer of the pool. Under stream capture the launches would be
    // replayed later, outside the pool's bookkeeping: the kernels that need no buffer serve then
    // (cet_rnn_rollout_static serves captured small batches).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    CET_HIP(hipStreamIsCapturing(stream, &cap));
    if (cap != hipStreamCaptureStatusNone) {
      if (m->variant != 0) return CET_ERR_UNSUPPORTED;
      kernel = rnn_choose(m, B, false);
    }
  }
  const bool split = (kernel == 2 || kernel == 3) && chunk > 0;
  const int shared = kernel == 5 || kernel == 7 ? 2 : (kernel == 6 ? 4 : (kernel == 8 ? 8 : 0));  // workgroups per half tile
  if ((kernel == 2 || kernel == 3) && !split) return CET_ERR_TOO_LARGE;
  if (split || shared) {
    if (m->device < 0 || m->device >= SPLIT_MAX_DEVICES) return CET_ERR_UNSUPPORTED;
    SplitPool& pool = g_split_pool[m->device];
    std::lock_guard<std::mutex> lock(pool.mu);  // claim, launches and event as one step (see SplitPool)
    // CUs for every workgroup of the (largest) launch to be resident at once
    long need = m->cus;
    if (shared) {
      need = (long)((B + MT - 1) / MT) * shared;
    } else {
      const int tiles = split_tiles(B < chunk ? B : chunk);
      const SplitPlan plan = split_plan(m, tiles);
      if (plan.uw) need = split_workgroups(m, tiles, plan);
    }
    if (need > m->cus) need = m->cus;
    int slot = -1;
    if (const int rc = split_claim(m, pool, stream, (int)need, &slot)) return rc;
    int rc = CET_OK;
    if (shared) rc = rollout_shared(m, x, out, B, seq_len, pred_len, shared, pool.slot[slot].buf, stream);
    for (int b0 = 0; !shared && b0 < B && rc == CET_OK; b0 += chunk) {
      const int nb = B - b0 < chunk ? B - b0 : chunk;
      rc = !split_fits(m, nb, (long)SPLIT_MAX_OCC * m->cus)
               ? CET_ERR_TOO_LARGE
               : rollout_split(m, x + (size_t)b0 * seq_len * m->F, out + (size_t)b0 * pred_len * m->F, nb,
                               seq_len, pred_len, pool.slot[slot].buf, stream);
    }
    // the slot is free again behind whatever was launched
    const hipError_t e = hipEventRecord(pool.slot[slot].done, stream);
    if (e != hipSuccess && rc == CET_OK) rc = hip_fail(e);
    return rc;
  }
  RArgs a = tiled_args(m, x, out, B, seq_len, pred_len);
  // Up to one half tile per CU the half-tile kernel (16 sequences per workgroup) takes 0.55 of the
  // batch-tiled kernel's time; beyond, its second round of workgroups costs more than full tiles.
  if (kernel == 4) {
    a.wstream = m->wstream_mid;
    return mid_dispatch(m, &a, 1, stream);
  }
#ifdef CET_PROF
  static unsigned long long* prof_dev = nullptr;
  const size_t prof_bytes = (size_t)RP_SLOTS * 4 * RP_WORDS * sizeof(unsigned long long);
  if (!prof_dev) CET_HIP(hipMalloc(&prof_dev, prof_bytes));
  CET_HIP(hipMemsetAsync(prof_dev, 0, prof_bytes, stream));
  a.prof = prof_dev;
#endif
  PrepareFn pf;
  LaunchFn lf;
  if (!variant(m->kind, m->H / 128, m->layers, &pf, &lf)) return CET_ERR_UNSUPPORTED;
  const int rc = lf(a, stream);
#ifdef CET_PROF
  if (rc == CET_OK) {
    static const char* names[RP_CATS] = {"encoder+barrier", "bias init", "x part", "h part",
                                         "cell update", "barrier (h read)", "h store+barrier",
                                         "decode tail"};
    unsigned long long h[RP_SLOTS * 4 * RP_WORDS];
    CET_HIP(hipStreamSynchronize(stream));
    CET_HIP(hipMemcpy(h, prof_dev, sizeof h, hipMemcpyDeviceToHost));
    double cat[RP_CATS + 2] = {};
    int n = 0;
    for (int s = 0; s < RP_SLOTS * 4; ++s) {
      if (!h[s * RP_WORDS + RP_CATS]) continue;
      ++n;
      for (int i = 0; i < RP_CATS + 2; ++i) cat[i] += (double)h[s * RP_WORDS + i];
    }
    if (n) {
      const int steps = seq_len + pred_len - 1;
      fprintf(stderr, "[cet rnn prof] kind %d B %d steps %d: %.0f ticks/step, shader clock %.3f GHz\n",
              m->kind, B, steps, cat[RP_CATS] / n / steps, cat[RP_CATS] / cat[RP_CATS + 1] * 0.1);
      for (int i = 0; i < RP_CATS; ++i)
        fprintf(stderr, "[cet rnn prof]   %-18s %9.0f ticks/step  %5.1f %%\n", names[i],
                cat[i] / n / steps, 100.0 * cat[i] / cat[RP_CATS]);
    }
  }
#endif
  return rc;
}

}  // namespace cet
