// LMMSE MIMO equalizer for gfx950: per resource element (RE)
//     A = H^H S^-1 H + I,  z = A^-1 H^H S^-1 y,  d = diag(A^-1 H^H S^-1 H),
//     x_hat = z / d,  no_eff = 1/d - 1
// (the arithmetic behind `lmmse_equ([y, h_hat, err_var, no])`,
// LocalSimulations/EndtoEndtest.py:133,191).
//
// The job is HBM-bound (688 B and ~2.9 kFLOP per RE at 16x4), so the kernel is built around
// the memory stream, not around the arithmetic:
//   * each wave owns a tile of T consecutive REs; H and y of a tile are pulled global->LDS by
//     LDS-DMA (global_load_lds, 16 B/lane, 1 KiB per wave instruction, fully coalesced); the
//     wave copies its operands from LDS to registers, hands the buffer back to the DMA engine
//     for the next tile at once and does the arithmetic from registers meanwhile (LDS is a
//     transposing staging area: one 20 KiB buffer per wave, 8 waves per CU);
//   * the only synchronisation is the wave's own counted s_waitcnt vmcnt(N): tiles are
//     wave-private, there is no workgroup barrier;
//   * in LDS the 16-byte chunks of each RE are XOR-permuted (done on the DMA *source*
//     address, the LDS side stays linear) so that the per-RE strided ds_read_b128 of the
//     compute phase are bank-conflict free;
//   * one RE is handled by LPR lanes that split the receive antennas; the partial Gram
//     matrices are combined with cross-lane adds and every lane then carries out the small
//     Hermitian solve (Cholesky, triangular inverse) in registers.
// Block fading (`hshare` > 1): H holds one matrix per `hshare` consecutive REs (the symbols of
// a slot over which a predicted channel is held). The DMA gathers every RE's H from its block,
// so HBM sees each matrix once (the re-reads of the following REs hit L1/L2) and the
// arithmetic per RE is exactly that of the per-RE layout: results are bit-identical to running
// on the expanded H.
// With S = no*I the kernel uses A' = H^H H + no*I, for which d = 1 - no*(A'^-1)_ss and
// no_eff = t/(1-t), t = no*(A'^-1)_ss: no cancellation in no_eff at high SNR.
#include <cstdint>
#include <cstdlib>

#include <atomic>

#include "cet_internal.h"

namespace cet {
namespace {

constexpr int pair_index(int ns, int s, int s2) { return s * ns - s * (s + 1) / 2 + (s2 - s - 1); }

// Complex values are kept as (re, im) pairs: a complex multiply-add is then two packed fp32
// instructions (v_pk_fma_f32 with half-select / negate modifiers) instead of four scalar ones.
// Per component the chain of fused multiply-adds is the one of the scalar statement
//   re += ar*br; re += ai*bi; im += ar*bi; im -= ai*br        (acc += conj(a) * b)
// so the results do not depend on which form the compiler emits.
typedef float pair2 __attribute__((ext_vector_type(2)));
__device__ inline pair2 fma2(pair2 a, pair2 b, pair2 c) { return __builtin_elementwise_fma(a, b, c); }
// acc += conj(a) * b
__device__ inline pair2 cmac_conj(pair2 acc, pair2 a, pair2 b) {
  acc = fma2(a.xx, b, acc);
  return fma2(a.yy, pair2{b.y, -b.x}, acc);
}

template <int NS>
struct Gram {
  static constexpr int NP = NS * (NS - 1) / 2 > 0 ? NS * (NS - 1) / 2 : 1;
  float diag[NS];  // sum |h_s|^2
  pair2 off[NP];   // s < s2: sum conj(h_s) h_s2
  pair2 b[NS];     // sum conj(h_s) y
  __device__ inline void clear() {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      diag[s] = 0.f;
      b[s] = pair2{0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) off[i] = pair2{0.f, 0.f};
  }
};

// One receive antenna's contribution; h = its row of H. WEIGHTED: row scaled by w = 1/S_r
// (diagonal S). gram_row_h: the Gram matrix part, gram_row_b: the matched-filter part.
template <int NS, bool WEIGHTED>
__device__ inline void gram_row_h(Gram<NS>& g, const pair2 (&h)[NS], float w) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const pair2 a = WEIGHTED ? h[s] * w : h[s];
    g.diag[s] = fmaf(a.x, h[s].x, g.diag[s]);
    g.diag[s] = fmaf(a.y, h[s].y, g.diag[s]);
#pragma unroll
    for (int s2 = s + 1; s2 < NS; ++s2) {
      const int p = pair_index(NS, s, s2);
      g.off[p] = cmac_conj(g.off[p], a, h[s2]);
    }
  }
}
template <int NS, bool WEIGHTED>
__device__ inline void gram_row_b(Gram<NS>& g, const pair2 (&h)[NS], pair2 y, float w) {
#pragma unroll
  for (int s = 0; s < NS; ++s) g.b[s] = cmac_conj(g.b[s], WEIGHTED ? h[s] * w : h[s], y);
}
// The same with the row given as separate real and imaginary parts.
template <int NS, bool WEIGHTED>
__device__ inline void gram_row(Gram<NS>& g, const float (&hr)[NS], const float (&hi)[NS],
                                float yr, float yi, float w) {
  pair2 h[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) h[s] = pair2{hr[s], hi[s]};
  gram_row_h<NS, WEIGHTED>(g, h, w);
  gram_row_b<NS, WEIGHTED>(g, h, pair2{yr, yi}, w);
}

// (G + lam*I)^-1 through its Cholesky factor: W = L^-1 (lower triangular), and per stream
// rd = 1 / d = 1 / (1 - t), ne = no_eff = t / (1 - t) with t = lam * (A^-1)_ss (file header).
template <int NS>
struct Factor {
  pair2 W[NS][NS];
  float rd[NS], ne[NS];
};

template <int NS>
__device__ inline void lmmse_factor(const Gram<NS>& g, float lam, Factor<NS>& f) {
  // Every multiply-add is written as the fma it should be and nothing else may be fused: which
  // operands the optimiser would contract depends on the surrounding code, and the kernels that
  // share this function must agree bit for bit.
#pragma clang fp contract(off)
  float Lr[NS][NS], Li[NS][NS], inv[NS], Wr[NS][NS], Wi[NS][NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    float d = g.diag[j] + lam;
#pragma unroll
    for (int k = 0; k < j; ++k) {
      d = fmaf(-Lr[j][k], Lr[j][k], d);
      d = fmaf(-Li[j][k], Li[j][k], d);
    }
    inv[j] = __builtin_amdgcn_rsqf(d);  // 1 ulp
    Lr[j][j] = d * inv[j];
    Li[j][j] = 0.f;
#pragma unroll
    for (int i = j + 1; i < NS; ++i) {
      const int p = pair_index(NS, j, i);
      float sr = g.off[p].x, si = -g.off[p].y;  // A_ij = conj(A_ji), i > j
#pragma unroll
      for (int k = 0; k < j; ++k) {  // s -= L_ik conj(L_jk)
        sr = fmaf(-Lr[i][k], Lr[j][k], sr);
        sr = fmaf(-Li[i][k], Li[j][k], sr);
        si = fmaf(-Li[i][k], Lr[j][k], si);
        si = fmaf(Lr[i][k], Li[j][k], si);
      }
      Lr[i][j] = sr * inv[j];
      Li[i][j] = si * inv[j];
    }
  }
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    Wr[j][j] = inv[j];
    Wi[j][j] = 0.f;
#pragma unroll
    for (int i = j + 1; i < NS; ++i) {
      float sr = 0.f, si = 0.f;
#pragma unroll
      for (int k = j; k < i; ++k) {  // s += L_ik W_kj
        sr = fmaf(Lr[i][k], Wr[k][j], sr);
        sr = fmaf(-Li[i][k], Wi[k][j], sr);
        si = fmaf(Lr[i][k], Wi[k][j], si);
        si = fmaf(Li[i][k], Wr[k][j], si);
      }
      Wr[i][j] = -sr * inv[i];
      Wi[i][j] = -si * inv[i];
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float a = 0.f;  // (A^-1)_ss = sum_k |W_ks|^2
#pragma unroll
    for (int k = s; k < NS; ++k) {
      a = fmaf(Wr[k][s], Wr[k][s], a);
      a = fmaf(Wi[k][s], Wi[k][s], a);
      f.W[k][s] = pair2{Wr[k][s], Wi[k][s]};
    }
    const float t = lam * a;
    f.rd[s] = __builtin_amdgcn_rcpf(1.0f - t);  // 1 ulp
    f.ne[s] = t * f.rd[s];
  }
}

// x_hat = diag(1/d) W^H W b for the matched-filter output b = H^H S^-1 y. Per component the
// chains are:  u_i.re += Wr*b.re; u_i.re -= Wi*b.im; u_i.im += Wr*b.im; u_i.im += Wi*b.re  (j <= i;
// the diagonal of W is real: its second step is left out), and likewise for z = W^H u.
template <int NS>
__device__ inline void lmmse_apply(const Factor<NS>& f, const pair2 (&b)[NS], pair2 (&x)[NS]) {
#pragma clang fp contract(off)
  pair2 u[NS];  // u = W b:  u_i = sum_j W_ij b_j
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    pair2 acc{0.f, 0.f};
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      acc = fma2(f.W[i][j].xx, b[j], acc);
      if (j < i) acc = fma2(f.W[i][j].yy, pair2{-b[j].y, b[j].x}, acc);
    }
    u[i] = acc;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {  // z = W^H u:  z_s = sum_k conj(W_ks) u_k
    pair2 acc{0.f, 0.f};
#pragma unroll
    for (int k = s; k < NS; ++k) {
      acc = fma2(f.W[k][s].xx, u[k], acc);
      if (k > s) acc = fma2(f.W[k][s].yy, pair2{u[k].y, -u[k].x}, acc);
    }
    x[s] = acc * f.rd[s];
  }
}

// Solves (G + lam*I) z = b by Cholesky and forms x_hat / no_eff (see the file header).
template <int NS>
__device__ inline void lmmse_solve(const Gram<NS>& g, float lam, float (&xr)[NS],
                                   float (&xi)[NS], float (&ne)[NS]) {
  Factor<NS> f;
  lmmse_factor<NS>(g, lam, f);
  pair2 x[NS];
  lmmse_apply<NS>(f, g.b, x);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    xr[s] = x[s].x;
    xi[s] = x[s].y;
    ne[s] = f.ne[s];
  }
}

// Epilogue of the LLR entry point: exact ("app") log-likelihood ratios ln(P(b=1)/P(b=0)) of
// Gray-mapped QPSK with bit 0 on the positive axis (the convention of the reference's mapper /
// demapper pair), from the unbiased estimate and its effective noise:
// llr = -2 sqrt(2) Re/Im(x_hat) / no_eff. Stored in place of x_hat.
template <int NS>
__device__ inline void to_llr(pair2 (&x)[NS], const float (&ne)[NS]) {
#pragma unroll
  for (int s = 0; s < NS; ++s) x[s] = x[s] * (-2.8284271247461903f * __builtin_amdgcn_rcpf(ne[s]));
}
template <int NS>
__device__ inline void to_llr(float (&xr)[NS], float (&xi)[NS], const float (&ne)[NS]) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float k = -2.8284271247461903f * __builtin_amdgcn_rcpf(ne[s]);
    xr[s] *= k;
    xi[s] *= k;
  }
}

// ---------------------------------------------------------------------------------------
// Straightforward kernel: one RE per thread, operands read straight from global memory.
// Serves every noise mode and is what the tiled kernel is cross-checked against.
template <int NR, int NS, int MODE>
__global__ __launch_bounds__(256) void lmmse_direct(const float* __restrict__ y,
                                                    const float* __restrict__ H, float no_scalar,
                                                    const float* __restrict__ no_dev,
                                                    float* __restrict__ x_hat,
                                                    float* __restrict__ no_eff, long N,
                                                    int hshare) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  Gram<NS> g;
  g.clear();
  const float2* h2 = reinterpret_cast<const float2*>(H) + (hshare > 1 ? n / hshare : n) * (NR * NS);
  const float2* y2 = reinterpret_cast<const float2*>(y) + n * NR;
  for (int r = 0; r < NR; ++r) {
    float hr[NS], hi[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float2 v = h2[r * NS + s];
      hr[s] = v.x;
      hi[s] = v.y;
    }
    const float2 yv = y2[r];
    if (MODE == CET_NO_PER_RE_RX) {
      gram_row<NS, true>(g, hr, hi, yv.x, yv.y, 1.0f / no_dev[n * NR + r]);
    } else {
      gram_row<NS, false>(g, hr, hi, yv.x, yv.y, 1.0f);
    }
  }
  const float lam = MODE == CET_NO_SCALAR ? no_scalar : (MODE == CET_NO_PER_RE ? no_dev[n] : 1.0f);
  float xr[NS], xi[NS], ne[NS];
  lmmse_solve<NS>(g, lam, xr, xi, ne);
  if (!no_eff) to_llr<NS>(xr, xi, ne);  // LLR entry point: x_hat receives the LLRs
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    reinterpret_cast<float2*>(x_hat)[n * NS + s] = make_float2(xr[s], xi[s]);
    if (no_eff) no_eff[n * NS + s] = ne[s];
  }
}

// ---------------------------------------------------------------------------------------
// Resource-grid kernel: the operands in the layouts the equalizer OBJECT of the reference
// receives and returns (LocalSimulations/EndtoEndtest.py:191), so that no gather / permute pass
// runs in front of or behind the equalizer:
//   y       [B][NR][n_sym][fft]            complex   (all OFDM symbols, all subcarriers)
//   h_hat   [B][NR][NS][n_sym][n_sc]       complex   (effective subcarriers only)
//   err_var null, or [B][NR][NS][n_sym][n_sc] float  (variance of the estimation error)
//   x_hat   [B][NS][n_ds * n_sc] complex, no_eff the same in float
// One data resource element per thread; consecutive threads take consecutive subcarriers, so
// every load and store of a wave is one contiguous run. The noise covariance of receive antenna
// r is no + sum over streams of err_var (diagonal); NOISE: 0 scalar `no`, 1 no_dev[B],
// 2 no_dev[B][NR]. Without err_var and with one noise value per RE the unweighted form is used
// (the arithmetic of lmmse_direct, antenna by antenna).
template <int NR, int NS, int NOISE, bool ERRVAR>
__global__ __launch_bounds__(256) void lmmse_grid(const float2* __restrict__ y,
                                                  const float2* __restrict__ h_hat,
                                                  const float* __restrict__ err_var, float no_scalar,
                                                  const float* __restrict__ no_dev,
                                                  const int* __restrict__ data_sym,
                                                  const int* __restrict__ eff_sc,
                                                  float2* __restrict__ x_hat,
                                                  float* __restrict__ no_eff, int B, int n_sym, int fft,
                                                  int n_sc, int n_ds) {
  const int per_b = n_ds * n_sc;
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (long)B * per_b) return;
  const int b = (int)(n / per_b);
  const int d = (int)(n - (long)b * per_b);
  const int sd = d / n_sc, sc = d - sd * n_sc;
  const int sym = data_sym[sd], f = eff_sc[sc];
  constexpr bool WEIGHTED = ERRVAR || NOISE == 2;
  Gram<NS> g;
  g.clear();
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const size_t plane = (size_t)b * NR + r;
    const float2 yv = y[(plane * n_sym + sym) * fft + f];
    float hr[NS], hi[NS];
    const float no_r = NOISE == 0 ? no_scalar : (NOISE == 1 ? no_dev[b] : no_dev[(size_t)b * NR + r]);
    float ev = 0.f;  // summed over the streams first, then added to the noise variance
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const size_t at = ((plane * NS + k) * n_sym + sym) * n_sc + sc;
      // the estimates are read exactly once, in whole lines: nontemporal
      const float2 v = make_float2(__builtin_nontemporal_load(&h_hat[at].x), __builtin_nontemporal_load(&h_hat[at].y));
      hr[k] = v.x;
      hi[k] = v.y;
      if (ERRVAR) ev += __builtin_nontemporal_load(err_var + at);
    }
    gram_row<NS, WEIGHTED>(g, hr, hi, yv.x, yv.y, WEIGHTED ? 1.0f / (no_r + ev) : 1.0f);
  }
  const float lam = WEIGHTED ? 1.0f : (NOISE == 0 ? no_scalar : no_dev[b]);
  float xr[NS], xi[NS], ne[NS];
  lmmse_solve<NS>(g, lam, xr, xi, ne);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const size_t at = ((size_t)b * NS + k) * per_b + d;
    // whole lines per wave, written once: nontemporal (0.18 -> 0.14 ms per 8192 slots of 4x2)
    typedef float v2f __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(v2f{xr[k], xi[k]}, reinterpret_cast<v2f*>(x_hat + at));
    __builtin_nontemporal_store(ne[k], no_eff + at);
  }
}

// ---------------------------------------------------------------------------------------
// Tiled, LDS-DMA fed kernel (noise modes: scalar, per RE).

// LDS-DMA: each lane supplies its own 16-byte (or 4-byte) global source; the wave's data lands
// at lds_addr + lane*size. Issued through inline asm on purpose: hipcc orders every ds_read
// behind an LDS-DMA it knows about with s_waitcnt vmcnt(0), which would drain the prefetch
// of the next tile before the current one is computed. Hidden from the compiler, the DMA is
// ordered only by the counted waits below. M0 carries the LDS byte address and is restored.
__device__ inline unsigned lds_address(const void* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}
// NT: with the nontemporal hint, for operands whose 128-byte lines are read exactly once (a bare
// stream of 16 KiB tiles reads 7.2 instead of 6.3 TB/s with it, tools/probe/stream.hip). Lines
// that a later stage reads again (received vectors of fewer than 128 bytes) must not carry it:
// they would come from HBM twice.
#ifndef CET_DMA_NT
#define CET_DMA_NT 1  // 0: timing experiment without the hint anywhere
#endif
template <bool NT = true>
__device__ inline void dma16(const void* src, unsigned lds_addr_wave_uniform) {
  unsigned keep;
  if (NT && CET_DMA_NT)
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_addr_wave_uniform)
        : "memory");
  else
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_addr_wave_uniform)
        : "memory");
}
// The same with the source as a wave-uniform base (scalar registers) plus a 32-bit offset per lane:
// no 64-bit vector arithmetic per instruction.
template <bool NT = true>
__device__ inline void dma16_base(const void* base_wave_uniform, unsigned lane_offset,
                                  unsigned lds_addr_wave_uniform) {
  unsigned keep;
  if (NT && CET_DMA_NT)
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_offset), "s"(base_wave_uniform), "s"(lds_addr_wave_uniform)
        : "memory");
  else
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_offset), "s"(base_wave_uniform), "s"(lds_addr_wave_uniform)
        : "memory");
}
__device__ inline void dma4(const void* src, unsigned lds_addr_wave_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_addr_wave_uniform)
      : "memory");
}

// XOR applied to the chunk index of RE `t` (CH 16-byte chunks per RE): makes the 16 lanes of
// a ds_read_b128 lane group, which read the same chunk of 16 different REs, hit 16 different
// 16-byte bank slots.
template <int CH>
__device__ inline int chunk_swizzle(int t) {
  if (CH >= 16) return t & 15;
  if (CH == 8) return (t >> 1) & 7;
  if (CH == 4) return (t >> 2) & 3;
  if (CH == 2) return (t >> 3) & 1;
  return 0;
}

template <int N>
__device__ inline void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Register budget (waves per SIMD the allocator must leave room for). The shapes with 32 or
// more matrix entries per RE need a few registers more than the 256 of a two-wave budget; the
// spills that budget forces cost 7 % extra HBM traffic (scratch) for no gain in rate, so those
// shapes run one wave per SIMD (4 tiles of 20 KB in flight per CU; the kernel is bound by the
// read/write mix of the stream, not by the bytes in flight).
#ifndef CET_TILED_WPS
#define CET_TILED_WPS(NR, NS, LPR) ((NR) * (NS) >= 32 ? 1 : 2)
#endif
template <int NR, int NS, int LPR, bool PER_RE_NOISE>
__global__ __launch_bounds__(64, CET_TILED_WPS(NR, NS, LPR)) void lmmse_tiled(const float* __restrict__ y,
                                                  const float* __restrict__ H, float no_scalar,
                                                  const float* __restrict__ no_dev,
                                                  float* __restrict__ x_hat,
                                                  float* __restrict__ no_eff, long N,
                                                  long n_tiles, int hshare) {
  constexpr int T = 64 / LPR;             // REs per tile
  constexpr int HB = NR * NS * 8;         // bytes of H per RE
  constexpr int YB = NR * 8;              // bytes of y per RE
  constexpr int CH = HB / 16, CY = YB / 16;
  constexpr int H_TILE = T * HB, Y_TILE = T * YB;
  constexpr int H_INSTR = H_TILE / 1024, Y_INSTR = Y_TILE / 1024;
  constexpr int NO_TILE = PER_RE_NOISE ? 256 : 0;  // T floats, padded to one DMA instruction
  constexpr int BUF = H_TILE + Y_TILE + NO_TILE;
  constexpr int RPP = NR / LPR;           // receive antennas per lane
  static_assert(H_TILE % 1024 == 0 && Y_TILE % 1024 == 0, "tiles are whole DMA instructions");
  static_assert(RPP % 2 == 0, "a lane reads y two antennas at a time");
  static_assert((CH & (CH - 1)) == 0 && (CY & (CY - 1)) == 0, "power-of-two chunk counts");

  __shared__ __attribute__((aligned(16))) char smem[BUF];

  const int lane = threadIdx.x;
  const int t = lane & (T - 1);  // RE within the tile
  const int p = lane / T;        // which slice of the receive antennas

  // Byte offsets of this lane's DMA sources inside a tile: the same for every tile.
  unsigned hoff[H_INSTR], yoff[Y_INSTR];
#pragma unroll
  for (int i = 0; i < H_INSTR; ++i) {
    const int P = i * 64 + lane;  // LDS chunk position written by this lane
    const int tt = P / CH, q = P % CH;
    hoff[i] = tt * HB + ((q ^ chunk_swizzle<CH>(tt)) << 4);
  }
#pragma unroll
  for (int i = 0; i < Y_INSTR; ++i) {
    const int P = i * 64 + lane;
    const int tt = P / CY, q = P % CY;
    yoff[i] = tt * YB + ((q ^ chunk_swizzle<CY>(tt)) << 4);
  }

  auto fill = [&](long tile) {
    const long re0 = tile * T;
    const unsigned hdst = lds_address(smem);
    const unsigned ydst = hdst + H_TILE;
    const char* hb = reinterpret_cast<const char*>(H) + re0 * HB;
    const char* yb = reinterpret_cast<const char*>(y) + re0 * YB;
    const long last = N - 1 - re0;  // >= T-1 for a full tile
    if (hshare > 1) {
      // block fading: RE re0 + tt takes H of block (re0 + tt) / hshare. The block and position
      // of re0 are wave-uniform; tt + rem < T + hshare is small, so the float quotient is exact.
      const long blk0 = re0 / hshare;
      const int rem = (int)(re0 - blk0 * hshare);
      const float inv = 1.0f / (float)hshare;
      const char* hb0 = reinterpret_cast<const char*>(H) + blk0 * HB;
#pragma unroll
      for (int i = 0; i < H_INSTR; ++i) {
        int tt = (i * 64 + lane) / CH;
        const unsigned chunk = hoff[i] - (unsigned)tt * HB;
        if (tt > last) tt = (int)last;
        const int q = (int)(((float)(tt + rem) + 0.5f) * inv);
        dma16(hb0 + (size_t)q * HB + chunk, hdst + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < Y_INSTR; ++i) {
        unsigned off = yoff[i];
        const long tt = (i * 64 + lane) / CY;
        if (tt > last) off -= (unsigned)(tt - last) * YB;
        dma16(yb + off, ydst + i * 1024);
      }
    } else if (last >= T - 1) {
#pragma unroll
      for (int i = 0; i < H_INSTR; ++i) dma16(hb + hoff[i], hdst + i * 1024);
#pragma unroll
      for (int i = 0; i < Y_INSTR; ++i) dma16(yb + yoff[i], ydst + i * 1024);
    } else {
      // A tail tile re-reads the last RE for the missing ones (results are discarded).
#pragma unroll
      for (int i = 0; i < H_INSTR; ++i) {
        unsigned off = hoff[i];
        const long tt = (i * 64 + lane) / CH;
        if (tt > last) off -= (unsigned)(tt - last) * HB;
        dma16(hb + off, hdst + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < Y_INSTR; ++i) {
        unsigned off = yoff[i];
        const long tt = (i * 64 + lane) / CY;
        if (tt > last) off -= (unsigned)(tt - last) * YB;
        dma16(yb + off, ydst + i * 1024);
      }
    }
    if (PER_RE_NOISE) {
      long re = re0 + (lane & (T - 1));
      re = re < N ? re : N - 1;
      dma4(no_dev + re, ydst + Y_TILE);
    }
  };

  // This lane's operands of one tile: y of its RPP antennas (two per 16-byte chunk) and the
  // matching rows of H. Reading them all into registers first frees the LDS buffer for the
  // next DMA before the arithmetic starts, so two tiles per wave are in flight all the time.
  constexpr int HV = NS == 1 ? RPP / 2 : RPP * (NS / 2);
  float4 yreg[RPP / 2], hreg[HV];
  auto load_regs = [&]() {
    const char* hsrc = smem + t * HB;
    const char* ysrc = smem + H_TILE + t * YB;
    const int swh = chunk_swizzle<CH>(t), swy = chunk_swizzle<CY>(t);
#pragma unroll
    for (int rr = 0; rr < RPP; rr += 2) {
      const int r = p * RPP + rr;
      yreg[rr / 2] = *reinterpret_cast<const float4*>(ysrc + (((r >> 1) ^ swy) << 4));
      if (NS == 1) {
        hreg[rr / 2] = *reinterpret_cast<const float4*>(hsrc + (((r >> 1) ^ swh) << 4));
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < NS / 2; ++j) {
            const int c = (r + a) * (NS / 2) + j;
            hreg[(rr + a) * (NS / 2) + j] = *reinterpret_cast<const float4*>(hsrc + ((c ^ swh) << 4));
          }
      }
    }
  };
  float lam_reg = no_scalar;
  auto load_noise = [&]() {
    if (PER_RE_NOISE) lam_reg = *reinterpret_cast<const float*>(smem + H_TILE + Y_TILE + 4 * t);
  };

  const bool llr_mode = no_eff == nullptr;  // LLR entry point: one output array
  auto compute = [&](long tile) {
    Gram<NS> g;
    g.clear();
#pragma unroll
    for (int rr = 0; rr < RPP; rr += 2) {
      const float4 yv = yreg[rr / 2];
      float hr[2][NS], hi[2][NS];
      if (NS == 1) {
        const float4 hv = hreg[rr / 2];
        hr[0][0] = hv.x; hi[0][0] = hv.y; hr[1][0] = hv.z; hi[1][0] = hv.w;
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < NS / 2; ++j) {
            const float4 hv = hreg[(rr + a) * (NS / 2) + j];
            hr[a][2 * j] = hv.x; hi[a][2 * j] = hv.y;
            hr[a][2 * j + 1] = hv.z; hi[a][2 * j + 1] = hv.w;
          }
      }
      gram_row<NS, false>(g, hr[0], hi[0], yv.x, yv.y, 1.0f);
      gram_row<NS, false>(g, hr[1], hi[1], yv.z, yv.w, 1.0f);
    }
    // combine the LPR antenna slices (lanes t, t+T, ...)
#pragma unroll
    for (int off = T; off < 64; off <<= 1) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        g.diag[s] += __shfl_xor(g.diag[s], off);
        g.b[s].x += __shfl_xor(g.b[s].x, off);
        g.b[s].y += __shfl_xor(g.b[s].y, off);
      }
#pragma unroll
      for (int i = 0; i < Gram<NS>::NP; ++i) {
        g.off[i].x += __shfl_xor(g.off[i].x, off);
        g.off[i].y += __shfl_xor(g.off[i].y, off);
      }
    }
    const float lam = lam_reg;
    float xr[NS], xi[NS], ne[NS];
    lmmse_solve<NS>(g, lam, xr, xi, ne);
    if (llr_mode) to_llr<NS>(xr, xi, ne);
    const long re = tile * T + t;
    if (re < N) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      typedef float v2f __attribute__((ext_vector_type(2)));
      if (LPR == 2 && NS == 4) {
        // both lanes of an RE hold the full result: lane p writes streams 2p, 2p+1, so every
        // lane stores 16 B of x_hat and 8 B of no_eff and a wave instruction covers one
        // contiguous 1 KiB / 512 B block
        constexpr int S2 = NS >= 4 ? 2 : 0, S3 = NS >= 4 ? 3 : 0;  // (in range for any NS)
        const v4f xv = p ? v4f{xr[S2], xi[S2], xr[S3], xi[S3]} : v4f{xr[0], xi[0], xr[NS > 1 ? 1 : 0], xi[NS > 1 ? 1 : 0]};
        const v2f nv = p ? v2f{ne[S2], ne[S3]} : v2f{ne[0], ne[NS > 1 ? 1 : 0]};
        __builtin_nontemporal_store(xv, reinterpret_cast<v4f*>(x_hat + (re * NS + 2 * p) * 2));
        if (!llr_mode)
          __builtin_nontemporal_store(nv, reinterpret_cast<v2f*>(no_eff + re * NS + 2 * p));
      } else if (p == 0) {
        float* xd = x_hat + re * NS * 2;
        float* nd = no_eff + re * NS;
        if (NS % 2 == 0) {
#pragma unroll
          for (int s = 0; s < NS; s += 2)
            __builtin_nontemporal_store(v4f{xr[s], xi[s], xr[s + 1], xi[s + 1]},
                                        reinterpret_cast<v4f*>(xd + 2 * s));
        } else {
#pragma unroll
          for (int s = 0; s < NS; ++s)
            __builtin_nontemporal_store(v2f{xr[s], xi[s]}, reinterpret_cast<v2f*>(xd + 2 * s));
        }
        if (llr_mode) {
          // no_eff is not written
        } else if (NS == 4) {
          __builtin_nontemporal_store(v4f{ne[0], ne[1], ne[2], ne[3]}, reinterpret_cast<v4f*>(nd));
        } else if (NS == 2) {
          __builtin_nontemporal_store(v2f{ne[0], ne[NS > 1 ? 1 : 0]}, reinterpret_cast<v2f*>(nd));
        } else {
          __builtin_nontemporal_store(ne[0], nd);
        }
      }
    }
  };

  // One LDS buffer per wave, used as a staging area only. Iteration j: wait for tile j, copy
  // this lane's operands to registers, hand the buffer straight back to the DMA engine for
  // tile j + 1, then do the arithmetic of tile j from registers while that DMA is in flight.
  // vmcnt retires vector-memory instructions (loads, LDS-DMA and stores alike) in issue order;
  // the only instructions younger than the DMA of tile j are the N_ST output stores of tile
  // j - 1, which the wait leaves outstanding.
  constexpr int N_ST = 2;  // at least two store instructions per tile (x_hat, no_eff)
  const long stride = gridDim.x;
  long tile = blockIdx.x;
  if (tile < n_tiles) fill(tile);
  bool first = true;
  for (; tile < n_tiles; tile += stride) {
    if (first) wait_vmcnt<0>();
    else if (llr_mode) wait_vmcnt<1>();  // at least one store instruction per tile there
    else wait_vmcnt<N_ST>();
    first = false;
    load_regs();
    load_noise();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tile + stride < n_tiles) fill(tile + stride);
    compute(tile);
  }
}

// s_waitcnt vmcnt(n) for a run-time n (the count is an immediate; n above 32 waits for 32).
template <int LO, int HI>
__device__ inline void wait_vmcnt_range(int n) {
  if constexpr (LO == HI) {
    wait_vmcnt<LO>();
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (n <= MID) wait_vmcnt_range<LO, MID>(n);
    else wait_vmcnt_range<MID + 1, HI>(n);
  }
}
__device__ inline void wait_vmcnt_upto32(int n) { wait_vmcnt_range<0, 32>(n < 32 ? n : 32); }

// Result stores of the block-fading kernels: x_hat (or the LLRs) and no_eff of resource element
// `re`. With two lanes per block and four streams each lane stores two streams (p = which);
// otherwise lane p == 0 stores everything. Plain (cacheable) stores: a store instruction covers
// 16-byte pieces of many lines, which the L2 merges with the pieces of the block's other symbols
// before they go to HBM (nontemporal stores of such pieces halve the kernel's rate).
template <int NS, int LPR>
__device__ inline void store_block_results(float* __restrict__ x_hat, float* __restrict__ no_eff,
                                           long re, int p, const pair2 (&x)[NS],
                                           const float (&ne)[NS]) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v2f __attribute__((ext_vector_type(2)));
  const bool llr_mode = no_eff == nullptr;
  if constexpr (LPR == 2 && NS == 4) {
    const pair2 xa = p ? x[2] : x[0], xb = p ? x[3] : x[1];
    const v2f nv = p ? v2f{ne[2], ne[3]} : v2f{ne[0], ne[1]};
    *reinterpret_cast<v4f*>(x_hat + (re * NS + 2 * p) * 2) = v4f{xa.x, xa.y, xb.x, xb.y};
    if (!llr_mode) *reinterpret_cast<v2f*>(no_eff + re * NS + 2 * p) = nv;
  } else {
    if (p != 0) return;
    float* xd = x_hat + re * NS * 2;
    if constexpr (NS % 2 == 0) {
#pragma unroll
      for (int s = 0; s < NS; s += 2) {
        const pair2 xa = x[s], xb = x[s + 1];
        *reinterpret_cast<v4f*>(xd + 2 * s) = v4f{xa.x, xa.y, xb.x, xb.y};
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<v2f*>(xd + 2 * s) = x[s];
    }
    if (!llr_mode) {
      float* nd = no_eff + re * NS;
      if constexpr (NS == 4) {
        *reinterpret_cast<v4f*>(nd) = v4f{ne[0], ne[1], ne[2], ne[3]};
      } else if constexpr (NS == 2) {
        *reinterpret_cast<v2f*>(nd) = v2f{ne[0], ne[1]};
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) nd[s] = ne[s];
      }
    }
  }
}

// x_hat (or the LLRs) of element `e` of a tile into the tile's LDS staging area, which has the
// layout of the output itself.
template <int NS, int LPR>
__device__ inline void stage_block_results(char* xs, int e, int p, const pair2 (&x)[NS]) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v2f __attribute__((ext_vector_type(2)));
  float* xd = reinterpret_cast<float*>(xs) + (size_t)e * NS * 2;
  if constexpr (LPR == 2 && NS == 4) {
    const pair2 xa = p ? x[2] : x[0], xb = p ? x[3] : x[1];
    *reinterpret_cast<v4f*>(xd + 4 * p) = v4f{xa.x, xa.y, xb.x, xb.y};
  } else {
    if (p != 0) return;
    if constexpr (NS % 2 == 0) {
#pragma unroll
      for (int s = 0; s < NS; s += 2) {
        const pair2 xa = x[s], xb = x[s + 1];
        *reinterpret_cast<v4f*>(xd + 2 * s) = v4f{xa.x, xa.y, xb.x, xb.y};
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<v2f*>(xd + 2 * s) = x[s];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Block fading with a shared factorisation (scalar noise): one channel matrix per `hs`
// consecutive REs. A wave owns a tile of T blocks. Its work is a sequence of equally sized
// stages that pass through the wave's one LDS buffer: stage 0 of a tile brings the T matrices
// (H_TILE bytes), which go to registers and are factorised once (Gram, Cholesky, W = L^-1,
// no_eff); every further stage brings the received vectors of GS = NS symbols of the T blocks
// (the same number of bytes), and each symbol costs a matched filter b = H^H y plus two
// triangular products. As in lmmse_tiled a stage is copied to registers and the buffer handed
// back to the DMA engine at once, so a wave always has a stage in flight. HBM traffic per RE:
// y + x_hat + no_eff + (H / hs). The arithmetic is that of the per-RE kernels in the same
// order, so the results are the same bit for bit.
// CET_BLOCK_ABLATE: timing experiments only (results invalid): bit 0 no result stores, bit 1 a
// quarter of the matched filter.
#ifndef CET_BLOCK_ABLATE
#define CET_BLOCK_ABLATE 0
#endif
// A stage of received vectors is CET_BLOCK_GM times the size of the matrix stage. 2 (eight
// symbols of 16x4 in registers) spills and halves the rate.
#ifndef CET_BLOCK_GM
#define CET_BLOCK_GM 1
#endif
// One wave per SIMD for the large shapes: the matrices, a stage of received vectors and the
// factor live in registers (about 300 of the 512 a lone wave may use). A variant for two waves
// per SIMD (only the matrices and the factor in registers, the received vectors consumed from a
// ring of two to four LDS buffers) was slower on every shape, as was the streaming cache policy
// on the DMA: neither the vector unit (the packed arithmetic removed a seventh of its
// instructions for 3 % of the time) nor the bytes in flight bound this kernel. What did was the
// shape of its stores (see `stage_out` below); with whole-line stores 16x4 moves 5.0 TB/s, which
// is what a bare DMA stream with any share of stores reaches on this memory system.
template <int NR, int NS, int LPR>
__global__ __launch_bounds__(64) void lmmse_block(const float* __restrict__ y,
                                                  const float* __restrict__ H, float lam,
                                                  float* __restrict__ x_hat,
                                                  float* __restrict__ no_eff, long N, int hs,
                                                  long n_blocks, long n_tiles, int stage_out) {
  constexpr int T = 64 / LPR;             // blocks per tile
  constexpr int HB = NR * NS * 8;         // bytes of H per block
  constexpr int YB = NR * 8;              // bytes of y per RE
  constexpr int CH = HB / 16, CY = YB / 16;
  constexpr int H_TILE = T * HB, Y_SYM = T * YB;
  constexpr int GS = NS * CET_BLOCK_GM;   // symbols per y stage: GS * Y_SYM == GM * H_TILE
  constexpr int H_INSTR = H_TILE / 1024, YS_INSTR = Y_SYM / 1024;
  constexpr int RPP = NR / LPR;
  static_assert(H_TILE % 1024 == 0 && Y_SYM % 1024 == 0, "stages are whole DMA instructions");
  static_assert(RPP % 2 == 0, "a lane reads y two antennas at a time");
  // A stage brings GS * YB bytes of a block's received vectors; with fewer than 8 antennas a
  // 128-byte line is shared by two or more stages (measured with the hint on every DMA: 4x2
  // 0.043 -> 0.072 ms per 4.2 M REs, on the matrices alone 0.046; 16x2 0.146 -> 0.126).
  constexpr bool STREAM_NT = NR >= 8;

  __shared__ __attribute__((aligned(16))) char smem[CET_BLOCK_GM * H_TILE];
  // stage_out: the results of a tile (T blocks x hs symbols: one contiguous range of x_hat and one
  // of no_eff) leave as whole 1 KiB store instructions when the tile is done. x_hat is collected
  // here (T * hs * NS * 8 bytes), followed by the no_eff of each block (T * NS * 4 bytes; the same
  // for all symbols of a block). Stored symbol by symbol a wave's store instruction touches 64
  // lines with 16 or 32 bytes each, a block's line is completed over two stages and a quarter of
  // the lines left the L2 before that (58.5 B/RE written for 48): 0.416 against 0.358 ms per
  // 8.4 M REs of 16x4, and twice the time on the small shapes.
  extern __shared__ __attribute__((aligned(16))) char out_stage[];
  char* const ne_stage = out_stage + (size_t)T * hs * NS * 8;

  const int lane = threadIdx.x;
  const int t = lane & (T - 1);  // block within the tile
  const int p = lane / T;        // which slice of the receive antennas
  const bool llr_mode = no_eff == nullptr;

  // Symbols a tile works on: those of its first block (every block but the very last one of the
  // array is complete).
  auto symbols_of = [&](long tile) {
    const long left = N - tile * T * hs;  // REs from the tile's first block on (> 0)
    return left < hs ? (int)left : hs;
  };

  // stage st of a tile: 0 = the matrices, g + 1 = symbols g*GS .. of every block
  // Offsets of this lane's 16-byte pieces inside a tile, for the DMA instructions of a stage: the
  // source of a complete tile is a wave-uniform base (scalar arithmetic) plus these, so that an
  // instruction costs no vector address arithmetic (it was 16 to 20 vector instructions each, a
  // third of the kernel's vector work). Tiles at the ragged end take the general path.
  unsigned yoff[YS_INSTR];
#pragma unroll
  for (int i = 0; i < YS_INSTR; ++i) {
    const int P = i * 64 + lane;
    const int tt = P / CY, q = P % CY;
    yoff[i] = (unsigned)tt * (unsigned)hs * YB + ((q ^ chunk_swizzle<CY>(tt)) << 4);
  }
  const bool small_tiles = (long)T * hs * YB < (1l << 31);  // the offsets fit 32 bits

  auto issue = [&](long tile, int st) {
    const unsigned dst = lds_address(smem);
    const long blk0 = tile * T;
    const int kmax = symbols_of(tile);
    const bool whole = small_tiles && (blk0 + T) * hs <= N;  // every block and symbol of the tile exists
    if (st == 0) {
      if (whole) {
        const char* base = reinterpret_cast<const char*>(H) + blk0 * HB;
#pragma unroll
        for (int i = 0; i < H_INSTR; ++i) {
          const int P = i * 64 + lane;
          const int tt = P / CH, q = P % CH;
          dma16_base<STREAM_NT>(base, (unsigned)(tt * HB + ((q ^ chunk_swizzle<CH>(tt)) << 4)), dst + i * 1024);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < H_INSTR; ++i) {
        const int P = i * 64 + lane;
        const int tt = P / CH, q = P % CH;
        long blk = blk0 + tt;
        blk = blk < n_blocks ? blk : n_blocks - 1;  // tail tile: re-read the last block
        dma16<STREAM_NT>(reinterpret_cast<const char*>(H) + blk * HB + ((q ^ chunk_swizzle<CH>(tt)) << 4),
                    dst + i * 1024);
      }
    } else {
      const int k0 = (st - 1) * GS;
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        if (k0 + j < kmax) {
          if (whole) {
            const char* base = reinterpret_cast<const char*>(y) + (blk0 * hs + k0 + j) * YB;
#pragma unroll
            for (int i = 0; i < YS_INSTR; ++i) dma16_base<STREAM_NT>(base, yoff[i], dst + (j * YS_INSTR + i) * 1024);
            continue;
          }
#pragma unroll
          for (int i = 0; i < YS_INSTR; ++i) {
            const int P = i * 64 + lane;
            const int tt = P / CY, q = P % CY;
            long re = (blk0 + tt) * hs + k0 + j;
            re = re < N ? re : N - 1;  // ragged end: results of these lanes are not stored
            dma16<STREAM_NT>(reinterpret_cast<const char*>(y) + re * YB + ((q ^ chunk_swizzle<CY>(tt)) << 4),
                        dst + (j * YS_INSTR + i) * 1024);
          }
        }
      }
    }
  };

  constexpr int HV = NS == 1 ? RPP / 2 : RPP * (NS / 2);
  float4 hreg[HV];            // this lane's rows of its block's matrix, for the whole tile
  float4 yreg[GS][RPP / 2];   // this lane's antennas of the stage's symbols
  Factor<NS> f;
  const int swh = chunk_swizzle<CH>(t), swy = chunk_swizzle<CY>(t);

  // rows rr and rr + 1 of this lane's part of the matrix
  auto h_rows = [&](int rr, pair2 (&h)[2][NS]) {
    if (NS == 1) {
      const float4 hv = hreg[rr / 2];
      h[0][0] = pair2{hv.x, hv.y};
      h[1][0] = pair2{hv.z, hv.w};
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NS / 2; ++j) {
          const float4 hv = hreg[(rr + a) * (NS / 2) + j];
          h[a][2 * j] = pair2{hv.x, hv.y};
          h[a][(2 * j + 1) % NS] = pair2{hv.z, hv.w};
        }
    }
  };

  auto copy_matrices = [&]() {
    const char* hsrc = smem + t * HB;
#pragma unroll
    for (int rr = 0; rr < RPP; rr += 2) {
      const int r = p * RPP + rr;
      if (NS == 1) {
        hreg[rr / 2] = *reinterpret_cast<const float4*>(hsrc + (((r >> 1) ^ swh) << 4));
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < NS / 2; ++j) {
            const int c = (r + a) * (NS / 2) + j;
            hreg[(rr + a) * (NS / 2) + j] = *reinterpret_cast<const float4*>(hsrc + ((c ^ swh) << 4));
          }
      }
    }
  };
  auto copy_symbols = [&](int nsym) {
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      if (j < nsym) {
        const char* ysrc = smem + j * Y_SYM + t * YB;
#pragma unroll
        for (int rr = 0; rr < RPP; rr += 2) {
          const int r = p * RPP + rr;
          yreg[j][rr / 2] = *reinterpret_cast<const float4*>(ysrc + (((r >> 1) ^ swy) << 4));
        }
      }
    }
  };

  auto factorise = [&]() {
    Gram<NS> g;
    g.clear();
#pragma unroll
    for (int rr = 0; rr < RPP; rr += 2) {
      pair2 h[2][NS];
      h_rows(rr, h);
      gram_row_h<NS, false>(g, h[0], 1.0f);
      gram_row_h<NS, false>(g, h[1], 1.0f);
    }
#pragma unroll
    for (int off = T; off < 64; off <<= 1) {
#pragma unroll
      for (int s = 0; s < NS; ++s) g.diag[s] += __shfl_xor(g.diag[s], off);
#pragma unroll
      for (int i = 0; i < Gram<NS>::NP; ++i) {
        g.off[i].x += __shfl_xor(g.off[i].x, off);
        g.off[i].y += __shfl_xor(g.off[i].y, off);
      }
    }
    lmmse_factor<NS>(g, lam, f);
    if (stage_out && !llr_mode && p == 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) reinterpret_cast<float*>(ne_stage)[t * NS + s] = f.ne[s];
    }
  };

  // The matched filter takes every matrix entry as a broadcast pair (re, re) / (im, im), which a
  // packed instruction forms by itself (half-select modifiers). Seen as common to all symbols,
  // those pairs would be built once with moves and kept (or rebuilt with moves per symbol);
  // passing the matrix through an empty asm per symbol makes them the symbol's own.
  auto launder_matrix = [&]() {
#pragma unroll
    for (int k = 0; k < HV; ++k) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      v4f v{hreg[k].x, hreg[k].y, hreg[k].z, hreg[k].w};
      asm volatile("" : "+v"(v));
      hreg[k] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  // Store instructions per symbol when every lane is in range.
  constexpr int ST_X = (LPR == 2 && NS == 4) ? 1 : (NS % 2 == 0 ? NS / 2 : NS);
  auto equalise = [&](long tile, int k0, int nsym) {
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      if (j >= nsym) break;
      launder_matrix();
      Gram<NS> g;
      g.clear();
#pragma unroll
      for (int rr = 0; rr < ((CET_BLOCK_ABLATE & 2) ? 2 : RPP); rr += 2) {
        const float4 yv = yreg[j][rr / 2];
        pair2 h[2][NS];
        h_rows(rr, h);
        gram_row_b<NS, false>(g, h[0], pair2{yv.x, yv.y}, 1.0f);
        gram_row_b<NS, false>(g, h[1], pair2{yv.z, yv.w}, 1.0f);
      }
#pragma unroll
      for (int off = T; off < 64; off <<= 1) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          g.b[s].x += __shfl_xor(g.b[s].x, off);
          g.b[s].y += __shfl_xor(g.b[s].y, off);
        }
      }
      pair2 x[NS];
      float ne[NS];
      lmmse_apply<NS>(f, g.b, x);
#pragma unroll
      for (int s = 0; s < NS; ++s) ne[s] = f.ne[s];
      if (llr_mode) to_llr<NS>(x, ne);
      const long blk = tile * T + t;
      const long re = blk * hs + k0 + j;
      if (stage_out) {
        stage_block_results<NS, LPR>(out_stage, t * hs + k0 + j, p, x);
      } else if (blk < n_blocks && re < N && (!(CET_BLOCK_ABLATE & 1) || lam < -1.0f)) {
        store_block_results<NS, LPR>(x_hat, no_eff, re, p, x, ne);
      }
    }
  };
  // The tile's staged results to global memory: 16 bytes per lane and instruction, contiguous.
  // Returns the number of store instructions issued.
  const float inv_hs = 1.0f / (float)hs;
  auto flush_tile = [&](long tile) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const long re0 = tile * T * hs;
    const long n_re = N - re0 < (long)T * hs ? N - re0 : (long)T * hs;  // elements of this tile
    const int xb = (int)n_re * NS * 8, nb = llr_mode ? 0 : (int)n_re * NS * 4;
    char* xg = reinterpret_cast<char*>(x_hat) + re0 * NS * 8;
    // xb is a multiple of 8 and nb of 4 only (one or two streams, ragged last tile): the piece
    // that holds the end of the range is written float by float up to the range's last byte.
    for (int o = lane * 16; o < xb; o += 1024) {
      if (o + 16 <= xb) {
        __builtin_nontemporal_store(*reinterpret_cast<const v4f*>(out_stage + o), reinterpret_cast<v4f*>(xg + o));
      } else {
        for (int b = o; b < xb; b += 4)
          *reinterpret_cast<float*>(xg + b) = *reinterpret_cast<const float*>(out_stage + b);
      }
    }
    if (nb) {
      char* ng = reinterpret_cast<char*>(no_eff) + re0 * NS * 4;
      const float* nsrc = reinterpret_cast<const float*>(ne_stage);
      constexpr int RP = 4 / NS;  // elements per 16-byte piece (NS is 1, 2 or 4)
      for (int o = lane * 16; o < nb; o += 1024) {
        float v[4];
#pragma unroll
        for (int r = 0; r < RP; ++r) {
          // block of element e: e / hs (e < 2^16: the rounded product is off by < 1e-3, the
          // quotient of e + 0.5 at least 0.5 / hs from an integer)
          const int e = o / (NS * 4) + r;
          const int blk = (int)(((float)e + 0.5f) * inv_hs);
#pragma unroll
          for (int s2 = 0; s2 < NS; ++s2) v[r * NS + s2] = nsrc[blk * NS + s2];
        }
        if (o + 16 <= nb) {
          __builtin_nontemporal_store(v4f{v[0], v[1], v[2], v[3]}, reinterpret_cast<v4f*>(ng + o));
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (o + q * 4 < nb) *reinterpret_cast<float*>(ng + o + q * 4) = v[q];
        }
      }
    }
    return (xb + 1023) / 1024 + (nb + 1023) / 1024;
  };

  // Stage pipeline. vmcnt retires vector-memory instructions in issue order; the instructions
  // younger than the DMA of the stage being waited for are the stores of the stage before it
  // (none after a matrix stage), which the wait leaves outstanding: waiting for them as well
  // would stall every stage for a store round trip. `younger` is a lower bound of their number:
  // exact for a tile whose T blocks are all complete (every lane stores in every instruction),
  // 1 otherwise (a store whose lanes are all out of range is branched over; the tile's first
  // block and a stage's first symbol always exist).
  long tile = blockIdx.x;
  int st = 0, younger = 0;
  if (tile < n_tiles) issue(tile, 0);
  while (tile < n_tiles) {
    if (CET_BLOCK_ABLATE & 1) younger = 0;
    wait_vmcnt_upto32(younger);
    const int kmax = symbols_of(tile);
    const int n_here = 1 + (kmax + GS - 1) / GS;  // stages of this tile
    const int k0 = (st - 1) * GS;
    const int nsym = st == 0 ? 0 : (kmax - k0 < GS ? kmax - k0 : GS);
    if (st == 0) copy_matrices();
    else copy_symbols(nsym);
    long ntile = tile;
    int nst = st + 1;
    if (nst >= n_here) {
      nst = 0;
      ntile = tile + gridDim.x;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ntile < n_tiles) issue(ntile, nst);
    if (st == 0) {
      factorise();
      younger = 0;
    } else {
      equalise(tile, k0, nsym);
      const bool full_tile = (tile * T + T) * hs <= N;
      if (stage_out) {
        younger = 0;
        if (st == n_here - 1) {  // the tile is complete
          const int issued = flush_tile(tile);
          younger = full_tile ? issued : 1;
        }
      } else {
        younger = full_tile ? nsym * (ST_X + (llr_mode ? 0 : 1)) : 1;
      }
    }
    tile = ntile;
    st = nst;
  }
}

// Output staging of the block-fading kernel is used while the stage buffer and a tile's results
// fit this many bytes of LDS together (16x4, 14 symbols per block: 16 + 14.5 KiB; 8x4: 16 + 29).
constexpr int BLOCK_LDS_MAX = 64 * 1024;

template <int NR, int NS, int LPR>
int launch_block(const float* y, const float* H, float no_scalar, float* x_hat, float* no_eff,
                 long N, int hshare, hipStream_t stream) {
  constexpr int T = 64 / LPR;
  const long n_blocks = (N + hshare - 1) / hshare;
  const long n_tiles = (n_blocks + T - 1) / T;
  int dev = 0, cus = 256;
  CET_HIP(hipGetDevice(&dev));
  CET_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const auto kernel = &lmmse_block<NR, NS, LPR>;
  // Results staged through LDS when a tile's fit: faster on every shape, also where the staging
  // area lowers the number of resident waves (8x4: two per CU instead of four).
  const long out_bytes = (long)T * NS * (hshare * 8l + 4);
  bool stage = out_bytes <= BLOCK_LDS_MAX - T * NR * NS * 8;
#ifdef CET_EXP  // experiment build only: CET_BLOCK_STAGE=0 switches the staging off
  if (const char* e = getenv("CET_BLOCK_STAGE")) stage = stage && e[0] == '1';
#endif
  const size_t dyn = stage ? (size_t)((out_bytes + 1023) / 1024 * 1024) : 0;
  // persistent waves: as many as are resident at once (registers and LDS decide), at most 8 per
  // CU; the answer for the last staging size is kept (a property of the kernel otherwise)
  static std::atomic<long> cached{-1};  // dyn * 16 + waves per CU
  const long c = cached.load(std::memory_order_relaxed);
  int per_cu = c >= 0 && (size_t)(c >> 4) == dyn ? (int)(c & 15) : 0;
  if (!per_cu) {
    int nb = 0;
    CET_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kernel), 64, dyn));
    per_cu = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
    cached.store((long)dyn * 16 + per_cu, std::memory_order_relaxed);
  }
  long grid = (long)cus * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64), dyn, stream, y, H, no_scalar, x_hat,
                     no_eff, N, hshare, n_blocks, n_tiles, stage ? 1 : 0);
  CET_HIP(hipGetLastError());
  return CET_OK;
}

template <int NR, int NS, int LPR>
int launch_tiled(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
                 float* x_hat, float* no_eff, long N, int hshare, hipStream_t stream) {
  constexpr int T = 64 / LPR;
  const long n_tiles = (N + T - 1) / T;
  int dev = 0, cus = 256;
  CET_HIP(hipGetDevice(&dev));
  CET_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // persistent waves: as many as are resident at once (registers and LDS decide), at most 8 per CU
  static std::atomic<int> per_cu_cached{0};  // a property of the kernel's registers and LDS
  int per_cu = per_cu_cached.load(std::memory_order_relaxed);
  if (!per_cu) {
    int nb = 0;
    CET_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &nb, reinterpret_cast<const void*>(&lmmse_tiled<NR, NS, LPR, false>), 64, 0));
    per_cu = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
    per_cu_cached.store(per_cu, std::memory_order_relaxed);
  }
  long grid = (long)cus * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  if (no_mode == CET_NO_SCALAR) {
    hipLaunchKernelGGL((lmmse_tiled<NR, NS, LPR, false>), dim3((unsigned)grid), dim3(64), 0, stream,
                       y, H, no_scalar, no_dev, x_hat, no_eff, N, n_tiles, hshare);
  } else {
    hipLaunchKernelGGL((lmmse_tiled<NR, NS, LPR, true>), dim3((unsigned)grid), dim3(64), 0, stream,
                       y, H, no_scalar, no_dev, x_hat, no_eff, N, n_tiles, hshare);
  }
  CET_HIP(hipGetLastError());
  return CET_OK;
}

template <int NR, int NS>
int launch_direct(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
                  float* x_hat, float* no_eff, long N, int hshare, hipStream_t stream) {
  const dim3 block(256), grid((unsigned)((N + 255) / 256));
  switch (no_mode) {
    case CET_NO_SCALAR:
      hipLaunchKernelGGL((lmmse_direct<NR, NS, CET_NO_SCALAR>), grid, block, 0, stream, y, H,
                         no_scalar, no_dev, x_hat, no_eff, N, hshare);
      break;
    case CET_NO_PER_RE:
      hipLaunchKernelGGL((lmmse_direct<NR, NS, CET_NO_PER_RE>), grid, block, 0, stream, y, H,
                         no_scalar, no_dev, x_hat, no_eff, N, hshare);
      break;
    default:
      hipLaunchKernelGGL((lmmse_direct<NR, NS, CET_NO_PER_RE_RX>), grid, block, 0, stream, y, H,
                         no_scalar, no_dev, x_hat, no_eff, N, hshare);
  }
  CET_HIP(hipGetLastError());
  return CET_OK;
}

// LPR: lanes per RE (per block in the block-fading kernel; the two kernels use the same value,
// which fixes the summation order of the Gram matrix and makes their results bit-identical).
// Two lanes for the four-stream shapes: half the registers per lane and, in the block-fading
// kernel, half the blocks and so half the result staging area per wave, which is what bounds the
// resident waves there. Measured (share of 8 TB/s, one / two lanes): block fading 8x4 0.56 / 0.65,
// 4x4 0.68 / 0.67; per RE 8x4 0.71 / 0.70, 4x4 0.70 / 0.74; two-stream shapes lose with two.
template <int NR, int NS>
constexpr int lanes_per_re() {
  return (NS == 4 && NR >= 4) ? 2 : 1;
}

template <int NR, int NS>
int dispatch(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
             float* x_hat, float* no_eff, long N, int hshare, int variant, hipStream_t stream) {
  if (variant == 1 || no_mode == CET_NO_PER_RE_RX)
    return launch_direct<NR, NS>(y, H, no_mode, no_scalar, no_dev, x_hat, no_eff, N, hshare, stream);
  constexpr int LPR = lanes_per_re<NR, NS>();
  // block fading with scalar noise: shared factorisation; the other noise modes run the per-RE
  // kernel with gathered matrices (the two are each other's cross-check)
  if (hshare > 1 && no_mode == CET_NO_SCALAR)
    return launch_block<NR, NS, LPR>(y, H, no_scalar, x_hat, no_eff, N, hshare, stream);
#ifdef CET_EXP  // experiment build only: one lane per RE instead of two (CET_LMMSE_LPR=1)
  static const int lpr_override = [] {
    const char* e = getenv("CET_LMMSE_LPR");
    return e ? atoi(e) : 0;
  }();
  if (LPR == 2 && lpr_override == 1)
    return launch_tiled<NR, NS, 1>(y, H, no_mode, no_scalar, no_dev, x_hat, no_eff, N, hshare, stream);
#endif
  return launch_tiled<NR, NS, LPR>(y, H, no_mode, no_scalar, no_dev, x_hat, no_eff, N, hshare, stream);
}

}  // namespace

template <int NR, int NS>
static int grid_dispatch(const float* y, const float* h_hat, const float* err_var, int no_mode,
                         float no_scalar, const float* no_dev, const int* data_sym, const int* eff_sc,
                         float* x_hat, float* no_eff, int B, int n_sym, int fft, int n_sc, int n_ds,
                         hipStream_t stream) {
  const long n = (long)B * n_ds * n_sc;
  // (two subcarriers per thread with 16-byte accesses run at the same 5.0 TB/s: not kept)
  const dim3 block(256), grid((unsigned)((n + 255) / 256));
#define CET_GRID(NOISE, EV)                                                                      \
  hipLaunchKernelGGL((lmmse_grid<NR, NS, NOISE, EV>), grid, block, 0, stream,                     \
                     reinterpret_cast<const float2*>(y), reinterpret_cast<const float2*>(h_hat),  \
                     err_var, no_scalar, no_dev, data_sym, eff_sc,                                \
                     reinterpret_cast<float2*>(x_hat), no_eff, B, n_sym, fft, n_sc, n_ds)
  if (err_var) {
    if (no_mode == 0) CET_GRID(0, true);
    else if (no_mode == 1) CET_GRID(1, true);
    else CET_GRID(2, true);
  } else {
    if (no_mode == 0) CET_GRID(0, false);
    else if (no_mode == 1) CET_GRID(1, false);
    else CET_GRID(2, false);
  }
#undef CET_GRID
  CET_HIP(hipGetLastError());
  return CET_OK;
}

int lmmse_grid_launch(const float* y, const float* h_hat, const float* err_var, int no_mode,
                      float no_scalar, const float* no_dev, const int* data_sym, const int* eff_sc,
                      float* x_hat, float* no_eff, int B, int Nr, int Ns, int n_sym, int fft, int n_sc,
                      int n_ds, hipStream_t stream) {
#define CET_CASE(NR, NS)         \
  if (Nr == NR && Ns == NS)      \
    return grid_dispatch<NR, NS>(y, h_hat, err_var, no_mode, no_scalar, no_dev, data_sym, eff_sc, \
                                 x_hat, no_eff, B, n_sym, fft, n_sc, n_ds, stream);
  CET_CASE(16, 4) CET_CASE(16, 2) CET_CASE(16, 1)
  CET_CASE(8, 4) CET_CASE(8, 2) CET_CASE(8, 1)
  CET_CASE(4, 4) CET_CASE(4, 2) CET_CASE(4, 1)
  CET_CASE(2, 2) CET_CASE(2, 1)
#undef CET_CASE
  return CET_ERR_UNSUPPORTED;
}

int lmmse_launch(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
                 float* x_hat, float* no_eff, long N, int Nr, int Ns, int hshare, int variant,
                 hipStream_t stream) {
#define CET_CASE(NR, NS) \
  if (Nr == NR && Ns == NS) \
    return dispatch<NR, NS>(y, H, no_mode, no_scalar, no_dev, x_hat, no_eff, N, hshare, variant, stream);
  CET_CASE(16, 4) CET_CASE(16, 2) CET_CASE(16, 1)
  CET_CASE(8, 4) CET_CASE(8, 2) CET_CASE(8, 1)
  CET_CASE(4, 4) CET_CASE(4, 2) CET_CASE(4, 1)
  CET_CASE(2, 2) CET_CASE(2, 1)
#undef CET_CASE
  return CET_ERR_UNSUPPORTED;
}

}  // namespace cet
