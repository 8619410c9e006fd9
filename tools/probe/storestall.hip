// How long a store instruction holds up a wave that is otherwise busy with MFMAs: the memory
// system is almost idle here (a few stores per microsecond and CU), so what shows is the issue
// cost of the store as the wave sees it, not bandwidth. A wave runs ITER rounds of NM dependent
// MFMAs followed by NS stores of one form, WPS waves per SIMD:
//   0 no stores   1 16 B per lane, rows of 25 floats (unaligned)   2 16 B per lane, coalesced 1 KiB
//   3 4 B per lane, rows of 25 floats   4 form 1 with three quarters of the lanes dropped by a
//   buffer bounds check   5 form 1 through a buffer descriptor, all lanes   6 form 1, 16 lanes only (exec)
// Prints clocks per round and the difference to form 0 per store instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o storestall storestall.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int FORM, int NM, int NS>
__global__ __launch_bounds__(512) void k(float* buf, int iters, unsigned long long* clocks, float* sink) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const long wave = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  float* base = buf + wave * 16384;  // 64 KiB per wave
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 65536, 0x00020000);
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  f32x4 v{1.f, 2.f, 3.f, 4.f};
  const float a = (float)lane, b = 1.0f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    float* p = base + (it & 7) * 2048;
    const unsigned row = (unsigned)(c * 25 + 4 * g);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (FORM == 1) *reinterpret_cast<f32x4_u*>(p + row + 400 * (s & 3)) = v;
      if (FORM == 2) *reinterpret_cast<f32x4*>(p + 4 * lane + 256 * (s & 3)) = v;
      if (FORM == 3) p[row + 400 * (s & 3)] = v[0];
      if (FORM == 4)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, g == 0 ? (row + 400 * (s & 3)) * 4u : 0x80000000u,
                                               (unsigned)(it & 7) * 8192u, 0);
      if (FORM == 5)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (row + 400 * (s & 3)) * 4u,
                                               (unsigned)(it & 7) * 8192u, 0);
      if (FORM == 6) {
        if (g == 0) *reinterpret_cast<f32x4_u*>(p + row + 400 * (s & 3)) = v;
      }
    }
    v += acc * 1e-30f;  // the stored values depend on the MFMAs, as in the kernel
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) clocks[wave] = t1 - t0;
  if (acc[0] == 12345.f) sink[0] = acc[1];
}

template <int FORM, int NM, int NS>
static double run(float* buf, int waves_per_block, unsigned long long* dclk, float* sink) {
  const int blocks = 256, iters = 2000;
  k<FORM, NM, NS><<<blocks, 64 * waves_per_block>>>(buf, iters, dclk, sink);
  k<FORM, NM, NS><<<blocks, 64 * waves_per_block>>>(buf, iters, dclk, sink);
  (void)hipDeviceSynchronize();
  static unsigned long long h[256 * 8];
  const int waves = blocks * waves_per_block;
  (void)hipMemcpy(h, dclk, waves * sizeof *h, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < waves; ++i) sum += (double)h[i];
  return sum / waves / iters;
}

template <int NM, int NS>
static void table(float* buf, int wpb, unsigned long long* dclk, float* sink) {
  const double base = run<0, NM, NS>(buf, wpb, dclk, sink);
  const double f1 = run<1, NM, NS>(buf, wpb, dclk, sink), f2 = run<2, NM, NS>(buf, wpb, dclk, sink);
  const double f3 = run<3, NM, NS>(buf, wpb, dclk, sink), f4 = run<4, NM, NS>(buf, wpb, dclk, sink);
  const double f5 = run<5, NM, NS>(buf, wpb, dclk, sink), f6 = run<6, NM, NS>(buf, wpb, dclk, sink);
  printf("%d waves/CU, %2d MFMAs + %d stores per round: %7.0f clocks without stores; per store: x4 rows %6.1f  "
         "x4 coalesced %6.1f  dword rows %6.1f  buffer x4 quarter %6.1f  buffer x4 all %6.1f  x4 rows 16 lanes %6.1f\n",
         wpb, NM, NS, base, (f1 - base) / NS, (f2 - base) / NS, (f3 - base) / NS, (f4 - base) / NS, (f5 - base) / NS,
         (f6 - base) / NS);
}

int main() {
  float *buf, *sink;
  unsigned long long* dclk;
  (void)hipMalloc(&buf, (size_t)256 * 8 * 65536);
  (void)hipMalloc(&sink, 64);
  (void)hipMalloc(&dclk, 256 * 8 * sizeof *dclk);
  table<32, 1>(buf, 4, dclk, sink);
  table<32, 4>(buf, 4, dclk, sink);
  table<32, 8>(buf, 4, dclk, sink);
  table<32, 4>(buf, 8, dclk, sink);
  table<32, 8>(buf, 8, dclk, sink);
  table<8, 4>(buf, 4, dclk, sink);
  return 0;
}
