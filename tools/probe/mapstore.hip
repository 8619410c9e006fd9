// What a wave-wide global store costs for the shapes the attention-map stores of the Transformer
// kernel can take. Every wave writes "maps" of 8 heads x 25 rows x 25 floats (20,000 B, the
// first encoder layer) over and over into its own 30 KB slot of a buffer far larger than the
// caches, with 8 waves per CU as in the workgroup kernel, in one of these forms:
//   0  rows of 25 floats, lane (c, g) writes 16 B at row c, float 4 g: unaligned 16-byte stores
//      (what the kernel does; ragged row ends as single floats)
//   1  the same rows padded to 28 floats: every 16-byte store is aligned, no ragged ends
//   2  the same bytes as fully coalesced 16-byte stores (lane l writes bytes 16 l .. 16 l + 15 of a
//      1 KiB piece): what an LDS-staged flush would issue
//   3  form 0 with single-float stores only
// Prints clocks per store instruction per wave and bytes per clock per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o mapstore mapstore.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

template <int FORM>
__global__ __launch_bounds__(512) void stores(float* buf, long slot_floats, int iters, unsigned long long* clocks,
                                               int* n_instr) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const long wave = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
  float* base = buf + wave * slot_floats;
  const f32x4 v{1.f, 2.f, 3.f, 4.f};
  const int L = 25, LP = FORM == 1 ? 28 : 25;
  int instr = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    float* m = base + (it & 3) * (8 * 25 * 28);  // four map sets per slot, so lines are not rewritten at once
    if (FORM == 2) {
      for (int k = 0; k < 8 * L * L / 256; ++k) {
        *reinterpret_cast<f32x4*>(m + 256 * k + 4 * lane) = v;
        ++instr;
      }
    } else {
      for (int h = 0; h < 8; ++h) {
        float* hb = m + h * L * LP;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const int q = 16 * qt + c;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            const int key0 = 16 * kt + 4 * g;
            if (FORM == 3) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (q < L && key0 + i < L) hb[q * LP + key0 + i] = v[i];
                ++instr;
              }
            } else if (FORM == 1) {
              if (q < L && key0 < LP) *reinterpret_cast<f32x4*>(hb + q * LP + key0) = v;
              ++instr;
            } else {
              if (q < L && key0 + 3 < L) *reinterpret_cast<f32x4_u*>(hb + q * LP + key0) = v;
              ++instr;
              if (kt == 1) {
                if (q < L && key0 == 24) hb[q * LP + 24] = v[0];
                ++instr;
              }
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) clocks[wave] = t1 - t0;
  if (wave == 0 && lane == 0) *n_instr = instr;
}

template <int FORM>
static void run(const char* name, float* buf, long slot, int blocks, int iters, unsigned long long* dclk, int* dn) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  stores<FORM><<<blocks, 512>>>(buf, slot, iters, dclk, dn);
  hipEventRecord(e0);
  stores<FORM><<<blocks, 512>>>(buf, slot, iters, dclk, dn);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int waves = blocks * 8;
  unsigned long long* h = (unsigned long long*)malloc(waves * sizeof *h);
  int n = 0;
  hipMemcpy(h, dclk, waves * sizeof *h, hipMemcpyDeviceToHost);
  hipMemcpy(&n, dn, sizeof n, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < waves; ++i) sum += (double)h[i];
  const double per_wave = sum / waves;
  const double bytes = 8.0 * 25 * 25 * 4 * iters;  // useful bytes per wave
  printf("%-28s %6d store instr/wave  %8.1f clocks/instr  %6.2f useful B/clk/CU  %7.3f ms  %6.2f TB/s useful\n", name, n,
         per_wave / n, 8 * bytes / per_wave, ms, bytes * waves / ms / 1e9);
  free(h);
}

int main() {
  const int blocks = 256, iters = 64;
  const long slot = 4 * 8 * 25 * 28 + 64;  // floats per wave
  float* buf;
  unsigned long long* dclk;
  int* dn;
  hipMalloc(&buf, (size_t)blocks * 8 * slot * sizeof(float));
  hipMalloc(&dclk, blocks * 8 * sizeof *dclk);
  hipMalloc(&dn, sizeof *dn);
  run<0>("unaligned x4, rows of 25", buf, slot, blocks, iters, dclk, dn);
  run<1>("aligned x4, rows of 28", buf, slot, blocks, iters, dclk, dn);
  run<2>("coalesced x4", buf, slot, blocks, iters, dclk, dn);
  run<3>("single floats, rows of 25", buf, slot, blocks, iters, dclk, dn);
  // one wave per SIMD instead of two
  printf("-- 128 blocks (half the CUs):\n");
  run<0>("unaligned x4, rows of 25", buf, slot, 128, iters, dclk, dn);
  run<2>("coalesced x4", buf, slot, 128, iters, dclk, dn);
  return 0;
}
