// Measures issue cost of v_mfma_f32_16x16x4_f32 / 32x32x2 in dependent and interleaved chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS, int N>
__global__ __launch_bounds__(512) void k16(float* o, unsigned long long* t, float a, float b) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float av = a + threadIdx.x, bv = b + threadIdx.x;
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  o[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
template <int CHAINS, int N>
__global__ __launch_bounds__(512) void k32(float* o, unsigned long long* t, float a, float b) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
  float av = a + threadIdx.x, bv = b + threadIdx.x;
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][5];
  o[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
template <typename F> void run(const char* name, F f, int threads, int n_mfma_per_wave) {
  float* o; unsigned long long* t; hipMalloc(&o, 4 * 1024 * 1024); hipMalloc(&t, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f<<<256, threads>>>(o, t, 1.f, 2.f); hipDeviceSynchronize();
  hipEventRecord(e0); f<<<256, threads>>>(o, t, 1.f, 2.f); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-44s threads/block %4d: %8.1f memtime ticks per MFMA, kernel %.3f ms (%.2f ns per MFMA per wave)\n", name, threads,
         (double)h / n_mfma_per_wave, ms, ms * 1e6 / n_mfma_per_wave);
}
int main() {
  constexpr int N = 2000;
  run("16x16x4 f32, 1 dependent chain", k16<1, N>, 256, N * 16);
  run("16x16x4 f32, 2 interleaved chains", k16<2, N>, 256, N * 32);
  run("16x16x4 f32, 4 interleaved chains", k16<4, N>, 256, N * 64);
  run("16x16x4 f32, 1 chain, 2 waves/SIMD", k16<1, N>, 512, N * 16);
  run("16x16x4 f32, 2 chains, 2 waves/SIMD", k16<2, N>, 512, N * 32);
  run("32x32x2 f32, 1 dependent chain", k32<1, N>, 256, N * 16);
  run("32x32x2 f32, 2 interleaved chains", k32<2, N>, 256, N * 32);
  run("32x32x2 f32, 1 chain, 2 waves/SIMD", k32<1, N>, 512, N * 16);
  return 0;
}
