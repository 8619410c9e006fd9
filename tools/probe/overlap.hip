// Which instruction classes of two waves on one SIMD overlap with the fp32 matrix pipe?
// Block = 8 waves: waves 0-3 play role A, waves 4-7 role B (wave w sits on SIMD w % 4).
// Each role is timed alone and together with the other.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum Role { NONE = 0, MFMA, VALU, TRANS, LDS, VMEM, MIXED, VMEM_PURE, LDS_PURE, SALU, MFMA_LD, MFMA_LDS };

template <int N>
__device__ float role_mfma(float a, float b) {
  f32x4 acc{0, 0, 0, 0};
  for (int it = 0; it < N; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  return acc[0] + acc[1] + acc[2] + acc[3];
}
template <int N>
__device__ float role_valu(float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  for (int it = 0; it < N; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  return s;
}
template <int N>
__device__ float role_trans(float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  for (int it = 0; it < N; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  return s;
}
template <int N>
__device__ float role_lds(float* lds) {
  f32x4 s{0, 0, 0, 0};
  const f32x4* p = reinterpret_cast<const f32x4*>(lds) + threadIdx.x % 64 + 64 * (threadIdx.x / 64);
  for (int it = 0; it < N; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      f32x4 v;
      asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p));
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      s += v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return s[0] + s[1] + s[2] + s[3];
}
template <int N>
__device__ float role_vmem(const float* g) {
  f32x4 s{0, 0, 0, 0};
  const f32x4* p = reinterpret_cast<const f32x4*>(g) + threadIdx.x % 64;
  for (int it = 0; it < N; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      f32x4 v = __builtin_nontemporal_load(p + 64 * ((it * 16 + r) & 15));
      s += v;
    }
  return s[0] + s[1] + s[2] + s[3];
}
// loads only: no VALU instruction in the loop (address and data registers are fixed)
template <int N>
__device__ float role_vmem_pure(const float* g) {
  const f32x4* p = reinterpret_cast<const f32x4*>(g) + threadIdx.x % 64;
  f32x4 v0, v1, v2, v3;
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v0) : "v"(p));
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(v1) : "v"(p));
      asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(v2) : "v"(p));
      asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(v3) : "v"(p));
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"memory");
  return v0[0] + v1[0] + v2[0] + v3[0];
}
template <int N>
__device__ float role_lds_pure(float* lds) {
  const unsigned a = (unsigned)(size_t)(reinterpret_cast<const f32x4*>(lds) + threadIdx.x % 64 + 64 * (threadIdx.x / 64));
  f32x4 v0, v1;
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(a));
      asm volatile("ds_read_b128 %0, %1" : "=v"(v1) : "v"(a));
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
  return v0[0] + v1[0];
}
template <int N>
__device__ float role_salu(int seed) {
  int x = seed;
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("s_add_u32 %0, %0, 3\n s_xor_b32 %0, %0, 5\n s_add_u32 %0, %0, 7\n s_xor_b32 %0, %0, 9" : "+s"(x) : : "scc");
  }
  return (float)x;
}
// one wave: per MFMA, K independent VALU FMAs
template <int N, int K>
__device__ float role_mixed(float a, float b) {
  f32x4 acc{0, 0, 0, 0};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  for (int it = 0; it < N; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i & 7]) : "v"(a), "v"(b));
    }
  float s = acc[0] + acc[1] + acc[2] + acc[3];
  for (int i = 0; i < 8; ++i) s += x[i];
  return s;
}

// one wave: K MFMAs per global_load_dwordx4 (no other VALU work)
template <int N, int K>
__device__ float role_mfma_ld(float a, float b, const float* g) {
  f32x4 acc{0, 0, 0, 0};
  const f32x4* p = reinterpret_cast<const f32x4*>(g) + threadIdx.x % 64;
  f32x4 v0;
  for (int it = 0; it < N; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < K; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(8)" : "=v"(v0) : "v"(p));
    }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0)::"memory");
  return acc[0] + acc[1] + acc[2] + acc[3] + v0[0];
}
template <int N, int K>
__device__ float role_mfma_lds(float a, float b, float* lds) {
  f32x4 acc{0, 0, 0, 0};
  const unsigned ad = (unsigned)(size_t)(reinterpret_cast<const f32x4*>(lds) + threadIdx.x % 64 + 64 * (threadIdx.x / 64));
  f32x4 v0;
  for (int it = 0; it < N; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < K; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(4)" : "=v"(v0) : "v"(ad));
    }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0)::"memory");
  return acc[0] + acc[1] + acc[2] + acc[3] + v0[0];
}
template <int N, int K>
__device__ float play(int role, float a, float b, float* lds, const float* g) {
  switch (role) {
    case MFMA: return role_mfma<N>(a, b);
    case VALU: return role_valu<N>(a, b);
    case TRANS: return role_trans<N>(a, b);
    case LDS: return role_lds<N>(lds);
    case VMEM: return role_vmem<N>(g);
    case MIXED: return role_mixed<N, K>(a, b);
    case VMEM_PURE: return role_vmem_pure<N>(g);
    case LDS_PURE: return role_lds_pure<N>(lds);
    case SALU: return role_salu<N>(role);
    case MFMA_LD: return role_mfma_ld<N, (K > 0 ? K : 1)>(a, b, g);
    case MFMA_LDS: return role_mfma_lds<N, (K > 0 ? K : 1)>(a, b, lds);
    default: return 0;
  }
}

template <int N, int K>
__global__ __launch_bounds__(512) void kern(float* o, unsigned long long* t, const float* g, int ra, int rb,
                                            float a, float b) {
  __shared__ float lds[8 * 64 * 4];
  for (int i = threadIdx.x; i < 8 * 64 * 4; i += 512) lds[i] = i;
  __syncthreads();
  const int wave = threadIdx.x / 64;
  const int role = __builtin_amdgcn_readfirstlane(wave < 4 ? ra : rb);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  const float s = play<N, K>(role, a + threadIdx.x, b, lds, g);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x % 64 == 0) t[wave] = t1 - t0;
}

static const char* NAMES[] = {"none", "mfma", "valu-fma", "v_exp", "ds_read_b128", "global_load x4", "mfma+Kfma", "loads only", "ds_reads only", "salu x4", "K mfma + 1 load", "K mfma + 1 ds_read"};
template <int N, int K>
void run(int ra, int rb, float* o, unsigned long long* t, float* g) {
  kern<N, K><<<256, 512>>>(o, t, g, ra, rb, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<N, K><<<256, 512>>>(o, t, g, ra, rb, 1.f, 2.f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
  printf("A=%-16s B=%-16s K=%d  ticks per loop step: A %7.1f  B %7.1f   kernel %.3f ms\n", NAMES[ra], NAMES[rb], K,
         (double)h[0] / (N * 16), (double)h[4] / (N * 16), ms);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *o, *g;
  unsigned long long* t;
  hipMalloc(&o, 4 * 1024 * 1024);
  hipMalloc(&g, 1 << 20);
  hipMemset(g, 0, 1 << 20);
  hipMalloc(&t, 64);
  constexpr int N = 500;
  // loop step = 1 MFMA | 8 FMAs | 8 v_exp | 1 ds_read_b128 | 1 global_load_dwordx4
  const int roles[] = {MFMA, VALU, TRANS, LDS, VMEM};
  for (int r : roles) run<N, 0>(r, NONE, o, t, g);
  for (int r : roles) run<N, 0>(MFMA, r, o, t, g);
  const int pure[] = {VMEM_PURE, LDS_PURE, SALU};
  for (int r : pure) run<N, 0>(r, NONE, o, t, g);
  for (int r : pure) run<N, 0>(MFMA, r, o, t, g);
  for (int r : pure) run<N, 0>(r, MFMA, o, t, g);
  run<N, 0>(VALU, MFMA, o, t, g);
  run<N, 0>(VALU, VALU, o, t, g);
  run<N, 0>(VALU, LDS, o, t, g);
  run<N, 0>(VALU, VMEM, o, t, g);
  run<N, 1>(MFMA_LD, NONE, o, t, g);
  run<N, 2>(MFMA_LD, NONE, o, t, g);
  run<N, 4>(MFMA_LD, NONE, o, t, g);
  run<N, 4>(MFMA_LD, MFMA_LD, o, t, g);
  run<N, 1>(MFMA_LDS, NONE, o, t, g);
  run<N, 4>(MFMA_LDS, NONE, o, t, g);
  run<N, 1>(MIXED, NONE, o, t, g);
  run<N, 2>(MIXED, NONE, o, t, g);
  run<N, 4>(MIXED, NONE, o, t, g);
  run<N, 6>(MIXED, NONE, o, t, g);
  run<N, 8>(MIXED, NONE, o, t, g);
  run<N, 4>(MIXED, MIXED, o, t, g);
  return 0;
}
