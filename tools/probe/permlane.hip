// Prints what v_permlane16_swap / v_permlane32_swap return for vdst = src = lane id.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned u = threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto t = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
  o[256 + threadIdx.x] = t[0]; o[320 + threadIdx.x] = t[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4); k<<<1, 64>>>(d); unsigned h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* n[6] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1", "p16(a,b) r0", "p16(a,b) r1"};
  for (int j = 0; j < 6; ++j) { printf("%s:", n[j]); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * j + i]); printf("\n"); }
  return 0;
}
