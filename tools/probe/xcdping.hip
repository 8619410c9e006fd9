// Round-trip time of the hidden-state exchange idiom (16-byte agent-scope store, acknowledgement,
// arrival on a counter, agent-scope poll, agent-scope 16-byte load) between two workgroups, by where
// the two run: the same XCD (workgroup ids 8 apart: ids go round-robin over the 8 XCDs) or two XCDs.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/xcdping.hip -o tools/probe/xcdping
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int AGENT = 16;  // sc1
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int scope_mode>
__global__ __launch_bounds__(64) void ping(float* buf, unsigned* cnt, int wa, int wb, int rounds, unsigned* xcc,
                                           unsigned long long* ticks, unsigned* sink) {
  const int wg = blockIdx.x, lane = threadIdx.x;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (lane == 0) xcc[wg] = id;
  if (wg != wa && wg != wb) return;
  const bool first = wg == wa;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 2 * 64 * 16, 0x00020000);
  unsigned* mine = cnt + (first ? 0 : 32);
  unsigned* theirs = cnt + (first ? 32 : 0);
  unsigned acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < rounds; ++i) {
    if (first || i > 0 || true) {
      if (!first) {  // wait for the ping
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(i + 1)) __builtin_amdgcn_s_sleep(1);
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 0, scope_mode);
        acc += v.x;
      }
      u32x4 o = {(unsigned)i, acc, 2u, 3u};
      __builtin_amdgcn_raw_buffer_store_b128(o, r, lane * 16, first ? 0 : 64 * 16, scope_mode);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (first) {  // wait for the pong
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(i + 1)) __builtin_amdgcn_s_sleep(1);
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 64 * 16, scope_mode);
        acc += v.x;
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0 && first) ticks[0] = t1 - t0;
  sink[wg * 64 + lane] = acc;
}


// The same exchange with the flag inside the data: every 8-byte pair is (value, step), aligned, so a
// pair is never seen half-written; no acknowledgement wait, no counter: the receiver re-reads the
// pieces until every pair carries the step it waits for.
__global__ __launch_bounds__(64) void ping_ll(float* buf, int wa, int wb, int rounds, unsigned long long* ticks, unsigned* sink) {
  const int wg = blockIdx.x, lane = threadIdx.x;
  if (wg != wa && wg != wb) return;
  const bool first = wg == wa;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4 * 64 * 16, 0x00020000);
  unsigned acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < rounds; ++i) {
    const unsigned flag = (unsigned)(i + 1);
    const unsigned par = (unsigned)(i & 1) * 2 * 64 * 16;
    auto recv = [&](unsigned off) {
      for (;;) {
        asm volatile("" ::: "memory");  // the read is repeated, not taken from the one before
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, off + par, AGENT);
        const bool ok = v.y == flag && v.w == flag;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
          if (v.x != (unsigned)i * 64u + (unsigned)lane || v.z != ~v.x) atomicAdd(sink + 4095, 1u);  // a pair seen half-written
          acc += v.x + v.z; break;
        }
      }
    };
    if (!first) recv(0);
    const unsigned val = (unsigned)i * 64u + (unsigned)lane;
    u32x4 o = {val, flag, ~val, flag};
    __builtin_amdgcn_raw_buffer_store_b128(o, r, lane * 16, (first ? 0 : 64 * 16) + par, AGENT);
    if (first) recv(64 * 16);
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0 && first) ticks[0] = t1 - t0;
  sink[wg * 64 + lane] = acc;
}

int main() {
  float* buf; unsigned *cnt, *xcc, *sink; unsigned long long* ticks;
  (void)hipMalloc(&buf, 4096); (void)hipMalloc(&cnt, 1024); (void)hipMalloc(&xcc, 64 * 4);
  (void)hipMalloc(&sink, 64 * 64 * 4); (void)hipMalloc(&ticks, 8);
  const int rounds = 2000;
  const int pairs[][2] = {{0, 8}, {0, 16}, {1, 9}, {0, 1}, {0, 4}, {0, 7}, {3, 12}};
  for (int mode : {16, 0}) for (auto& p : pairs) {
    (void)hipMemset(cnt, 0, 1024); (void)hipMemset(buf, 0, 4096);
    if (mode) ping<16><<<32, 64>>>(buf, cnt, p[0], p[1], rounds, xcc, ticks, sink);
    else ping<0><<<32, 64>>>(buf, cnt, p[0], p[1], rounds, xcc, ticks, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    unsigned hx[32]; unsigned long long t;
    (void)hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost); (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("data scope %s  workgroups %2d (XCC %u) <-> %2d (XCC %u): %.0f ns per round trip (two exchanges)\n",
           mode ? "agent(sc1)" : "none      ", p[0], hx[p[0]] & 15, p[1], hx[p[1]] & 15, t * 10.0 / rounds);
  }
  for (auto& p : pairs) {
    (void)hipMemset(buf, 0, 4096); (void)hipMemset(sink, 0, 64 * 64 * 4);
    ping_ll<<<32, 64>>>(buf, p[0], p[1], rounds, ticks, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    unsigned long long t; unsigned torn;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&torn, sink + 4095, 4, hipMemcpyDeviceToHost);
    printf("flag in data (agent-scope 16-byte pieces of two (value, step) pairs)  workgroups %2d <-> %2d: %.0f ns per round trip, %u wrong values\n",
           p[0], p[1], t * 10.0 / rounds, torn);
  }
  return 0;
}
