// Least-squares channel estimation at the pilots with nearest-neighbour interpolation over the
// resource grid (what `LSChannelEstimator(rg, interpolation_type="nn")([y, no])` provides in
// front of the equalizer on the estimated-CSI branch, LocalSimulations/EndtoEndtest.py:130,184).
// For every stream k and resource element e the host has resolved the nearest pilot of that
// stream into a table entry: where its observation sits in a receive antenna's [symbol][fft]
// plane, the reciprocal pilot 1/p and 1/|p|^2. The kernel is a gather-scale:
//     h_hat[b, r, k, e] = y[b, r, src[k, e]] / p[k, e],   err_var[b, r, k, e] = no / |p[k, e]|^2.
// HBM-bound on its writes (12 B per output element; the y plane of a block stays in L2/L1).
#include <cstdint>

#include "cet_internal.h"

namespace cet {
namespace {

// grid.y = (batch, antenna) plane, grid.x strides over the T = K * n_sym * n_eff table entries.
// General form: any T, any alignment.
__global__ __launch_bounds__(256) void ls_nn_kernel(const float2* __restrict__ y,
                                                    const int* __restrict__ src,
                                                    const float2* __restrict__ pinv,
                                                    const float* __restrict__ pabs2inv, float no,
                                                    float2* __restrict__ h_hat,
                                                    float* __restrict__ err_var, int plane, int T) {
  const float2* yp = y + (size_t)blockIdx.y * plane;
  float2* hp = h_hat + (size_t)blockIdx.y * T;
  float* ep = err_var + (size_t)blockIdx.y * T;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) {
    const float2 v = yp[src[i]];
    const float2 q = pinv[i];
    hp[i] = make_float2(v.x * q.x - v.y * q.y, v.x * q.y + v.y * q.x);
    ep[i] = no * pabs2inv[i];
  }
}

// Even T and 16-byte aligned outputs: a thread takes two neighbouring table entries, which it
// reads once, for PL consecutive planes; nontemporal 16-byte stores of h_hat and 8-byte stores of
// err_var. Stores per launch of 8192 slots (705 MB): one element and one plane per thread with 8-
// and 4-byte stores 0.27 ms, this shape with cached stores 0.18 ms, nontemporal 0.12 ms.
// Threads are numbered over (plane group, entry pair) so that no block is partly empty.
constexpr int LS_PL = 4;
__global__ __launch_bounds__(256) void ls_nn_pairs(const float2* __restrict__ y,
                                                   const int2* __restrict__ src,
                                                   const float4* __restrict__ pinv,
                                                   const float2* __restrict__ pabs2inv, float no,
                                                   float4* __restrict__ h_hat,
                                                   float2* __restrict__ err_var, long planes, int plane,
                                                   int pairs) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long group = idx / pairs;
  const int j = (int)(idx - group * pairs);
  const long p0 = group * LS_PL;
  if (p0 >= planes) return;
  const int2 s = src[j];
  const float4 q = pinv[j];
  const float2 a = pabs2inv[j];
  const float2 ev = make_float2(no * a.x, no * a.y);
  const int np = planes - p0 < LS_PL ? (int)(planes - p0) : LS_PL;
  float2 v0[LS_PL], v1[LS_PL];
#pragma unroll
  for (int k = 0; k < LS_PL; ++k) {
    if (k < np) {
      const float2* yp = y + (size_t)(p0 + k) * plane;
      v0[k] = yp[s.x];
      v1[k] = yp[s.y];
    }
  }
#pragma unroll
  for (int k = 0; k < LS_PL; ++k) {
    if (k < np) {
      const size_t at = (size_t)(p0 + k) * pairs + j;
      typedef float v4f __attribute__((ext_vector_type(4)));
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v4f hv{v0[k].x * q.x - v0[k].y * q.y, v0[k].x * q.y + v0[k].y * q.x,
                   v1[k].x * q.z - v1[k].y * q.w, v1[k].x * q.w + v1[k].y * q.z};
      __builtin_nontemporal_store(hv, reinterpret_cast<v4f*>(h_hat + at));
      __builtin_nontemporal_store(v2f{ev.x, ev.y}, reinterpret_cast<v2f*>(err_var + at));
    }
  }
}

}  // namespace

int ls_nn_launch(const float* y, const int* src, const float* pinv, const float* pabs2inv, float no,
                 float* h_hat, float* err_var, long planes, int plane, int T, hipStream_t stream) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(h_hat) | reinterpret_cast<uintptr_t>(pinv)) & 15) == 0 &&
                       ((reinterpret_cast<uintptr_t>(err_var) | reinterpret_cast<uintptr_t>(pabs2inv) |
                         reinterpret_cast<uintptr_t>(src)) & 7) == 0;
  if (T % 2 == 0 && aligned && planes > 0) {
    const int pairs = T / 2;
    const long threads = (planes + LS_PL - 1) / LS_PL * pairs;
    const long blocks = (threads + 255) / 256;
    if (blocks <= 0x7fffffffl) {
      hipLaunchKernelGGL(ls_nn_pairs, dim3((unsigned)blocks), dim3(256), 0, stream,
                         reinterpret_cast<const float2*>(y), reinterpret_cast<const int2*>(src),
                         reinterpret_cast<const float4*>(pinv), reinterpret_cast<const float2*>(pabs2inv),
                         no, reinterpret_cast<float4*>(h_hat), reinterpret_cast<float2*>(err_var), planes,
                         plane, pairs);
      CET_HIP(hipGetLastError());
      return CET_OK;
    }
  }
  int bx = (T + 255) / 256;
  if (bx > 64) bx = 64;
  for (long p0 = 0; p0 < planes; p0 += 65535) {  // grid.y limit
    const long n = planes - p0 < 65535 ? planes - p0 : 65535;
    hipLaunchKernelGGL(ls_nn_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(y) + (size_t)p0 * plane, src,
                       reinterpret_cast<const float2*>(pinv), pabs2inv, no,
                       reinterpret_cast<float2*>(h_hat) + (size_t)p0 * T, err_var + (size_t)p0 * T, plane,
                       T);
    CET_HIP(hipGetLastError());
  }
  return CET_OK;
}

}  // namespace cet
