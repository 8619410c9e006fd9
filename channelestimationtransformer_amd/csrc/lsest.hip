// Least-squares channel estimation at the pilots with nearest-neighbour interpolation over the
// resource grid (what `LSChannelEstimator(rg, interpolation_type="nn")([y, no])` provides in
// front of the equalizer on the estimated-CSI branch, LocalSimulations/EndtoEndtest.py:130,184).
// For every stream k and resource element e the host has resolved the nearest pilot of that
// stream into a table entry: where its observation sits in a receive antenna's [symbol][fft]
// plane, the reciprocal pilot 1/p and 1/|p|^2. The kernel is a gather-scale:
//     h_hat[b, r, k, e] = y[b, r, src[k, e]] / p[k, e],   err_var[b, r, k, e] = no / |p[k, e]|^2.
// HBM-bound on its writes (12 B per output element; the y plane of a block stays in L2/L1).
#include "cet_internal.h"

namespace cet {
namespace {

// grid.y = (batch, antenna) plane, grid.x strides over the T = K * n_sym * n_eff table entries.
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

}  // namespace

int ls_nn_launch(const float* y, const int* src, const float* pinv, const float* pabs2inv, float no,
                 float* h_hat, float* err_var, long planes, int plane, int T, hipStream_t stream) {
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
