// Evaluation sums on the GPU: per-slot NMSE partial sums and hard-decision QPSK bit errors.
// They are what a rank contributes to the one all-reduce of an evaluation
// (HPC/python/Evaluate.py:172-177 forms NMSE_s = sum_err_s / sum_pow_s over batch and antennas;
// LocalSimulations/EndtoEndtest.py:196-200 counts bit errors). HBM-bound single pass: every
// input element is read once, reduced in fp64 inside a wave, then across the four waves of a
// workgroup through LDS, and added with ONE atomic per workgroup and sum: atomics on one address
// serialise in L2 (8192 of them, one per wave, used to cost 100 us of a 110 us kernel), so the
// grids are kept to a few hundred workgroups that stride over the data.
#include <cstdint>

#include "cet_internal.h"

namespace cet {
namespace {

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Sum over the 256 threads of a workgroup; the result is valid in thread 0. `slot`: 4 doubles of
// LDS per quantity reduced.
__device__ inline double block_sum(double v, double* slot) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  return slot[0] + slot[1] + slot[2] + slot[3];
}

// pred, truth [B][P][C] fp32 (C % 4 == 0). blockIdx.y = slot p. sums[0][p] += sum (t-p)^2,
// sums[1][p] += sum t^2.
__global__ __launch_bounds__(256) void nmse_sums_kernel(const float* __restrict__ pred,
                                                        const float* __restrict__ truth,
                                                        double* __restrict__ sums, long B, int P,
                                                        int C) {
  const int p = blockIdx.y;
  const int c4 = C / 4;
  const long n = B * c4;  // float4 items of this slot
  double err = 0.0, pow = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / c4;
    const int q = (int)(i - b * c4);
    const size_t at = ((size_t)b * P + p) * C + 4 * q;
    const float4 a = *reinterpret_cast<const float4*>(pred + at);
    const float4 t = *reinterpret_cast<const float4*>(truth + at);
    const double d0 = (double)t.x - a.x, d1 = (double)t.y - a.y, d2 = (double)t.z - a.z,
                 d3 = (double)t.w - a.w;
    err += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    pow += (double)t.x * t.x + (double)t.y * t.y + (double)t.z * t.z + (double)t.w * t.w;
  }
  __shared__ double red[8];
  err = block_sum(err, red);
  pow = block_sum(pow, red + 4);
  if (threadIdx.x == 0) {
    atomicAdd(sums + p, err);
    atomicAdd(sums + P + p, pow);
  }
}

// The same sums one float per thread: any C, any alignment (the 16-byte form above needs C % 4 == 0
// and 16-byte aligned arrays). Same accumulation (fp64 per thread, wave, workgroup, one atomic).
__global__ __launch_bounds__(256) void nmse_sums_kernel_any(const float* __restrict__ pred,
                                                            const float* __restrict__ truth,
                                                            double* __restrict__ sums, long B, int P,
                                                            int C) {
  const int p = blockIdx.y;
  const long n = B * C;  // floats of this slot
  double err = 0.0, pow = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / C;
    const size_t at = ((size_t)b * P + p) * C + (size_t)(i - b * C);
    const double t = (double)truth[at], d = t - (double)pred[at];
    err += d * d;
    pow += t * t;
  }
  __shared__ double red[8];
  err = block_sum(err, red);
  pow = block_sum(pow, red + 4);
  if (threadIdx.x == 0) {
    atomicAdd(sums + p, err);
    atomicAdd(sums + P + p, pow);
  }
}

// x_hat, x_true [N] complex64 (interleaved). out[0] += bit errors of the hard decisions
// (real > 0, imag > 0), out[1] += 2 N.
__global__ __launch_bounds__(256) void qpsk_errors_kernel(const float2* __restrict__ x_hat,
                                                          const float2* __restrict__ x_true,
                                                          double* __restrict__ out, long N,
                                                          long n_vec) {
  // symbols [0, n_vec) two per 16-byte load (n_vec is even, and 0 unless both arrays are 16-byte
  // aligned), the rest one by one
  unsigned long long cnt = 0;
  const float4* a4 = reinterpret_cast<const float4*>(x_hat);
  const float4* t4 = reinterpret_cast<const float4*>(x_true);
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long)gridDim.x * blockDim.x;
  for (long i = tid; i < n_vec / 2; i += nthr) {
    const float4 a = a4[i], t = t4[i];
    cnt += ((a.x > 0.f) != (t.x > 0.f)) + ((a.y > 0.f) != (t.y > 0.f)) +
           ((a.z > 0.f) != (t.z > 0.f)) + ((a.w > 0.f) != (t.w > 0.f));
  }
  for (long i = n_vec + tid; i < N; i += nthr) {
    const float2 a = x_hat[i], t = x_true[i];
    cnt += ((a.x > 0.f) != (t.x > 0.f)) + ((a.y > 0.f) != (t.y > 0.f));
  }
  __shared__ double red[4];
  const double s = block_sum((double)cnt, red);
  if (threadIdx.x == 0) {
    atomicAdd(out, s);
    if (blockIdx.x == 0) atomicAdd(out + 1, 2.0 * (double)N);
  }
}

}  // namespace

int nmse_sums_launch(const float* pred, const float* truth, double* sums, long B, int P, int C,
                     hipStream_t stream) {
  CET_HIP(hipMemsetAsync(sums, 0, 2 * (size_t)P * sizeof(double), stream));
  if (B == 0) return CET_OK;
  const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(truth)) & 15) == 0;
  long blocks = (B * (vec ? C / 4 : C) + 255) / 256;
  if (blocks > 128) blocks = 128;  // x P slots
  if (vec)
    hipLaunchKernelGGL(nmse_sums_kernel, dim3((unsigned)blocks, (unsigned)P), dim3(256), 0, stream, pred,
                       truth, sums, B, P, C);
  else
    hipLaunchKernelGGL(nmse_sums_kernel_any, dim3((unsigned)blocks, (unsigned)P), dim3(256), 0, stream, pred,
                       truth, sums, B, P, C);
  CET_HIP(hipGetLastError());
  return CET_OK;
}

int qpsk_errors_launch(const float* x_hat, const float* x_true, double* out, long N,
                       hipStream_t stream) {
  CET_HIP(hipMemsetAsync(out, 0, 2 * sizeof(double), stream));
  if (N == 0) return CET_OK;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x_hat) | reinterpret_cast<uintptr_t>(x_true)) & 15) == 0;
  const long n_vec = aligned ? N / 2 * 2 : 0;
  long blocks = (N / 2 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(qpsk_errors_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     reinterpret_cast<const float2*>(x_hat), reinterpret_cast<const float2*>(x_true),
                     out, N, n_vec);
  CET_HIP(hipGetLastError());
  return CET_OK;
}

}  // namespace cet
