// Internal declarations shared by the translation units of libcet.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/cet.h"

namespace cet {

// Records a failing hipError_t for cet_last_hip_error() and maps it to CET_ERR_HIP.
int hip_fail(hipError_t e);

#define CET_HIP(call)                                 \
  do {                                                \
    hipError_t _e = (call);                           \
    if (_e != hipSuccess) return ::cet::hip_fail(_e); \
  } while (0)

enum HandleKind { HANDLE_INFORMER = 0x1F0, HANDLE_RNN = 0x2F0 };

struct InformerModel;
struct RnnModel;

// launchers implemented next to their kernels
int lmmse_launch(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
                 float* x_hat, float* no_eff, long N, int Nr, int Ns, int hshare, int variant,
                 hipStream_t stream);

int lmmse_grid_launch(const float* y, const float* h_hat, const float* err_var, int no_mode,
                      float no_scalar, const float* no_dev, const int* data_sym, const int* eff_sc,
                      float* x_hat, float* no_eff, int B, int Nr, int Ns, int n_sym, int fft, int n_sc,
                      int n_ds, hipStream_t stream);

int ls_nn_launch(const float* y, const int* src, const float* pinv, const float* pabs2inv, float no,
                 float* h_hat, float* err_var, long planes, int plane, int T, hipStream_t stream);
int nmse_sums_launch(const float* pred, const float* truth, double* sums, long B, int P, int C,
                     hipStream_t stream);
int qpsk_errors_launch(const float* x_hat, const float* x_true, double* out, long N,
                       hipStream_t stream);

}  // namespace cet

struct cet_handle_s {
  int kind;
  int device;
  void* impl;
};
