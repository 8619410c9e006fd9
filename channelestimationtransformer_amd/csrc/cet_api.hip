// C ABI of libcet.so (declared in include/cet.h): argument validation, status codes and the
// handle type. The kernels and their launchers live in lmmse.hip, informer.hip and rnn.hip.
#include <cstdint>
#include <initializer_list>
#include <new>

#include "cet_internal.h"

namespace cet {

static thread_local int g_last_hip_error = 0;

int hip_fail(hipError_t e) {
  g_last_hip_error = (int)e;
  (void)hipGetLastError();  // clear the sticky error so later calls start clean
  return e == hipErrorOutOfMemory ? CET_ERR_ALLOC : CET_ERR_HIP;
}

// implemented in informer.hip / rnn.hip
size_t informer_blob_size(const cet_informer_cfg* cfg);
int informer_create(const cet_informer_cfg* cfg, const float* w, size_t n, InformerModel** out);
void informer_destroy(InformerModel* m);
size_t informer_attn_size(const InformerModel* m, int B);
size_t informer_taps_size(const InformerModel* m, int B);
void informer_select(InformerModel* m, int kernel, int pack);
int informer_forward(const InformerModel* m, const float* x_enc, const float* x_dec, float* out,
                     float* attn, float* taps, int B, hipStream_t stream);

size_t rnn_blob_size(int kind, int F, int I, int H, int layers);
int rnn_create(int kind, int F, int I, int H, int layers, const float* w, size_t n, RnnModel** out);
void rnn_destroy(RnnModel* m);
int rnn_rollout(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len,
                hipStream_t stream);
int rnn_rollout_static(const RnnModel* m, const float* x, float* out, int B, int seq_len, int pred_len,
                       hipStream_t stream);
void rnn_select(RnnModel* m, int variant);

// A handle's weights live on the device that was current when it was created; launching from
// another device would read them across GPUs (or fault without peer access).
static int check_device(const cet_handle_s* h) {
  int dev = -1;
  CET_HIP(hipGetDevice(&dev));
  return dev == h->device ? CET_OK : CET_ERR_INVALID_ARG;
}

// Arrays of interleaved complex64 are read and written as (re, im) pairs.
static bool pairs_aligned(std::initializer_list<const void*> ptrs) {
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 7) return false;
  return true;
}

}  // namespace cet

using namespace cet;

extern "C" {

const char* cet_strerror(int code) {
  switch (code) {
    case CET_OK: return "success";
    case CET_ERR_INVALID_ARG:
      return "invalid argument (null pointer, bad size or enum, or the handle belongs to another device)";
    case CET_ERR_UNSUPPORTED: return "configuration not supported by the gfx950 kernels";
    case CET_ERR_HIP: return "HIP runtime call failed";
    case CET_ERR_ALLOC: return "allocation failed";
    case CET_ERR_BLOB_SIZE: return "weight blob length does not match the configuration";
    case CET_ERR_TOO_LARGE:
      return "batch too large for the requested kernel or buffer (hidden-split rollout: ceil(B / 16) * "
             "hidden / 16 workgroups must not exceed the CU count, nor 256 on the static buffer)";
    default: return "unknown error code";
  }
}

int cet_last_hip_error(void) { return g_last_hip_error; }

int cet_abi_version(void) { return 1; }

int cet_destroy(cet_handle h) {
  if (!h) return CET_ERR_INVALID_ARG;
  if (h->kind == HANDLE_INFORMER) {
    informer_destroy(static_cast<InformerModel*>(h->impl));
  } else if (h->kind == HANDLE_RNN) {
    rnn_destroy(static_cast<RnnModel*>(h->impl));
  } else {
    return CET_ERR_INVALID_ARG;
  }
  h->kind = 0;
  delete h;
  return CET_OK;
}

size_t cet_informer_blob_size(const cet_informer_cfg* cfg) {
  return cfg ? informer_blob_size(cfg) : 0;
}

int cet_informer_create(const cet_informer_cfg* cfg, const float* weights, size_t n_weights,
                        cet_handle* out) {
  if (!cfg || !weights || !out) return CET_ERR_INVALID_ARG;
  InformerModel* m = nullptr;
  const int rc = informer_create(cfg, weights, n_weights, &m);
  if (rc != CET_OK) return rc;
  cet_handle h = new (std::nothrow) cet_handle_s;
  if (!h) {
    informer_destroy(m);
    return CET_ERR_ALLOC;
  }
  h->kind = HANDLE_INFORMER;
  h->impl = m;
  (void)hipGetDevice(&h->device);
  *out = h;
  return CET_OK;
}

size_t cet_informer_attn_size(cet_handle h, int B) {
  if (!h || h->kind != HANDLE_INFORMER || B < 0) return 0;
  return informer_attn_size(static_cast<InformerModel*>(h->impl), B);
}

size_t cet_informer_taps_size(cet_handle h, int B) {
  if (!h || h->kind != HANDLE_INFORMER || B < 0) return 0;
  return informer_taps_size(static_cast<InformerModel*>(h->impl), B);
}

int cet_informer_forward_taps(cet_handle h, const float* x_enc, const float* x_dec, float* out,
                              float* attn, float* taps, int B, void* stream) {
  if (!h || h->kind != HANDLE_INFORMER || B < 0) return CET_ERR_INVALID_ARG;
  if (B == 0) return CET_OK;
  if (!x_enc || !x_dec || !out) return CET_ERR_INVALID_ARG;
  if (const int rc = check_device(h)) return rc;
  return informer_forward(static_cast<InformerModel*>(h->impl), x_enc, x_dec, out, attn, taps, B,
                          static_cast<hipStream_t>(stream));
}

int cet_informer_select(cet_handle h, int kernel, int pack) {
  if (!h || h->kind != HANDLE_INFORMER || kernel < -1 || kernel > 2 || pack < -1 || pack > 1)
    return CET_ERR_INVALID_ARG;
  informer_select(static_cast<InformerModel*>(h->impl), kernel, pack);
  return CET_OK;
}

int cet_informer_forward(cet_handle h, const float* x_enc, const float* x_dec, float* out,
                         float* attn, int B, void* stream) {
  return cet_informer_forward_taps(h, x_enc, x_dec, out, attn, nullptr, B, stream);
}

size_t cet_rnn_blob_size(int kind, int features, int input_size, int hidden_size,
                         int num_layers) {
  return rnn_blob_size(kind, features, input_size, hidden_size, num_layers);
}

int cet_rnn_create(int kind, int features, int input_size, int hidden_size, int num_layers,
                   const float* weights, size_t n_weights, cet_handle* out) {
  if (!weights || !out) return CET_ERR_INVALID_ARG;
  RnnModel* m = nullptr;
  const int rc = rnn_create(kind, features, input_size, hidden_size, num_layers, weights,
                            n_weights, &m);
  if (rc != CET_OK) return rc;
  cet_handle h = new (std::nothrow) cet_handle_s;
  if (!h) {
    rnn_destroy(m);
    return CET_ERR_ALLOC;
  }
  h->kind = HANDLE_RNN;
  h->impl = m;
  (void)hipGetDevice(&h->device);
  *out = h;
  return CET_OK;
}

int cet_rnn_rollout(cet_handle h, const float* x, float* out, int B, int seq_len, int pred_len,
                    void* stream) {
  if (!h || h->kind != HANDLE_RNN || B < 0 || seq_len < 1 || pred_len < 1)
    return CET_ERR_INVALID_ARG;
  if (B == 0) return CET_OK;
  if (!x || !out) return CET_ERR_INVALID_ARG;
  if (const int rc = check_device(h)) return rc;
  return rnn_rollout(static_cast<RnnModel*>(h->impl), x, out, B, seq_len, pred_len,
                     static_cast<hipStream_t>(stream));
}

int cet_rnn_rollout_static(cet_handle h, const float* x, float* out, int B, int seq_len, int pred_len,
                           void* stream) {
  if (!h || h->kind != HANDLE_RNN || B < 0 || seq_len < 1 || pred_len < 1)
    return CET_ERR_INVALID_ARG;
  if (B == 0) return CET_OK;
  if (!x || !out) return CET_ERR_INVALID_ARG;
  if (const int rc = check_device(h)) return rc;
  return rnn_rollout_static(static_cast<RnnModel*>(h->impl), x, out, B, seq_len, pred_len,
                            static_cast<hipStream_t>(stream));
}

int cet_rnn_select(cet_handle h, int kernel) {
  if (!h || h->kind != HANDLE_RNN || kernel < 0 || kernel > 8) return CET_ERR_INVALID_ARG;
  rnn_select(static_cast<RnnModel*>(h->impl), kernel);
  return CET_OK;
}

// `x_hat` receives LLRs when `llr` is set (no_eff is then not written and may be null).
static int lmmse_checked(const float* y, const float* H, int no_mode, float no_scalar,
                         const float* no_dev, float* x_hat, float* no_eff, bool llr, long N,
                         int h_share, int Nr, int Ns, int variant, void* stream) {
  if (N < 0 || Nr <= 0 || Ns <= 0 || h_share < 1 || h_share > 65536) return CET_ERR_INVALID_ARG;
  if (no_mode < CET_NO_SCALAR || no_mode > CET_NO_PER_RE_RX) return CET_ERR_INVALID_ARG;
  if (N == 0) return CET_OK;
  if (!y || !H || !x_hat || (!llr && !no_eff)) return CET_ERR_INVALID_ARG;
  if (no_mode != CET_NO_SCALAR && !no_dev) return CET_ERR_INVALID_ARG;
  if (no_mode == CET_NO_SCALAR && !(no_scalar > 0.f)) return CET_ERR_INVALID_ARG;
  // the kernels move 16-byte pieces (LDS-DMA, vector loads and stores)
  for (const void* p : {(const void*)y, (const void*)H, (const void*)x_hat, (const void*)no_eff, (const void*)no_dev})
    if (reinterpret_cast<uintptr_t>(p) & 15) return CET_ERR_INVALID_ARG;
  if (Ns > Nr) return CET_ERR_UNSUPPORTED;
  return lmmse_launch(y, H, no_mode, no_scalar, no_dev, x_hat, llr ? nullptr : no_eff, N, Nr, Ns,
                      h_share, variant, static_cast<hipStream_t>(stream));
}

int cet_lmmse(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
              float* x_hat, float* no_eff, long N, int Nr, int Ns, void* stream) {
  return lmmse_checked(y, H, no_mode, no_scalar, no_dev, x_hat, no_eff, false, N, 1, Nr, Ns, 0,
                       stream);
}

int cet_lmmse_direct(const float* y, const float* H, int no_mode, float no_scalar,
                     const float* no_dev, float* x_hat, float* no_eff, long N, int Nr, int Ns,
                     void* stream) {
  return lmmse_checked(y, H, no_mode, no_scalar, no_dev, x_hat, no_eff, false, N, 1, Nr, Ns, 1,
                       stream);
}

int cet_lmmse_llr(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
                  float* llr, long N, int Nr, int Ns, void* stream) {
  return lmmse_checked(y, H, no_mode, no_scalar, no_dev, llr, nullptr, true, N, 1, Nr, Ns, 0,
                       stream);
}

int cet_lmmse_block(const float* y, const float* H, int no_mode, float no_scalar,
                    const float* no_dev, float* x_hat, float* no_eff, long N, int h_share, int Nr,
                    int Ns, void* stream) {
  return lmmse_checked(y, H, no_mode, no_scalar, no_dev, x_hat, no_eff, false, N, h_share, Nr, Ns,
                       0, stream);
}

int cet_lmmse_block_llr(const float* y, const float* H, int no_mode, float no_scalar,
                        const float* no_dev, float* llr, long N, int h_share, int Nr, int Ns,
                        void* stream) {
  return lmmse_checked(y, H, no_mode, no_scalar, no_dev, llr, nullptr, true, N, h_share, Nr, Ns, 0,
                       stream);
}

int cet_lmmse_grid(const float* y, const float* h_hat, const float* err_var, int no_mode,
                   float no_scalar, const float* no_dev, const int32_t* data_sym,
                   const int32_t* eff_sc, float* x_hat, float* no_eff, int B, int Nr, int Ns,
                   int n_sym, int fft, int n_sc, int n_ds, void* stream) {
  if (B < 0 || Nr <= 0 || Ns <= 0 || n_sym < 1 || fft < 1 || n_sc < 0 || n_sc > fft || n_ds < 0 ||
      n_ds > n_sym)
    return CET_ERR_INVALID_ARG;
  if (no_mode < 0 || no_mode > 2) return CET_ERR_INVALID_ARG;
  if ((long)B * n_sc * n_ds == 0) return CET_OK;
  if (!y || !h_hat || !data_sym || !eff_sc || !x_hat || !no_eff) return CET_ERR_INVALID_ARG;
  if (no_mode != 0 && !no_dev) return CET_ERR_INVALID_ARG;
  if (no_mode == 0 && !(no_scalar > 0.f) && !err_var) return CET_ERR_INVALID_ARG;
  if (!pairs_aligned({y, h_hat, x_hat})) return CET_ERR_INVALID_ARG;
  if (Ns > Nr) return CET_ERR_UNSUPPORTED;
  return lmmse_grid_launch(y, h_hat, err_var, no_mode, no_scalar, no_dev, data_sym, eff_sc, x_hat,
                           no_eff, B, Nr, Ns, n_sym, fft, n_sc, n_ds,
                           static_cast<hipStream_t>(stream));
}

int cet_ls_nn(const float* y, const int32_t* src, const float* pinv, const float* pabs2inv,
              float no, float* h_hat, float* err_var, long planes, int plane, int T, void* stream) {
  if (planes < 0 || plane < 1 || T < 1 || !(no >= 0.f)) return CET_ERR_INVALID_ARG;
  if (planes == 0) return CET_OK;
  if (!y || !src || !pinv || !pabs2inv || !h_hat || !err_var) return CET_ERR_INVALID_ARG;
  if (!pairs_aligned({y, pinv, h_hat})) return CET_ERR_INVALID_ARG;
  return ls_nn_launch(y, src, pinv, pabs2inv, no, h_hat, err_var, planes, plane, T,
                      static_cast<hipStream_t>(stream));
}

int cet_nmse_sums(const float* pred, const float* truth, double* sums, long B, int P, int C,
                  void* stream) {
  if (!pred || !truth || !sums || B < 0 || P < 1 || C < 1 || !pairs_aligned({sums}))
    return CET_ERR_INVALID_ARG;
  return nmse_sums_launch(pred, truth, sums, B, P, C, static_cast<hipStream_t>(stream));
}

int cet_qpsk_bit_errors(const float* x_hat, const float* x_true, double* out, long N, void* stream) {
  if (!x_hat || !x_true || !out || N < 0 || !pairs_aligned({x_hat, x_true, out}))
    return CET_ERR_INVALID_ARG;
  return qpsk_errors_launch(x_hat, x_true, out, N, static_cast<hipStream_t>(stream));
}

}  // extern "C"
