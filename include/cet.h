/*
 * cet.h -- C ABI of libcet.so: MI355X (gfx950) channel-predictor + LMMSE inference path.
 *
 * Every entry point takes plain pointers and sizes. Device buffers are owned by the
 * caller; the library only owns the weights held inside a handle. Kernels are enqueued
 * on the HIP stream passed in (`stream` is a hipStream_t cast to void*, NULL = default
 * stream) and the calls do not synchronise. Return value: 0 on success, a CET_ERR_* code
 * otherwise (never an exception across this boundary).
 *
 * The reference has no FFI of its own (it is a pure-Python PyTorch/Sionna program), so each
 * entry point below names the Python call it stands in for.
 */
#ifndef CET_H
#define CET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  CET_OK = 0,
  CET_ERR_INVALID_ARG = 1, /* null pointer, negative size, bad enum value, or a handle used
                              while another HIP device than the one it was created on is current */
  CET_ERR_UNSUPPORTED = 2, /* well-formed request outside what the kernels are built for */
  CET_ERR_HIP = 3,         /* a HIP runtime call failed; see cet_last_hip_error()         */
  CET_ERR_ALLOC = 4,       /* host or device allocation failed                           */
  CET_ERR_BLOB_SIZE = 5,   /* weight blob length does not match the configuration        */
  CET_ERR_TOO_LARGE = 6    /* the batch exceeds what the requested kernel or buffer serves */
};

typedef struct cet_handle_s* cet_handle;

/* Library / device ------------------------------------------------------------------ */

/* Text for a CET_ERR_* code (static storage). */
const char* cet_strerror(int code);
/* hipError_t value of the most recent failing HIP call on this thread (0 if none). */
int cet_last_hip_error(void);
/* ABI version, bumped when a signature changes. */
int cet_abi_version(void);
/* Releases the device weights and the handle (models and recurrent nets alike). */
int cet_destroy(cet_handle h);

/* Transformer predictor -------------------------------------------------------------- */

/* Constructor arguments of Informer / InformerStack / InformerStack_e2e
 * (LocalSimulations/models/model.py:12-16, 85-89, 161-165). `attn` is always 'full',
 * `embed` always 'fixed'; dropout is inactive in eval mode and has no field. */
typedef struct cet_informer_cfg {
  int32_t enc_in, dec_in, c_out;
  int32_t seq_len, label_len, pred_len;
  int32_t d_model, n_heads, e_layers, d_layers, d_ff;
  int32_t stack;      /* 0: Informer (one encoder). 1: InformerStack / InformerStack_e2e */
  int32_t distil;     /* 1: ConvLayer between encoder layers (encoder.py:5-23)           */
  int32_t activation; /* 0 relu, 1 gelu (exact erf form)                                 */
} cet_informer_cfg;

/* Number of fp32 values cet_informer_create expects for `cfg` (0 if cfg is unsupported).
 * Supported: 1 <= n_heads <= 12 with d_model / n_heads <= 8, d_model <= 64, 1 <= d_ff <= 64,
 * enc_in/dec_in in {16, 32}, seq_len <= 46 (48 without distilling), label_len + pred_len <= 48,
 * at most 32 rows of concatenated encoder outputs. The blob always has the model's own dimensions;
 * models narrower than 64 features or with fewer than 8 features per head are widened with zeros
 * inside cet_informer_create. Models of up to 8 heads, 30 slots of history and 32 decoder rows
 * run on both launch shapes of the kernel, larger ones on the one-wave shape only.
 * Below, d_inner = (d_model / n_heads) * n_heads (models/attn.py:126-133; 48 for d_model 54).
 *
 * Blob order (all tensors in their PyTorch layout, row-major, concatenated):
 *   enc_embedding : tokenConv.weight[d_model,enc_in,3], tokenConv.bias[d_model],
 *                   pe[0:seq_len, d_model]
 *   dec_embedding : tokenConv.weight[d_model,dec_in,3], tokenConv.bias[d_model],
 *                   pe[0:label_len+pred_len, d_model]
 *   for each encoder of the stack (stack=1: e_layers, e_layers-1, ... 3 layers; stack=0: one
 *   encoder of e_layers layers):
 *       for each attention layer: query/key/value projection weight[d_inner,d_model] +
 *           bias[d_inner], out projection weight[d_model,d_inner] + bias[d_model],
 *           conv1.weight[d_ff,d_model], conv1.bias[d_ff],
 *           conv2.weight[d_model,d_ff], conv2.bias[d_model], norm1.weight, norm1.bias,
 *           norm2.weight, norm2.bias
 *       then, if distil, for each of the (layers-1) ConvLayers: downConv.weight[d_model,
 *           d_model,3], downConv.bias, norm.weight, norm.bias, norm.running_mean,
 *           norm.running_var
 *       then norm.weight, norm.bias
 *   for each decoder layer: self_attention q/k/v/out (weight,bias), cross_attention
 *       q/k/v/out (weight,bias), conv1 (w,b), conv2 (w,b), norm1, norm2, norm3 (w,b each)
 *   decoder.norm.weight, decoder.norm.bias, projection.weight[c_out,d_model],
 *   projection.bias[c_out]
 */
size_t cet_informer_blob_size(const cet_informer_cfg* cfg);

/* Stands in for `InformerStack(...)` + `load_state_dict` + `.eval()` + `.to(device)`
 * (LocalSimulations/Predict.py:44-64, 90-94). `weights` is a HOST pointer; the packed copy
 * lives on the current HIP device. */
int cet_informer_create(const cet_informer_cfg* cfg, const float* weights, size_t n_weights,
                        cet_handle* out);

/* Number of fp32 values of the attention-map buffer for batch B: the encoder self-attention
 * maps [B, n_heads, L, L] of every encoder layer, concatenated in execution order. */
size_t cet_informer_attn_size(cet_handle h, int B);

/* Stands in for `informer(enc_inp, dec_inp)` (LocalSimulations/models/model.py:220-234,
 * call site Predict.py:114). Device pointers:
 *   x_enc [B, seq_len, enc_in], x_dec [B, label_len+pred_len, dec_in],
 *   out   [B, pred_len, c_out],
 *   attn  NULL, or cet_informer_attn_size(h,B) floats (what output_attention=True returns). */
int cet_informer_forward(cet_handle h, const float* x_enc, const float* x_dec, float* out,
                         float* attn, int B, void* stream);

/* Pins the launch shape of the Transformer kernel for later forwards of this handle (tests and
 * timing; the default picks by batch size). kernel: 0 by batch size, 1 one wave per sequence,
 * 2 workgroups of 8 sequences with the weights staged in LDS; pack: 1 / 0 two sequences per tile
 * in the short encoder stages of the workgroup kernel, or not; -1 leaves a setting as it is.
 * Not to be called while a forward of the same handle runs on another thread. No counterpart in
 * the reference (PyTorch picks its kernels internally). */
int cet_informer_select(cet_handle h, int kernel, int pack);

/* Debug variant: additionally copies the activations after every stage into `taps`
 * (device, cet_informer_taps_size(h,B) floats; layout [stage][B][rows_of_stage][64]; features
 * d_model..63 of a narrower model are zero). */
size_t cet_informer_taps_size(cet_handle h, int B);
int cet_informer_forward_taps(cet_handle h, const float* x_enc, const float* x_dec, float* out,
                              float* attn, float* taps, int B, void* stream);

/* Recurrent predictors ---------------------------------------------------------------- */

enum { CET_RNN_TANH = 0, CET_RNN_GRU = 1, CET_RNN_LSTM = 2 };

/* fp32 count expected by cet_rnn_create. Blob order (PyTorch layouts):
 *   model.encoder.0.weight[input,features], model.encoder.0.bias[input],
 *   for l in 0..layers-1: weight_ih_l[G*hidden, in_l], weight_hh_l[G*hidden, hidden],
 *                         bias_ih_l[G*hidden], bias_hh_l[G*hidden]
 *                         (G = 1 / 3 / 4 for RNN / GRU / LSTM; in_0 = input, else hidden)
 *   model.decoder.0.weight[features,hidden], model.decoder.0.bias[features] */
size_t cet_rnn_blob_size(int kind, int features, int input_size, int hidden_size,
                         int num_layers);

/* Stands in for `GRU/LSTM/RNN(features, input_size, hidden_size, num_layers)` +
 * `load_state_dict` + `.eval()` (LocalSimulations/models/model.py:276, 350, 426;
 * HPC/python/Evaluate.py:60-98). `weights` is a HOST pointer. Supported: features a multiple of
 * 4 up to 32, hidden_size 64, 128, 256 or 512, 1 to 4 layers (3 at 512), input_size 1 to 4096
 * (the encoder Linear is folded into the first layer's input weights at create time, in double
 * precision). Hidden 128 / 256 with up to 3 layers are served by every kernel of cet_rnn_select;
 * the other shapes by the half-tile kernels alone (choices 0 and 4 .. 8; 1 .. 3 and
 * cet_rnn_rollout_static return CET_ERR_UNSUPPORTED). A model with another hidden size runs as the
 * next of these sizes with zero units added (zero weights and biases on every side: the added
 * units stay at zero and no result changes); the Python classes lay the blob out that way. */
int cet_rnn_create(int kind, int features, int input_size, int hidden_size, int num_layers,
                   const float* weights, size_t n_weights, cet_handle* out);

/* Stands in for `model.test_data(x, pred_len, device)` (LocalSimulations/models/model.py:
 * 298-312, 373-388, 449-465): zero initial state, seq_len teacher-forced steps on x, then
 * pred_len-1 closed-loop steps on the model's own output. Device pointers:
 *   x [B, seq_len, features], out [B, pred_len, features]. */
int cet_rnn_rollout(cet_handle h, const float* x, float* out, int B, int seq_len,
                    int pred_len, void* stream);

/* Pins the kernel of later rollouts of this handle (tests and timing; 0..8): 0 by batch size, 1 the
 * batch-tiled kernel (a workgroup takes 32 sequences and all hidden units), 2 the hidden-split
 * kernel for small batches (16 sequences and 4, 8 or 16 hidden units per workgroup - as few as
 * keeps the launch's workgroups resident at once, so that a small batch spreads over the whole
 * device -, hidden states exchanged through a buffer of the library once per layer and step; one
 * launch takes as many sequences as ceil(B / 16) * hidden / 16 workgroups fit the device's CU
 * count, twice that where two such workgroups fit a CU; larger batches go through it in chunks,
 * launch after launch - which choice 0 does for as long as it beats the batch-tiled kernel, a
 * few chunks), 3 the same with an exchange that gives up at its first
 * unanswered poll (tests: every chunk is then computed by the stand-by launch described next), 4
 * the half-tile kernel (the batch-tiled design with 16 sequences per workgroup: 0.55 of its time
 * up to 16 sequences per CU), 5 / 6 the same with a tile's hidden units shared by 2 / 4 workgroups
 * that exchange their states like the hidden-split kernel's (up to 8 / 4 sequences per CU; 6 needs
 * 256 hidden units; CET_ERR_TOO_LARGE beyond), 7 as 5 with an exchange that gives up like 3, 8 as 5
 * on 8 workgroups (512 hidden units; up to 2 sequences per CU). Choice 0 takes the kernel with the
 * least estimated time for the batch, and a batch beyond 16 sequences per CU in parts (whole rounds
 * of full or half tiles, then the rest: several launches on the stream, disjoint ranges of x and
 * out). Behind a launch of 5 .. 8 the stand-by launch is choice 4's.
 * Rollouts on the kernels that exchange (2, 3, 5, 6, 7) may overlap freely, on any streams and handles and from any
 * host threads: each call claims one of four exchange buffers per GPU (a call that finds all four in flight on other
 * streams queues its stream behind one of them), and behind every hidden-split launch the
 * batch-tiled kernel is launched on the same stream on stand-by; it returns at once unless the
 * exchange gave up (other work kept part of the grid from becoming resident for about a tenth of a second),
 * in which case it computes that chunk. Under stream capture choice 0 uses a kernel that needs no
 * exchange and the exchanging choices return CET_ERR_UNSUPPORTED (cet_rnn_rollout_static serves captured small
 * batches). Not to be called while a rollout of the same handle runs on another thread. No
 * counterpart in the reference. */
int cet_rnn_select(cet_handle h, int kernel);

/* cet_rnn_rollout for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc): always the hidden-split kernel, with its
 * exchange buffer in a static device array of the library (1.6 MiB per GPU) instead of the buffer
 * cet_rnn_rollout allocates on its first small batch. Calls hipMalloc / hipFree on no path.
 * Limit: ceil(B / 16) * hidden / 16 <= 256 (B <= 256 at hidden 256, B <= 512 at hidden 128) and
 * the launch's workgroups resident at once (CU count, twice that where two fit a CU), else
 * CET_ERR_TOO_LARGE. All calls on one GPU share the array: calls that overlap in
 * time on one GPU (other streams, other handles) race (cet_rnn_rollout has no such limit). The
 * stand-by launch of the batch-tiled kernel follows here too. No counterpart in the reference. */
int cet_rnn_rollout_static(cet_handle h, const float* x, float* out, int B, int seq_len,
                           int pred_len, void* stream);

/* LMMSE MIMO equalizer ---------------------------------------------------------------- */

enum {
  CET_NO_SCALAR = 0,    /* one noise variance for all resource elements (`no_scalar`)      */
  CET_NO_PER_RE = 1,    /* no_dev[N]     : noise variance per resource element             */
  CET_NO_PER_RE_RX = 2  /* no_dev[N,Nr]  : diagonal noise+interference covariance per RE   */
};

/* Stands in for the per-resource-element arithmetic of Sionna's
 * `LMMSEEqualizer(rg, sm)([y, h_hat, err_var, no])` (call sites
 * LocalSimulations/EndtoEndtest.py:133,191; HPC/python/EvaluateE2E.py:254,317):
 *   G = (H^H S^-1 H + I)^-1 H^H S^-1,  x_hat = diag(GH)^-1 G y,  no_eff = Re(1/diag(GH) - 1)
 * with S diagonal. Device pointers, interleaved (re,im) float32:
 *   y [N,Nr,2], H [N,Nr,Ns,2]  ->  x_hat [N,Ns,2], no_eff [N,Ns].
 * Supported (Nr,Ns): Nr in {2,4,8,16}, Ns in {1,2,4}, Ns <= Nr. Every device pointer of
 * cet_lmmse, cet_lmmse_direct, cet_lmmse_llr, cet_lmmse_block and cet_lmmse_block_llr must be
 * 16-byte aligned (CET_ERR_INVALID_ARG otherwise; the kernels move 16-byte pieces); the kernels
 * write exactly the N * Ns elements of each output and nothing behind them. */
int cet_lmmse(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
              float* x_hat, float* no_eff, long N, int Nr, int Ns, void* stream);

/* Same contract as cet_lmmse, served by the plain one-resource-element-per-thread kernel
 * (no LDS staging). Kept as an independent implementation the tiled kernel is checked
 * against on the GPU; it is also what cet_lmmse uses for CET_NO_PER_RE_RX. */
int cet_lmmse_direct(const float* y, const float* H, int no_mode, float no_scalar,
                     const float* no_dev, float* x_hat, float* no_eff, long N, int Nr, int Ns,
                     void* stream);

/* cet_lmmse with the QPSK demapper fused into its epilogue: instead of (x_hat, no_eff) the kernel
 * writes llr [N,Ns,2] float32, the exact bit log-likelihood ratios ln(P(b=1)/P(b=0)) =
 * -2 sqrt(2) Re/Im(x_hat) / no_eff of Gray-mapped QPSK with bit 0 on the positive axis (the
 * "app" demapper that follows the equalizer in the reference's receive chain,
 * HPC/python/EvaluateE2E.py:292-319); 16 B less HBM traffic per stream than equalizing and
 * demapping in two passes. Same argument meaning and support as cet_lmmse. */
int cet_lmmse_llr(const float* y, const float* H, int no_mode, float no_scalar, const float* no_dev,
                  float* llr, long N, int Nr, int Ns, void* stream);

/* Block fading: cet_lmmse / cet_lmmse_llr for a channel estimate that is held over `h_share`
 * consecutive REs, as when one predicted channel serves the OFDM symbols of a slot (the
 * prediction of LocalSimulations/Predict.py:99-120 feeding the equalizer call of
 * EndtoEndtest.py:191). H is [ceil(N / h_share), Nr, Ns]: RE n uses H[n / h_share]; y, the noise
 * arguments and the outputs stay per RE, and the results are bit-identical to cet_lmmse on the
 * expanded H. HBM reads H once per block (512 / h_share instead of 512 B per RE at 16x4).
 * 1 <= h_share <= 65536. */
int cet_lmmse_block(const float* y, const float* H, int no_mode, float no_scalar,
                    const float* no_dev, float* x_hat, float* no_eff, long N, int h_share, int Nr,
                    int Ns, void* stream);
int cet_lmmse_block_llr(const float* y, const float* H, int no_mode, float no_scalar,
                        const float* no_dev, float* llr, long N, int h_share, int Nr, int Ns,
                        void* stream);

/* The equalizer on the operands of the reference's equalizer OBJECT, in their own layouts
 * (`lmmse_equ([y, h_hat, err_var, no])`, LocalSimulations/EndtoEndtest.py:133,191), with no
 * gather or permute pass around it:
 *   y        [B, Nr, n_sym, fft]          complex64 (num_rx = 1 squeezed out)
 *   h_hat    [B, Nr, Ns, n_sym, n_sc]     complex64 (Ns = num_tx * num_streams_per_tx)
 *   err_var  NULL, or [B, Nr, Ns, n_sym, n_sc] float32: variance of the estimation error; the
 *            noise covariance of receive antenna r is no + sum over the streams of err_var
 *   no_mode  0: no_scalar; 1: no_dev[B]; 2: no_dev[B, Nr]
 *   data_sym [n_ds] int32 OFDM symbols that carry data, eff_sc [n_sc] int32 subcarrier index in
 *            the fft grid of every effective subcarrier (device arrays)
 *   x_hat    [B, Ns, n_ds * n_sc] complex64, no_eff the same in float32 (OFDM symbol first, then
 *            subcarrier: pilot symbols, guard and DC carriers removed).
 * The complex arrays must be 8-byte aligned (CET_ERR_INVALID_ARG otherwise). */
int cet_lmmse_grid(const float* y, const float* h_hat, const float* err_var, int no_mode,
                   float no_scalar, const float* no_dev, const int32_t* data_sym,
                   const int32_t* eff_sc, float* x_hat, float* no_eff, int B, int Nr, int Ns,
                   int n_sym, int fft, int n_sc, int n_ds, void* stream);

/* ---- LS channel estimate with nearest-neighbour interpolation ------------------------------ */

/* Per-stream least-squares estimate at the pilots, spread over the resource grid by nearest
 * pilot: stands in for `LSChannelEstimator(rg, interpolation_type="nn")([y, no])`
 * (LocalSimulations/EndtoEndtest.py:130,184), with the pilot geometry resolved by the caller
 * into a table of T = streams * symbols * effective subcarriers entries:
 *   src[t]      index of the nearest pilot's observation inside one receive antenna's
 *               [symbols * fft_size] plane of y,
 *   pinv[t]     1 / pilot (complex, interleaved),   pabs2inv[t] = 1 / |pilot|^2.
 * y: device [planes, plane] complex64 (planes = batch * receive antennas);
 * h_hat: device [planes, T] complex64, err_var: device [planes, T] float32 = no / |pilot|^2.
 * The complex arrays must be 8-byte aligned (CET_ERR_INVALID_ARG otherwise). */
int cet_ls_nn(const float* y, const int32_t* src, const float* pinv, const float* pabs2inv,
              float no, float* h_hat, float* err_var, long planes, int plane, int T, void* stream);

/* ---- evaluation sums (what a rank contributes to the single all-reduce) ------------------- */

/* Per prediction slot s: sums[s] = sum (truth - pred)^2, sums[P + s] = sum truth^2 over batch and
 * the C real components (2 per antenna pair); NMSE_s = sums[s] / sums[P + s] is the quantity of
 * HPC/python/Evaluate.py:172-177. pred, truth: device [B,P,C] float32 (any C >= 1; 16-byte loads
 * when C % 4 == 0 and both arrays are 16-byte aligned, single floats otherwise);
 * sums: device double[2*P], overwritten. Accumulation in fp64. */
int cet_nmse_sums(const float* pred, const float* truth, double* sums, long B, int P, int C,
                  void* stream);

/* Hard-decision QPSK (bit = sign of the real / imaginary part) of x_hat against x_true, both
 * device complex64 [N] interleaved: out[0] = bit errors, out[1] = bits compared (2 N); the count
 * behind the BER of LocalSimulations/EndtoEndtest.py:196-200. out: device double[2], overwritten.
 * All three arrays 8-byte aligned (CET_ERR_INVALID_ARG otherwise). */
int cet_qpsk_bit_errors(const float* x_hat, const float* x_true, double* out, long N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CET_H */
