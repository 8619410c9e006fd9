"""Transformer prediction feeding LMMSE equalization, with BER / NMSE bookkeeping.

The reference's end-to-end scripts stop short of this coupling: `EndtoEndtest.py:182-192`
equalizes with the perfect channel (`h_hat = h_freq`, `err_var = 0.`) and
`HPC/python/EvaluateE2E.py:292-319` selects its predictor branch with a key that never
matches. The coupling implemented here is the one those scripts set up but do not reach:

    h_hat[slot] := reshape(real2complex(transformer(LoadBatch(h_hist), dec_inp)[0]))

for every subcarrier sample, held constant over the OFDM symbols of the predicted slot, then
per-resource-element LMMSE, hard-decision QPSK, bit errors and per-slot NMSE of the predicted
channel. Every subcarrier is independent: ranks take contiguous subcarrier ranges and the
only exchange is one all-reduce of the sums returned by `run` (`metrics.all_reduce_sums`).
"""
import math

import torch

from . import metrics
from .lmmse import lmmse
from .utils import decoder_input, load_batch_device, real2complex


def predict_slots(model, h_hist, label_len, pred_len):
    """h_hist complex [S, seq_len, Nr, Nt] (CUDA) -> predicted channels complex
    [S, pred_len, Nr, Nt]; lines 99-120 of LocalSimulations/Predict.py without the host
    round trip (`.cpu().detach()` at :117)."""
    S, _, Nr, Nt = h_hist.shape
    enc = load_batch_device(h_hist)
    dec = decoder_input(enc, label_len, pred_len)
    out = model(enc, dec)
    if isinstance(out, tuple):
        out = out[0]
    return real2complex(out).reshape(S, pred_len, Nr, Nt)


def group_antennas(h, nr0=4, nt0=2):
    """Splits a channel [S, T, Nr, Nt] into the (Nr/nr0)*(Nt/nt0) sub-arrays of nr0 x nt0 antenna
    pairs a predictor trained on nr0 x nt0 channels takes: returns [S*G, T, nr0, nt0], the groups
    of a subcarrier sample adjacent (group g = (receive block, transmit block) row-major)."""
    S, T, Nr, Nt = h.shape
    if Nr % nr0 or Nt % nt0:
        raise RuntimeError(f"{Nr}x{Nt} does not split into {nr0}x{nt0} groups")
    h = h.reshape(S, T, Nr // nr0, nr0, Nt // nt0, nt0).permute(0, 2, 4, 1, 3, 5)
    return h.reshape(S * (Nr // nr0) * (Nt // nt0), T, nr0, nt0)


def ungroup_antennas(hg, Nr, Nt):
    """Inverse of `group_antennas`: [S*G, T, nr0, nt0] -> [S, T, Nr, Nt]."""
    SG, T, nr0, nt0 = hg.shape
    gr, gt = Nr // nr0, Nt // nt0
    h = hg.reshape(SG // (gr * gt), gr, gt, T, nr0, nt0).permute(0, 3, 1, 4, 2, 5)
    return h.reshape(SG // (gr * gt), T, Nr, Nt)


def predict_slots_grouped(model, h_hist, label_len, pred_len, nr0=4, nt0=2):
    """`predict_slots` for an array larger than the one the predictor was trained on (the
    16 x 4 geometry with the reference's 4 x 2 checkpoints): every nr0 x nt0 group of antenna
    pairs is one sequence of the batch. h_hist [S, seq_len, Nr, Nt] -> [S, pred_len, Nr, Nt]."""
    _, _, Nr, Nt = h_hist.shape
    return ungroup_antennas(predict_slots(model, group_antennas(h_hist, nr0, nt0), label_len, pred_len),
                            Nr, Nt)


def predict_slots_recurrent(model, h_hist, pred_len):
    """Same for the GRU / LSTM / RNN predictors (`test_data`, HPC/python/Evaluate.py:147-157)."""
    S, _, Nr, Nt = h_hist.shape
    out = model.test_data(load_batch_device(h_hist), pred_len, h_hist.device)
    return real2complex(out).reshape(S, pred_len, Nr, Nt)


def qpsk(bits):
    """bits [..., 2] in {0,1} -> Gray-mapped QPSK ((1 - 2 b0) + j (1 - 2 b1)) / sqrt(2): bit 0 on
    the positive axis, the mapping of the reference's `Mapper("qam", 2)`. The hard decision
    compares signs (metrics.qpsk_bit_errors), so BER does not depend on the convention."""
    s = (1.0 - 2.0 * bits.to(torch.float32)) * (1.0 / math.sqrt(2.0))
    return torch.complex(s[..., 0], s[..., 1])


def transmit(h_true, n_sym, no, generator):
    """Synthetic uplink over the true channel: h_true complex [S, P, Nr, Nt]; draws QPSK
    symbols x [S, P, n_sym, Nt] and returns (y [S, P, n_sym, Nr], x) with
    y = h_true x + n, n ~ CN(0, no), the channel constant over the n_sym symbols of a slot."""
    S, P, Nr, Nt = h_true.shape
    dev = h_true.device
    bits = torch.randint(0, 2, (S, P, n_sym, Nt, 2), device=dev, generator=generator)
    x = qpsk(bits)
    y = torch.einsum("sprt,spnt->spnr", h_true, x)
    sigma = math.sqrt(no / 2.0)
    n = torch.complex(torch.randn(y.shape, device=dev, generator=generator),
                      torch.randn(y.shape, device=dev, generator=generator)) * sigma
    return (y + n).to(torch.complex64), x.to(torch.complex64)


def equalize_slots(h_hat, y, no, err_var=0.0):
    """h_hat complex [S, P, Nr, Nt] (one estimate per slot), y complex [S, P, n_sym, Nr].
    Returns (x_hat [S, P, n_sym, Nt], no_eff [S, P, n_sym, Nt]). `err_var` (scalar or
    broadcastable to [S, P, Nr, Nt]): variance of the estimation error of every h_hat entry;
    it enters the noise covariance as no + sum over streams of err_var per receive antenna. The
    slot's estimate is not expanded over the symbols: the equalizer's block-fading entry point
    reads one matrix per slot."""
    S, P, n_sym, Nr = y.shape
    Nt = h_hat.shape[3]
    N = S * P * n_sym
    H = h_hat.reshape(S * P, Nr, Nt)  # one matrix per slot, shared by its n_sym elements
    if not isinstance(err_var, torch.Tensor) or err_var.dim() == 0:
        noise = float(no) + Nt * float(err_var)  # the same variance on every receive antenna
    else:
        ev = torch.as_tensor(err_var, dtype=torch.float32, device=y.device)
        ev = torch.broadcast_to(ev, h_hat.shape).sum(-1)  # [S, P, Nr]
        noise = (ev + float(no)).unsqueeze(2).expand(S, P, n_sym, Nr).reshape(N, Nr).contiguous()
    x_hat, no_eff = lmmse(y.reshape(N, Nr), H, noise, h_share=n_sym)
    return x_hat.reshape(S, P, n_sym, Nt), no_eff.reshape(S, P, n_sym, Nt)


def qpsk_llr(x_hat, no_eff):
    """Exact ("app") bit log-likelihood ratios of the QPSK of `qpsk` after unbiased LMMSE:
    llr = ln(P(b=1)/P(b=0)) = -2 sqrt(2) Re/Im(x_hat) / no_eff. Returns [..., 2] float32.
    `equalize_slots_llr` produces the same numbers inside the equalizer kernel."""
    c = -2.0 * math.sqrt(2.0)
    return torch.stack([c * x_hat.real / no_eff, c * x_hat.imag / no_eff], dim=-1)


def equalize_slots_llr(h_hat, y, no):
    """`equalize_slots` followed by `qpsk_llr`, fused: returns llr [S, P, n_sym, Nt, 2]."""
    from .lmmse import lmmse_llr
    S, P, n_sym, Nr = y.shape
    Nt = h_hat.shape[3]
    N = S * P * n_sym
    H = h_hat.reshape(S * P, Nr, Nt)
    return lmmse_llr(y.reshape(N, Nr), H, float(no), h_share=n_sym).reshape(S, P, n_sym, Nt, 2)


def run(model, h_hist, h_future, label_len, pred_len, n_sym=14, no=0.1, seed=0, err_var=0.0,
        recurrent=False):
    """One evaluation pass over this rank's subcarrier samples.
    h_hist [S, seq_len, Nr, Nt], h_future [S, pred_len, Nr, Nt] complex (CUDA).
    Returns a dict of float64 tensors of partial sums:
      'nmse'  [2, pred_len]  (sum |H - H_hat|^2, sum |H|^2) per predicted slot
      'bits'  [2]            (bit errors, bits) with the predicted channel
      'bits_perfect' [2]     the same with the true channel as the estimate
    Sum them over ranks (`reduce`) before forming NMSE / BER."""
    g = torch.Generator(device=h_hist.device)
    g.manual_seed(seed)
    if recurrent:
        h_hat = predict_slots_recurrent(model, h_hist, pred_len)
    else:
        h_hat = predict_slots(model, h_hist, label_len, pred_len)
    y, x = transmit(h_future, n_sym, no, g)
    x_hat, _ = equalize_slots(h_hat, y, no, err_var)
    x_ref, _ = equalize_slots(h_future.to(torch.complex64), y, no)
    return {"nmse": metrics.nmse_partial_sums(h_hat, h_future.to(torch.complex64)),
            "bits": metrics.qpsk_bit_errors(x_hat, x),
            "bits_perfect": metrics.qpsk_bit_errors(x_ref, x)}


def reduce(sums):
    """All-reduces the partial sums of `run` over the ranks in ONE collective and returns
    {'nmse_db': [pred_len], 'ber': float, 'ber_perfect_csi': float}."""
    keys = ("nmse", "bits", "bits_perfect")
    flat = torch.cat([sums[k].reshape(-1) for k in keys])
    metrics.all_reduce_sums(flat)
    P = sums["nmse"].shape[1]
    nm = flat[:2 * P].reshape(2, P)
    b, bp = flat[2 * P:2 * P + 2], flat[2 * P + 2:2 * P + 4]
    return {"nmse_db": metrics.nmse_db(nm).tolist(), "ber": float(b[0] / b[1]),
            "ber_perfect_csi": float(bp[0] / bp[1])}
