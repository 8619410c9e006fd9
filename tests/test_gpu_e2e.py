"""Transformer prediction -> LMMSE equalization -> BER / NMSE on the GPU, checked step by step
against the CPU checkers (same inputs) and through size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import DEFAULT_ARGS, cfg_from_args, dec_input_np, load_batch_np, load_npz
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import blobs, e2e, synth
from test_gpu_informer import build

pytestmark = pytest.mark.gpu

TOL_H = 2e-5     # predicted channel entries (the predictor's output tolerance)
TOL_X = 2e-4     # equalized symbols, magnitude ~1 (4x2 Gram in fp32)


@pytest.fixture(scope="module")
def model():
    sd = dict(load_npz("informer_ckpt.npz"))
    return build("InformerStack", DEFAULT_ARGS, sd).cuda()


def _channels(S, seed):
    noisy, clean = synth.channel_histories(S, T=30, Nr=4, Nt=2, seed=seed)
    dev = torch.device("cuda:0")
    return (torch.from_numpy(noisy[:, :25]).to(dev), torch.from_numpy(clean[:, 25:]).to(dev), noisy)


def _cdl_b_channels():
    """The reference's stored CDL-B sequences (CR_data_{30,36,60}: 192 x 30 slots of a 4x2
    channel), the data the shipped checkpoint was trained for."""
    d = load_npz("cr_pairs.npz")
    H = np.concatenate([d["data_30"], d["data_36"], d["data_60"]])
    dev = torch.device("cuda:0")
    return torch.from_numpy(H[:, :25]).to(dev), torch.from_numpy(H[:, 25:]).to(dev)


def test_prediction_step_matches_checker(model, oracle):
    h_hist, _, noisy = _channels(96, 3)
    h_hat = e2e.predict_slots(model, h_hist, 10, 5)
    x_enc = load_batch_np(noisy[:, :25])
    cfg = cfg_from_args(DEFAULT_ARGS, 1)
    blob = blobs.pack_informer(dict(load_npz("informer_ckpt.npz")), cfg)
    want, _, _ = oracle.informer_forward(cfg, blob, x_enc, dec_input_np(x_enc, 10, 5))
    want = cet.real2complex(want).reshape(96, 5, 4, 2)
    assert np.abs(h_hat.cpu().numpy() - want).max() < TOL_H


def test_equalization_step_matches_checker(oracle):
    S, P, n_sym, Nr, Nt = 40, 5, 14, 4, 2
    g = torch.Generator(device="cuda:0").manual_seed(5)
    _, h_future, _ = _channels(S, 4)
    h_hat = h_future + 0.05 * torch.complex(torch.randn(h_future.shape, device="cuda:0", generator=g),
                                            torch.randn(h_future.shape, device="cuda:0", generator=g))
    y, x = e2e.transmit(h_future, n_sym, 0.05, g)
    for err_var in (0.0, 0.005):
        x_hat, no_eff = e2e.equalize_slots(h_hat, y, 0.05, err_var)
        H = h_hat.unsqueeze(2).expand(S, P, n_sym, Nr, Nt).reshape(-1, Nr, Nt).cpu().numpy()
        no = 0.05 + Nt * err_var
        xw, nw = oracle.lmmse(y.reshape(-1, Nr).cpu().numpy(), H, no)
        assert np.abs(x_hat.reshape(-1, Nt).cpu().numpy() - xw).max() < TOL_X
        assert np.abs(no_eff.reshape(-1, Nt).cpu().numpy() / nw - 1).max() < 1e-3
    llr = e2e.qpsk_llr(x_hat, no_eff)
    assert llr.shape == (S, P, n_sym, Nt, 2)
    assert torch.equal(llr[..., 0] < 0, x_hat.real > 0)  # llr = ln P(b=1)/P(b=0), bit 0 positive
    # the fused equalizer + demapper kernel gives the same LLRs as the two-step path
    x2, ne2 = e2e.equalize_slots(h_hat, y, 0.05)
    fused = e2e.equalize_slots_llr(h_hat, y, 0.05)
    two_step = e2e.qpsk_llr(x2, ne2)
    assert torch.allclose(fused, two_step, rtol=1e-5, atol=1e-5)
    # ... and they decide the transmitted bits: a positive LLR means bit 1 = negative axis
    assert ((fused[..., 0] > 0) == (x.real < 0)).float().mean() > 0.99


def test_end_to_end_properties(model):
    h_hist, h_future = _cdl_b_channels()
    S = h_hist.shape[0]
    sums = e2e.run(model, h_hist, h_future, 10, 5, n_sym=14, no=0.05, seed=1)
    res = e2e.reduce(sums)
    assert sums["bits"][1] == 2 * S * 5 * 14 * 2
    # a noiseless link equalized with the true channel makes no errors
    clean = e2e.run(model, h_hist, h_future, 10, 5, n_sym=14, no=1e-6, seed=1)
    assert clean["bits_perfect"][0] == 0
    # the reference's own outputs for these sequences reach about -14 dB per slot
    assert max(res["nmse_db"]) < -12.0
    # the predicted channel cannot beat the true one by more than noise, and is close to it
    assert res["ber_perfect_csi"] <= res["ber"] + 2e-3
    assert res["ber"] < 0.1
    # sharding the subcarriers and summing the partial sums gives the same sums
    a = e2e.run(model, h_hist[:100], h_future[:100], 10, 5, no=0.05, seed=1)
    b = e2e.run(model, h_hist[100:], h_future[100:], 10, 5, no=0.05, seed=1)
    assert torch.allclose(a["nmse"] + b["nmse"], sums["nmse"], rtol=1e-9)
    assert (a["bits"] + b["bits"])[1] == sums["bits"][1]


def test_evaluation_sum_kernels_match_float64():
    """NMSE partial sums and QPSK bit errors from the HIP reductions vs numpy float64."""
    from channelestimationtransformer_amd import metrics
    rng = np.random.default_rng(3)
    # C % 4 != 0 (an odd number of antenna pairs) and 1 float: the one-float-per-thread kernel
    for B, P, C in ((1, 5, 16), (777, 5, 16), (4096, 25, 32), (777, 5, 6), (300, 3, 18), (50, 7, 1)):
        t = rng.standard_normal((B, P, C)).astype(np.float32)
        p = (t + 0.1 * rng.standard_normal((B, P, C))).astype(np.float32)
        got = metrics.nmse_partial_sums(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
        d = t.astype(np.float64) - p.astype(np.float64)
        want = np.stack([(d * d).sum((0, 2)), (t.astype(np.float64) ** 2).sum((0, 2))])
        assert got.shape == (2, P) and np.allclose(got, want, rtol=1e-12)
    # complex inputs [B, P, Nr, Nt] take the same path
    tc = (rng.standard_normal((33, 5, 4, 2)) + 1j * rng.standard_normal((33, 5, 4, 2))).astype(np.complex64)
    pc = (tc + 0.05).astype(np.complex64)
    got = metrics.nmse_partial_sums(torch.from_numpy(pc).cuda(), torch.from_numpy(tc).cuda()).cpu().numpy()
    want = np.stack([(np.abs(tc.astype(np.complex128) - pc) ** 2).sum((0, 2, 3)), (np.abs(tc.astype(np.complex128)) ** 2).sum((0, 2, 3))])
    assert np.allclose(got, want, rtol=1e-12)
    tc = (rng.standard_normal((41, 4, 3, 1)) + 1j * rng.standard_normal((41, 4, 3, 1))).astype(np.complex64)  # 3 x 1 antennas
    pc = (tc * 0.9).astype(np.complex64)
    got = metrics.nmse_partial_sums(torch.from_numpy(pc).cuda(), torch.from_numpy(tc).cuda()).cpu().numpy()
    want = np.stack([(np.abs(tc.astype(np.complex128) - pc) ** 2).sum((0, 2, 3)), (np.abs(tc.astype(np.complex128)) ** 2).sum((0, 2, 3))])
    assert np.allclose(got, want, rtol=1e-12)
    # a view that starts 4 bytes into its storage: the one-float kernel, same sums
    tt = torch.from_numpy(rng.standard_normal(64 * 5 * 16 + 1).astype(np.float32)).cuda()
    t1, p1 = tt[1:].view(64, 5, 16), (tt[1:] * 1.05).view(64, 5, 16)
    assert t1.data_ptr() % 16 != 0
    got = metrics.nmse_partial_sums(p1, t1)
    assert torch.allclose(got, metrics.nmse_partial_sums(p1.clone(), t1.clone()), rtol=1e-12)
    for N in (1, 2, 1000, 1001, 1 << 20, (1 << 21) + 7):
        xt = ((2 * rng.integers(0, 2, N) - 1) + 1j * (2 * rng.integers(0, 2, N) - 1)).astype(np.complex64)
        xh = (xt + 1.2 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
        got = metrics.qpsk_bit_errors(torch.from_numpy(xh).cuda(), torch.from_numpy(xt).cuda()).cpu().numpy()
        want = ((xh.real > 0) != (xt.real > 0)).sum() + ((xh.imag > 0) != (xt.imag > 0)).sum()
        assert got[0] == want and got[1] == 2 * N
        if N > 2:  # views that start 8 bytes into their storage: not 16-byte aligned
            got = metrics.qpsk_bit_errors(torch.from_numpy(xh).cuda()[1:], torch.from_numpy(xt).cuda()[1:]).cpu().numpy()
            want = ((xh.real > 0) != (xt.real > 0))[1:].sum() + ((xh.imag > 0) != (xt.imag > 0))[1:].sum()
            assert got[0] == want and got[1] == 2 * (N - 1)


def test_device_channel_generator():
    noisy, clean = synth.channel_histories_device(2048, "cuda:0", seed=3)
    assert noisy.shape == clean.shape == (2048, 30, 4, 2) and noisy.dtype == torch.complex64
    assert abs(float((clean.abs() ** 2).mean()) - 1.0) < 1e-3            # unit mean power
    snr = float((clean.abs() ** 2).mean() / ((noisy - clean).abs() ** 2).mean())
    assert abs(10 * np.log10(snr) - 20.0) < 0.2                          # 20 dB by default
    again, _ = synth.channel_histories_device(2048, "cuda:0", seed=3)
    assert torch.equal(noisy, again)                                     # seeded


def test_scaled_geometry_groups_antenna_pairs(model):
    """16 x 4 channel with the 4 x 2 checkpoint: every 4 x 2 group of antenna pairs is one
    predictor sequence. The grouped prediction must place exactly the per-group predictions,
    and the 16 x 4 block-fading equalization with it must work end to end."""
    h_hist, h_future = _cdl_b_channels()          # 192 sequences of a 4x2 channel
    big_hist = e2e.ungroup_antennas(h_hist, 16, 4)   # 24 subcarrier samples of a 16x4 channel
    big_future = e2e.ungroup_antennas(h_future, 16, 4)
    assert big_hist.shape == (24, 25, 16, 4)
    assert torch.equal(e2e.group_antennas(big_hist), h_hist)
    # entry (rx, tx) of sample s comes from group (rx // 4, tx // 2) = sequence 8 s + 2 (rx // 4) + tx // 2
    assert torch.equal(big_hist[3, :, 9, 3], h_hist[8 * 3 + 2 * 2 + 1, :, 1, 1])
    small = e2e.predict_slots(model, h_hist, 10, 5)
    big = e2e.predict_slots_grouped(model, big_hist, 10, 5)
    assert torch.equal(e2e.group_antennas(big), small)
    g = torch.Generator(device="cuda").manual_seed(5)
    y, x = e2e.transmit(big_future.to(torch.complex64), 14, 0.05, g)
    x_hat, no_eff = e2e.equalize_slots(big, y, 0.05)
    assert x_hat.shape == (24, 5, 14, 4) and no_eff.shape == (24, 5, 14, 4)
    # the eight stored sequences that make up a sample are strongly correlated, so the 16 x 4
    # matrix is poorly conditioned and the -14 dB prediction error costs bit errors; telling the
    # equalizer about that error (err_var) must help, and the true channel must do best
    ber = lambda xh: float((lambda b: b[0] / b[1])(cet.metrics.qpsk_bit_errors(xh, x)))
    err_var = float((big - big_future).abs().pow(2).mean())
    x_ev, _ = e2e.equalize_slots(big, y, 0.05, err_var)
    x_true, _ = e2e.equalize_slots(big_future.to(torch.complex64), y, 0.05)
    assert ber(x_true) <= ber(x_ev) <= ber(x_hat) < 0.3
    # equalizing with the slot's matrix expanded over the symbols gives the same numbers
    H = big.unsqueeze(2).expand(24, 5, 14, 16, 4).reshape(-1, 16, 4)
    x_ref, ne_ref = cet.lmmse(y.reshape(-1, 16), H, 0.05)
    assert torch.equal(torch.view_as_real(x_ref.reshape(x_hat.shape)), torch.view_as_real(x_hat))
    assert torch.equal(ne_ref.reshape(no_eff.shape), no_eff)
