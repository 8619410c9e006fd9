"""Parity of the HIP LMMSE equalizer with the float64 CPU checker (PARITY UNPINNED: the
reference holds no vector for this path; see oracle/cet_oracle.c)."""
import numpy as np
import pytest
import torch

import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import synth

pytestmark = pytest.mark.gpu

# complex64 Gram + Cholesky vs float64: relative tolerance on well-conditioned 16x4 problems
RTOL = 2e-4


def run(y, H, no, direct=False):
    no_t = torch.from_numpy(no).cuda() if isinstance(no, np.ndarray) else no
    x, ne = cet.lmmse(torch.from_numpy(y).cuda(), torch.from_numpy(H).cuda(), no_t, direct=direct)
    return x.cpu().numpy(), ne.cpu().numpy()


def close(a, b):
    return np.abs(a - b).max() <= RTOL * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("Nr,Ns", [(16, 4), (16, 2), (16, 1), (8, 4), (8, 2), (8, 1), (4, 4), (4, 2),
                                   (4, 1), (2, 2), (2, 1)])
@pytest.mark.parametrize("N", [1, 63, 64, 1000, 4099])
def test_all_shapes_ragged_sizes(oracle, Nr, Ns, N):
    y, H, _ = synth.lmmse_problem(N, Nr, Ns, 0.1, seed=N + Nr)
    want_x, want_ne = oracle.lmmse(y, H, 0.1)
    for direct in (False, True):
        x, ne = run(y, H, 0.1, direct)
        assert close(x, want_x) and close(ne, want_ne), f"direct={direct}"


def test_noise_modes(oracle):
    rng = np.random.default_rng(0)
    y, H, _ = synth.lmmse_problem(3000, 16, 4, 0.1, seed=1)
    per_re = rng.uniform(0.02, 1.0, 3000).astype(np.float32)
    per_rx = rng.uniform(0.02, 1.0, (3000, 16)).astype(np.float32)
    for no in (per_re, per_rx):
        want_x, want_ne = oracle.lmmse(y, H, no)
        for direct in (False, True):
            x, ne = run(y, H, no, direct)
            assert close(x, want_x) and close(ne, want_ne)


def test_known_answers_and_edge_cases():
    N, n = 70, 4
    H = np.tile(np.eye(n, dtype=np.complex64)[None], (N, 1, 1))
    y = (np.arange(N * n).reshape(N, n) % 7 + 1j).astype(np.complex64)
    x, ne = run(y, H, 0.25)
    assert np.allclose(x, y, atol=1e-5) and np.allclose(ne, 0.25, atol=1e-6)
    g = np.array([2.0, 0.5j, -1.0, 3.0 + 4.0j], dtype=np.complex64)
    H = np.tile(np.diag(g)[None], (N, 1, 1)).astype(np.complex64)
    x, ne = run(y, H, 0.5)
    assert np.allclose(x, y / g, atol=1e-4) and np.allclose(ne, 0.5 / np.abs(g) ** 2, rtol=1e-4)
    H = np.ones((N, 4, 2), np.complex64)  # rank deficient, regularised by no > 0
    x, ne = run(np.ones((N, 4), np.complex64), H, 0.1)
    assert np.isfinite(x).all() and np.isfinite(ne).all()
    e = torch.empty(0, 16, dtype=torch.complex64, device="cuda")
    x0, ne0 = cet.lmmse(e, torch.empty(0, 16, 4, dtype=torch.complex64, device="cuda"), 0.1)
    assert x0.shape == (0, 4) and ne0.shape == (0, 4)


def test_snr_extremes(oracle):
    for no in (1e-4, 10.0):
        y, H, _ = synth.lmmse_problem(2000, 16, 4, no, seed=7)
        want_x, want_ne = oracle.lmmse(y, H, no)
        x, ne = run(y, H, no)
        assert np.abs(x - want_x).max() <= 5e-4 * max(1.0, np.abs(want_x).max())
        assert np.abs(ne - want_ne).max() <= 5e-4 * np.abs(want_ne).max()


def test_tiled_equals_direct_at_scale():
    """2^21 REs (1.4 GB of H+y): the two independent kernels agree, QPSK decisions recover
    the transmitted symbols at this SNR, and a sub-range call returns the same bits."""
    N = 1 << 21
    g = torch.Generator(device="cuda").manual_seed(0)
    H = torch.view_as_complex(torch.randn(N, 16, 4, 2, device="cuda", generator=g) * 0.5 ** 0.5)
    xs = torch.view_as_complex((torch.randint(0, 2, (N, 4, 2), device="cuda", generator=g).float() * 2 - 1) * 0.5 ** 0.5)
    noise = torch.view_as_complex(torch.randn(N, 16, 2, device="cuda", generator=g) * (0.05 / 2) ** 0.5)
    y = torch.einsum("nrs,ns->nr", H, xs) + noise
    x1, ne1 = cet.lmmse(y, H, 0.05)
    x2, ne2 = cet.lmmse(y, H, 0.05, direct=True)
    assert (x1 - x2).abs().max() < 1e-4 and (ne1 - ne2).abs().max() < 1e-5
    ber = ((x1.real > 0) != (xs.real > 0)).float().mean() + ((x1.imag > 0) != (xs.imag > 0)).float().mean()
    assert ber < 1e-3
    lo, hi = 12345, 12345 + 5000
    x3, _ = cet.lmmse(y[lo:hi], H[lo:hi], 0.05)
    assert torch.equal(x3, x1[lo:hi])


def test_equalizer_object_on_the_reference_geometry(oracle):
    """`LMMSEEqualizer(rg, sm)([y, h_hat, err_var, no])` with the grid of
    LocalSimulations/EndtoEndtest.py:50-59 (4 rx antennas, 2 streams, 14 symbols, fft 76)."""
    rg = cet.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=15e3, num_tx=1,
                          num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                          dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[0])
    sm = cet.StreamManagement(np.array([[1]]), 2)
    equ = cet.LMMSEEqualizer(rg, sm)
    rng = np.random.default_rng(0)
    b = 3
    h = (rng.standard_normal((b, 1, 4, 1, 2, 14, 64)) + 1j * rng.standard_normal((b, 1, 4, 1, 2, 14, 64))).astype(np.complex64)
    y = (rng.standard_normal((b, 1, 4, 14, 76)) + 1j * rng.standard_normal((b, 1, 4, 14, 76))).astype(np.complex64)
    no = 0.2
    x_hat, no_eff = equ([torch.from_numpy(y).cuda(), torch.from_numpy(h).cuda(), 0., no])
    assert x_hat.shape == (b, 1, 2, 832) and no_eff.shape == (b, 1, 2, 832)
    sc = rg.effective_subcarrier_ind
    yd = y[:, 0][:, :, 1:, :][:, :, :, sc].transpose(0, 2, 3, 1).reshape(-1, 4)
    hd = h[:, 0, :, 0][:, :, :, 1:, :].transpose(0, 3, 4, 1, 2).reshape(-1, 4, 2)
    wx, wne = oracle.lmmse(yd, hd, no)
    wx = wx.reshape(b, 832, 2).transpose(0, 2, 1)[:, None]
    wne = wne.reshape(b, 832, 2).transpose(0, 2, 1)[:, None]
    assert close(x_hat.cpu().numpy(), wx) and close(no_eff.cpu().numpy(), wne)
    # estimation-error variance folds into a per-antenna noise term
    ev = 0.05
    x2, ne2 = equ([torch.from_numpy(y).cuda(), torch.from_numpy(h).cuda(), torch.tensor(ev), no])
    wx2, wne2 = oracle.lmmse(yd, hd, np.full((yd.shape[0], 4), no + 2 * ev, np.float32))
    assert close(x2.cpu().numpy(), wx2.reshape(b, 832, 2).transpose(0, 2, 1)[:, None])


@pytest.mark.parametrize("Nr,Ns", [(16, 4), (8, 2), (4, 2), (4, 1)])
def test_fused_llr_epilogue(Nr, Ns, oracle):
    """cet_lmmse_llr = -2 sqrt(2) x_hat / no_eff of the checker, per noise mode (the per-antenna
    mode runs the direct kernel, the others the tiled one), ragged N."""
    N = 4099
    y, H, _ = synth.lmmse_problem(N, Nr, Ns, 0.2, seed=11)
    yt, Ht = torch.from_numpy(y).cuda(), torch.from_numpy(H).cuda()
    rng = np.random.default_rng(1)
    for no in (0.2, rng.uniform(0.05, 0.5, N).astype(np.float32),
               rng.uniform(0.05, 0.5, (N, Nr)).astype(np.float32)):
        wx, wne = oracle.lmmse(y, H, no)
        want = -2.0 * np.sqrt(2.0) * np.stack([wx.real, wx.imag], -1) / wne[..., None]
        no_t = no if isinstance(no, float) else torch.from_numpy(no).cuda()
        got = cet.lmmse_llr(yt, Ht, no_t).cpu().numpy()
        assert got.shape == (N, Ns, 2)
        assert np.abs(got - want).max() < 2e-4 * np.abs(want).max()


def test_ls_estimator_nearest_neighbour(oracle):
    """LS estimate + nearest-neighbour interpolation against the brute-force statement, on the
    reference's grid (14 symbols, fft 76, guards 5/6, DC nulled, pilots on symbol 0; here two
    pilot symbols to exercise the time axis) and on the estimator -> equalizer chain."""
    rg = cet.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=15e3, num_tx=1,
                          num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                          dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    assert rg.num_effective_subcarriers == 64 and rg.pilots.shape == (1, 2, 2, 64)
    assert np.allclose((np.abs(rg.pilots) ** 2).mean(-1), 1.0)
    est = cet.LSChannelEstimator(rg, interpolation_type="nn")
    rng = np.random.default_rng(4)
    b, ant = 3, 4
    y = (rng.standard_normal((b, 1, ant, 14, 76)) + 1j * rng.standard_normal((b, 1, ant, 14, 76))).astype(np.complex64)
    h_hat, err_var = est([torch.from_numpy(y).cuda(), 0.3])
    assert h_hat.shape == err_var.shape == (b, 1, ant, 1, 2, 14, 64)
    wh, wev = oracle.ls_nn_numpy(y[:, 0], rg.pilots.reshape(2, 2, 64), rg.pilot_ofdm_symbol_indices,
                                 rg.effective_subcarrier_ind, 0.3)
    assert np.abs(h_hat.cpu().numpy().reshape(wh.shape) - wh).max() < 1e-5
    assert np.abs(err_var.cpu().numpy().reshape(wev.shape) - wev).max() < 1e-6
    # a flat channel observed without noise through the pilots is recovered exactly everywhere
    H = (rng.standard_normal((b, ant, 2)) + 1j * rng.standard_normal((b, ant, 2))).astype(np.complex64)
    grid = np.zeros((b, 1, ant, 14, 76), np.complex64)
    for ip, s in enumerate(rg.pilot_ofdm_symbol_indices):
        grid[:, 0, :, s, rg.effective_subcarrier_ind] = np.einsum("bak,kf->fba", H, rg.pilots[0, :, ip, :])
    h2, _ = est([torch.from_numpy(grid).cuda(), 0.0])
    assert np.abs(h2.cpu().numpy() - H[:, None, :, None, :, None, None]).max() < 1e-5
    # estimator output feeds the equalizer object unchanged
    x_hat, no_eff = cet.LMMSEEqualizer(rg, cet.StreamManagement(np.array([[1]]), 2))(
        [torch.from_numpy(y).cuda(), h_hat, err_var, 0.3])
    assert x_hat.shape == (b, 1, 2, 12 * 64) and torch.isfinite(no_eff).all()


@pytest.mark.parametrize("fft,guards,dc,n_sym,streams,pil,b,ant", [
    (76, [5, 6], True, 14, 2, [2, 11], 5, 2),    # 10 planes: the last group of four is ragged
    (15, [0, 0], False, 7, 1, [3], 3, 3),        # 105 table entries (odd): the general kernel
    (24, [2, 1], True, 5, 2, [0], 1, 1),         # a single plane
])
def test_ls_estimator_kernel_forms(oracle, fft, guards, dc, n_sym, streams, pil, b, ant):
    """The estimator runs a kernel that takes two table entries and four antenna planes per
    thread when the table is even, and a general one otherwise: both against the brute-force
    statement, with plane counts that are no multiple of four."""
    rg = cet.ResourceGrid(num_ofdm_symbols=n_sym, fft_size=fft, subcarrier_spacing=15e3, num_tx=1,
                          num_streams_per_tx=streams, cyclic_prefix_length=1, num_guard_carriers=guards,
                          dc_null=dc, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=pil)
    n_eff = rg.num_effective_subcarriers
    est = cet.LSChannelEstimator(rg, interpolation_type="nn")
    rng = np.random.default_rng(fft + b)
    y = (rng.standard_normal((b, 1, ant, n_sym, fft)) + 1j * rng.standard_normal((b, 1, ant, n_sym, fft))).astype(np.complex64)
    h_hat, err_var = est([torch.from_numpy(y).cuda(), 0.2])
    assert h_hat.shape == err_var.shape == (b, 1, ant, 1, streams, n_sym, n_eff)
    wh, wev = oracle.ls_nn_numpy(y[:, 0], rg.pilots.reshape(streams, len(pil), n_eff), rg.pilot_ofdm_symbol_indices,
                                 rg.effective_subcarrier_ind, 0.2)
    assert np.abs(h_hat.cpu().numpy().reshape(wh.shape) - wh).max() < 1e-5
    assert np.abs(err_var.cpu().numpy().reshape(wev.shape) - wev).max() < 1e-6


@pytest.mark.parametrize("Nr,Ns", [(16, 4), (8, 4), (4, 2), (2, 1)])
@pytest.mark.parametrize("N,h_share", [(14 * 300, 14), (14 * 300 - 5, 14), (4099, 3), (1000, 1000),
                                       (77, 2000), (64, 1)])
@pytest.mark.parametrize("kernel", ["shared_factorisation", "gather"])
def test_block_fading_equals_expanded_channel(Nr, Ns, N, h_share, kernel, oracle):
    """One channel matrix held over h_share consecutive resource elements (a predicted slot):
    the block-fading entry points must return, bit for bit, what the per-element entry points
    return on the expanded H — ragged last block, ragged last tile, every noise mode, the fused
    demapper — and the float64 checker's values within the usual tolerance. Two kernels serve
    the entry points: one factorises each matrix once and applies it to the block's symbols
    (scalar noise), the other is the per-element kernel gathering its matrix (the other noise
    modes; a per-element noise array that holds one value everywhere sends the scalar case
    through it too, and must reproduce the scalar entry point's bits)."""
    nb = -(-N // h_share)
    y, _, _ = synth.lmmse_problem(N, Nr, Ns, 0.1, seed=N + Nr + h_share)
    _, Hb, _ = synth.lmmse_problem(nb, Nr, Ns, 0.1, seed=7 * N + Ns)
    H = np.repeat(Hb, h_share, axis=0)[:N]
    yt, Hbt, Ht = (torch.from_numpy(a).cuda() for a in (y, Hb, H))
    rng = np.random.default_rng(N)
    per_re = torch.from_numpy(rng.uniform(0.02, 1.0, N).astype(np.float32)).cuda()
    per_rx = torch.from_numpy(rng.uniform(0.02, 1.0, (N, Nr)).astype(np.float32)).cuda()
    for no in (0.1, per_re, per_rx):
        x0, ne0 = cet.lmmse(yt, Ht, no)
        x1, ne1 = cet.lmmse(yt, Hbt, no, h_share=h_share)
        assert torch.equal(torch.view_as_real(x0), torch.view_as_real(x1)) and torch.equal(ne0, ne1)
    want_x, want_ne = oracle.lmmse(y, H, 0.1)
    no_arg = 0.1 if kernel == "shared_factorisation" else torch.full((N,), 0.1, device="cuda")
    x1, ne1 = cet.lmmse(yt, Hbt, no_arg, h_share=h_share)
    assert close(x1.cpu().numpy(), want_x) and close(ne1.cpu().numpy(), want_ne)
    x0, ne0 = cet.lmmse(yt, Ht, 0.1)
    assert torch.equal(torch.view_as_real(x0), torch.view_as_real(x1)) and torch.equal(ne0, ne1)
    assert torch.equal(cet.lmmse_llr(yt, Ht, 0.1), cet.lmmse_llr(yt, Hbt, no_arg, h_share=h_share))
    with pytest.raises(RuntimeError):
        cet.lmmse(yt, Hbt[:-1] if nb > 1 else Ht[:0], 0.1, h_share=h_share)


def test_equalizer_object_scalars_on_host_or_device():
    """`no` and `err_var` as Python numbers (kept on the host: no synchronisation per call), as
    0-d device tensors (kept on the device) and mixed give the same results."""
    rg = cet.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=15e3, num_tx=1,
                          num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                          dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    equ = cet.LMMSEEqualizer(rg, cet.StreamManagement(np.array([[1]]), 2))
    g = torch.Generator(device="cuda").manual_seed(5)
    y = torch.view_as_complex(torch.randn(6, 1, 4, 14, 76, 2, device="cuda", generator=g))
    h = torch.view_as_complex(torch.randn(6, 1, 4, 1, 2, 14, 64, 2, device="cuda", generator=g))
    x0, n0 = equ([y, h, 0.02, 0.1])
    dev = lambda v: torch.tensor(v, device="cuda")
    for ev, no in ((dev(0.02), dev(0.1)), (0.02, dev(0.1)), (dev(0.02), 0.1), (np.float32(0.02), np.float64(0.1))):
        x1, n1 = equ([y, h, ev, no])
        assert torch.allclose(torch.view_as_real(x1), torch.view_as_real(x0), rtol=1e-5, atol=1e-6)
        assert torch.allclose(n1, n0, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("Nr,Ns", [(16, 4), (16, 2), (16, 1), (8, 4), (8, 2), (4, 4), (4, 2), (2, 2), (2, 1)])
def test_block_fading_staged_results_every_shape(Nr, Ns):
    """The shared-factorisation kernel collects a tile's results in LDS and stores them as whole
    lines when they fit (and stores them directly when they do not): block lengths on both sides
    of that limit, odd ones (a 16-byte piece of no_eff then spans two blocks when Ns < 4), a
    ragged last block and tile, and enough tiles that every wave works on several in a row — bit
    for bit what the per-element kernel returns on the expanded channel, demapper included."""
    for h_share, nb in ((14, 150000), (7, 9001), (13, 5000), (28, 3000), (97, 700)):
        N = h_share * nb - (h_share // 2)
        g = torch.Generator(device="cuda").manual_seed(Nr * 100 + Ns + h_share)
        yt = torch.view_as_complex(torch.randn(N, Nr, 2, device="cuda", generator=g))
        Hbt = torch.view_as_complex(torch.randn(nb, Nr, Ns, 2, device="cuda", generator=g))
        Ht = Hbt.repeat_interleave(h_share, dim=0)[:N]
        x0, ne0 = cet.lmmse(yt, Ht, 0.07)
        x1, ne1 = cet.lmmse(yt, Hbt, 0.07, h_share=h_share)
        assert torch.equal(torch.view_as_real(x0), torch.view_as_real(x1)) and torch.equal(ne0, ne1)
        assert torch.equal(cet.lmmse_llr(yt, Ht, 0.07), cet.lmmse_llr(yt, Hbt, 0.07, h_share=h_share))
        del Ht, x0, x1, ne0, ne1


@pytest.mark.parametrize("ant,ntx,nst", [(4, 1, 2), (16, 2, 2), (8, 1, 1), (2, 1, 2)])
def test_equalizer_object_reads_the_grid_in_place(ant, ntx, nst, oracle):
    """The equalizer object hands y [b,1,ant,sym,fft] and h_hat [b,1,ant,tx,st,sym,sc] to one
    kernel that gathers data symbols / effective subcarriers itself. Its arithmetic is that of
    the one-RE-per-thread kernel on the gathered operands: the results must be bit-identical to
    that route for a scalar, per-batch and per-antenna `no` and for a full err_var tensor, and
    agree with the float64 checker."""
    K = ntx * nst
    rg = cet.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=15e3, num_tx=ntx,
                          num_streams_per_tx=nst, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                          dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    equ = cet.LMMSEEqualizer(rg, cet.StreamManagement(np.ones((1, ntx), int), nst))
    rng = np.random.default_rng(ant + K)
    b, n_sc = 5, rg.num_effective_subcarriers
    cplx = lambda *s: (rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(np.complex64)
    y, h = cplx(b, 1, ant, 14, 76), cplx(b, 1, ant, ntx, nst, 14, n_sc)
    ev = rng.uniform(0.0, 0.2, h.shape).astype(np.float32)
    sym, sc = np.asarray(rg.data_symbol_ind), np.asarray(rg.effective_subcarrier_ind)
    n_d = len(sym) * n_sc
    yd = y[:, 0][:, :, sym][:, :, :, sc].transpose(0, 2, 3, 1).reshape(-1, ant)
    hd = h[:, 0].reshape(b, ant, K, 14, n_sc)[:, :, :, sym].transpose(0, 3, 4, 1, 2).reshape(-1, ant, K)
    evd = ev[:, 0].reshape(b, ant, K, 14, n_sc)[:, :, :, sym].transpose(0, 3, 4, 1, 2).reshape(-1, ant, K).sum(-1)
    yt, ht = torch.from_numpy(y).cuda(), torch.from_numpy(h).cuda()
    no_b = rng.uniform(0.05, 0.5, b).astype(np.float32)
    no_ba = rng.uniform(0.05, 0.5, (b, 1, ant)).astype(np.float32)
    per_re = lambda v: np.broadcast_to(v, (b, len(sym), n_sc, ant)).reshape(-1, ant).astype(np.float32).copy()
    cases = [
        (0.0, 0.2, 0.2),
        (0.0, torch.from_numpy(no_b).cuda(), np.repeat(no_b, n_d)),
        (0.0, torch.from_numpy(no_ba).cuda(), per_re(no_ba.reshape(b, 1, 1, ant))),
        (torch.from_numpy(ev).cuda(), 0.2, np.float32(0.2) + evd),
        (torch.from_numpy(ev).cuda(), torch.from_numpy(no_ba).cuda(), per_re(no_ba.reshape(b, 1, 1, ant)) + evd),
        (torch.tensor(0.03), torch.from_numpy(no_b).cuda(), np.repeat(no_b, n_d) + K * 0.03),
    ]
    back = lambda a: a.reshape(b, n_d, ntx, nst).transpose(0, 2, 3, 1)
    for err_var, no, flat_no in cases:
        x_hat, no_eff = equ([yt, ht, err_var, no])
        assert x_hat.shape == no_eff.shape == (b, ntx, nst, n_d)
        fn = flat_no if np.isscalar(flat_no) else torch.from_numpy(np.ascontiguousarray(flat_no, np.float32)).cuda()
        xf, nf = cet.lmmse(torch.from_numpy(yd).cuda(), torch.from_numpy(hd).cuda(), fn, direct=True)
        assert np.array_equal(x_hat.cpu().numpy(), back(xf.cpu().numpy()))
        assert np.array_equal(no_eff.cpu().numpy(), back(nf.cpu().numpy()))
        wx, wne = oracle.lmmse(yd, hd, flat_no)
        assert close(x_hat.cpu().numpy(), back(wx)) and close(no_eff.cpu().numpy(), back(wne))
