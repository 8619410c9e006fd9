"""Edges of the LMMSE entry points' buffers: nothing is written behind the N * Ns results (called
through the C ABI on exactly sized sub-buffers), misaligned pointers are refused, and element /
byte offsets beyond 2^31 and 2^32 index correctly (PARITY UNPINNED like every LMMSE test: the
checker is the float64 restatement, see oracle/cet_oracle.c)."""
import ctypes

import numpy as np
import pytest
import torch

import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import _cabi

pytestmark = pytest.mark.gpu

RTOL = 2e-4
GUARD = 64  # floats in front of and behind every output view


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _guarded(n):
    """A NaN-filled buffer and its exactly sized, 16-byte aligned view of n floats in the middle."""
    buf = torch.full((n + 2 * GUARD,), float("nan"), device="cuda")
    return buf, buf[GUARD:GUARD + n]


def _guards_untouched(buf, n):
    return bool(torch.isnan(buf[:GUARD]).all()) and bool(torch.isnan(buf[GUARD + n:]).all())


@pytest.mark.parametrize("Nr,Ns", [(2, 1), (4, 1), (16, 1), (4, 2), (16, 2), (2, 2), (8, 4), (16, 4)])
@pytest.mark.parametrize("N,h_share", [(4195, 14), (4197, 14), (611, 1), (613, 3), (64 * 14 + 1, 14), (5, 7)])
def test_no_byte_behind_the_results(Nr, Ns, N, h_share):
    """Every LMMSE entry point, called through the C ABI with x_hat / no_eff / llr as views into
    NaN-filled buffers: odd N and N % 4 != 0 make the end of the results fall inside a 16-byte
    piece for one and two streams. The bytes around the views stay NaN; the views hold what the
    torch-allocated route returns."""
    L = _cabi.lib()
    g = torch.Generator(device="cuda").manual_seed(N * 31 + Nr * 4 + Ns)
    nb = -(-N // h_share)
    y = torch.randn(N, Nr, 2, device="cuda", generator=g)
    Hb = torch.randn(nb, Nr, Ns, 2, device="cuda", generator=g)
    H = Hb.repeat_interleave(h_share, dim=0)[:N].contiguous()
    stream = _cabi.current_stream_ptr(torch.device("cuda"))
    want_x, want_ne = cet.lmmse(torch.view_as_complex(y), torch.view_as_complex(H), 0.07)
    want_llr = cet.lmmse_llr(torch.view_as_complex(y), torch.view_as_complex(H), 0.07)
    nx, nn = N * Ns * 2, N * Ns
    no = ctypes.c_float(0.07)
    calls = [
        ("cet_lmmse", lambda x, n: L.cet_lmmse(_ptr(y), _ptr(H), 0, no, None, _ptr(x), _ptr(n), N, Nr, Ns, stream), False),
        ("cet_lmmse_direct", lambda x, n: L.cet_lmmse_direct(_ptr(y), _ptr(H), 0, no, None, _ptr(x), _ptr(n), N, Nr, Ns, stream), False),
        ("cet_lmmse_block", lambda x, n: L.cet_lmmse_block(_ptr(y), _ptr(Hb), 0, no, None, _ptr(x), _ptr(n), N, h_share, Nr, Ns, stream), False),
        ("cet_lmmse_llr", lambda x, n: L.cet_lmmse_llr(_ptr(y), _ptr(H), 0, no, None, _ptr(x), N, Nr, Ns, stream), True),
        ("cet_lmmse_block_llr", lambda x, n: L.cet_lmmse_block_llr(_ptr(y), _ptr(Hb), 0, no, None, _ptr(x), N, h_share, Nr, Ns, stream), True),
    ]
    for name, call, llr in calls:
        xbuf, xv = _guarded(nx)
        nbuf, nv = _guarded(nn)
        assert call(xv, nv) == 0, name
        torch.cuda.synchronize()
        assert _guards_untouched(xbuf, nx), f"{name} wrote outside x_hat / llr"
        assert _guards_untouched(nbuf, nn), f"{name} wrote outside no_eff"
        if llr:
            assert torch.isnan(nv).all(), f"{name} must not write no_eff"
            assert torch.equal(xv.view(N, Ns, 2), want_llr) or name == "cet_lmmse_direct"
        elif name == "cet_lmmse_direct":  # other summation order than the tiled kernels
            assert torch.allclose(xv.view(N, Ns, 2), torch.view_as_real(want_x), rtol=1e-3, atol=1e-4)
        else:
            assert torch.equal(xv.view(N, Ns, 2), torch.view_as_real(want_x)), name
            assert torch.equal(nv.view(N, Ns), want_ne), name


def test_misaligned_pointers_are_refused():
    L = _cabi.lib()
    N, Nr, Ns = 100, 4, 2
    y = torch.randn(N * Nr * 2 + 4, device="cuda")
    H = torch.randn(N * Nr * Ns * 2 + 4, device="cuda")
    x = torch.empty(N * Ns * 2 + 4, device="cuda")
    ne = torch.empty(N * Ns + 4, device="cuda")
    stream = _cabi.current_stream_ptr(torch.device("cuda"))
    ok = L.cet_lmmse(_ptr(y), _ptr(H), 0, ctypes.c_float(0.1), None, _ptr(x), _ptr(ne), N, Nr, Ns, stream)
    assert ok == 0
    for bad in range(4):
        args = [y, H, x, ne]
        args[bad] = args[bad][1:]  # 4 bytes off
        rc = L.cet_lmmse(_ptr(args[0]), _ptr(args[1]), 0, ctypes.c_float(0.1), None, _ptr(args[2]),
                         _ptr(args[3]), N, Nr, Ns, stream)
        assert rc == 1  # CET_ERR_INVALID_ARG
    # complex arrays of the other entry points: 8 bytes
    xc = torch.randn(2 * N + 2, device="cuda")
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    assert L.cet_qpsk_bit_errors(_ptr(xc), _ptr(xc), _ptr(out), N, stream) == 0
    assert L.cet_qpsk_bit_errors(_ptr(xc[1:]), _ptr(xc), _ptr(out), N, stream) == 1
    assert L.cet_qpsk_bit_errors(_ptr(xc), _ptr(xc[1:]), _ptr(out), N, stream) == 1
    # the torch-level op copies a misaligned slice instead of failing
    yc = torch.view_as_complex(torch.randn(N + 1, 2, 2, device="cuda"))[1:]  # 16 B rows: stays aligned
    assert yc.data_ptr() % 16 == 0
    no = (torch.rand(N + 1, device="cuda") + 0.1)[1:]  # starts 4 bytes into its allocation
    assert no.data_ptr() % 16 != 0
    Hc = torch.view_as_complex(torch.randn(N, 2, 2, 2, device="cuda"))
    x1, n1 = cet.lmmse(yc, Hc, no)
    x2, n2 = cet.lmmse(yc, Hc, no.clone())
    assert torch.equal(torch.view_as_real(x1), torch.view_as_real(x2)) and torch.equal(n1, n2)


def _windows(N, marks, half=2500, align=1):
    out = [((N - 2 * half) // align * align, N)]
    for m in marks:
        lo = max(0, (m - half) // align * align)
        out.append((lo, min(N, lo + 2 * half)))
    return out


def test_offsets_beyond_2_31_per_re_kernels(oracle):
    """2^24 + 4099 REs at 16x4: H is 8.6 GB (element offsets to 2^31 + 524k, byte offsets beyond
    2^33), y 2.1 GB. Windows at the end and around element offset 2^31 / byte offset 2^32 of H
    must equal, bit for bit, the same entry point called on the window alone (whose offsets are
    small), the independent one-RE-per-thread kernel within rounding, and the float64 checker;
    the two kernels agree over all REs."""
    Nr, Ns = 16, 4
    N = (1 << 24) + 4099
    g = torch.Generator(device="cuda").manual_seed(3)
    H = torch.empty(N, Nr, Ns, 2, device="cuda")
    step = 1 << 21
    for lo in range(0, N, step):  # in pieces: one randn of 2.1 G elements needs 64-bit indexing inside torch
        H[lo:lo + step].normal_(0.0, 0.5 ** 0.5, generator=g)
    y = torch.randn(N, Nr, 2, device="cuda", generator=g)
    Hc, yc = torch.view_as_complex(H), torch.view_as_complex(y)
    assert H.numel() > 2 ** 31
    x1, ne1 = cet.lmmse(yc, Hc, 0.05)
    llr = cet.lmmse_llr(yc, Hc, 0.05)
    x2, ne2 = cet.lmmse(yc, Hc, 0.05, direct=True)
    assert bool(torch.isfinite(torch.view_as_real(x1)).all())
    scale = float(x1.abs().max())
    assert float((x1 - x2).abs().max()) < 1e-3 * scale  # 17 M random matrices: some poorly conditioned
    assert float(((ne1 - ne2).abs() / ne2).max()) < 1e-3
    per_re_H = Nr * Ns * 2
    for lo, hi in _windows(N, [2 ** 31 // per_re_H, 2 ** 32 // (per_re_H * 4), 2 ** 33 // (per_re_H * 4)]):
        ys, Hs = yc[lo:hi].clone(), Hc[lo:hi].clone()
        xs, nes = cet.lmmse(ys, Hs, 0.05)
        assert torch.equal(torch.view_as_real(xs), torch.view_as_real(x1[lo:hi])) and torch.equal(nes, ne1[lo:hi])
        assert torch.equal(cet.lmmse_llr(ys, Hs, 0.05), llr[lo:hi])
        xd, ned = cet.lmmse(ys, Hs, 0.05, direct=True)
        assert torch.equal(torch.view_as_real(xd), torch.view_as_real(x2[lo:hi])) and torch.equal(ned, ne2[lo:hi])
        wx, wne = oracle.lmmse(ys.cpu().numpy(), Hs.cpu().numpy(), 0.05)
        gx, gne = xs.cpu().numpy(), nes.cpu().numpy()
        assert np.abs(gx - wx).max() <= RTOL * max(1.0, np.abs(wx).max())
        assert np.abs(gne - wne).max() <= RTOL * max(1.0, np.abs(wne).max())


def test_offsets_beyond_2_31_block_fading(oracle):
    """2^26 + 4099 REs at 16x4 with one matrix per 14 REs: y is 8.6 GB (element offsets beyond
    2^31), x_hat 2.1 GB (byte offsets beyond 2^31). Windows at the end and around those marks
    equal, bit for bit, the per-RE entry point on the window's expanded channel, demapper
    included, and the float64 checker."""
    Nr, Ns, hs = 16, 4, 14
    N = (1 << 26) + 4099
    nb = -(-N // hs)
    g = torch.Generator(device="cuda").manual_seed(4)
    y = torch.empty(N, Nr, 2, device="cuda")
    step = 1 << 24
    for lo in range(0, N, step):
        y[lo:lo + step].normal_(0.0, 1.0, generator=g)
    assert y.numel() > 2 ** 31
    Hb = torch.randn(nb, Nr, Ns, 2, device="cuda", generator=g) * 0.5 ** 0.5
    yc, Hbc = torch.view_as_complex(y), torch.view_as_complex(Hb)
    x1, ne1 = cet.lmmse(yc, Hbc, 0.05, h_share=hs)
    llr = cet.lmmse_llr(yc, Hbc, 0.05, h_share=hs)
    per_re_y = Nr * 2
    marks = [2 ** 31 // per_re_y, 2 ** 32 // (per_re_y * 4), 2 ** 33 // (per_re_y * 4), 2 ** 31 // (Ns * 8)]
    for lo, hi in _windows(N, marks, align=hs):
        ys = yc[lo:hi].clone()
        Hs = Hbc[lo // hs:-(-hi // hs)].repeat_interleave(hs, dim=0)[:hi - lo].contiguous()
        xs, nes = cet.lmmse(ys, Hs, 0.05)
        assert torch.equal(torch.view_as_real(xs), torch.view_as_real(x1[lo:hi])) and torch.equal(nes, ne1[lo:hi])
        assert torch.equal(cet.lmmse_llr(ys, Hs, 0.05), llr[lo:hi])
        wx, wne = oracle.lmmse(ys.cpu().numpy(), Hs.cpu().numpy(), 0.05)
        assert np.abs(xs.cpu().numpy() - wx).max() <= RTOL * max(1.0, np.abs(wx).max())
        assert np.abs(nes.cpu().numpy() - wne).max() <= RTOL * max(1.0, np.abs(wne).max())


def test_offsets_beyond_2_31_grid_entry(oracle):
    """The equalizer object on 18,800 slots of a 16-antenna, 2 x 2-stream grid: h_hat holds
    2.16 G floats. The slots around element offset 2^31 and the last ones equal, bit for bit, the
    object called on those slots alone; one of them is checked against the float64 checker."""
    ant, ntx, nst = 16, 2, 2
    K = ntx * nst
    rg = cet.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=15e3, num_tx=ntx,
                          num_streams_per_tx=nst, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                          dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    equ = cet.LMMSEEqualizer(rg, cet.StreamManagement(np.ones((1, ntx), int), nst))
    n_sc = rg.num_effective_subcarriers
    b = 18800
    per_slot = ant * K * 14 * n_sc * 2
    assert b * per_slot > 2 ** 31
    g = torch.Generator(device="cuda").manual_seed(5)
    h = torch.empty(b, 1, ant, ntx, nst, 14, n_sc, 2, device="cuda")
    for lo in range(0, b, 2048):
        h[lo:lo + 2048].normal_(0.0, 0.5 ** 0.5, generator=g)
    y = torch.randn(b, 1, ant, 14, 76, 2, device="cuda", generator=g)
    hc, yc = torch.view_as_complex(h), torch.view_as_complex(y)
    ev = 0.01
    x_hat, no_eff = equ([yc, hc, ev, 0.1])
    cross = 2 ** 31 // per_slot
    for lo, hi in ((cross - 3, cross + 4), (b - 5, b), (2 ** 32 // (per_slot * 4) - 2, 2 ** 32 // (per_slot * 4) + 3)):
        xs, ns = equ([yc[lo:hi].clone(), hc[lo:hi].clone(), ev, 0.1])
        assert torch.equal(torch.view_as_real(xs), torch.view_as_real(x_hat[lo:hi]))
        assert torch.equal(ns, no_eff[lo:hi])
    sym, sc = np.asarray(rg.data_symbol_ind), np.asarray(rg.effective_subcarrier_ind)
    s = cross  # the slot whose estimates straddle element 2^31
    yn, hn = yc[s:s + 1].cpu().numpy(), hc[s:s + 1].cpu().numpy()
    yd = yn[:, 0][:, :, sym][:, :, :, sc].transpose(0, 2, 3, 1).reshape(-1, ant)
    hd = hn[:, 0].reshape(1, ant, K, 14, n_sc)[:, :, :, sym].transpose(0, 3, 4, 1, 2).reshape(-1, ant, K)
    wx, wne = oracle.lmmse(yd, hd, np.full((yd.shape[0], ant), 0.1 + K * ev, np.float32))
    n_d = len(sym) * n_sc
    back = lambda a: a.reshape(1, n_d, ntx, nst).transpose(0, 2, 3, 1)
    gx, gne = x_hat[s:s + 1].cpu().numpy(), no_eff[s:s + 1].cpu().numpy()
    assert np.abs(gx - back(wx)).max() <= RTOL * max(1.0, np.abs(wx).max())
    assert np.abs(gne - back(wne)).max() <= RTOL * max(1.0, np.abs(wne).max())
