"""LMMSE MIMO equalization on the GPU.

`lmmse(y, H, no)` is the flat per-resource-element form. `LMMSEEqualizer(resource_grid,
stream_management)([y, h_hat, err_var, no])` mirrors the object the reference builds from
Sionna (LocalSimulations/EndtoEndtest.py:133,191) for the single-receiver OFDM set-up it
uses; `ResourceGrid` / `StreamManagement` are small stand-ins that carry the geometry that
call needs (Sionna itself is not a dependency).
"""
import numpy as np
import torch

from . import ops  # noqa: F401


def _prepare(y, H, no):
    """Validates the flat operands; returns (y, H, no tensor, scalar no, noise mode)."""
    if not (torch.is_complex(y) and torch.is_complex(H)):
        raise RuntimeError("y and H must be complex tensors")
    y = y.to(torch.complex64).contiguous()
    H = H.to(torch.complex64).contiguous()
    if H.dim() != 3 or y.dim() != 2 or y.shape != H.shape[:2]:
        raise RuntimeError(f"expected y [N,Nr] and H [N,Nr,Ns], got {tuple(y.shape)}, {tuple(H.shape)}")
    if isinstance(no, torch.Tensor) and no.dim() > 0:
        no_t = no.to(device=y.device, dtype=torch.float32).contiguous()
        if tuple(no_t.shape) == (H.shape[0],):
            mode = 1
        elif tuple(no_t.shape) == tuple(H.shape[:2]):
            mode = 2
        else:
            raise RuntimeError(f"`no` must be a scalar, [N] or [N,Nr]; got {tuple(no_t.shape)}")
        no_s = 0.0
    else:
        mode, no_s = 0, float(no)
        no_t = torch.empty(0, dtype=torch.float32, device=y.device)
    return y, H, no_t, no_s, mode


def lmmse(y, H, no, direct=False):
    """y [N, Nr] complex64, H [N, Nr, Ns] complex64 (CUDA); `no`: python float, tensor [N]
    (per resource element) or tensor [N, Nr] (diagonal noise+interference covariance).
    Returns (x_hat [N, Ns] complex64, no_eff [N, Ns] float32)."""
    y, H, no_t, no_s, mode = _prepare(y, H, no)
    x, ne = torch.ops.cet.lmmse(torch.view_as_real(y), torch.view_as_real(H), no_t, no_s, mode,
                                bool(direct))
    return torch.view_as_complex(x), ne


def lmmse_llr(y, H, no):
    """Equalizer and QPSK demapper in one kernel: same operands as `lmmse`, returns the bit
    log-likelihood ratios [N, Ns, 2] float32, ln(P(b=1)/P(b=0)) with bit 0 on the positive axis
    (what a `Demapper("app", "qam", 2)` makes of `lmmse`'s outputs, HPC/python/EvaluateE2E.py:
    292-319)."""
    y, H, no_t, no_s, mode = _prepare(y, H, no)
    return torch.ops.cet.lmmse_llr(torch.view_as_real(y), torch.view_as_real(H), no_t, no_s, mode)


class StreamManagement:
    """Which streams go to which receiver. Carries `rx_tx_association` and
    `num_streams_per_tx` like the reference's `StreamManagement(np.array([[1]]), n)`
    (LocalSimulations/EndtoEndtest.py:41-47)."""

    def __init__(self, rx_tx_association, num_streams_per_tx):
        self.rx_tx_association = np.asarray(rx_tx_association)
        self.num_streams_per_tx = int(num_streams_per_tx)
        self.num_rx, self.num_tx = self.rx_tx_association.shape
        if self.num_rx != 1 or not np.all(self.rx_tx_association == 1):
            raise NotImplementedError("only one receiver that detects every transmitter's "
                                      "streams is supported (the reference's set-up)")


class ResourceGrid:
    """OFDM grid geometry, same constructor arguments as the reference's call
    (LocalSimulations/EndtoEndtest.py:50-59). Only the 'kronecker' pilot pattern (whole OFDM
    symbols reserved for pilots) is implemented."""

    def __init__(self, num_ofdm_symbols, fft_size, subcarrier_spacing, num_tx=1,
                 num_streams_per_tx=1, cyclic_prefix_length=0, num_guard_carriers=(0, 0),
                 dc_null=False, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=None):
        if pilot_pattern not in ("kronecker", None, "empty"):
            raise NotImplementedError("pilot_pattern must be 'kronecker' or 'empty'")
        self.num_ofdm_symbols = int(num_ofdm_symbols)
        self.fft_size = int(fft_size)
        self.subcarrier_spacing = float(subcarrier_spacing)
        self.num_tx = int(num_tx)
        self.num_streams_per_tx = int(num_streams_per_tx)
        self.cyclic_prefix_length = int(cyclic_prefix_length)
        self.num_guard_carriers = tuple(int(g) for g in num_guard_carriers)
        self.dc_null = bool(dc_null)
        pil = [] if pilot_pattern != "kronecker" else list(pilot_ofdm_symbol_indices or [])
        self.pilot_ofdm_symbol_indices = pil
        lo, hi = self.num_guard_carriers
        sc = np.arange(lo, self.fft_size - hi)
        if self.dc_null:
            # the DC carrier sits in the middle of the (fft-shifted) grid
            sc = sc[sc != self.fft_size // 2]
        self.effective_subcarrier_ind = sc
        self.num_effective_subcarriers = len(sc)
        self.data_symbol_ind = np.array([s for s in range(self.num_ofdm_symbols) if s not in pil])
        self.num_data_symbols = len(self.data_symbol_ind) * self.num_effective_subcarriers
        self.bandwidth = self.fft_size * self.subcarrier_spacing


class LMMSEEqualizer:
    """`LMMSEEqualizer(rg, sm)([y, h_hat, err_var, no]) -> (x_hat, no_eff)`.

    y      [batch, num_rx=1, num_rx_ant, num_ofdm_symbols, fft_size]                complex
    h_hat  [batch, 1, num_rx_ant, num_tx, num_streams_per_tx, num_ofdm_symbols, n_eff] complex
    err_var broadcastable to h_hat (0. for perfect CSI), no: scalar or [batch] / [batch,1,ant]
    x_hat, no_eff  [batch, num_tx, num_streams_per_tx, num_data_symbols]
    Data symbols are ordered OFDM symbol first, then subcarrier; pilot symbols are dropped.
    """

    def __init__(self, resource_grid, stream_management, whiten_interference=True):
        self.rg, self.sm = resource_grid, stream_management

    def __call__(self, inputs):
        y, h_hat, err_var, no = inputs
        rg = self.rg
        if y.dim() != 5 or h_hat.dim() != 7 or y.shape[1] != 1:
            raise RuntimeError("expected y [b,1,ant,sym,fft] and h_hat [b,1,ant,tx,streams,sym,sc]")
        b, _, ant, nsym, _ = y.shape
        ntx, nst = h_hat.shape[3], h_hat.shape[4]
        dev = y.device
        sc = torch.as_tensor(rg.effective_subcarrier_ind, device=dev)
        sym = torch.as_tensor(rg.data_symbol_ind, device=dev)
        # [b, ant, sym_d, sc] -> [b, sym_d, sc, ant]
        yd = y[:, 0].index_select(3, sc).index_select(2, sym).permute(0, 2, 3, 1)
        # [b, ant, tx, st, sym_d, sc] -> [b, sym_d, sc, ant, tx*st]
        hd = h_hat[:, 0].index_select(4, sym).permute(0, 4, 5, 1, 2, 3)
        hd = hd.reshape(b, len(sym), len(sc), ant, ntx * nst)
        N = b * len(sym) * len(sc)
        no_t = torch.as_tensor(no, dtype=torch.float32, device=dev)
        ev = torch.as_tensor(err_var, dtype=torch.float32, device=dev)
        if ev.dim() == 0 and float(ev) == 0.0 and no_t.dim() == 0:
            noise = float(no_t)
        else:
            # S = no*I + diag(sum over streams of the estimation-error variance)
            if ev.dim() > 0:
                ev = torch.broadcast_to(ev, h_hat.shape)[:, 0].index_select(4, sym)
                ev = ev.permute(0, 4, 5, 1, 2, 3).reshape(b, len(sym), len(sc), ant, ntx * nst).sum(-1)
            else:
                ev = ev * (ntx * nst)  # the same variance on every stream's estimate
            nb = no_t.reshape(b, 1, 1, -1) if no_t.dim() > 0 else no_t
            noise = torch.broadcast_to(ev + nb, (b, len(sym), len(sc), ant)).reshape(N, ant).contiguous()
        x, ne = lmmse(yd.reshape(N, ant), hd.reshape(N, ant, ntx * nst), noise)
        # [b, sym_d, sc, tx, st] -> [b, tx, st, sym_d*sc]
        x = x.reshape(b, len(sym) * len(sc), ntx, nst).permute(0, 2, 3, 1).contiguous()
        ne = ne.reshape(b, len(sym) * len(sc), ntx, nst).permute(0, 2, 3, 1).contiguous()
        return x, ne
