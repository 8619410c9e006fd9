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


def _prepare(y, H, no, h_share=1):
    """Validates the flat operands; returns (y, H, no tensor, scalar no, noise mode)."""
    if not (torch.is_complex(y) and torch.is_complex(H)):
        raise RuntimeError("y and H must be complex tensors")
    y = y.to(torch.complex64).contiguous()
    H = H.to(torch.complex64).contiguous()
    h_share = int(h_share)
    if (H.dim() != 3 or y.dim() != 2 or h_share < 1 or y.shape[1] != H.shape[1]
            or H.shape[0] != -(-y.shape[0] // h_share)):
        raise RuntimeError(f"expected y [N,Nr] and H [ceil(N/h_share),Nr,Ns], got {tuple(y.shape)}, "
                           f"{tuple(H.shape)}, h_share={h_share}")
    if isinstance(no, torch.Tensor) and no.dim() > 0:
        no_t = no.to(device=y.device, dtype=torch.float32).contiguous()
        if tuple(no_t.shape) == (y.shape[0],):
            mode = 1
        elif tuple(no_t.shape) == tuple(y.shape):
            mode = 2
        else:
            raise RuntimeError(f"`no` must be a scalar, [N] or [N,Nr]; got {tuple(no_t.shape)}")
        no_s = 0.0
    else:
        mode, no_s = 0, float(no)
        no_t = torch.empty(0, dtype=torch.float32, device=y.device)
    return y, H, no_t, no_s, mode


def lmmse(y, H, no, direct=False, h_share=1):
    """y [N, Nr] complex64, H [N, Nr, Ns] complex64 (CUDA); `no`: python float, tensor [N]
    (per resource element) or tensor [N, Nr] (diagonal noise+interference covariance).
    Returns (x_hat [N, Ns] complex64, no_eff [N, Ns] float32).
    `h_share` > 1 (block fading): H is [ceil(N / h_share), Nr, Ns] and resource element n uses
    H[n // h_share], one estimate held over h_share consecutive elements; the results equal
    those of the expanded H bit for bit while HBM reads every matrix once."""
    y, H, no_t, no_s, mode = _prepare(y, H, no, h_share)
    x, ne = torch.ops.cet.lmmse(torch.view_as_real(y), torch.view_as_real(H), no_t, no_s, mode,
                                bool(direct), int(h_share))
    return torch.view_as_complex(x), ne


def lmmse_llr(y, H, no, h_share=1):
    """Equalizer and QPSK demapper in one kernel: same operands as `lmmse`, returns the bit
    log-likelihood ratios [N, Ns, 2] float32, ln(P(b=1)/P(b=0)) with bit 0 on the positive axis
    (what a `Demapper("app", "qam", 2)` makes of `lmmse`'s outputs, HPC/python/EvaluateE2E.py:
    292-319)."""
    y, H, no_t, no_s, mode = _prepare(y, H, no, h_share)
    return torch.ops.cet.lmmse_llr(torch.view_as_real(y), torch.view_as_real(H), no_t, no_s, mode,
                                   int(h_share))


def _host_scalar(v):
    """float(v) for Python / numpy numbers and 0-d CPU tensors, None for anything else."""
    if isinstance(v, (int, float, np.integer, np.floating)):
        return float(v)
    if isinstance(v, np.ndarray) and v.ndim == 0:
        return float(v)
    if torch.is_tensor(v) and v.dim() == 0 and v.device.type == "cpu":
        return float(v)
    return None


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
        self.pilots = self._kronecker_pilots(seed=0) if pil else None

    def _kronecker_pilots(self, seed):
        """Pilots of the 'kronecker' pattern, complex64 [num_tx, num_streams_per_tx,
        num_pilot_symbols, num_effective_subcarriers]: on every pilot OFDM symbol stream k (of
        K = num_tx * num_streams_per_tx) occupies the effective subcarriers k, k + K, ... and is
        silent on the others; the occupied positions carry QPSK symbols scaled so that a
        stream's pilots have unit mean energy over all pilot positions (amplitude sqrt(K)).
        The symbols come from numpy's generator (Sionna draws its own from TensorFlow's);
        `set_pilots` installs other values."""
        K = self.num_tx * self.num_streams_per_tx
        n_p, n_sc = len(self.pilot_ofdm_symbol_indices), self.num_effective_subcarriers
        rng = np.random.default_rng(seed)
        bits = rng.integers(0, 2, (K, n_p, n_sc, 2))
        qpsk = ((1 - 2 * bits[..., 0]) + 1j * (1 - 2 * bits[..., 1])) / np.sqrt(2.0)
        mask = (np.arange(n_sc)[None, :] % K) == np.arange(K)[:, None]
        p = qpsk * mask[:, None, :] * np.sqrt(K)
        return p.reshape(self.num_tx, self.num_streams_per_tx, n_p, n_sc).astype(np.complex64)

    def set_pilots(self, pilots):
        pilots = np.asarray(pilots, dtype=np.complex64)
        want = (self.num_tx, self.num_streams_per_tx, len(self.pilot_ofdm_symbol_indices),
                self.num_effective_subcarriers)
        if pilots.shape != want:
            raise ValueError(f"pilots must have shape {want}")
        self.pilots = pilots


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
        self._dev_tables = {}

    def __call__(self, inputs):
        y, h_hat, err_var, no = inputs
        rg = self.rg
        if y.dim() != 5 or h_hat.dim() != 7 or y.shape[1] != 1 or h_hat.shape[1] != 1:
            raise RuntimeError("expected y [b,1,ant,sym,fft] and h_hat [b,1,ant,tx,streams,sym,sc]")
        if not (torch.is_complex(y) and torch.is_complex(h_hat)):
            raise RuntimeError("y and h_hat must be complex tensors")
        b, _, ant, nsym, fft = y.shape
        ntx, nst, n_sc = h_hat.shape[3], h_hat.shape[4], h_hat.shape[6]
        K = ntx * nst
        dev = y.device
        sym, sc = self._tables(dev)
        if n_sc != sc.numel() or h_hat.shape[5] != nsym or h_hat.shape[2] != ant:
            raise RuntimeError(f"h_hat {tuple(h_hat.shape)} does not match the resource grid")
        # the kernel reads both tensors where they are: [b, ant, sym, fft] and [b, ant, K, sym, sc]
        yv = torch.view_as_real(y.to(torch.complex64).contiguous()[:, 0])
        hv = torch.view_as_real(h_hat.to(torch.complex64).contiguous()[:, 0].reshape(b, ant, K, nsym, n_sc))
        # Scalars given as Python / numpy numbers stay on the host: a device copy read back with
        # float() would synchronise with the GPU on every call.
        no_h, ev_h = _host_scalar(no), _host_scalar(err_var)
        empty = torch.empty(0, dtype=torch.float32, device=dev)
        if ev_h is not None:
            extra, ev = K * ev_h, empty  # the same variance on every stream's estimate
        else:
            ev = torch.as_tensor(err_var, dtype=torch.float32, device=dev)
            if ev.dim() == 0:
                extra, ev = K * ev, empty  # a device scalar: stays there
            else:
                extra = 0.0
                ev = torch.broadcast_to(ev, h_hat.shape)[:, 0].reshape(b, ant, K, nsym, n_sc).contiguous()
        no_t = None if no_h is not None else torch.as_tensor(no, dtype=torch.float32, device=dev)
        if no_h is not None and not torch.is_tensor(extra):
            mode, no_s, no_d = 0, no_h + extra, empty
        elif no_t is None or no_t.dim() == 0:  # a device scalar somewhere: one value per batch entry
            base = no_h if no_t is None else no_t
            mode, no_s, no_d = 1, 0.0, (base + extra).reshape(1).expand(b).contiguous()
        elif no_t.numel() == b:
            mode, no_s, no_d = 1, 0.0, (no_t.reshape(b) + extra).contiguous()
        elif no_t.numel() == b * ant:
            mode, no_s, no_d = 2, 0.0, (no_t.reshape(b, ant) + extra).contiguous()
        else:
            raise RuntimeError(f"`no` must be a scalar, [batch] or [batch, 1, ant]; got {tuple(no_t.shape)}")
        x, ne = torch.ops.cet.lmmse_grid(yv, hv, ev, no_d, no_s, mode, sym, sc)
        n_data = x.shape[2]
        return (torch.view_as_complex(x).reshape(b, ntx, nst, n_data), ne.reshape(b, ntx, nst, n_data))

    def _tables(self, dev):
        """Device copies of the data-symbol and effective-subcarrier index tables (int32)."""
        t = self._dev_tables.get(dev)
        if t is None:
            t = (torch.as_tensor(np.asarray(self.rg.data_symbol_ind), dtype=torch.int32, device=dev),
                 torch.as_tensor(np.asarray(self.rg.effective_subcarrier_ind), dtype=torch.int32, device=dev))
            self._dev_tables[dev] = t
        return t


class LSChannelEstimator:
    """`LSChannelEstimator(rg, interpolation_type="nn")([y, no]) -> (h_hat, err_var)`, the
    estimator in front of the equalizer on the reference's estimated-CSI branch
    (LocalSimulations/EndtoEndtest.py:130,184).

    y      [batch, num_rx=1, num_rx_ant, num_ofdm_symbols, fft_size]                  complex
    h_hat  [batch, 1, num_rx_ant, num_tx, num_streams_per_tx, num_ofdm_symbols, n_eff] complex
    err_var same shape, float32
    At a pilot position of stream k: h = y / p, err_var = no / |p|^2. Every other resource
    element takes the values of the stream's nearest pilot, nearest by |symbol distance| +
    |subcarrier distance| (effective subcarriers), the first one in (symbol, subcarrier) order
    among equals. The geometry is resolved once on the host into a gather table; the kernel
    (csrc/lsest.hip) gathers and scales."""

    def __init__(self, resource_grid, interpolation_type="nn"):
        if interpolation_type != "nn":
            raise NotImplementedError("only nearest-neighbour interpolation is implemented")
        if resource_grid.pilots is None:
            raise ValueError("the resource grid carries no pilots")
        self.rg = resource_grid
        self._tables = self._build_tables()
        self._dev = {}

    def _build_tables(self):
        rg = self.rg
        K = rg.num_tx * rg.num_streams_per_tx
        n_sym, n_sc = rg.num_ofdm_symbols, rg.num_effective_subcarriers
        pil = rg.pilots.reshape(K, len(rg.pilot_ofdm_symbol_indices), n_sc)
        src = np.empty((K, n_sym, n_sc), np.int32)
        pinv = np.empty((K, n_sym, n_sc), np.complex64)
        s_idx = np.arange(n_sym)[:, None, None]
        f_idx = np.arange(n_sc)[None, :, None]
        for k in range(K):
            pp, pf = np.nonzero(pil[k])                       # (pilot symbol, subcarrier) order
            ps = np.asarray(rg.pilot_ofdm_symbol_indices)[pp]
            dist = np.abs(s_idx - ps[None, None, :]) + np.abs(f_idx - pf[None, None, :])
            near = dist.argmin(-1)                             # first minimum
            src[k] = ps[near] * rg.fft_size + rg.effective_subcarrier_ind[pf[near]]
            pinv[k] = 1.0 / pil[k][pp[near], pf[near]]
        return src.reshape(-1), pinv.reshape(-1), (np.abs(pinv) ** 2).astype(np.float32).reshape(-1)

    def __call__(self, inputs):
        y, no = inputs
        rg = self.rg
        if y.dim() != 5 or y.shape[1] != 1 or y.shape[3] != rg.num_ofdm_symbols or y.shape[4] != rg.fft_size:
            raise RuntimeError("expected y [b, 1, ant, num_ofdm_symbols, fft_size]")
        dev = y.device
        if dev not in self._dev:
            src, pinv, p2 = self._tables
            self._dev[dev] = (torch.from_numpy(src).to(dev),
                              torch.view_as_real(torch.from_numpy(pinv)).contiguous().to(dev),
                              torch.from_numpy(p2).to(dev))
        src, pinv, p2 = self._dev[dev]
        b, _, ant = y.shape[:3]
        yr = torch.view_as_real(y.to(torch.complex64).contiguous()).reshape(b * ant, -1, 2)
        h, ev = torch.ops.cet.ls_nn(yr, src, pinv, p2, float(no))
        shape = (b, 1, ant, rg.num_tx, rg.num_streams_per_tx, rg.num_ofdm_symbols,
                 rg.num_effective_subcarriers)
        return torch.view_as_complex(h).reshape(shape), ev.reshape(shape)
