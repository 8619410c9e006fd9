"""ctypes front end of oracle/libcet_oracle.so plus a float64 numpy LMMSE.

TEST INFRASTRUCTURE ONLY (see oracle/cet_oracle.c). Functions take and return numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcet_oracle.so")
_lib = None


def build():
    """Compiles the C restatement with gcc (seconds)."""
    subprocess.run(["make", "-C", _HERE, "libcet_oracle.so"], check=True,
                   stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_informer_attn_size.restype = ctypes.c_size_t
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def informer_forward(cfg, blob, x_enc, x_dec, want_attn=False, want_taps=False):
    """-> (pred [B,pred_len,c_out], attn list or None, taps list or None)."""
    L = lib()
    blob, x_enc, x_dec = _f32(blob), _f32(x_enc), _f32(x_dec)
    B = x_enc.shape[0]
    out = np.empty((B, cfg.pred_len, cfg.c_out), np.float32)
    attn = taps = None
    rows = (ctypes.c_int * 128)()
    n_st = L.orc_informer_tap_rows(ctypes.byref(cfg), rows, 128)
    if want_attn:
        attn = np.empty(L.orc_informer_attn_size(ctypes.byref(cfg), B), np.float32)
    if want_taps:
        taps = np.empty(sum(rows[i] for i in range(n_st)) * B * cfg.d_model, np.float32)
    rc = L.orc_informer_forward(ctypes.byref(cfg), _p(blob), _p(x_enc), _p(x_dec), _p(out),
                                _p(attn), _p(taps), B)
    assert rc == 0
    if want_taps:
        taps = split_taps(taps, [rows[i] for i in range(n_st)], B, cfg.d_model)
    if want_attn:
        attn = split_attn(attn, attn_lengths(cfg), B, cfg.n_heads)
    return out, attn, taps


def tap_rows(cfg):
    rows = (ctypes.c_int * 128)()
    n = lib().orc_informer_tap_rows(ctypes.byref(cfg), rows, 128)
    return [rows[i] for i in range(n)]


def attn_lengths(cfg):
    """Sequence length of each encoder self-attention map, in execution order."""
    lens, inp_len, cur = [], cfg.seq_len, cfg.seq_len
    depths = list(range(cfg.e_layers, 2, -1)) if cfg.stack else [cfg.e_layers]
    for el in depths:
        L = min(inp_len, cur)
        for l in range(el):
            lens.append(L)
            if cfg.distil and l < el - 1:
                L = (L + 1) // 2 + 1
        cur, inp_len = L, inp_len // 2
    return lens


def split_taps(flat, rows, B, D):
    out, off = [], 0
    for r in rows:
        n = B * r * D
        out.append(flat[off:off + n].reshape(B, r, D))
        off += n
    assert off == flat.size
    return out


def split_attn(flat, lens, B, H):
    out, off = [], 0
    for L in lens:
        n = B * H * L * L
        out.append(flat[off:off + n].reshape(B, H, L, L))
        off += n
    assert off == flat.size
    return out


def rnn_rollout(kind, features, input_size, hidden, layers, blob, x, pred_len):
    blob, x = _f32(blob), _f32(x)
    B, seq_len, F = x.shape
    assert F == features
    out = np.empty((B, pred_len, F), np.float32)
    rc = lib().orc_rnn_rollout(kind, features, input_size, hidden, layers, _p(blob), _p(x),
                               _p(out), B, seq_len, pred_len)
    assert rc == 0
    return out


def lmmse(y, H, no):
    """C float64 restatement. y [N,Nr] complex, H [N,Nr,Ns] complex, `no` scalar, [N] or
    [N,Nr]  ->  (x_hat [N,Ns] complex64, no_eff [N,Ns] float32)."""
    y = np.ascontiguousarray(y, dtype=np.complex64)
    H = np.ascontiguousarray(H, dtype=np.complex64)
    N, Nr, Ns = H.shape
    no = np.asarray(no, dtype=np.float32)
    mode = {0: 0, 1: 1, 2: 2}[no.ndim]
    no_arr = np.ascontiguousarray(no) if mode else None
    x = np.empty((N, Ns), np.complex64)
    ne = np.empty((N, Ns), np.float32)
    rc = lib().orc_lmmse(_p(y), _p(H), mode, ctypes.c_float(float(no) if mode == 0 else 0.0),
                         _p(no_arr), _p(x), _p(ne), ctypes.c_long(N), Nr, Ns)
    assert rc == 0
    return x, ne


def lmmse_numpy(y, H, no):
    """Independent float64 statement of the same definition with numpy.linalg:
    G = (H^H S^-1 H + I)^-1 H^H S^-1, x_hat = (G y)/diag(GH), no_eff = Re(1/diag(GH) - 1)."""
    y = np.asarray(y, dtype=np.complex128)
    H = np.asarray(H, dtype=np.complex128)
    N, Nr, Ns = H.shape
    s = np.asarray(no, dtype=np.float64)
    if s.ndim == 1:
        s = s[:, None]
    s = np.broadcast_to(s, (N, Nr))
    Hh_Sinv = np.conj(np.swapaxes(H, 1, 2)) / s[:, None, :]
    A = Hh_Sinv @ H + np.eye(Ns)[None]
    G = np.linalg.solve(A, Hh_Sinv)
    GH = G @ H
    d = np.einsum("nii->ni", GH)
    x = (G @ y[..., None])[..., 0] / d
    return x, np.real(1.0 / d - 1.0)


def num_threads():
    return lib().orc_num_threads()


def ls_nn_numpy(y, pilots, pilot_symbols, eff_sc, no):
    """Brute-force float64 statement of the LS channel estimate with nearest-neighbour
    interpolation (the checker of csrc/lsest.hip; no reference vector exists for it: Sionna, whose
    `LSChannelEstimator(rg, interpolation_type="nn")` it stands in for at
    LocalSimulations/EndtoEndtest.py:130,184, is absent -> parity unpinned).
    y [b, ant, n_sym, fft] complex; pilots [K, n_pilot_sym, n_eff] complex (0 = no pilot);
    pilot_symbols [n_pilot_sym] OFDM symbol indices; eff_sc [n_eff] fft indices.
    -> h_hat [b, ant, K, n_sym, n_eff] complex128, err_var same shape float64."""
    y = np.asarray(y, np.complex128)
    b, ant, n_sym, _ = y.shape
    K, _, n_eff = pilots.shape
    h = np.zeros((b, ant, K, n_sym, n_eff), np.complex128)
    ev = np.zeros((b, ant, K, n_sym, n_eff))
    for k in range(K):
        cand = [(ip, f) for ip in range(len(pilot_symbols)) for f in range(n_eff) if pilots[k, ip, f] != 0]
        for s in range(n_sym):
            for f in range(n_eff):
                best, bd = None, None
                for ip, pf in cand:
                    d = abs(s - int(pilot_symbols[ip])) + abs(f - pf)
                    if bd is None or d < bd:
                        best, bd = (ip, pf), d
                ip, pf = best
                p = complex(pilots[k, ip, pf])
                h[:, :, k, s, f] = y[:, :, int(pilot_symbols[ip]), int(eff_sc[pf])] / p
                ev[:, :, k, s, f] = no / abs(p) ** 2
    return h, ev
