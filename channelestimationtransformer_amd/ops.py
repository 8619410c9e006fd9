"""PyTorch custom ops (`torch.ops.cet.*`) in front of the C ABI.

Each op checks device / dtype / contiguity, allocates the outputs with torch (the library
never allocates results) and enqueues the HIP kernel on torch's current stream through the
ctypes binding. Handles are created and owned by the host-side model objects.
"""
import ctypes

import torch

from . import _cabi

_NO_ARG = torch.empty(0)


def _dev_check(*tensors):
    dev = tensors[0].device
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "channelestimationtransformer_amd computes on the GPU only: got a tensor on "
                f"{t.device}; move inputs to a ROCm device (there is no CPU path)")
        if t.device != dev:
            raise RuntimeError(f"tensors on different devices: {t.device} vs {dev}")
    return dev


def _f32(name, t, *shape):
    """A contiguous float32 tensor of exactly this shape (None = any size in that position):
    the kernels index raw pointers, so a wrong dtype or a short buffer would read out of bounds."""
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    if t.dim() != len(shape) or any(s is not None and int(d) != int(s) for d, s in zip(t.shape, shape)):
        raise RuntimeError(f"{name} must have shape {list(shape)}, got {list(t.shape)}")
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t  # a slice may start anywhere; the C ABI wants 16 B


def _noise(no, no_mode, N, Nr):
    """The noise operand of the per-resource-element kernels for `no_mode`."""
    if no_mode == 0:
        return None
    if no_mode == 1:
        return _f32("no", no, N)
    if no_mode == 2:
        return _f32("no", no, N, Nr)
    raise RuntimeError(f"no_mode must be 0, 1 or 2, got {no_mode}")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None


@torch.library.custom_op("cet::informer_forward", mutates_args=())
def informer_forward(x_enc: torch.Tensor, x_dec: torch.Tensor, handle: int, pred_len: int,
                     c_out: int, attn_floats_per_sample: int) -> tuple[torch.Tensor, torch.Tensor]:
    dev = _dev_check(x_enc, x_dec)
    x_enc = x_enc.contiguous().float()
    x_dec = x_dec.contiguous().float()
    B = x_enc.shape[0]
    out = torch.empty((B, pred_len, c_out), dtype=torch.float32, device=dev)
    attn = torch.empty((B * attn_floats_per_sample,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _cabi.lib().cet_informer_forward(
            ctypes.c_void_p(handle), _ptr(x_enc), _ptr(x_dec), _ptr(out), _ptr(attn), B,
            _cabi.current_stream_ptr(dev))
    _cabi.check(rc, "cet_informer_forward")
    return out, attn


@informer_forward.register_fake
def _(x_enc, x_dec, handle, pred_len, c_out, attn_floats_per_sample):
    B = x_enc.shape[0]
    return (x_enc.new_empty((B, pred_len, c_out)), x_enc.new_empty((B * attn_floats_per_sample,)))


@torch.library.custom_op("cet::rnn_rollout", mutates_args=())
def rnn_rollout(x: torch.Tensor, handle: int, pred_len: int) -> torch.Tensor:
    dev = _dev_check(x)
    x = x.contiguous().float()
    B, seq_len, F = x.shape
    out = torch.empty((B, pred_len, F), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _cabi.lib().cet_rnn_rollout(ctypes.c_void_p(handle), _ptr(x), _ptr(out), B, seq_len,
                                         pred_len, _cabi.current_stream_ptr(dev))
    _cabi.check(rc, "cet_rnn_rollout")
    return out


@rnn_rollout.register_fake
def _(x, handle, pred_len):
    return x.new_empty((x.shape[0], pred_len, x.shape[2]))


@torch.library.custom_op("cet::lmmse", mutates_args=())
def lmmse(y: torch.Tensor, H: torch.Tensor, no: torch.Tensor, no_scalar: float,
          no_mode: int, direct: bool, h_share: int = 1) -> tuple[torch.Tensor, torch.Tensor]:
    """y [N,Nr,2], H [ceil(N/h_share),Nr,Ns,2] float32 (interleaved complex) -> x_hat [N,Ns,2],
    no_eff [N,Ns]. h_share > 1: block fading, RE n uses H[n // h_share]."""
    dev = _dev_check(y, H) if no_mode == 0 else _dev_check(y, H, no)
    y = _f32("y", y, None, None, 2)
    N, Nr = y.shape[0], y.shape[1]
    H = _f32("H", H, None, Nr, None, 2)
    Ns = H.shape[2]
    _block_check(N, H.shape[0], h_share, direct)
    x_hat = torch.empty((N, Ns, 2), dtype=torch.float32, device=dev)
    no_eff = torch.empty((N, Ns), dtype=torch.float32, device=dev)
    no_dev = _noise(no, no_mode, N, Nr)
    L = _cabi.lib()
    with torch.cuda.device(dev):
        if h_share > 1:
            rc = L.cet_lmmse_block(_ptr(y), _ptr(H), no_mode, ctypes.c_float(no_scalar), _ptr(no_dev),
                                   _ptr(x_hat), _ptr(no_eff), ctypes.c_long(N), h_share, Nr, Ns,
                                   _cabi.current_stream_ptr(dev))
        else:
            fn = L.cet_lmmse_direct if direct else L.cet_lmmse
            rc = fn(_ptr(y), _ptr(H), no_mode, ctypes.c_float(no_scalar), _ptr(no_dev), _ptr(x_hat),
                    _ptr(no_eff), ctypes.c_long(N), Nr, Ns, _cabi.current_stream_ptr(dev))
    _cabi.check(rc, "cet_lmmse")
    return x_hat, no_eff


def _block_check(N, n_h, h_share, direct=False):
    if h_share < 1 or n_h != -(-N // h_share):
        raise RuntimeError(f"H holds {n_h} matrices; {N} REs with h_share={h_share} need {-(-N // max(h_share, 1))}")
    if direct and h_share > 1:
        raise RuntimeError("the direct kernel has no block-fading entry point")


@lmmse.register_fake
def _(y, H, no, no_scalar, no_mode, direct, h_share=1):
    N, Ns = y.shape[0], H.shape[2]
    return y.new_empty((N, Ns, 2)), y.new_empty((N, Ns))


@torch.library.custom_op("cet::nmse_sums", mutates_args=())
def nmse_sums(pred: torch.Tensor, truth: torch.Tensor) -> torch.Tensor:
    """pred, truth [B,P,C] float32 -> float64 [2,P]: (sum (truth-pred)^2, sum truth^2) per slot."""
    dev = _dev_check(pred, truth)
    pred = pred.contiguous().float()
    truth = truth.contiguous().float()
    B, P, C = pred.shape
    sums = torch.empty((2, P), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = _cabi.lib().cet_nmse_sums(_ptr(pred), _ptr(truth), ctypes.c_void_p(sums.data_ptr()),
                                       ctypes.c_long(B), P, C, _cabi.current_stream_ptr(dev))
    _cabi.check(rc, "cet_nmse_sums")
    return sums


@nmse_sums.register_fake
def _(pred, truth):
    return pred.new_empty((2, pred.shape[1]), dtype=torch.float64)


@torch.library.custom_op("cet::qpsk_bit_errors", mutates_args=())
def qpsk_bit_errors(x_hat: torch.Tensor, x_true: torch.Tensor) -> torch.Tensor:
    """x_hat, x_true [..., 2] float32 (interleaved complex) -> float64 [2]: (bit errors, bits)."""
    dev = _dev_check(x_hat, x_true)
    x_hat = x_hat.contiguous().float()
    x_true = x_true.contiguous().float()
    x_hat = x_hat.clone() if x_hat.data_ptr() % 8 else x_hat  # (re, im) pairs are read as one piece
    x_true = x_true.clone() if x_true.data_ptr() % 8 else x_true
    N = x_hat.numel() // 2
    out = torch.empty((2,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = _cabi.lib().cet_qpsk_bit_errors(_ptr(x_hat), _ptr(x_true), ctypes.c_void_p(out.data_ptr()),
                                             ctypes.c_long(N), _cabi.current_stream_ptr(dev))
    _cabi.check(rc, "cet_qpsk_bit_errors")
    return out


@qpsk_bit_errors.register_fake
def _(x_hat, x_true):
    return x_hat.new_empty((2,), dtype=torch.float64)


@torch.library.custom_op("cet::lmmse_llr", mutates_args=())
def lmmse_llr(y: torch.Tensor, H: torch.Tensor, no: torch.Tensor, no_scalar: float,
              no_mode: int, h_share: int = 1) -> torch.Tensor:
    """y [N,Nr,2], H [ceil(N/h_share),Nr,Ns,2] float32 -> QPSK bit LLRs [N,Ns,2] (equalizer +
    demapper fused)."""
    dev = _dev_check(y, H) if no_mode == 0 else _dev_check(y, H, no)
    y = _f32("y", y, None, None, 2)
    N, Nr = y.shape[0], y.shape[1]
    H = _f32("H", H, None, Nr, None, 2)
    Ns = H.shape[2]
    _block_check(N, H.shape[0], h_share)
    llr = torch.empty((N, Ns, 2), dtype=torch.float32, device=dev)
    no_dev = _noise(no, no_mode, N, Nr)
    L = _cabi.lib()
    with torch.cuda.device(dev):
        if h_share > 1:
            rc = L.cet_lmmse_block_llr(_ptr(y), _ptr(H), no_mode, ctypes.c_float(no_scalar),
                                       _ptr(no_dev), _ptr(llr), ctypes.c_long(N), h_share, Nr, Ns,
                                       _cabi.current_stream_ptr(dev))
        else:
            rc = L.cet_lmmse_llr(_ptr(y), _ptr(H), no_mode, ctypes.c_float(no_scalar),
                                 _ptr(no_dev), _ptr(llr), ctypes.c_long(N), Nr, Ns,
                                 _cabi.current_stream_ptr(dev))
    _cabi.check(rc, "cet_lmmse_llr")
    return llr


@lmmse_llr.register_fake
def _(y, H, no, no_scalar, no_mode, h_share=1):
    return y.new_empty((y.shape[0], H.shape[2], 2))


@torch.library.custom_op("cet::ls_nn", mutates_args=())
def ls_nn(y: torch.Tensor, src: torch.Tensor, pinv: torch.Tensor, pabs2inv: torch.Tensor,
          no: float) -> tuple[torch.Tensor, torch.Tensor]:
    """y [planes, plane, 2] float32 (interleaved complex), tables of T entries ->
    h_hat [planes, T, 2], err_var [planes, T]."""
    dev = _dev_check(y, src, pinv, pabs2inv)
    y = _f32("y", y, None, None, 2)
    planes, plane = y.shape[0], y.shape[1]
    T = src.numel()
    if src.dtype != torch.int32 or src.dim() != 1:
        raise RuntimeError(f"src must be a 1-d int32 table, got {src.dtype} {list(src.shape)}")
    src = src.contiguous()
    pinv = _f32("pinv", pinv, T, 2)
    pabs2inv = _f32("pabs2inv", pabs2inv, T)
    h_hat = torch.empty((planes, T, 2), dtype=torch.float32, device=dev)
    err_var = torch.empty((planes, T), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _cabi.lib().cet_ls_nn(_ptr(y), _ptr(src), _ptr(pinv), _ptr(pabs2inv), ctypes.c_float(no),
                                   _ptr(h_hat), _ptr(err_var), ctypes.c_long(planes), plane, T,
                                   _cabi.current_stream_ptr(dev))
    _cabi.check(rc, "cet_ls_nn")
    return h_hat, err_var


@ls_nn.register_fake
def _(y, src, pinv, pabs2inv, no):
    return y.new_empty((y.shape[0], src.numel(), 2)), y.new_empty((y.shape[0], src.numel()))


@torch.library.custom_op("cet::lmmse_grid", mutates_args=())
def lmmse_grid(y: torch.Tensor, h_hat: torch.Tensor, err_var: torch.Tensor, no: torch.Tensor,
               no_scalar: float, no_mode: int, data_sym: torch.Tensor,
               eff_sc: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """y [B,Nr,n_sym,fft,2], h_hat [B,Nr,Ns,n_sym,n_sc,2] float32 (interleaved complex), err_var
    empty or [B,Nr,Ns,n_sym,n_sc]; index tables int32 -> x_hat [B,Ns,n_ds*n_sc,2], no_eff
    [B,Ns,n_ds*n_sc]."""
    dev = _dev_check(y, h_hat, data_sym, eff_sc)
    y = _f32("y", y, None, None, None, None, 2)
    B, Nr, n_sym, fft = y.shape[:4]
    h_hat = _f32("h_hat", h_hat, B, Nr, None, n_sym, None, 2)
    Ns, n_sc = h_hat.shape[2], h_hat.shape[4]
    n_ds = data_sym.numel()
    for name, t in (("data_sym", data_sym), ("eff_sc", eff_sc)):
        if t.dtype != torch.int32 or t.dim() != 1:
            raise RuntimeError(f"{name} must be a 1-d int32 table, got {t.dtype} {list(t.shape)}")
    data_sym, eff_sc = data_sym.contiguous(), eff_sc.contiguous()
    if eff_sc.numel() != n_sc or n_sc > fft or n_ds > n_sym:
        raise RuntimeError(f"inconsistent shapes: y {tuple(y.shape)}, h_hat {tuple(h_hat.shape)}, "
                           f"{n_ds} data symbols, {eff_sc.numel()} effective subcarriers")
    ev = _f32("err_var", err_var, B, Nr, Ns, n_sym, n_sc) if err_var.numel() else None
    if ev is not None:
        _dev_check(y, ev)
    x_hat = torch.empty((B, Ns, n_ds * n_sc, 2), dtype=torch.float32, device=dev)
    no_eff = torch.empty((B, Ns, n_ds * n_sc), dtype=torch.float32, device=dev)
    if no_mode not in (0, 1, 2):
        raise RuntimeError(f"no_mode must be 0, 1 or 2, got {no_mode}")
    no_dev = None
    if no_mode != 0:
        _dev_check(y, no)
        no_dev = _f32("no", no, B) if no_mode == 1 else _f32("no", no, B, Nr)
    with torch.cuda.device(dev):
        rc = _cabi.lib().cet_lmmse_grid(_ptr(y), _ptr(h_hat), _ptr(ev), no_mode, ctypes.c_float(no_scalar),
                                        _ptr(no_dev), _ptr(data_sym), _ptr(eff_sc), _ptr(x_hat),
                                        _ptr(no_eff), B, Nr, Ns, n_sym, fft, n_sc, n_ds,
                                        _cabi.current_stream_ptr(dev))
    _cabi.check(rc, "cet_lmmse_grid")
    return x_hat, no_eff


@lmmse_grid.register_fake
def _(y, h_hat, err_var, no, no_scalar, no_mode, data_sym, eff_sc):
    B, Ns, n = y.shape[0], h_hat.shape[2], data_sym.numel() * h_hat.shape[4]
    return y.new_empty((B, Ns, n, 2)), y.new_empty((B, Ns, n))
