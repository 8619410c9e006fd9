"""ctypes binding of the C ABI declared in include/cet.h (library: csrc/libcet.so).

There is deliberately no fallback: if the HIP library is missing or a call fails, the
caller gets an exception. The product path never computes on the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CET_LIB selects another build of the same ABI (the profiling build of tools/)
LIB_PATH = os.environ.get("CET_LIB") or os.path.join(_HERE, "csrc", "libcet.so")

ERR_NAMES = {0: "CET_OK", 1: "CET_ERR_INVALID_ARG", 2: "CET_ERR_UNSUPPORTED", 3: "CET_ERR_HIP",
             4: "CET_ERR_ALLOC", 5: "CET_ERR_BLOB_SIZE", 6: "CET_ERR_TOO_LARGE"}

# Every symbol include/cet.h declares; tests check that the library exports all of them.
EXPORTS = (
    "cet_strerror", "cet_last_hip_error", "cet_abi_version", "cet_destroy",
    "cet_informer_blob_size", "cet_informer_create", "cet_informer_attn_size",
    "cet_informer_forward", "cet_informer_select", "cet_informer_taps_size",
    "cet_informer_forward_taps",
    "cet_rnn_blob_size", "cet_rnn_create", "cet_rnn_rollout", "cet_rnn_select", "cet_rnn_rollout_static", "cet_lmmse", "cet_lmmse_direct",
    "cet_lmmse_llr", "cet_lmmse_block", "cet_lmmse_block_llr", "cet_lmmse_grid", "cet_ls_nn", "cet_nmse_sums", "cet_qpsk_bit_errors",
)


class CetError(RuntimeError):
    pass


class InformerCfg(ctypes.Structure):
    """Mirror of `cet_informer_cfg`."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "enc_in", "dec_in", "c_out", "seq_len", "label_len", "pred_len", "d_model", "n_heads",
        "e_layers", "d_layers", "d_ff", "stack", "distil", "activation")]


_lib = None


def lib():
    """Loads libcet.so once and declares the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CetError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or "
            f"`make -C channelestimationtransformer_amd/csrc`). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, fp = ctypes.c_void_p, ctypes.c_void_p  # device pointers travel as integers
    L.cet_strerror.restype = ctypes.c_char_p
    L.cet_strerror.argtypes = [ctypes.c_int]
    L.cet_last_hip_error.restype = ctypes.c_int
    L.cet_abi_version.restype = ctypes.c_int
    L.cet_destroy.argtypes = [vp]
    L.cet_informer_blob_size.restype = ctypes.c_size_t
    L.cet_informer_blob_size.argtypes = [ctypes.POINTER(InformerCfg)]
    L.cet_informer_create.argtypes = [ctypes.POINTER(InformerCfg), fp, ctypes.c_size_t,
                                      ctypes.POINTER(vp)]
    L.cet_informer_attn_size.restype = ctypes.c_size_t
    L.cet_informer_attn_size.argtypes = [vp, ctypes.c_int]
    L.cet_informer_forward.argtypes = [vp, fp, fp, fp, fp, ctypes.c_int, vp]
    L.cet_informer_select.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.cet_informer_taps_size.restype = ctypes.c_size_t
    L.cet_informer_taps_size.argtypes = [vp, ctypes.c_int]
    L.cet_informer_forward_taps.argtypes = [vp, fp, fp, fp, fp, fp, ctypes.c_int, vp]
    L.cet_rnn_blob_size.restype = ctypes.c_size_t
    L.cet_rnn_blob_size.argtypes = [ctypes.c_int] * 5
    L.cet_rnn_create.argtypes = [ctypes.c_int] * 5 + [fp, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.cet_rnn_rollout.argtypes = [vp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.cet_rnn_select.argtypes = [vp, ctypes.c_int]
    L.cet_rnn_rollout_static.argtypes = L.cet_rnn_rollout.argtypes
    L.cet_lmmse.argtypes = [fp, fp, ctypes.c_int, ctypes.c_float, fp, fp, fp, ctypes.c_long,
                            ctypes.c_int, ctypes.c_int, vp]
    L.cet_lmmse_direct.argtypes = L.cet_lmmse.argtypes
    L.cet_lmmse_llr.argtypes = [fp, fp, ctypes.c_int, ctypes.c_float, fp, fp, ctypes.c_long,
                                ctypes.c_int, ctypes.c_int, vp]
    L.cet_lmmse_block.argtypes = [fp, fp, ctypes.c_int, ctypes.c_float, fp, fp, fp, ctypes.c_long,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.cet_lmmse_block_llr.argtypes = [fp, fp, ctypes.c_int, ctypes.c_float, fp, fp, ctypes.c_long,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.cet_lmmse_grid.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_float, fp, fp, fp, fp, fp] + \
        [ctypes.c_int] * 7 + [vp]
    L.cet_ls_nn.argtypes = [fp, fp, fp, fp, ctypes.c_float, fp, fp, ctypes.c_long, ctypes.c_int,
                            ctypes.c_int, vp]
    L.cet_nmse_sums.argtypes = [fp, fp, fp, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp]
    L.cet_qpsk_bit_errors.argtypes = [fp, fp, fp, ctypes.c_long, vp]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here means header and library disagree
    _lib = L
    return L


def check(code, what):
    """Turns a non-zero status into an exception (the reference raises from torch)."""
    if code != 0:
        L = lib()
        msg = L.cet_strerror(code).decode()
        extra = f" (hipError {L.cet_last_hip_error()})" if code == 3 else ""
        raise CetError(f"{what}: {ERR_NAMES.get(code, code)}: {msg}{extra}")


def current_stream_ptr(device):
    """hipStream_t of torch's current stream on `device`, as an integer."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
