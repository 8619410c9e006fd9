"""Boundary marshalling between complex channel tensors and the predictors' real layout."""
import numpy as np
import torch


def LoadBatch(H):
    """complex [M, T, Nr, Nt] -> float32 torch tensor [M, T, 2*Nr*Nt].

    Antenna pair index nr*Nt + nt, (re, im) interleaved innermost: the layout of
    `LoadBatch` in LocalSimulations/Predict.py:66-77 (same in HPC/python/data.py:20-31).
    """
    H = np.asarray(H)
    M, T = H.shape[0], H.shape[1]
    flat = H.reshape(M, T, -1)
    out = np.empty((M, T, flat.shape[2], 2), dtype=np.float32)
    out[..., 0] = flat.real
    out[..., 1] = flat.imag
    return torch.from_numpy(out.reshape(M, T, -1))


def real2complex(data):
    """[B, P, 2N] real -> [B, P, N] complex (LocalSimulations/Predict.py:79-83).

    Accepts numpy arrays (returned as numpy, like the reference) and torch tensors
    (returned as torch complex tensors on the same device, no host round trip)."""
    if isinstance(data, torch.Tensor):
        B, P, N = data.shape
        return torch.view_as_complex(data.reshape(B, P, N // 2, 2).contiguous())
    data = np.asarray(data)
    B, P, N = data.shape
    d = data.reshape(B, P, N // 2, 2)
    return d[..., 0] + 1j * d[..., 1]


def load_batch_device(H):
    """Device-side LoadBatch: complex torch tensor [M, T, Nr, Nt] -> float32 [M, T, 2*Nr*Nt]
    without leaving the GPU (a reinterpretation of the complex64 storage)."""
    H = H.to(torch.complex64).contiguous()
    M, T = H.shape[0], H.shape[1]
    return torch.view_as_real(H.reshape(M, T, -1)).reshape(M, T, -1)


def decoder_input(enc_inp, label_len, pred_len):
    """`cat(enc_inp[:, seq_len-label_len:], zeros[B, pred_len, C])`: the decoder input the
    reference's drivers build (LocalSimulations/Predict.py:105-106)."""
    zeros = torch.zeros_like(enc_inp[:, -pred_len:, :])
    return torch.cat([enc_inp[:, enc_inp.shape[1] - label_len:, :], zeros], dim=1)


class TriangularCausalMask:
    """bool triu(diagonal=1) mask [B, 1, L, L] (LocalSimulations/utils.py:108-116). The HIP
    kernel applies the same mask by index comparison; this object only exists so that code
    written against the reference's `utils` module keeps importing."""

    def __init__(self, B, L, device="cpu"):
        with torch.no_grad():
            self._mask = torch.triu(torch.ones((B, 1, L, L), dtype=torch.bool), diagonal=1).to(device)

    @property
    def mask(self):
        return self._mask
