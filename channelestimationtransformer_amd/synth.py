"""Synthetic MIMO channel histories and equalizer inputs (seeded, numpy)."""
import numpy as np


def channel_histories(M, T=30, Nr=4, Nt=2, speed_kmh=30.0, fc=28e9, slot_s=0.625e-3, paths=8,
                      snr_db=20.0, seed=0):
    """Sum-of-sinusoids Rayleigh fading per antenna pair: h[t] = sum_p a_p exp(j(2 pi f_p t dt
    + phi_p)), f_p = f_D cos(theta_p), f_D = v fc / c; normalised to unit mean power, then
    AWGN at `snr_db` (carrier and slot length as in HPC/python/GenerateChannels.py:64,89; the
    normalisation and noise model as in HPC/python/data.py:39-55).
    Returns (noisy [M,T,Nr,Nt] complex64, clean [M,T,Nr,Nt] complex64)."""
    rng = np.random.default_rng(seed)
    fD = speed_kmh / 3.6 * fc / 3e8
    theta = rng.uniform(0, 2 * np.pi, (M, 1, Nr, Nt, paths))
    phi = rng.uniform(0, 2 * np.pi, (M, 1, Nr, Nt, paths))
    amp = (rng.standard_normal((M, 1, Nr, Nt, paths)) + 1j * rng.standard_normal((M, 1, Nr, Nt, paths)))
    amp *= np.sqrt(0.5 / paths)
    t = np.arange(T).reshape(1, T, 1, 1, 1) * slot_s
    h = (amp * np.exp(1j * (2 * np.pi * fD * np.cos(theta) * t + phi))).sum(-1)
    h = h / np.sqrt(np.mean(np.abs(h) ** 2))
    sigma = 10 ** (-snr_db / 10)
    n = np.sqrt(sigma / 2) * (rng.standard_normal(h.shape) + 1j * rng.standard_normal(h.shape))
    return (h + n).astype(np.complex64), h.astype(np.complex64)


def channel_histories_device(M, device, T=30, Nr=4, Nt=2, speed_kmh=30.0, fc=28e9, slot_s=0.625e-3,
                             paths=8, snr_db=20.0, seed=0):
    """The same process generated on the GPU with torch (for batches too large to stage
    through the host); same model and normalisation as `channel_histories`, another random
    stream. Returns (noisy, clean) complex64 tensors [M, T, Nr, Nt] on `device`."""
    import math

    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    fD = speed_kmh / 3.6 * fc / 3e8
    shape = (M, 1, Nr, Nt, paths)
    theta = torch.rand(shape, device=device, generator=g) * (2 * math.pi)
    phi = torch.rand(shape, device=device, generator=g) * (2 * math.pi)
    amp = torch.complex(torch.randn(shape, device=device, generator=g),
                        torch.randn(shape, device=device, generator=g)) * math.sqrt(0.5 / paths)
    t = torch.arange(T, device=device, dtype=torch.float32).reshape(1, T, 1, 1, 1) * slot_s
    phase = 2 * math.pi * fD * torch.cos(theta) * t + phi
    h = (amp * torch.polar(torch.ones_like(phase), phase)).sum(-1)
    h = h / torch.sqrt((h.abs() ** 2).mean())
    sigma = 10 ** (-snr_db / 10)
    n = torch.complex(torch.randn(h.shape, device=device, generator=g),
                      torch.randn(h.shape, device=device, generator=g)) * math.sqrt(sigma / 2)
    return (h + n).to(torch.complex64), h.to(torch.complex64)


def lmmse_problem(N, Nr=16, Ns=4, no=0.1, seed=0):
    """i.i.d. CN(0,1) channel, QPSK symbols, y = Hx + n with n ~ CN(0, no).
    Returns (y [N,Nr], H [N,Nr,Ns], x [N,Ns]) complex64."""
    rng = np.random.default_rng(seed)
    H = ((rng.standard_normal((N, Nr, Ns), dtype=np.float32)
          + 1j * rng.standard_normal((N, Nr, Ns), dtype=np.float32)) * np.float32(np.sqrt(0.5)))
    bits = rng.integers(0, 2, (N, Ns, 2))
    x = ((2.0 * bits[..., 0] - 1) + 1j * (2.0 * bits[..., 1] - 1)) / np.sqrt(2.0)
    n = np.sqrt(no / 2) * (rng.standard_normal((N, Nr)) + 1j * rng.standard_normal((N, Nr)))
    y = np.einsum("nrs,ns->nr", H, x) + n
    return y.astype(np.complex64), H.astype(np.complex64), x.astype(np.complex64)
