"""Times the per-RE equalizer for every antenna shape, and the block-fading entry points served
by it (per-RE noise: the kernel gathers its matrix):
    CET_LIB=... python tools/lmmseshapes.py
Prints ms per call, G RE/s and the share of the 8 TB/s HBM peak on the algorithmic bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import channelestimationtransformer_amd as cet


def timed(step, n=100):
    for _ in range(200):  # about 80 ms of load: the shader clock has settled
        step()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        step()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


N = 1 << 23
for Nr, Ns in [(16, 4), (16, 2), (8, 4), (8, 2), (4, 4), (4, 2), (16, 1), (2, 2)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.view_as_complex(torch.randn(N, Nr, 2, device="cuda", generator=g))
    H = torch.view_as_complex(torch.randn(N, Nr, Ns, 2, device="cuda", generator=g))
    ms = timed(lambda: cet.lmmse(y, H, 0.1))
    byt = N * (Nr * Ns * 8 + Nr * 8 + Ns * 12)
    print(f"{Nr}x{Ns}: {ms:.4f} ms  {N / ms / 1e6:.2f} G RE/s  {byt / ms / 1e6 / 8000:.3f} of HBM peak", flush=True)
    if (Nr, Ns) in ((16, 4), (4, 2)):
        no = torch.full((N,), 0.1, device="cuda")
        Hb = H[: N // 14 + 1].contiguous()
        ms = timed(lambda: cet.lmmse(y, Hb, no, h_share=14))
        print(f"{Nr}x{Ns} gathered (one matrix per 14 REs, per-RE noise): {ms:.4f} ms  {N / ms / 1e6:.2f} G RE/s", flush=True)
    del y, H
