"""Times the block-fading equalizer (one matrix per 14 REs) for every antenna shape:
    CET_LIB=... python tools/blockshapes.py
Prints ms per call, G RE/s and the share of the 8 TB/s HBM peak on the algorithmic bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import channelestimationtransformer_amd as cet

N, HS = 14 * 300000, 14
for Nr, Ns in [(16, 4), (16, 2), (8, 4), (8, 2), (4, 4), (4, 2), (16, 1), (2, 2)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.view_as_complex(torch.randn(N, Nr, 2, device="cuda", generator=g))
    H = torch.view_as_complex(torch.randn(N // HS, Nr, Ns, 2, device="cuda", generator=g))
    for _ in range(600):  # about 80 ms of load: the shader clock has settled
        cet.lmmse(y, H, 0.1, h_share=HS)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300):
        cet.lmmse(y, H, 0.1, h_share=HS)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 300
    byt = N * (Nr * 8 + Ns * 12) + (N // HS) * Nr * Ns * 8
    print(f"{Nr}x{Ns}: {ms:.4f} ms  {N / ms / 1e6:.2f} G RE/s  {byt / ms / 1e6 / 8000:.3f} of HBM peak", flush=True)
