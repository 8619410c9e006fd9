import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import synth
for (Nr, Ns) in [(16, 4), (8, 4), (4, 2), (2, 1), (16, 2), (16, 1), (4, 4)]:
    for (N, hs) in [(4200, 14), (4195, 14), (4099, 3), (1000, 1000), (77, 2000), (100000, 14), (5, 2)]:
        nb = -(-N // hs)
        y, _, _ = synth.lmmse_problem(N, Nr, Ns, 0.1, seed=N + Nr + hs)
        _, Hb, _ = synth.lmmse_problem(nb, Nr, Ns, 0.1, seed=7 * N + Ns)
        H = np.repeat(Hb, hs, axis=0)[:N]
        yt, Hbt, Ht = (torch.from_numpy(a).cuda() for a in (y, Hb, H))
        x0, n0 = cet.lmmse(yt, Ht, 0.1)
        x1, n1 = cet.lmmse(yt, Hbt, 0.1, h_share=hs)
        l0 = cet.lmmse_llr(yt, Ht, 0.1); l1 = cet.lmmse_llr(yt, Hbt, 0.1, h_share=hs)
        print(Nr, Ns, N, hs, "x", float((x0 - x1).abs().max()), "ne", float((n0 - n1).abs().max()),
              "llr rel", float(((l0 - l1).abs() / (1 + l0.abs())).max()), flush=True)
