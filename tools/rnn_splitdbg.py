"""Debug aid: hidden-split against batch-tiled rollouts, error per predicted step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import channelestimationtransformer_amd as cet

torch.manual_seed(0)
for cls, F, I, H, L in (("RNN", 16, 16, 128, 1), ("RNN", 16, 16, 256, 1), ("RNN", 16, 16, 256, 2), ("GRU", 16, 16, 256, 2), ("LSTM", 16, 16, 256, 2), ("GRU", 32, 32, 128, 3)):
    m = getattr(cet, cls)(F, I, H, L).eval()
    for B, seq, P in ((16, 1, 1), (16, 2, 1), (16, 1, 2), (16, 3, 1), (16, 25, 5), (300, 12, 4), (700, 12, 4)):
        x = torch.randn(B, seq, F, device="cuda") * 0.5
        a = m.select_kernel("tiled").test_data(x, P, "cuda")
        b = m.select_kernel("split").test_data(x, P, "cuda")
        err = (a - b).abs().amax(dim=(0, 2)).cpu().numpy()
        bad = (a - b).abs().amax(dim=(1, 2)) > 1e-3
        print(cls, (F, I, H, L), "B", B, "seq", seq, "P", P, "err/step", [f"{e:.1e}" for e in err], "bad seqs", int(bad.sum()), "nan", int(torch.isnan(b).sum()), flush=True)
