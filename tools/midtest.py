"""The half-tile rollout kernels (16 sequences per workgroup; a tile's hidden units on 1, 2 or 4
workgroups) against the batch-tiled one: results and times by batch size, other shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import synth
dev = torch.device("cuda:0")
noisy, _ = synth.channel_histories(4100, seed=3)
xall = cet.LoadBatch(noisy[:, :25]).to(dev)
LIMIT = {"half_tile": 4096, "half_tile_2": 2048, "half_tile_4": 1024}
for kind in ("gru", "lstm", "rnn"):
    m, _sd = bench.make_rnn(kind)
    for B in (45, 600, 1024, 1536, 2048, 4096):
        x = xall[:B].contiguous()
        t = m.select_kernel("tiled").test_data(x, 7, dev)
        row = []
        for variant in ("split", "half_tile", "half_tile_2", "half_tile_4", "auto"):
            if B > LIMIT.get(variant, 1 << 30):
                continue
            h = m.select_kernel(variant).test_data(x, 7, dev)
            err = float((t - h).abs().max())
            d = bench.timed_steady(lambda: m.test_data(x, 5, dev), dev, warm_ms=30.0, run_ms=60.0)
            row.append(f"{variant} {d*1e3:.3f} ms (err {err:.1e})")
        print(f"{kind} B={B}: " + "  ".join(row), flush=True)
for cls, H, NL in (("GRU", 128, 3), ("LSTM", 256, 3), ("RNN", 128, 1), ("LSTM", 128, 2), ("GRU", 256, 1)):
    torch.manual_seed(0)
    m = getattr(cet, cls)(16, 32, H, NL).eval()
    x = xall[:300].contiguous()
    t = m.select_kernel("tiled").test_data(x, 7, dev)
    errs = []
    for variant in ("half_tile", "half_tile_2") + (("half_tile_4",) if H == 256 else ()):
        h = m.select_kernel(variant).test_data(x, 7, dev)
        errs.append(f"{variant} {float((t - h).abs().max()):.1e}")
    print(cls, H, NL, " ".join(errs), "scale", float(t.abs().max()), flush=True)
