"""Times the recurrent predictors beyond hidden 128 / 256 x 3 layers (the shapes only the half-tile
kernels take) per batch size and kernel choice; prints ms per rollout and the automatic choice's time."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import synth
dev = torch.device("cuda:0")
noisy, _ = synth.channel_histories(8192, seed=3)
xall = cet.LoadBatch(noisy[:, :25]).to(dev)
for cls, H, NL in (("GRU", 512, 2), ("LSTM", 512, 3), ("GRU", 64, 2), ("LSTM", 256, 4), ("RNN", 512, 2), ("GRU", 128, 4)):
    torch.manual_seed(0)
    m = getattr(cet, cls)(16, 32, H, NL).eval()
    for B in (64, 256, 512, 1024, 2048, 4096, 8192):
        x = xall[:B].contiguous()
        row = []
        for k in ("auto", "half_tile", "half_tile_2", "half_tile_4", "half_tile_8"):
            m.select_kernel(k)
            try:
                d = bench.timed_steady(lambda: m.test_data(x, 5, dev), dev, warm_ms=15.0, run_ms=30.0)
                row.append(f"{k}={d*1e3:.3f}")
            except RuntimeError:
                row.append(f"{k}=-")
        print(f"{cls} H={H} L={NL} B={B}: " + "  ".join(row), flush=True)
