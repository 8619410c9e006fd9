import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import synth
dev = torch.device("cuda:0")
noisy, _ = synth.channel_histories(4100, seed=3)
xall = cet.LoadBatch(noisy[:, :25]).to(dev)
for kind in ("gru", "lstm", "rnn"):
    m, _sd = bench.make_rnn(kind)
    for B in (45, 2048, 3000, 4096):
        x = xall[:B].contiguous()
        t = m.select_kernel("tiled").test_data(x, 7, dev)
        h = m.select_kernel("half_tile").test_data(x, 7, dev)
        err = float((t - h).abs().max()); sc = float(t.abs().max())
        row = []
        for variant in ("tiled", "half_tile", "auto"):
            m.select_kernel(variant)
            d = bench.timed_steady(lambda: m.test_data(x, 5, dev), dev, warm_ms=30.0, run_ms=60.0)
            row.append(f"{variant} {d*1e3:.3f} ms")
        print(f"{kind} B={B}: err {err:.2e} (scale {sc:.2f})  " + "  ".join(row), flush=True)
for cls, H, NL in (("GRU", 128, 3), ("LSTM", 256, 3), ("RNN", 128, 1), ("LSTM", 128, 2), ("GRU", 256, 1)):
    torch.manual_seed(0)
    m = getattr(cet, cls)(16, 32, H, NL).eval()
    x = xall[:300].contiguous()
    t = m.select_kernel("tiled").test_data(x, 7, dev)
    h = m.select_kernel("half_tile").test_data(x, 7, dev)
    print(cls, H, NL, "err", float((t - h).abs().max()), "scale", float(t.abs().max()), flush=True)
