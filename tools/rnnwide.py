import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import synth
dev = torch.device("cuda:0")
noisy, _ = synth.channel_histories(8192, seed=3)
xall = cet.LoadBatch(noisy[:, :25]).to(dev)
for cls, H, NL in (("GRU", 512, 2), ("LSTM", 512, 3), ("GRU", 64, 2), ("LSTM", 256, 4), ("RNN", 512, 2)):
    torch.manual_seed(0)
    m = getattr(cet, cls)(16, 32, H, NL).eval()
    for B in (64, 4096, 8192):
        x = xall[:B].contiguous()
        d = bench.timed_steady(lambda: m.test_data(x, 5, dev), dev, warm_ms=30.0, run_ms=60.0)
        G = {"RNN": 1, "GRU": 3, "LSTM": 4}[cls]
        flop = 29 * 2 * G * (H * (16 + H) + (NL - 1) * 2 * H * H) * B
        print(f"{cls} H={H} L={NL} B={B}: {d*1e3:.3f} ms  {flop / d / 1e12:.1f} TF", flush=True)
