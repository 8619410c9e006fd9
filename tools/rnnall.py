"""Rollout time of every recurrent kernel by shape and batch size (random weights, 25 + 5 steps):
python tools/rnnall.py  ->  one line per (cell, hidden, layers, batch): ms per kernel choice and the
automatic choice's; '-' where a kernel does not take the batch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import channelestimationtransformer_amd as cet  # noqa: E402
from channelestimationtransformer_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
sizes = [16, 64, 128, 256, 512, 1024, 1536, 2048, 4096]
noisy, _ = synth.channel_histories(max(sizes), seed=3)
xall = cet.LoadBatch(noisy[:, :25]).to(dev)
shapes = ((256, 2), (128, 2), (256, 1), (256, 3), (128, 3), (128, 1))
if len(sys.argv) > 1:
    shapes = tuple(tuple(int(v) for v in a.split("x")) for a in sys.argv[1:])
for cls in ("GRU", "LSTM", "RNN"):
    for H, NL in shapes:
        torch.manual_seed(1)
        m = getattr(cet, cls)(16, 32, H, NL).eval()
        for B in sizes:
            x = xall[:B].contiguous()
            row = []
            for variant in ("split", "half_tile_4", "half_tile_2", "half_tile", "tiled", "auto"):
                m.select_kernel(variant)
                try:
                    d = bench.timed_steady(lambda: m.test_data(x, 5, dev), dev, warm_ms=10.0, run_ms=25.0)
                    row.append(f"{variant} {d * 1e3:.3f}")
                except RuntimeError:
                    row.append(f"{variant} -")
            print(f"{cls} H={H} L={NL} B={B}: " + "  ".join(row), flush=True)
