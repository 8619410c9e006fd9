"""Rollout time of batches between the kernels' natural sizes (16 and 32 sequences per CU and
beyond): the automatic choice (part by part) against the batch-tiled kernel alone.
python tools/rnnparts.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import channelestimationtransformer_amd as cet  # noqa: E402
from channelestimationtransformer_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
sizes = [4096, 4500, 5000, 5500, 6000, 6500, 7000, 8192, 9000, 10000, 12288, 13000]
noisy, _ = synth.channel_histories(max(sizes), seed=3)
xall = cet.LoadBatch(noisy[:, :25]).to(dev)
for cls, H, NL in (("GRU", 256, 2), ("LSTM", 256, 2), ("RNN", 256, 2), ("GRU", 128, 3), ("GRU", 512, 2)):
    torch.manual_seed(1)
    m = getattr(cet, cls)(16, 16, H, NL).eval()
    for B in sizes:
        x = xall[:B].contiguous()
        row = []
        ref = None
        for variant in ("tiled", "half_tile", "auto"):
            m.select_kernel(variant)
            try:
                d = bench.timed_steady(lambda: m.test_data(x, 5, dev), dev, warm_ms=10.0, run_ms=30.0)
                out = m.test_data(x, 5, dev)
                ref = out if ref is None else ref
                row.append(f"{variant} {d * 1e3:.3f} ms (diff {float((out - ref).abs().max()):.1e})")
            except RuntimeError:
                row.append(f"{variant} -")
        print(f"{cls} H={H} L={NL} B={B}: " + "  ".join(row), flush=True)
