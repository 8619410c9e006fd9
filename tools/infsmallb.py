"""Transformer forward time by batch size and launch shape (one wave per sequence / 8-wave
workgroups) next to the automatic choice: python tools/infsmallb.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import channelestimationtransformer_amd as cet  # noqa: E402
from channelestimationtransformer_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
model, _ck = bench.make_informer(True)
noisy, _ = synth.channel_histories(6144, seed=3)
xall = cet.LoadBatch(noisy[:, :25]).to(dev)
for B in (64, 256, 512, 768, 1024, 1280, 1536, 2048, 2304, 2560, 3072, 4096, 4352, 5120, 6144):
    x = xall[:B].contiguous()
    xd = cet.decoder_input(x, 10, 5)
    row = []
    for k in ("one_wave", "workgroup", "auto"):
        model.select_kernel(k)
        d = bench.timed_steady(lambda: model(x, xd), dev, warm_ms=10.0, run_ms=30.0)
        row.append(f"{k} {d * 1e3:.3f}")
    print(f"B={B}: " + "  ".join(row) + " ms", flush=True)
