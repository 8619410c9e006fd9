"""The shipped 12-head / 35-slot checkpoint (large launch shape of the Transformer kernel) on 4096
sequences: step time and MFMA fraction. CET_LIB selects the build (libcet_prof.so prints the
per-phase clocks of sampled waves)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import channelestimationtransformer_amd as cet  # noqa: E402
from channelestimationtransformer_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
h12 = np.load(os.path.join(ROOT, "tests", "golden", "informer_h12.npz"))
m = cet.InformerStack(16, 16, 16, 35, 20, 5, 5, 64, 12, 4, 3, 64, 0.05, "full", "fixed", "gelu", False, True)
m.load_state_dict(bench.torch_sd({k[len("h12/sd/"):].replace("module.", "", 1): h12[k]
                                  for k in h12.files if k.startswith("h12/sd/")}))
m = m.eval().to(dev)
noisy, _ = synth.channel_histories_device(B, dev, T=40, seed=12)
x = cet.load_batch_device(noisy[:, :35]).contiguous()
xd = cet.decoder_input(x, 20, 5)
if "prof" in os.environ.get("CET_LIB", ""):
    m(x, xd)
    torch.cuda.synchronize()
    m(x, xd)
    torch.cuda.synchronize()
else:
    d = bench.timed_steady(lambda: m(x, xd), dev)
    f = bench.informer_flop(35, 20, 5, H=12)
    print(f"B={B}: {d * 1e3:.4f} ms, {B / d / 1e6:.3f} M pred/s, frac {f * B / d / 1e12 / bench.PEAK_FP32_MFMA_TFLOPS:.4f}")
