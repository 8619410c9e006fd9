"""Golden vectors for recurrent shapes beyond hidden 128 / 256 and three layers: GRU(16, 32, 512, 1),
LSTM(16, 16, 64, 4), RNN(16, 32, 256, 4), and for hidden sizes between the kernels' (100, 200, 300,
40), 48 sequences of 25 slots, pred_len 5.

Runs ONLY in the build container: imports the reference's model classes from /root/reference on
the CPU (LocalSimulations/models/model.py:276, 350, 426 and their test_data methods :298-312,
:373-388, :449-465) and stores weights, inputs and outputs. No checkpoint of these shapes ships
with the reference, so the weights are PyTorch's default initialisation under a fixed seed.

    python tests/golden/make_golden_shapes.py   ->  rnn_shapes.npz
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REF, "LocalSimulations"))
sys.path.insert(1, ROOT)

from models.model import GRU, LSTM, RNN  # noqa: E402

from channelestimationtransformer_amd import synth  # noqa: E402
from channelestimationtransformer_amd.utils import LoadBatch  # noqa: E402

torch.set_num_threads(4)
CASES = (("GRU", GRU, (16, 32, 512, 1)), ("LSTM", LSTM, (16, 16, 64, 4)), ("RNN", RNN, (16, 32, 256, 4)),
         # hidden sizes between the kernels' (the host side adds zero units up to 128 / 256 / 512 / 64)
         ("GRU100", GRU, (16, 16, 100, 2)), ("LSTM200", LSTM, (16, 32, 200, 3)), ("RNN300", RNN, (16, 16, 300, 1)),
         ("GRU40", GRU, (16, 16, 40, 1)))


def main():
    out = {}
    noisy, _ = synth.channel_histories(48, T=30, seed=17)
    x = LoadBatch(noisy[:, :25])
    assert tuple(x.shape) == (48, 25, 16)
    out["x"] = x.numpy()
    for seed, (name, cls, shape) in enumerate(CASES):
        torch.manual_seed(200 + seed)
        m = cls(*shape)
        m.eval()
        out[f"{name}/shape"] = np.array(shape)
        for k, v in m.state_dict().items():
            out[f"{name}/sd/{k}"] = v.numpy()
        with torch.no_grad():
            out[f"{name}/pred5"] = m.test_data(x, 5, "cpu").numpy()
        print(name, shape, float(np.abs(out[f"{name}/pred5"]).max()))
    np.savez_compressed(os.path.join(HERE, "rnn_shapes.npz"), **out)


if __name__ == "__main__":
    main()
