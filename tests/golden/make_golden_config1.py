"""Golden vectors for the literal "4x4 MIMO" recurrent configuration: GRU / LSTM / RNN
(features 32, input 32, hidden 256, 2 layers), 64 sequences of 25 slots, pred_len 5 and 25.

Runs ONLY in the build container: imports the reference's model classes from /root/reference on
the CPU (LocalSimulations/models/model.py:276, 350, 426 and their test_data methods :298-312,
:373-388, :449-465) and stores weights, inputs and outputs. No checkpoint of this shape ships
with the reference, so the weights are PyTorch's default initialisation under a fixed seed.

    python tests/golden/make_golden_config1.py   ->  rnn_config1.npz
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


def main():
    out = {}
    noisy, _ = synth.channel_histories(64, T=30, Nr=4, Nt=4, seed=11)  # 16 antenna pairs: 32 features
    x = LoadBatch(noisy[:, :25])
    assert tuple(x.shape) == (64, 25, 32)
    out["x"] = x.numpy()
    for seed, (name, cls) in enumerate((("GRU", GRU), ("LSTM", LSTM), ("RNN", RNN))):
        torch.manual_seed(100 + seed)
        m = cls(32, 32, 256, 2)
        m.eval()
        for k, v in m.state_dict().items():
            out[f"{name}/sd/{k}"] = v.numpy()
        with torch.no_grad():
            for P in (5, 25):
                out[f"{name}/pred{P}"] = m.test_data(x, P, "cpu").numpy()
        print(name, {P: float(np.abs(out[f"{name}/pred{P}"]).max()) for P in (5, 25)})
    np.savez_compressed(os.path.join(HERE, "rnn_config1.npz"), **out)


if __name__ == "__main__":
    main()
