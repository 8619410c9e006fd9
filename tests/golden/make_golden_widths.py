"""Golden vectors for feature counts other than the kernels' widths (the model classes add zero
channels on the host side): InformerStack with enc_in = dec_in = c_out = 8 (a 2x2 channel),
Informer with enc_in = dec_in = 24 and c_out = 22, GRU with 6 features (3x1), LSTM with 10.

Runs ONLY in the build container: imports the reference's model classes from /root/reference on
the CPU (LocalSimulations/models/model.py:11-81, 161-234, 350-388, 426-465) and stores weights,
inputs and outputs. Weights: PyTorch's default initialisation under a fixed seed, BatchNorm
statistics randomised (no checkpoint of these widths ships with the reference).

    python tests/golden/make_golden_widths.py   ->  widths.npz
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "LocalSimulations"))

from models.model import GRU, LSTM, Informer, InformerStack  # noqa: E402

torch.set_num_threads(4)


def main():
    out = {}
    rng = np.random.default_rng(23)
    # (name, class, enc_in, dec_in, c_out, seq_len, label_len, pred_len, e_layers, d_layers)
    for seed, (name, cls, ei, di, co, sl, ll, pl, el, dl) in enumerate((
            ("stack8", InformerStack, 8, 8, 8, 25, 10, 5, 4, 3),
            ("informer24", Informer, 24, 24, 22, 20, 8, 4, 3, 2))):
        torch.manual_seed(300 + seed)
        m = cls(ei, di, co, sl, ll, pl, 5, 64, 8, el, dl, 64, 0.0, "full", "fixed", "gelu", True, True,
                torch.device("cpu"))
        sd = m.state_dict()
        for k, v in sd.items():  # BatchNorm statistics other than (0, 1)
            if k.endswith("running_var"):
                v.copy_(torch.rand(v.shape) * 1.5 + 0.25)
            elif k.endswith("running_mean"):
                v.copy_(0.1 * torch.randn(v.shape))
        m.load_state_dict(sd)
        m.eval()
        B = 6
        x_enc = torch.from_numpy(rng.standard_normal((B, sl, ei)).astype(np.float32))
        x_dec = torch.cat([x_enc[:, sl - ll:, :di] if di <= ei else x_enc[:, sl - ll:],
                           torch.zeros(B, pl, di)], 1)
        with torch.no_grad():
            pred, attns = m(x_enc, x_dec)
        out[f"{name}/args"] = np.array([ei, di, co, sl, ll, pl, el, dl])
        out[f"{name}/x_enc"], out[f"{name}/x_dec"] = x_enc.numpy(), x_dec.numpy()
        out[f"{name}/pred"] = pred.numpy()
        flat = []
        for a in attns:
            flat += list(a) if isinstance(a, (list, tuple)) else [a]
        for i, a in enumerate(flat):
            out[f"{name}/attn{i}"] = a.numpy()
        for k, v in m.state_dict().items():
            if not k.endswith("position_embedding.pe"):  # the fixed sin / cos table (models/embed.py:7-24): not stored
                out[f"{name}/sd/{k}"] = v.numpy()
        print(name, tuple(pred.shape), float(pred.abs().max()), len(flat), "maps")
    for seed, (name, cls, shape) in enumerate((("gru6", GRU, (6, 16, 128, 2)), ("lstm10", LSTM, (10, 10, 256, 1)))):
        torch.manual_seed(310 + seed)
        m = cls(*shape)
        m.eval()
        x = torch.from_numpy((0.7 * rng.standard_normal((20, 25, shape[0]))).astype(np.float32))
        with torch.no_grad():
            out[f"{name}/pred5"] = m.test_data(x, 5, "cpu").numpy()
        out[f"{name}/shape"], out[f"{name}/x"] = np.array(shape), x.numpy()
        for k, v in m.state_dict().items():
            out[f"{name}/sd/{k}"] = v.numpy()
        print(name, shape, float(np.abs(out[f"{name}/pred5"]).max()))
    np.savez_compressed(os.path.join(HERE, "widths.npz"), **out)


if __name__ == "__main__":
    main()
