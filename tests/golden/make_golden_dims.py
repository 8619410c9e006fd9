"""Golden vectors for the shipped Transformer checkpoints with d_model 32 and 54
(ExampleCodeTsinghua/prediciton_code/checkpoints/checkpoints/informerstack_*_dm{32,54}_*): the
reference imported in the build container, the shipped weights (weights_only=True), the first 8
sequences of the reference's CR_data_30 as input.

    python tests/golden/make_golden_dims.py      ->  informer_dims.npz
"""
import os
import sys

import numpy as np
import scipy.io as scio
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, dec_input, load_batch, run_informer, sd_to_np  # noqa: E402
from models.model import InformerStack  # noqa: E402  (reference, via make_golden's sys.path)

DIR = "ExampleCodeTsinghua/prediciton_code/checkpoints/checkpoints/"
CKPT = {
    "dm32": (32, DIR + "informerstack_data_0_sl25_ll10_pl3_dm32_nh8_el4_dl3_df64_atfull_fc5_ebfixed_dtTrue/checkpoint.pth"),
    "dm54": (54, DIR + "informerstack_data_0_sl25_ll10_pl3_dm54_nh8_el4_dl3_df64_atfull_fc5_ebfixed_dtTrue/checkpoint.pth"),
}


def main():
    out = {}
    enc = load_batch(scio.loadmat(os.path.join(REF, "ExampleCodeTsinghua/data/CR_data_30.mat"))["data"][:8, :25])
    dec = dec_input(enc, 10, 3)
    for name, (dm, path) in CKPT.items():
        ck = torch.load(os.path.join(REF, path), map_location="cpu", weights_only=True)
        ck = ck["state_dict"] if "state_dict" in ck else ck
        ck = {(k[7:] if k.startswith("module.") else k): v for k, v in ck.items()}
        m = InformerStack(16, 16, 16, 25, 10, 3, 5, dm, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu",
                          True, True, "cpu")
        m.load_state_dict(ck)
        m.eval()
        pred, attns, taps = run_informer(m, enc, dec, True)
        for k, v in sd_to_np(m.state_dict()).items():
            out[f"{name}/sd/{k}"] = v
        out[f"{name}/x_enc"], out[f"{name}/x_dec"], out[f"{name}/pred"] = enc.numpy(), dec.numpy(), pred
        for i, a in enumerate(attns):
            out[f"{name}/attn_{i}"] = a
        for i, t in enumerate(taps):
            out[f"{name}/tap_{i}"] = t
        print(name, "pred", pred.shape, "taps", [t.shape[1:] for t in taps][:4], "attn", [a.shape[-1] for a in attns],
              "q_proj", tuple(ck["encoder.encoders.0.attn_layers.0.attention.query_projection.weight"].shape))
    np.savez_compressed(os.path.join(HERE, "informer_dims.npz"), **out)


if __name__ == "__main__":
    main()
