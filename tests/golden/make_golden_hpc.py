"""Golden vectors for two more of the shipped Transformer checkpoints (HPC/TrainedTransformers/
InformerStack_best_model_params_V30_uplink.pt and best_model_params_V30_MultiGPU.pt, the latter
saved from a DataParallel model: `module.` key prefix): the reference imported in the build
container, the shipped weights (weights_only=True) WITH the key names as stored, the first 8
sequences of the reference's CR_data_36 as input.

    python tests/golden/make_golden_hpc.py      ->  informer_hpc.npz
"""
import os
import sys

import numpy as np
import scipy.io as scio
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import PE_ROWS, REF, dec_input, load_batch, run_informer  # noqa: E402
from models.model import InformerStack  # noqa: E402  (reference, via make_golden's sys.path)

CKPT = {
    "uplink": "HPC/TrainedTransformers/InformerStack_best_model_params_V30_uplink.pt",
    "multigpu": "HPC/TrainedTransformers/best_model_params_V30_MultiGPU.pt",
}


def main():
    out = {}
    data = scio.loadmat(os.path.join(REF, "ExampleCodeTsinghua/prediciton_code/results/CR_data_36.mat"))["data"]
    enc = load_batch(data[:8, :25])
    dec = dec_input(enc, 10, 5)
    for name, path in CKPT.items():
        ck = torch.load(os.path.join(REF, path), map_location="cpu", weights_only=True)
        ck = ck["state_dict"] if "state_dict" in ck else ck
        plain = {(k[7:] if k.startswith("module.") else k): v for k, v in ck.items()}
        m = InformerStack(16, 16, 16, 25, 10, 5, 5, 64, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu",
                          True, True, "cpu")
        m.load_state_dict(plain)
        m.eval()
        pred, attns, _taps = run_informer(m, enc, dec, True)
        for k, v in ck.items():  # key names exactly as stored in the file
            a = v.detach().cpu().numpy()
            out[f"{name}/sd/{k}"] = a[:, :PE_ROWS] if k.endswith("position_embedding.pe") else a
        out[f"{name}/x_enc"], out[f"{name}/x_dec"], out[f"{name}/pred"] = enc.numpy(), dec.numpy(), pred
        out[f"{name}/attn_0"] = attns[0]
        print(name, len(ck), "keys, first:", next(iter(ck)), "pred", pred.shape)
    np.savez_compressed(os.path.join(HERE, "informer_hpc.npz"), **out)


if __name__ == "__main__":
    main()
