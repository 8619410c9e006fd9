"""Generates the golden fixtures in this directory from the reference implementation.

Runs ONLY in the build container (the reference tree is read from /root/reference and never
copied): it imports the reference's model classes on the CPU, loads the shipped checkpoints
with `weights_only=True`, feeds them the reference's own stored inputs plus seeded random
inputs, and stores tensors (weights, inputs, outputs, intermediate activations) as .npz.

    python tests/golden/make_golden.py

Fixtures written (all float32 / complex as noted):
  cr_pairs.npz          the reference's stored input/output pairs CR_data_{30,36,60} ->
                        CR_transformer_{30,36,60} (ExampleCodeTsinghua/data, .../results)
  informer_ckpt.npz     state_dict of LocalSimulations/models/checkpoint/checkpoint.pth
                        (positional tables cut to their first 64 rows)
  informer_ref.npz      outputs, attention maps and per-stage activations of the reference
                        InformerStack on the first 8 sequences of CR_data_30
  informer_variants.npz random-weight Informer / InformerStack variants (other wirings)
  rnn_ckpt.npz          state_dicts of the shipped GRU / LSTM / RNN checkpoints
  rnn_ref.npz           their test_data outputs (pred_len 5 and 25) + a random-weight GRU
  loadbatch.npz         LoadBatch / real2complex input/output pair
"""
import os
import sys

import numpy as np
import scipy.io as scio
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "LocalSimulations"))

from models.model import GRU, LSTM, RNN, Informer, InformerStack  # noqa: E402

torch.set_num_threads(4)
PE_ROWS = 64


def load_batch(H):
    """complex [M,T,Nr,Nt] -> float32 [M,T,2*Nr*Nt], (re,im) innermost."""
    M, T = H.shape[:2]
    Hr = H.reshape(M, T, -1)
    return torch.tensor(np.stack([Hr.real, Hr.imag], -1).reshape(M, T, -1), dtype=torch.float32)


def sd_to_np(sd):
    out = {}
    for k, v in sd.items():
        a = v.detach().cpu().numpy()
        if k.endswith("position_embedding.pe"):
            a = a[:, :PE_ROWS]
        out[k] = a
    return out


def dec_input(enc, label_len, pred_len):
    z = torch.zeros_like(enc[:, -pred_len:, :])
    return torch.cat([enc[:, enc.shape[1] - label_len:, :], z], dim=1)


def capture_taps(model, stacked):
    """Forward hooks in execution order: embedding, every encoder layer / distil conv, every
    encoder's output, the concatenated memory, decoder embedding, decoder layers, decoder."""
    taps, handles = [], []

    def hook(_m, _i, o):
        taps.append((o[0] if isinstance(o, tuple) else o).detach().numpy().copy())

    def reg(m):
        handles.append(m.register_forward_hook(hook))

    reg(model.enc_embedding)
    encs = list(model.encoder.encoders) if stacked else [model.encoder]
    for enc in encs:
        for m in enc.attn_layers:
            reg(m)
        if enc.conv_layers is not None:
            for m in enc.conv_layers:
                reg(m)
        reg(enc)
    if stacked:
        reg(model.encoder)
    reg(model.dec_embedding)
    for m in model.decoder.layers:
        reg(m)
    reg(model.decoder)
    return taps, handles


def run_informer(model, enc, dec, stacked):
    taps, handles = capture_taps(model, stacked)
    with torch.no_grad():
        res = model(enc, dec)
    for h in handles:
        h.remove()
    pred, attns = res if isinstance(res, tuple) else (res, None)
    # hooks fire when a module RETURNS, so sort into execution-completion order == the order
    # they were appended; an encoder's own hook fires after its layers, as wanted. A
    # non-stacked model has no separate "memory" stage: duplicate the encoder output.
    if not stacked:
        n_enc_taps = 1 + len(model.encoder.attn_layers) + (
            len(model.encoder.conv_layers) if model.encoder.conv_layers is not None else 0) + 1
        taps.insert(n_enc_taps, taps[n_enc_taps - 1].copy())
    flat_attn = []
    if attns is not None:
        for a in (attns if stacked else [attns]):
            flat_attn.extend(x.numpy() for x in a)
    return pred.numpy(), flat_attn, taps


def main():
    out = {}
    # ---- the reference's stored pairs -------------------------------------------------
    pairs = {"30": ("ExampleCodeTsinghua/data/CR_data_30.mat",
                    "ExampleCodeTsinghua/data/CR_transformer_30.mat"),
             "36": ("ExampleCodeTsinghua/prediciton_code/results/CR_data_36.mat",
                    "ExampleCodeTsinghua/prediciton_code/results/CR_transformer_36.mat"),
             "60": ("ExampleCodeTsinghua/prediciton_code/results/CR_data_60.mat",
                    "ExampleCodeTsinghua/prediciton_code/results/CR_transformer_60.mat")}
    cr = {}
    for tag, (fi, fo) in pairs.items():
        cr[f"data_{tag}"] = scio.loadmat(os.path.join(REF, fi))["data"].astype(np.complex64)
        cr[f"transformer_{tag}"] = scio.loadmat(os.path.join(REF, fo))["data"].astype(np.complex64)
    np.savez_compressed(os.path.join(HERE, "cr_pairs.npz"), **cr)

    # ---- shipped InformerStack checkpoint ----------------------------------------------
    ck = torch.load(os.path.join(REF, "LocalSimulations/models/checkpoint/checkpoint.pth"),
                    map_location="cpu", weights_only=True)["state_dict"]
    model = InformerStack(16, 16, 16, 25, 10, 5, 5, 64, 8, 4, 3, 64, 0.05, "full", "fixed",
                          "gelu", True, True, "cpu")
    model.load_state_dict(ck)
    model.eval()
    np.savez_compressed(os.path.join(HERE, "informer_ckpt.npz"), **sd_to_np(model.state_dict()))

    # the stored pairs must be reproduced by the imported reference
    for tag in pairs:
        enc = load_batch(scio.loadmat(os.path.join(REF, pairs[tag][0]))["data"][:, :25])
        with torch.no_grad():
            pred = model(enc, dec_input(enc, 10, 5))[0].numpy()
        got = pred.reshape(64, 5, 8, 2)
        got = (got[..., 0] + 1j * got[..., 1]).reshape(64, 5, 4, 2)
        err = np.abs(got - cr[f"transformer_{tag}"]).max()
        print(f"stored pair {tag}: reference-vs-stored max abs {err:.3e}")
        assert err < 5e-6

    enc = load_batch(scio.loadmat(os.path.join(REF, pairs["30"][0]))["data"][:8, :25])
    dec = dec_input(enc, 10, 5)
    pred, attns, taps = run_informer(model, enc, dec, True)
    ref = {"x_enc": enc.numpy(), "x_dec": dec.numpy(), "pred": pred}
    for i, a in enumerate(attns):
        ref[f"attn_{i}"] = a
    for i, t in enumerate(taps):
        ref[f"tap_{i}"] = t
    ref["n_attn"], ref["n_taps"] = len(attns), len(taps)
    np.savez_compressed(os.path.join(HERE, "informer_ref.npz"), **ref)
    print("informer_ref taps:", [t.shape[1] for t in taps], "attn:", [a.shape[-1] for a in attns])

    # ---- other wirings, random weights ---------------------------------------------------
    var = {}

    def variant(name, cls, stacked, args, B=4, seed=0):
        torch.manual_seed(seed)
        m = cls(*args)
        # non-trivial BatchNorm statistics so that the eval-mode folding is exercised
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.normal_(0, 0.2)
        m.eval()
        enc_in, dec_in, _c, seq_len, label_len, out_len = args[:6]
        x = torch.randn(B, seq_len, enc_in)
        xd = torch.cat([x[:, seq_len - label_len:, :dec_in], torch.zeros(B, out_len, dec_in)], 1)
        p, at, tp = run_informer(m, x, xd, stacked)
        for k, v in sd_to_np(m.state_dict()).items():
            var[f"{name}/sd/{k}"] = v
        var[f"{name}/x_enc"], var[f"{name}/x_dec"], var[f"{name}/pred"] = x.numpy(), xd.numpy(), p
        for i, a in enumerate(at):
            var[f"{name}/attn_{i}"] = a
        for i, t in enumerate(tp):
            var[f"{name}/tap_{i}"] = t
        print(name, "taps", [t.shape[1] for t in tp], "attn", [a.shape[-1] for a in at])

    # Informer (single encoder), e_layers 3, d_layers 2
    variant("informer_e3d2", Informer, False,
            (16, 16, 16, 25, 10, 5, 5, 64, 8, 3, 2, 64, 0.0, "full", "fixed", "gelu", True, True, "cpu"))
    # InformerStack on a 4x4 MIMO history (enc_in 32), shorter sequence
    variant("stack_in32", InformerStack, True,
            (32, 32, 32, 20, 8, 4, 5, 64, 8, 4, 3, 64, 0.0, "full", "fixed", "gelu", True, True, "cpu"))
    # no distilling, relu, output_attention False
    variant("informer_nodistil_relu", Informer, False,
            (16, 16, 16, 12, 6, 3, 5, 64, 8, 2, 1, 64, 0.0, "full", "fixed", "relu", False, False, "cpu"))
    np.savez_compressed(os.path.join(HERE, "informer_variants.npz"), **var)

    # ---- recurrent predictors --------------------------------------------------------------
    rnn_ck, rnn_ref = {}, {}
    x = load_batch(scio.loadmat(os.path.join(REF, pairs["30"][0]))["data"][:16, :25])
    rnn_ref["x"] = x.numpy()
    base = "HPC/TrainedTransformers/{}_enc16_hs256_hl2_seq25_label10_best_model_params_V30_uplink.pt"
    for name, cls in (("GRU", GRU), ("LSTM", LSTM), ("RNN", RNN)):
        sd = torch.load(os.path.join(REF, base.format(name)), map_location="cpu", weights_only=True)
        m = cls(16, 16, 256, 2)
        m.load_state_dict(sd)
        m.eval()
        for k, v in m.state_dict().items():
            rnn_ck[f"{name}/{k}"] = v.numpy()
        with torch.no_grad():
            for P in (5, 25):
                rnn_ref[f"{name}/pred{P}"] = m.test_data(x, P, "cpu").numpy()
    torch.manual_seed(1)
    m = GRU(32, 32, 128, 3)
    m.eval()
    xr = 0.5 * torch.randn(8, 25, 32)
    for k, v in m.state_dict().items():
        rnn_ck[f"GRU_f32_h128_l3/{k}"] = v.numpy()
    rnn_ref["GRU_f32_h128_l3/x"] = xr.numpy()
    with torch.no_grad():
        rnn_ref["GRU_f32_h128_l3/pred5"] = m.test_data(xr, 5, "cpu").numpy()
    torch.manual_seed(2)
    m = LSTM(16, 24, 128, 1)
    m.eval()
    xr = 0.5 * torch.randn(8, 12, 16)
    for k, v in m.state_dict().items():
        rnn_ck[f"LSTM_i24_h128_l1/{k}"] = v.numpy()
    rnn_ref["LSTM_i24_h128_l1/x"] = xr.numpy()
    with torch.no_grad():
        rnn_ref["LSTM_i24_h128_l1/pred5"] = m.test_data(xr, 5, "cpu").numpy()
    np.savez_compressed(os.path.join(HERE, "rnn_ckpt.npz"), **rnn_ck)
    np.savez_compressed(os.path.join(HERE, "rnn_ref.npz"), **rnn_ref)

    # ---- LoadBatch / real2complex (HPC/python/data.py:20-37) ----------------------------------
    sys.path.insert(0, os.path.join(REF, "HPC/python"))
    import data as refdata  # noqa: E402
    Hc = cr["data_30"][:4].astype(np.complex128)
    lb = refdata.LoadBatch(Hc).numpy()
    r2c = refdata.real2complex(lb[:, :5])
    np.savez_compressed(os.path.join(HERE, "loadbatch.npz"), H=Hc, loadbatch=lb, real2complex=r2c)
    print("done")


if __name__ == "__main__":
    main()
