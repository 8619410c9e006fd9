"""Golden vectors for the remaining shipped checkpoints, from the reference imported in the build
container (weights read with weights_only=True, key names kept exactly as stored).

    python tests/golden/make_golden_shipped.py

  rnn_cr.npz          the reference's 192 stored CDL-B sequences (CR_data_30/36/60): x = LoadBatch of
                      slots 0..24, truth = LoadBatch of slots 25..29, and the pred_len-5 rollouts of
                      the three recurrent checkpoints that rnn_ckpt.npz holds (the ones bench.py
                      runs) and of the three trained on this data set (prediciton_code/checkpoints/
                      checkpoints_30-60_L5/{GRU,LSTM,RNN}_..., stored here with their weights)
                      -> per-slot NMSE of the reference's recurrent predictors
  informer_more.npz   informerstack_e2e (checkpoints_30-60_L5), best_model_params_V30.pt,
                      best_model_params_V60.pt: state_dict + 8 CDL-B sequences -> prediction and
                      first attention map
  rnn_more.npz        {GRU,LSTM,RNN}_CDLBFixedDirection_... and {GRU,LSTM,RNN}_best_model_params_
                      V30_uplink.pt: state_dict + 16 sequences -> rollouts at pred_len 5 and 25
  rnn_pl3.npz         prediciton_code/checkpoints/checkpoints/{GRU,LSTM,RNN}_data_0_sl25_pl3_hs256_hl2
                      (trained for pred_len 3): state_dict + 16 sequences -> rollouts at pred_len 3 and 5
  informer_h12.npz    InformerStack_5x_d_model64_n_heads12_..._seq_len35_label_len20_pred_len5_
                      ..._uplink.pt (12 heads of 5 features, 35-slot history): state_dict, 8 seeded
                      synthetic 40-slot histories -> every stage, every attention map, prediction
"""
import os
import sys

import numpy as np
import scipy.io as scio
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import PE_ROWS, REF, dec_input, load_batch, run_informer  # noqa: E402
from models.model import GRU, LSTM, RNN, InformerStack, InformerStack_e2e  # noqa: E402  (reference)

CR = {"30": "ExampleCodeTsinghua/data/CR_data_30.mat",
      "36": "ExampleCodeTsinghua/prediciton_code/results/CR_data_36.mat",
      "60": "ExampleCodeTsinghua/prediciton_code/results/CR_data_60.mat"}
HPC = "HPC/TrainedTransformers/"
L5 = "ExampleCodeTsinghua/prediciton_code/checkpoints/checkpoints_30-60_L5/"
DEFAULT = (16, 16, 16, 25, 10, 5, 5, 64, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu", True, True, "cpu")


def load_sd(path):
    ck = torch.load(os.path.join(REF, path), map_location="cpu", weights_only=True)
    ck = ck["state_dict"] if "state_dict" in ck else ck
    plain = {(k[7:] if k.startswith("module.") else k): v for k, v in ck.items()}
    return ck, plain


def store_sd(out, prefix, ck):
    for k, v in ck.items():  # key names exactly as stored in the file
        a = v.detach().cpu().numpy()
        out[f"{prefix}/sd/{k}"] = a[:, :PE_ROWS] if k.endswith("position_embedding.pe") else a


def main():
    torch.set_num_threads(4)
    data = {t: scio.loadmat(os.path.join(REF, p))["data"] for t, p in CR.items()}
    allseq = np.concatenate([data["30"], data["36"], data["60"]])

    # ---- recurrent predictors on all stored sequences ------------------------------------------
    out = {"x": load_batch(allseq[:, :25]).numpy(), "truth": load_batch(allseq[:, 25:30]).numpy()}
    x = torch.from_numpy(out["x"])
    base = HPC + "{}_enc16_hs256_hl2_seq25_label10_best_model_params_V30_uplink.pt"
    for name, cls in (("GRU", GRU), ("LSTM", LSTM), ("RNN", RNN)):
        _ck, plain = load_sd(base.format(name))
        m = cls(16, 16, 256, 2)
        m.load_state_dict(plain)
        m.eval()
        with torch.no_grad():
            out[f"{name}/pred5"] = m.test_data(x, 5, "cpu").numpy()
        err = ((out[f"{name}/pred5"] - out["truth"]) ** 2).sum((0, 2)) / (out["truth"] ** 2).sum((0, 2))
        print(name, "NMSE dB per slot", np.round(10 * np.log10(err), 2))
        # the checkpoint trained on this very data set (prediciton_code/checkpoints_30-60_L5)
        ck, plain = load_sd(L5 + f"{name}_data_0_sl25_pl5_hs256_hl2/checkpoint.pth")
        m = cls(16, 16, 256, 2)
        m.load_state_dict(plain)
        m.eval()
        store_sd(out, f"{name}_L5", ck)
        with torch.no_grad():
            out[f"{name}_L5/pred5"] = m.test_data(x, 5, "cpu").numpy()
        err = ((out[f"{name}_L5/pred5"] - out["truth"]) ** 2).sum((0, 2)) / (out["truth"] ** 2).sum((0, 2))
        print(name, "L5 checkpoint: NMSE dB per slot", np.round(10 * np.log10(err), 2))
    np.savez_compressed(os.path.join(HERE, "rnn_cr.npz"), **out)

    # ---- more Transformer checkpoints ---------------------------------------------------------------
    out = {}
    cases = {"e2e_L5": (InformerStack_e2e, L5 + "informerstack_e2e_data_0_sl25_ll10_pl5_dm64_nh8_el4_dl3_df64_"
                        "atfull_fc5_ebfixed_dtTrue/checkpoint.pth", "30"),
             "V30": (InformerStack, HPC + "best_model_params_V30.pt", "30"),
             "V60": (InformerStack, HPC + "best_model_params_V60.pt", "60")}
    for name, (cls, path, tag) in cases.items():
        ck, plain = load_sd(path)
        m = cls(*DEFAULT)
        m.load_state_dict(plain)
        m.eval()
        enc = load_batch(data[tag][8:16, :25])
        dec = dec_input(enc, 10, 5)
        pred, attns, _taps = run_informer(m, enc, dec, True)
        store_sd(out, name, ck)
        out[f"{name}/x_enc"], out[f"{name}/x_dec"], out[f"{name}/pred"] = enc.numpy(), dec.numpy(), pred
        out[f"{name}/attn_0"] = attns[0]
        print(name, len(ck), "keys, first:", next(iter(ck)), "pred", pred.shape)
    np.savez_compressed(os.path.join(HERE, "informer_more.npz"), **out)

    # ---- more recurrent checkpoints --------------------------------------------------------------------
    out = {"x": load_batch(data["36"][:16, :25]).numpy()}
    x = torch.from_numpy(out["x"])
    for name, cls in (("GRU", GRU), ("LSTM", LSTM), ("RNN", RNN)):
        for tag, fname in (("fixed_direction", f"{name}_CDLBFixedDirection_enc16_hs256_hl2_seq25_label10_"
                                                "best_model_params_V30_uplink.pt"),
                           ("best", f"{name}_best_model_params_V30_uplink.pt")):
            ck, plain = load_sd(HPC + fname)
            m = cls(16, 16, 256, 2)
            m.load_state_dict(plain)
            m.eval()
            store_sd(out, f"{name}_{tag}", ck)
            with torch.no_grad():
                for P in (5, 25):
                    out[f"{name}_{tag}/pred{P}"] = m.test_data(x, P, "cpu").numpy()
            print(name, tag, len(ck), "keys", np.abs(out[f"{name}_{tag}/pred25"]).max())
    np.savez_compressed(os.path.join(HERE, "rnn_more.npz"), **out)

    # ---- the recurrent checkpoints trained for pred_len 3 ---------------------------------------------
    out = {"x": load_batch(data["60"][:16, :25]).numpy()}
    x = torch.from_numpy(out["x"])
    PL3 = "ExampleCodeTsinghua/prediciton_code/checkpoints/checkpoints/"
    for name, cls in (("GRU", GRU), ("LSTM", LSTM), ("RNN", RNN)):
        ck, plain = load_sd(PL3 + f"{name}_data_0_sl25_pl3_hs256_hl2/checkpoint.pth")
        m = cls(16, 16, 256, 2)
        m.load_state_dict(plain)
        m.eval()
        store_sd(out, name, ck)
        with torch.no_grad():
            for P in (3, 5):
                out[f"{name}/pred{P}"] = m.test_data(x, P, "cpu").numpy()
        print(name, "pl3", len(ck), "keys", np.abs(out[f"{name}/pred3"]).max())
    np.savez_compressed(os.path.join(HERE, "rnn_pl3.npz"), **out)

    # ---- the 12-head, 35-slot checkpoint ------------------------------------------------------------------
    from channelestimationtransformer_amd import synth
    out = {}
    ck, plain = load_sd(HPC + "InformerStack_5x_d_model64_n_heads12_e_layers4_d_layers3_d_ff64_dropout0.05_"
                        "attn_full_embed_fixed_activation_gelu_enc_in16_dec_in16_c_out16_seq_len35_"
                        "label_len20_pred_len5_output_attentionTrue_distilTrue_uplink.pt")
    m = InformerStack(16, 16, 16, 35, 20, 5, 5, 64, 12, 4, 3, 64, 0.05, "full", "fixed", "gelu", True, True, "cpu")
    m.load_state_dict(plain)
    m.eval()
    noisy, _ = synth.channel_histories(8, T=40, seed=12)
    enc = load_batch(noisy[:, :35])
    dec = dec_input(enc, 20, 5)
    pred, attns, taps = run_informer(m, enc, dec, True)
    store_sd(out, "h12", ck)
    out["h12/x_enc"], out["h12/x_dec"], out["h12/pred"] = enc.numpy(), dec.numpy(), pred
    out["h12/truth"] = load_batch(noisy[:, 35:40]).numpy()
    for i, a in enumerate(attns):
        out[f"h12/attn_{i}"] = a
    for i, t in enumerate(taps):
        out[f"h12/tap_{i}"] = t
    print("h12 taps", [t.shape[1] for t in taps], "attn", [a.shape for a in attns])
    np.savez_compressed(os.path.join(HERE, "informer_h12.npz"), **out)


if __name__ == "__main__":
    main()
