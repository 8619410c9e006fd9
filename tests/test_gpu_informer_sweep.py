"""Randomized configurations of the Transformer predictor against the CPU checker: every debug
stage, every attention map and the prediction, on each kernel path (one wave per sequence,
8-wave workgroups with and without packed tiles). Weights and BatchNorm statistics are random;
the checker itself is pinned to the reference by tests/test_oracle.py."""
import numpy as np
import pytest
import torch

import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import blobs
from test_gpu_informer import TOL_ATTN, TOL_PRED, TOL_STAGE, run_taps

pytestmark = pytest.mark.gpu

# (class, enc_in, seq_len, label_len, pred_len, e_layers, d_layers, activation, distil, batch)
CONFIGS = [
    ("InformerStack", 16, 25, 10, 5, 4, 3, "gelu", True, 13),   # the default shape, ragged batch
    ("InformerStack", 16, 30, 12, 4, 5, 1, "relu", True, 9),    # longest input, three encoders
    ("InformerStack", 32, 16, 8, 8, 3, 2, "gelu", True, 8),     # one-tile input, 32 channels
    ("InformerStack", 16, 24, 20, 12, 4, 2, "gelu", True, 5),   # 32 decoder rows (two tiles)
    ("Informer", 16, 8, 4, 2, 3, 1, "gelu", True, 20),          # short from the start
    ("Informer", 16, 28, 6, 3, 4, 2, "relu", True, 12),
    ("Informer", 32, 20, 10, 10, 2, 3, "gelu", False, 7),       # no distilling: 20 rows throughout
    ("Informer", 16, 7, 3, 1, 1, 1, "gelu", True, 33),          # single layer, 7 rows
    ("InformerStack", 16, 12, 6, 6, 3, 1, "gelu", False, 11),   # stack without distilling
    ("InformerStack", 16, 29, 1, 2, 6, 1, "gelu", True, 6),     # deepest stack the planner takes
    ("InformerStack", 16, 14, 7, 7, 4, 2, "gelu", False, 10),   # slice 14 -> 7, memory of 21 rows
    ("InformerStack", 32, 16, 3, 13, 5, 2, "relu", True, 17),   # packed from the third step on
]


def randomized(cls, enc_in, seq_len, label_len, pred_len, e_layers, d_layers, act, distil, seed):
    g = torch.Generator().manual_seed(seed)
    m = getattr(cet, cls)(enc_in, enc_in, enc_in, seq_len, label_len, pred_len, 5, 64, 8, e_layers,
                          d_layers, 64, 0.0, "full", "fixed", act, True, distil)
    sd = m.state_dict()
    for k, v in sd.items():
        if k.endswith("position_embedding.pe") or k.endswith("num_batches_tracked"):
            continue
        if k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) * 1.5 + 0.25)
        elif k.endswith("norm.weight") or ".norm1.weight" in k or ".norm2.weight" in k or ".norm3.weight" in k:
            v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=g))
        elif v.dim() >= 2:
            v.copy_(torch.randn(v.shape, generator=g) * (0.9 / np.sqrt(v[0].numel())))
        else:
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
    m.load_state_dict(sd)
    return m.eval(), {k: v.numpy() for k, v in sd.items()}


@pytest.mark.parametrize("kernel", ["one_wave", "workgroup", "workgroup_unpacked"])
@pytest.mark.parametrize("case", range(len(CONFIGS)))
def test_random_configuration(case, kernel, oracle, monkeypatch):
    monkeypatch.setenv("CET_INFORMER_KERNEL", "one_wave" if kernel == "one_wave" else "workgroup")
    monkeypatch.setenv("CET_INFORMER_PACK", "0" if kernel.endswith("unpacked") else "1")
    cls, enc_in, seq_len, label_len, pred_len, e_layers, d_layers, act, distil, B = CONFIGS[case]
    m, sd = randomized(cls, enc_in, seq_len, label_len, pred_len, e_layers, d_layers, act, distil, case)
    cfg = m._cfg
    rng = np.random.default_rng(100 + case)
    x_enc = rng.standard_normal((B, seq_len, enc_in)).astype(np.float32)
    x_dec = np.concatenate([x_enc[:, seq_len - label_len:], np.zeros((B, pred_len, enc_in), np.float32)], 1)
    want, want_attn, want_taps = oracle.informer_forward(cfg, blobs.pack_informer(sd, cfg), x_enc, x_dec,
                                                         want_attn=True, want_taps=True)
    pred, taps, attn = run_taps(m, x_enc, x_dec)
    rows = oracle.tap_rows(cfg)
    for i, (t, w) in enumerate(zip(oracle.split_taps(taps, rows, B, 64), want_taps)):
        err = np.abs(t - w).max()
        assert err < TOL_STAGE, f"stage {i} (rows {rows[i]}): {err}"
    for i, (a, w) in enumerate(zip(oracle.split_attn(attn, oracle.attn_lengths(cfg), B, 8), want_attn)):
        assert np.abs(a - w).max() < TOL_ATTN, f"attention map {i}"
    assert np.abs(pred - want).max() < TOL_PRED


# (class, d_model, n_heads, seq_len, label_len, pred_len, e_layers, d_layers, distil, batch): head
# counts other than 8 (d_keys = d_model // n_heads <= 8, up to 12 heads), histories of up to 46
# slots and decoder inputs of up to 48 rows (three 16-row tiles)
WIDE_CONFIGS = [
    ("InformerStack", 64, 12, 46, 30, 18, 4, 2, True, 5),   # the longest history and decoder input
    ("InformerStack", 64, 10, 40, 20, 5, 4, 1, True, 7),    # 10 heads of 6
    ("InformerStack", 60, 12, 33, 10, 5, 3, 2, True, 6),    # 60 features, 12 heads of 5
    ("InformerStack", 64, 8, 40, 24, 8, 4, 2, True, 9),     # 8 heads, three row tiles
    ("InformerStack", 48, 6, 20, 10, 5, 4, 2, True, 11),    # 6 heads of 8: two padded heads
    ("InformerStack", 64, 9, 25, 10, 5, 4, 3, True, 4),     # odd head count: 9 heads of 7
    ("Informer", 64, 8, 44, 12, 4, 3, 1, True, 3),          # 44 -> 23 -> 13 rows, single encoder
    ("Informer", 64, 12, 20, 10, 6, 2, 2, False, 8),        # 12 heads, memory of 20 rows
    ("Informer", 32, 4, 36, 33, 3, 3, 1, True, 5),          # 4 heads of 8, 36 decoder rows
    ("InformerStack", 58, 11, 30, 12, 5, 3, 2, True, 5),    # 11 heads of 5: three to a tile, the last tile two
    ("Informer", 36, 9, 25, 10, 5, 2, 1, True, 6),          # 9 heads of 4: three to a tile, no fifth feature
]


@pytest.mark.parametrize("kernel", ["one_wave", "workgroup"])
@pytest.mark.parametrize("case", range(len(WIDE_CONFIGS)))
def test_random_wide_configuration(case, kernel, oracle, monkeypatch):
    monkeypatch.setenv("CET_INFORMER_KERNEL", kernel)
    cls, dm, nh, seq_len, label_len, pred_len, e_layers, d_layers, distil, B = WIDE_CONFIGS[case]
    g = torch.Generator().manual_seed(50 + case)
    m = getattr(cet, cls)(16, 16, 16, seq_len, label_len, pred_len, 5, dm, nh, e_layers, d_layers, 40,
                          0.0, "full", "fixed", "gelu", True, distil)
    sd = m.state_dict()
    for k, v in sd.items():
        if k.endswith("position_embedding.pe") or k.endswith("num_batches_tracked"):
            continue
        if k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) * 1.5 + 0.25)
        elif v.dim() >= 2:
            v.copy_(torch.randn(v.shape, generator=g) * (0.9 / np.sqrt(v[0].numel())))
        else:
            v.copy_((1.0 if k.endswith("weight") else 0.0) + 0.1 * torch.randn(v.shape, generator=g))
    m.load_state_dict(sd)
    m.eval()
    sd = {k: v.numpy() for k, v in sd.items()}
    cfg = m._cfg
    rng = np.random.default_rng(300 + case)
    x_enc = rng.standard_normal((B, seq_len, 16)).astype(np.float32)
    x_dec = np.concatenate([x_enc[:, seq_len - label_len:], np.zeros((B, pred_len, 16), np.float32)], 1)
    want, want_attn, want_taps = oracle.informer_forward(cfg, blobs.pack_informer(sd, cfg), x_enc, x_dec,
                                                         want_attn=True, want_taps=True)
    pred, taps, attn = run_taps(m, x_enc, x_dec)
    rows = oracle.tap_rows(cfg)
    for i, (t, w) in enumerate(zip(oracle.split_taps(taps, rows, B, 64), want_taps)):
        err = np.abs(t[..., :dm] - w).max()
        assert err < TOL_STAGE, f"stage {i} (rows {rows[i]}): {err}"
        assert not t[..., dm:].any(), f"stage {i}: padded features are not zero"
    for i, (a, w) in enumerate(zip(oracle.split_attn(attn, oracle.attn_lengths(cfg), B, nh), want_attn)):
        assert np.abs(a - w).max() < TOL_ATTN, f"attention map {i}"
    assert np.abs(pred - want).max() < TOL_PRED
