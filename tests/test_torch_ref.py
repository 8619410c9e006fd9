"""The second CPU checker (oracle/torch_ref.py: plain torch.nn.functional calls on the
state_dict) against the golden vectors of the reference and against the C checker."""
import numpy as np
import pytest
import torch

from conftest import (DEFAULT_ARGS, INFORMER_VARIANTS, RNN_CASES, cfg_from_args, load_npz, sub_dict)
from channelestimationtransformer_amd import blobs
from oracle import torch_ref

TOL = 5e-6


def _kw(args, stack):
    return dict(pred_len=args[5], n_heads=args[8], e_layers=args[9], d_layers=args[10], stack=stack,
                distil=args[17], activation=args[15])


def test_default_checkpoint_against_the_reference():
    ref, ck = load_npz("informer_ref.npz"), dict(load_npz("informer_ckpt.npz"))
    pred, attn = torch_ref.informer_forward(ck, ref["x_enc"], ref["x_dec"], **_kw(DEFAULT_ARGS, True))
    assert np.abs(pred.numpy() - ref["pred"]).max() < TOL
    assert len(attn) == int(ref["n_attn"])
    for i, a in enumerate(attn):
        assert np.abs(a.numpy() - ref[f"attn_{i}"]).max() < TOL, i


@pytest.mark.parametrize("name", sorted(INFORMER_VARIANTS))
def test_other_wirings(name):
    v = load_npz("informer_variants.npz")
    cls, args = INFORMER_VARIANTS[name]
    pred, _ = torch_ref.informer_forward(sub_dict(v, f"{name}/sd"), v[f"{name}/x_enc"], v[f"{name}/x_dec"],
                                         **_kw(args, cls != "Informer"))
    assert np.abs(pred.numpy() - v[f"{name}/pred"]).max() < TOL


@pytest.mark.parametrize("name,dm", [("dm32", 32), ("dm54", 54)])
def test_narrow_checkpoints(name, dm):
    v = load_npz("informer_dims.npz")
    args = (16, 16, 16, 25, 10, 3, 5, dm, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu", True, True)
    pred, attn = torch_ref.informer_forward(sub_dict(v, f"{name}/sd"), v[f"{name}/x_enc"], v[f"{name}/x_dec"],
                                            **_kw(args, True))
    assert np.abs(pred.numpy() - v[f"{name}/pred"]).max() < TOL
    assert np.abs(attn[0].numpy() - v[f"{name}/attn_0"]).max() < TOL


def test_agrees_with_the_c_checker_on_random_input(oracle):
    """Two independent restatements on inputs no fixture holds."""
    ck = dict(load_npz("informer_ckpt.npz"))
    cfg = cfg_from_args(DEFAULT_ARGS, True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 25, 16, generator=g)
    xd = torch.cat([x[:, 15:], torch.zeros(16, 5, 16)], dim=1)
    want, _, _ = oracle.informer_forward(cfg, blobs.pack_informer(ck, cfg), x.numpy(), xd.numpy())
    got, _ = torch_ref.informer_forward(ck, x, xd, **_kw(DEFAULT_ARGS, True))
    assert np.abs(got.numpy() - want).max() < 2e-5


@pytest.mark.parametrize("name", sorted(RNN_CASES))
def test_recurrent_rollouts(name):
    ck, ref = load_npz("rnn_ckpt.npz"), load_npz("rnn_ref.npz")
    cls = RNN_CASES[name][0]
    x = ref["x"] if f"{name}/x" not in ref.files else ref[f"{name}/x"]
    for P in (5, 25):
        key = f"{name}/pred{P}"
        if key in ref.files:
            out = torch_ref.rnn_rollout(sub_dict(ck, name), cls.lower(), x, P).numpy()
            assert np.abs(out - ref[key]).max() < 1e-5 * max(1.0, float(np.abs(ref[key]).max()))
