"""Parity of the HIP Transformer predictor (through the host module and the C ABI) with the
golden vectors from the reference and with the CPU checker."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import (DEFAULT_ARGS, INFORMER_VARIANTS, cfg_from_args, dec_input_np, load_batch_np,
                      load_npz, sub_dict)
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import _cabi, blobs, synth

pytestmark = pytest.mark.gpu

# fp32 throughout; the reference itself reproduces its stored outputs to ~1.2e-6.
TOL_PRED = 2e-5   # max-abs on outputs of magnitude <= ~2.5
TOL_STAGE = 5e-5  # max-abs on internal activations (LayerNorm outputs, magnitude <= ~10)
TOL_ATTN = 2e-5   # attention probabilities (softmax outputs after up to 7 layers)


@pytest.fixture(autouse=True, params=["one_wave", "workgroup", "workgroup_unpacked"])
def kernel_variant(request, monkeypatch):
    """Every test of this file runs on every Transformer kernel path: the library picks one by
    batch size (one wave per sequence up to 4 sequences per CU, else 8-wave workgroups with
    LDS-staged weights, which also pack two short sequences into one tile at the end of the
    encoder stack); CET_INFORMER_KERNEL / CET_INFORMER_PACK pin the choice for a call."""
    monkeypatch.setenv("CET_INFORMER_KERNEL", request.param.split("_")[0] if request.param != "one_wave" else "one_wave")
    monkeypatch.setenv("CET_INFORMER_PACK", "0" if request.param.endswith("unpacked") else "1")
    return request.param


def to_torch_sd(d):
    sd = {}
    for k, v in d.items():
        t = torch.from_numpy(np.array(v))
        if k.endswith("position_embedding.pe"):
            full = torch.zeros(1, 5000, t.shape[-1])
            full[:, :t.shape[1]] = t
            t = full
        sd[k] = t
    return sd


def build(cls_name, args, sd):
    m = getattr(cet, cls_name)(*args)
    m.load_state_dict(to_torch_sd(sd))
    return m.eval()


def run_taps(m, x_enc, x_dec):
    """Direct C-ABI call of the debug entry point -> (pred, taps list, attn list)."""
    L = _cabi.lib()
    dev = torch.device("cuda:0")
    h, _ = m._handle_for(dev)
    B = x_enc.shape[0]
    c = m._cfg
    xe = torch.from_numpy(x_enc).to(dev)
    xd = torch.from_numpy(x_dec).to(dev)
    out = torch.empty(B, c.pred_len, c.c_out, device=dev)
    taps = torch.full((L.cet_informer_taps_size(h, B),), float("nan"), device=dev)
    attn = torch.full((L.cet_informer_attn_size(h, B),), float("nan"), device=dev)
    rc = L.cet_informer_forward_taps(h, xe.data_ptr(), xd.data_ptr(), out.data_ptr(), attn.data_ptr(),
                                     taps.data_ptr(), B, None)
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu().numpy(), taps.cpu().numpy(), attn.cpu().numpy()


def test_stored_pairs_of_the_reference():
    """CR_data_{30,36,60} -> CR_transformer_{30,36,60} with the shipped checkpoint, following
    LocalSimulations/Predict.py:99-120 line by line."""
    cr = load_npz("cr_pairs.npz")
    m = build("InformerStack", DEFAULT_ARGS, dict(load_npz("informer_ckpt.npz"))).to("cuda")
    for tag in ("30", "36", "60"):
        channel = cr[f"data_{tag}"]
        data = cet.LoadBatch(channel[:, :25, :, :])
        enc_inp = data.to("cuda")
        dec_inp = torch.zeros_like(enc_inp[:, -5:, :])
        dec_inp = torch.cat([enc_inp[:, 25 - 10:25, :], dec_inp], dim=1)
        outputs = m(enc_inp, dec_inp)[0]
        outputs = cet.real2complex(outputs.cpu().detach().numpy()).reshape([64, 5, 4, 2])
        assert np.abs(outputs - cr[f"transformer_{tag}"]).max() < TOL_PRED


def test_every_stage_against_reference_activations(oracle):
    ref = load_npz("informer_ref.npz")
    m = build("InformerStack", DEFAULT_ARGS, dict(load_npz("informer_ckpt.npz")))
    pred, taps, attn = run_taps(m, ref["x_enc"], ref["x_dec"])
    rows = oracle.tap_rows(m._cfg)
    taps = oracle.split_taps(taps, rows, 8, 64)
    attn = oracle.split_attn(attn, oracle.attn_lengths(m._cfg), 8, 8)
    for i, t in enumerate(taps):
        err = np.abs(t - ref[f"tap_{i}"]).max()
        assert err < TOL_STAGE, f"stage {i} (rows {rows[i]}): max abs err {err}"
    for i, a in enumerate(attn):
        err = np.abs(a - ref[f"attn_{i}"]).max()
        assert err < TOL_ATTN, f"attention map {i}: {err}"
    assert np.abs(pred - ref["pred"]).max() < TOL_PRED


@pytest.mark.parametrize("name", sorted(INFORMER_VARIANTS))
def test_other_wirings(name, oracle):
    v = load_npz("informer_variants.npz")
    cls, args = INFORMER_VARIANTS[name]
    m = build(cls, args, sub_dict(v, f"{name}/sd"))
    pred, taps, attn = run_taps(m, v[f"{name}/x_enc"], v[f"{name}/x_dec"])
    B = v[f"{name}/x_enc"].shape[0]
    rows = oracle.tap_rows(m._cfg)
    for i, t in enumerate(oracle.split_taps(taps, rows, B, 64)):
        err = np.abs(t - v[f"{name}/tap_{i}"]).max()
        assert err < TOL_STAGE, f"{name} stage {i}: {err}"
    assert np.abs(pred - v[f"{name}/pred"]).max() < TOL_PRED
    for i, a in enumerate(oracle.split_attn(attn, oracle.attn_lengths(m._cfg), B, 8)):
        if f"{name}/attn_{i}" in v.files:
            assert np.abs(a - v[f"{name}/attn_{i}"]).max() < TOL_ATTN


@pytest.mark.parametrize("name,dm", [("dm32", 32), ("dm54", 54)])
def test_shipped_narrow_checkpoints(name, dm, oracle):
    """The reference ships Transformer checkpoints with d_model 32 and 54 (d_keys 4 and 6; the
    projections of the 54-wide model are 48 wide because d_keys = d_model // n_heads,
    models/attn.py:126-133). The library widens them to its 64-feature tiles with zeros; every
    stage, every attention map and the prediction must still equal the reference's, and the
    padded feature columns of every stage must be exactly zero."""
    v = load_npz("informer_dims.npz")
    args = (16, 16, 16, 25, 10, 3, 5, dm, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu", True, True)
    m = build("InformerStack", args, sub_dict(v, f"{name}/sd"))
    pred, taps, attn = run_taps(m, v[f"{name}/x_enc"], v[f"{name}/x_dec"])
    B = v[f"{name}/x_enc"].shape[0]
    rows = oracle.tap_rows(m._cfg)
    for i, t in enumerate(oracle.split_taps(taps, rows, B, 64)):
        err = np.abs(t[..., :dm] - v[f"{name}/tap_{i}"]).max()
        assert err < TOL_STAGE, f"{name} stage {i}: {err}"
        assert not t[..., dm:].any(), f"{name} stage {i}: padded features are not zero"
    assert np.abs(pred - v[f"{name}/pred"]).max() < TOL_PRED
    for i, a in enumerate(oracle.split_attn(attn, oracle.attn_lengths(m._cfg), B, 8)):
        assert np.abs(a - v[f"{name}/attn_{i}"]).max() < TOL_ATTN, f"{name} attention map {i}"
    out = m.to("cuda")(torch.from_numpy(v[f"{name}/x_enc"]).cuda(), torch.from_numpy(v[f"{name}/x_dec"]).cuda())
    assert np.array_equal(out[0].cpu().numpy(), pred)


@pytest.mark.parametrize("name", ["uplink", "multigpu"])
def test_shipped_hpc_checkpoints(name):
    """Two more shipped checkpoints (HPC/TrainedTransformers/InformerStack_best_model_params_V30_
    uplink.pt, best_model_params_V30_MultiGPU.pt), loaded with their stored key names (the
    `module.` prefix of a DataParallel model), against the imported reference's outputs."""
    v = load_npz("informer_hpc.npz")
    m = build("InformerStack", DEFAULT_ARGS, sub_dict(v, f"{name}/sd")).cuda()
    pred, attns = m(torch.from_numpy(v[f"{name}/x_enc"]).cuda(), torch.from_numpy(v[f"{name}/x_dec"]).cuda())
    assert np.abs(pred.cpu().numpy() - v[f"{name}/pred"]).max() < TOL_PRED
    assert np.abs(attns[0][0].cpu().numpy() - v[f"{name}/attn_0"]).max() < TOL_ATTN


@pytest.mark.parametrize("name,cls", [("e2e_L5", "InformerStack_e2e"), ("V30", "InformerStack"), ("V60", "InformerStack")])
def test_more_shipped_checkpoints(name, cls):
    """prediciton_code/checkpoints/checkpoints_30-60_L5/informerstack_e2e_.../checkpoint.pth and
    HPC/TrainedTransformers/best_model_params_V30.pt / _V60.pt with their stored key names, against
    the imported reference's outputs (tests/golden/make_golden_shipped.py)."""
    v = load_npz("informer_more.npz")
    m = build(cls, DEFAULT_ARGS, sub_dict(v, f"{name}/sd")).cuda()
    pred, attns = m(torch.from_numpy(v[f"{name}/x_enc"]).cuda(), torch.from_numpy(v[f"{name}/x_dec"]).cuda())
    assert np.abs(pred.cpu().numpy() - v[f"{name}/pred"]).max() < TOL_PRED
    assert np.abs(attns[0][0].cpu().numpy() - v[f"{name}/attn_0"]).max() < TOL_ATTN


def test_shipped_12_head_35_slot_checkpoint(kernel_variant, oracle):
    """HPC/TrainedTransformers/InformerStack_5x_d_model64_n_heads12_..._seq_len35_label_len20_
    pred_len5_..._uplink.pt: 12 heads of 64 // 12 = 5 features (60-wide projections,
    models/attn.py:126-133), 35-slot history (three 16-row tiles), 25-row decoder input. Every
    stage, every attention map and the prediction against the imported reference; the library
    serves this family with its one-wave launch shape whatever shape is asked for."""
    v = load_npz("informer_h12.npz")
    args = (16, 16, 16, 35, 20, 5, 5, 64, 12, 4, 3, 64, 0.05, "full", "fixed", "gelu", True, True)
    m = build("InformerStack", args, sub_dict(v, "h12/sd"))
    pred, taps, attn = run_taps(m, v["h12/x_enc"], v["h12/x_dec"])
    B = v["h12/x_enc"].shape[0]
    rows = oracle.tap_rows(m._cfg)
    for i, t in enumerate(oracle.split_taps(taps, rows, B, 64)):
        err = np.abs(t - v[f"h12/tap_{i}"]).max()
        assert err < TOL_STAGE, f"stage {i} (rows {rows[i]}): {err}"
    for i, a in enumerate(oracle.split_attn(attn, oracle.attn_lengths(m._cfg), B, 12)):
        assert a.shape == v[f"h12/attn_{i}"].shape
        assert np.abs(a - v[f"h12/attn_{i}"]).max() < TOL_ATTN, f"attention map {i}"
    assert np.abs(pred - v["h12/pred"]).max() < TOL_PRED
    # through the module API, with the reference's nesting of the maps
    out, attns = m.cuda()(torch.from_numpy(v["h12/x_enc"]).cuda(), torch.from_numpy(v["h12/x_dec"]).cuda())
    assert np.array_equal(out.cpu().numpy(), pred)
    assert [a.shape[1:] for a in attns[0]] == [(12, 35, 35), (12, 19, 19), (12, 11, 11), (12, 7, 7)]
    assert [a.shape[1:] for a in attns[1]] == [(12, 7, 7), (12, 5, 5), (12, 4, 4)]


def test_e2e_class_is_the_same_wiring():
    """InformerStack_e2e (LocalSimulations/models/model.py:84-158) differs from InformerStack in
    its training target only: same state_dict keys, same outputs."""
    ck = dict(load_npz("informer_ckpt.npz"))
    ref = load_npz("informer_ref.npz")
    a = build("InformerStack", DEFAULT_ARGS, ck).cuda()
    b = build("InformerStack_e2e", DEFAULT_ARGS, ck).cuda()
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    x, xd = torch.from_numpy(ref["x_enc"]).cuda(), torch.from_numpy(ref["x_dec"]).cuda()
    assert torch.equal(a(x, xd)[0], b(x, xd)[0])


def test_module_api_and_attention_nesting(oracle):
    ck = dict(load_npz("informer_ckpt.npz"))
    m = build("InformerStack", DEFAULT_ARGS, ck).cuda()
    from channelestimationtransformer_amd import synth
    noisy, _ = synth.channel_histories(37, seed=5)  # ragged batch (not a multiple of anything)
    x = cet.LoadBatch(noisy[:, :25]).cuda()
    xd = cet.decoder_input(x, 10, 5)
    pred, attns = m(x, xd)
    assert pred.shape == (37, 5, 16) and len(attns) == 2 and [len(a) for a in attns] == [4, 3]
    assert [a.shape[-1] for a in attns[0]] == [25, 14, 8, 5] and [a.shape[-1] for a in attns[1]] == [5, 4, 3]
    cfg = m._cfg
    want, want_attn, _ = oracle.informer_forward(cfg, blobs.pack_informer(ck, cfg), x.cpu().numpy(),
                                                 xd.cpu().numpy(), want_attn=True)
    assert np.abs(pred.cpu().numpy() - want).max() < TOL_PRED
    flat = [a for grp in attns for a in grp]
    for a, w in zip(flat, want_attn):
        assert np.abs(a.cpu().numpy() - w).max() < TOL_ATTN
        assert torch.allclose(a.sum(-1), torch.ones_like(a.sum(-1)), atol=1e-5)
    # output_attention=False returns the bare tensor, bit-identical predictions
    m.output_attention = False
    p2 = m(x, xd)
    assert isinstance(p2, torch.Tensor) and torch.equal(p2, pred)
    # batch of one, and an empty batch
    assert torch.equal(m(x[:1], xd[:1]), pred[:1])
    assert m(x[:0], xd[:0]).shape == (0, 5, 16)


def test_full_size_batch_properties(oracle):
    """Batch 4096 (the benchmark size): per-sample independence (any sub-batch gives the
    same bits) and agreement with the CPU checker on a strided sample of the batch."""
    from channelestimationtransformer_amd import synth
    ck = dict(load_npz("informer_ckpt.npz"))
    m = build("InformerStack", DEFAULT_ARGS, ck).cuda()
    m.output_attention = False
    noisy, _ = synth.channel_histories(4096, seed=0)
    x = cet.LoadBatch(noisy[:, :25]).cuda()
    xd = cet.decoder_input(x, 10, 5)
    full = m(x, xd)
    assert torch.isfinite(full).all()
    idx = torch.arange(0, 4096, 61, device="cuda")
    assert torch.equal(m(x[idx].contiguous(), xd[idx].contiguous()), full[idx])
    cfg = m._cfg
    want, _, _ = oracle.informer_forward(cfg, blobs.pack_informer(ck, cfg), x[idx].cpu().numpy(),
                                         xd[idx].cpu().numpy())
    assert np.abs(full[idx].cpu().numpy() - want).max() < TOL_PRED
    # delta-NMSE vs the checker on the same sample, in dB
    truth = cet.LoadBatch(noisy[:, 25:30])[idx.cpu()].numpy()
    def nmse(p):
        return 10 * np.log10(((truth - p) ** 2).sum() / (truth ** 2).sum())
    assert abs(nmse(full[idx].cpu().numpy()) - nmse(want)) < 0.05


def test_default_kernel_choice_agrees(monkeypatch):
    """Without the override the library switches kernels with the batch size; both choices
    must agree to the output tolerance on the same sequences."""
    from channelestimationtransformer_amd import synth
    monkeypatch.delenv("CET_INFORMER_KERNEL")
    m = build("InformerStack", DEFAULT_ARGS, dict(load_npz("informer_ckpt.npz"))).cuda()
    m.output_attention = False
    noisy, _ = synth.channel_histories(4099, seed=2)  # not a multiple of the 8-wave workgroup
    x = cet.LoadBatch(noisy[:, :25]).cuda()
    xd = cet.decoder_input(x, 10, 5)
    big = m(x, xd)             # workgroup kernel
    small = m(x[-67:].contiguous(), xd[-67:].contiguous())  # one-wave kernel
    assert torch.isfinite(big).all()
    assert (big[-67:] - small).abs().max() < TOL_PRED


@pytest.mark.parametrize("name,cls", [("stack8", "InformerStack"), ("informer24", "Informer")])
def test_feature_counts_other_than_the_kernels_widths(name, cls):
    """enc_in = dec_in = c_out = 8 (InformerStack) and enc_in = dec_in = 24, c_out = 22 (Informer)
    against the imported reference (tests/golden/make_golden_widths.py;
    LocalSimulations/models/model.py:67-81, 220-234): the classes add zero input channels and
    output rows up to the kernels' widths (blobs.padded_informer_cfg) and drop them again."""
    v = load_npz("widths.npz")
    ei, di, co, sl, ll, pl, el, dl = (int(n) for n in v[f"{name}/args"])
    m = getattr(cet, cls)(ei, di, co, sl, ll, pl, 5, 64, 8, el, dl, 64, 0.0, "full", "fixed", "gelu", True, True)
    sd = m.state_dict()
    for k, a in sub_dict(v, f"{name}/sd").items():
        sd[k] = torch.from_numpy(np.array(a))
    m.load_state_dict(sd)
    m.eval()
    x_enc, x_dec = torch.from_numpy(v[f"{name}/x_enc"]).cuda(), torch.from_numpy(v[f"{name}/x_dec"]).cuda()
    for kernel in ("one_wave", "workgroup"):
        pred, attns = m.select_kernel(kernel)(x_enc, x_dec)
        assert tuple(pred.shape) == (6, pl, co)
        assert np.abs(pred.cpu().numpy() - v[f"{name}/pred"]).max() < TOL_PRED, kernel
        flat = []
        for a in attns:
            flat += list(a) if isinstance(a, (list, tuple)) else [a]
        for i, a in enumerate(flat):
            assert np.abs(a.cpu().numpy() - v[f"{name}/attn{i}"]).max() < TOL_ATTN, (kernel, i)


@pytest.mark.parametrize("B", [2048 + 3, 2048 + 500, 4096 + 517])
def test_rounds_of_the_workgroup_kernel_and_a_one_wave_tail(B):
    """The automatic choice runs a batch of whole workgroup rounds (8 sequences per CU) plus a small
    rest in two launches, the rest on the one-wave kernel: every sequence's prediction and maps
    are those of the kernel that computed it, in place (on a GPU with another CU count the batch
    may go in one launch: then the first comparison holds for all of it)."""
    inf = build("InformerStack", DEFAULT_ARGS, dict(load_npz("informer_ckpt.npz"))).cuda()
    noisy, _ = synth.channel_histories(B, seed=8)
    x = cet.LoadBatch(noisy[:, :25]).cuda()
    xd = cet.decoder_input(x, 10, 5)
    pred, attns = inf.select_kernel("auto")(x, xd)
    pw, aw = inf.select_kernel("workgroup")(x, xd)
    po, ao = inf.select_kernel("one_wave")(x, xd)
    assert torch.isfinite(pred).all() and (pred - pw).abs().max() < TOL_PRED
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n = B // (8 * cus) * (8 * cus) if 0 < B % (8 * cus) <= 2 * cus else B
    assert torch.equal(pred[:n], pw[:n]) and torch.equal(pred[n:], po[n:])
    for e in range(len(attns)):
        for l in range(len(attns[e])):
            assert torch.equal(attns[e][l][:n], aw[e][l][:n]) and torch.equal(attns[e][l][n:], ao[e][l][n:])
