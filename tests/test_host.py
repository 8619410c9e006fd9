"""Host logic and the C ABI surface; nothing here launches a kernel."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import DEFAULT_ARGS, INFORMER_VARIANTS, RNN_CASES, ROOT, cfg_from_args, load_npz, sub_dict
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import _cabi, blobs


def header_functions():
    text = open(os.path.join(ROOT, "include", "cet.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cet_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _cabi.lib()
    names = header_functions()
    assert set(names) >= set(_cabi.EXPORTS)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/cet.h but not exported by libcet.so"
    assert L.cet_abi_version() == 1
    assert L.cet_strerror(2).decode().startswith("configuration not supported")


def test_blob_sizes_agree_between_library_and_host():
    L = _cabi.lib()
    for args, stack in [(DEFAULT_ARGS, True)] + [(a, c != "Informer") for c, a in INFORMER_VARIANTS.values()]:
        cfg = cfg_from_args(args, stack)
        assert L.cet_informer_blob_size(ctypes.byref(cfg)) == blobs.informer_blob_len(cfg)
    ck = load_npz("rnn_ckpt.npz")
    for name, (_cls, kind, F, I, H, layers) in RNN_CASES.items():
        blob = blobs.pack_rnn(sub_dict(ck, name), kind, F, I, H, layers)
        assert L.cet_rnn_blob_size(kind, F, I, H, layers) == blob.size


def test_unsupported_configurations_are_refused():
    L = _cabi.lib()
    cfg = cfg_from_args(DEFAULT_ARGS, True)
    cfg.d_model = 128  # wider than the 64-feature tiles
    assert L.cet_informer_blob_size(ctypes.byref(cfg)) == 0
    h = ctypes.c_void_p()
    dummy = np.zeros(4, np.float32)
    rc = L.cet_informer_create(ctypes.byref(cfg), dummy.ctypes.data_as(ctypes.c_void_p), 4, ctypes.byref(h))
    assert rc == 2
    for narrow in (32, 54):  # narrower models are widened with zeros by the library
        cfg.d_model = narrow
        assert L.cet_informer_blob_size(ctypes.byref(cfg)) > 0
    cfg.d_model, cfg.n_heads = 64, 4  # 16 features per head: more than the kernels' 8
    assert L.cet_informer_blob_size(ctypes.byref(cfg)) == 0
    cfg.n_heads = 12  # 12 heads of 5 features: served
    assert L.cet_informer_blob_size(ctypes.byref(cfg)) == blobs.informer_blob_len(cfg)
    cfg.n_heads, cfg.seq_len = 8, 47  # with distilling, 46 slots is the longest history
    assert L.cet_informer_blob_size(ctypes.byref(cfg)) == 0
    cfg.distil = 0  # without distilling the encoder outputs (40 + 20 rows) exceed the 32-row memory
    assert L.cet_informer_blob_size(ctypes.byref(cfg)) == 0
    cfg.distil = 1
    cfg.seq_len, cfg.label_len = 40, 43  # decoder input of 48 rows: served
    assert L.cet_informer_blob_size(ctypes.byref(cfg)) == blobs.informer_blob_len(cfg)
    rc = L.cet_rnn_create(1, 16, 16, 100, 2, dummy.ctypes.data_as(ctypes.c_void_p), 4, ctypes.byref(h))
    assert rc == 2
    cfg = cfg_from_args(DEFAULT_ARGS, True)
    rc = L.cet_informer_create(ctypes.byref(cfg), dummy.ctypes.data_as(ctypes.c_void_p), 4, ctypes.byref(h))
    assert rc == 5  # blob length mismatch
    assert L.cet_lmmse(None, None, 0, ctypes.c_float(0.1), None, None, None, ctypes.c_long(4), 16, 4, None) == 1
    assert L.cet_lmmse(None, None, 0, ctypes.c_float(0.1), None, None, None, ctypes.c_long(0), 16, 4, None) == 0
    assert L.cet_destroy(None) == 1


def test_state_dict_keys_match_reference_checkpoints():
    m = cet.InformerStack(*DEFAULT_ARGS)
    ck = dict(load_npz("informer_ckpt.npz"))
    assert set(m.state_dict().keys()) == set(ck.keys())
    # the three on-disk conventions load
    sd = {k: torch.from_numpy(v if not k.endswith(".pe") else np.resize(v, (1, 5000, 64))) for k, v in ck.items()}
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0)
    m.load_state_dict(sd)
    m.load_state_dict({"module." + k: v for k, v in sd.items()})
    m.load_state_dict({"state_dict": sd})
    rk = load_npz("rnn_ckpt.npz")
    for name, (cls, _k, F, I, H, layers) in RNN_CASES.items():
        mod = getattr(cet, cls)(F, I, H, layers)
        assert set(mod.state_dict().keys()) == set(sub_dict(rk, name).keys())
    for name, (cls, args) in INFORMER_VARIANTS.items():
        mod = getattr(cet, cls)(*args)
        v = load_npz("informer_variants.npz")
        assert set(mod.state_dict().keys()) == set(sub_dict(v, f"{name}/sd").keys())


def test_cpu_tensors_are_refused_not_computed():
    m = cet.InformerStack(*DEFAULT_ARGS).eval()
    with pytest.raises(RuntimeError, match="CPU"):
        m(torch.zeros(2, 25, 16), torch.zeros(2, 15, 16))
    g = cet.GRU(16, 16, 256, 2).eval()
    with pytest.raises(RuntimeError, match="CPU"):
        g.test_data(torch.zeros(2, 25, 16), 5, "cpu")
    with pytest.raises(RuntimeError):
        cet.lmmse(torch.zeros(4, 16, dtype=torch.complex64), torch.zeros(4, 16, 4, dtype=torch.complex64), 0.1)
    with pytest.raises(NotImplementedError):
        cet.InformerStack(16, 16, 16, 25, 10, 5, attn="prob")
    with pytest.raises(RuntimeError, match="x_enc must be"):
        m(torch.zeros(2, 24, 16), torch.zeros(2, 15, 16))


def test_resource_grid_geometry():
    rg = cet.ResourceGrid(14, 76, 15e3, 1, 2, 6, [5, 6], True, "kronecker", [0])
    assert rg.num_effective_subcarriers == 64 and rg.num_data_symbols == 13 * 64
    rg = cet.ResourceGrid(14, 72, 15e3, 4, 1, 6, (0, 0), False, "kronecker", [2, 11])
    assert rg.num_effective_subcarriers == 72 and rg.num_data_symbols == 12 * 72


def test_synthetic_generators_are_seeded():
    from channelestimationtransformer_amd import synth
    a, _ = synth.channel_histories(4, seed=1)
    b, _ = synth.channel_histories(4, seed=1)
    assert a.shape == (4, 30, 4, 2) and np.array_equal(a, b)
    y, H, x = synth.lmmse_problem(10, 16, 4, 0.1, 0)
    assert y.shape == (10, 16) and H.shape == (10, 16, 4) and x.shape == (10, 4)


def test_antenna_grouping_round_trip():
    from channelestimationtransformer_amd import e2e
    g = torch.Generator().manual_seed(0)
    h = torch.randn(6, 7, 16, 4, generator=g)
    grouped = e2e.group_antennas(h)
    assert grouped.shape == (6 * 8, 7, 4, 2)
    assert torch.equal(e2e.ungroup_antennas(grouped, 16, 4), h)
    assert torch.equal(grouped[8 * 2 + 2 * 3 + 1, :, 2, 1], h[2, :, 4 * 3 + 2, 2 * 1 + 1])
    with pytest.raises(RuntimeError):
        e2e.group_antennas(torch.zeros(1, 2, 6, 4))


def test_copies_of_a_module_do_not_share_library_handles():
    """copy.copy, copy.deepcopy, pickling and DataParallel's replicate must start with an empty
    handle cache: a shared cache would have every copy free the original's handles when it is
    collected (HPC/python/Evaluate.py:74 wraps the model in torch.nn.DataParallel)."""
    import copy
    import pickle
    for m in (cet.InformerStack(*DEFAULT_ARGS).eval(), cet.GRU(16, 16, 256, 2).eval()):
        m._handles[0] = (m._weights_version, 0xDEAD0, None)  # stands for a live library handle
        try:
            copies = [copy.copy(m), copy.deepcopy(m), pickle.loads(pickle.dumps(m)),
                      m._replicate_for_data_parallel()]
            for c in copies:
                assert c._handles == {} and c._handles is not m._handles
                assert set(c.state_dict().keys()) == set(m.state_dict().keys())
            assert m._handles[0][1] == 0xDEAD0
            # the shallow copies share the parameters, the deep ones own theirs
            p0 = next(m.parameters())
            assert next(copies[0].parameters()) is p0 and next(copies[1].parameters()) is not p0
        finally:
            m._handles.clear()  # nothing real to free
        del copies


def test_train_data_is_not_part_of_the_inference_path():
    with pytest.raises(NotImplementedError, match="training"):
        cet.GRU(16, 16, 256, 2).train_data(torch.zeros(2, 5, 16), "cpu")
