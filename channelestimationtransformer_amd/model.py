"""Host-side mirrors of the reference's predictor classes.

Same constructor signatures, same `forward` / `test_data` call signatures and the same
`state_dict` key names as LocalSimulations/models/model.py, so the reference's checkpoints
load with `load_state_dict` unchanged and its driver scripts run as written. The modules
only *hold* parameters: all arithmetic happens in the HIP kernels of csrc/ (reached through
`torch.ops.cet.*`). There is no CPU implementation behind them; calling one with CPU tensors
raises.
"""
import ctypes
import os
import math

import numpy as np
import torch
import torch.nn as nn

from . import _cabi, blobs
from . import ops  # noqa: F401  (registers torch.ops.cet.*)


class _Node(nn.Module):
    """Bare container; children are attached under the reference's attribute names."""


class _DeviceWeights(nn.Module):
    """Keeps one library handle (packed weights on the GPU) per device.

    Packing walks the whole state_dict, so it is not repeated per call: the handle is rebuilt
    only after `load_state_dict`, after `.to()/.cuda()/.float()` style conversions, on leaving
    training mode (`.eval()` after `.train()`), or when `refresh_weights()` is called (needed
    after modifying parameters in place while in eval mode)."""

    def __init__(self):
        super().__init__()
        self._weights_version = 0
        self._handles = {}  # device index -> (version, handle as int, extra)

    # A copy of the module must not share the handle cache: every holder of the dict would free
    # the same handles in __del__. `copy.copy`, `copy.deepcopy`, pickling and
    # `nn.Module._replicate_for_data_parallel` (torch.nn.DataParallel, HPC/python/Evaluate.py:74)
    # all start the copy with an empty cache; it packs its own weights on first use.
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handles"] = {}
        return state

    def __copy__(self):
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update(self.__getstate__())
        for name in ("_parameters", "_buffers", "_modules"):
            new.__dict__[name] = self.__dict__[name].copy()
        return new

    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        replica._handles = {}
        return replica

    def refresh_weights(self):
        self._weights_version += 1

    def load_state_dict(self, state_dict, *args, **kwargs):
        self._weights_version += 1
        return super().load_state_dict(blobs.normalize_state_dict(state_dict), *args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._weights_version += 1
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        if self.training and not mode:  # leaving training mode: the parameters may have moved
            self._weights_version += 1
        return super().train(mode)

    def _create(self, device_index):  # -> (handle, extra)
        raise NotImplementedError

    def _handle_for(self, device):
        if self.training:
            raise RuntimeError("the HIP path implements the eval-mode forward only: call .eval() "
                               "first (the reference's drivers do, Predict.py:94)")
        key = device.index if device.index is not None else torch.cuda.current_device()
        cached = self._handles.get(key)
        if cached is not None and cached[0] == self._weights_version:
            return cached[1], cached[2]
        if cached is not None:
            del self._handles[key]
            _cabi.lib().cet_destroy(ctypes.c_void_p(cached[1]))
        with torch.cuda.device(key):
            handle, extra = self._create(key)
        self._handles[key] = (self._weights_version, int(handle.value), extra)
        return int(handle.value), extra

    def __del__(self):
        try:
            handles = self.__dict__.get("_handles") or {}
            while handles:
                _, (_v, h, _x) = handles.popitem()
                _cabi.lib().cet_destroy(ctypes.c_void_p(h))
        except Exception:
            pass


def _attention_holder(d_model, n_heads):
    """AttentionLayer's parameters (attn.py:121-138): d_keys = d_values = d_model // n_heads."""
    n = _Node()
    inner = (d_model // n_heads) * n_heads
    for name in ("query_projection", "key_projection", "value_projection"):
        setattr(n, name, nn.Linear(d_model, inner))
    n.out_projection = nn.Linear(inner, d_model)
    return n


def _ffn_into(node, d_model, d_ff, norms):
    node.conv1 = nn.Conv1d(d_model, d_ff, kernel_size=1)
    node.conv2 = nn.Conv1d(d_ff, d_model, kernel_size=1)
    for name in norms:
        setattr(node, name, nn.LayerNorm(d_model))


def _embedding_holder(c_in, d_model, max_len=5000):
    """`value_embedding.tokenConv` + the fixed sinusoid table `position_embedding.pe`
    (LocalSimulations/models/embed.py:7-38); the table is a registered buffer there, so it is
    part of the state_dict here too."""
    emb = _Node()
    emb.value_embedding = _Node()
    conv = nn.Conv1d(c_in, d_model, kernel_size=3, padding=1, padding_mode="circular")
    nn.init.kaiming_normal_(conv.weight, mode="fan_in", nonlinearity="leaky_relu")
    emb.value_embedding.tokenConv = conv
    pos = torch.arange(0, max_len).float().unsqueeze(1)
    div = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    emb.position_embedding = _Node()
    emb.position_embedding.register_buffer("pe", pe.unsqueeze(0))
    return emb


def _encoder_holder(n_layers, d_model, n_heads, d_ff, distil):
    enc = _Node()
    layers = []
    for _ in range(n_layers):
        lay = _Node()
        lay.attention = _attention_holder(d_model, n_heads)
        _ffn_into(lay, d_model, d_ff, ("norm1", "norm2"))
        layers.append(lay)
    enc.attn_layers = nn.ModuleList(layers)
    if distil:
        convs = []
        for _ in range(n_layers - 1):
            c = _Node()
            c.downConv = nn.Conv1d(d_model, d_model, kernel_size=3, padding=2, padding_mode="circular")
            c.norm = nn.BatchNorm1d(d_model)
            convs.append(c)
        enc.conv_layers = nn.ModuleList(convs)
    enc.norm = nn.LayerNorm(d_model)
    return enc


class _InformerBase(_DeviceWeights):
    _stack = True

    def __init__(self, enc_in, dec_in, c_out, seq_len, label_len, out_len, factor=5, d_model=512,
                 n_heads=8, e_layers=3, d_layers=2, d_ff=512, dropout=0.0, attn="prob",
                 embed="fixed", activation="gelu", output_attention=False, distil=True,
                 device=None):
        super().__init__()
        if attn != "full":
            raise NotImplementedError(
                "only attn='full' is implemented (every driver of the reference sets it; "
                "ProbSparse attention samples keys at random)")
        self.pred_len = out_len
        self.attn = attn
        self.output_attention = output_attention
        # `device` is accepted and ignored, as in the reference (model.py:165).
        self._cfg = blobs.make_informer_cfg(enc_in, dec_in, c_out, seq_len, label_len, out_len,
                                            d_model, n_heads, e_layers, d_layers, d_ff,
                                            self._stack, distil, activation)
        self._kernel_choice = (-1, -1)
        self.enc_embedding = _embedding_holder(enc_in, d_model)
        self.dec_embedding = _embedding_holder(dec_in, d_model)
        depths = blobs.encoder_depths(e_layers, self._stack)
        if self._stack:
            self.encoder = _Node()
            self.encoder.encoders = nn.ModuleList(
                [_encoder_holder(el, d_model, n_heads, d_ff, distil) for el in depths])
        else:
            self.encoder = _encoder_holder(e_layers, d_model, n_heads, d_ff, distil)
        self.decoder = _Node()
        dec_layers = []
        for _ in range(d_layers):
            lay = _Node()
            lay.self_attention = _attention_holder(d_model, n_heads)
            lay.cross_attention = _attention_holder(d_model, n_heads)
            _ffn_into(lay, d_model, d_ff, ("norm1", "norm2", "norm3"))
            dec_layers.append(lay)
        self.decoder.layers = nn.ModuleList(dec_layers)
        self.decoder.norm = nn.LayerNorm(d_model)
        self.projection = nn.Linear(d_model, c_out, bias=True)

    def _create(self, device_index):
        # feature counts other than the kernels' widths: zero channels added (blobs.padded_informer_cfg)
        dev_cfg = blobs.padded_informer_cfg(self._cfg)
        blob = np.ascontiguousarray(blobs.pack_informer(self.state_dict(), self._cfg, dev_cfg))
        L = _cabi.lib()
        h = ctypes.c_void_p()
        rc = L.cet_informer_create(ctypes.byref(dev_cfg), blob.ctypes.data_as(ctypes.c_void_p),
                                   blob.size, ctypes.byref(h))
        _cabi.check(rc, "cet_informer_create")
        choice = self._kernel_choice
        if choice == (-1, -1):  # nothing pinned on the object: the environment may pin (tools, tests)
            k = os.environ.get("CET_INFORMER_KERNEL", "")
            p = os.environ.get("CET_INFORMER_PACK", "")
            choice = ({"o": 1, "w": 2}.get(k[:1], -1 if not k else 0), -1 if not p else int(p[0] != "0"))
        if choice != (-1, -1):
            _cabi.check(L.cet_informer_select(h, *choice), "cet_informer_select")
        return h, L.cet_informer_attn_size(h, 1)

    _KERNELS = {None: -1, "auto": 0, "one_wave": 1, "workgroup": 2}

    def select_kernel(self, kernel=None, pack=None):
        """Pins the launch shape of the Transformer kernel (tests, timing): kernel "auto" |
        "one_wave" | "workgroup", pack True/False = two sequences per tile in the short encoder
        stages of the workgroup kernel; None leaves a setting unchanged. The library picks by
        batch size by default."""
        choice = (self._KERNELS[kernel], -1 if pack is None else int(bool(pack)))
        self._kernel_choice = tuple(c if c >= 0 else old for c, old in zip(choice, self._kernel_choice))
        for _v, h, _x in self._handles.values():
            _cabi.check(_cabi.lib().cet_informer_select(ctypes.c_void_p(h), *choice), "cet_informer_select")
        return self

    # -- the reference's call signature ---------------------------------------------------
    def forward(self, x_enc, x_dec, enc_self_mask=None, dec_self_mask=None, dec_enc_mask=None):
        """`[B, seq_len, enc_in]`, `[B, label_len+out_len, dec_in]` -> `[B, out_len, c_out]`,
        or `(pred, attns)` when `output_attention` (LocalSimulations/models/model.py:220-234)."""
        if enc_self_mask is not None or dec_self_mask is not None or dec_enc_mask is not None:
            raise NotImplementedError("explicit attention masks are not supported (the "
                                      "reference's callers always pass None)")
        c = self._cfg
        if x_enc.dim() != 3 or tuple(x_enc.shape[1:]) != (c.seq_len, c.enc_in):
            raise RuntimeError(f"x_enc must be [B, {c.seq_len}, {c.enc_in}], got {tuple(x_enc.shape)}")
        if x_dec.dim() != 3 or tuple(x_dec.shape) != (x_enc.shape[0], c.label_len + c.pred_len, c.dec_in):
            raise RuntimeError(f"x_dec must be [{x_enc.shape[0]}, {c.label_len + c.pred_len}, "
                               f"{c.dec_in}], got {tuple(x_dec.shape)}")
        if not x_enc.is_cuda:
            raise RuntimeError("x_enc is on the CPU: this implementation runs on ROCm GPUs only")
        handle, per_sample = self._handle_for(x_enc.device)
        want = per_sample if self.output_attention else 0
        d = blobs.padded_informer_cfg(c)
        if d.enc_in != c.enc_in:
            x_enc = torch.nn.functional.pad(x_enc, (0, d.enc_in - c.enc_in))
        if d.dec_in != c.dec_in:
            x_dec = torch.nn.functional.pad(x_dec, (0, d.dec_in - c.dec_in))
        pred, attn = torch.ops.cet.informer_forward(x_enc, x_dec, handle, c.pred_len,
                                                    d.c_out, want)
        if d.c_out != c.c_out:
            pred = pred[:, :, :c.c_out].contiguous()
        if not self.output_attention:
            return pred
        return pred, self._split_attention(attn, x_enc.shape[0])

    def _split_attention(self, flat, B):
        """Flat map buffer -> the reference's nesting: one list per encoder of the stack
        (a flat list for Informer), one `[B, n_heads, L, L]` tensor per attention layer."""
        c = self._cfg
        depths = blobs.encoder_depths(c.e_layers, c.stack)
        out, off = [], 0
        inp_len = cur = c.seq_len
        for el in depths:
            L = min(inp_len, cur)
            maps = []
            for l in range(el):
                n = B * c.n_heads * L * L
                maps.append(flat[off:off + n].view(B, c.n_heads, L, L))
                off += n
                if c.distil and l < el - 1:
                    L = (L + 1) // 2 + 1
            out.append(maps)
            cur, inp_len = L, inp_len // 2
        return out if c.stack else out[0]


class InformerStack(_InformerBase):
    """LocalSimulations/models/model.py:160-234."""
    _stack = True


class InformerStack_e2e(_InformerBase):
    """LocalSimulations/models/model.py:84-158 (same wiring as InformerStack)."""
    _stack = True


class Informer(_InformerBase):
    """LocalSimulations/models/model.py:11-81 (single encoder)."""
    _stack = False


# ---------------------------------------------------------------------------------------


class _RecurrentBase(_DeviceWeights):
    _cell = None  # "rnn" | "gru" | "lstm"

    def __init__(self, features, input_size, hidden_size, num_layers=2):
        super().__init__()
        self._kernel_choice = 0
        self.num_layers = num_layers
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.features = features
        unit = _Node()
        unit.encoder = nn.Sequential(nn.Linear(features, input_size))
        cell_cls = {"rnn": nn.RNN, "gru": nn.GRU, "lstm": nn.LSTM}[self._cell]
        setattr(unit, self._cell, cell_cls(input_size, hidden_size, num_layers))
        unit.decoder = nn.Sequential(nn.Linear(hidden_size, features))
        self.model = unit

    def _create(self, device_index):
        kind = blobs.RNN_KINDS[self._cell]
        # hidden sizes between the kernels' (64 / 128 / 256 / 512) run with zero units added
        # and feature counts that are no multiple of 4 with zero features added
        hp, fp = blobs.padded_hidden(self.hidden_size), blobs.padded_features(self.features)
        blob = np.ascontiguousarray(blobs.pack_rnn(self.state_dict(), kind, self.features,
                                                   self.input_size, self.hidden_size,
                                                   self.num_layers, pad_to=hp, pad_features=fp))
        h = ctypes.c_void_p()
        rc = _cabi.lib().cet_rnn_create(kind, fp, self.input_size, hp,
                                        self.num_layers, blob.ctypes.data_as(ctypes.c_void_p),
                                        blob.size, ctypes.byref(h))
        _cabi.check(rc, "cet_rnn_create")
        choice = self._kernel_choice or {"t": 1, "s": 2}.get(os.environ.get("CET_RNN_KERNEL", "")[:1], 0)
        if choice:
            _cabi.check(_cabi.lib().cet_rnn_select(h, choice), "cet_rnn_select")
        return h, None

    _KERNELS = {"auto": 0, "tiled": 1, "split": 2, "split_giveup": 3, "half_tile": 4, "half_tile_2": 5, "half_tile_4": 6,
                "half_tile_2_giveup": 7, "half_tile_8": 8}

    def select_kernel(self, kernel):
        """Pins the rollout kernel (tests, timing): "tiled" (a workgroup takes 32 sequences and all
        hidden units), "split" (small batches: 16 sequences and 4, 8 or 16 hidden units per workgroup,
        hidden states exchanged once per layer and step), "half_tile" (the tiled design with 16
        sequences per workgroup, for batches of up to 16 per CU; "half_tile_2" / "half_tile_4": a tile's
        hidden units on 2 / 4 workgroups, for up to 8 / 4 per CU), "auto" (by estimated time, the default) or
        "split_giveup" (tests: the hidden-split kernel abandons its exchange at once, so that the
        stand-by launch of the tiled kernel behind it computes the result)."""
        self._kernel_choice = self._KERNELS[kernel]
        for _v, h, _x in self._handles.values():
            _cabi.check(_cabi.lib().cet_rnn_select(ctypes.c_void_p(h), self._kernel_choice), "cet_rnn_select")
        return self

    def test_data(self, x, pred_len, device=None):
        """`[B, seq_len, features]` -> `[B, pred_len, features]`: seq_len teacher-forced steps,
        then closed-loop prediction (LocalSimulations/models/model.py:298-312, 373-388,
        449-465). `device` is accepted for signature compatibility; the rollout runs where
        `x` lives."""
        if x.dim() != 3 or x.shape[2] != self.features:
            raise RuntimeError(f"x must be [B, seq_len, {self.features}], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("x is on the CPU: this implementation runs on ROCm GPUs only")
        fp = blobs.padded_features(self.features)
        if fp != self.features:
            x = torch.nn.functional.pad(x, (0, fp - self.features))
            return torch.ops.cet.rnn_rollout(x, self._handle_for(x.device)[0], int(pred_len))[:, :, :self.features].contiguous()
        return torch.ops.cet.rnn_rollout(x, self._handle_for(x.device)[0], int(pred_len))

    def train_data(self, x, device=None):
        """Training is outside this inference path: the two teacher-forced variants of the
        reference (LocalSimulations/models/model.py:285-297 and the output-aligned one of
        HPC/python/models/model.py:285-296) are not provided."""
        raise NotImplementedError(
            "train_data belongs to the reference's training loops, which this inference-only "
            "path does not provide; train with the reference and load the checkpoint here")


class RNN(_RecurrentBase):
    """LocalSimulations/models/model.py:272-312."""
    _cell = "rnn"


class GRU(_RecurrentBase):
    """LocalSimulations/models/model.py:346-388."""
    _cell = "gru"


class LSTM(_RecurrentBase):
    """LocalSimulations/models/model.py:422-465."""
    _cell = "lstm"
