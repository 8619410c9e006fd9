"""Second, independent CPU statement of the predictors in plain `torch.nn.functional` calls.

TEST INFRASTRUCTURE ONLY, like oracle/cet_oracle.c: used by tests/ as a checker that shares no
code with the C restatement or the HIP kernels, and by bench.py as the "stock PyTorch on the host
cores" baseline. It works on a reference-style state_dict (name -> tensor or numpy array) and never touches
the GPU library; with the state_dict and the inputs on a ROCm device it runs stock PyTorch-ROCm
there (the full-size cross-check of tests/test_gpu_torch_ref.py).

The functions follow the reference's modules:
  embedding     LocalSimulations/models/embed.py:7-53 (circular Conv1d k=3 + stored `pe`)
  attention     LocalSimulations/models/attn.py:18-36, 121-156 (mask before scale, d_keys = d_model // H)
  encoder       LocalSimulations/models/encoder.py:5-101 (distil ConvLayer, stack quirk at :96)
  decoder       LocalSimulations/models/decoder.py:20-51
  model         LocalSimulations/models/model.py:67-81, 220-234
  recurrent     LocalSimulations/models/model.py:253-270, 298-312 (and the GRU / LSTM twins)
"""
import math

import torch
import torch.nn.functional as F


def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def _embed(sd, pre, x):
    w, b = _t(sd, pre + ".value_embedding.tokenConv.weight"), _t(sd, pre + ".value_embedding.tokenConv.bias")
    y = F.conv1d(F.pad(x.permute(0, 2, 1), (1, 1), mode="circular"), w, b).transpose(1, 2)
    return y + _t(sd, pre + ".position_embedding.pe")[:, :x.shape[1]]


def _attention(sd, pre, q_in, kv_in, n_heads, causal):
    B, L, _ = q_in.shape
    S = kv_in.shape[1]
    lin = lambda n, v: F.linear(v, _t(sd, f"{pre}.{n}_projection.weight"), _t(sd, f"{pre}.{n}_projection.bias"))
    q = lin("query", q_in).view(B, L, n_heads, -1)
    k = lin("key", kv_in).view(B, S, n_heads, -1)
    v = lin("value", kv_in).view(B, S, n_heads, -1)
    scores = torch.einsum("blhe,bshe->bhls", q, k)
    if causal:
        scores = scores.masked_fill(torch.triu(torch.ones(L, S, dtype=torch.bool, device=scores.device), 1), -math.inf)
    a = torch.softmax(scores / math.sqrt(q.shape[-1]), dim=-1)
    ctx = torch.einsum("bhls,bshd->blhd", a, v).reshape(B, L, -1)
    return lin("out", ctx), a


def _ffn(sd, pre, x, act):
    y = F.conv1d(x.transpose(1, 2), _t(sd, pre + ".conv1.weight"), _t(sd, pre + ".conv1.bias"))
    y = F.conv1d(act(y), _t(sd, pre + ".conv2.weight"), _t(sd, pre + ".conv2.bias"))
    return y.transpose(1, 2)


def _ln(sd, pre, x):
    return F.layer_norm(x, x.shape[-1:], _t(sd, pre + ".weight"), _t(sd, pre + ".bias"), 1e-5)


def _encoder(sd, pre, x, n_layers, distil, n_heads, act, attns):
    for i in range(n_layers):
        lp = f"{pre}.attn_layers.{i}"
        y, a = _attention(sd, lp + ".attention", x, x, n_heads, False)
        attns.append(a)
        x = _ln(sd, lp + ".norm1", x + y)
        x = _ln(sd, lp + ".norm2", x + _ffn(sd, lp, x, act))
        if distil and i < n_layers - 1:
            cp = f"{pre}.conv_layers.{i}"
            y = F.conv1d(F.pad(x.permute(0, 2, 1), (2, 2), mode="circular"),
                         _t(sd, cp + ".downConv.weight"), _t(sd, cp + ".downConv.bias"))
            y = F.batch_norm(y, _t(sd, cp + ".norm.running_mean"), _t(sd, cp + ".norm.running_var"),
                             _t(sd, cp + ".norm.weight"), _t(sd, cp + ".norm.bias"), False, 0.0, 1e-5)
            x = F.max_pool1d(F.elu(y), 3, 2, 1).transpose(1, 2)
    return _ln(sd, pre + ".norm", x)


def informer_forward(sd, x_enc, x_dec, pred_len, n_heads=8, e_layers=4, d_layers=3, stack=True,
                     distil=True, activation="gelu"):
    """-> (pred [B, pred_len, c_out], list of encoder self-attention maps in execution order)."""
    act = F.gelu if activation == "gelu" else F.relu
    x_enc, x_dec = torch.as_tensor(x_enc), torch.as_tensor(x_dec)
    attns = []
    x = _embed(sd, "enc_embedding", x_enc)
    if stack:
        outs = []
        for j, n in enumerate(range(e_layers, 2, -1)):
            # the loop variable is overwritten in the reference (encoder.py:96): encoder j + 1
            # consumes the (whole) output of encoder j
            x = _encoder(sd, f"encoder.encoders.{j}", x, n, distil, n_heads, act, attns)
            outs.append(x)
        mem = torch.cat(outs, dim=1)
    else:
        mem = _encoder(sd, "encoder", x, e_layers, distil, n_heads, act, attns)
    y = _embed(sd, "dec_embedding", x_dec)
    for i in range(d_layers):
        lp = f"decoder.layers.{i}"
        y = _ln(sd, lp + ".norm1", y + _attention(sd, lp + ".self_attention", y, y, n_heads, True)[0])
        y = _ln(sd, lp + ".norm2", y + _attention(sd, lp + ".cross_attention", y, mem, n_heads, False)[0])
        y = _ln(sd, lp + ".norm3", y + _ffn(sd, lp, y, act))
    y = _ln(sd, "decoder.norm", y)
    y = F.linear(y, _t(sd, "projection.weight"), _t(sd, "projection.bias"))
    return y[:, -pred_len:], attns


def rnn_rollout(sd, cell, x, pred_len):
    """cell in {'rnn', 'gru', 'lstm'}; x [B, seq_len, F] -> [B, pred_len, F] with a stock torch
    recurrent module carrying the state_dict's weights (teacher-forced, then closed loop)."""
    x = torch.as_tensor(x)
    w = lambda k: _t(sd, k)
    layers = sum(1 for k in sd if k.startswith(f"model.{cell}.weight_hh_l"))
    hidden = w(f"model.{cell}.weight_hh_l0").shape[1]
    inp = w("model.encoder.0.weight").shape[0]
    mod = {"rnn": torch.nn.RNN, "gru": torch.nn.GRU, "lstm": torch.nn.LSTM}[cell](inp, hidden, layers)
    mod.load_state_dict({k[len(f"model.{cell}."):]: w(k) for k in sd if k.startswith(f"model.{cell}.")})
    mod = mod.to(x.device)
    B, seq_len, _ = x.shape
    state = torch.zeros(layers, B, hidden, device=x.device)
    if cell == "lstm":
        state = (state, state.clone())
    outs, out = [], None
    with torch.no_grad():
        for idx in range(seq_len + pred_len - 1):
            step = x[:, idx:idx + 1].permute(1, 0, 2) if idx < seq_len else out
            h, state = mod(F.linear(step, w("model.encoder.0.weight"), w("model.encoder.0.bias")), state)
            out = F.linear(h, w("model.decoder.0.weight"), w("model.decoder.0.bias"))
            if idx >= seq_len - 1:
                outs.append(out)
    return torch.cat(outs, dim=0).permute(1, 0, 2).contiguous()
