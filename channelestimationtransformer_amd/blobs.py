"""Flattening of reference-style state_dicts into the fp32 blobs the C ABI takes.

The order is the one documented next to `cet_informer_blob_size` / `cet_rnn_blob_size` in
include/cet.h. All tensors keep their PyTorch layout; any re-packing for the kernels happens
inside the library.
"""
import numpy as np

from ._cabi import InformerCfg

RNN_KINDS = {"rnn": 0, "gru": 1, "lstm": 2}
RNN_GATES = {0: 1, 1: 3, 2: 4}


def normalize_state_dict(sd):
    """Accepts the three on-disk conventions of the reference's checkpoints.

    `{'state_dict': ...}` wrapper (LocalSimulations/Predict.py:90-91), bare dict with the
    `module.` prefix DataParallel adds (HPC/python/Evaluate.py:71-73), bare dict without
    prefix (HPC/python/Evaluate.py:78-79).
    """
    if "state_dict" in sd and not hasattr(sd["state_dict"], "shape"):
        sd = sd["state_dict"]
    out = {}
    for k, v in sd.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out


def _np(t):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def encoder_depths(e_layers, stack):
    """Layer counts of the encoders: `list(range(e_layers, 2, -1))` for the stacked models
    (LocalSimulations/models/model.py:178), one encoder otherwise."""
    return list(range(e_layers, 2, -1)) if stack else [e_layers]


def make_informer_cfg(enc_in, dec_in, c_out, seq_len, label_len, out_len, d_model, n_heads,
                      e_layers, d_layers, d_ff, stack, distil, activation):
    return InformerCfg(enc_in=enc_in, dec_in=dec_in, c_out=c_out, seq_len=seq_len,
                       label_len=label_len, pred_len=out_len, d_model=d_model,
                       n_heads=n_heads, e_layers=e_layers, d_layers=d_layers, d_ff=d_ff,
                       stack=int(bool(stack)), distil=int(bool(distil)),
                       activation=0 if activation == "relu" else 1)


def padded_informer_cfg(cfg):
    """The configuration the kernels run for `cfg`: enc_in / dec_in of 16 or 32 input features and a
    c_out that is a multiple of 4 (the kernels' widths). The added input features have zero
    token-convolution weights and the added outputs zero projection rows and biases
    (pack_informer(sd, cfg, dev_cfg)); the model classes add zero columns to the inputs and drop
    the added outputs. Equal to `cfg` for the widths the kernels take as they are."""
    def width(n, what):
        if n <= 16:
            return 16
        if n <= 32:
            return 32
        raise ValueError(f"{what} {n}: the Transformer kernel takes up to 32 input features")
    d = InformerCfg.from_buffer_copy(cfg)
    d.enc_in, d.dec_in = width(cfg.enc_in, "enc_in"), width(cfg.dec_in, "dec_in")
    d.c_out = (cfg.c_out + 3) // 4 * 4
    return d


def pack_informer(sd, cfg, dev_cfg=None):
    """state_dict (reference key names) -> 1-D float32 array in the C ABI's order; with `dev_cfg`
    (padded_informer_cfg) laid out for its enc_in / dec_in / c_out, zeros for what is added."""
    sd = normalize_state_dict(sd)
    dev_cfg = dev_cfg or cfg
    parts = []

    def put(key, shape=None):
        a = _np(sd[key])
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: shape {tuple(a.shape)} != expected {tuple(shape)}")
        parts.append(a.reshape(-1))

    D, F = cfg.d_model, cfg.d_ff
    HE = (D // cfg.n_heads) * cfg.n_heads  # d_keys * n_heads (attn.py:121-138)
    for emb, cin, L in (("enc_embedding", cfg.enc_in, cfg.seq_len),
                        ("dec_embedding", cfg.dec_in, cfg.label_len + cfg.pred_len)):
        w = _np(sd[f"{emb}.value_embedding.tokenConv.weight"])
        if tuple(w.shape) != (D, cin, 3):
            raise ValueError(f"{emb}.value_embedding.tokenConv.weight: shape {tuple(w.shape)} != expected {(D, cin, 3)}")
        cpad = dev_cfg.enc_in if emb == "enc_embedding" else dev_cfg.dec_in
        wp = np.zeros((D, cpad, 3), np.float32)
        wp[:, :cin] = w
        parts.append(wp.reshape(-1))
        put(f"{emb}.value_embedding.tokenConv.bias", (D,))
        pe = _np(sd[f"{emb}.position_embedding.pe"])
        parts.append(np.ascontiguousarray(pe.reshape(-1, D)[:L]).reshape(-1))

    def put_attn(prefix):
        for proj in ("query", "key", "value"):
            put(f"{prefix}.{proj}_projection.weight", (HE, D))
            put(f"{prefix}.{proj}_projection.bias", (HE,))
        put(f"{prefix}.out_projection.weight", (D, HE))
        put(f"{prefix}.out_projection.bias", (D,))

    def put_ffn(prefix):
        # Conv1d(kernel_size=1) weights are [out, in, 1]
        put(f"{prefix}.conv1.weight", (F, D, 1))
        put(f"{prefix}.conv1.bias", (F,))
        put(f"{prefix}.conv2.weight", (D, F, 1))
        put(f"{prefix}.conv2.bias", (D,))

    depths = encoder_depths(cfg.e_layers, cfg.stack)
    for ei, el in enumerate(depths):
        enc = f"encoder.encoders.{ei}" if cfg.stack else "encoder"
        for l in range(el):
            p = f"{enc}.attn_layers.{l}"
            put_attn(f"{p}.attention")
            put_ffn(p)
            for n in ("norm1", "norm2"):
                put(f"{p}.{n}.weight", (D,))
                put(f"{p}.{n}.bias", (D,))
        if cfg.distil:
            for l in range(el - 1):
                p = f"{enc}.conv_layers.{l}"
                put(f"{p}.downConv.weight", (D, D, 3))
                put(f"{p}.downConv.bias", (D,))
                for n in ("weight", "bias", "running_mean", "running_var"):
                    put(f"{p}.norm.{n}", (D,))
        put(f"{enc}.norm.weight", (D,))
        put(f"{enc}.norm.bias", (D,))
    for l in range(cfg.d_layers):
        p = f"decoder.layers.{l}"
        put_attn(f"{p}.self_attention")
        put_attn(f"{p}.cross_attention")
        put_ffn(p)
        for n in ("norm1", "norm2", "norm3"):
            put(f"{p}.{n}.weight", (D,))
            put(f"{p}.{n}.bias", (D,))
    put("decoder.norm.weight", (D,))
    put("decoder.norm.bias", (D,))
    pw, pb = _np(sd["projection.weight"]), _np(sd["projection.bias"])
    if tuple(pw.shape) != (cfg.c_out, D) or tuple(pb.shape) != (cfg.c_out,):
        raise ValueError(f"projection: shapes {tuple(pw.shape)}, {tuple(pb.shape)} != expected {(cfg.c_out, D)}, {(cfg.c_out,)}")
    pwp, pbp = np.zeros((dev_cfg.c_out, D), np.float32), np.zeros((dev_cfg.c_out,), np.float32)
    pwp[:cfg.c_out], pbp[:cfg.c_out] = pw, pb
    parts += [pwp.reshape(-1), pbp]
    return np.concatenate(parts)


def informer_blob_len(cfg):
    """Host-side recomputation of `cet_informer_blob_size` (cross-checked in the tests)."""
    D, F = cfg.d_model, cfg.d_ff
    n = D * cfg.enc_in * 3 + D + cfg.seq_len * D
    n += D * cfg.dec_in * 3 + D + (cfg.label_len + cfg.pred_len) * D
    HE = (D // cfg.n_heads) * cfg.n_heads
    attn = 3 * (HE * D + HE) + D * HE + D
    ffn = F * D + F + D * F + D
    for el in encoder_depths(cfg.e_layers, cfg.stack):
        n += el * (attn + ffn + 4 * D)
        if cfg.distil:
            n += (el - 1) * (3 * D * D + D + 4 * D)
        n += 2 * D
    n += cfg.d_layers * (2 * attn + ffn + 6 * D)
    n += 2 * D + cfg.c_out * D + cfg.c_out
    return n


RNN_HIDDEN_SIZES = (64, 128, 256, 512)  # hidden sizes the kernels are instantiated for


def padded_hidden(hidden_size):
    """The kernels' hidden size that serves `hidden_size`: the next of 64 / 128 / 256 / 512. The
    units in between are padding (see pack_rnn)."""
    for h in RNN_HIDDEN_SIZES:
        if hidden_size <= h:
            return h
    raise ValueError(f"hidden_size {hidden_size}: the recurrent kernels take up to {RNN_HIDDEN_SIZES[-1]} hidden units")


def padded_features(features):
    """The recurrent kernels read rows of a multiple of 4 features, up to 32."""
    fp = (features + 3) // 4 * 4
    if fp > 32:
        raise ValueError(f"features {features}: the recurrent kernels take up to 32 features")
    return fp


def pack_rnn(sd, kind, features, input_size, hidden_size, num_layers, pad_to=None, pad_features=None):
    """state_dict of RNN/GRU/LSTM (keys `model.encoder.0.*`, `model.<cell>.weight_ih_l0` ...)
    -> 1-D float32 array in the C ABI's order.

    pad_to: lay the parameters out for `pad_to` >= hidden_size hidden units. The added units have
    zero weights and biases on every side: their gates are sigma(0) / tanh(0), so starting from the
    zero state they stay at exactly zero (tanh cell: tanh(0); GRU: 0.5 * 0 + 0.5 * h; LSTM: c = 0.5
    c + 0.5 * 0, h = 0.5 tanh(c)), and nothing reads them through the zero columns. The real units'
    sums gain terms of exactly zero: same results as the unpadded model.
    pad_features: lay the encoder and decoder Linear out for `pad_features` >= features (zero encoder
    columns, zero decoder rows and biases): the added outputs are exactly zero and feed zeros back."""
    sd = normalize_state_dict(sd)
    cell = {0: "rnn", 1: "gru", 2: "lstm"}[kind]
    G = RNN_GATES[kind]
    H, Hp = hidden_size, pad_to or hidden_size
    if Hp < H:
        raise ValueError(f"pad_to {Hp} < hidden_size {H}")
    parts = []

    def get(key, shape):
        a = _np(sd[key])
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: shape {tuple(a.shape)} != expected {tuple(shape)}")
        return a

    def gates(a, cols):
        """[G*H, ...] -> [G*Hp, cols]: every gate's rows at g * Hp, zero rows and columns behind."""
        a = a.reshape(G, H, -1)
        out = np.zeros((G, Hp, cols), np.float32)
        out[:, :H, :a.shape[2]] = a
        return out.reshape(-1)

    Fp = pad_features or features
    if Fp < features:
        raise ValueError(f"pad_features {Fp} < features {features}")
    enc = np.zeros((input_size, Fp), np.float32)
    enc[:, :features] = get("model.encoder.0.weight", (input_size, features))
    parts.append(enc.reshape(-1))
    parts.append(get("model.encoder.0.bias", (input_size,)).reshape(-1))
    for l in range(num_layers):
        inp = input_size if l == 0 else H
        parts.append(gates(get(f"model.{cell}.weight_ih_l{l}", (G * H, inp)), input_size if l == 0 else Hp))
        parts.append(gates(get(f"model.{cell}.weight_hh_l{l}", (G * H, H)), Hp))
        parts.append(gates(get(f"model.{cell}.bias_ih_l{l}", (G * H,)), 1))
        parts.append(gates(get(f"model.{cell}.bias_hh_l{l}", (G * H,)), 1))
    dec, dec_b = np.zeros((Fp, Hp), np.float32), np.zeros((Fp,), np.float32)
    dec[:features, :H] = get("model.decoder.0.weight", (features, H))
    dec_b[:features] = get("model.decoder.0.bias", (features,))
    parts += [dec.reshape(-1), dec_b]
    return np.concatenate(parts)
