"""Recurrent rollout of two builds of the library (argv: libA libB) against each other and
against the CPU checker, per predicted step: shows whether a difference is rounding noise that
grows along the closed loop or a jump."""
import os, subprocess, sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

if len(sys.argv) == 4:  # child: lib, kind name, output file
    os.environ["CET_LIB"] = sys.argv[1]
    import torch
    import channelestimationtransformer_amd as cet
    from channelestimationtransformer_amd import synth
    from test_gpu_rnn import build
    m = build(sys.argv[2])
    noisy, _ = synth.channel_histories(48, seed=9)
    x = cet.LoadBatch(noisy[:, :25]).cuda()
    np.save(sys.argv[3], m.test_data(x, 100, "cuda").cpu().numpy())
else:
    import channelestimationtransformer_amd as cet
    from channelestimationtransformer_amd import blobs, synth
    from conftest import load_npz, sub_dict
    from test_gpu_rnn import RNN_CASES
    from oracle import pyoracle
    noisy, _ = synth.channel_histories(48, seed=9)
    x = cet.LoadBatch(noisy[:, :25]).numpy()
    for name in ("LSTM", "GRU", "RNN"):
        _cls, kind, F, I, H, layers = RNN_CASES[name]
        blob = blobs.pack_rnn(sub_dict(load_npz("rnn_ckpt.npz"), name), kind, F, I, H, layers)
        want = pyoracle.rnn_rollout(kind, F, I, H, layers, blob, x, 100)
        outs = []
        for i, lib in enumerate(sys.argv[1:3]):
            f = f"/tmp/rnn_cmp_{i}.npy"
            subprocess.run([sys.executable, __file__, os.path.abspath(lib), name, f], check=True, timeout=200)
            outs.append(np.load(f))
        steps = [0, 1, 2, 5, 10, 20, 40, 60, 80, 99]
        print(name, "A vs B    ", " ".join(f"{np.abs(outs[0] - outs[1])[:, t].max():.1e}" for t in steps))
        print(name, "A vs check", " ".join(f"{np.abs(outs[0] - want)[:, t].max():.1e}" for t in steps))
        print(name, "B vs check", " ".join(f"{np.abs(outs[1] - want)[:, t].max():.1e}" for t in steps), flush=True)
