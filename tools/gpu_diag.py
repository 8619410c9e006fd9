"""Prints per-stage differences between the HIP kernels and the golden vectors (run on the
GPU box while bringing a kernel up; the pytest suite asserts the same things)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import DEFAULT_ARGS, INFORMER_VARIANTS, RNN_CASES, load_npz, sub_dict  # noqa: E402
from oracle import pyoracle  # noqa: E402
import channelestimationtransformer_amd as cet  # noqa: E402
from channelestimationtransformer_amd import synth  # noqa: E402
import test_gpu_informer as ti  # noqa: E402
import test_gpu_rnn as tr  # noqa: E402


def main():
    which = sys.argv[1:] or ["lmmse", "informer", "rnn"]
    if "lmmse" in which:
        for Nr, Ns in ((16, 4), (8, 4), (4, 2), (2, 1)):
            y, H, _ = synth.lmmse_problem(1000, Nr, Ns, 0.1, seed=1)
            wx, wne = pyoracle.lmmse(y, H, 0.1)
            for direct in (True, False):
                x, ne = cet.lmmse(torch.from_numpy(y).cuda(), torch.from_numpy(H).cuda(), 0.1, direct=direct)
                print(f"lmmse {Nr}x{Ns} direct={direct}: x err {np.abs(x.cpu().numpy() - wx).max():.3e} "
                      f"no_eff err {np.abs(ne.cpu().numpy() - wne).max():.3e}", flush=True)
    if "informer" in which:
        ref = load_npz("informer_ref.npz")
        m = ti.build("InformerStack", DEFAULT_ARGS, dict(load_npz("informer_ckpt.npz")))
        pred, taps, attn = ti.run_taps(m, ref["x_enc"], ref["x_dec"])
        rows = pyoracle.tap_rows(m._cfg)
        for i, t in enumerate(pyoracle.split_taps(taps, rows, 8, 64)):
            d = np.abs(t - ref[f"tap_{i}"])
            print(f"informer stage {i:2d} rows {rows[i]:2d}: max err {np.nanmax(d):.3e} nan {np.isnan(t).sum()} "
                  f"worst row {np.unravel_index(np.nanargmax(d), d.shape)}", flush=True)
        for i, a in enumerate(pyoracle.split_attn(attn, pyoracle.attn_lengths(m._cfg), 8, 8)):
            print(f"informer attn {i}: max err {np.nanmax(np.abs(a - ref[f'attn_{i}'])):.3e} nan {np.isnan(a).sum()}")
        print(f"informer pred: max err {np.abs(pred - ref['pred']).max():.3e}", flush=True)
        v = load_npz("informer_variants.npz")
        for name, (cls, args) in INFORMER_VARIANTS.items():
            mm = ti.build(cls, args, sub_dict(v, f"{name}/sd"))
            p, tp, _ = ti.run_taps(mm, v[f"{name}/x_enc"], v[f"{name}/x_dec"])
            rw = pyoracle.tap_rows(mm._cfg)
            errs = [float(np.nanmax(np.abs(t - v[f"{name}/tap_{i}"]))) for i, t in
                    enumerate(pyoracle.split_taps(tp, rw, p.shape[0], 64))]
            print(f"{name}: pred err {np.abs(p - v[f'{name}/pred']).max():.3e} stage errs "
                  + " ".join(f"{e:.1e}" for e in errs), flush=True)
    if "rnn" in which:
        ref = load_npz("rnn_ref.npz")
        for name in RNN_CASES:
            m = tr.build(name)
            x = ref["x"] if f"{name}/x" not in ref.files else ref[f"{name}/x"]
            for P in (5, 25):
                key = f"{name}/pred{P}"
                if key not in ref.files:
                    continue
                out = m.test_data(torch.from_numpy(x).cuda(), P, "cuda").cpu().numpy()
                d = np.abs(out - ref[key]).max(axis=(0, 2))
                print(f"rnn {name} P={P}: per-step max err " + " ".join(f"{e:.1e}" for e in d), flush=True)


if __name__ == "__main__":
    main()
