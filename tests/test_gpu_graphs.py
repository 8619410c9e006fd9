"""The forward calls under HIP graph capture (serving loops replay a captured step): the library
allocates nothing on these paths and launches on the capturing stream, so a captured step replays
with new inputs in the same buffers."""
import pytest
import torch

from conftest import DEFAULT_ARGS, load_npz
import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import synth
from test_gpu_informer import build as build_informer
from test_gpu_rnn import build as build_rnn

pytestmark = pytest.mark.gpu


def _inputs(B, seed):
    noisy, _ = synth.channel_histories(B, seed=seed)
    return cet.LoadBatch(noisy[:, :25]).cuda()


@pytest.mark.parametrize("B", [96, 2500])  # one-wave kernel, workgroup kernel
def test_transformer_forward_replays_from_a_graph(B):
    inf = build_informer("InformerStack", DEFAULT_ARGS, dict(load_npz("informer_ckpt.npz"))).cuda()
    x, x2 = _inputs(B, 1), _inputs(B, 2)
    xd, xd2 = cet.decoder_input(x, 10, 5), cet.decoder_input(x2, 10, 5)
    want1, want2 = inf(x, xd), inf(x2, xd2)
    sx, sxd = x.clone(), xd.clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pred, attns = inf(sx, sxd)
    for (xi, xdi), want in (((x, xd), want1), ((x2, xd2), want2), ((x, xd), want1)):
        sx.copy_(xi)
        sxd.copy_(xdi)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(pred, want[0])
        assert torch.equal(attns[0][0], want[1][0][0]) and torch.equal(attns[1][-1], want[1][1][-1])


@pytest.mark.parametrize("name", ["GRU", "LSTM", "RNN"])
@pytest.mark.parametrize("B", [40, 1500, 5000])
def test_recurrent_rollout_replays_from_a_graph(name, B):
    """Under capture the automatic choice keeps to the kernels that need no exchange buffer of the
    pool (their launches would be replayed outside its bookkeeping)."""
    m = build_rnn(name)
    x, x2 = _inputs(B, 3), _inputs(B, 4)
    want1, want2 = m.test_data(x, 5, "cuda"), m.test_data(x2, 5, "cuda")
    sx = x.clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.test_data(sx, 5, "cuda")
    scale = max(1.0, float(want1.abs().max()))
    for xi, want in ((x, want1), (x2, want2), (x, want1)):
        sx.copy_(xi)
        g.replay()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all() and (out - want).abs().max() < 1e-4 * scale  # another kernel than outside capture


def test_equalizer_replays_from_a_graph():
    y, H, _ = synth.lmmse_problem(30_000, 16, 4, 0.1, seed=5)
    y2, H2, _ = synth.lmmse_problem(30_000, 16, 4, 0.1, seed=6)
    y, H, y2, H2 = (torch.from_numpy(a).cuda() for a in (y, H, y2, H2))
    want1, want2 = cet.lmmse(y, H, 0.1), cet.lmmse(y2, H2, 0.1)
    sy, sH = y.clone(), H.clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        xh, ne = cet.lmmse(sy, sH, 0.1)
    for (yi, Hi), want in (((y, H), want1), ((y2, H2), want2)):
        sy.copy_(yi)
        sH.copy_(Hi)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(xh), torch.view_as_real(want[0])) and torch.equal(ne, want[1])
