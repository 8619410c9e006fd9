"""Host-resident batches through the three-stream pipeline give the results of the plain calls
(LocalSimulations/Predict.py:99-117 is the loop it replaces)."""
import numpy as np
import pytest
import torch

import channelestimationtransformer_amd as cet
from channelestimationtransformer_amd import synth

pytestmark = pytest.mark.gpu


def _histories(n, B, seed):
    out = []
    for i in range(n):
        noisy, _ = synth.channel_histories(B, T=30, seed=seed + i)
        out.append(cet.LoadBatch(noisy[:, :25]).float())
    return out


@pytest.mark.parametrize("depth", [2, 3])
def test_transformer_batches_from_host(depth):
    torch.manual_seed(0)
    model = cet.InformerStack(16, 16, 16, 25, 10, 5, 5, 64, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu",
                              False, True, "cuda").cuda().eval()
    B, n = 37, 7
    enc = _histories(n, B, 100)
    dec = [cet.decoder_input(e, 10, 5) for e in enc]
    want = [model(e.cuda(), d.cuda()).cpu() for e, d in zip(enc, dec)]
    got = {}

    def fill(i, host):
        host[0].copy_(enc[i])
        host[1].copy_(dec[i])

    def consume(i, out):
        got[i] = out.clone()

    pipe = cet.HostPipeline(lambda xe, xd: model(xe, xd), [(B, 25, 16), (B, 15, 16)], (B, 5, 16), depth=depth)
    pipe.run(n, fill, consume)
    assert sorted(got) == list(range(n))
    for i in range(n):
        assert torch.equal(got[i], want[i])
    pipe.run(2, fill, consume)  # a pipeline can be run again
    assert torch.equal(got[1], want[1])


def test_recurrent_batches_from_host():
    torch.manual_seed(1)
    model = cet.GRU(16, 16, 256, 2).cuda().eval()
    B, n = 20, 5
    xs = _histories(n, B, 300)
    want = [model.test_data(x.cuda(), 5, "cuda").cpu() for x in xs]
    got, order = {}, []

    def consume(i, out):
        order.append(i)
        got[i] = out.clone()

    pipe = cet.HostPipeline(lambda x: model.test_data(x, 5, "cuda"), [(B, 25, 16)], (B, 5, 16))
    pipe.run(n, lambda i, host: host[0].copy_(xs[i]), consume)
    assert order == list(range(n))
    for i in range(n):
        assert torch.equal(got[i], want[i])
