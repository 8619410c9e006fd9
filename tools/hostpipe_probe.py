"""Where the time of a host-fed Transformer step goes: transfers alone, forward alone, serial,
pipelined (HostPipeline). python tools/hostpipe_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import channelestimationtransformer_amd as cet

dev = torch.device("cuda")
torch.manual_seed(0)
model = cet.InformerStack(16, 16, 16, 25, 10, 5, 5, 64, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu",
                          False, True, "cuda").cuda().eval()
model_maps = cet.InformerStack(16, 16, 16, 25, 10, 5, 5, 64, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu",
                               True, True, "cuda").cuda().eval()
B = 4096
shapes = [(B, 25, 16), (B, 15, 16)]


def wall(fn, n=40):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def pipe_run(step, depth=3):
    pipe = cet.HostPipeline(step, shapes, (B, 5, 16), depth=depth)
    for sl in pipe.slots:
        for h in sl.host_in:
            h.normal_()
    return lambda n: pipe.run(n, lambda i, h: None, lambda i, o: None)


out_dev = torch.zeros(B, 5, 16, device=dev)
print("pipeline, forward replaced by a no-op: %.3f ms" % wall(pipe_run(lambda a, b: out_dev)))
print("pipeline depth 2: %.3f ms" % wall(pipe_run(lambda a, b: model(a, b), 2)))
print("pipeline depth 3: %.3f ms" % wall(pipe_run(lambda a, b: model(a, b), 3)))
print("pipeline depth 4: %.3f ms" % wall(pipe_run(lambda a, b: model(a, b), 4)))
for d in (2, 3, 4):
    print("with attention maps, pipeline depth %d: %.3f ms" % (d, wall(pipe_run(lambda a, b: model_maps(a, b), d), 100)))
print("depth 3 again, 200 batches: %.3f ms" % wall(pipe_run(lambda a, b: model(a, b), 3), 200))
xh = [torch.randn(s).pin_memory() for s in shapes]
oh = torch.empty(B, 5, 16).pin_memory()
xd = [t.cuda() for t in xh]


def serial(n):
    for _ in range(n):
        r = model(xh[0].to(dev, non_blocking=True), xh[1].to(dev, non_blocking=True))
        oh.copy_(r, non_blocking=True)


def fwd_only(n):
    for _ in range(n):
        model(xd[0], xd[1])


def copies_only(n):
    for _ in range(n):
        xd[0].copy_(xh[0], non_blocking=True)
        xd[1].copy_(xh[1], non_blocking=True)
        oh.copy_(out_dev, non_blocking=True)


print("serial on one stream: %.3f ms" % wall(serial))
print("forward only: %.3f ms" % wall(fwd_only))
print("copies only (one stream): %.3f ms" % wall(copies_only))
s2 = torch.cuda.Stream()


def fwd_and_copies_two_streams(n):
    for _ in range(n):
        model(xd[0], xd[1])
        with torch.cuda.stream(s2):
            xd2[0].copy_(xh[0], non_blocking=True)
            xd2[1].copy_(xh[1], non_blocking=True)


xd2 = [torch.empty_like(t) for t in xd]
print("forward + independent uploads on a second stream: %.3f ms" % wall(fwd_and_copies_two_streams))
