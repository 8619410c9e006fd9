"""Throughput benchmark of the predictor + LMMSE inference path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--workload informer|informer12|lmmse|lstm|gru|rnn|e2e|selftest] [--batch B]
                    [--pred-len P] [--h-share n] [--geometry reference|scaled]
                    [--no-attention-maps]

One "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM. The default workload is the Transformer predictor forward on 4096 channel
sequences per GPU (InformerStack, the constructor arguments of LocalSimulations/Predict.py:
21-38, shipped checkpoint weights). With N > 1 every rank (one process per GPU) runs the same
per-GPU batch (weak scaling); the only collective is one RCCL all-reduce of the NMSE partial sums
at the end of the timed region. Under torch.distributed.run (WORLD_SIZE set) the process is one
rank of N; started without a launcher, `--gpus N` makes it start the N ranks itself, as a child
process tree and before it touches a GPU, and exit with their status (this replaces the
reference's torch.nn.DataParallel wrapper, HPC/python/Evaluate.py:74). Rank 0 prints ONE JSON
line. `--workload selftest` runs the launcher, the rendezvous and the all-reduce on the CPU with
the gloo backend (no GPU work; used by the CPU tests).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import channelestimationtransformer_amd as cet  # noqa: E402
from channelestimationtransformer_amd import blobs, e2e, metrics, synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Algorithmic work / bytes per unit (dense contractions only for the FLOP figures).
INFORMER_FLOP = 8.897e6          # per prediction, default configuration


def informer_flop(seq, label, pred, D=64, H=8, e_layers=4, d_layers=3, F=64, cin=16, cout=16, stack=True,
                  distil=True):
    """GEMM-type work of one forward (multiply-adds x 2), counted as for the default configuration's
    8.897 MFLOP: embeddings (3-tap conv), per attention layer Q/K/V/O + scores + context, FFN, the
    distil conv on its L + 2 output rows, the projection of all decoder rows."""
    HE, Ld = (D // H) * H, label + pred
    f = 2 * (seq + Ld) * 3 * cin * D
    cur, inp, mem = seq, seq, 0
    for el in (range(e_layers, 2, -1) if stack else [e_layers]):
        L = min(inp, cur)
        for layer in range(el):
            f += 8 * L * D * HE + 4 * L * L * HE + 4 * L * D * F
            if distil and layer < el - 1:
                f += 2 * (L + 2) * 3 * D * D
                L = (L + 1) // 2 + 1
        cur, mem, inp = L, mem + L, inp // 2
    f += d_layers * (8 * Ld * D * HE + 4 * Ld * Ld * HE + 4 * Ld * D * HE + 4 * mem * D * HE + 4 * Ld * mem * HE
                     + 4 * Ld * D * F)
    return f + 2 * Ld * D * cout
RNN_FLOP_PER_STEP = {"lstm": 1.614e6, "gru": 1.213e6, "rnn": 0.410e6}
LMMSE_BYTES = {(16, 4): 688.0, (8, 4): 368.0, (4, 4): 208.0, (4, 2): 120.0}
PEAK_FP32_MFMA_TFLOPS = 157.3    # dense fp32 matrix peak of MI355X
PEAK_HBM_GBS = 8000.0
CLOCK_SETTLE_MS = 80.0  # the shader clock reaches its steady value some tens of ms into a load


def measured_traffic(workload, units, per_unit_key=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json:
    FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes), scaled to this launch's unit count.
    None when no profile of this workload is on disk."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_hbm_traffic.json"):
            with open(os.path.join(pdir, name)) as f:
                d = json.load(f)
            if workload in d and d[workload].get("units"):
                best = d[workload]["traffic_bytes_per_launch"] / d[workload]["units"] * units
    return best


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the shader clock under this load settles within some tens of milliseconds (5 warm-up
    # launches of the 0.5 ms Transformer step measured 4 % slower than 50 or 500)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="informer",
                    choices=["informer", "informer12", "lmmse", "lstm", "gru", "rnn", "e2e", "selftest"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (sequences or REs)")
    ap.add_argument("--pred-len", type=int, default=25, help="rollout horizon of the recurrent workloads")
    ap.add_argument("--no-attention-maps", action="store_true",
                    help="informer: do not materialise the attention maps (output_attention=False)")
    ap.add_argument("--h-share", type=int, default=1,
                    help="lmmse: block fading, one channel matrix per this many consecutive REs")
    ap.add_argument("--geometry", default="reference", choices=["reference", "scaled"],
                    help="e2e: 4x2 antennas per subcarrier sample, or 16x4 (eight 4x2 predictor "
                         "sequences per subcarrier)")
    ap.add_argument("--clock-settle-ms", type=float, default=CLOCK_SETTLE_MS,
                    help="duration of the untimed clock-settle phase in front of the warm-up steps")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-secondary", action="store_true")
    return ap.parse_args(argv)


def torch_sd(npz_dict):
    sd = {}
    for k, v in npz_dict.items():
        t = torch.from_numpy(np.array(v))
        if k.endswith("position_embedding.pe"):
            full = torch.zeros(1, 5000, t.shape[-1])
            full[:, :t.shape[1]] = t
            t = full
        sd[k] = t
    return sd


def make_informer(output_attention):
    m = cet.InformerStack(16, 16, 16, 25, 10, 5, 5, 64, 8, 4, 3, 64, 0.05, "full", "fixed", "gelu",
                          output_attention, True)
    ck = dict(np.load(os.path.join(GOLDEN, "informer_ckpt.npz")))
    m.load_state_dict(torch_sd(ck))
    return m.eval(), ck


def make_rnn(kind):
    name = kind.upper()
    rk = np.load(os.path.join(GOLDEN, "rnn_ckpt.npz"))
    sd = {k[len(name) + 1:]: rk[k] for k in rk.files if k.startswith(name + "/")}
    m = getattr(cet, name)(16, 16, 256, 2)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.eval(), sd


def clock_settle(step, dev, min_ms=CLOCK_SETTLE_MS):
    """Runs `step` back to back for at least `min_ms` of wall clock and returns the time spent (ms).
    A GPU that has been idle starts a load at a lower shader clock and needs some tens of
    milliseconds to settle; without this phase a short `--warmup` measures the ramp (5 warm-up
    launches of the 0.5 ms Transformer step: 7 % slower than steady state). Not part of the
    warm-up or step counts of the command line."""
    t0 = time.perf_counter()
    if min_ms <= 0:
        return 0.0
    step()  # calibration: how many launches make up the phase
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    step()
    b.record()
    torch.cuda.synchronize(dev)
    per_step = max(a.elapsed_time(b), 1e-3)
    # launched back to back with no host synchronisation in between: gaps in the load keep the
    # clock from ramping (with a synchronisation every four launches 300 ms were not enough)
    for _ in range(int(min_ms / per_step) + 1):
        step()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3


def timed_loop(step, steps, warmup, dist_on, dev, settle=True):
    """A clock-settle phase of the same step, W untimed steps, then exactly K steps between
    barrier+synchronize pairs. Returns (wall seconds for the K steps, mean per-step device time
    in seconds). The device time is one pair of HIP events around the K launches (on the launch
    stream) divided by K, so it includes the gaps between back-to-back launches; an event pair per
    step would be the kernel alone but adds two marker packets per launch, which cost 4 % of the
    step (0.535 against 0.515 ms)."""
    import torch.distributed as dist
    if settle:
        timed_loop.settle_ms = clock_settle(step, dev, timed_loop.min_settle_ms)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    first, last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    first.record()
    for _ in range(steps):
        step()
    last.record()
    torch.cuda.synchronize(dev)
    if dist_on:
        dist.barrier()
    t1 = time.perf_counter()
    dev_s = first.elapsed_time(last) / steps * 1e-3
    return t1 - t0, dev_s


timed_loop.min_settle_ms = CLOCK_SETTLE_MS


def timed_steady(step, dev, warm_ms=40.0, run_ms=120.0):
    """timed_loop for the secondary lines: step counts from one calibration call, so that every
    line is measured after `warm_ms` of its own load (the shader clock settles within some tens of
    milliseconds) and over `run_ms`. Returns the mean per-step device time in seconds."""
    step()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    step()
    b.record()
    torch.cuda.synchronize(dev)
    t = max(a.elapsed_time(b), 1e-3)
    warm, steps = max(2, int(warm_ms / t + 0.999)), max(5, int(run_ms / t + 0.999))
    return timed_loop(step, steps, warm, False, dev, settle=False)[1]


def cpu_baseline_informer(ck, x, xd, budget_s=15.0):
    from oracle import pyoracle
    model_cfg = blobs.make_informer_cfg(16, 16, 16, 25, 10, 5, 64, 8, 4, 3, 64, True, True, "gelu")
    blob = blobs.pack_informer(ck, model_cfg)
    threads = pyoracle.num_threads()
    n = 64
    t0 = time.perf_counter()
    pyoracle.informer_forward(model_cfg, blob, x[:n], xd[:n])
    dt = time.perf_counter() - t0
    n2 = int(min(x.shape[0], max(n, n * budget_s / max(dt, 1e-3))))
    reps = int(min(50, max(1, round(budget_s * n / max(dt, 1e-3) / n2))))  # ~budget_s of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        pyoracle.informer_forward(model_cfg, blob, x[:n2], xd[:n2])
    dt = time.perf_counter() - t0
    return {"value": n2 * reps / dt, "unit": "predictions/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x {n2} of the benchmark's sequences ({dt:.1f} s) through oracle/cet_oracle.c "
                      f"(C restatement, OpenMP over sequences, {threads} threads)"}


def cpu_baseline_torch(run, n_total, what, budget_s=10.0):
    """Stock PyTorch on the host cores (the reference's CPU path runs the same ATen kernels):
    `run(n)` processes the first n sequences with oracle/torch_ref.py. Timed with all the
    threads the box grants (at most 16) for about `budget_s` seconds, and briefly with one.
    Returns (cpu_baseline object, one-thread object)."""
    res = []
    for threads, budget in ((min(16, os.cpu_count() or 1), budget_s), (1, budget_s / 3)):
        torch.set_num_threads(threads)
        n = 64
        run(n)  # warm-up
        t0 = time.perf_counter()
        run(n)
        dt = time.perf_counter() - t0
        n2 = int(min(n_total, max(n, n * budget / max(dt, 1e-4))))
        t0 = time.perf_counter()
        run(n2)  # large batches run faster per sequence than the 64 used for sizing
        dt = time.perf_counter() - t0
        reps = int(min(500, max(1, round(budget / max(dt, 1e-4)))))
        t0 = time.perf_counter()
        for _ in range(reps):
            run(n2)
        dt = time.perf_counter() - t0
        res.append({"value": n2 * reps / dt, "unit": "predictions/s", "cores": threads, "kind": "port",
                    "sample": f"{reps} x {n2} of the benchmark's sequences ({dt:.1f} s): {what}"})
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # also the OpenMP setting of the C checker
    return res[0], res[1]


def cpu_baseline_rnn(kind, sd, x, pred_len, budget_s=15.0):
    from oracle import pyoracle
    kid = blobs.RNN_KINDS[kind]
    blob = blobs.pack_rnn(sd, kid, 16, 16, 256, 2)
    threads = pyoracle.num_threads()
    n = 32
    t0 = time.perf_counter()
    pyoracle.rnn_rollout(kid, 16, 16, 256, 2, blob, x[:n], pred_len)
    dt = time.perf_counter() - t0
    n2 = int(min(x.shape[0], max(n, n * budget_s / max(dt, 1e-3))))
    t0 = time.perf_counter()
    pyoracle.rnn_rollout(kid, 16, 16, 256, 2, blob, x[:n2], pred_len)
    dt = time.perf_counter() - t0
    return {"value": n2 / dt, "unit": "predictions/s", "cores": threads, "kind": "port",
            "sample": f"{n2} sequences, pred_len {pred_len}, oracle/cet_oracle.c with {threads} threads"}


def cpu_baseline_lmmse(Nr, Ns, no, budget_s=12.0):
    from oracle import pyoracle
    threads = pyoracle.num_threads()
    n = 200_000
    y, H, _ = synth.lmmse_problem(n, Nr, Ns, no, seed=0)
    t0 = time.perf_counter()
    pyoracle.lmmse(y, H, no)
    dt = time.perf_counter() - t0
    reps = max(1, int(budget_s / max(dt, 1e-3)))
    t0 = time.perf_counter()
    for _ in range(min(reps, 20)):
        pyoracle.lmmse(y, H, no)
    dt = (time.perf_counter() - t0) / min(reps, 20)
    return {"value": n / dt, "unit": "REs/s", "cores": threads, "kind": "port",
            "sample": f"{n} REs {Nr}x{Ns}, float64 C restatement (oracle/cet_oracle.c), {threads} threads"}


def lmmse_inputs(N, Nr, Ns, no, dev, seed, h_share=1):
    """y [N,Nr], H [ceil(N/h_share),Nr,Ns] (one matrix per h_share consecutive REs), x [N,Ns]."""
    g = torch.Generator(device=dev).manual_seed(seed)
    nb = -(-N // h_share)
    H = torch.randn(nb, Nr, Ns, 2, device=dev, generator=g) * 0.5 ** 0.5
    bits = torch.randint(0, 2, (N, Ns, 2), device=dev, generator=g).float() * 2 - 1
    xs = torch.view_as_complex((bits * 0.5 ** 0.5).contiguous())
    Hc = torch.view_as_complex(H)
    if h_share == 1:
        y = torch.einsum("nrs,ns->nr", Hc, xs)
    else:
        pad = nb * h_share - N
        xp = torch.cat([xs, xs.new_zeros(pad, Ns)]) if pad else xs
        y = torch.einsum("brs,bks->bkr", Hc, xp.reshape(nb, h_share, Ns)).reshape(nb * h_share, Nr)[:N]
    y = y + torch.view_as_complex(torch.randn(N, Nr, 2, device=dev, generator=g) * (no / 2) ** 0.5)
    return y.contiguous(), Hc, xs


class Ctx:
    """Where this rank runs: device, rank numbering, whether a process group exists."""

    def __init__(self, dev, rank, n_gpus, dist_on):
        self.dev, self.rank, self.n_gpus, self.dist_on = dev, rank, n_gpus, dist_on

    def max_over_ranks(self, seconds):
        if not self.dist_on:
            return seconds
        import torch.distributed as dist
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)


def free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def visible_gpus():
    """GPUs this process's children will see, counted without any HIP or torch.cuda call (the
    launcher parent must not open the device): the KFD topology nodes that have SIMDs, cut down
    by the *_VISIBLE_DEVICES lists. 0 when the kernel driver is not loaded (no /sys/class/kfd),
    None when its topology cannot be read."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir("/sys/class/kfd"):
        return 0
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip()]))
    return n


def launch_ranks(n, argv):
    """Starts `n` ranks of this script on this node through torch.distributed.run, as a child
    process (this process has not touched a GPU: no HIP call and no torch.cuda call precedes
    this), waits for them and returns their exit status. Rank 0 of the children prints the JSON
    line on the shared stdout."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC, which RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def rnn_nmse_check(kind, dev):
    """NMSE of this path's rollout and of the reference's own outputs (captured from the imported
    reference, tests/golden/rnn_cr.npz) on the reference's stored CDL-B sequences, against the same
    truth (slots 25..29 of CR_data_*), with the reference's checkpoint that was trained on that
    data set (prediciton_code/checkpoints/checkpoints_30-60_L5): the accuracy contract of the
    recurrent predictors."""
    g = np.load(os.path.join(GOLDEN, "rnn_cr.npz"))
    name = f"{kind.upper()}_L5"
    sd = {k[len(name) + 4:]: g[k] for k in g.files if k.startswith(name + "/sd/")}
    model = getattr(cet, kind.upper())(16, 16, 256, 2)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model.eval()
    x = torch.from_numpy(g["x"]).to(dev)
    truth = torch.from_numpy(g["truth"]).to(dev)
    ref = torch.from_numpy(g[f"{name}/pred5"]).to(dev)
    ours = model.test_data(x, 5, dev)
    nm_ours = metrics.nmse_db(metrics.nmse_partial_sums(ours, truth)).cpu()
    nm_ref = metrics.nmse_db(metrics.nmse_partial_sums(ref, truth)).cpu()
    return {"data": f"CR_data_30/36/60 ({x.shape[0]} CDL-B sequences shipped with the reference), "
                    f"checkpoint checkpoints_30-60_L5/{kind.upper()}_data_0_sl25_pl5_hs256_hl2",
            "nmse_db_per_slot": [round(float(v), 4) for v in nm_ours],
            "reference_nmse_db_per_slot": [round(float(v), 4) for v in nm_ref],
            "max_abs_delta_db": float((nm_ours - nm_ref).abs().max()),
            "max_abs_err_vs_reference_output": float((ours - ref).abs().max())}


def run_rnn(ctx, kind, B, P, steps, warmup, cpu_baseline=False):
    """Recurrent rollout (`test_data`, seq_len 25, pred_len P) on B sequences per GPU."""
    dev = ctx.dev
    model, sd = make_rnn(kind)
    noisy, _ = synth.channel_histories(B, T=25 + 5, seed=ctx.rank)
    x = cet.LoadBatch(noisy[:, :25]).to(dev)
    holder = {}

    def step():
        holder["pred"] = model.test_data(x, P, dev)

    wall, dev_s = timed_loop(step, steps, warmup, ctx.dist_on, dev)
    wall = ctx.max_over_ranks(wall)
    flop = RNN_FLOP_PER_STEP[kind] * (24 + P) * B
    achieved = flop / dev_s / 1e12
    out = {
        "metric": f"channel predictions/s ({kind.upper()} rollout)",
        "value": B * ctx.n_gpus * steps / wall, "unit": "predictions/s",
        "ms_per_step": wall / steps * 1e3, "dtype": "f32",
        "config": {"workload": f"{kind.upper()}(16,16,256,2).test_data, seq_len 25, pred_len {P}, shipped checkpoint",
                   "batch_per_gpu": B, "global_batch": B * ctx.n_gpus, "parallelism": f"dp{ctx.n_gpus}"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                     "traffic": measured_traffic(kind, B) if P == 5 else None,  # profiled at pred_len 5
                     # the library's choice for the batch at 256 hidden units on 256 CUs
                     "kernel": ("rnn_rollout_kernel" if B > 4096 else "rnn_mid_kernel" if B > 2048 else
                                "rnn_mid_shared_kernel" if B > 512 else "rnn_split_kernel"),
                     "kernel_ms": dev_s * 1e3, "flop_per_launch": flop},
    }
    out["nmse_check"] = rnn_nmse_check(kind, dev)
    if cpu_baseline:
        from oracle import torch_ref
        xc = x.cpu()
        out["cpu_baseline"], one = cpu_baseline_torch(
            lambda n: torch_ref.rnn_rollout(sd, kind, xc[:n], P), B,
            f"stock torch.nn.{kind.upper()} rollout (oracle/torch_ref.py)")
        out["cpu_baseline_other"] = {
            "torch_one_thread": one,
            "c_checker": cpu_baseline_rnn(kind, sd, x.cpu().numpy(), P, budget_s=5.0)}
    return out


def run_lmmse(ctx, N, hs, steps, warmup, cpu_baseline=False):
    """LMMSE 16x4 on N resource elements per GPU (hs > 1: one channel matrix per hs elements)."""
    dev = ctx.dev
    Nr, Ns, no = 16, 4, 0.1
    y, H, xs = lmmse_inputs(N, Nr, Ns, no, dev, ctx.rank, hs)
    holder = {}

    def step():
        holder["x"], holder["ne"] = cet.lmmse(y, H, no, h_share=hs)

    wall, dev_s = timed_loop(step, steps, warmup, ctx.dist_on, dev)
    bits = metrics.qpsk_bit_errors(holder["x"], xs)
    metrics.all_reduce_sums(bits)
    wall = ctx.max_over_ranks(wall)
    bytes_re = LMMSE_BYTES[(Nr, Ns)] - Nr * Ns * 8 * (1.0 - 1.0 / hs)  # H once per block
    gbs = bytes_re * N / dev_s / 1e9
    out = {
        "metric": "LMMSE-equalized REs/s", "value": N * ctx.n_gpus * steps / wall, "unit": "REs/s",
        "ms_per_step": wall / steps * 1e3, "dtype": "complex64 (f32 arithmetic)",
        "config": {"workload": f"LMMSE {Nr}x{Ns}, {N} REs per GPU, no=0.1"
                               + (" (3276 subcarriers x 14 symbols x 1024 slots)" if N == 3276 * 14 * 1024 else "")
                               + (f", block fading: one H per {hs} REs, {bytes_re:.1f} B/RE" if hs > 1 else ""),
                   "parallelism": f"dp{ctx.n_gpus}"},
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": gbs / PEAK_HBM_GBS,
                     "traffic": measured_traffic("lmmse", N) if hs == 1 and (Nr, Ns) == (16, 4)
                     else (measured_traffic("lmmse_block", N) if hs == 14 and (Nr, Ns) == (16, 4) else None),
                     "kernel": "lmmse_tiled<16,4,2>" if hs == 1 else "lmmse_block<16,4,2>",
                     "kernel_ms": dev_s * 1e3, "bytes_per_launch": bytes_re * N},
        "ber": float(bits[0] / bits[1]),
    }
    if cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_lmmse(Nr, Ns, no)
    return out


def run_e2e(ctx, scaled, S, steps, warmup):
    """Transformer prediction of 5 future slots per subcarrier sample -> LMMSE over the 14 OFDM
    symbols of each predicted slot -> hard-decision QPSK bit errors and per-slot NMSE; ranks hold
    disjoint subcarrier ranges. S = predictor sequences (4x2 antenna groups) per GPU."""
    dev, rank = ctx.dev, ctx.rank
    n_sym, no = 14, 0.05
    Nr, Nt = (16, 4) if scaled else (4, 2)
    model, ck = make_informer(False)
    # the checkpoint predicts CDL-B channels: the reference's 192 stored sequences, tiled
    cr = np.load(os.path.join(GOLDEN, "cr_pairs.npz"))
    real = np.concatenate([cr["data_30"], cr["data_36"], cr["data_60"]])
    real = np.roll(real, 7 * rank, axis=0)[np.arange(S) % real.shape[0]]
    real = torch.from_numpy(real).to(dev).to(torch.complex64)
    if scaled:  # eight stored 4x2 sequences make one 16x4 subcarrier sample
        real = e2e.ungroup_antennas(real[:S - S % 8], Nr, Nt)
    h_hist, h_future = real[:, :25].contiguous(), real[:, 25:30].contiguous()
    n_sc = h_hist.shape[0]
    g = torch.Generator(device=dev).manual_seed(rank)
    y, xs = e2e.transmit(h_future, n_sym, no, g)
    holder = {}
    predict = e2e.predict_slots_grouped if scaled else e2e.predict_slots

    def step():
        h_hat = predict(model, h_hist, 10, 5)
        x_hat, _ = e2e.equalize_slots(h_hat, y, no)
        bits = metrics.qpsk_bit_errors(x_hat, xs)
        holder["sums"] = {"nmse": metrics.nmse_partial_sums(h_hat, h_future), "bits": bits,
                          "bits_perfect": bits}  # replaced after the timed region

    wall, dev_s = timed_loop(step, steps, warmup, ctx.dist_on, dev)
    # outside the timed region: the same link equalized with the true channel, and with the
    # prediction error's variance passed to the equalizer
    x_true, _ = e2e.equalize_slots(h_future.to(torch.complex64), y, no)
    holder["sums"]["bits_perfect"] = metrics.qpsk_bit_errors(x_true, xs)
    h_hat = predict(model, h_hist, 10, 5)
    err_var = float((h_hat - h_future).abs().pow(2).mean())
    x_ev, _ = e2e.equalize_slots(h_hat, y, no, err_var)
    bits_ev = metrics.qpsk_bit_errors(x_ev, xs)
    metrics.all_reduce_sums(bits_ev)
    res = e2e.reduce(holder["sums"])  # the path's only collective (RCCL)
    wall = ctx.max_over_ranks(wall)
    n_re = n_sc * 5 * n_sym
    frac = INFORMER_FLOP * S / dev_s / 1e12 / PEAK_FP32_MFMA_TFLOPS
    return {
        "metric": "channel predictions/s (Transformer prediction + LMMSE equalization + BER/NMSE)",
        "value": S * ctx.n_gpus * steps / wall, "unit": "predictions/s",
        "ms_per_step": wall / steps * 1e3, "dtype": "f32",
        "data": "the reference's stored CDL-B sequences (CR_data_30/36/60), tiled; synthetic QPSK + AWGN",
        "config": {"workload": f"end to end: InformerStack default configuration on {S} 4x2 antenna "
                               f"groups = {n_sc} subcarrier samples per GPU ({Nr}x{Nt} channel), "
                               f"block-fading LMMSE {Nr}x{Nt} on {n_re} REs "
                               f"(5 slots x {n_sym} symbols), QPSK, no={no}",
                   "parallelism": f"dp{ctx.n_gpus} (subcarrier ranges)"},
        "roofline": {"bound": "mfma", "achieved": frac * PEAK_FP32_MFMA_TFLOPS,
                     "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": frac, "traffic": None,
                     "kernel": "whole step (informer_kernel + lmmse + reductions + torch glue)",
                     "kernel_ms": dev_s * 1e3},
        "equalized_REs_per_s": n_re * ctx.n_gpus * steps / wall,
        "ber": res["ber"], "ber_perfect_csi": res["ber_perfect_csi"],
        "ber_with_err_var": float(bits_ev[0] / bits_ev[1]), "err_var": err_var,
        "nmse_db_per_slot": [round(v, 3) for v in res["nmse_db"]],
    }


def run_informer12(ctx, B, steps, warmup, steady=False):
    """The shipped 12-head / 35-slot / 25-decoder-row checkpoint (large launch shape of the
    Transformer kernel, one wave per sequence), output_attention=False, B sequences per GPU."""
    dev = ctx.dev
    h12 = np.load(os.path.join(GOLDEN, "informer_h12.npz"))
    m12 = cet.InformerStack(16, 16, 16, 35, 20, 5, 5, 64, 12, 4, 3, 64, 0.05, "full", "fixed", "gelu",
                            False, True)
    m12.load_state_dict(torch_sd({k[len("h12/sd/"):].replace("module.", "", 1): h12[k]
                                  for k in h12.files if k.startswith("h12/sd/")}))
    m12 = m12.eval().to(dev)
    noisy12, _ = synth.channel_histories_device(B, dev, T=40, seed=12 + ctx.rank)
    x12 = cet.load_batch_device(noisy12[:, :35]).contiguous()
    xd12 = cet.decoder_input(x12, 20, 5)
    if steady:
        wall = dev_s = timed_steady(lambda: m12(x12, xd12), dev)
        steps = 1
    else:
        wall, dev_s = timed_loop(lambda: m12(x12, xd12), steps, warmup, ctx.dist_on, dev)
        wall = ctx.max_over_ranks(wall)
    f12 = informer_flop(35, 20, 5, H=12)
    achieved = f12 * B / dev_s / 1e12
    return {
        "metric": "channel predictions/s (Transformer predictor forward, 12-head checkpoint)",
        "value": B * ctx.n_gpus * steps / wall, "unit": "predictions/s", "ms_per_step": wall / steps * 1e3,
        "dtype": "f32",
        "config": {"workload": "InformerStack forward, shipped InformerStack_5x_d_model64_n_heads12_..._seq_len35_"
                               "label_len20_pred_len5 checkpoint, output_attention=False",
                   "batch_per_gpu": B, "global_batch": B * ctx.n_gpus, "parallelism": f"dp{ctx.n_gpus}"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": measured_traffic("informer12", B),
                     "kernel": "informer_kernel<3,2,1,false,4,3>", "kernel_ms": dev_s * 1e3,
                     "flop_per_launch": f12 * B, "flop_per_prediction": f12},
    }


def run_selftest(ctx):
    """No GPU work: every rank forms the NMSE partial sums of its contiguous shard of a seeded
    batch on the CPU, one all-reduce sums them, rank 0 checks the total against the whole batch.
    Exercises the launcher, the rendezvous and the collective exactly as the GPU workloads use
    them (backend gloo instead of RCCL)."""
    per_rank = 8
    B = per_rank * ctx.n_gpus
    g = torch.Generator().manual_seed(0)
    truth = torch.randn(B, 5, 16, generator=g)
    pred = truth + 0.1 * torch.randn(B, 5, 16, generator=g)
    lo = ctx.rank * per_rank
    t0 = time.perf_counter()
    sums = metrics.nmse_partial_sums(pred[lo:lo + per_rank], truth[lo:lo + per_rank])
    metrics.all_reduce_sums(sums)
    wall = ctx.max_over_ranks(time.perf_counter() - t0)
    want = metrics.nmse_partial_sums(pred, truth)
    ok = bool(torch.allclose(sums, want, rtol=1e-12))
    return {"metric": "launcher self-test (CPU, gloo)", "value": B / max(wall, 1e-9), "unit": "sequences/s",
            "ms_per_step": wall * 1e3, "dtype": "f64", "config": {"workload": "selftest", "global_batch": B},
            "all_reduce_matches_single_process": ok,
            "nmse_db_per_slot": [round(float(v), 6) for v in metrics.nmse_db(sums)]}


def secondary_lines(ctx, args, model, ck, x, xd, want_maps, dev_s):
    """The other kernels and configurations of the path, shorter runs, each with its own roofline
    object; all inside the default run so that one driver record covers the whole metric."""
    dev = ctx.dev
    B = x.shape[0]
    extra = {}

    def roof(bound, achieved, kernel_ms, **more):
        peak = PEAK_FP32_MFMA_TFLOPS if bound == "mfma" else PEAK_HBM_GBS
        r = {"bound": bound, "achieved": achieved, "peak": peak,
             "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": achieved / peak, "kernel_ms": kernel_ms}
        r.update(more)
        return r

    model2, _ = make_informer(not want_maps)
    def step2():
        model2(x, xd)
    _w, d2 = timed_loop(step2, args.steps, args.warmup, False, dev)
    extra[f"informer_output_attention_{not want_maps}"] = {
        "predictions_per_s": B / d2, "kernel_ms": d2 * 1e3,
        "mfma_frac": INFORMER_FLOP * B / d2 / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "roofline": roof("mfma", INFORMER_FLOP * B / d2 / 1e12, d2 * 1e3)}
    # the shipped 12-head / 35-slot / 25-decoder-row checkpoint (large one-wave launch shape)
    r12 = run_informer12(Ctx(dev, 0, 1, False), B, 0, 0, steady=True)
    extra["informer_12_heads_35_slots"] = {
        "predictions_per_s": r12["value"], "kernel_ms": r12["roofline"]["kernel_ms"],
        "flop_per_prediction": r12["roofline"]["flop_per_prediction"], "roofline": r12["roofline"],
        "note": "shipped InformerStack_5x_d_model64_n_heads12_..._seq_len35_label_len20 checkpoint, "
                f"B={B}, output_attention=False"}
    # the reference's own GPU path on this card: the model in stock PyTorch-ROCm eager
    # calls (oracle/torch_ref.py on device tensors), same weights, same batch
    from oracle import torch_ref as _tr
    sd_dev = {k: torch.from_numpy(np.array(v)).to(dev) for k, v in ck.items()}
    def step_eager():
        with torch.no_grad():
            _tr.informer_forward(sd_dev, x, xd, 5)
    de = timed_steady(step_eager, dev)
    extra["stock_pytorch_rocm_eager_same_gpu"] = {
        "predictions_per_s": B / de, "step_ms": de * 1e3,
        "speedup_of_the_hip_kernel": de / dev_s,
        "note": f"InformerStack forward, B={B}, torch.nn.functional calls on the MI355X (about 190 kernel launches)"}
    del sd_dev
    # the same step fed from and returning to pinned host memory (Predict.py:99-117
    # moves every batch numpy -> GPU -> numpy); never the headline value
    xh, xdh = x.cpu().pin_memory(), xd.cpu().pin_memory()
    oh = torch.empty(B, 5, 16).pin_memory()
    def step_host():
        r = model(xh.to(dev, non_blocking=True), xdh.to(dev, non_blocking=True))
        oh.copy_(r[0] if want_maps else r, non_blocking=True)
    dh = timed_steady(step_host, dev)
    extra["informer_with_host_transfers"] = {
        "predictions_per_s": B / dh, "step_ms": dh * 1e3,
        "note": "pinned host input (x_enc, x_dec) -> device -> forward -> prediction to pinned host"}
    # the same with three batches in flight on three streams (HostPipeline): upload of batch i + 1
    # and download of batch i - 1 behind the forward of batch i
    from channelestimationtransformer_amd import HostPipeline
    pipe = HostPipeline(lambda a, b: model(a, b), [tuple(x.shape), tuple(xd.shape)], (B, 5, 16), device=dev)
    for sl in pipe.slots:
        sl.host_in[0].copy_(xh)
        sl.host_in[1].copy_(xdh)
    n_pipe = max(20, 2 * args.steps)
    pipe.run(10, lambda i, h: None, lambda i, o: None)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    pipe.run(n_pipe, lambda i, h: None, lambda i, o: None)
    torch.cuda.synchronize(dev)
    dp = (time.perf_counter() - t0) / n_pipe
    extra["informer_with_host_transfers_pipelined"] = {
        "predictions_per_s": B / dp, "step_ms": dp * 1e3, "batches": n_pipe,
        "note": "the same from / to pinned host buffers through HostPipeline (3 batches in flight, wall clock per batch)"}

    # ---- LMMSE: the stated size of the metric's second half, then the other shapes -------------
    one = Ctx(dev, 0, 1, False)
    r = run_lmmse(one, 3276 * 14 * 1024, 1, 20, 7)  # about 45 ms of warm-up, 125 ms timed
    extra["lmmse_16x4_46964736_REs"] = {"REs_per_s": r["value"], "kernel_ms": r["roofline"]["kernel_ms"],
                                        "N": 3276 * 14 * 1024, "ber": r["ber"], "roofline": r["roofline"]}
    torch.cuda.empty_cache()
    N = 1 << 23
    r = run_lmmse(one, N, 1, 110, 40)
    extra["lmmse_16x4"] = {"REs_per_s": r["value"], "kernel_ms": r["roofline"]["kernel_ms"], "N": N,
                           "hbm_GBps": r["roofline"]["achieved"], "hbm_frac": r["roofline"]["frac"],
                           "roofline": r["roofline"]}
    r = run_lmmse(one, N, 14, 280, 100)
    extra["lmmse_16x4_block_fading_14"] = {
        "REs_per_s": r["value"], "kernel_ms": r["roofline"]["kernel_ms"], "N": N,
        "bytes_per_RE": r["roofline"]["bytes_per_launch"] / N, "hbm_GBps": r["roofline"]["achieved"],
        "hbm_frac": r["roofline"]["frac"], "roofline": r["roofline"]}
    torch.cuda.empty_cache()
    # the reference-sized equalizer shapes (4 x 2 of EndtoEndtest.py, 8 x 4)
    for (nr, ns) in ((8, 4), (4, 2)):
        Nv = 1 << 24
        yv, Hv, _xs = lmmse_inputs(Nv, nr, ns, 0.1, dev, 0)
        def step3b():
            cet.lmmse(yv, Hv, 0.1)
        dv = timed_steady(step3b, dev)
        bpr = LMMSE_BYTES[(nr, ns)]
        extra[f"lmmse_{nr}x{ns}"] = {"REs_per_s": Nv / dv, "kernel_ms": dv * 1e3, "N": Nv,
                                     "hbm_GBps": bpr * Nv / dv / 1e9,
                                     "hbm_frac": bpr * Nv / dv / 1e9 / PEAK_HBM_GBS,
                                     "roofline": roof("hbm", bpr * Nv / dv / 1e9, dv * 1e3)}
        del yv, Hv
    # block fading (one matrix per 14 REs) on the other four-stream shapes
    for (nr, ns) in ((8, 4), (4, 4)):
        Nv = 1 << 24
        yv, Hv, _xs = lmmse_inputs(Nv, nr, ns, 0.1, dev, 0, 14)
        dv = timed_steady(lambda: cet.lmmse(yv, Hv, 0.1, h_share=14), dev)
        bpr = LMMSE_BYTES[(nr, ns)] - nr * ns * 8 * (1.0 - 1.0 / 14)
        extra[f"lmmse_{nr}x{ns}_block_fading_14"] = {
            "REs_per_s": Nv / dv, "kernel_ms": dv * 1e3, "N": Nv, "bytes_per_RE": bpr,
            "hbm_GBps": bpr * Nv / dv / 1e9, "hbm_frac": bpr * Nv / dv / 1e9 / PEAK_HBM_GBS,
            "roofline": roof("hbm", bpr * Nv / dv / 1e9, dv * 1e3, kernel=f"lmmse_block<{nr},{ns},2>")}
        del yv, Hv
    # the reference's receive chain on its own grid (EndtoEndtest.py:50-59, 130-133, 184-191):
    # LS estimate with nearest-neighbour interpolation -> LMMSEEqualizer object, 4 rx x 2 streams
    rg = cet.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=15e3, num_tx=1,
                          num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                          dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    est = cet.LSChannelEstimator(rg, interpolation_type="nn")
    equ = cet.LMMSEEqualizer(rg, cet.StreamManagement(np.array([[1]]), 2))
    nb = 8192
    # four input sets used in turn (4 GB in all), so that nothing a call reads was left in the
    # 256 MB memory-side cache by the call before it
    ygs = [torch.view_as_complex(torch.randn(nb, 1, 4, 14, 76, 2, device=dev)) for _ in range(4)]
    ests = [est([yg, 0.1]) for yg in ygs]
    turn = {"ls": 0, "eq": 0}
    def step_ls():
        turn["ls"] = (turn["ls"] + 1) % 4
        est([ygs[turn["ls"]], 0.1])
    dls = timed_steady(step_ls, dev)
    def step_eq():
        k = turn["eq"] = (turn["eq"] + 1) % 4
        equ([ygs[k], ests[k][0], ests[k][1], 0.1])
    deq = timed_steady(step_eq, dev)
    n_out = nb * 4 * 2 * 14 * 64
    extra["receive_chain_reference_grid"] = {
        "batch": nb, "ls_nn_ms": dls * 1e3,
        # 12 bytes per output element + the two pilot symbols of every antenna plane
        "ls_nn_GBps": (n_out * 12 + nb * 4 * 2 * 76 * 8) / dls / 1e9,
        "equalizer_object_ms": deq * 1e3, "equalizer_object_REs_per_s": nb * 12 * 64 / deq,
        # per data RE: y 4x8 + h_hat 8x8 + err_var 8x4 + x_hat 2x8 + no_eff 2x4 bytes
        "equalizer_object_GBps": nb * 12 * 64 * 152.0 / deq / 1e9,
        "roofline": roof("hbm", nb * 12 * 64 * 152.0 / deq / 1e9, deq * 1e3, kernel="lmmse_grid<4,2>")}
    del ygs, ests
    torch.cuda.empty_cache()

    # ---- the reference's own CPU-runnable case: 64 sequences (latency, not throughput) ---------
    x64, xd64 = x[:64].contiguous(), xd[:64].contiguous()
    def step64():
        model(x64, xd64)
    d64 = timed_steady(step64, dev)
    lat = {"informer_ms": d64 * 1e3, "informer_predictions_per_s": 64 / d64}
    for kind in ("gru", "lstm", "rnn"):
        gm, _sd = make_rnn(kind)
        def step64g():
            gm.test_data(x64, 5, dev)
        d64g = timed_steady(step64g, dev)
        lat[f"{kind}_pred5_ms"] = d64g * 1e3
        lat[f"{kind}_predictions_per_s"] = 64 / d64g
    extra["batch_64_latency"] = lat
    # medium batches of the recurrent predictors (the kernel the library picks for the batch: half-tile
    # workgroups of 16 sequences, a tile's hidden units on 4 / 2 / 1 workgroups at 1024 / 2048 / 4096)
    for nb in (1024, 2048, 4096):
        xm = x[:nb].contiguous()
        mid = {}
        for kind in ("gru", "lstm", "rnn"):
            gm, _sd = make_rnn(kind)
            dm = timed_steady(lambda: gm.test_data(xm, 5, dev), dev)
            mid[f"{kind}_pred5_ms"] = dm * 1e3
            mid[f"{kind}_predictions_per_s"] = nb / dm
        extra[f"batch_{nb}_rollout"] = mid
    # shapes beyond the shipped checkpoints' (hidden 512; random weights): the half-tile kernels alone,
    # a tile's hidden units on 8 workgroups at 512 sequences, one workgroup per tile at 4096
    wide = {}
    for cls, G, H, NL in (("GRU", 3, 512, 2), ("LSTM", 4, 512, 3)):
        torch.manual_seed(0)
        wm = getattr(cet, cls)(16, 16, H, NL).eval()
        for nb in (512, 4096):
            xm = x[:nb].contiguous()
            dm = timed_steady(lambda: wm.test_data(xm, 5, dev), dev)
            fl = 29 * 2 * G * (H * (16 + H) + (NL - 1) * 2 * H * H) * nb
            wide[f"{cls.lower()}_{H}x{NL}_B{nb}_pred5"] = {
                "ms": dm * 1e3, "predictions_per_s": nb / dm, "mfma_frac": fl / dm / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        del wm
    extra["recurrent_hidden_512"] = wide

    # ---- recurrent rollouts, 8192 sequences, pred_len 5 / 25 / 100 ------------------------------
    for kind in ("lstm", "gru", "rnn"):
        rm, _sd = make_rnn(kind)
        nbh, _ = synth.channel_histories(8192, seed=1)
        xr = cet.LoadBatch(nbh[:, :25]).to(dev)
        sd_dev = {k: torch.from_numpy(np.array(v)).to(dev) for k, v in _sd.items()}
        for P in (5, 25, 100):
            def step4():
                rm.test_data(xr, P, dev)
            d4 = timed_steady(step4, dev)
            fl = RNN_FLOP_PER_STEP[kind] * (24 + P) * 8192
            line = {"predictions_per_s": 8192 / d4, "kernel_ms": d4 * 1e3,
                    "mfma_frac": fl / d4 / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    "roofline": roof("mfma", fl / d4 / 1e12, d4 * 1e3, flop_per_launch=fl)}
            if P == 5:
                def step4e():
                    _tr.rnn_rollout(sd_dev, kind, xr, 5)
                d4e = timed_steady(step4e, dev)
                line["stock_pytorch_rocm_eager_ms"] = d4e * 1e3
                line["speedup_of_the_hip_kernel"] = d4e / d4
                line["nmse_check"] = rnn_nmse_check(kind, dev)
            extra[f"{kind}_rollout_B8192_pred{P}"] = line
        del sd_dev

    # ---- end to end, both geometries --------------------------------------------------------------
    for geometry in ("reference", "scaled"):
        r = run_e2e(one, geometry == "scaled", 3276 * 8, 36, 12)
        extra[f"e2e_{geometry}_geometry"] = {
            k: r[k] for k in ("value", "unit", "ms_per_step", "equalized_REs_per_s", "ber", "ber_perfect_csi",
                              "ber_with_err_var", "nmse_db_per_slot", "roofline")}
        extra[f"e2e_{geometry}_geometry"]["workload"] = r["config"]["workload"]
    return extra


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    timed_loop.min_settle_ms = args.clock_settle_ms
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: become one (nothing here has touched a GPU yet)
        seen = None if args.workload == "selftest" else visible_gpus()
        if seen is not None and seen < args.gpus:  # unknown: the ranks fail by themselves
            raise SystemExit(f"--gpus {args.gpus}: only {seen} GPU(s) visible")
        raise SystemExit(launch_ranks(args.gpus, argv))

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun, also at N = 1
    cpu_only = args.workload == "selftest"
    if cpu_only:
        dev = torch.device("cpu")
        if dist_on:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if dist_on:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctx = Ctx(dev, rank, world, dist_on)
    n_gpus = world

    out = {"n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
    extra = {}
    with_cpu = rank == 0 and n_gpus == 1 and not args.skip_cpu_baseline

    if args.workload == "informer":
        B = args.batch or 4096
        want_maps = not args.no_attention_maps
        model, ck = make_informer(want_maps)
        noisy, _ = synth.channel_histories(B, seed=rank)
        x = cet.LoadBatch(noisy[:, :25]).to(dev)
        xd = cet.decoder_input(x, 10, 5)
        truth = cet.LoadBatch(noisy[:, 25:30]).to(dev)
        holder = {}

        def step():
            r = model(x, xd)
            holder["pred"] = r[0] if want_maps else r

        wall, dev_s = timed_loop(step, args.steps, args.warmup, dist_on, dev)
        sums = metrics.nmse_partial_sums(holder["pred"], truth)
        metrics.all_reduce_sums(sums)  # the path's only collective (RCCL)
        wall = ctx.max_over_ranks(wall)
        achieved = INFORMER_FLOP * B / dev_s / 1e12
        out.update({
            "metric": "channel predictions/s (Transformer predictor forward)",
            "value": B * n_gpus * args.steps / wall, "unit": "predictions/s",
            "ms_per_step": wall / args.steps * 1e3, "dtype": "f32",
            "config": {"workload": "InformerStack forward, enc_in=16 seq_len=25 label_len=10 pred_len=5 "
                                   "d_model=64 n_heads=8 e_layers=4 d_layers=3 d_ff=64, "
                                   f"output_attention={want_maps}, shipped checkpoint",
                       "batch_per_gpu": B, "global_batch": B * n_gpus, "parallelism": f"dp{n_gpus}"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": measured_traffic("informer" if want_maps else "informer_nomaps", B),
                         "kernel": "informer_kernel", "kernel_ms": dev_s * 1e3,
                         "flop_per_launch": INFORMER_FLOP * B},
            "nmse_db_per_slot_synthetic": [round(float(v), 3) for v in metrics.nmse_db(sums)],
        })
        # NMSE on the reference's stored CDL-B sequences next to the NMSE of the reference's own
        # stored outputs for them (CR_data_* -> CR_transformer_*): the accuracy contract.
        cr = np.load(os.path.join(GOLDEN, "cr_pairs.npz"))
        real = np.concatenate([cr[f"data_{v}"] for v in ("30", "36", "60")])
        ref_out = np.concatenate([cr[f"transformer_{v}"] for v in ("30", "36", "60")])
        xr = cet.LoadBatch(real[:, :25]).to(dev)
        r = model(xr, cet.decoder_input(xr, 10, 5))
        ours = cet.real2complex(r[0] if want_maps else r).reshape(-1, 5, 4, 2)
        truth_r = torch.from_numpy(real[:, 25:30]).to(dev)
        nm_ours = metrics.nmse_db(metrics.nmse_partial_sums(ours, truth_r)).cpu()
        nm_ref = metrics.nmse_db(metrics.nmse_partial_sums(torch.from_numpy(ref_out).to(dev), truth_r)).cpu()
        out["nmse_check"] = {
            "data": "CR_data_30/36/60 (192 CDL-B sequences shipped with the reference)",
            "nmse_db_per_slot": [round(float(v), 4) for v in nm_ours],
            "reference_nmse_db_per_slot": [round(float(v), 4) for v in nm_ref],
            "max_abs_delta_db": float((nm_ours - nm_ref).abs().max()),
            "max_abs_err_vs_reference_output": float((ours.cpu() - torch.from_numpy(ref_out)).abs().max())}
        if with_cpu:
            # the baseline: stock PyTorch on the host cores (the ATen CPU kernels the reference's own
            # CPU path runs); beside it the same with one thread and the scalar C checker
            from oracle import torch_ref
            xc, xdc = x.cpu(), xd.cpu()
            with torch.no_grad():
                out["cpu_baseline"], one = cpu_baseline_torch(
                    lambda n: torch_ref.informer_forward(ck, xc[:n], xdc[:n], 5), B,
                    "InformerStack forward in stock torch.nn.functional calls (oracle/torch_ref.py)")
            out["cpu_baseline_other"] = {
                "torch_one_thread": one,
                "c_checker": cpu_baseline_informer(ck, x.cpu().numpy(), xd.cpu().numpy(), budget_s=5.0)}
        if rank == 0 and n_gpus == 1 and not args.skip_secondary:
            extra = secondary_lines(ctx, args, model, ck, x, xd, want_maps, dev_s)
    elif args.workload == "informer12":
        out.update(run_informer12(ctx, args.batch or 4096, args.steps, args.warmup))
    elif args.workload == "e2e":
        out.update(run_e2e(ctx, args.geometry == "scaled", args.batch or 3276 * 8, args.steps, args.warmup))
    elif args.workload == "lmmse":
        out.update(run_lmmse(ctx, args.batch or 3276 * 14 * 1024, max(1, args.h_share), args.steps,
                             args.warmup, with_cpu))
    elif args.workload == "selftest":
        out.update(run_selftest(ctx))
    else:
        out.update(run_rnn(ctx, args.workload, args.batch or 8192, args.pred_len, args.steps, args.warmup,
                           with_cpu))

    if not cpu_only:
        out["clock_settle_ms"] = round(getattr(timed_loop, "settle_ms", 0.0), 1)
    if extra:
        out["secondary"] = extra
    # one compact object as the LAST key: every configuration's [value, unit, roofline fraction]
    summary = {}
    if "roofline" in out:
        summary[out["config"]["workload"].split(",")[0] + f" (headline, B={out['config'].get('batch_per_gpu', '')})"] = [
            round(out["value"]), out["unit"], round(out["roofline"]["frac"], 4)]
    for key, line in extra.items():
        r = line.get("roofline")
        if not isinstance(r, dict):
            continue
        val = line.get("predictions_per_s", line.get("REs_per_s", line.get("equalizer_object_REs_per_s", line.get("value"))))
        unit = "REs/s" if ("REs_per_s" in line or "equalizer_object_REs_per_s" in line) else "predictions/s"
        summary[key] = [round(val) if val else None, unit, round(r["frac"], 4)]
    if summary:
        out["summary"] = summary
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
