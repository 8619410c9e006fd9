"""The multi-rank reduction of the NMSE/BER partial sums, on CPU with gloo (world size 2), and
bench.py's own launcher."""
import json
import os
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from channelestimationtransformer_amd import metrics
    g = torch.Generator().manual_seed(0)
    truth = torch.randn(8, 5, 16, generator=g)
    pred = truth + 0.1 * torch.randn(8, 5, 16, generator=g)
    lo, hi = rank * 4, rank * 4 + 4  # contiguous batch shard per rank
    sums = metrics.nmse_partial_sums(pred[lo:hi], truth[lo:hi])
    bits = metrics.qpsk_bit_errors(torch.view_as_complex(pred[lo:hi].reshape(4, 5, 8, 2).contiguous()),
                                   torch.view_as_complex(truth[lo:hi].reshape(4, 5, 8, 2).contiguous()))
    flat = torch.cat([sums.reshape(-1), bits])
    metrics.all_reduce_sums(flat)
    if rank == 0:
        full = torch.cat([metrics.nmse_partial_sums(pred, truth).reshape(-1),
                          metrics.qpsk_bit_errors(torch.view_as_complex(pred.reshape(8, 5, 8, 2).contiguous()),
                                                  torch.view_as_complex(truth.reshape(8, 5, 8, 2).contiguous()))])
        q.put((flat.numpy(), full.numpy()))
    dist.destroy_process_group()


def test_two_rank_all_reduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, want = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.allclose(got, want, rtol=1e-12)


def _e2e_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from channelestimationtransformer_amd import e2e
    # per-rank partial sums of an end-to-end pass (made-up numbers, the reduction is what is tested)
    sums = {"nmse": torch.tensor([[1.0 + rank, 2.0, 3.0], [10.0, 20.0 + rank, 30.0]], dtype=torch.float64),
            "bits": torch.tensor([3.0 + rank, 1000.0], dtype=torch.float64),
            "bits_perfect": torch.tensor([1.0, 1000.0], dtype=torch.float64)}
    res = e2e.reduce(sums)
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


def test_two_rank_e2e_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_e2e_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert abs(res["ber"] - 7.0 / 2000.0) < 1e-15
    assert abs(res["ber_perfect_csi"] - 2.0 / 2000.0) < 1e-15
    want = 10 * np.log10(np.array([3.0 / 20.0, 4.0 / 41.0, 6.0 / 60.0]))
    assert np.allclose(res["nmse_db"], want, rtol=1e-12)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (here on the CPU:
    --workload selftest runs the launcher, the rendezvous and the all-reduce over gloo) and rank 0
    prints one JSON line for the whole job; a request that cannot be served exits non-zero."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "selftest"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16
    assert out["all_reduce_matches_single_process"] is True
    # under a launcher (WORLD_SIZE set) the same script is one rank and does not start others
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "selftest"],
                        env=env1, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
    import torch
    if torch.cuda.device_count() < 2:  # more ranks than GPUs: refused before anything starts
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                            capture_output=True, text=True, timeout=300)
        assert r2.returncode != 0 and "GPU(s) visible" in r2.stderr
