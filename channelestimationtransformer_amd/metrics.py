"""NMSE / BER bookkeeping: per-rank partial sums on the GPU, one all-reduce across ranks.

On ROCm tensors the sums come from the HIP reduction kernels (csrc/metrics.hip, fp64
accumulation); tensors on the CPU (host-side bookkeeping, the gloo tests) take the equivalent
torch expressions."""
import torch


def nmse_partial_sums(pred, truth):
    """Per prediction slot: [sum |truth - pred|^2, sum |truth|^2] over batch and antennas
    (the two sums whose ratio HPC/python/Evaluate.py:172-177 averages). Works on the real
    interleaved layout [B, P, 2N] or on complex [B, P, ...]. Returns float64 [2, P]."""
    if torch.is_complex(pred):
        pred, truth = torch.view_as_real(pred), torch.view_as_real(truth)
    B, P = pred.shape[0], pred.shape[1]
    if pred.is_cuda:  # GPU tensors: the HIP kernel, whatever the antenna count (no second arithmetic on the GPU)
        from . import ops  # noqa: F401
        return torch.ops.cet.nmse_sums(pred.reshape(B, P, -1), truth.reshape(B, P, -1))
    d = (truth.double() - pred.double()).reshape(B, P, -1)
    t = truth.double().reshape(B, P, -1)
    return torch.stack([(d * d).sum(dim=(0, 2)), (t * t).sum(dim=(0, 2))])


def qpsk_bit_errors(x_hat, x_true):
    """Hard-decision QPSK: bit errors and bit count, float64 [2]."""
    if x_hat.is_cuda and x_hat.dtype == torch.complex64 and x_true.dtype == torch.complex64:
        from . import ops  # noqa: F401
        return torch.ops.cet.qpsk_bit_errors(torch.view_as_real(x_hat.contiguous()),
                                             torch.view_as_real(x_true.contiguous()))
    e = ((x_hat.real > 0) != (x_true.real > 0)).sum() + ((x_hat.imag > 0) != (x_true.imag > 0)).sum()
    return torch.stack([e.double(), torch.tensor(2.0 * x_hat.numel(), dtype=torch.float64,
                                                 device=x_hat.device)])


def all_reduce_sums(t):
    """Sums a small float64 tensor over all ranks (RCCL when the default group is 'nccl');
    a no-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def nmse_db(sums):
    """10*log10(sum_err / sum_pow) per slot from the (all-reduced) partial sums."""
    return 10.0 * torch.log10(sums[0] / sums[1])


def NMSELoss(x_hat, x):
    """`sum((x - x_hat)^2) / sum(x_hat^2)`: the reference's metric normalises by the power of
    its FIRST argument (HPC/python/metrics.py:76-103, line 84)."""
    return ((x - x_hat) ** 2).sum() / (x_hat ** 2).sum()
