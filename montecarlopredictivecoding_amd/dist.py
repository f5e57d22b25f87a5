"""Sharding of Langevin chains over the GPUs of a node (one process per GPU, RCCL over xGMI).

Chains are independent given the weights for a whole ``train_on_batch`` call, so the batch dimension
is partitioned with NO per-step communication.  The only exchange is one
sum-all-reduce per learning call of the flat Hebbian gradient bucket (W0,b0,W1,b1,...), performed
BEFORE the 1/(n_acc * B_global) normalisation's optimizer step; energies are per-shard partial sums
that the caller may reduce the same way.  ``torch.distributed`` (backend "nccl" = RCCL on ROCm,
"gloo" in CPU tests) is used as plumbing; the reference has no counterpart (it is single-device).
"""
import typing

import torch


def shard_bounds(total: int, rank: int, world: int) -> typing.Tuple[int, int]:
    """[begin, end) of the chains owned by ``rank`` when ``total`` chains are split as evenly as possible
    (the first ``total % world`` ranks get one extra chain)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def flat_param_count(linears) -> int:
    return sum(lin.weight.numel() + (0 if lin.bias is None else lin.bias.numel()) for lin in linears)


def allreduce_flat(flat: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the gradient bucket over all shards in place (no-op without an initialised process group)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        if flat.is_cuda and torch.distributed.get_backend(group) == "gloo":
            # gloo is the CPU backend (tests, rehearsals of several ranks on one GPU): stage the bucket through the host
            host = flat.cpu()
            torch.distributed.all_reduce(host, op=torch.distributed.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=group)
    return flat


def sum_over_group(value: int, group=None) -> int:
    """Sum of a per-rank integer over the group (the global batch of a sharded job); the value itself without one."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return int(value)
    backend = torch.distributed.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return int(t.item())


def assign_flat_grads(linears, flat: torch.Tensor) -> None:
    """Point every parameter's ``.grad`` at its slice of the flat bucket (layout of mcpc_read_param_grads_flat)."""
    if flat.numel() != flat_param_count(linears):
        raise ValueError(f"flat bucket has {flat.numel()} floats, parameters need {flat_param_count(linears)}")
    off = 0
    for lin in linears:
        n = lin.weight.numel()
        lin.weight.grad = flat[off:off + n].view_as(lin.weight)
        off += n
        if lin.bias is not None:
            n = lin.bias.numel()
            lin.bias.grad = flat[off:off + n].view_as(lin.bias)
            off += n


def grad_scale(n_accumulate: int, global_batch: int) -> float:
    """Normalisation the reference applies at an update step (pc_trainer.py:905-913)."""
    return 1.0 / (n_accumulate * global_batch) if n_accumulate > 0 else 1.0 / global_batch
