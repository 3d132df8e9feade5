"""One-process-per-GPU data parallelism for the SE-UNet step: a single flat-bucket RCCL all-reduce.

The reference wraps the model in single-process ``torch.nn.DataParallel`` (train.py:197,396,577).
Here every rank owns one MI355X and a B/N shard of the patch batch; the only exchange steps are
  (1) the 7 loss sums per head (``losses.*(group=...)``), so the objective stays the global-batch ratio;
  (2) ONE all-reduce(SUM) of the 1,520,314-element gradient bucket (6.08 MB) over xGMI.
``SE_UNet``'s backward already writes every parameter gradient into one contiguous buffer, so (2) is
zero-copy when gradients are fresh (``zero_grad(set_to_none=True)``, PyTorch's default).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> int:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); returns the local rank."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" is RCCL on ROCm
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend)
    return local


def _flat_view(grads: List[torch.Tensor]) -> Optional[torch.Tensor]:
    """The single contiguous buffer the gradients are views of, if they are laid out back to back."""
    if not grads:
        return None
    base = grads[0]
    try:
        storage_ptr = base.untyped_storage().data_ptr()
    except Exception:
        return None
    start = base.data_ptr()
    expect = start
    for g in grads:
        if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != storage_ptr \
                or g.data_ptr() != expect:
            return None
        expect += g.numel() * 4
    total = (expect - start) // 4
    off = (start - storage_ptr) // 4
    return torch.empty(0, dtype=torch.float32, device=base.device).set_(base.untyped_storage(), off, (total,))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, average: bool = False) -> int:
    """Sum (or average) the gradients of ``params`` across ranks with one collective.  Returns the number of
    elements reduced.  Parameters without a gradient (the dead ``dc62`` block) are skipped on every rank alike."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return sum(g.numel() for g in grads)
    flat = _flat_view(grads)
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(dist.get_world_size(group))
        return flat.numel()
    bucket = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    if average:
        bucket.div_(dist.get_world_size(group))
    off = 0
    for g in grads:
        g.copy_(bucket[off:off + g.numel()].view_as(g))
        off += g.numel()
    return bucket.numel()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s weights (DataParallel's per-forward broadcast, done once)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for p in module.parameters():
        dist.broadcast(p.data, src=src, group=group)
