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
    """The single contiguous buffer the gradients are views of, if together they tile one range of one storage (in any order:
    ``SE_UNet``'s backward lays the decoder's gradients out last)."""
    if not grads:
        return None
    base = grads[0]
    try:
        storage_ptr = base.untyped_storage().data_ptr()
    except Exception:
        return None
    spans = []
    for g in grads:
        if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != storage_ptr:
            return None
        spans.append((g.data_ptr(), g.numel() * 4))
    spans.sort()
    expect = spans[0][0]
    for start, nbytes in spans:
        if start != expect:
            return None
        expect += nbytes
    total = (expect - spans[0][0]) // 4
    off = (spans[0][0] - storage_ptr) // 4
    return torch.empty(0, dtype=torch.float32, device=base.device).set_(base.untyped_storage(), off, (total,))


class GradSync:
    """The gradient exchange of a data-parallel step, overlapped with the backward pass: ``model.grad_sync = GradSync(group)``.

    ``SE_UNet``'s backward then (1) has the library record an event once the decoder's parameter gradients are final
    (``seunet_net_backward_ev``), (2) all-reduces that tail of the flat gradient buffer on a side stream behind the event, while
    the encoder is still being differentiated on the compute stream, (3) all-reduces the head of the buffer (encoder + the two
    1x1x1 heads) on the compute stream when the backward pass is done, and (4) makes the compute stream wait for the side
    stream: when ``loss.backward()`` returns, the stream-ordered gradients are the global sums (no ``allreduce_gradients`` call).
    Two collectives of ~2.4 MB and ~3.7 MB instead of one of 6.08 MB; the first one is off the critical path.
    ``elapsed_ms()``: stream time of the exchange that was NOT hidden (the second collective + the join), for ``bench.py``."""

    def __init__(self, group=None, average: bool = False, timing: bool = False):
        self.group, self.average, self.timing = group, average, timing
        self._side = None
        self._marks = []

    def decoder_event(self):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return None
        # A torch event has no HIP handle until its first record(): ``ev.cuda_event`` would be 0, the library would record nothing
        # and ``wait_event`` on a never-recorded event is a no-op -- the side stream would reduce the decoder's gradients before
        # they are written.  Record once here (the library's record at the decoder boundary supersedes it).
        ev = torch.cuda.Event()
        ev.record()
        if not ev.cuda_event:
            raise RuntimeError("GradSync: could not create the decoder-boundary event")
        return ev

    def exchange(self, flat: torch.Tensor, split: int, ev) -> None:
        if ev is None:
            return
        cur = torch.cuda.current_stream(flat.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=flat.device)
        world = dist.get_world_size(self.group)
        tail, head = flat[split:], flat[:split]
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            if tail.numel():
                dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
                tail.record_stream(self._side)
        e0 = e1 = None
        if self.timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        if head.numel():
            dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group)
        cur.wait_stream(self._side)
        if self.average:
            flat.div_(world)
        if self.timing:
            e1.record(cur)
            self._marks.append((e0, e1))

    def elapsed_ms(self) -> List[float]:
        out = [a.elapsed_time(b) for a, b in self._marks]
        self._marks.clear()
        return out


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
