"""Whole-volume sliding-window inference (reference prediction.py:39-49, 65-109; validation form train.py:631-699 with
the window table of data.py:731-773).  Window gathering, the float64 overlap accumulation and the final division run
on the device (csrc/window.hip through ``seunet_window_*``); there is one D2H copy of the finished volume, where the
reference copies 8 MB per window (prediction.py:104-107)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


def two_channel(data: np.ndarray):
    """prediction.py:39-49: two HU windows scaled to [0, 1]."""
    data = data.astype(float)
    c1 = (np.clip(data, -1000, 500) + 1000) / 1500
    c0 = (np.clip(data, -1024, 1024) + 1024) / 2048
    return c0, c1


def window_starts(dim: int, cube: int = 128, step: int = 64) -> List[int]:
    """prediction.py:80-100: stride ``step``; the last window is shifted back to end at ``dim``."""
    if dim < cube:
        raise ValueError(f"axis of {dim} voxels is shorter than the {cube}-voxel window (unsupported by the reference)")
    n = (dim - cube) // step + 1 if (dim - cube) % step == 0 else (dim - cube) // step + 2
    return [min(step * i, dim - cube) for i in range(n)]


def window_table(shape: Sequence[int], cube: int = 128, step: int = 64, pad_to_batch: Optional[int] = None
                 ) -> List[Tuple[int, int, int]]:
    """All (xl, yl, zl) window origins in the reference's loop order (x outer, z inner: prediction.py:83-100).
    ``pad_to_batch``: append copies of window 0 until the count is a multiple of it (``SegValCropData.crop_pos``,
    data.py:764-765 -- the validation / test loops then run AND accumulate those copies)."""
    X, Y, Z = shape
    pos = [(a, b, c) for a in window_starts(X, cube, step) for b in window_starts(Y, cube, step)
           for c in window_starts(Z, cube, step)]
    if pad_to_batch:
        while len(pos) % pad_to_batch:
            pos.append(pos[0])
    return pos


def _assemble(model, x: torch.Tensor, pos, cube: int, step: int, batch: int, n_dup: int, graph: bool = False, group=None,
              sigmoid: bool = True) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("HIP path needs a GPU tensor (no CPU fallback)")
    if x.dim() != 5 or x.shape[0] != 1:
        raise ValueError(f"expected one case (1, C, X, Y, Z), got {tuple(x.shape)}")
    if getattr(model, "n_classes", 1) != 1:
        raise NotImplementedError("the window loops of the reference accumulate ONE probability volume (prediction.py:104-107): "
                                  "n_classes must be 1")
    lib = _lib.load()
    x = x.contiguous().float()
    _, C_, X, Y, Z = x.shape
    max_call = 64                                       # windows per native call (kernel-argument table)
    # one case over several GPUs (one process per GPU, every rank holds the volume): the WINDOWS of the list are dealt
    # round-robin by their index (rank r takes windows r, r + world, ...: a partition that does not depend on any rank's
    # batch size, which auto_batch() derives from that rank's free memory), each rank batches and accumulates its own
    # windows, and the float64 accumulators are summed by one all-reduce before the division by the overlap count -- the
    # only exchange step of the loop
    world, rank, pg = 1, 0, None
    if group is not None and group is not False:
        import torch.distributed as dist
        pg = None if group is True else group
        world, rank = dist.get_world_size(pg), dist.get_rank(pg)
        pos = pos[rank::world]
    with torch.cuda.device(x.device):
        st = _lib.stream_ptr()
        acc = torch.zeros((X, Y, Z), dtype=torch.float64, device=x.device)
        # full batches replay ONE recorded HIP graph of the forward pass on fixed buffers; a ragged last batch (another
        # descriptor) runs eagerly
        cap = None
        if graph and len(pos) >= 2 * batch:
            from .SE_UNet import CapturedForward
            cap = CapturedForward(model, batch, (cube, cube, cube), decoder_only=True)
        for i in range(0, len(pos), batch):
            chunk = pos[i:i + batch]
            use_cap = cap is not None and len(chunk) == batch
            xin = cap.x if use_cap else torch.empty((len(chunk), C_, cube, cube, cube), dtype=torch.float32, device=x.device)
            for j in range(0, len(chunk), max_call):
                sub = chunk[j:j + max_call]
                arr = _lib.int_array([v for p in sub for v in p])
                _lib.check(lib.seunet_window_gather(x.data_ptr(), C_, X, Y, Z, cube, len(sub), arr, xin[j:].data_ptr(), st), "window_gather")
            # logits of the decoder head (prediction.py:103 `p0, p = model(...)` keeps only p): the inference form of the
            # forward, which does not evaluate the encoder head
            p = cap()[1] if use_cap else model.predict_logits(xin)
            for j in range(0, len(chunk), max_call):
                sub = chunk[j:j + max_call]
                arr = _lib.int_array([v for q in sub for v in q])
                _lib.check(lib.seunet_window_accumulate(p[j:].data_ptr(), 1 if sigmoid else 0, len(sub), arr, cube, acc.data_ptr(), X, Y, Z, st),
                           "window_accumulate")
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=pg)
        xs, ys, zs = window_starts(X, cube, step), window_starts(Y, cube, step), window_starts(Z, cube, step)
        out = torch.empty_like(acc)
        _lib.check(lib.seunet_window_finalize(acc.data_ptr(), X, Y, Z, cube, len(xs), _lib.int_array(xs), len(ys), _lib.int_array(ys),
                                              len(zs), _lib.int_array(zs), n_dup, out.data_ptr(), st), "window_finalize")
    return out


def auto_batch(model, device, cube: int = 128, cap: int = 16) -> int:
    """Windows per network call for the prediction-form loop when the caller does not say: the reference runs one window
    per call (prediction.py:103); in eval mode any batch gives the same volume up to f32 rounding of the InstanceNorm
    statistics (InstanceNorm is per sample; the number of partial sums per sample follows the batch size), and larger
    batches amortise the small coarse-level kernels (512^3 on MI355X: 0.65 s at 1, 0.49 s at 4, 0.455 s at 16).  The largest
    power of two <= ``cap`` whose workspace (3.3 GB per 128^3 window in 16-bit storage) fits the free HBM with room to spare;
    1 under ``model.train()`` (DropLayer's scale depends on the batch size, SE_UNet.py:91-96)."""
    if model.training:
        return 1
    import ctypes as C
    from .SE_UNet import make_desc
    lib = _lib.load()
    desc = make_desc(1, model.in_channel, model.n_classes, cube, cube, cube, model.width_mult, _lib.dtype_code(model.act_dtype),
                     model.conv_impl, model.negative_slope)
    per = lib.seunet_net_workspace_bytes(C.byref(desc))
    if per == 0:
        return 1
    free, _ = torch.cuda.mem_get_info(device)
    b = cap
    while b > 1 and b * per * 1.25 + (2 << 30) > free:
        b //= 2
    return max(b, 1)


@torch.no_grad()
def sliding_window_predict(model, x: torch.Tensor, cube: int = 128, step: int = 64, batch: Optional[int] = None,
                           return_tensor: bool = False, graph: bool = False, group=None, sigmoid: bool = True):
    """prediction.py:78-109.  x: (1, C, X, Y, Z) on the GPU.  Returns the overlap-averaged sigmoid(pred1) volume as
    float64 numpy (like the reference's host accumulators), or the float64 CUDA tensor with ``return_tensor=True`` (what
    ``double_threshold_iteration`` takes next, prediction.py:110).  ``batch`` windows go through the network per call
    (the reference uses 1; in eval mode the result does not depend on it beyond f32 rounding of the statistics, because
    InstanceNorm is per sample; ``None`` = ``auto_batch``).  ``graph``: replay the forward pass as one recorded HIP graph per
    batch (``CapturedForward``: same kernels, same bits as the eager call at the same batch size).  Off by default: on
    MI355X the loop is bound by the kernels, not by their launches (512^3: 0.499 s replayed vs 0.493 s launched one by
    one at batch 4, 0.656 vs 0.648 s at batch 1); it pays only when the host thread is slow or busy.
    ``group`` (``True`` = the default process group, or a ``ProcessGroup``): shard the windows of this ONE case over the
    ranks (every rank passes the same volume and gets the full result; the windows are dealt round-robin by index -- ranks
    may use different batch sizes -- and the float64 accumulators all-reduced once, RCCL on GPUs).  Eval mode only gives rank-count-independent results
    (DropLayer draws are per call).
    ``sigmoid=False``: the variant of save_gradients.py:129-137 / weight_br.py:95-102, which accumulate the decoder head's RAW
    logits (``pred += p``), average them and threshold the average at 0.5 (those scripts run under ``case_net.train()``, one
    window per call: ``model.train()`` + ``batch=None`` gives exactly that)."""
    pos = window_table(x.shape[2:], cube, step)
    if batch is None:
        batch = auto_batch(model, x.device, cube) if x.is_cuda else 1
    out = _assemble(model, x, pos, cube, step, batch, 0, graph, group, sigmoid)
    return out if return_tensor else out.cpu().numpy()


@torch.no_grad()
def sliding_window_validate(model, x: torch.Tensor, batch: int = 24, cube: int = 128, step: int = 64,
                            return_tensor: bool = False, graph: bool = False, sigmoid: bool = True):
    """The validation / test form of the loop (train.py:682-693 with ``SegValCropData``, data.py:731-773; test.py:151-161
    with batch 8): the window list is padded with copies of window 0 to a multiple of ``batch`` and the copies are run and
    accumulated like any other window.  The reference runs this loop under ``model.train()`` (train.py:632):
    DropLayer is then active, its scale depends on the batch size, and each copy of window 0 gets its own draw -- call
    ``model.train()`` / ``model.eval()`` yourself, as the reference does.  ``sigmoid=False`` accumulates raw logits."""
    pos = window_table(x.shape[2:], cube, step, pad_to_batch=batch)
    n_real = len(window_table(x.shape[2:], cube, step))
    out = _assemble(model, x, pos, cube, step, batch, len(pos) - n_real, graph, None, sigmoid)
    return out if return_tensor else out.cpu().numpy()
