"""Losses of the reference training loops (train.py:51-76) on HIP reduction kernels.

``dice_loss`` / ``general_union_loss_lib`` / ``atr_loss`` keep the reference signatures: they take
*probabilities* (the reference applies ``torch.sigmoid`` first, train.py:595-596) of any
broadcast-compatible shape and return a scalar tensor that supports ``.backward()``.
``fused_stage_loss`` is the same arithmetic taken from raw logits in one pass per head (sigmoid and
its derivative folded into the kernels).

All three are ratios of WHOLE-BATCH sums.  Under one-process-per-GPU data parallelism
pass ``group=`` (a torch.distributed process group): the 7 partial sums are all-reduced before the
ratio is formed, which reproduces the reference's single-process global-batch objective exactly
(then SUM, not average, the parameter gradients across ranks).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _prep(t: Optional[torch.Tensor], like: torch.Tensor) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return torch.broadcast_to(t.to(like.device, torch.float32), like.shape).contiguous()


def _launch_sums(pred, apply_sigmoid, target, weight, skel, terms, sums):
    lib = _lib.load()
    with torch.cuda.device(pred.device):    # launch on pred's GPU even when it is not the current device
        partial = torch.empty(lib.seunet_loss_partial_floats(), dtype=torch.float32, device=pred.device)
        _lib.check(lib.seunet_loss_sums(pred.data_ptr(), int(apply_sigmoid), target.data_ptr(), _lib.ptr(weight), _lib.ptr(skel),
                                        pred.numel(), partial.data_ptr(), sums.data_ptr(), int(terms), _lib.stream_ptr()), "loss_sums")


def _reduce(sums, group):
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group if group is not True else None)


def _sums(pred, apply_sigmoid, target, weight, skel, group, terms=0):
    sums = torch.empty(_lib.LOSS_NSUMS, dtype=torch.float64, device=pred.device)
    _launch_sums(pred, apply_sigmoid, target, weight, skel, terms, sums)
    _reduce(sums, group)
    return sums


def _terms(c_dice, c_gul, c_atr):
    # only the sums of the losses with a non-zero coefficient are formed (the others stay 0 and are multiplied by 0)
    return (1 if c_dice else 0) | (2 if c_gul else 0) | (4 if c_atr else 0)


def _value(sums, c_dice, c_gul, c_atr):
    """The ratio arithmetic on the 7 whole-batch sums, as torch scalar operations (the readable definition; a dozen one-element
    kernels on a GPU, which is why the losses below call ``_value_dev``: the same operations in one launch)."""
    out = sums.new_zeros(())
    if c_dice:
        out = out + c_dice * (1 - (2 * sums[0] + 1) / (sums[1] + sums[2] + 1))
    if c_gul:
        out = out + c_gul * (1 - (sums[3] + 1) / (sums[4] + 1))
    if c_atr:
        out = out + c_atr * (1 - (sums[5] + 1) / (sums[6] + 1))
    return out.to(torch.float32)


def _value_dev(sums, coef, sums1=None, coef1=(0.0, 0.0, 0.0)):
    """f32 scalar tensor: the loss of one head, or the sum of two heads' losses, from the whole-batch sums (one launch)."""
    out = torch.empty((), dtype=torch.float32, device=sums.device)
    with torch.cuda.device(sums.device):
        _lib.check(_lib.load().seunet_loss_value(sums.data_ptr(), float(coef[0]), float(coef[1]), float(coef[2]), _lib.ptr(sums1),
                                                 float(coef1[0]), float(coef1[1]), float(coef1[2]), out.data_ptr(),
                                                 _lib.stream_ptr()), "loss_value")
    return out


def _launch_grad(p, sig, t, w, s, sums, coef, g, gp):
    with torch.cuda.device(p.device):
        gs = g.detach().reshape(1).to(p.device, torch.float32).contiguous()
        _lib.check(_lib.load().seunet_loss_grad(p.data_ptr(), int(sig), t.data_ptr(), _lib.ptr(w), _lib.ptr(s), p.numel(),
                                                sums.data_ptr(), coef[0], coef[1], coef[2], 1.0, gs.data_ptr(), gp.data_ptr(),
                                                _lib.stream_ptr()), "loss_grad")


class _RatioLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, skel, c_dice, c_gul, c_atr, apply_sigmoid, group):
        if not pred.is_cuda:
            raise RuntimeError("HIP losses need GPU tensors (no CPU fallback; the CPU oracle is oracle/seunet_oracle.py)")
        p = pred.detach().contiguous().float()
        t, w, s = _prep(target, p), _prep(weight, p), _prep(skel, p)
        sums = _sums(p, apply_sigmoid, t, w, s, group, _terms(c_dice, c_gul, c_atr))
        ctx.saved = (p, t, w, s, sums)
        ctx.coef = (float(c_dice), float(c_gul), float(c_atr), bool(apply_sigmoid))
        ctx.shape = pred.shape
        return _value_dev(sums, ctx.coef)

    @staticmethod
    def backward(ctx, g):
        p, t, w, s, sums = ctx.saved
        gp = torch.empty_like(p)
        _launch_grad(p, ctx.coef[3], t, w, s, sums, ctx.coef, g, gp)
        return gp.reshape(ctx.shape), None, None, None, None, None, None, None, None


class _StageLoss(torch.autograd.Function):
    """loss(head A; coefficients ca) + loss(head B; cb) from raw logits: two reduction passes, ONE exchange of the 14 sums
    under data parallelism, one value launch; the backward is the two gradient passes."""

    @staticmethod
    def forward(ctx, pred_a, pred_b, target, weight, skel, ca, cb, group):
        if not (pred_a.is_cuda and pred_b.is_cuda):
            raise RuntimeError("HIP losses need GPU tensors (no CPU fallback; the CPU oracle is oracle/seunet_oracle.py)")
        pa, pb = pred_a.detach().contiguous().float(), pred_b.detach().contiguous().float()
        t, w, s = _prep(target, pa), _prep(weight, pa), _prep(skel, pa)
        sums = torch.empty((2, _lib.LOSS_NSUMS), dtype=torch.float64, device=pa.device)
        _launch_sums(pa, True, t, w, s, _terms(*ca), sums[0])
        _launch_sums(pb, True, t, w, s, _terms(*cb), sums[1])
        _reduce(sums, group)
        ctx.saved = (pa, pb, t, w, s, sums)
        ctx.coef = (tuple(float(c) for c in ca), tuple(float(c) for c in cb))
        ctx.shapes = (pred_a.shape, pred_b.shape)
        return _value_dev(sums[0], ctx.coef[0], sums[1], ctx.coef[1])

    @staticmethod
    def backward(ctx, g):
        pa, pb, t, w, s, sums = ctx.saved
        ga, gb = torch.empty_like(pa), torch.empty_like(pb)
        _launch_grad(pa, True, t, w, s, sums[0], ctx.coef[0], g, ga)
        _launch_grad(pb, True, t, w, s, sums[1], ctx.coef[1], g, gb)
        return ga.reshape(ctx.shapes[0]), gb.reshape(ctx.shapes[1]), None, None, None, None, None, None


def dice_loss(pred, target, group=None):
    """train.py:51-57."""
    return _RatioLoss.apply(pred, target, None, None, 1.0, 0.0, 0.0, False, group)


def general_union_loss_lib(pred, target, weight, group=None):
    """train.py:59-68 (alpha 0.2, exponent 0.7)."""
    return _RatioLoss.apply(pred, target, weight, None, 0.0, 1.0, 0.0, False, group)


def atr_loss(pred, target, skel, weight, group=None):
    """train.py:70-76 (``target`` is ignored by the reference and here)."""
    return _RatioLoss.apply(pred, target, weight, skel, 0.0, 0.0, 1.0, False, group)


def fused_logit_loss(logits, target, weight=None, skel=None, c_dice=1.0, c_gul=0.0, c_atr=0.0, group=None):
    """c_dice*dice + c_gul*GUL + c_atr*ATR of sigmoid(logits), one reduction pass + one gradient pass."""
    return _RatioLoss.apply(logits, target, weight, skel, c_dice, c_gul, c_atr, True, group)


def fused_stage_loss(stage: int, pred_en, pred_de, label, weight=None, skel=None, group=None):
    """Loss of training stage 1/2/3 from the raw logits of both heads
    (train.py:595-599 ; 429-435 ; 235-243): decoder-head term + encoder-head term, in that order."""
    if pred_en.shape != pred_de.shape:
        raise ValueError("fused_stage_loss: the two heads' logits must have the same shape")
    if stage == 1:
        ca, cb = (1.0, 0.0, 0.0), (1.0, 0.0, 0.0)
    elif stage == 2:
        ca, cb, skel = (0.0, 1.0, 0.0), (0.0, 0.5, 0.0), None
    else:
        ca, cb = (0.0, 1.0, 0.5), (0.0, 0.5, 0.5)
    return _StageLoss.apply(pred_de, pred_en, label, weight, skel, ca, cb, group)
