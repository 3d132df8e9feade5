"""Fused AdamW for the MI355X hot path.

Drop-in for the reference's ``torch.optim.AdamW(model.parameters(), lr=0.0001)`` (train.py:188,386,569) and its
``optimizer.step()`` (train.py:247,439,603): same constructor arguments, ``param_groups`` / ``state`` layout (so
``torch.optim.lr_scheduler.MultiStepLR`` at train.py:189-191 and ``state_dict()`` keep working), but one native
multi-tensor launch sequence per step (``seunet_adamw_step``, csrc/optim.hip) instead of PyTorch's per-op foreach
kernels.  There is no CPU path: parameters must be contiguous float32 CUDA tensors.
"""
import ctypes as C
from typing import Iterable, Tuple

import torch

from . import _lib


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, no amsgrad) on the HIP library."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, amsgrad: bool = False, *, maximize: bool = False):
        if amsgrad:
            raise ValueError("seunet AdamW: amsgrad is not implemented (the reference never enables it)")
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameters: {betas}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=maximize))

    def _init_group_state(self, group):
        """exp_avg / exp_avg_sq of a group live in one flat buffer each (views per parameter)."""
        todo = [p for p in group["params"] if p.grad is not None and len(self.state[p]) == 0]
        if not todo:
            return
        dev = todo[0].device
        total = sum(p.numel() for p in todo)
        flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in todo:
            n = p.numel()
            st = self.state[p]
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = flat_m[off:off + n].view_as(p)
            st["exp_avg_sq"] = flat_v[off:off + n].view_as(p)
            off += n

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            self._init_group_state(group)
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("seunet AdamW: parameters must be contiguous float32 CUDA tensors (no CPU path)")
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("seunet AdamW: gradients must be dense float32 tensors on the parameter's device")
                st = self.state[p]
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append((p, g if g.is_contiguous() else g.contiguous(), st))
            beta1, beta2 = group["betas"]
            for step, items in by_step.items():
                n = len(items)
                pa, ga, ma, va = ((C.c_void_p * n)() for _ in range(4))
                cnt = (C.c_longlong * n)()
                for i, (p, g, st) in enumerate(items):
                    pa[i], ga[i] = p.data_ptr(), g.data_ptr()
                    ma[i], va[i] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    cnt[i] = p.numel()
                with torch.cuda.device(items[0][0].device):
                    _lib.check(lib.seunet_adamw_step(pa, ga, ma, va, cnt, n, float(group["lr"]), float(beta1), float(beta2),
                                                     float(group["eps"]), float(group["weight_decay"]), step,
                                                     int(bool(group["maximize"])), _lib.stream_ptr()), "adamw_step")
        return loss
