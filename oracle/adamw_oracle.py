"""CPU oracle for the optimizer step.  TEST INFRASTRUCTURE ONLY -- only tests/ may import it.

The reference takes its optimizer from a third-party dependency: ``torch.optim.AdamW(model.parameters(), lr=0.0001)``
(train.py:188,386,569; pytorch pinned at 2.0.1 by environment.yml:78, source not under /root/reference) driven by
``torch.optim.lr_scheduler.MultiStepLR(milestones=[40, 60], gamma=0.1)`` (train.py:189-191).  This file restates the
published single-tensor algorithm (torch/optim/adamw.py ``_single_tensor_adamw``, amsgrad=False, maximize=False) in
numpy float32, operation by operation:

    param.mul_(1 - lr * weight_decay)
    exp_avg.lerp_(grad, 1 - beta1)                      # == exp_avg + (grad - exp_avg) * (1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bias_correction1 = 1 - beta1 ** step ; bias_correction2 = 1 - beta2 ** step
    step_size = lr / bias_correction1
    denom = (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-step_size)

Pinning: ``oracle/make_golden_adamw.py`` runs the real ``torch.optim.AdamW`` + ``MultiStepLR`` of this container's
PyTorch on CPU and stores inputs and results in ``tests/golden/adamw_known.npz``; ``tests/test_oracle_golden.py``
checks this restatement against that fixture on every run.
"""
import math
from typing import Dict, List

import numpy as np

F32 = np.float32


def adamw_step(param: np.ndarray, grad: np.ndarray, exp_avg: np.ndarray, exp_avg_sq: np.ndarray, step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-2) -> None:
    """One in-place AdamW update of float32 arrays; ``step`` counts from 1."""
    assert param.dtype == F32 and grad.dtype == F32 and exp_avg.dtype == F32 and exp_avg_sq.dtype == F32
    param *= F32(1.0 - lr * weight_decay)
    exp_avg += (grad - exp_avg) * F32(1.0 - beta1)
    exp_avg_sq *= F32(beta2)
    exp_avg_sq += F32(1.0 - beta2) * grad * grad
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = np.sqrt(exp_avg_sq) / F32(math.sqrt(bc2)) + F32(eps)
    param -= F32(step_size) * (exp_avg / denom)


def multistep_lr(base_lr: float, epoch: int, milestones=(40, 60), gamma: float = 0.1) -> float:
    """MultiStepLR: lr after ``epoch`` scheduler steps (train.py:189-191)."""
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)


def run(params: List[np.ndarray], grads_per_step: List[List[np.ndarray]], lrs: List[float], **hyper) -> Dict[str, List[np.ndarray]]:
    """Apply len(grads_per_step) steps to copies of ``params``; returns final params and moments."""
    p = [a.astype(F32).copy() for a in params]
    m = [np.zeros_like(a) for a in p]
    v = [np.zeros_like(a) for a in p]
    for t, (grads, lr) in enumerate(zip(grads_per_step, lrs), start=1):
        for i in range(len(p)):
            adamw_step(p[i], grads[i].astype(F32), m[i], v[i], t, lr, **hyper)
    return {"params": p, "exp_avg": m, "exp_avg_sq": v}
