"""Whole-volume sliding-window inference (reference prediction.py:39-49, 65-109; data.py:731-773)."""
from __future__ import annotations

from typing import List

import numpy as np
import torch


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


@torch.no_grad()
def sliding_window_predict(model, x: torch.Tensor, cube: int = 128, step: int = 64, batch: int = 1) -> np.ndarray:
    """x: (1, C, X, Y, Z) on the GPU.  Returns the overlap-averaged sigmoid(pred1) volume (float64 numpy,
    like the reference's host accumulators) -- accumulated on the device in float64, one D2H at the end
    instead of the reference's 8 MB copy per window (prediction.py:104-107)."""
    if not x.is_cuda:
        raise RuntimeError("HIP path needs a GPU tensor (no CPU fallback)")
    _, _, X, Y, Z = x.shape
    acc = torch.zeros((X, Y, Z), dtype=torch.float64, device=x.device)
    cnt = torch.zeros((X, Y, Z), dtype=torch.float64, device=x.device)
    pos = [(a, b, c) for a in window_starts(X, cube, step) for b in window_starts(Y, cube, step)
           for c in window_starts(Z, cube, step)]
    for i in range(0, len(pos), batch):
        chunk = pos[i:i + batch]
        xin = torch.cat([x[:, :, a:a + cube, b:b + cube, c:c + cube] for a, b, c in chunk], 0)
        _, p = model(xin)
        p = torch.sigmoid(p).double()
        for k, (a, b, c) in enumerate(chunk):
            acc[a:a + cube, b:b + cube, c:c + cube] += p[k, 0]
            cnt[a:a + cube, b:b + cube, c:c + cube] += 1
    return (acc / cnt).cpu().numpy()
