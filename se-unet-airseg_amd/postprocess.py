"""Prediction post-processing on the GPU: the reference's ``double_threshold_iteration`` and the 15 %
border clearing of prediction.py:111-114.

The reference has three copies of the function.  prediction.py:13-37 keeps ``pred*255`` in float64; train.py:25-49 and
test.py:18-42 (the validation / test loops) round it to float32 first (train.py:31, test.py:24), which moves voxels
within a float32 ulp of a threshold to the other class.  ``pred_dtype`` selects the copy ("float64" = prediction.py,
the default; "float32" = train.py / test.py).

The reference's version is a pure-Python triple loop over every voxel with 26 neighbour look-ups (hours for a 512^3
volume); here it is ``seunet_dti`` (csrc/dti.hip), which reproduces the single raster-order sweep bit for bit.
There is no CPU path."""
from typing import Union

import numpy as np
import torch

from . import _lib


def double_threshold_iteration(pred: Union[np.ndarray, torch.Tensor], h_thresh: float, l_thresh: float,
                               pred_dtype: str = "float64"):
    """Same arguments and meaning as prediction.py:13 (``pred_dtype="float32"``: train.py:25 / test.py:18).
    ``pred``: (h, w, z) probabilities.
    numpy in -> float64 numpy of zeros / ones out (what the reference returns, ``gbin / 255``);
    CUDA tensor in -> uint8 CUDA tensor out (stays on the device)."""
    as_numpy = isinstance(pred, np.ndarray)
    if as_numpy:
        if not torch.cuda.is_available():
            raise RuntimeError("seunet double_threshold_iteration needs a GPU (no CPU path)")
        t = torch.from_numpy(np.ascontiguousarray(pred, dtype=np.float64)).cuda()
    else:
        if not pred.is_cuda:
            raise RuntimeError("seunet double_threshold_iteration needs a CUDA tensor or a numpy array (no CPU path)")
        t = pred.detach().to(torch.float64).contiguous()
    if t.dim() != 3:
        raise ValueError(f"double_threshold_iteration expects a 3-D volume, got shape {tuple(t.shape)}")
    if pred_dtype not in ("float64", "float32"):
        raise ValueError(f"pred_dtype {pred_dtype!r}: 'float64' (prediction.py:19) or 'float32' (train.py:31, test.py:24)")
    code = _lib.DTI_F32 if pred_dtype == "float32" else _lib.DTI_F64
    lib = _lib.load()
    h, w, z = (int(v) for v in t.shape)
    with torch.cuda.device(t.device):
        nbytes = lib.seunet_dti_workspace_bytes(h, w, z)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
        out = torch.empty((h, w, z), dtype=torch.uint8, device=t.device)
        _lib.check(lib.seunet_dti(t.data_ptr(), h, w, z, float(h_thresh), float(l_thresh), code, out.data_ptr(), ws.data_ptr(),
                                  nbytes, _lib.stream_ptr()), "dti")
    return out.cpu().numpy().astype(np.float64) if as_numpy else out


def zero_borders(pred, lo: float = 0.15, hi: float = 0.85):
    """prediction.py:111-114: clear the slabs below 15 % and above 85 % of the first two axes (in place; numpy or
    tensor).  The two literals are the reference's own (int(0.15 * n), int(0.85 * n))."""
    a, b = pred.shape[0], pred.shape[1]
    pred[0:int(lo * a), :, :] = 0
    pred[int(hi * a):, :, :] = 0
    pred[:, 0:int(lo * b), :] = 0
    pred[:, int(hi * b):, :] = 0
    return pred


def postprocess_prediction(pred, h_thresh: float = 0.5, l_thresh: float = 0.4):
    """prediction.py:110-114: double threshold (0.5 / 0.4), then border clearing.  The largest-component filter that
    follows in the reference (``maximum_3d``, util.py:58-75) is SURVEY 8(f4) and not part of this package."""
    return zero_borders(double_threshold_iteration(pred, h_thresh, l_thresh))
