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
    """prediction.py:110-114: double threshold (0.5 / 0.4), then border clearing.  ``maximum_3d`` (prediction.py:116)
    is the next step: ``maximum_3d(postprocess_prediction(pred))``."""
    return zero_borders(double_threshold_iteration(pred, h_thresh, l_thresh))


# ----------------------------------------------------------------------------------------------------------------------
# largest 26-connected component (+ hole filling) and the ATM'22 metrics, on the device
# ----------------------------------------------------------------------------------------------------------------------
def _as_u8_cuda(a, name):
    if isinstance(a, np.ndarray):
        if not torch.cuda.is_available():
            raise RuntimeError(f"seunet {name}: needs a GPU (no CPU path)")
        return torch.from_numpy(np.ascontiguousarray(a != 0).view(np.uint8)).cuda(), True
    if not a.is_cuda:
        raise RuntimeError(f"seunet {name}: needs a CUDA tensor or a numpy array (no CPU path)")
    return (a != 0).to(torch.uint8).contiguous(), False


def _largest(vol, rule, name):
    t, as_numpy = _as_u8_cuda(vol, name)
    if t.dim() != 3:
        raise ValueError(f"{name} expects a 3-D volume, got shape {tuple(t.shape)}")
    lib = _lib.load()
    h, w, z = (int(v) for v in t.shape)
    with torch.cuda.device(t.device):
        nbytes = lib.seunet_cc_workspace_bytes(h, w, z)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
        out = torch.empty((h, w, z), dtype=torch.uint8, device=t.device)
        status = torch.zeros(1, dtype=torch.int32, device=t.device)
        _lib.check(lib.seunet_largest_component(t.data_ptr(), h, w, z, rule, out.data_ptr(), status.data_ptr(), ws.data_ptr(), nbytes,
                                                _lib.stream_ptr()), name)
    return out, int(status.item()), as_numpy


def largest_component(pred):
    """The component rule of ``evaluation_case`` (train.py:749-757, test.py:243-251): the 26-connected component with the
    most voxels as a uint8 mask (``large_cd``); an empty prediction gives an empty mask (``pred.astype(np.uint8)``).
    numpy in -> uint8 numpy out; CUDA tensor in -> uint8 CUDA tensor out."""
    out, _, as_numpy = _largest(pred, _lib.CC_EVALUATION, "largest_component")
    return out.cpu().numpy() if as_numpy else out


def maximum_3d(region01):
    """util.py:58-75 (called at prediction.py:116): largest 26-connected component, replaced by the second largest when it
    misses the three test slices of the last axis, then ``binary_fill_holes``.  Returns a bool array (numpy in) like the
    reference, or a uint8 CUDA tensor (tensor in).  Raises IndexError where the reference does (no component / no second
    component)."""
    out, status, as_numpy = _largest(region01, _lib.CC_MAXIMUM_3D, "maximum_3d")
    if status != 0:
        raise IndexError("list index out of range (maximum_3d: %s, util.py:%d)" %
                         (("the volume has no foreground component", 65) if status == 1 else ("no second component to fall back to", 71)))
    return out.cpu().numpy().astype(bool) if as_numpy else out


class MetricSums:
    """The integer sums metrics.py:14-78 is built from, computed in one pass on the device (``seunet_metric_sums``)."""

    def __init__(self, pred, label=None, skeleton=None, parsing=None, nbins: int = 4096):
        p, _ = _as_u8_cuda(pred, "metrics")
        lab = _as_u8_cuda(label, "metrics")[0] if label is not None else None
        sk = _as_u8_cuda(skeleton, "metrics")[0] if skeleton is not None else None
        pa = None
        if parsing is not None:
            pa = (torch.from_numpy(np.ascontiguousarray(parsing)).cuda() if isinstance(parsing, np.ndarray) else parsing)
            pa = pa.to(device=p.device, dtype=torch.int32).contiguous()
        for t in (lab, sk, pa):
            if t is not None and t.numel() != p.numel():
                raise ValueError("metrics: all volumes must have the prediction's size")
        lib = _lib.load()
        with torch.cuda.device(p.device):
            nb = lib.seunet_metric_out_bytes(nbins)
            out = torch.empty(nb, dtype=torch.uint8, device=p.device)
            _lib.check(lib.seunet_metric_sums(p.data_ptr(), _lib.ptr(lab), _lib.ptr(sk), _lib.ptr(pa), p.numel(), nbins, out.data_ptr(), nb,
                                              _lib.stream_ptr()), "metric_sums")
        raw = out.cpu().numpy()
        sums = raw[:64].view(np.uint64)
        self.n = p.numel()
        self.inter, self.pred, self.label, self.pred_skel, self.skel = (np.uint64(v) for v in sums[:5])
        self.branch_label = raw[64:64 + 4 * nbins].view(np.uint32).astype(np.int64)
        self.branch_pred = raw[64 + 4 * nbins:64 + 8 * nbins].view(np.uint32).astype(np.int64)
        extra = raw[64 + 8 * nbins:64 + 8 * nbins + 8].view(np.int32)
        self.max_id = int(extra[0])
        if int(extra[1]):
            raise ValueError(f"metrics: a branch id >= nbins ({nbins}) was met; pass a larger nbins")


def branch_detected_calculation(pred, label_parsing, label_skeleton, thresh=0.8, sums: "MetricSums" = None):
    """metrics.py:14-29 -> (total_branch_num, detected_branch_num, detected_branch_ratio), from the per-branch skeleton voxel
    counts the device histogram returns: a branch counts as detected when the prediction covers at least ``thresh`` of its
    skeleton voxels.  The reference divides two bincounts (the predicted one zero-padded to the labelled one's length) and
    compares the ratio with ``thresh``; a branch id without skeleton voxels gives 0/0 = NaN there, which never passes."""
    s = sums or MetricSums(pred, None, label_skeleton, label_parsing)
    n_branches = int(s.max_id)                       # ids 1 .. max_id (np.bincount's length follows the largest id present)
    in_label = np.asarray(s.branch_label[1:n_branches + 1], dtype=np.float64)
    covered = np.zeros(n_branches, dtype=np.float64)
    have = np.asarray(s.branch_pred[1:n_branches + 1], dtype=np.float64)
    covered[:have.shape[0]] = have
    frac = np.divide(covered, in_label, out=np.full(n_branches, -1.0), where=in_label > 0)    # (0/0: a miss, like NaN >= thresh)
    hit = frac >= thresh
    detected = int(np.count_nonzero(hit))
    return n_branches, detected, round(detected * 100 / n_branches, 2)


def dice_coefficient_score_calculation(pred, label, smooth=1e-5, sums: "MetricSums" = None):
    """metrics.py:32-37."""
    s = sums or MetricSums(pred, label)
    return round(((2.0 * s.inter + smooth) / (s.pred + s.label + smooth)) * 100, 2)


def tree_length_calculation(pred, label_skeleton, smooth=1e-5, sums: "MetricSums" = None):
    """metrics.py:40-44."""
    s = sums or MetricSums(pred, None, label_skeleton)
    return round((s.pred_skel + smooth) / (s.skel + smooth) * 100, 2)


def false_positive_rate_calculation(pred, label, smooth=1e-5, sums: "MetricSums" = None):
    """metrics.py:47-52."""
    s = sums or MetricSums(pred, label)
    fp = np.uint64(s.pred - s.inter) + smooth
    return round(fp * 100 / (np.float64(s.n - int(s.label)) + smooth), 3)


def false_negative_rate_calculation(pred, label, smooth=1e-5, sums: "MetricSums" = None):
    """metrics.py:55-60."""
    s = sums or MetricSums(pred, label)
    fn = np.uint64(s.label - s.inter) + smooth
    return round(fn * 100 / (s.label + smooth), 3)


def sensitivity_calculation(pred, label, sums: "MetricSums" = None):
    """metrics.py:63-65."""
    return round(100 - false_negative_rate_calculation(pred, label, sums=sums), 3)


def specificity_calculation(pred, label, sums: "MetricSums" = None):
    """metrics.py:68-70."""
    return round(100 - false_positive_rate_calculation(pred, label, sums=sums), 3)


def precision_calculation(pred, label, smooth=1e-5, sums: "MetricSums" = None):
    """metrics.py:73-78."""
    s = sums or MetricSums(pred, label)
    tp = s.inter + smooth
    return round(tp * 100 / (s.pred + smooth), 3)


def evaluation_case(pred, label, skeleton, parsing, nbins: int = 4096):
    """The arithmetic of ``evaluation_case`` (train.py:740-775, minus its file reads): largest 26-connected component of
    ``pred``, then (TD, BD, DSC, Pre, Sen, Spe) against the mask, the skeleton (``skeleton > 0``) and the branch parsing.
    One labelling pass + one reduction pass on the device; the percentages are rounded on the host like metrics.py."""
    large_cd = largest_component(pred if not isinstance(pred, np.ndarray) else torch.from_numpy(np.ascontiguousarray(pred != 0)).cuda())
    s = MetricSums(large_cd, label, skeleton, parsing, nbins)
    _, _, bd = branch_detected_calculation(None, None, None, sums=s)
    dsc = dice_coefficient_score_calculation(None, None, sums=s)
    td = tree_length_calculation(None, None, sums=s)
    sen = sensitivity_calculation(None, None, sums=s)
    spe = specificity_calculation(None, None, sums=s)
    pre = precision_calculation(None, None, sums=s)
    return td, bd, dsc, pre, sen, spe
