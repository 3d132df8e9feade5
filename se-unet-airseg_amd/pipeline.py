"""GPU input pipeline: one resident CT case -> the tensors of a training step, on the device.

Mirrors what the reference's Datasets do between the file reads and the step (data.py): crop extraction
(``CropSegData.crop`` :645-664, the ``*_sample`` helpers :85-252), the two HU windows (``process_imgmsk`` :667-677,
``process_img`` :286-299, ``two_channel`` prediction.py:39-49), ``weight ** (U + 2) * label + (1 - label)`` (:701) and the
flip / rotate augmentation (``random_flip`` / ``random_rotate`` :40-67).  All random draws stay on the host, in the
reference's generators (python ``random`` and ``numpy.random``) and in its order, so a seeded run selects the same crops and
augmentations; the data movement and arithmetic are one launch of ``seunet_crop_batch`` that writes the layout
train.py:582-592 builds (``data (B, 2, n, n, n)``, ``label / weight (B, 1, n, n, n)``, float32), instead of four H2D copies
of host-built float tensors per step.  There is no CPU path."""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

_IMG = {torch.int16: 0, torch.float32: 1}
_WGT = {torch.float16: 0, torch.float32: 1, torch.float64: 2}


def aug_code(flipid: Optional[Sequence[int]] = None, rotate: Optional[str] = None) -> int:
    """``random_rotate(random_flip(x))`` (data.py:40-67) as one signed axis map.  flipid in {-1, +1}^3 (or None);
    rotate in {None, "left", "right"}: rotate_left is out[a,b,c] = y[a, c, n-1-b] (:50-53), rotate_right is
    out[a,b,c] = y[a, n-1-c, n-1-b] (:54-58).  Code: bit k = source axis k read reversed, bit 3 = axes 1, 2 swapped."""
    r = [False, False, False] if flipid is None else [int(f) == -1 for f in flipid]
    if rotate == "left":
        r = [r[0], r[1], not r[2]]
    elif rotate == "right":
        r = [r[0], not r[1], not r[2]]
    elif rotate is not None:
        raise ValueError(f"rotate {rotate!r}: None, 'left' or 'right'")
    return int(r[0]) | int(r[1]) << 1 | int(r[2]) << 2 | int(rotate is not None) << 3


def draw_augmentation() -> int:
    """One ``augment`` call (data.py:679-686) on the reference's generators: ``random.random() > 0.5`` -> random_flip
    (three ``np.random.randint(2)``, redrawn while no axis flips, :42-44); ``random.random() > 0.5`` -> random_rotate
    (``random.random() > 0.5`` -> left, else right, :61-66)."""
    flipid, rot = None, None
    if random.random() > 0.5:
        while True:                                   # three numpy draws per attempt (axis order z, y, x), until some axis flips
            bits = [int(np.random.randint(2)) for _ in range(3)]
            if any(b == 0 for b in bits):             # bit 0 -> step -1 (that axis is reversed), bit 1 -> step +1
                break
        flipid = np.array([2 * b - 1 for b in bits])
    if random.random() > 0.5:
        rot = "left" if random.random() > 0.5 else "right"
    return aug_code(flipid, rot)


def draw_stage1_plan(shape: Sequence[int], batch_size: int, cube: int = 128, aug_flag: int = 1) -> Dict:
    """The random choices of one ``CropSegData.__getitem__`` (data.py:689-715) in its draw order: the weight exponent
    ``np.random.random()`` (:701), then per crop the centre ``random.randint`` z, y, x with INCLUSIVE bounds
    [cube/2, dim - cube/2] (:648-656), then per crop one ``augment``."""
    u = float(np.random.random())
    half = cube // 2
    centres = [(random.randint(half, shape[0] - half), random.randint(half, shape[1] - half), random.randint(half, shape[2] - half))
               for _ in range(batch_size)]
    starts = [(z - half, y - half, x - half) for z, y, x in centres]
    codes = [draw_augmentation() if aug_flag == 1 else 0 for _ in range(batch_size)]
    return {"u": u, "starts": starts, "codes": codes}


def _loc_len(loc) -> int:
    return int(len(loc[0]))


def _start_near(loc, shape: Sequence[int], cube: int) -> Tuple[int, int, int]:
    """The crop origin of the reference's location-guided samplers (``skeleton_sample`` / ``small_airway_sample`` /
    ``hard_sample`` / ``*_sample_wg``, data.py:85-252): one ``np.random.randint(len(loc[0]))`` picks a voxel of the candidate
    list, then per axis one ``np.random.randint(max(0, v - cube // 2), v + cube // 2)``; an origin whose cube would leave the
    volume is moved back to ``dim - cube``."""
    k = int(np.random.randint(_loc_len(loc)))
    half = cube // 2
    start = []
    for ax in range(3):
        v = int(loc[ax][k])
        start.append(int(np.random.randint(max(0, v - half), v + half)))
    return tuple(min(st, int(shape[ax]) - cube) if st + cube > int(shape[ax]) else st for ax, st in enumerate(start))


def _start_uniform(shape: Sequence[int], cube: int) -> Tuple[int, int, int]:
    """``random_sample`` (data.py:159-172): three ``np.random.randint(0, dim - cube)`` (exclusive upper bound)."""
    return tuple(int(np.random.randint(0, int(shape[ax]) - cube)) for ax in range(3))


def draw_stage2_plan(shape: Sequence[int], batch_size: int, loc_skeleton, loc_small, cube: int = 128, hard_ratio: float = 0.4,
                     aug_flag: int = 1) -> Dict:
    """The random choices of one ``AirwayHMData.__getitem__`` (data.py:301-324, 359-408) in its draw order: the weight exponent
    ``np.random.random()`` (:388); then per crop (``crop``, :301-324) ``np.random.random() < hard_ratio`` -> ``hard_sample``
    (:120-157: ``np.random.random() > 0.5`` and a non-empty missed-skeleton list -> that list, else the small-airway list if
    non-empty, else a uniform origin), otherwise ``random_sample``; then per crop one ``augment``.  ``loc_skeleton`` /
    ``loc_small``: the (z, y, x) index triples ``np.where(skeleton * (1 - pred))`` / ``np.where(dis * skeleton < 2)`` of the case
    (:305-306; any indexable triple: numpy arrays or tensors).  ``kinds``: 'skeleton' | 'small' | 'random' per crop."""
    u = float(np.random.random())
    starts, kinds = [], []
    for _ in range(batch_size):
        if np.random.random() < hard_ratio:
            pick_skel = np.random.random() > 0.5           # (drawn whether or not the list is empty: `and` evaluates it first)
            if pick_skel and _loc_len(loc_skeleton) > 0:
                starts.append(_start_near(loc_skeleton, shape, cube)); kinds.append("skeleton")
            elif _loc_len(loc_small) > 0:
                starts.append(_start_near(loc_small, shape, cube)); kinds.append("small")
            else:
                starts.append(_start_uniform(shape, cube)); kinds.append("random")
        else:
            starts.append(_start_uniform(shape, cube)); kinds.append("random")
    codes = [draw_augmentation() if aug_flag == 1 else 0 for _ in range(batch_size)]
    return {"u": u, "starts": starts, "codes": codes, "kinds": kinds}


def draw_stage3_plan(shape: Sequence[int], batch_size: int, loc_skeleton, loc_small, loc_break, cube: int = 128,
                     hard_ratio: float = 0.8, break_ratio: float = 0.625, aug_flag: int = 1) -> Dict:
    """The random choices of one ``AirwayHMData3.__getitem__`` (data.py:449-491, 546-584): the weight exponent (:566); per
    crop ``np.random.random() < hard_ratio`` -> (``np.random.random() < break_ratio`` and a non-empty break list -> a break
    sample, else ``np.random.random() < 0.5`` -> small-airway sample, else missed-skeleton sample), otherwise a uniform
    origin; then per crop one ``augment``.  ``loc_break``: the saved ``np.where(br_skel == 1)`` triple (weight_br.py:171)."""
    u = float(np.random.random())
    starts, kinds = [], []
    for _ in range(batch_size):
        if np.random.random() < hard_ratio:
            take_break = np.random.random() < break_ratio
            if take_break and _loc_len(loc_break) != 0:
                starts.append(_start_near(loc_break, shape, cube)); kinds.append("break")
            elif np.random.random() < 0.5:
                starts.append(_start_near(loc_small, shape, cube)); kinds.append("small")
            else:
                starts.append(_start_near(loc_skeleton, shape, cube)); kinds.append("skeleton")
        else:
            starts.append(_start_uniform(shape, cube)); kinds.append("random")
    codes = [draw_augmentation() if aug_flag == 1 else 0 for _ in range(batch_size)]
    return {"u": u, "starts": starts, "codes": codes, "kinds": kinds}


def _dev(t, name, dtypes):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"seunet pipeline: `{name}` must be a CUDA tensor resident on the GPU (there is no CPU path)")
    if t.dtype not in dtypes:
        raise TypeError(f"seunet pipeline: `{name}` has dtype {t.dtype}; supported: {[str(d) for d in dtypes]}")
    return t.contiguous()


def crop_batch(img: torch.Tensor, starts: Sequence[Tuple[int, int, int]], cube: int = 128, label: Optional[torch.Tensor] = None,
               weight: Optional[torch.Tensor] = None, skeleton: Optional[torch.Tensor] = None,
               codes: Optional[Sequence[int]] = None, u: Optional[float] = None, f64_math: Optional[bool] = None
               ) -> Dict[str, torch.Tensor]:
    """img: (D, H, W) HU volume (file value - 1024, data.py:692), int16 or float32, on the GPU.  label / skeleton: uint8
    volumes; weight: the LIB map (float16 as stored, or float32 / float64).  starts: (z, y, x) origin per crop; codes:
    ``aug_code`` per crop; u: the ``np.random.random()`` draw of data.py:701 (exponent u + 2).
    f64_math defaults to the reference's arithmetic for the volume's dtype: int16 crops divide in float64
    (``process_img``, data.py:286-299), a float32 volume divides in float32 (``process_imgmsk``, data.py:667-677).
    Returns {"data": (B,2,n,n,n), "label": (B,1,n,n,n), "weight": ..., "skel": ...} float32 CUDA tensors."""
    img = _dev(img, "img", _IMG)
    label, skeleton = _dev(label, "label", {torch.uint8: 0}), _dev(skeleton, "skeleton", {torch.uint8: 0})
    weight = _dev(weight, "weight", _WGT)
    if img.dim() != 3:
        raise ValueError(f"img must be (D, H, W), got {tuple(img.shape)}")
    for t, nm in ((label, "label"), (weight, "weight"), (skeleton, "skeleton")):
        if t is not None and t.shape != img.shape:
            raise ValueError(f"{nm} shape {tuple(t.shape)} differs from the image's {tuple(img.shape)}")
    if weight is not None and (label is None or u is None):
        raise ValueError("the weight map needs the label volume and the draw `u` (data.py:701)")
    if f64_math is None:
        f64_math = img.dtype == torch.int16
    b = len(starts)
    codes = [0] * b if codes is None else list(codes)
    lib = _lib.load()
    D, H, W = (int(v) for v in img.shape)
    dev = img.device
    with torch.cuda.device(dev):
        out = {"data": torch.empty((b, 2, cube, cube, cube), dtype=torch.float32, device=dev)}
        if label is not None:
            out["label"] = torch.empty((b, 1, cube, cube, cube), dtype=torch.float32, device=dev)
        if weight is not None:
            out["weight"] = torch.empty((b, 1, cube, cube, cube), dtype=torch.float32, device=dev)
        if skeleton is not None:
            out["skel"] = torch.empty((b, 1, cube, cube, cube), dtype=torch.float32, device=dev)
        for i in range(0, b, 32):                                   # 32 crops per native call (kernel-argument table)
            n = min(32, b - i)
            st = _lib.int_array([int(v) for s in starts[i:i + n] for v in s])
            cd = _lib.int_array([int(c) for c in codes[i:i + n]])
            _lib.check(lib.seunet_crop_batch(
                img.data_ptr(), _IMG[img.dtype], _lib.ptr(label), _lib.ptr(weight), _WGT[weight.dtype] if weight is not None else 0,
                _lib.ptr(skeleton), D, H, W, cube, n, st, cd, float(u) + 2.0 if u is not None else 2.0, int(bool(f64_math)),
                out["data"][i:].data_ptr(), out["label"][i:].data_ptr() if label is not None else None,
                out["weight"][i:].data_ptr() if weight is not None else None,
                out["skel"][i:].data_ptr() if skeleton is not None else None, _lib.stream_ptr()), "crop_batch")
    return out


def two_channel_volume(img: torch.Tensor, f64_math: Optional[bool] = None) -> torch.Tensor:
    """(X, Y, Z) HU volume on the GPU -> the (1, 2, X, Y, Z) float32 network input of the whole-volume loops:
    prediction.py:39-49,71-75 (float64 math, then ``astype(np.float32)``: ``f64_math=True``, the default for int16) or
    ``SegValCropData.process_imgmsk`` data.py:775-784,796 (float32 math, the default for a float32 volume)."""
    img = _dev(img, "img", _IMG)
    if f64_math is None:
        f64_math = img.dtype == torch.int16
    with torch.cuda.device(img.device):
        out = torch.empty((1, 2) + tuple(img.shape), dtype=torch.float32, device=img.device)
        _lib.check(_lib.load().seunet_hu_two_channel(img.data_ptr(), _IMG[img.dtype], img.numel(), int(bool(f64_math)), out.data_ptr(),
                                                     _lib.stream_ptr()), "hu_two_channel")
    return out


class CropSegDataGPU:
    """Stage-1 sampler with the interface of the reference's ``CropSegData`` minus the file IO (data.py:632-715): the
    case volumes are handed over once, resident on the GPU; ``sample()`` = one ``__getitem__`` + the step's
    ``.float().cuda()`` / transpose / cat (train.py:582-592)."""

    def __init__(self, img: torch.Tensor, label: torch.Tensor, weight: torch.Tensor, batch_size: int, aug_flag: int = 1,
                 cube: int = 128):
        # process_imgmsk casts the volume to float32 first (data.py:668) -> float32 division
        self.img = img.to(torch.float32) if img.dtype != torch.float32 else img
        self.label, self.weight = label, weight
        self.batch_size, self.aug_flag, self.cube = batch_size, aug_flag, cube

    def sample(self) -> Dict[str, torch.Tensor]:
        plan = draw_stage1_plan(self.img.shape, self.batch_size, self.cube, self.aug_flag)
        return crop_batch(self.img, plan["starts"], self.cube, self.label, self.weight, None, plan["codes"], plan["u"], f64_math=False)


class AirwayHMDataGPU:
    """Stage-2 sampler with the interface of the reference's ``AirwayHMData`` minus the file IO (data.py:254-408): the case
    volumes are resident on the GPU (HU int16 = file value - 1024, uint8 label, float16 LIB weight), the two candidate lists
    of the hard-mining samplers (``loc_skeleton`` = where(skeleton * (1 - pred)), ``loc_small`` = where(EDT(label) * skeleton
    < 2), data.py:305-306) are computed once per case by the caller; ``sample()`` = one ``__getitem__`` + the step's
    ``.float().cuda()`` / transpose / cat (train.py:416-426).  ``hard_ratio`` is a plain attribute with the reference's initial
    value (:273-283); the curriculum policy that moves it between epochs (``update_scheduler``, :326-349) is training control
    plane, outside this path: the caller's loop sets ``ds.hard_ratio``."""

    def __init__(self, img, label, weight, loc_skeleton, loc_small, batch_size: int, aug_flag: int = 1, cube: int = 128):
        self.img, self.label, self.weight = img, label, weight
        self.loc_skeleton, self.loc_small = loc_skeleton, loc_small
        self.batch_size, self.aug_flag, self.cube = batch_size, aug_flag, cube
        self.random_ratio, self.hard_ratio = 0.6, 0.4

    def sample(self) -> Dict[str, torch.Tensor]:
        plan = draw_stage2_plan(self.img.shape, self.batch_size, self.loc_skeleton, self.loc_small, self.cube, self.hard_ratio,
                                self.aug_flag)
        out = crop_batch(self.img, plan["starts"], self.cube, self.label, self.weight, None, plan["codes"], plan["u"])
        out["kinds"] = plan["kinds"]
        return out


class AirwayHMData3GPU:
    """Stage-3 sampler (reference ``AirwayHMData3``, data.py:410-584): as stage 2 plus the break-point list ``loc_break`` and
    the skeleton crop; the weight volume handed over is already ``LIB + 0.6 * break weight`` in float16 (:553-557)."""

    def __init__(self, img, label, weight, skeleton, loc_skeleton, loc_small, loc_break, batch_size: int, aug_flag: int = 1,
                 cube: int = 128):
        self.img, self.label, self.weight, self.skeleton = img, label, weight, skeleton
        self.loc_skeleton, self.loc_small, self.loc_break = loc_skeleton, loc_small, loc_break
        self.batch_size, self.aug_flag, self.cube = batch_size, aug_flag, cube
        self.hard_ratio, self.break_ratio = 0.8, 0.625      # (set by the caller's curriculum between epochs)

    def sample(self) -> Dict[str, torch.Tensor]:
        plan = draw_stage3_plan(self.img.shape, self.batch_size, self.loc_skeleton, self.loc_small, self.loc_break, self.cube,
                                self.hard_ratio, self.break_ratio, self.aug_flag)
        out = crop_batch(self.img, plan["starts"], self.cube, self.label, self.weight, self.skeleton, plan["codes"], plan["u"])
        out["kinds"] = plan["kinds"]
        return out
