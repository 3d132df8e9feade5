"""CPU oracle for the input pipeline.  TEST INFRASTRUCTURE ONLY.

numpy restatement of what the reference's Datasets do to one loaded case between the file reads and the tensors of the
training step (data.py); each function cites the lines it follows.  Pinned by tests/golden/pipeline_known.npz, which
oracle/make_golden_pipeline.py produces by running the reference's OWN helpers (ast-extracted from data.py:
random_flip, random_rotate, CropSegData.crop / process_imgmsk / augment, AirwayHMData.process_img, the weight statement of
data.py:701) under seeded generators.  Only tests/ import this file.
"""
from __future__ import annotations

import random
from copy import deepcopy
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


# ---- HU windows -------------------------------------------------------------------------------------------------
def process_imgmsk(data: np.ndarray, mask: Optional[np.ndarray] = None):
    """CropSegData.process_imgmsk (data.py:667-677) / SegValCropData.process_imgmsk (data.py:775-784): float32 math."""
    data = data.astype(np.float32)
    data2 = data.copy()
    data2[data2 > 500] = 500
    data2[data2 < -1000] = -1000
    data2 = (data2 + 1000) / 1500
    data[data > 1024] = 1024
    data[data < -1024] = -1024
    data = (data + 1024) / 2048
    if mask is None:
        return data, data2
    return data, data2, (mask > 0).astype(np.int32).astype(np.float32)


def process_img_crop(crop: np.ndarray):
    """AirwayHMData.process_img on one crop (data.py:286-299; AirwayHMData3: 433-446): the crop keeps the volume's dtype
    (int16 from SimpleITK), so the divisions are numpy true divisions -> float64."""
    crop = crop.copy()
    crop2 = crop.copy()
    crop2[crop2 > 500] = 500
    crop2[crop2 < -1000] = -1000
    crop2 = (crop2 + 1000) / 1500
    crop[crop > 1024] = 1024
    crop[crop < -1024] = -1024
    crop = (crop + 1024) / 2048
    return crop, crop2


def two_channel(data: np.ndarray):
    """prediction.py:39-49: float64 math (`data.astype(float)`); the caller rounds with astype(np.float32) (:74)."""
    data = data.astype(float)
    data2 = data.copy()
    data2[data2 > 500] = 500
    data2[data2 < -1000] = -1000
    data2 = (data2 + 1000) / 1500
    data[data > 1024] = 1024
    data[data < -1024] = -1024
    data = (data + 1024) / 2048
    return data, data2


# ---- weight map ---------------------------------------------------------------------------------------------------
def lib_weight(weight: np.ndarray, label: np.ndarray, u: float) -> np.ndarray:
    """data.py:701 (also :389, :561): ``weight ** (np.random.random() + 2) * label + (1 - label)`` with the draw ``u``
    passed in.  numpy evaluates the power in the weight array's dtype (float16 for the LIB maps, lib_weight.py:50)."""
    return weight ** (u + 2) * label + (1 - label)


# ---- augmentation as index maps -------------------------------------------------------------------------------------
def flip(arr: np.ndarray, flipid: Sequence[int]) -> np.ndarray:
    """random_flip's body for a given flipid in {-1, +1}^3 (data.py:40-47)."""
    return np.ascontiguousarray(arr[::flipid[0], ::flipid[1], ::flipid[2]])


def rotate_left(data: np.ndarray) -> np.ndarray:
    """data.py:50-53: out[a, b, c] = in[a, c, n-1-b]."""
    return np.ascontiguousarray(data.transpose((0, 2, 1))[:, ::-1])


def rotate_right(data: np.ndarray) -> np.ndarray:
    """data.py:54-58: out[a, b, c] = in[a, n-1-c, n-1-b]."""
    data = np.ascontiguousarray(data[:, ::-1])
    data = data.transpose((0, 2, 1))
    return np.ascontiguousarray(data[:, ::-1])


def aug_code(flipid: Optional[Sequence[int]], rot: Optional[str]) -> int:
    """The signed axis map of `rotate(flip(x))` as the code the HIP kernel takes: bit k (k = 0..2) = source axis k is read
    reversed, bit 3 = source axes 1 and 2 are fed by output axes 2 and 1.
      flip f:          y[a,b,c] = x[f0(a), f1(b), f2(c)]
      rotate_left:     out[a,b,c] = y[a, c, n-1-b]     = x[f0(a), f1(c), f2(n-1-b)]
      rotate_right:    out[a,b,c] = y[a, n-1-c, n-1-b] = x[f0(a), f1(n-1-c), f2(n-1-b)]"""
    r = [False, False, False] if flipid is None else [f == -1 for f in flipid]
    swap = rot is not None
    if rot == "left":
        r = [r[0], r[1], not r[2]]
    elif rot == "right":
        r = [r[0], not r[1], not r[2]]
    return int(r[0]) | int(r[1]) << 1 | int(r[2]) << 2 | int(swap) << 3


def apply_code(arr: np.ndarray, code: int) -> np.ndarray:
    """Inverse view of aug_code, for testing the map itself: out[o] = in[s(o)]."""
    n = arr.shape[0]
    o0, o1, o2 = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    t1, t2 = (o2, o1) if code & 8 else (o1, o2)
    s0 = n - 1 - o0 if code & 1 else o0
    s1 = n - 1 - t1 if code & 2 else t1
    s2 = n - 1 - t2 if code & 4 else t2
    return arr[s0, s1, s2]


# ---- the random plan of one CropSegData.__getitem__ (stage 1), in the reference's draw order -----------------------
def draw_stage1_plan(shape: Sequence[int], batch_size: int, cube: int = 128, aug_flag: int = 1) -> Dict:
    """Order of draws in CropSegData.__getitem__ (data.py:689-715): (1) np.random.random() for the weight exponent (:701);
    (2) crop centres, python `random.randint` z, y, x per crop, INCLUSIVE upper bound (:650-656); (3) per crop,
    `augment` (:679-686): random.random() > 0.5 -> random_flip (np.random.randint(2) x 3, redrawn while all +1, :42-44);
    random.random() > 0.5 -> random_rotate (random.random() > 0.5 -> left else right, :61-66)."""
    u = np.random.random()
    rng = [[cube // 2, shape[i] - cube // 2] for i in range(3)]
    centres = []
    for _ in range(batch_size):
        z = random.randint(rng[0][0], rng[0][1])
        y = random.randint(rng[1][0], rng[1][1])
        x = random.randint(rng[2][0], rng[2][1])
        centres.append((z, y, x))
    starts = [(z - cube // 2, y - cube // 2, x - cube // 2) for z, y, x in centres]
    codes = []
    for _ in range(batch_size):
        flipid, rot = None, None
        if aug_flag == 1:
            if random.random() > 0.5:
                flipid = np.array([np.random.randint(2), np.random.randint(2), np.random.randint(2)]) * 2 - 1
                while (flipid == [1, 1, 1]).all():
                    flipid = np.array([np.random.randint(2), np.random.randint(2), np.random.randint(2)]) * 2 - 1
            if random.random() > 0.5:
                rot = "left" if random.random() > 0.5 else "right"
        codes.append(aug_code(flipid, rot))
    return {"u": float(u), "starts": starts, "codes": codes}


def crop_batch(img: np.ndarray, starts, codes, cube: int, label=None, weight=None, skeleton=None, u: Optional[float] = None,
               f64_math: Optional[bool] = None) -> Dict[str, np.ndarray]:
    """The step's tensors for a planned batch: data (B,2,n,n,n), label / weight / skel (B,1,n,n,n), all float32 -- what
    train.py:582-592 builds from the Dataset's return values (`.float()`, transpose(0,1), cat).
    f64_math (default: integer volume) selects process_img_crop (int crops, data.py:286-299) over process_imgmsk
    (float32 volume, data.py:667-677)."""
    if f64_math is None:
        f64_math = np.issubdtype(img.dtype, np.integer)
    out = {"data": [], "label": [], "weight": [], "skel": []}
    lab01 = None if label is None else (label > 0)
    w_full = None
    if weight is not None:
        w_full = lib_weight(weight, lab01.astype(np.float32) if weight.dtype != np.float16 else lab01.astype(np.uint8), u)
    for (z, y, x), code in zip(starts, codes):
        sl = (slice(z, z + cube), slice(y, y + cube), slice(x, x + cube))
        c = img[sl]
        if f64_math:
            c0, c1 = process_img_crop(c)
        else:
            c0, c1 = process_imgmsk(c)
        out["data"].append(np.stack([apply_code(c0, code), apply_code(c1, code)]).astype(np.float32))
        if label is not None:
            out["label"].append(apply_code(lab01[sl].astype(np.float32), code)[None])
        if w_full is not None:
            out["weight"].append(apply_code(w_full[sl], code).astype(np.float32)[None])
        if skeleton is not None:
            out["skel"].append(apply_code(skeleton[sl], code).astype(np.float32)[None])
    return {k: np.stack(v) for k, v in out.items() if v}


# ---- the random plans of AirwayHMData.__getitem__ (stage 2) and AirwayHMData3.__getitem__ (stage 3) ---------------------
def _guided_start(loc, origin_size, cube):
    """skeleton_sample / small_airway_sample / hard_sample / *_sample_wg (data.py:85-252): one randint picks the voxel, one
    randint per axis places the origin in [max(0, v - cube//2), v + cube//2), origins past the end are clamped."""
    k = np.random.randint(len(loc[0]))
    start = [np.random.randint(max(0, loc[a][k] - cube // 2), loc[a][k] + cube // 2) for a in range(3)]
    for a in range(3):
        if start[a] + cube > origin_size[a]:
            start[a] = origin_size[a] - cube
    return tuple(int(v) for v in start)


def _uniform_start(origin_size, cube):
    """random_sample / random_sample_wg (data.py:159-172, 240-252)."""
    return tuple(int(np.random.randint(0, origin_size[a] - cube)) for a in range(3))


def _augment_codes(batch_size, aug_flag):
    codes = []
    for _ in range(batch_size):
        flipid, rot = None, None
        if aug_flag == 1:
            if random.random() > 0.5:
                flipid = np.array([np.random.randint(2), np.random.randint(2), np.random.randint(2)]) * 2 - 1
                while (flipid == [1, 1, 1]).all():
                    flipid = np.array([np.random.randint(2), np.random.randint(2), np.random.randint(2)]) * 2 - 1
            if random.random() > 0.5:
                rot = "left" if random.random() > 0.5 else "right"
        codes.append(aug_code(flipid, rot))
    return codes


def draw_stage2_plan(shape, batch_size, loc_skeleton, loc_small, cube=128, hard_ratio=0.4, aug_flag=1) -> Dict:
    """AirwayHMData.__getitem__ (data.py:359-408): exponent draw (:388), then ``crop`` (:301-324) with ``hard_sample`` (:120-157)
    / ``random_sample`` (:159-172), then one ``augment`` per crop (:351-357)."""
    u = np.random.random()
    starts = []
    for _ in range(batch_size):
        if np.random.random() < hard_ratio:
            if np.random.random() > 0.5 and len(loc_skeleton[0]) > 0:
                starts.append(_guided_start(loc_skeleton, shape, cube))
            elif len(loc_small[0]) > 0:
                starts.append(_guided_start(loc_small, shape, cube))
            else:
                starts.append(_uniform_start(shape, cube))
        else:
            starts.append(_uniform_start(shape, cube))
    return {"u": float(u), "starts": starts, "codes": _augment_codes(batch_size, aug_flag)}


def draw_stage3_plan(shape, batch_size, loc_skeleton, loc_small, loc_break, cube=128, hard_ratio=0.8, break_ratio=0.625,
                     aug_flag=1) -> Dict:
    """AirwayHMData3.__getitem__ (data.py:546-584) with ``crop`` (:449-491)."""
    u = np.random.random()
    starts = []
    for _ in range(batch_size):
        if np.random.random() < hard_ratio:
            if np.random.random() < break_ratio and len(loc_break[0]) != 0:
                starts.append(_guided_start(loc_break, shape, cube))
            elif np.random.random() < 0.5:
                starts.append(_guided_start(loc_small, shape, cube))
            else:
                starts.append(_guided_start(loc_skeleton, shape, cube))
        else:
            starts.append(_uniform_start(shape, cube))
    return {"u": float(u), "starts": starts, "codes": _augment_codes(batch_size, aug_flag)}
