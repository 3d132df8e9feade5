"""CPU oracle for the largest-component / hole-filling / metrics step.  TEST INFRASTRUCTURE ONLY.

* metrics (metrics.py:14-78): numpy restatement, PINNED by tests/golden/metrics_known.npz, which
  oracle/make_golden_components.py produces by importing the reference's own metrics.py (it needs numpy only).
* maximum_3d (util.py:58-75) and the component rule of evaluation_case (train.py:749-757): the reference calls
  cc3d.connected_components (connected-components-3d, requirements.txt:3 pins 3.19.0) and skimage.measure.regionprops
  (scikit-image 0.24.0); neither is installed here and util.py / train.py cannot be imported, so these two
  functions are restated on scipy.ndimage.label with the full 3x3x3 structure (= 26-connectivity; same partition into
  components as cc3d, any correct labelling gives it) and scipy.ndimage.binary_fill_holes (the very function util.py:73
  calls; scipy IS installed).  What the partition does not fix is the ORDER of equal-sized components: the reference
  picks `sorted(num_list, key=area)[::-1][0]`, i.e. the highest cc3d label among the largest; cc3d numbers components by
  first appearance in memory order, which is also how scipy.ndimage.label numbers them, so label order = raster order of
  each component's first voxel.  That tie rule rests on cc3d's documented numbering, not on a run of cc3d:
  PARITY UNPINNED for ties between equal-sized components; everything else is pinned by construction (known-answer
  cases in tests/test_oracle_golden.py) and by scipy's own binary_fill_holes.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

S26 = np.ones((3, 3, 3), dtype=bool)


def _sorted_labels(region01):
    """label image + label numbers sorted like util.py:59-64: ascending area (stable), then reversed."""
    label, num = ndimage.label(region01 != 0, structure=S26)
    area = np.bincount(label.ravel(), minlength=num + 1)          # regionprops(...).area = voxel count
    num_list = [i for i in range(1, num + 1)]
    return label, sorted(num_list, key=lambda x: area[x])[::-1]


def largest_component(pred: np.ndarray) -> np.ndarray:
    """train.py:749-757: `large_cd`; an empty prediction stays as it is."""
    label, order = _sorted_labels(pred)
    if order != []:
        return (label == order[0]).astype(np.uint8)
    return (pred != 0).astype(np.uint8)


def maximum_3d(region01: np.ndarray) -> np.ndarray:
    """util.py:58-75, statement by statement (IndexError where the reference raises one)."""
    label, order = _sorted_labels(region01)
    max_region01 = (label == order[0])
    z = region01.shape[2]
    if max_region01[:, :, z // 2].any() == 0 and max_region01[:, :, z // 3].any() == 0 and max_region01[:, :, z // 3 * 2].any() == 0:
        max_region01 = (label == order[1])
    max_region01 = max_region01.astype(np.int8)
    return ndimage.binary_fill_holes(max_region01)


# ---- metrics.py:14-78 ------------------------------------------------------------------------------------------------
def branch_detected_calculation(pred, label_parsing, label_skeleton, thresh=0.8):
    label_branch = label_skeleton * label_parsing
    label_branch_bincount = np.bincount(label_branch.flatten())[1:]
    total_branch_num = label_branch_bincount.shape[0]
    pred_branch = label_branch * pred
    pred_branch_bincount = np.bincount(pred_branch.flatten().astype(np.int32))[1:]
    if total_branch_num != pred_branch_bincount.shape[0]:
        lack_num = total_branch_num - pred_branch_bincount.shape[0]
        pred_branch_bincount = np.concatenate((pred_branch_bincount, np.zeros(lack_num)))
    with np.errstate(divide="ignore", invalid="ignore"):
        branch_ratio_array = pred_branch_bincount / label_branch_bincount
    branch_ratio_array = np.where(branch_ratio_array >= thresh, 1, 0)
    detected_branch_num = np.count_nonzero(branch_ratio_array)
    return total_branch_num, detected_branch_num, round((detected_branch_num * 100) / total_branch_num, 2)


def dice_coefficient_score_calculation(pred, label, smooth=1e-5):
    pred, label = pred.flatten(), label.flatten()
    intersection = np.sum(pred * label)
    return round(((2.0 * intersection + smooth) / (np.sum(pred) + np.sum(label) + smooth)) * 100, 2)


def tree_length_calculation(pred, label_skeleton, smooth=1e-5):
    pred, label_skeleton = pred.flatten(), label_skeleton.flatten()
    return round((np.sum(pred * label_skeleton) + smooth) / (np.sum(label_skeleton) + smooth) * 100, 2)


def false_positive_rate_calculation(pred, label, smooth=1e-5):
    pred, label = pred.flatten(), label.flatten()
    fp = np.sum(pred - pred * label) + smooth
    return round(fp * 100 / (np.sum((1.0 - label)) + smooth), 3)


def false_negative_rate_calculation(pred, label, smooth=1e-5):
    pred, label = pred.flatten(), label.flatten()
    fn = np.sum(label - pred * label) + smooth
    return round(fn * 100 / (np.sum(label) + smooth), 3)


def sensitivity_calculation(pred, label):
    return round(100 - false_negative_rate_calculation(pred, label), 3)


def specificity_calculation(pred, label):
    return round(100 - false_positive_rate_calculation(pred, label), 3)


def precision_calculation(pred, label, smooth=1e-5):
    pred, label = pred.flatten(), label.flatten()
    tp = np.sum(pred * label) + smooth
    return round(tp * 100 / (np.sum(pred) + smooth), 3)


def evaluation_case(pred, label, skeleton, parsing):
    """train.py:740-775 without the file reads: (TD, BD, DSC, Pre, Sen, Spe)."""
    large_cd = largest_component(pred)
    skeleton = (skeleton > 0).astype("uint8")
    _, _, bd = branch_detected_calculation(large_cd, parsing, skeleton)
    return (tree_length_calculation(large_cd, skeleton), bd, dice_coefficient_score_calculation(large_cd, label),
            precision_calculation(large_cd, label), sensitivity_calculation(large_cd, label), specificity_calculation(large_cd, label))


def synthetic_tree(shape, seed):
    """A branching tube structure + noise blobs + an enclosed cavity: prediction, mask, skeleton, branch parsing."""
    rng = np.random.default_rng(seed)
    h, w, z = shape
    label = np.zeros(shape, dtype=np.uint8)
    skel = np.zeros(shape, dtype=np.uint8)
    parsing = np.zeros(shape, dtype=np.int32)
    bid = 0
    for _ in range(6):
        p = np.array([rng.integers(2, h - 2), rng.integers(2, w - 2), rng.integers(2, z - 2)], dtype=float)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        bid += 1
        for step in range(int(rng.integers(10, 40))):
            q = np.clip(np.round(p).astype(int), 1, np.array(shape) - 2)
            label[q[0] - 1:q[0] + 2, q[1] - 1:q[1] + 2, q[2] - 1:q[2] + 2] = 1
            skel[tuple(q)] = 1
            parsing[q[0] - 1:q[0] + 2, q[1] - 1:q[1] + 2, q[2] - 1:q[2] + 2] = bid
            p += d
            if step % 12 == 11:
                bid += 1
                d = d + 0.6 * rng.normal(size=3); d /= np.linalg.norm(d)
    pred = label.copy()
    pred[rng.random(shape) < 0.02] = 1                   # false-positive specks (separate components)
    pred[rng.random(shape) < 0.15] = 0                   # misses (break the tree into pieces, open cavities)
    return pred, label, skel, parsing
