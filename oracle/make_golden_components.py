"""Generate tests/golden/metrics_known.npz by running the REFERENCE's own metrics.py (build container only; the module
needs numpy only, so it is imported as it stands).  Only data is written.  Usage: python oracle/make_golden_components.py"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "metrics_known.npz")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import components_oracle as co   # noqa: E402  (only for the synthetic case generator)


def main():
    spec = importlib.util.spec_from_file_location("ref_metrics", "/root/reference/metrics.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    data = {"n": 4}
    for c, (shape, seed) in enumerate([((40, 44, 48), 1), ((33, 31, 65), 2), ((24, 24, 24), 3), ((50, 20, 70), 4)]):
        pred, label, skel, parsing = co.synthetic_tree(shape, seed)
        if c == 2:
            pred = pred & label            # no false positives: FPR = smooth / ..., precision ~ 100
        with np.errstate(divide="ignore", invalid="ignore"):
            tot, det, bd = m.branch_detected_calculation(pred, parsing, skel)
        vals = [bd, m.dice_coefficient_score_calculation(pred, label), m.tree_length_calculation(pred, skel),
                m.false_positive_rate_calculation(pred, label), m.false_negative_rate_calculation(pred, label),
                m.sensitivity_calculation(pred, label), m.specificity_calculation(pred, label), m.precision_calculation(pred, label)]
        data[f"pred_{c}"], data[f"label_{c}"], data[f"skel_{c}"], data[f"parsing_{c}"] = pred, label, skel, parsing.astype(np.int16)
        data[f"branches_{c}"] = np.array([tot, det])
        data[f"values_{c}"] = np.array(vals, dtype=np.float64)
        print(shape, "branches", tot, det, "BD DSC TD FPR FNR Sen Spe Pre =", vals)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
