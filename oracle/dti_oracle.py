"""Loader of the C oracle for double_threshold_iteration (oracle/dti_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _load():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libdti_oracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", HERE], check=True)
        _LIB = C.CDLL(path)
        _LIB.dti_oracle.restype = C.c_int
        _LIB.dti_oracle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
    return _LIB


def double_threshold_iteration(pred: np.ndarray, h_thresh: float, l_thresh: float, pred_dtype: str = "float64") -> np.ndarray:
    """prediction.py:13-37 (pred_dtype "float64") or train.py:25-49 / test.py:18-42 (pred_dtype "float32": pred*255 is
    rounded to float32 first) on a (h, w, z) array; returns float64 zeros/ones like the reference."""
    p = np.ascontiguousarray(pred, dtype=np.float64)
    out = np.empty(p.shape, dtype=np.uint8)
    f32 = {"float64": 0, "float32": 1}[pred_dtype]
    if _load().dti_oracle(p.ctypes.data, p.shape[0], p.shape[1], p.shape[2], float(h_thresh), float(l_thresh), f32, out.ctypes.data):
        raise MemoryError("dti_oracle")
    return out.astype(np.float64)
