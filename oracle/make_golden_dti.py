"""Generate tests/golden/dti_known.npz by running the REFERENCE's own double_threshold_iteration (build container only).

prediction.py cannot be imported whole (pyvista, skimage, stl, ... are not installed: SURVEY 8(c)); the function is pure
numpy, so it is ast-extracted from the source text and executed here on seeded volumes.  Only data is written.
Usage (from the repo root):  python oracle/make_golden_dti.py
"""
import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "dti_known.npz")


def reference_function():
    src = open("/root/reference/prediction.py").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "double_threshold_iteration")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "prediction.py", "exec"), ns)
    return ns["double_threshold_iteration"]


def volume(rng, shape, kind):
    if kind == "noise":          # independent probabilities: many isolated weak voxels
        return rng.random(shape)
    if kind == "blobs":          # smooth field: strong cores with weak halos that the sweep grows in raster order
        f = rng.random(shape)
        for ax in range(3):
            f = (f + np.roll(f, 1, ax) + np.roll(f, -1, ax)) / 3.0
        f = (f - f.min()) / (f.max() - f.min())
        return f
    if kind == "chain":          # a long weak chain fed from one strong voxel at its END: only the order decides
        f = np.zeros(shape)
        f[1, 1, :] = 0.45
        f[1, 1, -1] = 0.9
        f[2, :, 3] = 0.45
        f[2, 0, 3] = 0.9
        return f
    raise ValueError(kind)


def main():
    dti = reference_function()
    rng = np.random.default_rng(20240502)
    cases = [((6, 5, 70), "noise", 0.5, 0.4), ((9, 8, 130), "blobs", 0.5, 0.4), ((4, 4, 66), "chain", 0.5, 0.4),
             ((1, 1, 1), "noise", 0.5, 0.4), ((3, 1, 64), "blobs", 0.6, 0.3), ((2, 7, 65), "noise", 0.7, 0.2)]
    data = {"n": len(cases)}
    for c, (shape, kind, h, l) in enumerate(cases):
        v = volume(rng, shape, kind)
        data[f"pred_{c}"] = v
        data[f"h_{c}"], data[f"l_{c}"] = h, l
        data[f"out_{c}"] = dti(v.copy(), h_thresh=h, l_thresh=l).astype(np.uint8)
        print(shape, kind, "strong", int((v >= h).sum()), "result", int(data[f"out_{c}"].sum()))
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
