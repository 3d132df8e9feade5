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


def reference_function(script="prediction.py"):
    src = open("/root/reference/" + script).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "double_threshold_iteration")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), script, "exec"), ns)
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
    if kind == "ulp":            # values within a float32 ulp of the thresholds: the float64 copy (prediction.py:19) and the
        f = rng.random(shape) * 0.38          # float32 copies (train.py:31, test.py:24) classify these differently
        flat = f.reshape(-1)
        idx = rng.permutation(flat.size)[: flat.size // 3]
        near = np.array([0.5 - 1e-9, 0.5 - 3e-8, 0.5 + 1e-9, 0.4 - 1e-9, 0.4 - 2e-8, 0.4 + 1e-9, 0.5, 0.4, 0.45, 0.9])
        flat[idx] = near[rng.integers(0, near.size, idx.size)]
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

    # the float32 copies used by validation / test (train.py:25-49, test.py:18-42): same cases + near-threshold volumes
    dti32, dti32b = reference_function("train.py"), reference_function("test.py")
    cases32 = cases + [((5, 6, 67), "ulp", 0.5, 0.4), ((3, 9, 130), "ulp", 0.5, 0.4)]
    rng = np.random.default_rng(20240503)
    data, differ = {"n": len(cases32)}, 0
    for c, (shape, kind, h, l) in enumerate(cases32):
        v = volume(rng, shape, kind)
        data[f"pred_{c}"] = v
        data[f"h_{c}"], data[f"l_{c}"] = h, l
        a = dti32(v.copy(), h_thresh=h, l_thresh=l)
        assert np.array_equal(a, dti32b(v.copy(), h_thresh=h, l_thresh=l)), "train.py and test.py copies disagree"
        data[f"out_{c}"] = a.astype(np.uint8)
        d64 = dti(v.copy(), h_thresh=h, l_thresh=l).astype(np.uint8)
        data[f"out64_{c}"] = d64
        differ += int((d64 != data[f"out_{c}"]).sum())
        print(shape, kind, "float32 copy result", int(a.sum()), "differs from the float64 copy in", int((d64 != data[f"out_{c}"]).sum()), "voxels")
    assert differ > 0, "the fixture must contain voxels on which the two variants disagree"
    out32 = OUT.replace("dti_known", "dti_known_f32")
    np.savez_compressed(out32, **data)
    print("wrote", out32, os.path.getsize(out32), "bytes")


if __name__ == "__main__":
    main()
