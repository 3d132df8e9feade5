"""Generate tests/golden/pipeline_known.npz by running the REFERENCE's own data.py helpers (build container only).

data.py cannot be imported whole (SimpleITK, nibabel are not installed: SURVEY 8(c)); its augmentation / normalisation /
cropping helpers are pure numpy, so they are ast-extracted from the source text and executed here under seeded generators
on a synthetic case.  Only data (inputs, draws, outputs) is written.   Usage: python oracle/make_golden_pipeline.py
"""
import ast
import os
import random
import sys
from copy import deepcopy

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "pipeline_known.npz")
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def extract():
    src = open("/root/reference/data.py").read()
    tree = ast.parse(src)
    ns = {"np": np, "random": random, "deepcopy": deepcopy}
    funcs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("random_flip", "random_rotate")]
    exec(compile(ast.Module(body=funcs, type_ignores=[]), "data.py", "exec"), ns)
    keep = {"CropSegData": ("crop", "process_imgmsk", "augment"), "AirwayHMData": ("process_img",)}
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in keep]:
        cls.bases = []
        cls.body = [m for m in cls.body if isinstance(m, ast.FunctionDef) and m.name in keep[cls.name]]
        exec(compile(ast.Module(body=[cls], type_ignores=[]), "data.py", "exec"), ns)
    # the weight statement of CropSegData.__getitem__ (data.py:701)
    getitem = next(m for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "CropSegData"
                   for m in n.body if isinstance(m, ast.FunctionDef) and m.name == "__getitem__")
    stmt = next(s for s in getitem.body if isinstance(s, ast.Assign) and getattr(s.targets[0], "id", "") == "weight"
                and "random" in ast.unparse(s))
    ns["_weight_stmt"] = compile(ast.Module(body=[stmt], type_ignores=[]), "data.py:701", "exec")
    return ns


def main():
    ns = extract()
    rng = np.random.default_rng(20240601)
    D, H, W, cube, B = 44, 48, 52, 32, 3
    img = rng.integers(-1500, 1700, (D, H, W)).astype(np.int16)            # HU (= file value - 1024, data.py:692)
    label = (rng.random((D, H, W)) < 0.2).astype(np.uint8)
    weight16 = (rng.random((D, H, W)) * 2.6).astype(np.float16)            # LIB weights are stored as float16 (lib_weight.py:50)
    data = {"img": img, "label": label, "weight16": weight16, "cube": cube, "batch": B}

    # ---- stage 1: CropSegData.__getitem__ (data.py:689-715) composed from the reference's own pieces, seeded
    ds = ns["CropSegData"].__new__(ns["CropSegData"])
    ds.batch_size = B
    for seed in (1, 2):
        random.seed(100 + seed)
        np.random.seed(200 + seed)
        im, im2, lab = ds.process_imgmsk(img.copy(), label.copy())
        loc = {"weight": weight16.copy(), "label": lab, "np": np}
        exec(ns["_weight_stmt"], loc)                                       # weight = weight ** (np.random.random() + 2) * label + (1 - label)
        dl = ds.crop([im, im2, lab, loc["weight"]], crop_size=[cube, cube, cube])
        for i in range(len(dl[0])):
            aug = ds.augment([dl[j][i] for j in range(len(dl))])
            for j in range(len(dl)):
                dl[j][i] = aug[j]
        packs = [np.array(x) for x in dl]
        data[f"s1_{seed}_data"] = np.stack([packs[0], packs[1]], 1).astype(np.float32)      # train.py:582-592
        data[f"s1_{seed}_label"] = packs[2][:, None].astype(np.float32)
        data[f"s1_{seed}_weight"] = packs[3][:, None].astype(np.float32)
        print("stage-1 batch", seed, data[f"s1_{seed}_data"].shape, "weight dtype before .float():", packs[3].dtype)

    # ---- stage 2/3 normalisation of int16 crops: AirwayHMData.process_img (data.py:286-299): float64 true division
    hm = ns["AirwayHMData"].__new__(ns["AirwayHMData"])
    crops = [img[4:36, 8:40, 12:44].copy(), img[12:44, 0:32, 20:52].copy()]
    c0, c1 = hm.process_img([c.copy() for c in crops])
    data["s2_starts"] = np.array([[4, 8, 12], [12, 0, 20]])
    data["s2_data"] = np.stack([np.stack([a, b]) for a, b in zip(c0, c1)]).astype(np.float32)
    print("stage-2 crops: dtype before .float():", c0[0].dtype)

    # ---- every flip / rotate combination on an index-coded cube (the index maps themselves)
    n = 6
    code = np.arange(n ** 3, dtype=np.int32).reshape(n, n, n)
    combos = []
    for f0 in (1, -1):
        for f1 in (1, -1):
            for f2 in (1, -1):
                for rot in (0, 1, 2):
                    x = code
                    x = np.ascontiguousarray(x[::f0, ::f1, ::f2])
                    if rot:
                        # random_rotate picks by `k > 0.5`: monkey-free selection by seeding random so that k falls on the side wanted
                        random.seed(0 if rot == 1 else 1)   # seed 0 -> k = 0.84 (rotate_left), seed 1 -> k = 0.13 (rotate_right)
                        k = random.random()
                        assert (k > 0.5) == (rot == 1)
                        random.seed(0 if rot == 1 else 1)   # seed 0 -> k = 0.84 (rotate_left), seed 1 -> k = 0.13 (rotate_right)
                        x = ns["random_rotate"]([x])[0]
                    combos.append((f0, f1, f2, rot, x))
    data["aug_params"] = np.array([c[:4] for c in combos])
    data["aug_out"] = np.stack([c[4] for c in combos])
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
