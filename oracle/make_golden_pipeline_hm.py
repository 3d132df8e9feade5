"""Generate tests/golden/pipeline_hm_known.npz: the stage-2 / stage-3 samplers of the REFERENCE (build container only).

data.py cannot be imported whole (SimpleITK, nibabel: SURVEY 8(c)); the samplers (hard_sample, random_sample, *_sample_wg) and
the methods AirwayHMData.crop / process_img / augment, AirwayHMData3.crop / process_img / augment are pure numpy + scipy, so
they are ast-extracted from the source text and run here under seeded generators on a synthetic case; the weight statements of
the two __getitem__ bodies (data.py:388, :566) are executed from the source as well.  Only data (inputs, outputs) is written.
Usage: python oracle/make_golden_pipeline_hm.py
"""
import ast
import os
import random
from copy import deepcopy

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "pipeline_hm_known.npz")
FUNCS = ("random_flip", "random_rotate", "hard_sample", "random_sample", "skeleton_sample_wg", "break_sample_wg",
         "small_airway_sample_wg", "random_sample_wg")


def extract():
    src = open("/root/reference/data.py").read()
    tree = ast.parse(src)
    ns = {"np": np, "random": random, "deepcopy": deepcopy, "ndimage": ndimage}
    funcs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in FUNCS]
    exec(compile(ast.Module(body=funcs, type_ignores=[]), "data.py", "exec"), ns)
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("AirwayHMData", "AirwayHMData3")]:
        getitem = next(m for m in cls.body if isinstance(m, ast.FunctionDef) and m.name == "__getitem__")
        stmt = [s for s in getitem.body if isinstance(s, ast.Assign) and getattr(s.targets[0], "id", "") == "weight"
                and "random" in ast.unparse(s)][0]
        ns["_weight_stmt_" + cls.name] = compile(ast.Module(body=[stmt], type_ignores=[]), "data.py", "exec")
        cls.bases = []
        cls.body = [m for m in cls.body if isinstance(m, ast.FunctionDef) and m.name in ("crop", "process_img", "augment")]
        exec(compile(ast.Module(body=[cls], type_ignores=[]), "data.py", "exec"), ns)
    return ns


def run(ns, stage, vols, B, cube, seed):
    random.seed(300 + seed)
    np.random.seed(400 + seed)
    cls = ns["AirwayHMData" if stage == 2 else "AirwayHMData3"]
    ds = cls.__new__(cls)
    ds.batch_size, ds.cube_size = B, cube
    if stage == 2:
        ds.hard_ratio = 0.4
    else:
        ds.hard_ratio, ds.break_ratio = 0.8, 0.625
    loc = {"weight": vols["weight16"].copy(), "label": vols["label"], "np": np}
    exec(ns["_weight_stmt_" + cls.__name__], loc)
    if stage == 2:
        crops = ds.crop(vols["img"], vols["label"], loc["weight"], vols["pred"], vols["skeleton"], None)
    else:
        crops = ds.crop(vols["img"], vols["label"], loc["weight"], vols["pred"], vols["skeleton"], None, vols["br_skel"])
    crops = [list(c) for c in crops]
    img_crops, img2_crops = ds.process_img([c.copy() for c in crops[0]])
    dl = [img_crops, img2_crops] + crops[1:]
    for i in range(len(dl[0])):
        aug = ds.augment([dl[j][i] for j in range(len(dl))])
        for j in range(len(dl)):
            dl[j][i] = aug[j]
    packs = [np.array(x) for x in dl]
    out = {"data": np.stack([packs[0], packs[1]], 1).astype(np.float32), "label": packs[2][:, None].astype(np.float32),
           "weight": packs[3][:, None].astype(np.float32)}
    if stage == 3:
        out["skel"] = packs[4][:, None].astype(np.float32)
    return out


def main():
    ns = extract()
    rng = np.random.default_rng(20250704)
    D, H, W, cube, B = 44, 48, 52, 32, 6
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    img = (((zz * 131 + yy * 17 + xx * 7) % 3200) - 1500).astype(np.int16)    # (a pattern, so that the fixture compresses)
    label = ndimage.binary_dilation(rng.random((D, H, W)) < 0.004, iterations=2).astype(np.uint8)
    skeleton = (label * (rng.random((D, H, W)) < 0.3)).astype(np.uint8)
    pred = (label * (rng.random((D, H, W)) < 0.7)).astype(np.float64)        # nibabel get_fdata() is float64
    weight16 = (((zz * 5 + yy * 3 + xx) % 64) / 64.0 * 2.6).astype(np.float16)
    brw16 = ((((zz + yy * 7 + xx * 3) % 32) / 32.0 * 1.4) * label).astype(np.float16)
    weight3 = weight16 + 0.6 * brw16                                          # data.py:553-557 (float16 arithmetic)
    assert weight3.dtype == np.float16
    br_skel = np.where((skeleton * (rng.random((D, H, W)) < 0.2)) == 1)        # weight_br.py:171 saves np.where(br_skel == 1)
    data = {"img": img, "label": label, "skeleton": skeleton, "pred": pred, "weight16": weight16, "weight3": weight3,
            "br_skel": np.stack(br_skel), "cube": cube, "batch": B}
    for seed in (1, 2):
        o = run(ns, 2, {"img": img, "label": label, "weight16": weight16, "pred": pred, "skeleton": skeleton}, B, cube, seed)
        for k, v in o.items():
            data[f"s2_{seed}_{k}"] = v
        o = run(ns, 3, {"img": img, "label": label, "weight16": weight3, "pred": pred, "skeleton": skeleton, "br_skel": br_skel},
                B, cube, seed)
        for k, v in o.items():
            data[f"s3_{seed}_{k}"] = v
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes", {k: v.shape for k, v in data.items() if hasattr(v, "shape") and k.startswith("s")})


if __name__ == "__main__":
    main()
