"""Race hunt, forward only: repeated eval forwards with every block's raw conv output / statistics / output read back, while a
second process keeps the GPU busy; reports WHICH tensors differ from the first repetition, in graph order.
usage: python scripts/r4_stress2.py <dtype> <batch> <size> <reps>"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import seunet_amd as A
import seunet_oracle as orc
import forced_oracle as FO

dtype, batch, size, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
noise = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "stress_shared_gpu.py"), dtype, "1", "64", "0", "noise"])
try:
    b = orc.synthetic_batch(batch, (size,) * 3, 2, seed=3)
    x = b["image"].cuda()
    m = A.SE_UNet(2, 1, act_dtype=dtype)
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, seed=0))
    m = m.cuda().eval()
    import time
    time.sleep(12)          # (the perturbing process needs ~10 s to import torch and start its loop)
    first, bad = None, 0
    for i in range(reps):
        p0, p1, inter = m.forward_with_intermediates(x, FO.LRELU_ORDER)
        rec = {}
        for n in FO.LRELU_ORDER:
            for k, t in inter[n].items():
                rec[n + "." + k] = t.clone()
        rec["pred0"], rec["pred1"] = p0.clone(), p1.clone()
        if first is None:
            first = rec
            continue
        diff = [k for k in rec if not torch.equal(rec[k], first[k])]
        if diff:
            bad += 1
            k = diff[0]
            a, r = rec[k].double(), first[k].double()
            nbad = int((a != r).sum())
            print(f"rep {i}: {len(diff)} tensors differ; first {k}: {nbad} elements, rel {float((a - r).norm() / max(float(r.norm()), 1e-30)):.3e}; {diff[:10]}", flush=True)
    print(f"{dtype} {batch}x{size}^3 forward: {reps} repetitions, {bad} differ from the first", flush=True)
finally:
    noise.kill()
    noise.wait()
