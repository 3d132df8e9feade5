"""Race hunt at op level: head_fwd on fixed level maps, and the gated-block epilogue with level-map accumulation, repeated while a
second process keeps the GPU busy."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import seunet_amd as A
from seunet_amd import ops as S

size, reps = int(sys.argv[1]), int(sys.argv[2])
noise = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "stress_shared_gpu.py"), "bf16", "1", "64", "0", "noise"])
try:
    import time
    torch.zeros(1, device="cuda")
    time.sleep(12)          # (the perturbing process needs ~10 s to import torch and start its loop)
    g = torch.Generator(device="cuda").manual_seed(1)
    maps = [torch.randn((1, size >> l, size >> l, size >> l), generator=g, device="cuda") for l in range(4)]
    bias = torch.zeros(1, device="cuda")
    first, bad = None, 0
    for i in range(reps):
        p = S.head_fwd(maps, bias)
        if first is None:
            first = p.clone()
        elif not torch.equal(p, first):
            bad += 1
            print(f"head_fwd rep {i}: {int((p != first).sum())} elements differ", flush=True)
    print(f"head_fwd {size}^3: {reps} repetitions, {bad} differ", flush=True)
    # gated epilogue writing / accumulating a level map: every channel count / gate count / storage type of the network
    for dt in ("fp32", "bf16"):
        for C, two in ((8, False), (16, False), (32, False), (32, True), (64, True)):
            raw = S.to_cl(torch.randn((1, C, size, size, size), generator=g, device="cuda"), dt)
            part, slots = S.channel_stats(raw)
            mean, rstd = S.stats_finalize(part, slots, size ** 3)
            w_se = torch.randn(C, generator=g, device="cuda") * 0.2
            w_se2 = torch.randn(C, generator=g, device="cuda") * 0.2 if two else None
            w_side = torch.randn((2, C), generator=g, device="cuda") * 0.2
            b_side = torch.zeros(2, device="cuda")
            head_w = torch.tensor([0.3, -0.2], device="cuda")
            first, bad = None, 0
            for i in range(reps // 4):
                lm = torch.zeros((1, size, size, size), device="cuda")
                for k in range(3):
                    out = S.gate_epilogue_fwd(raw, mean, rstd, w_se, w_se2, w_side, b_side, level_map=lm, level_accumulate=1 if k else 0, head_w=head_w, want_side=False)
                rec = (lm.clone(), out[0].clone())
                if first is None:
                    first = rec
                elif not (torch.equal(rec[0], first[0]) and torch.equal(rec[1], first[1])):
                    bad += 1
                    msk = rec[0] != first[0]
                    idx = msk.flatten().nonzero().flatten()
                    if bad <= 4:
                        print(f"gate_epilogue {dt} C={C} gates={2 if two else 1} rep {i}: level map {int(msk.sum())} / e {int((rec[1] != first[1]).sum())} elements differ; "
                              f"flat indices {idx[:12].tolist()}", flush=True)
            print(f"gate_epilogue_fwd x3 {dt} C={C} gates={2 if two else 1} {size}^3: {reps // 4} repetitions, {bad} differ", flush=True)
finally:
    noise.kill()
    noise.wait()
