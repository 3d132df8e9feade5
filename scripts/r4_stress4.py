"""Race hunt over the rest of the path while a second process keeps the GPU busy: the window loop (gather / accumulate /
finalize around the inference forward), the double-threshold sweep, the largest component, and a fused AdamW step -- every
repetition must reproduce the first bit for bit."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import seunet_amd as A
import seunet_oracle as orc

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
noise = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "stress_shared_gpu.py"), "bf16", "1", "64", "0", "noise"])
try:
    m = A.SE_UNet(2, 1, act_dtype="bf16")
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, seed=0))
    m = m.cuda().eval()
    x = orc.synthetic_batch(1, (48, 40, 64), 2, seed=5)["image"].cuda()
    torch.zeros(1, device="cuda")
    time.sleep(12)
    first = None
    bad = {"window": 0, "dti": 0, "cc": 0, "adamw": 0}
    for i in range(reps):
        vol = A.sliding_window_predict(m, x, cube=32, step=16, batch=3, return_tensor=True)
        seg = A.double_threshold_iteration(vol, 0.5, 0.4)          # CUDA tensor in -> uint8 CUDA tensor out
        blob = (torch.rand((64, 64, 64), generator=torch.Generator().manual_seed(3)) > 0.6).to(torch.uint8).cuda()
        cc = A.largest_component(blob)
        cc = torch.as_tensor(cc)
        # one fused AdamW step on fixed gradients
        p = [torch.nn.Parameter(torch.linspace(-1, 1, 100_003, device="cuda")), torch.nn.Parameter(torch.ones(777, device="cuda"))]
        opt = A.AdamW(p, lr=1e-3)
        for q in p:
            q.grad = torch.sin(q.detach() * 3.0)
        opt.step()
        rec = {"window": vol.clone(), "dti": torch.as_tensor(seg).clone(), "cc": cc.clone(), "adamw": torch.cat([q.detach().reshape(-1) for q in p]).clone()}
        if first is None:
            first = rec
            continue
        for k in rec:
            if not torch.equal(rec[k].cpu(), first[k].cpu()):
                bad[k] += 1
                if bad[k] <= 3:
                    print(f"rep {i}: {k} differs", flush=True)
    print(f"{reps} repetitions; differing: {bad}", flush=True)
finally:
    noise.kill()
    noise.wait()
