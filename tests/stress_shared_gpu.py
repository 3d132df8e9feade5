"""The same training step repeated while a SECOND PROCESS keeps the GPU busy; every repetition must give the bits of the first
(loss, both logit volumes, all 116 gradients).  Run by tests/test_net_gpu.py::test_step_is_bitwise_stable_while_another_process_
shares_the_gpu; also a command-line tool:  python tests/stress_shared_gpu.py <dtype> <batch> <size> <reps>

Why it exists (round 4): with two processes on one GPU -- the data-parallel tests run two ranks that way -- cross-lane instructions
(DPP, ds_bpermute behind __shfl_*, ds_swizzle) that the compiler had placed right in front of an s_and_saveexec saw the NARROWED
EXEC mask in their last 16 lanes in ~5 % of the launches (disabled source lanes read as 0): side values, InstanceNorm-backward sums
and with them logits and gradients were off by 1e-5 .. 7e-3, never reproducibly, never with the GPU to itself.  Every cross-lane
result now passes through a plain vector move before anything can change EXEC (csrc/seunet_common.h, dpp_settle)."""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import seunet_amd as A
import seunet_oracle as orc

dtype, batch, size, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if len(sys.argv) > 5 and sys.argv[5] == "noise":       # the perturbing process: steps of another shape until killed
    m = A.SE_UNet(2, 1, act_dtype="bf16").cuda()
    x = torch.rand(1, 2, 64, 64, 64, device="cuda")
    lab = (torch.rand(1, 1, 64, 64, 64, device="cuda") < 0.05).float()
    t0 = time.time()
    while time.time() - t0 < 600:
        for p in m.parameters():
            p.grad = None
        e, d = m(x)
        A.fused_stage_loss(1, e, d, lab).backward()
        torch.cuda.synchronize()
    sys.exit(0)

noise = subprocess.Popen([sys.executable, os.path.abspath(__file__), dtype, "1", "64", "0", "noise"])
try:
    b = orc.synthetic_batch(batch, (size,) * 3, 2, seed=3)
    x, lab = b["image"].cuda(), b["label"].cuda()
    m = A.SE_UNet(2, 1, act_dtype=dtype)
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, seed=0))
    m = m.cuda().eval()
    time.sleep(12)          # (the perturbing process needs ~10 s to import torch and start its loop)
    first, bad = None, 0
    for i in range(reps):
        for p in m.parameters():
            p.grad = None
        e, d = m(x)
        loss = A.fused_stage_loss(1, e, d, lab)
        loss.backward()
        rec = {"loss": loss.detach().clone(), "e": e.detach().clone(), "d": d.detach().clone()}
        rec.update({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        if first is None:
            first = rec
            continue
        diff = [k for k in rec if not torch.equal(rec[k], first[k])]
        if diff:
            bad += 1
            k = diff[0]
            a, r = rec[k].double(), first[k].double()
            print(f"rep {i}: {len(diff)} tensors differ, first {k}: rel {float((a - r).norm() / max(float(r.norm()), 1e-30)):.3e}; all: {diff[:12]}", flush=True)
    print(f"{dtype} {batch}x{size}^3: {reps} repetitions, {bad} differ from the first", flush=True)
finally:
    noise.kill()
    noise.wait()
