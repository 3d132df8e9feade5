"""Host-side timeline of the training step (no synchronisation inside): when does each phase return on the host, against
the GPU's step time?  Finds host stalls (a host that is not ahead of the GPU shows up as idle gaps between kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import seunet_amd as A

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = A.SE_UNet(in_channel=2, n_classes=1, act_dtype="bf16").to(dev).eval()
opt = A.AdamW(m.parameters(), lr=1e-4)
x = torch.rand((4, 2, 128, 128, 128), device=dev)
label = (torch.rand((4, 1, 128, 128, 128), device=dev) < 0.03).float()
rows = []
for i in range(12):
    if i == 4:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    pe, pd = m(x)
    t2 = time.perf_counter()
    loss = A.fused_stage_loss(1, pe, pd, label)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    rows.append((t0, t1, t2, t3, t4, t5))
torch.cuda.synchronize()
tend = time.perf_counter()
for i, r in enumerate(rows):
    print("step %2d  start %8.3f ms | zero %.3f  fwd %.3f  loss %.3f  bwd %.3f  opt %.3f  = %.3f ms host" %
          (i, (r[0] - rows[4][0]) * 1e3, *[(r[k + 1] - r[k]) * 1e3 for k in range(5)], (r[5] - r[0]) * 1e3))
print("GPU done at %.3f ms after step 4 started -> %.3f ms per step" % ((tend - rows[4][0]) * 1e3, (tend - rows[4][0]) * 1e3 / 8))
