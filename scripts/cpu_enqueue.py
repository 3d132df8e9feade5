"""How long does the host need to enqueue one training step (it must stay well below the GPU time of the step)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import seunet_amd as A
torch.manual_seed(0)
m = A.SE_UNet(2, 1, act_dtype="bf16").cuda().eval()
opt = A.AdamW(m.parameters(), lr=1e-4)
x = torch.rand(4, 2, 128, 128, 128, device="cuda")
lab = (torch.rand(4, 1, 128, 128, 128, device="cuda") < 0.03).float()
def step():
    opt.zero_grad(set_to_none=True)
    e, d = m(x)
    A.fused_stage_loss(1, e, d, lab).backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
cpu, tot = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cpu.append(t1 - t0); tot.append(t2 - t0)
print("host enqueue per step: %.2f ms; step (enqueue + drain): %.2f ms" % (1e3 * min(cpu), 1e3 * min(tot)))
