import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import seunet_amd as A
from seunet_amd import ops as S
torch.set_printoptions(precision=2, linewidth=200, sci_mode=False)
def run(xc, cin, cout, dil, shape, mode):
    n, d, h, w = shape
    g = torch.Generator().manual_seed(1)
    if mode == "ones":
        x = torch.ones(n, xc, d, h, w); dy = torch.ones(n, cout, d, h, w)
    elif mode == "xz":   # x depends on z only, dy = 1
        x = torch.arange(d).float().view(1, 1, d, 1, 1).expand(n, xc, d, h, w).clone(); dy = torch.ones(n, cout, d, h, w)
    elif mode == "chan":
        x = torch.arange(xc).float().view(1, xc, 1, 1, 1).expand(n, xc, d, h, w).clone() + 1; dy = (torch.arange(cout).float().view(1, cout, 1, 1, 1).expand(n, cout, d, h, w).clone() + 1)
    elif mode.startswith("row"):   # dy = 1 only on output row k
        k = int(mode[3:])
        x = torch.ones(n, xc, d, h, w); dy = torch.zeros(n, cout, d, h, w); dy[:, :, :, k] = 1
    elif mode.startswith("pln"):   # dy = 1 only on plane k
        k = int(mode[3:])
        x = torch.ones(n, xc, d, h, w); dy = torch.zeros(n, cout, d, h, w); dy[:, :, k] = 1
    else:
        x = (torch.rand(n, xc, d, h, w, generator=g) * 2 - 1).bfloat16().float(); dy = (torch.rand(n, cout, d, h, w, generator=g) * 2 - 1).bfloat16().float()
    x[:, cin:] = 0
    wt = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x[:, :cin], wt, padding=dil, dilation=dil).backward(dy)
    got = S.conv3d_wgrad_stream(S.to_cl(x.cuda(), "bf16"), S.to_cl(dy.cuda(), "bf16"), cin, cout, dil).cpu()
    err = (got - wt.grad).abs().max().item()
    print(f"--- x_c {xc} cin {cin} cout {cout} dil {dil} shape {shape} mode {mode}: max err {err:.3f} (ref max {wt.grad.abs().max().item():.2f})")
    if err > 1e-2 * wt.grad.abs().max().item():
        print("ref[0,0]:\n", wt.grad[0, 0]); print("got[0,0]:\n", got[0, 0])

for k in range(8):
    run(8, 8, 8, 1, (1, 8, 8, 32), f"row{k}")
for k in range(8):
    run(8, 8, 8, 1, (1, 8, 8, 32), f"pln{k}")
