"""Accuracy of the conv epilogue's InstanceNorm partial sums against f64 sums of the stored f32 output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import seunet_amd
from seunet_amd import ops as S, _lib
torch.manual_seed(0)
for (cin, cout, dil, size, bias_v) in [(8, 8, 1, 32, 0.0), (16, 32, 2, 32, 0.0), (32, 16, 1, 48, 3.0), (64, 64, 1, 32, 0.5), (8, 32, 1, 40, 0.0),
                                       (64, 128, 1, 4, 0.2), (128, 128, 2, 4, 0.2), (32, 64, 2, 8, 0.1), (64, 64, 1, 16, 0.1), (128, 128, 1, 2, 0.3), (16, 32, 2, 6, 0.0)]:
    x = torch.randn((2, size, size, size, cin), device="cuda") + 0.3
    w = torch.randn((cout, cin, 3, 3, 3), device="cuda") * 0.1
    b = torch.full((cout,), bias_v, device="cuda")
    (raw,), part, slots = S.conv3d([x], w, b, dil, 0, want_stats=True)
    tot = part.sum(1)                       # [n][c][2] f64
    r64 = raw.double()
    import torch.nn.functional as F
    ref = F.conv3d(x.permute(0, 4, 1, 2, 3).double().cpu(), w.double().cpu(), b.double().cpu(), padding=dil, dilation=dil).permute(0, 2, 3, 4, 1)
    e_out = (r64.cpu() - ref).abs().max().item() / ref.abs().max().item()
    s1 = r64.sum(dim=(1, 2, 3)); s2 = (r64 * r64).sum(dim=(1, 2, 3))
    cnt = size ** 3
    mean = s1 / cnt; var = s2 / cnt - mean * mean; sd = var.sqrt()
    e_mean = ((tot[..., 0] - s1) / cnt / sd).abs().max().item()
    e_sq = ((tot[..., 1] - s2) / s2).abs().max().item()
    mean_k = tot[..., 0] / cnt; var_k = tot[..., 1] / cnt - mean_k * mean_k
    e_var = ((var_k - var) / var).abs().max().item()
    print("cin %d cout %d dil %d size %d bias %.1f: out rel %.2e  mean err/sd %.2e  sumsq rel %.2e  var rel %.2e" % (cin, cout, dil, size, bias_v, e_out, e_mean, e_sq, e_var), flush=True)
