"""Diagnostic (GPU): the dc6 block backward in isolation on the real network tensors, step by step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import torch.nn.functional as F
import seunet_amd as A
from seunet_amd import ops as S
import seunet_oracle as orc

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm())

o = orc.build_oracle(2, 1, 1, 0).double()
b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
cap = {}
blk = o.dc6
def hook_conv(mod, inp, out):
    cap["x"] = inp[0]; cap["raw"] = out
    out.retain_grad()
h = blk.conv1.register_forward_hook(hook_conv)
pe, pd = o(b["image"].double())
pd.retain_grad()
loss = orc.stage_loss(1, pe, pd, b["label"].double())
loss.backward()
h.remove()
x, raw, draw_ref, gl = cap["x"].detach(), cap["raw"].detach(), cap["raw"].grad, pd.grad
print("ref shapes", x.shape, raw.shape, draw_ref.shape, gl.shape)
w = {k: v.detach() for k, v in blk.named_parameters()}
hw = o.dc0_1.weight.detach().reshape(-1)[10:12]
xc = S.to_cl(x.float().cuda(), "fp32")
for impl in (0, 1):
    (rawc,), part, slots = S.conv3d([xc], w["conv1.weight"].float().cuda(), w["conv1.bias"].float().cuda(), 1, impl, want_stats=True)
    print("impl", impl, "raw rel", rel(S.from_cl(rawc), raw))
    mean, rstd = S.stats_finalize(part, slots, 32 ** 3)
    rm = raw.mean(dim=(2, 3, 4)); rv = raw.var(dim=(2, 3, 4), unbiased=False)
    print("   mean rel", rel(mean, rm), "rstd rel", rel(rstd, (rv + 1e-5).rsqrt()))
    out = S.gate_epilogue_bwd(rawc, mean, rstd, w["conv_se.weight"].float().cuda(), None, w["conv2.weight"].float().cuda(),
                              w["conv2.bias"].float().cuda(), g_level=gl.float().reshape(2, 32, 32, 32).contiguous().cuda(),
                              head_w=hw.float().cuda())
    d = S.from_cl(out["draw"])
    print("   draw rel", rel(d, draw_ref), " max|ref|", float(draw_ref.abs().max()), " max|err|", float((d.cpu().double() - draw_ref).abs().max()))
    print("   draw mean per (n,c) ours", d.mean(dim=(2, 3, 4))[0, :4].tolist(), "ref", draw_ref.mean(dim=(2, 3, 4))[0, :4].tolist())
    print("   dw_se rel", rel(out["dw_se"], blk.conv_se.weight.grad.reshape(-1)), "dw_side rel", rel(out["dw_side"], blk.conv2.weight.grad.reshape(-1)))
    dw = S.conv3d_wgrad([xc], out["draw"], 32, 16, 27, 1, impl)
    print("   wgrad(ours draw) rel", rel(dw, blk.conv1.weight.grad))
    dw2 = S.conv3d_wgrad([xc], S.to_cl(draw_ref.float().cuda(), "fp32"), 32, 16, 27, 1, impl)
    print("   wgrad(ref draw)  rel", rel(dw2, blk.conv1.weight.grad))
    # where is the draw error?
    err = (d.cpu().double() - draw_ref).abs()
    print("   err by channel", err.amax(dim=(0, 2, 3, 4))[:8].tolist())
    print("   err at border z=0 %.3e interior %.3e" % (float(err[:, :, 0].max()), float(err[:, :, 8:24, 8:24, 8:24].max())))
