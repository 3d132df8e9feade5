"""Diagnostic (GPU): dc6 dgrad -> dc5 gate backward -> dc5 wgrad on the real network tensors (f64 reference)."""
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
def mk(name):
    def hook(mod, inp, out):
        cap[name + ".x"] = inp[0]; cap[name + ".raw"] = out; out.retain_grad()
    return hook
hs = [getattr(o, n).conv1.register_forward_hook(mk(n)) for n in ("dc5", "dc6")]
orig = o._gated
def gated(name, t):
    e, s = orig(name, t)
    if name == "dc5":
        e.retain_grad(); cap["dc5.e"] = e
    return e, s
o._gated = gated
pe, pd = o(b["image"].double())
pd.retain_grad()
orc.stage_loss(1, pe, pd, b["label"].double()).backward()
gl = pd.grad.float().reshape(2, 32, 32, 32).contiguous().cuda()
x5, raw5, draw5 = cap["dc5.x"].detach(), cap["dc5.raw"].detach(), cap["dc5.raw"].grad
raw6, draw6 = cap["dc6.raw"].detach(), cap["dc6.raw"].grad
e5 = cap["dc5.e"]; ge5_total = e5.grad
W6 = o.dc6.conv1.weight.detach()
xx = e5.detach().clone().requires_grad_(True)
F.conv3d(xx, W6, padding=1).backward(draw6)
dgrad_ref = xx.grad
print("side-path share of g_e: |total - dgrad| / |total| = %.3e" % rel(dgrad_ref, ge5_total))
for impl in (0, 1):
    (g,), _, _ = S.conv3d([S.to_cl(draw6.float().cuda(), "fp32")], W6.float().cuda(), None, 1, impl, transpose_flip=True)
    print("impl", impl, "dgrad(dc6) rel", rel(S.from_cl(g), dgrad_ref))
    w = {k: v.detach().float().cuda() for k, v in o.dc5.named_parameters()}
    hw = o.dc0_1.weight.detach().reshape(-1)[8:10].float().cuda()
    rawc = S.to_cl(raw5.float().cuda(), "fp32")
    part, slots = S.channel_stats(rawc)
    mean, rstd = S.stats_finalize(part, slots, 32 ** 3)
    for label, ge in (("ref dgrad as g_e", S.to_cl(dgrad_ref.float().cuda(), "fp32")), ("own dgrad as g_e", g)):
        out = S.gate_epilogue_bwd(rawc, mean, rstd, w["conv_se.weight"], None, w["conv2.weight"], w["conv2.bias"], g_e=ge, g_level=gl, head_w=hw)
        print("   [%s] draw(dc5) rel %.3e  dw_se rel %.3e" % (label, rel(S.from_cl(out["draw"]), draw5), rel(out["dw_se"], o.dc5.conv_se.weight.grad.reshape(-1))))
        srcs = [S.to_cl(x5[:, :32].float().cuda(), "fp32"), S.to_cl(x5[:, 32:].float().cuda(), "fp32")]
        print("   [%s] wgrad(dc5) rel %.3e" % (label, rel(S.conv3d_wgrad(srcs, out["draw"], 64, 32, 27, 1, impl), o.dc5.conv1.weight.grad)))
    print("   wgrad(dc5) with ref draw rel %.3e" % rel(S.conv3d_wgrad(srcs, S.to_cl(draw5.float().cuda(), "fp32"), 64, 32, 27, 1, impl), o.dc5.conv1.weight.grad))
    # dgrad of dc5 into its two sources
    x2 = x5.detach().clone().requires_grad_(True)
    F.conv3d(x2, o.dc5.conv1.weight.detach(), padding=1).backward(draw5)
    d0 = torch.empty((2, 32, 32, 32, 32), device="cuda"); d1 = torch.empty_like(d0)
    S.conv3d([S.to_cl(draw5.float().cuda(), "fp32")], o.dc5.conv1.weight.detach().float().cuda(), None, 1, impl, transpose_flip=True, dsts=[d0, d1])
    print("   dgrad(dc5) rel dst0 %.3e dst1 %.3e" % (rel(S.from_cl(d0), x2.grad[:, :32]), rel(S.from_cl(d1), x2.grad[:, 32:])))
