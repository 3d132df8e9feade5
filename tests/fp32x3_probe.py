"""CPU probe for an "fp32x3" parity mode: fp32 storage, every 3x3x3 / 1x1x1 block convolution computed as three
bf16 products (hi*hi + hi*lo + lo*hi, f32 accumulation) in the forward pass, the data gradient and the weight gradient --
what three bf16 MFMAs per product would give.  Emulated on the CPU oracle with exact f32 products of bf16-representable values;
gradients are compared with the float64 oracle run with the SAME discrete choices (tests/forced_oracle.py), the gate the fp32
mode has to hold (3e-5 per tensor).  Also runs the plain fp32 oracle through the same harness for reference.

usage: python tests/fp32x3_probe.py [size=32] [batch=2] [out.md]      (test infrastructure: imports oracle/)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import torch.nn.functional as F
from torch.nn.grad import conv3d_input, conv3d_weight

import seunet_oracle as orc
import forced_oracle as FO

size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
out_md = sys.argv[3] if len(sys.argv) > 3 else None
torch.set_num_threads(8)


def split(t):
    hi = t.to(torch.bfloat16).float()
    return hi, (t - hi).to(torch.bfloat16).float()


class Conv3Split(torch.autograd.Function):
    """y = conv(x, w) with every product formed as hi*hi + hi*lo + lo*hi of the operands' bf16 halves."""

    @staticmethod
    def forward(ctx, x, w, bias, pad, dil):
        xh, xl = split(x)
        wh, wl = split(w)
        y = F.conv3d(xh, wh, None, 1, pad, dil) + F.conv3d(xh, wl, None, 1, pad, dil) + F.conv3d(xl, wh, None, 1, pad, dil)
        if bias is not None:
            y = y + bias.view(1, -1, 1, 1, 1)
        ctx.save_for_backward(x, w)
        ctx.cfg = (pad, dil, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        pad, dil, has_bias = ctx.cfg
        gh, gl = split(g)
        xh, xl = split(x)
        wh, wl = split(w)
        gx = (conv3d_input(x.shape, wh, gh, 1, pad, dil) + conv3d_input(x.shape, wl, gh, 1, pad, dil)
              + conv3d_input(x.shape, wh, gl, 1, pad, dil))
        gw = (conv3d_weight(xh, w.shape, gh, 1, pad, dil) + conv3d_weight(xl, w.shape, gh, 1, pad, dil)
              + conv3d_weight(xh, w.shape, gl, 1, pad, dil))
        return gx, gw, (g.sum(dim=(0, 2, 3, 4)) if has_bias else None), None, None


def patch_split(model):
    for name in FO.LRELU_ORDER:
        conv = getattr(model, name).conv1
        conv.forward = (lambda t, c=conv: Conv3Split.apply(t, c.weight, c.bias, c.padding[0], c.dilation[0]))


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


b = orc.synthetic_batch(batch, (size,) * 3, 2, seed=3)
lines = ["# fp32x3 probe: %d x 2 x %d^3, stage-1 Dice, gradients against float64 with the same discrete choices" % (batch, size), "",
         "| path | choices differing from float64 (signs / arg-max) | logits max abs err | gradient rel-L2: median | p90 | max | worst tensor |",
         "|---|---|---|---|---|---|---|"]
for tag, patched in (("torch fp32 (f32 products)", False), ("fp32 storage, 3 bf16 products per product", True)):
    m = orc.build_oracle(2, 1, 1, seed=0).float()
    signs, pools = FO.oracle_choices(orc, m, b["image"].float())        # (choices of the unpatched fp32 forward)
    if patched:
        patch_split(m)
        # the choices of THIS path: hook the patched forward's raw conv outputs
        raws = {}
        for n in FO.LRELU_ORDER:
            conv = getattr(m, n).conv1
            f = conv.forward
            conv.forward = (lambda t, f=f, n=n: raws.__setitem__(n, f(t).detach()) or raws[n].requires_grad_(False) * 0 + f(t))
        pool_in = []
        real = F.max_pool3d
        orc.F.max_pool3d = lambda t, *a, **k: (pool_in.append(t.detach()), real(t, *a, **k))[1]
        try:
            with torch.no_grad():
                m(b["image"].float())
        finally:
            orc.F.max_pool3d = real
        signs = {n: (r.double() - r.double().mean(dim=(2, 3, 4), keepdim=True)) > 0 for n, r in raws.items()}
        pools = {nm: F.max_pool3d(t.double(), 2, 2, return_indices=True)[1] for nm, t in zip(FO.POOL_ORDER, pool_in) if nm is not None}
        m = orc.build_oracle(2, 1, 1, seed=0).float()
        patch_split(m)
    pe, pd = m(b["image"].float())
    loss = orc.stage_loss(1, pe, pd, b["label"].float())
    loss.backward()
    o64, pe64, pd64, loss64, sf, pf = FO.forced_step(orc, b, 1, signs, pools)
    errs = {}
    for (n, p), (_, q) in zip(m.named_parameters(), o64.named_parameters()):
        if p.grad is None or q.grad is None or n.endswith("conv1.bias"):    # (conv1.bias feeds an InstanceNorm: its gradient is 0)
            continue
        errs[n] = rel(p.grad, q.grad)
    v = sorted(errs.values())
    worst = max(errs, key=errs.get)
    lines.append("| %s | %d / %d | %.2e | %.2e | %.2e | %.2e | %s |" % (
        tag, sf, pf, float((pd.detach().double() - pd64).abs().max()), v[len(v) // 2], v[int(0.9 * (len(v) - 1))], v[-1], worst))
    print(lines[-1], flush=True)
    if patched:
        lines += ["", "## per tensor, 3-product path (conv weights)", "", "| tensor | rel-L2 vs same-choice float64 |", "|---|---|"]
        lines += ["| %s | %.2e |" % (n, e) for n, e in errs.items() if n.endswith("conv1.weight")]
text = "\n".join(lines) + "\n"
if out_md:
    open(out_md, "w").write(text)
print(text)
