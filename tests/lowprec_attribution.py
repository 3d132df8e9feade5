"""Where does the 16-bit storage modes' gradient error come from?  (GPU; imports the oracle, hence under tests/)

For bf16 / fp16 activation storage: parameter-gradient distance from (a) the plain float64 oracle and (b) the float64 oracle
with the path's OWN LeakyReLU-sign / max-pool arg-max choices imposed (tests/forced_oracle.py).  (a) - (b) is what the
discrete choices of a 16-bit forward cost (any 16-bit implementation pays it: PyTorch's bf16 autocast of the oracle is listed
beside it); (b) is the arithmetic / storage error of this implementation's backward given its forward.

usage: python tests/lowprec_attribution.py [size=32] [batch=2] [out.md]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import seunet_amd as A
import seunet_oracle as orc
import forced_oracle as FO

size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
out_md = sys.argv[3] if len(sys.argv) > 3 else None
b = orc.synthetic_batch(batch, (size,) * 3, 2, seed=3)

o64 = orc.build_oracle(2, 1, 1, seed=0).double()
pe, pd = o64(b["image"].double())
l64 = orc.stage_loss(1, pe, pd, b["label"].double())
l64.backward()
p64 = pd.detach()


def errs(named, ref):
    out = {}
    for (name, p), (_, q) in zip(named, ref.named_parameters()):
        if q.grad is None or name.endswith("conv1.bias"):
            continue
        out[name] = (float((p.grad.detach().cpu().double() - q.grad).norm() / max(float(q.grad.norm()), 1e-30)), q.grad.numel())
    return out


def summary(e, big_only=False):
    v = np.array([x for x, n in e.values() if (n >= 4096 or not big_only)])
    return "%.2e / %.2e / %.2e" % (np.median(v), np.percentile(v, 90), v.max())


rows = []
# PyTorch's bf16 autocast of the oracle, for scale
oa = orc.build_oracle(2, 1, 1, seed=0)
with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
    ae, ad = oa(b["image"])
orc.stage_loss(1, ae.float(), ad.float(), b["label"]).backward()
rows.append(("torch bf16 autocast (CPU)", float((ad.detach().double() - p64).abs().max()), errs(oa.named_parameters(), o64), None, None))
for dtype in ("bf16", "fp16", "fp32"):
    m = A.SE_UNet(2, 1, act_dtype=dtype)
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, 0))
    m = m.cuda().eval()
    _, _, inter = m.forward_with_intermediates(b["image"].cuda(), FO.LRELU_ORDER)
    ge, gd = m(b["image"].cuda())
    A.fused_stage_loss(1, ge, gd, b["label"].cuda()).backward()
    sg, pl = FO.path_choices(inter)
    of, _, fd, lf, nsf, npf = FO.forced_step(orc, b, 1, sg, pl)
    rows.append((f"HIP {dtype}", float((gd.detach().cpu().double() - p64).abs().max()), errs(m.named_parameters(), o64),
                 errs(m.named_parameters(), of), (nsf, npf, float((gd.detach().cpu().double() - fd).abs().max()))))

L = []
P = L.append
P(f"# 16-bit storage modes: gradient error attribution ({batch} x 2 x {size}^3, stage-1 Dice, against float64)")
P("")
P("median / p90 / max of the per-tensor relative L2 error; 'large' = tensors of >= 4096 elements (the conv weights)")
P("")
P("| path | logits max abs err vs f64 | grads vs plain f64 (all) | grads vs plain f64 (large) | choices differing from f64 (signs / arg-max) | logits vs same-choice f64 | grads vs same-choice f64 (all) | (large) |")
P("|---|---|---|---|---|---|---|---|")
for tag, le, e, ef, fl in rows:
    if ef is None:
        P(f"| {tag} | {le:.2e} | {summary(e)} | {summary(e, True)} | - | - | - | - |")
    else:
        P(f"| {tag} | {le:.2e} | {summary(e)} | {summary(e, True)} | {fl[0]} / {fl[1]} | {fl[2]:.2e} | {summary(ef)} | {summary(ef, True)} |")
P("")
P("## per tensor (large tensors): vs plain f64 | vs same-choice f64")
P("")
P("| tensor | " + " | ".join(f"{r[0]}" for r in rows) + " | " + " | ".join(f"{r[0]} (same choices)" for r in rows[1:]) + " |")
P("|---|" + "---|" * (2 * len(rows) - 1))
for nm, (_, n) in rows[0][2].items():
    if n < 4096:
        continue
    P(f"| {nm} | " + " | ".join(f"{r[2][nm][0]:.2e}" for r in rows) + " | " + " | ".join(f"{r[3][nm][0]:.2e}" for r in rows[1:]) + " |")
text = "\n".join(L)
print(text)
if out_md:
    open(out_md, "w").write(text + "\n")
