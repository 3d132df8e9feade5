"""Diagnostic (GPU): per-parameter gradient error map of the HIP path vs the CPU oracle + forward errors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import seunet_amd as A
import seunet_oracle as orc

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
impl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
size = int(sys.argv[3]) if len(sys.argv) > 3 else 32
m = A.SE_UNet(2, 1, act_dtype=dtype, conv_impl=impl)
m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, 0))
m = m.cuda().eval()
o = orc.build_oracle(2, 1, 1, 0)
b = orc.synthetic_batch(2, (size,) * 3, 2, seed=3)
pe, pd = o(b["image"])
l_ref = orc.stage_loss(1, pe, pd, b["label"])
l_ref.backward()
ge, gd = m(b["image"].cuda())
print("fwd max|err| pred0 %.3e pred1 %.3e (|ref| max %.3f)" % (float((ge.cpu() - pe).abs().max()), float((gd.cpu() - pd).abs().max()), float(pd.abs().max())))
loss = A.fused_stage_loss(1, ge, gd, b["label"].cuda())
loss.backward()
print("loss %.7f ref %.7f" % (float(loss), float(l_ref)))
o64 = orc.build_oracle(2, 1, 1, 0).double()
pe64, pd64 = o64(b["image"].double())
orc.stage_loss(1, pe64, pd64, b["label"].double()).backward()
print("fwd vs fp64 oracle: HIP pred1 %.3e | fp32 oracle pred1 %.3e" % (float((gd.detach().cpu().double() - pd64).abs().max()), float((pd.double() - pd64).abs().max())))
print("%-22s %-12s %-12s %-12s" % ("param", "HIP-vs-f64", "orc32-vs-f64", "HIP-vs-orc32"))
for (n, p), (_, q), (_, q64) in zip(m.named_parameters(), o.named_parameters(), o64.named_parameters()):
    if q.grad is None or n.endswith("conv1.bias"):
        continue
    r = q64.grad; g = p.grad.cpu().double(); r32 = q.grad.double()
    nr = max(float(r.norm()), 1e-30)
    print("%-22s %.3e    %.3e    %.3e" % (n, float((g - r).norm() / nr), float((r32 - r).norm() / nr), float((g - r32).norm() / nr)))
# is the HIP backward consistent with the HIP forward?  directional finite difference in fp32
if dtype == "fp32":
    torch.manual_seed(0)
    names = ["dc5.conv1.weight", "ec3.conv1.weight", "ec1.conv1.weight", "ec8.conv1.weight", "ec33.conv1.weight"]
    params = dict(m.named_parameters())
    for nm in names:
        p = params[nm]
        u = torch.randn_like(p); u /= u.norm()
        ana = float((p.grad * u).sum())
        eps = 1e-2
        vals = []
        with torch.no_grad():
            for sgn in (1, -1):
                p.add_(sgn * eps * u)
                e, d = m(b["image"].cuda())
                vals.append(float(A.fused_stage_loss(1, e, d, b["label"].cuda())))
                p.sub_(sgn * eps * u)
        fd = (vals[0] - vals[1]) / (2 * eps)
        qo = dict(o.named_parameters())[nm]
        print("%-20s analytic %.6e  finite-diff %.6e  oracle-analytic %.6e" % (nm, ana, fd, float((qo.grad * u.cpu()).sum())))
