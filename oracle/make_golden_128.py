"""Generate tests/golden/bwd128_stage1.npz from the REAL reference at the size north_star names (build container only).

One 1 x 2 x 128^3 patch, the reference's stage-1 step body (train.py:594-602: forward, sigmoid, Dice on both heads, backward)
run by the imported ``/root/reference/SE_UNet.py`` in fp32 on the deterministic weights / inputs of ``oracle/seunet_oracle.py``
(seed 21, the batch of tests/test_net_gpu.py::test_forward_backward_128_vs_oracle_fp32).  Stored (numbers only):

  * strided samples (::8) of both logit volumes, their sums and absolute sums, the loss;
  * norm / sum / first 8 elements of every parameter gradient;
  * ``<name>|ref32_same_choice_err``: the fp32 REFERENCE's own relative-L2 distance from the float64 oracle run with the
    reference's own LeakyReLU-sign / arg-max choices imposed (tests/forced_oracle.py) -- the arithmetic error of the fp32
    reference proper, flip noise removed.  The HIP path's x-branch weight gradients are gated against these.

Takes ~4 minutes on 8 cores.  Usage (from the repo root):  python oracle/make_golden_128.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, os.path.join(ROOT, "tests")]
import seunet_oracle as orc  # noqa: E402
import forced_oracle as FO  # noqa: E402
from make_golden import OUT, grads_summary, load_reference_losses, load_reference_model_module  # noqa: E402


def main():
    torch.set_num_threads(8)
    ref = load_reference_model_module()
    dice = load_reference_losses()["dice_loss"]
    b = orc.synthetic_batch(1, (128, 128, 128), 2, seed=21)
    m = ref.SE_UNet(in_channel=2, n_classes=1).eval()
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, seed=0))
    pe, pd = m(b["image"])
    loss = dice(torch.sigmoid(pd), b["label"]) + dice(torch.sigmoid(pe), b["label"])      # train.py:595-599
    loss.backward()
    rec = grads_summary(m)
    rec["loss"] = np.array(float(loss.detach()))
    for key, t in (("pred0", pe.detach()), ("pred1", pd.detach())):
        rec[key + "_s"] = t[0, 0, ::8, ::8, ::8].numpy()
        rec[key + "_sum"] = np.array(float(t.double().sum()))
        rec[key + "_abs"] = np.array(float(t.double().abs().sum()))
    rec["meta"] = np.array([1, 2, 128, 0, 21, 1])      # B, inch, S, wseed, xseed, stage
    # the restatement takes the same step bit for bit: its choices are the reference's choices
    o = orc.build_oracle(2, 1, 1, seed=0)
    sg, pl = FO.oracle_choices(orc, o, b["image"])
    qe, qd = o(b["image"])
    orc.stage_loss(1, qe, qd, b["label"]).backward()
    assert float((qd.detach() - pd.detach()).abs().max()) == 0.0
    for (n1, p), (n2, q) in zip(m.named_parameters(), o.named_parameters()):
        assert n1 == n2 and (p.grad is None) == (q.grad is None) and (p.grad is None or torch.equal(p.grad, q.grad)), n1
    of, _, _, lf, nsf, npf = FO.forced_step(orc, b, 1, sg, pl)
    rec["ref32_choices_differing_from_f64"] = np.array([nsf, npf])
    rec["loss_f64_same_choices"] = np.array(lf)
    errs = {}
    for (name, p), (_, q) in zip(m.named_parameters(), of.named_parameters()):
        if q.grad is None or name.endswith("conv1.bias"):
            continue
        errs[name] = float((p.grad.double() - q.grad).norm() / max(float(q.grad.norm()), 1e-30))
        rec[name + "|ref32_same_choice_err"] = np.array(errs[name])
        rec[name + "|f64_same_choice_norm"] = np.array(float(q.grad.norm()))
    v = np.array(list(errs.values()))
    print("fp32 reference vs float64 with its own %d sign / %d arg-max choices: median %.2e p90 %.2e max %.2e" %
          (nsf, npf, np.median(v), np.percentile(v, 90), v.max()))
    print("   raw-input branches:", {k: "%.2e" % e for k, e in errs.items() if k.startswith("x")})
    np.savez_compressed(os.path.join(OUT, "bwd128_stage1.npz"), **rec)
    print("written", os.path.join(OUT, "bwd128_stage1.npz"))


if __name__ == "__main__":
    main()
