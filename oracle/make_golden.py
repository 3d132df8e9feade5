"""Generate tests/golden/*.npz from the REAL reference (build container only).

Imports ``/root/reference/SE_UNet.py`` (pure torch) and ast-extracts the three
self-contained loss functions from ``/root/reference/train.py`` (train.py as a whole
cannot be imported here: cc3d/SimpleITK/nibabel/... are not installed -- SURVEY.md
section 8(c)).  Runs them on the deterministic weights/inputs defined in
``oracle/seunet_oracle.py`` and stores inputs' recipe + expected outputs.  Only data
(numbers) is written; no reference source text is stored anywhere.

Usage (from the repo root):  python oracle/make_golden.py
"""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, HERE)
import seunet_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def load_reference_model_module():
    spec = importlib.util.spec_from_file_location("ref_SE_UNet", os.path.join(REF, "SE_UNet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_losses():
    src = open(os.path.join(REF, "train.py")).read()
    tree = ast.parse(src)
    want = {"dice_loss", "general_union_loss_lib", "atr_loss"}
    ns = {"torch": torch, "np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            exec(compile(ast.Module([node], []), "train.py", "exec"), ns)
    return {k: ns[k] for k in want}


def grads_summary(model):
    out = {}
    for name, p in model.named_parameters():
        if p.grad is None:
            out[name + "|none"] = np.array(1.0)
            continue
        g = p.grad.detach().double().reshape(-1)
        out[name + "|norm"] = np.array(float(g.norm()))
        out[name + "|sum"] = np.array(float(g.sum()))
        out[name + "|head"] = g[:8].float().numpy()
    return out


def main():
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference_model_module()
    losses = load_reference_losses()

    # ---- 1. eval forward, full outputs, 2x2x32^3 and in_channel=1 ------------------
    for inch, tag in ((2, "fwd32_in2"), (1, "fwd32_in1")):
        m = ref.SE_UNet(in_channel=inch, n_classes=1).eval()
        m.load_state_dict(orc.deterministic_state_dict(inch, 1, 1, seed=0))
        x = orc.synthetic_batch(2, (32, 32, 32), inch, seed=1)["image"]
        with torch.no_grad():
            p0, p1 = m(x)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), pred0=p0.numpy(), pred1=p1.numpy(),
                            meta=np.array([2, inch, 32, 0, 1]))  # B, inch, S, wseed, xseed

    # ---- 2. eval forward 1x2x64^3 (config 1): strided samples + checksums ----------
    m = ref.SE_UNet(in_channel=2, n_classes=1).eval()
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, seed=0))
    x = orc.synthetic_batch(1, (64, 64, 64), 2, seed=2)["image"]
    with torch.no_grad():
        p0, p1 = m(x)
    np.savez_compressed(os.path.join(OUT, "fwd64_in2.npz"),
                        pred0_s=p0[0, 0, ::4, ::4, ::4].numpy(), pred1_s=p1[0, 0, ::4, ::4, ::4].numpy(),
                        pred0_sum=np.array(float(p0.double().sum())), pred1_sum=np.array(float(p1.double().sum())),
                        pred0_abs=np.array(float(p0.double().abs().sum())), pred1_abs=np.array(float(p1.double().abs().sum())),
                        meta=np.array([1, 2, 64, 0, 2]))

    # ---- 3. forward+backward, stage 1 and stage 3 losses, 2x2x32^3 ------------------
    for stage in (1, 3):
        m = ref.SE_UNet(in_channel=2, n_classes=1).eval()
        m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, seed=0))
        b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
        pe, pd = m(b["image"])
        se, sd_ = torch.sigmoid(pe), torch.sigmoid(pd)
        if stage == 1:      # train.py:595-599
            loss = losses["dice_loss"](sd_, b["label"]) + losses["dice_loss"](se, b["label"])
        else:               # train.py:235-243
            gul = losses["general_union_loss_lib"]
            atr = losses["atr_loss"]
            loss = (gul(sd_, b["label"], b["weight"]) + 0.5 * gul(se, b["label"], b["weight"])
                    + 0.5 * (atr(se, b["label"], b["skel"], b["weight"]) + atr(sd_, b["label"], b["skel"], b["weight"])))
        loss.backward()
        g = grads_summary(m)
        g["loss"] = np.array(float(loss.detach()))
        g["meta"] = np.array([2, 2, 32, 0, 3, stage])
        np.savez_compressed(os.path.join(OUT, f"bwd32_stage{stage}.npz"), **g)

    # ---- 4. train-mode forward (DropLayer active) ------------------------------------
    # DropLayer hard-codes .cuda() (SE_UNet.py:91); on this CPU-only host make .cuda()
    # the identity for the duration of the call (harness-side, reference untouched).
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        m = ref.SE_UNet(in_channel=2, n_classes=1).train()
        m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, seed=0))
        x = orc.synthetic_batch(2, (32, 32, 32), 2, seed=4)["image"]
        torch.manual_seed(123)
        with torch.no_grad():
            p0, p1 = m(x)
    finally:
        torch.Tensor.cuda = orig_cuda
    np.savez_compressed(os.path.join(OUT, "fwd32_train.npz"), pred0=p0.numpy(), pred1=p1.numpy(),
                        meta=np.array([2, 2, 32, 0, 4, 123]))

    # ---- 5. loss known answers (inputs documented in SURVEY.md section 8(c)) ---------
    g = torch.Generator().manual_seed(1234)
    p = torch.rand(2, 1, 16, 16, 16, generator=g)
    t = (torch.rand(2, 1, 16, 16, 16, generator=g) > 0.9).float()
    w = 1 + torch.rand(2, 1, 16, 16, 16, generator=g)
    s = t * (torch.rand(2, 1, 16, 16, 16, generator=g) > 0.5).float()
    rec = {}
    for name, args in (("dice_loss", (t,)), ("general_union_loss_lib", (t, w)), ("atr_loss", (t, s, w))):
        pp = p.clone().requires_grad_(True)
        l = losses[name](pp, *args)
        l.backward()
        rec[name] = np.array(float(l))
        rec[name + "|gradnorm"] = np.array(float(pp.grad.double().norm()))
        rec[name + "|grad"] = pp.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_known.npz"), **rec)

    # ---- 6. sliding-window start tables from the formula at prediction.py:80-100 -----
    tab = {}
    for dim in (128, 129, 191, 192, 200, 256, 300, 512):
        cube, step = 128, 64
        n = (dim - cube) // step + 1 if (dim - cube) % step == 0 else (dim - cube) // step + 2
        st = []
        for i in range(n):
            lo, hi = step * i, step * i + cube
            if hi > dim:
                hi = dim
                lo = dim - cube
            st.append(lo)
        tab[str(dim)] = np.array(st)
    np.savez_compressed(os.path.join(OUT, "window_starts.npz"), **tab)

    # ---- 7. restatement vs reference, recorded for DESIGN.md -------------------------
    o = orc.build_oracle(2, 1, 1, seed=0)
    m = ref.SE_UNet(in_channel=2, n_classes=1).eval()
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, seed=0))
    assert list(o.state_dict().keys()) == list(m.state_dict().keys())
    x = orc.synthetic_batch(1, (64, 64, 64), 2, seed=2)["image"]
    with torch.no_grad():
        a0, a1 = o(x)
        b0, b1 = m(x)
    print("restatement vs reference, 1x2x64^3: max|d pred0| %.3e  max|d pred1| %.3e" %
          (float((a0 - b0).abs().max()), float((a1 - b1).abs().max())))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
