"""Whole-network parity of the HIP path (SE_UNet nn.Module -> libseunet_hip.so) against the CPU oracle
and the golden fixtures generated from the reference (tests/golden, oracle/make_golden.py).

Tolerance stated by BASELINE.json's north_star: 1e-3 (fp32) on outputs: checked on logits (2e-3 abs, measured
1e-6), sigmoid outputs (1e-3) and the loss (1e-5).  Parameter gradients are compared with the FLOAT64 oracle with
the tolerance derived in test_forward_backward_vs_oracle_fp32; the backward kernels themselves are checked
flip-free to 1e-5 on the network's own tensors.  bf16 mode (the benchmark dtype) must be at least as accurate as
PyTorch's bf16 autocast of the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FP32_ATOL = 1e-3


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import seunet_amd
    seunet_amd._lib.load()
    return seunet_amd


@pytest.fixture(scope="module")
def orc():
    import seunet_oracle
    return seunet_oracle


def build(A, orc, inch=2, dtype="fp32", impl=0, train=False, width_mult=1):
    m = A.SE_UNet(in_channel=inch, n_classes=1, width_mult=width_mult, act_dtype=dtype, conv_impl=impl)
    m.load_state_dict(orc.deterministic_state_dict(inch, 1, width_mult, seed=0))
    return m.cuda().train(train)


@pytest.mark.parametrize("tag,inch", [("fwd32_in2", 2), ("fwd32_in1", 1)])
@pytest.mark.parametrize("impl", [0, 1])
def test_eval_forward_vs_reference_golden_fp32(A, orc, golden_dir, tag, inch, impl):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    m = build(A, orc, inch, "fp32", impl)
    x = orc.synthetic_batch(2, (32, 32, 32), inch, seed=1)["image"]
    with torch.no_grad():
        p0, p1 = m(x.cuda())
    for got, key in ((p0, "pred0"), (p1, "pred1")):
        ref = torch.from_numpy(g[key])
        err = float((got.cpu() - ref).abs().max())
        serr = float((torch.sigmoid(got.cpu()) - torch.sigmoid(ref)).abs().max())
        assert err < 2e-3 and serr < FP32_ATOL, f"{key}: logits {err:.3e} sigmoid {serr:.3e}"


def test_eval_forward_64_config1_fp32(A, orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "fwd64_in2.npz"))
    m = build(A, orc, 2, "fp32")
    x = orc.synthetic_batch(1, (64, 64, 64), 2, seed=2)["image"]
    with torch.no_grad():
        p0, p1 = m(x.cuda())
    assert float((p0.cpu()[0, 0, ::4, ::4, ::4] - torch.from_numpy(g["pred0_s"])).abs().max()) < 2e-3
    assert float((p1.cpu()[0, 0, ::4, ::4, ::4] - torch.from_numpy(g["pred1_s"])).abs().max()) < 2e-3
    assert abs(float(p1.double().abs().sum()) - float(g["pred1_abs"])) < 1e-3 * float(g["pred1_abs"])


def test_train_mode_droplayer_rng_fp32(A, orc, golden_dir):
    """train(): DropLayer draws from the CPU generator in the reference's order."""
    g = np.load(os.path.join(golden_dir, "fwd32_train.npz"))
    m = build(A, orc, 2, "fp32", train=True)
    x = orc.synthetic_batch(2, (32, 32, 32), 2, seed=4)["image"]
    torch.manual_seed(123)
    with torch.no_grad():
        p0, p1 = m(x.cuda())
    assert float((p0.cpu() - torch.from_numpy(g["pred0"])).abs().max()) < 3e-3
    assert float((p1.cpu() - torch.from_numpy(g["pred1"])).abs().max()) < 3e-3


def _oracle64(orc, stage, batch):
    o = orc.build_oracle(2, 1, 1, seed=0).double()
    pe, pd = o(batch["image"].double())
    loss = orc.stage_loss(stage, pe, pd, batch["label"].double(), batch["weight"].double(), batch["skel"].double())
    loss.backward()
    return o, float(loss.detach())


def _rel_errors(model, ref64):
    """relative L2 error of every live parameter gradient against the float64 oracle."""
    out = {}
    for (name, p), (_, q) in zip(model.named_parameters(), ref64.named_parameters()):
        if name.startswith("dc62."):
            assert p.grad is None and q.grad is None, name          # dead block
            continue
        assert p.grad is not None, name
        if name.endswith("conv1.bias"):
            # bias in front of an affine-less InstanceNorm: gradient is zero up to rounding (Q4)
            assert float(p.grad.abs().max()) <= 1e-6, name
            continue
        r = q.grad
        out[name] = float((p.grad.detach().cpu().double() - r).norm() / max(float(r.norm()), 1e-30))
    return out


def _check_grad_noise(err, ref_noise):
    """Coarse companion of the same-choice gate below: the HIP path's distance from the PLAIN float64 oracle, next to the fp32
    reference's own distance from it on the same case.  Both are flip noise -- which handful of LeakyReLU signs / arg-maxes the
    fp32 forward takes differently from float64, and where they sit (measured on MI355X at 2 x 2 x 32^3, stage 1: HIP median 4.4e-4,
    p90 2.1e-3, worst 3.2e-3; fp32 reference 3.5e-6 / 1.8e-3 / 2.5e-3; in train mode one tensor reached 6.4e-3 with every tensor
    within 9e-6 of float64 under the same choices) -- so this is only a tripwire for SYSTEMATIC error (f32 instead of f64
    InstanceNorm sums shows up as a median of ~1e-2): median <= max(2e-3, 4 x ref), worst <= max(2e-2, 6 x ref).  The arithmetic
    gate is _check_vs_same_choice_f64 (3e-5 per tensor), which every caller of this function also runs."""
    v, r = np.array(list(err.values())), np.array(list(ref_noise.values()))
    stats = (float(np.median(v)), float(np.percentile(v, 90)), float(v.max()))
    ref = (float(np.median(r)), float(np.percentile(r, 90)), float(r.max()))
    print("gradient rel-L2 vs f64 (median, p90, max): HIP %.2e %.2e %.2e | fp32 reference %.2e %.2e %.2e" % (stats + ref))
    assert stats[0] <= max(2e-3, 4 * ref[0]) and stats[2] <= max(2e-2, 6 * ref[2]), \
        (stats, ref, sorted(err.items(), key=lambda kv: -kv[1])[:6])


def _check_vs_same_choice_f64(orc, m, b, stage, width_mult=1, tol=3e-5, what="", xtol=None, max_flip_frac=2e-6, drops=None):
    """The flip-free network-level gradient gate (tests/forced_oracle.py): the float64 oracle is run with the LeakyReLU signs and
    max-pool arg-maxes that THIS forward of the HIP path took (read back from its workspace; the raw-input branches x33 / x63 /
    x93, which leave no tensor when in_channel <= 2, are recomputed by the device function the aggregation epilogue uses), and
    every parameter gradient of the HIP path must agree with it to rounding.  Measured on MI355X
    (profiles/r03_flip_census_32.md): median 3.5e-6, every tensor <= 1e-5 -- the same as the fp32 reference against float64
    with ITS choices imposed (3.0e-6 / 7e-6).  `tol` = 3x the worst measured tensor.  `xtol` (default: `tol`) is for the weights of
    the three raw-input branches: their gradient sum_v draw[v] * x[v] is what is left of terms that cancel (sum_v draw[v] = 0 and
    sum_v draw[v] * xhat[v] = 0 with xhat linear in x), which costs every plain fp32 evaluation 1e-4 at 2 x 32^3 and 4e-4 ... 1.6e-3
    at 128^3 (the fp32 reference with its choices imposed, tests/golden/bwd128_stage1.npz); the HIP path forms it in f64 from sums
    (xw_finalize_kernel) and needs no allowance.
    `max_flip_frac` bounds how many choices the gate imposes (fp32 mode): the path's forward may differ from float64's in at most
    that fraction of the LeakyReLU signs + arg-maxes (measured 3e-7 ... 7e-7 at 32^3 ... 128^3, the fp32 reference alike), so a
    forward that takes many wrong branches on near-zero inputs is not forgiven by having them imposed on the oracle."""
    import forced_oracle as FO
    _, _, inter = m.forward_with_intermediates(b["image"].cuda(), FO.LRELU_ORDER)
    signs, pools = FO.path_choices(inter)
    of, fe, fd, lf, nsf, npf = FO.forced_step(orc, b, stage, signs, pools, width_mult=width_mult, drops=drops)
    err = _rel_errors(m, of)
    v = np.array(list(err.values()))
    choices = sum(int(t.numel()) for t in signs.values()) + sum(int(t.numel()) for t in pools.values())
    print(f"{what} gradient rel-L2 vs float64 with the same {nsf} sign / {npf} arg-max flips (of {choices} choices) imposed: "
          f"median {np.median(v):.2e} p90 {np.percentile(v, 90):.2e} max {v.max():.2e}")
    print("   raw-input branches:", {k: "%.2e" % e for k, e in err.items() if k.startswith("x")})
    if max_flip_frac is not None:
        assert nsf + npf <= max(4, max_flip_frac * choices), (nsf, npf, choices)
    xtol = tol if xtol is None else xtol
    bad = {k: e for k, e in err.items() if e > (xtol if k.startswith("x") else tol)}
    assert not bad and float(np.median(v)) <= tol / 3, (bad, float(np.median(v)))
    return {"loss": lf, "err": err, "pred0": fe, "pred1": fd}


@pytest.mark.parametrize("stage", [1, 3])
@pytest.mark.parametrize("impl", [0, 1])
def test_forward_backward_vs_oracle_fp32(A, orc, golden_dir, stage, impl):
    """Gradients are compared with the FLOAT64 oracle.  This network's gradient is ill-conditioned (InstanceNorm's
    backward cancels the dominant part of the Dice gradient) and contains discrete choices (LeakyReLU sign, max-pool
    argmax): one flipped element out of ~2M moves a tensor's gradient by ~1e-3 relative.  The fp32 reference itself
    differs from its own float64 run by up to 2.5e-3 on ec1..ec63 at this size (printed below, same inputs).  The bar
    is therefore distributional and relative to that measured reference noise (_check_grad_noise).  A systematic
    error (e.g. f32 instead of f64 InstanceNorm sums) shows up as median ~1e-2, two orders above the band.
    The backward KERNELS are checked flip-free to 1e-5 in test_block_backward_exact_on_real_tensors_fp32."""
    golden = np.load(os.path.join(golden_dir, f"bwd32_stage{stage}.npz"))
    m = build(A, orc, 2, "fp32", impl)
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
    o64, l64 = _oracle64(orc, stage, b)
    o32 = orc.build_oracle(2, 1, 1, seed=0)
    pe, pd = o32(b["image"])
    l32 = orc.stage_loss(stage, pe, pd, b["label"], b["weight"], b["skel"])
    l32.backward()
    assert abs(float(l32.detach()) - float(golden["loss"])) < 1e-6          # oracle == reference fixture
    for name, q in o32.named_parameters():
        if q.grad is not None and not name.endswith("conv1.bias"):      # those are pure rounding noise (Q4)
            gn = float(golden[name + "|norm"])
            assert abs(float(q.grad.double().norm()) - gn) <= 1e-4 * max(gn, 1e-9) + 1e-9, name
    c = {k: v.cuda() for k, v in b.items()}
    ge, gd = m(c["image"])
    assert float((gd.detach().cpu() - pd.detach()).abs().max()) < 1e-4
    loss = A.fused_stage_loss(stage, ge, gd, c["label"], c["weight"], c["skel"])
    loss.backward()
    assert abs(float(loss.detach()) - l64) < 1e-5
    err = _rel_errors(m, o64)
    ref_noise = _rel_errors(o32, o64)
    worst = max(err.values())
    med = float(np.median(list(err.values())))
    print(f"stage {stage} impl {impl}: HIP-vs-f64 max {worst:.2e} median {med:.2e}; fp32-reference-vs-f64 max {max(ref_noise.values()):.2e}")
    _check_grad_noise(err, ref_noise)
    # ... and flip-free: against float64 with this forward's own discrete choices every tensor agrees to rounding
    # (impl 1, the naive cross-check kernels, materialises the raw-input branches and sums their weight gradient from the f32 draw
    # like any plain fp32 evaluation: 1.2e-4 on x93, the fp32 reference's own figure at this size)
    _check_vs_same_choice_f64(orc, m, b, stage, what=f"stage {stage} impl {impl}:", xtol=1e-3 if impl else None)


def test_forward_backward_128_vs_oracle_fp32(A, orc, golden_dir):
    """The size north_star names: one 1 x 2 x 128^3 patch (reference step train.py:594-603: forward, sigmoid, Dice on both
    heads, backward) in the fp32 parity mode
      * against the fixture the imported reference wrote at this size (oracle/make_golden_128.py -> bwd128_stage1.npz: strided
        logit samples, loss, per-parameter gradient norms, SE_UNet.py:181-238 + train.py:594-602): logits 2e-3, loss 1e-5, gradient
        norms to the flip-noise band (2e-2: the fp32 reference's small raw-input-branch gradients are themselves 1.6e-3 from
        float64 under its own choices, and a handful of differing choices moves them by several 1e-3);
      * against the FLOAT64 oracle run with the SAME LeakyReLU-sign / arg-max choices (_check_vs_same_choice_f64; at most 2e-6 of the
        choices may differ from float64's own): logits and sigmoid outputs to the 1e-3 bar, the loss to 1e-5, every parameter
        gradient to rounding (3e-5; the three raw-input branch weights 1e-3);
      * the raw-input branch weights additionally against the fp32 REFERENCE's own same-choice error recorded in the fixture
        (x33 8e-4, x63 1.6e-3, x93 4e-4): the HIP path must not be worse than twice that.
    The float64 oracle step takes about a minute of host time (the plain, choice-free float64 and fp32 oracle runs this test used
    to repeat for a printed noise band live in the fixture now)."""
    golden = np.load(os.path.join(golden_dir, "bwd128_stage1.npz"))
    b = orc.synthetic_batch(1, (128, 128, 128), 2, seed=21)
    m = build(A, orc, 2, "fp32")
    ge, gd = m(b["image"].cuda())
    for got, key in ((ge, "pred0"), (gd, "pred1")):         # the reference's own numbers at this size
        gs = got.detach().cpu()
        err = float((gs[0, 0, ::8, ::8, ::8] - torch.from_numpy(golden[key + "_s"])).abs().max())
        print(f"128^3 {key}: max|logit - fp32 reference| on the fixture's samples {err:.3e}")
        assert err < 2e-3, key
        assert abs(float(gs.double().abs().sum()) - float(golden[key + "_abs"])) < 1e-4 * float(golden[key + "_abs"]), key
    loss = A.fused_stage_loss(1, ge, gd, b["label"].cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(golden["loss"])) < 1e-5
    for name, p in m.named_parameters():
        if p.grad is not None and not name.endswith("conv1.bias"):
            gn = float(golden[name + "|norm"])
            assert abs(float(p.grad.double().norm()) - gn) <= 2e-2 * gn, (name, float(p.grad.double().norm()), gn)
    # The raw-input branch weights: their gradient is what is left of terms that cancel to ~1e-6 of their size at this extent (in
    # float64 the direct sum is only good to 1e-10), so ANY systematic 1e-10-class deviation of the incoming f32 gradient shows up
    # at 1e-4: north_star's 1e-3 is the absolute bar, and the relative one is the fp32 reference's own same-choice error below.
    # Measured on MI355X: x33 6.1e-6, x63 1.1e-4, x93 3.2e-4 (reference: 8.0e-4, 1.6e-3, 4.5e-4; this path before the f64
    # formulation of round 4: x93 4.6e-2).
    f = _check_vs_same_choice_f64(orc, m, b, 1, what="128^3:", xtol=1e-3)
    for got, ref, key in ((ge, f["pred0"], "pred0"), (gd, f["pred1"], "pred1")):
        err = float((got.detach().cpu().double() - ref).abs().max())
        serr = float((torch.sigmoid(got.detach().cpu().double()) - torch.sigmoid(ref)).abs().max())
        print(f"128^3 {key} vs float64 (same choices): logits max|err| {err:.3e}  sigmoid {serr:.3e}")
        assert err < 2e-3 and serr < FP32_ATOL, f"{key}: logits {err:.3e} sigmoid {serr:.3e}"
    assert abs(float(loss.detach()) - f["loss"]) < 1e-5, (float(loss.detach()), f["loss"])
    for k in ("x33.conv1.weight", "x63.conv1.weight", "x93.conv1.weight"):
        ref_err = float(golden[k + "|ref32_same_choice_err"])
        print(f"   {k}: HIP {f['err'][k]:.2e} vs the fp32 reference's own same-choice error {ref_err:.2e}")
        assert f["err"][k] <= 2 * ref_err, (k, f["err"][k], ref_err)


def test_block_backward_exact_on_real_tensors_fp32(A, orc):
    """Flip-free check of the backward kernels on the network's own tensors: the dc6 dgrad -> dc5 gate/InstanceNorm
    backward -> dc5 wgrad/dgrad chain, fed with the float64 oracle's tensors, must agree to 1e-5 (measured ~5e-7)."""
    import torch.nn.functional as F
    from seunet_amd import ops as S
    o = orc.build_oracle(2, 1, 1, seed=0).double()
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
    cap = {}

    def mk(name):
        def hook(mod, inp, out):
            cap[name + ".x"], cap[name + ".raw"] = inp[0], out
            out.retain_grad()
        return hook
    hooks = [getattr(o, n).conv1.register_forward_hook(mk(n)) for n in ("dc5", "dc6")]
    pe, pd = o(b["image"].double())
    pd.retain_grad()
    orc.stage_loss(1, pe, pd, b["label"].double()).backward()
    for h in hooks:
        h.remove()
    rel = lambda a, r: float((a.detach().cpu().double() - r).norm() / r.norm())
    gl = pd.grad.float().reshape(2, 32, 32, 32).contiguous().cuda()
    x5, raw5, draw5, draw6 = cap["dc5.x"].detach(), cap["dc5.raw"].detach(), cap["dc5.raw"].grad, cap["dc6.raw"].grad
    (g,), _, _ = S.conv3d([S.to_cl(draw6.float().cuda(), "fp32")], o.dc6.conv1.weight.detach().float().cuda(), None, 1, 0,
                          transpose_flip=True)
    xx = cap["dc6.x"].detach().clone().requires_grad_(True)
    F.conv3d(xx, o.dc6.conv1.weight.detach(), padding=1).backward(draw6)
    assert rel(S.from_cl(g), xx.grad) < 1e-5
    w = {k: v.detach().float().cuda() for k, v in o.dc5.named_parameters()}
    rawc = S.to_cl(raw5.float().cuda(), "fp32")
    part, slots = S.channel_stats(rawc)
    mean, rstd = S.stats_finalize(part, slots, 32 ** 3)
    out = S.gate_epilogue_bwd(rawc, mean, rstd, w["conv_se.weight"], None, w["conv2.weight"], w["conv2.bias"], g_e=g, g_level=gl,
                              head_w=o.dc0_1.weight.detach().reshape(-1)[8:10].float().cuda())
    assert rel(S.from_cl(out["draw"]), draw5) < 1e-5
    assert rel(out["dw_se"], o.dc5.conv_se.weight.grad.reshape(-1)) < 1e-5
    srcs = [S.to_cl(x5[:, :32].float().cuda(), "fp32"), S.to_cl(x5[:, 32:].float().cuda(), "fp32")]
    assert rel(S.conv3d_wgrad(srcs, out["draw"], 64, 32, 27, 1, 0), o.dc5.conv1.weight.grad) < 1e-5


def test_reference_style_step_api_fp32(A, orc):
    """The train.py:594-603 body verbatim: sigmoid -> dice_loss x2 -> backward -> AdamW step."""
    m = build(A, orc, 2, "fp32", train=False)
    o = orc.build_oracle(2, 1, 1, seed=0)
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=7)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    opt_o = torch.optim.AdamW([p for n, p in o.named_parameters()], lr=1e-4)
    data, label = b["image"].cuda(), b["label"].cuda()
    pred_en, pred_de = m(data)
    pred_en, pred_de = torch.sigmoid(pred_en), torch.sigmoid(pred_de)
    loss = A.dice_loss(pred_de, label) + A.dice_loss(pred_en, label)
    opt.zero_grad()
    loss.backward()
    opt.step()
    pe, pd = o(b["image"])
    l_ref = orc.dice_loss(torch.sigmoid(pd), b["label"]) + orc.dice_loss(torch.sigmoid(pe), b["label"])
    opt_o.zero_grad()
    l_ref.backward()
    opt_o.step()
    assert abs(float(loss.detach()) - float(l_ref.detach())) < 1e-4
    sd, so = m.state_dict(), o.state_dict()
    assert list(sd.keys()) == list(so.keys())
    for k in sd:                                                      # AdamW's sign-like first step: compare loosely
        assert float((sd[k].cpu() - so[k]).abs().max()) < 2.5e-4, k
    assert torch.equal(sd["dc62.conv1.weight"].cpu(), orc.deterministic_state_dict(2, 1, 1, 0)["dc62.conv1.weight"])


def test_width_mult_2_forward_backward_fp32(A, orc):
    """BASELINE configs[4] uses 2x channel width (a build-side extension; oracle = this repo's restatement)."""
    m = A.SE_UNet(2, 1, width_mult=2, act_dtype="fp32")
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 2, seed=0))
    m = m.cuda().eval()
    o = orc.build_oracle(2, 1, 2, seed=0).double()
    b = orc.synthetic_batch(1, (32, 32, 32), 2, seed=13)
    pe, pd = o(b["image"].double())
    l64 = orc.stage_loss(1, pe, pd, b["label"].double())
    l64.backward()
    ge, gd = m(b["image"].cuda())
    assert float((gd.detach().cpu().double() - pd.detach()).abs().max()) < 1e-4
    assert float((ge.detach().cpu().double() - pe.detach()).abs().max()) < 1e-4
    loss = A.fused_stage_loss(1, ge, gd, b["label"].cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(l64.detach())) < 1e-5
    o32 = orc.build_oracle(2, 1, 2, seed=0)
    qe, qd = o32(b["image"])
    orc.stage_loss(1, qe, qd, b["label"]).backward()
    _check_grad_noise(_rel_errors(m, o), _rel_errors(o32, o))
    _check_vs_same_choice_f64(orc, m, b, 1, width_mult=2, what="width x2:")


def test_noncontiguous_input_and_determinism_fp32(A, orc):
    m = build(A, orc, 2, "fp32")
    big = orc.synthetic_batch(1, (40, 40, 40), 2, seed=8)["image"].cuda()
    view = big[:, :, 4:36, 8:40, 0:32]                                  # strided window
    with torch.no_grad():
        a = m(view)[1]
        b = m(view.contiguous())[1]
        c = m(view)[1]
    assert torch.equal(a, b) and torch.equal(a, c)


def test_cpu_tensor_raises(A, orc):
    m = A.SE_UNet(2, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 16, 16, 16))


def test_bf16_mode_no_worse_than_bf16_autocast(A, orc):
    """bf16 activation storage is the benchmark dtype (BASELINE.json configs[1]).  Stated tolerance: against the
    float64 oracle the HIP bf16 path must be at least as accurate as PyTorch's own bf16 autocast of the oracle
    network (logits, loss, and every large gradient tensor within 1.25x of autocast's error).  Measured at
    2x32^3: logits 2.2e-2 (autocast 2.9e-2), gradients 19-52 % (autocast 21-59 %): the Dice gradient through 28
    InstanceNorm layers is ill-conditioned, 8-bit mantissas show it in ANY bf16 implementation."""
    m = build(A, orc, 2, "bf16")
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
    o64, l64 = _oracle64(orc, 1, b)
    with torch.no_grad():
        p64 = o64(b["image"].double())[1]
    oa = orc.build_oracle(2, 1, 1, seed=0)
    with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
        ae, ad = oa(b["image"])
    la = orc.stage_loss(1, ae.float(), ad.float(), b["label"])
    la.backward()
    c = {k: v.cuda() for k, v in b.items()}
    ge, gd = m(c["image"])
    loss = A.fused_stage_loss(1, ge, gd, c["label"])
    loss.backward()
    e_hip = float((gd.detach().cpu().double() - p64).abs().max())
    e_amp = float((ad.detach().double() - p64).abs().max())
    print(f"bf16 logits: HIP {e_hip:.3e} autocast {e_amp:.3e}")
    assert e_hip <= 1.25 * e_amp and e_hip < 5e-2
    assert abs(float(loss.detach()) - l64) <= max(2 * abs(float(la.detach()) - l64), 1e-4)
    worse = []
    for (name, p), (_, q), (_, r) in zip(m.named_parameters(), oa.named_parameters(), o64.named_parameters()):
        if r.grad is None or r.numel() < 4096:
            continue
        eh = float((p.grad.cpu().double() - r.grad).norm() / r.grad.norm())
        ea = float((q.grad.double() - r.grad).norm() / r.grad.norm())
        if eh > 1.25 * ea + 0.02:
            worse.append((name, eh, ea))
    assert not worse, worse


@pytest.mark.parametrize("batch,size", [(2, 32), (1, 64)])
@pytest.mark.parametrize("dtype,med_bar,max_bar,dc5_bar", [("bf16", 1e-1, 1.5e-1, 5e-2), ("fp16", 1.5e-2, 2.5e-2, 1e-2)])
def test_16bit_modes_against_same_choice_float64(A, orc, dtype, med_bar, max_bar, dc5_bar, batch, size):
    """The 16-bit storage modes against FLOAT64 (not against another 16-bit implementation): with the LeakyReLU-sign and max-pool
    arg-max choices of the mode's own forward imposed on the float64 oracle (tests/forced_oracle.py), what is left is the
    arithmetic / storage error of the implementation.  Against the PLAIN float64 oracle a bf16 forward is 35 % off on the large
    gradient tensors (76,572 of 12 M signs and 3,230 of 336 k arg-maxes differ at 2 x 32^3) -- PyTorch's bf16 autocast of the
    oracle is 45 % off for the same reason (profiles/r03_lowprec_attribution_32.md); with the choices imposed: bf16 median 5.3e-2
    / max 8.5e-2 over the large tensors (dc5.conv1.weight 3.2e-2), fp16 6.4e-3 / 1.0e-2 (3.9e-3): the error is the 8- (11-) bit
    rounding of the stored activations and activation gradients, amplified ~13 x by the InstanceNorm backward's cancellation.
    Bars = ~2 x measured.  The 1 x 64^3 case runs the full-resolution level on the marching kernels (conv_march.hip,
    wgrad_march.hip: levels of >= 48^3 voxels with rows of >= 32), the 2 x 32^3 case on the tiled / streaming ones."""
    import forced_oracle as FO
    m = build(A, orc, 2, dtype)
    b = orc.synthetic_batch(batch, (size,) * 3, 2, seed=3)
    _, _, inter = m.forward_with_intermediates(b["image"].cuda(), FO.LRELU_ORDER)
    ge, gd = m(b["image"].cuda())
    A.fused_stage_loss(1, ge, gd, b["label"].cuda()).backward()
    signs, pools = FO.path_choices(inter)
    of, _, fd, lf, nsf, npf = FO.forced_step(orc, b, 1, signs, pools)
    big = {}
    for (name, p), (_, q) in zip(m.named_parameters(), of.named_parameters()):
        if q.grad is not None and q.numel() >= 4096:
            big[name] = float((p.grad.cpu().double() - q.grad).norm() / q.grad.norm())
    v = np.array(list(big.values()))
    lerr = float((gd.detach().cpu().double() - fd).abs().max())
    print(f"{dtype}: {nsf} sign / {npf} arg-max choices differ from float64's; logits vs same-choice f64 {lerr:.2e}; large-tensor "
          f"gradient rel-L2 median {np.median(v):.2e} max {v.max():.2e}; dc5.conv1.weight {big['dc5.conv1.weight']:.2e}")
    assert float(np.median(v)) <= med_bar and float(v.max()) <= max_bar and big["dc5.conv1.weight"] <= dc5_bar, big
    assert lerr <= (3e-2 if dtype == "bf16" else 4e-3)


def test_bf16_mfma_matches_naive(A, orc):
    x = orc.synthetic_batch(1, (32, 32, 32), 2, seed=9)["image"].cuda()
    with torch.no_grad():
        a = build(A, orc, 2, "bf16", 0)(x)[1]
        b = build(A, orc, 2, "bf16", 1)(x)[1]
    assert float((a - b).abs().max()) < 5e-2


def test_sliding_window_matches_oracle_assembly(A, orc):
    """prediction.py:78-109 loop: window positions, overlap averaging (float64)."""
    m = build(A, orc, 2, "fp32")
    o = orc.build_oracle(2, 1, 1, seed=0)
    x = orc.synthetic_batch(1, (40, 32, 48), 2, seed=10)["image"]
    got = A.sliding_window_predict(m, x.cuda(), cube=32, step=16, batch=2)
    ref = orc.sliding_window_predict(o, x, cube=32, step=16)
    assert got.shape == ref.shape == (40, 32, 48)
    assert float(np.abs(got - ref).max()) < FP32_ATOL


@pytest.mark.parametrize("train", [False, True])
def test_sliding_window_raw_logit_variant_matches_oracle(A, orc, train):
    """save_gradients.py:129-137 / weight_br.py:95-102: the same loop accumulating the decoder head's RAW logits (no sigmoid),
    averaged and thresholded at 0.5, run under ``case_net.train()`` (DropLayer active, one window per call, one CPU-generator
    draw per window in both implementations)."""
    m = build(A, orc, 2, "fp32", train=train)
    o = orc.build_oracle(2, 1, 1, seed=0, train=train)
    x = orc.synthetic_batch(1, (40, 32, 48), 2, seed=11)["image"]
    torch.manual_seed(91)
    got = A.sliding_window_predict(m, x.cuda(), cube=32, step=16, batch=None if train else 2, sigmoid=False)
    torch.manual_seed(91)
    ref = orc.sliding_window_predict(o, x, cube=32, step=16, sigmoid=False)
    assert got.shape == ref.shape == (40, 32, 48)
    assert float(np.abs(got - ref).max()) < 3e-3 and float(ref.min()) < 0.0               # (logits, not probabilities)
    with_sig = A.sliding_window_predict(m.eval(), x.cuda(), cube=32, step=16, batch=2)
    assert float(with_sig.min()) > 0.0 and float(got.min()) < 0.0
    near = np.abs(ref - 0.5) < 1e-2                                                    # the scripts' threshold of the average
    assert np.array_equal((got >= 0.5)[~near], (ref >= 0.5)[~near])


def test_block_modules_standalone(A, orc):
    os.environ["SEUNET_DTYPE"] = "fp32"
    try:
        o = orc.build_oracle(2, 1, 1, seed=0)
        m = build(A, orc, 2, "fp32")
        x = orc.synthetic_batch(1, (16, 16, 16), 2, seed=11)["image"]
        e_ref, s_ref = o._gated("ec1", x)
        e, s = m.ec1(x.cuda())
        assert float((e.detach().cpu() - e_ref.detach()).abs().max()) < 1e-4 and float((s.detach().cpu() - s_ref.detach()).abs().max()) < 1e-4
        t = torch.rand(1, 56, 8, 8, 8)
        assert float((m.ec33(t.cuda()).detach().cpu() - o._cat("ec33", t).detach()).abs().max()) < 1e-4
    finally:
        os.environ.pop("SEUNET_DTYPE", None)


@pytest.mark.parametrize("name,cin,shape", [("ec1", 2, (16, 16, 16)), ("ec5", 32, (8, 16, 16)), ("dc2", 64, (8, 8, 8)), ("ec33", 56, (8, 8, 16))])
def test_block_modules_are_differentiable_fp32(A, orc, name, cin, shape):
    """SSEConv / SSEConv2 / CATConv used on their own (SE_UNet.py:9-82) back-propagate like the reference's modules: the
    gradients w.r.t. the input and every parameter of the block against the float64 oracle block, for a loss that uses
    BOTH outputs (the gated tensor and the up-sampled 2-channel side map; down_sample 1, 2 and 4, dilation 1 and 2)."""
    os.environ["SEUNET_DTYPE"] = "fp32"
    try:
        o = orc.build_oracle(2, 1, 1, seed=0).double()
        m = build(A, orc, 2, "fp32")
        g0 = torch.Generator().manual_seed(77)
        x = torch.rand((2, cin) + shape, generator=g0)
        xo = x.double().requires_grad_(True)
        xm = x.cuda().requires_grad_(True)
        blk_o, blk_m = getattr(o, name), getattr(m, name)
        if name == "ec33":
            out_o, out_m = (o._cat(name, xo),), (blk_m(xm),)
        else:
            out_o, out_m = o._gated(name, xo), blk_m(xm)
        ws = [torch.rand(t.shape, generator=g0) - 0.5 for t in out_o]
        sum((t * w.double()).sum() for t, w in zip(out_o, ws)).backward()
        sum((t * w.cuda()).sum() for t, w in zip(out_m, ws)).backward()
        for a, b in zip(out_m, out_o):
            assert float((a.detach().cpu().double() - b.detach()).abs().max()) < 1e-4
        rel = lambda a, r: float((a.detach().cpu().double() - r).norm() / (r.norm() + 1e-30))
        assert rel(xm.grad, xo.grad) < 2e-4, ("x", rel(xm.grad, xo.grad))
        for (n_, p_m), (_, p_o) in zip(blk_m.named_parameters(), blk_o.named_parameters()):
            if n_ == "conv1.bias":      # identically zero (affine-less InstanceNorm follows); the oracle holds rounding noise
                assert float(p_m.grad.abs().max()) == 0.0
                continue
            assert rel(p_m.grad, p_o.grad) < 2e-4, (n_, rel(p_m.grad, p_o.grad))
    finally:
        os.environ.pop("SEUNET_DTYPE", None)


@pytest.mark.parametrize("dtype", ["bf16"])
def test_full_size_properties_128(A, orc, dtype):
    """BASELINE configs[1] as benchmarked (4 x 2 x 128^3, bf16): size-independent properties."""
    m = build(A, orc, 2, dtype)
    x = orc.synthetic_batch(4, (128, 128, 128), 2, seed=12)["image"].cuda()
    with torch.no_grad():
        p0, p1 = m(x)
        q0, q1 = m(x[1:2])                       # InstanceNorm network: samples are independent
        r0, r1 = m(x)
        assert torch.equal(p1[3:4], m(x[3:4])[1])
    assert p0.shape == p1.shape == (4, 1, 128, 128, 128)
    assert torch.isfinite(p0).all() and torch.isfinite(p1).all()
    assert torch.equal(p1, r1) and torch.equal(p0, r0)                # deterministic
    assert torch.equal(p1[1:2], q1) and torch.equal(p0[1:2], q0)       # batch independence (eval mode)
    label = (torch.rand(4, 1, 128, 128, 128, device="cuda") < 0.03).float()
    e, d = m(x)
    A.fused_stage_loss(1, e, d, label).backward()
    gs = [p.grad for n, p in m.named_parameters() if not n.startswith("dc62.")]
    assert all(g is not None and torch.isfinite(g).all() for g in gs)
    assert m.dc62.conv1.weight.grad is None


def test_three_channel_input_takes_the_materialised_x_branch_fp32(A, orc):
    """in_channel = 3: the x-branches (x33 / x63 / x93) are not recomputed from the input (that path handles <= 2
    channels) but run as 1x1x1 convolutions with their own MFMA weight-gradient launch.  Forward and gradients against
    the oracle on the same weights."""
    torch.manual_seed(0)
    o = orc.build_oracle(3, 1, 1, seed=0)
    m = A.SE_UNet(3, 1, act_dtype="fp32", conv_impl=0)
    m.load_state_dict(orc.deterministic_state_dict(3, 1, 1, 0))
    m = m.cuda().eval()
    b = orc.synthetic_batch(1, (32, 32, 32), 3, seed=9)
    pe, pd = o(b["image"])
    orc.stage_loss(1, pe, pd, b["label"]).backward()
    e, d = m(b["image"].cuda())
    assert float((d.detach().cpu() - pd.detach()).abs().max()) < 1e-4 and float((e.detach().cpu() - pe.detach()).abs().max()) < 1e-4
    A.fused_stage_loss(1, e, d, b["label"].cuda()).backward()
    for name in ("x33.conv1.weight", "x63.conv1.weight", "x93.conv1.weight", "ec33.conv1.weight"):
        gq, gp = dict(o.named_parameters())[name].grad, dict(m.named_parameters())[name].grad.cpu()
        e = float((gp - gq).norm() / gq.norm())
        print(f"in_channel=3 {name}: rel-L2 vs fp32 oracle {e:.2e} (two fp32 paths: flip noise, informational)")
    # the gate: float64 with this forward's own choices imposed (here the x-branches are stored tensors)
    # (this path accumulates the x-branch weight gradient from the stored f32 draw like every plain fp32 evaluation: 1e-4-class)
    _check_vs_same_choice_f64(orc, m, b, 1, what="in_channel=3:", xtol=1e-3)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_gradients_are_bitwise_reproducible(A, orc, dtype):
    """Every reduction on the path (InstanceNorm partial sums, parameter-gradient records, weight-gradient slabs, loss
    sums) is summed in a fixed order and nothing uses atomics: two runs of the same step give identical bits."""
    b = orc.synthetic_batch(2, (64, 64, 64), 2, seed=11)
    x, lab = b["image"].cuda(), b["label"].cuda()
    runs = []
    for _ in range(2):
        m = build(A, orc, 2, dtype)
        e, d = m(x)
        loss = A.fused_stage_loss(1, e, d, lab)
        loss.backward()
        runs.append((loss.detach().clone(), d.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][2].keys() == runs[1][2].keys() and len(runs[0][2]) == 116
    for n in runs[0][2]:
        assert torch.equal(runs[0][2][n], runs[1][2][n]), n


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("batch,size", [(2, (32, 32, 32)), (1, (64, 64, 64)), (1, (40, 48, 56))])
def test_results_do_not_depend_on_the_prior_contents_of_the_workspace(A, orc, dtype, batch, size, monkeypatch):
    """The caller owns the workspace and hands it over uninitialised (torch.empty: whatever an earlier step left there).  A step over
    a workspace, outputs and gradient buffer pre-filled with 0xFF bytes (NaN patterns in bf16 / fp16 / f32 / f64) must give the very
    bits of a step over zero-filled ones: the library reads no byte that it has not written in the same pass."""
    import importlib
    host = importlib.import_module("seunet_amd.SE_UNet")      # (the module; the package attribute of that name is the class)
    b = orc.synthetic_batch(batch, size, 2, seed=13)
    x, lab = b["image"].cuda(), b["label"].cuda()
    runs = []
    for fill in (0x00, 0xFF):
        monkeypatch.setattr(host, "_DEBUG_FILL", fill)
        m = build(A, orc, 2, dtype)
        e, d = m(x)
        loss = A.fused_stage_loss(1, e, d, lab)
        loss.backward()
        runs.append((loss.detach().clone(), e.detach().clone(), d.detach().clone(),
                     {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    monkeypatch.setattr(host, "_DEBUG_FILL", None)
    assert torch.isfinite(runs[1][0]) and torch.equal(runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    bad = [n for n in runs[0][3] if not torch.equal(runs[0][3][n], runs[1][3][n])]
    assert not bad, bad


@pytest.mark.parametrize("train", [False, True])
def test_three_classes_forward_backward_fp32(A, orc, train):
    """SE_UNet(in_channel, n_classes) with n_classes = 3 (SE_UNet.py:100,150-151: the two 1x1x1 heads map 24 / 12 side channels to
    n_classes logits).  No reference caller uses it; the HIP path runs the heads on its general form (csrc/classes.hip).  Logits
    (B, 3, D, H, W) against the oracle, loss (Dice over all classes, the reference's flattening ratio) and every gradient against
    float64 with the same discrete choices -- eval mode and train mode (DropLayer scales injected)."""
    import forced_oracle as FO
    K = 3
    m = A.SE_UNet(in_channel=2, n_classes=K, act_dtype="fp32")
    m.load_state_dict(orc.deterministic_state_dict(2, K, 1, seed=0))
    m = m.cuda().train(train)
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=31)
    g = torch.Generator().manual_seed(7)
    label = (torch.rand(2, K, 32, 32, 32, generator=g) < 0.05).float()
    drops = None
    if train:
        drops = (orc.drop_scale_from_uniform(torch.rand(2, 24, 1, 1, 1, generator=g), 24),
                 orc.drop_scale_from_uniform(torch.rand(2, 12, 1, 1, 1, generator=g), 12))
    o = orc.build_oracle(2, K, 1, seed=0, train=train)
    with torch.no_grad():
        pe, pd = o(b["image"], *drops) if train else o(b["image"])
    ge, gd = m(b["image"].cuda(), drop_scales=drops) if train else m(b["image"].cuda())
    assert tuple(ge.shape) == (2, K, 32, 32, 32) == tuple(gd.shape)
    assert float((ge.detach().cpu() - pe).abs().max()) < 1e-4 and float((gd.detach().cpu() - pd).abs().max()) < 1e-4
    loss = A.fused_stage_loss(1, ge, gd, label.cuda())
    loss.backward()
    m.eval()
    _, _, inter = m.forward_with_intermediates(b["image"].cuda(), FO.LRELU_ORDER)
    m.train(train)
    signs, pools = FO.path_choices(inter)
    bb = dict(b, label=label, weight=torch.ones_like(label), skel=torch.zeros_like(label))
    of, fe, fd, lf, nsf, npf = FO.forced_step(orc, bb, 1, signs, pools, drops=drops, n_classes=K)
    assert abs(float(loss.detach()) - lf) < 1e-5
    err = _rel_errors(m, of)
    v = np.array(list(err.values()))
    print(f"n_classes=3 train={train}: gradient rel-L2 vs same-choice float64 ({nsf} / {npf} choices differ): median {np.median(v):.2e} max {v.max():.2e}")
    bad = {k: e for k, e in err.items() if e > 3e-5}
    assert not bad, bad
    assert tuple(m.dc0_0.weight.grad.shape) == (K, 24, 1, 1, 1) and tuple(m.dc0_1.bias.grad.shape) == (K,)
    with pytest.raises(NotImplementedError):
        A.sliding_window_predict(m.eval(), torch.zeros(1, 2, 32, 32, 32, device="cuda"), cube=32, step=16)


@pytest.mark.parametrize("dtype,reps", [("fp32", 300), ("bf16", 400)])
def test_step_is_bitwise_stable_while_another_process_shares_the_gpu(dtype, reps):
    """tests/stress_shared_gpu.py: one process repeats a 1 x 2 x 32^3 step while a second one runs steps of another shape on the same
    GPU; every repetition must reproduce the first bit for bit.  Before round 4's fix (cross-lane results settled before any change
    of EXEC) 5-8 % of the repetitions differed under exactly these conditions -- the conditions of the two-rank data-parallel tests."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_shared_gpu.py")
    r = subprocess.run([sys.executable, script, dtype, "1", "32", str(reps)], capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and f"{reps} repetitions, 0 differ from the first" in r.stdout, tail


# ---------------------------------------------------------------------------------------------------------------
# round 2: train-mode backward, stage 2, ragged tiles (40^3 / 160^3), bf16 trainability, data-parallel equivalence
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stage", [1, 2])
def test_train_mode_forward_backward_injected_drop_fp32(A, orc, stage):
    """DropLayer ACTIVE (every reference training / validation caller runs model.train(): train.py:578,632) with the
    two scale tensors injected into both implementations: logits, loss and all gradients against the float64 oracle
    ('enabled with injected mask for one train-mode parity case').  Stage 2 = GUL on both heads
    (train.py:428-435)."""
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=21)
    g = torch.Generator().manual_seed(99)
    d1 = orc.drop_scale_from_uniform(torch.rand(2, 24, 1, 1, 1, generator=g), 24)
    d2 = orc.drop_scale_from_uniform(torch.rand(2, 12, 1, 1, 1, generator=g), 12)
    assert int((d1 == 0).sum()) > 0 and int((d2 == 0).sum()) > 0          # some side maps really are dropped
    o64 = orc.build_oracle(2, 1, 1, seed=0, train=True).double()
    pe, pd = o64(b["image"].double(), d1.double(), d2.double())
    l64 = orc.stage_loss(stage, pe, pd, b["label"].double(), b["weight"].double(), b["skel"].double())
    l64.backward()
    o32 = orc.build_oracle(2, 1, 1, seed=0, train=True)
    qe, qd = o32(b["image"], d1, d2)
    orc.stage_loss(stage, qe, qd, b["label"], b["weight"], b["skel"]).backward()
    m = build(A, orc, 2, "fp32", train=True)
    c = {k: v.cuda() for k, v in b.items()}
    ge, gd = m(c["image"], drop_scales=(d1, d2))
    assert float((gd.detach().cpu().double() - pd.detach()).abs().max()) < 1e-4
    assert float((ge.detach().cpu().double() - pe.detach()).abs().max()) < 1e-4
    loss = A.fused_stage_loss(stage, ge, gd, c["label"], c["weight"], c["skel"])
    loss.backward()
    assert abs(float(loss.detach()) - float(l64.detach())) < 1e-5
    # Gradients: against float64 with this forward's own discrete choices AND the same injected drop scales, every tensor to
    # rounding (measured 8.7e-6 worst).  The distance from the plain float64 run is printed only: it measures where the handful of
    # differing LeakyReLU signs fell (round 4: four of 15 million moved dc42.conv1.weight to 6.4e-3, with every tensor within 9e-6 of
    # float64 under the same choices).
    err, ref = _rel_errors(m, o64), _rel_errors(o32, o64)
    print("train mode, plain float64 (flip noise): HIP median %.2e max %.2e | fp32 oracle median %.2e max %.2e" %
          (np.median(list(err.values())), max(err.values()), np.median(list(ref.values())), max(ref.values())))
    was_training = m.training
    _check_vs_same_choice_f64(orc, m.eval(), b, stage, what=f"train mode stage {stage}:", drops=(d1, d2))   # (eval(): the diagnostic forward reads block tensors, which do not depend on DropLayer)
    m.train(was_training)
    # a dropped side map contributes nothing: the head weight of a channel dropped in EVERY sample gets a zero gradient
    dead1 = (d1.reshape(2, 24) == 0).all(0)
    if bool(dead1.any()):
        assert float(m.dc0_0.weight.grad.reshape(-1)[dead1.cuda()].abs().max()) == 0.0


@pytest.mark.parametrize("width", [1, 2])
def test_ragged_tiles_40_forward_backward_fp32(A, orc, width):
    """1 x 2 x 40^3: levels 40 / 20 / 10 / 5 -- no extent is a multiple of the 32-voxel x-tile or the 4-voxel y/z
    tile at the deeper levels (the small analogue of BASELINE configs[4]'s 160^3 = 160 / 80 / 40 / 20)."""
    m = A.SE_UNet(2, 1, width_mult=width, act_dtype="fp32")
    m.load_state_dict(orc.deterministic_state_dict(2, 1, width, seed=0))
    m = m.cuda().eval()
    o64 = orc.build_oracle(2, 1, width, seed=0).double()
    o32 = orc.build_oracle(2, 1, width, seed=0)
    b = orc.synthetic_batch(1, (40, 40, 40), 2, seed=31)
    pe, pd = o64(b["image"].double())
    l64 = orc.stage_loss(1, pe, pd, b["label"].double())
    l64.backward()
    qe, qd = o32(b["image"])
    orc.stage_loss(1, qe, qd, b["label"]).backward()
    ge, gd = m(b["image"].cuda())
    assert float((gd.detach().cpu().double() - pd.detach()).abs().max()) < 1e-4
    assert float((ge.detach().cpu().double() - pe.detach()).abs().max()) < 1e-4
    loss = A.fused_stage_loss(1, ge, gd, b["label"].cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(l64.detach())) < 1e-5
    _check_grad_noise(_rel_errors(m, o64), _rel_errors(o32, o64))
    _check_vs_same_choice_f64(orc, m, b, 1, width_mult=width, what=f"40^3 width x{width}:")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_config4_shape_160_width2_properties(A, orc, dtype):
    """BASELINE configs[4]: 2x channel width, 160^3 (levels 160/80/40/20), 2-byte activations: size-independent
    properties (finite, deterministic, batch independent, gradients for all 116 live tensors), plus agreement of the two
    2-byte storage modes with each other at the level their mantissas allow."""
    m = A.SE_UNet(2, 1, width_mult=2, act_dtype=dtype)
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 2, seed=0))
    m = m.cuda().eval()
    b = orc.synthetic_batch(2, (160, 160, 160), 2, seed=41)
    x, lab = b["image"].cuda(), b["label"].cuda()
    with torch.no_grad():
        p0, p1 = m(x)
        q0, q1 = m(x[1:2])
    assert p1.shape == (2, 1, 160, 160, 160) and torch.isfinite(p0).all() and torch.isfinite(p1).all()
    assert torch.equal(p1[1:2], q1) and torch.equal(p0[1:2], q0)
    e, d = m(x)
    assert torch.equal(d.detach(), p1)
    A.fused_stage_loss(1, e, d, lab).backward()
    gs = [p.grad for n, p in m.named_parameters() if not n.startswith("dc62.")]
    assert len(gs) == 116 and all(g is not None and torch.isfinite(g).all() for g in gs)


def test_bf16_mode_trains_like_fp32_mode(A, orc):
    """bf16 activation storage is the benchmark dtype; its per-step gradients differ from float64 by tens of percent on
    the ill-conditioned tensors (test_bf16_mode_no_worse_than_bf16_autocast), so check what matters: 40 AdamW steps
    (train.py:569: AdamW, here lr 1e-3 so that 40 steps move the loss) on a fixed learnable batch give the same loss
    curve in bf16 mode as in fp32 mode.  Measured on MI355X: both fall 1.887 -> 1.783, max |bf16 - fp32| over the 40 steps
    8e-4.  Band: |bf16 - fp32| <= 5e-3 at every step, and the loss must fall by at least 0.08 in both modes."""
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 2, 64, 64, 64, generator=g)
    label = (img[:, 0:1] > 0.97).float()              # learnable from the image; ~3 % foreground like an airway mask
    x, lab = img.cuda(), label.cuda()
    curves = {}
    for dtype in ("fp32", "bf16"):
        m = build(A, orc, 2, dtype)
        opt = A.AdamW(m.parameters(), lr=1e-3)
        losses = []
        for _ in range(40):
            opt.zero_grad(set_to_none=True)
            e, d = m(x)
            loss = A.fused_stage_loss(1, e, d, lab)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        curves[dtype] = np.array(losses)
    f, h = curves["fp32"], curves["bf16"]
    print("loss curve fp32:", np.round(f[::5], 4), "bf16:", np.round(h[::5], 4), "max |diff| %.4f" % np.abs(f - h).max())
    assert np.isfinite(f).all() and np.isfinite(h).all()
    assert f[-1] < f[0] - 0.08 and h[-1] < h[0] - 0.08
    assert np.abs(f - h).max() <= 5e-3


def _dp_worker(rank, world, port, backend, q, overlap=False, dtype="fp32", poison=False):
    import os as _os
    import sys as _sys
    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                       LOCAL_RANK=str(rank if backend == "nccl" else 0), HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    for p_ in (root, _os.path.join(root, "oracle")):
        if p_ not in _sys.path:
            _sys.path.insert(0, p_)
    import torch as _t
    import torch.distributed as _dist
    import seunet_amd as _A
    import seunet_oracle as _orc
    from seunet_amd import ddp as _ddp
    try:
        _ddp.init_from_env(backend)
        dev = _t.device("cuda", rank if backend == "nccl" else 0)
        _t.cuda.set_device(dev)
        b = _orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
        m = _A.SE_UNet(2, 1, act_dtype=dtype)
        m.load_state_dict(_orc.deterministic_state_dict(2, 1, 1, seed=0))
        m = m.to(dev).eval()
        x, lab = b["image"][rank:rank + 1].to(dev), b["label"][rank:rank + 1].to(dev)
        if overlap:      # the exchange runs inside backward(): decoder bucket on a side stream, the rest after the backward
            sync_cls = _ddp.GradSync
            if poison:   # ONE rank's buffer holds an inf before the exchange (a scaled fp16 gradient that left half precision's range)
                class sync_cls(_ddp.GradSync):
                    def exchange(self, flat, split, ev):
                        if rank == 1:
                            flat[:1].fill_(float("inf"))
                        super().exchange(flat, split, ev)
            m.grad_sync = sync_cls(timing=True)
            assert m.grad_sync.decoder_event().cuda_event != 0    # (a torch event has no handle before its first record)
        e, d = m(x)
        loss = _A.fused_stage_loss(1, e, d, lab, group=True)      # global-batch ratio: sums all-reduced first
        if overlap:
            # put the launch stream ~100 ms behind the host: a side stream that does NOT wait for the decoder-boundary event then
            # reduces gradient memory the backward pass has not written yet, and the comparison below fails
            _t.cuda._sleep(200_000_000)
        loss.backward()
        grads = [p.grad for p in m.parameters() if p.grad is not None]
        zero_copy = _ddp._flat_view(grads) is not None            # real backward -> one contiguous bucket
        n = sum(g.numel() for g in grads) if overlap else _ddp.allreduce_gradients(m.parameters())
        if overlap:
            assert len(m.grad_sync.elapsed_ms()) == 1
        out = {k: p.grad.cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}   # by value (the worker exits)
        if poison:
            out = {"overflow_steps": int(m.overflow_steps), "max_abs": max(float(np.abs(v).max()) for v in out.values())}
        q.put((rank, float(loss.detach()), bool(zero_copy), int(n), out if (rank == 0 or poison) else None))
        _dist.barrier()
        _dist.destroy_process_group()
    except Exception as ex:     # report instead of hanging the parent
        q.put((rank, repr(ex), False, 0, None))


def _spawn_dp(backend, overlap, dtype="fp32", poison=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + (7 if backend == "nccl" else 0) + (13 if overlap else 0) + {"fp32": 0, "bf16": 29, "fp16": 53}[dtype] \
        + (101 if poison else 0)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, backend, q, overlap, dtype, poison)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
    assert all(isinstance(r[1], float) for r in res), res
    return res


def _run_dp_equivalence(A, orc, backend, overlap=False, dtype="fp32", tol=1e-4, loss_tol=1e-6):
    res = _spawn_dp(backend, overlap, dtype)
    # single process, batch 2
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
    m = build(A, orc, 2, dtype)
    e, d = m(b["image"].cuda())
    loss = A.fused_stage_loss(1, e, d, b["label"].cuda())
    loss.backward()
    assert abs(res[0][1] - float(loss.detach())) < loss_tol and abs(res[1][1] - float(loss.detach())) < loss_tol
    assert res[0][2] and res[1][2], "gradients of a real backward are not one contiguous bucket"
    assert res[0][3] == 1_520_314 - 768
    worst, worst_name = 0.0, ""
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        ref, got = p.grad.cpu().double(), torch.from_numpy(res[0][4][k]).double()
        if k.endswith("conv1.bias"):
            assert float(got.abs().max()) <= 1e-6
            continue
        e = float((got - ref).norm() / max(float(ref.norm()), 1e-30))
        if e > worst:
            worst, worst_name = e, k
    print(f"1 GPU x B=2 vs 2 ranks x B=1 ({backend}, {dtype}, {'overlapped' if overlap else 'serial'}): worst gradient rel-L2 {worst:.2e} ({worst_name})")
    assert worst < tol


def test_data_parallel_two_ranks_equal_one_rank_batch2_gloo_shared_gpu(A, orc):
    """SURVEY 8(e) exact-sum form on hardware that has ONE GPU: two ranks share cuda:0 and exchange over gloo
    (loss sums all-reduced before the ratio, train.py:53-57; parameter gradients SUMMED in place in the flat bucket).
    Must equal the single-process batch-2 step.  Also asserts the zero-copy bucket on a real backward."""
    _run_dp_equivalence(A, orc, "gloo")


def test_data_parallel_overlapped_exchange_equals_one_rank_batch2_gloo_shared_gpu(A, orc):
    """The same equivalence with the exchange overlapped with the backward pass (ddp.GradSync: the library records an event when
    the decoder's gradients are final, that tail of the flat buffer is reduced on a side stream while the encoder is still being
    differentiated, the head afterwards; train.py:577's DataParallel reduce)."""
    _run_dp_equivalence(A, orc, "gloo", overlap=True)


# 16-bit storage modes (BASELINE configs[2] is bf16, configs[4] fp16).  The sample-wise kernels give a sample the same bits whatever
# the batch it sits in (at this size the InstanceNorm partial counts do not depend on the batch either), so 1 x B=2 and 2 x B=1 differ
# only in the order of the weight-gradient sums: measured on MI355X 1.1e-7 (bf16) and 9.6e-8 (fp16), serial and overlapped alike.
# The bound is 100 x that: one discrete choice taken differently on one rank costs 1e-3 (which is what this test showed once, 2.5e-3,
# before the cross-lane / EXEC fix of round 4 -- two ranks sharing a GPU are exactly the condition that exposed it), a missing or
# doubled exchange costs ~1.
DP_16BIT = [("bf16", 1e-5), ("fp16", 1e-5)]


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("dtype,tol", DP_16BIT)
def test_data_parallel_16bit_modes_equal_one_rank_batch2_gloo_shared_gpu(A, orc, dtype, tol, overlap):
    """train.py:577's DataParallel reduce in the benchmark dtypes: serial (``allreduce_gradients``) and overlapped
    (``GradSync``, which with fp16's loss scale reduces the SCALED buffer and tests finiteness on the global sum)."""
    _run_dp_equivalence(A, orc, "gloo", overlap=overlap, dtype=dtype, tol=tol, loss_tol=2e-3)


def test_data_parallel_fp16_overflow_on_one_rank_skips_the_step_on_all_ranks(A, orc):
    """fp16 storage carries gradients times a static loss scale; a backward whose scaled gradients overflow is dropped.  Under data
    parallelism the decision must be global or the ranks' weights diverge: the exchange sums the scaled buffers FIRST, so one
    rank's inf is every rank's inf.  Rank 1 plants an inf in its buffer just before the exchange: both ranks must end with
    all-zero gradients and overflow_steps == 1."""
    res = _spawn_dp("gloo", True, "fp16", poison=True)
    for r in res:
        assert r[4] == {"overflow_steps": 1, "max_abs": 0.0}, res


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4)] + DP_16BIT)
def test_data_parallel_two_gpus_equal_one_gpu_batch2_rccl(A, orc, overlap, dtype, tol):
    """The same equivalence over RCCL with one GPU per rank (skipped on a single-GPU box)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_dp_equivalence(A, orc, "nccl", overlap=overlap, dtype=dtype, tol=tol, loss_tol=1e-6 if dtype == "fp32" else 2e-3)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_inference_forward_skips_encoder_head_bit_identical(A, orc, dtype):
    """prediction.py:102-103 keeps only the decoder head's logits: ``predict_logits`` (seunet_net_forward with pred0 = NULL)
    evaluates neither the encoder head nor the side convs / level maps of the twelve encoder blocks, and must return the very
    bits of ``forward(x)[1]`` -- in eval mode and, with the same CPU-generator state, under model.train() (the validation
    loops run that way, train.py:632); repeated calls reuse one workspace arena."""
    m = build(A, orc, 2, dtype)
    x = orc.synthetic_batch(2, (64, 64, 64), 2, seed=23)["image"].cuda()
    with torch.no_grad():
        q = m(x)[1]
    p = m.predict_logits(x)
    p2 = m.predict_logits(x)
    assert torch.equal(p, q) and torch.equal(p2, q) and len(m._arena) == 1
    m.train()
    torch.manual_seed(5)
    with torch.no_grad():
        qt = m(x)[1]
    torch.manual_seed(5)
    pt = m.predict_logits(x)
    assert torch.equal(pt, qt) and not torch.equal(pt, q)
    m.release_arena()
    assert not m._arena


def _window_worker(rank, world, port, q):
    import os as _os
    import sys as _sys
    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    for p_ in (root, _os.path.join(root, "oracle")):
        if p_ not in _sys.path:
            _sys.path.insert(0, p_)
    import torch as _t
    import torch.distributed as _dist
    import seunet_amd as _A
    import seunet_oracle as _orc
    from seunet_amd import ddp as _ddp
    try:
        _ddp.init_from_env("gloo")
        _t.cuda.set_device(0)
        m = _A.SE_UNet(2, 1, act_dtype="fp32")
        m.load_state_dict(_orc.deterministic_state_dict(2, 1, 1, seed=0))
        m = m.cuda().eval()
        x = _orc.synthetic_batch(1, (192, 128, 192), 2, seed=19)["image"].cuda()
        # (different batch sizes per rank, as auto_batch() may pick from each rank's free memory: the partition of the window
        # list must not depend on them)
        out = _A.sliding_window_predict(m, x, batch=1 + rank, group=True)
        q.put((rank, out if rank == 0 else None, float(out.sum())))
        _dist.barrier()
        _dist.destroy_process_group()
    except Exception as ex:
        q.put((rank, repr(ex), None))


def test_window_loop_sharded_over_two_ranks_equals_one_rank_gloo_shared_gpu(A, orc):
    """One case, its 4 windows dealt to two ranks (sharing cuda:0, exchanging over gloo), float64 accumulators all-reduced
    once before the division: every rank ends with the single-process result (float64 sums in another order: <= 1e-12)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 977) % 2000
    procs = [ctx.Process(target=_window_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
    assert all(r[2] is not None for r in res), res
    m = build(A, orc, 2, "fp32")
    x = orc.synthetic_batch(1, (192, 128, 192), 2, seed=19)["image"].cuda()
    ref = A.sliding_window_predict(m, x, batch=1)
    assert len(A.window_table((192, 128, 192))) == 4
    assert float(np.abs(res[0][1] - ref).max()) <= 1e-12
    assert abs(res[0][2] - res[1][2]) <= 1e-9 * abs(res[0][2])


def test_second_backward_and_input_grad_fail_clearly(A, orc):
    m = build(A, orc, 2, "fp32")
    x = orc.synthetic_batch(1, (16, 16, 16), 2, seed=1)["image"].cuda()
    e, d = m(x)
    (e.sum() + d.sum()).backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        (e.sum() + d.sum()).backward()
    with pytest.raises(NotImplementedError, match="input"):
        m(x.clone().requires_grad_(True))


@pytest.mark.parametrize("train", [False, True])
def test_validation_form_window_loop_matches_oracle(A, orc, train):
    """train.py:682-693 / data.py:731-773: batches of windows, table padded with copies of window 0 that are run and
    accumulated too.  12 windows of 32^3 in batches of 5 -> 3 copies of window 0.  In train mode (what the reference's validation runs,
    train.py:632) DropLayer draws from the CPU generator once per batch in both implementations."""
    m = build(A, orc, 2, "fp32", train=train)
    o = orc.build_oracle(2, 1, 1, seed=0, train=train)
    x = orc.synthetic_batch(1, (40, 48, 56), 2, seed=17)["image"]
    assert len(orc.validation_window_table((40, 48, 56), 5, 32, 16)) == 15 and len(A.window_table((40, 48, 56), 32, 16, 5)) == 15
    torch.manual_seed(77)
    got = A.sliding_window_validate(m, x.cuda(), batch=5, cube=32, step=16)
    torch.manual_seed(77)
    ref = orc.sliding_window_validate(o, x, batch=5, cube=32, step=16)
    assert got.shape == ref.shape == (40, 48, 56) and got.dtype == np.float64
    err = float(np.abs(got - ref).max())
    print(f"validation-form assembly (train={train}): max |diff| {err:.2e}")
    assert err < FP32_ATOL


def test_window_kernels_exact_bookkeeping(A):
    """gather / accumulate / finalize against numpy on integer-valued data: indices, order and counts are exact."""
    lib = A._lib.load()
    rng = np.random.default_rng(3)
    X, Y, Z, cube, C_ = 20, 16, 28, 8, 2
    vol = torch.from_numpy(rng.integers(-50, 50, (C_, X, Y, Z)).astype(np.float32)).cuda()
    pos = A.window_table((X, Y, Z), cube, 6, pad_to_batch=7)
    n_real = len(A.window_table((X, Y, Z), cube, 6))
    starts = A._lib.int_array([v for p in pos[:5] for v in p])
    out = torch.empty((5, C_, cube, cube, cube), dtype=torch.float32, device="cuda")
    A._lib.check(lib.seunet_window_gather(vol.data_ptr(), C_, X, Y, Z, cube, 5, starts, out.data_ptr(), None))
    for k, (a, b, c) in enumerate(pos[:5]):
        assert torch.equal(out[k], vol[:, a:a + cube, b:b + cube, c:c + cube])
    acc = torch.zeros((X, Y, Z), dtype=torch.float64, device="cuda")
    ref, cnt = np.zeros((X, Y, Z)), np.zeros((X, Y, Z))
    vals = torch.from_numpy(rng.integers(0, 9, (len(pos), 1, cube, cube, cube)).astype(np.float32)).cuda()
    for i in range(0, len(pos), 7):
        chunk = pos[i:i + 7]
        A._lib.check(lib.seunet_window_accumulate(vals[i:].data_ptr(), 0, len(chunk), A._lib.int_array([v for p in chunk for v in p]),
                                                  cube, acc.data_ptr(), X, Y, Z, None))
        for k, (a, b, c) in enumerate(chunk):
            ref[a:a + cube, b:b + cube, c:c + cube] += vals[i + k, 0].cpu().numpy()
            cnt[a:a + cube, b:b + cube, c:c + cube] += 1
    assert np.array_equal(acc.cpu().numpy(), ref)
    xs, ys, zs = (A.window_starts(d, cube, 6) for d in (X, Y, Z))
    fin = torch.empty_like(acc)
    A._lib.check(lib.seunet_window_finalize(acc.data_ptr(), X, Y, Z, cube, len(xs), A._lib.int_array(xs), len(ys), A._lib.int_array(ys),
                                            len(zs), A._lib.int_array(zs), len(pos) - n_real, fin.data_ptr(), None))
    assert np.array_equal(fin.cpu().numpy(), ref / cnt)
    bad = A._lib.int_array([X - cube + 1, 0, 0])
    assert lib.seunet_window_gather(vol.data_ptr(), C_, X, Y, Z, cube, 1, bad, out.data_ptr(), None) != 0
    assert "leaves" in A._lib.last_error()


def test_sliding_window_192x160x128_matches_oracle_and_512_speed(A, orc):
    """BASELINE configs[3] path (prediction.py:78-109) with the real 128^3 window / stride 64: 192 x 160 x 128 = 2x2x1
    windows against the oracle loop (fp32 mode, 1e-3), then the 512^3 volume (343 windows) in bf16: finite, in (0, 1),
    identical for batch 1 and batch 4 (eval mode: InstanceNorm is per sample), and timed."""
    import time
    m = build(A, orc, 2, "fp32")
    o = orc.build_oracle(2, 1, 1, seed=0)
    x = orc.synthetic_batch(1, (192, 160, 128), 2, seed=19)["image"]
    got = A.sliding_window_predict(m, x.cuda(), batch=2)
    ref = orc.sliding_window_predict(o, x)
    err = float(np.abs(got - ref).max())
    print(f"192x160x128 assembly vs oracle: max |diff| {err:.2e}")
    assert got.shape == (192, 160, 128) and err < FP32_ATOL
    del m
    mb = build(A, orc, 2, "bf16")
    g = torch.Generator(device="cuda").manual_seed(23)
    vol = torch.rand((1, 2, 512, 512, 512), generator=g, device="cuda")
    A.sliding_window_predict(mb, vol[:, :, :128, :128, :256], batch=1, return_tensor=True)      # warm-up
    res = {}
    for batch, graph in ((4, True), (1, True), (4, False), (1, False), (16, False), (None, False)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res[batch, graph] = A.sliding_window_predict(mb, vol, batch=batch, return_tensor=True, graph=graph)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"512^3 sliding window, 343 windows, batch {batch if batch else 'auto = %d' % A.sliding_window.auto_batch(mb, vol.device)}, {'one HIP graph per batch' if graph else 'eager launches'}: "
              f"{dt:.3f} s ({512 ** 3 / dt / 1e6:.0f} M output voxels/s)")
    assert torch.isfinite(res[4, True]).all() and float(res[4, True].min()) > 0.0 and float(res[4, True].max()) < 1.0
    for k in res:
        assert torch.equal(res[k], res[4, True]), k


def test_captured_forward_replays_match_eager_and_follow_weight_updates(A, orc):
    """CapturedForward (seunet_net_forward_capture): replays of the recorded graph give the eager forward's bits, see
    in-place weight updates, draw DropLayer scales like the eager path under model.train(), and re-record when a
    parameter tensor is replaced."""
    m = build(A, orc, 2, "bf16")
    x1 = orc.synthetic_batch(2, (64, 64, 64), 2, seed=31)["image"].cuda()
    x2 = orc.synthetic_batch(2, (64, 64, 64), 2, seed=32)["image"].cuda()
    cap = A.CapturedForward(m, 2, (64, 64, 64))
    with torch.no_grad():
        for x in (x1, x2, x1):
            cap.x.copy_(x)
            p0, p1 = cap()
            e0, e1 = m(x)
            assert torch.equal(p0, e0) and torch.equal(p1, e1)
        for p in m.parameters():            # in-place update: same pointers, new values
            p.mul_(1.01)
        cap.x.copy_(x2)
        p0, p1 = cap()
        e0, e1 = m(x2)
        assert torch.equal(p0, e0) and torch.equal(p1, e1)
        m.train()                           # DropLayer active: same CPU-generator draws as the eager call
        torch.manual_seed(5)
        p0, p1 = cap()
        p0, p1 = p0.clone(), p1.clone()
        torch.manual_seed(5)
        e0, e1 = m(x2)
        assert torch.equal(p0, e0) and torch.equal(p1, e1)
        m.eval()
        m.dc6.conv1.weight.data = m.dc6.conv1.weight.data.clone() * 0.5     # REPLACED tensor: new pointer
        p0, p1 = cap()
        e0, e1 = m(x2)
        assert torch.equal(p0, e0) and torch.equal(p1, e1)


def test_fp16_mode_at_least_as_accurate_as_bf16_mode(A, orc):
    """fp16 activation storage (BASELINE configs[4], `v_mfma_*_f16`): three more mantissa bits than bf16, so against the float64
    oracle its logits, loss and large gradient tensors must be no worse than the bf16 mode's (x1.1 + a floor); the static
    loss scale (65536) keeps the ~1e-8 activation gradients inside half precision's range: no tensor may come back zero,
    infinite or NaN."""
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
    o64, l64 = _oracle64(orc, 1, b)
    with torch.no_grad():
        p64 = o64(b["image"].double())[1]
    c = {k: v.cuda() for k, v in b.items()}
    res = {}
    for dtype in ("bf16", "fp16"):
        m = build(A, orc, 2, dtype)
        ge, gd = m(c["image"])
        loss = A.fused_stage_loss(1, ge, gd, c["label"])
        loss.backward()
        errs = {}
        for (name, p), (_, r) in zip(m.named_parameters(), o64.named_parameters()):
            if r.grad is None or r.numel() < 4096:
                continue
            assert torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, name
            errs[name] = float((p.grad.cpu().double() - r.grad).norm() / r.grad.norm())
        res[dtype] = (float((gd.detach().cpu().double() - p64).abs().max()), abs(float(loss.detach()) - l64), errs)
    print("logits err bf16 %.3e fp16 %.3e | loss err bf16 %.2e fp16 %.2e | median grad err bf16 %.3f fp16 %.3f" % (
        res["bf16"][0], res["fp16"][0], res["bf16"][1], res["fp16"][1],
        float(np.median(list(res["bf16"][2].values()))), float(np.median(list(res["fp16"][2].values())))))
    assert res["fp16"][0] <= 1.1 * res["bf16"][0] + 1e-3
    assert res["fp16"][1] <= 1.1 * res["bf16"][1] + 1e-4
    worse = [(k, v, res["bf16"][2][k]) for k, v in res["fp16"][2].items() if v > 1.1 * res["bf16"][2][k] + 0.02]
    assert not worse, worse


def test_fp16_mode_trains(A, orc):
    """30 AdamW steps in fp16 mode track the fp32 mode's loss curve like the bf16 mode does (band 5e-3)."""
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 2, 64, 64, 64, generator=g)
    label = (img[:, 0:1] > 0.97).float()
    x, lab = img.cuda(), label.cuda()
    curves = {}
    for dtype in ("fp32", "fp16"):
        m = build(A, orc, 2, dtype)
        opt = A.AdamW(m.parameters(), lr=1e-3)
        losses = []
        for _ in range(30):
            opt.zero_grad(set_to_none=True)
            e, d = m(x)
            loss = A.fused_stage_loss(1, e, d, lab)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        curves[dtype] = np.array(losses)
    f, h = curves["fp32"], curves["fp16"]
    print("loss curve fp32:", np.round(f[::5], 4), "fp16:", np.round(h[::5], 4), "max |diff| %.4f" % np.abs(f - h).max())
    assert np.isfinite(h).all() and h[-1] < h[0] - 0.05 and np.abs(f - h).max() <= 5e-3
