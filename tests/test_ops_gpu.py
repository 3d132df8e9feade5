"""Op-level parity of the HIP kernels (through the C ABI) against the same op on the CPU in fp32 torch.

f32 kernels must agree to ~1e-5 relative; bf16 kernels are compared against the fp32 op applied to
bf16-rounded inputs (so what is measured is the kernel, not input quantisation) with a tolerance of a few
bf16 ulps of the output range.  Shapes are deliberately not multiples of the tile sizes (masking)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import seunet_amd
    from seunet_amd import _lib, ops
    _lib.load()
    return ops


DT = ["fp32", "bf16", "fp16"]
IMPL = [0, 1]  # MFMA, naive


def rnd(dtype, t):
    if dtype == "bf16":
        return t.to(torch.bfloat16).float()
    return t.to(torch.float16).float() if dtype == "fp16" else t


def tol(dtype, ref):
    scale = float(ref.abs().max()) + 1e-12
    return {"fp32": 2e-5, "bf16": 1.2e-2, "fp16": 2e-3}[dtype] * scale


def assert_close(got, ref, dtype, what=""):
    got, ref = got.detach().cpu().float(), ref.detach().cpu().float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = float((got - ref).abs().max())
    assert err <= tol(dtype, ref), f"{what}: max|err|={err:.3e} tol={tol(dtype, ref):.3e} ref max={float(ref.abs().max()):.3e}"


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("dtype", DT)
def test_layout_roundtrip(S, dtype):
    x = gen(2, 5, 4, 6, 10, seed=1)
    cl = S.to_cl(x.cuda(), dtype)
    assert cl.shape == (2, 4, 6, 10, 8)
    back = S.from_cl(cl, 5).cpu()
    assert_close(back, rnd(dtype, x), dtype, "roundtrip")
    assert float(S.from_cl(cl).cpu()[:, 5:].abs().max()) == 0.0


CONV_CASES = [
    # (src channel split, logical cin, cout, dilation)
    ([8], 2, 8, 1),            # ec1: padded network input
    ([8], 8, 16, 1),           # ec2
    ([16], 16, 32, 2),         # ec3
    ([32, 32], 64, 32, 1),     # dc5 (fused cat)
    ([64], 64, 64, 2),         # ec8/ec9
    ([64, 64], 128, 64, 1),    # dc1/dc3
    ([32], 32, 16, 1),         # dc6
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("impl", IMPL)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3x3_forward_and_stats(S, dtype, impl, case):
    split, cin, cout, dil = case
    n, d, h, w = 2, 6, 9, 40
    x = rnd(dtype, gen(n, sum(split), d, h, w, seed=2))
    if cin < sum(split):
        x[:, cin:] = 0
    wt = rnd(dtype, gen(cout, cin, 3, 3, 3, seed=3, scale=(27 * cin) ** -0.5))
    b = gen(cout, seed=4, scale=0.1)
    ref = F.conv3d(x[:, :cin], wt, b, padding=dil, dilation=dil)
    srcs, o = [], 0
    for c in split:
        srcs.append(S.to_cl(x[:, o:o + c].cuda(), dtype))
        o += c
    (raw,), part, slots = S.conv3d(srcs, wt.cuda(), b.cuda(), dil, impl, cin=cin, want_stats=True)
    got = S.from_cl(raw, cout)
    assert_close(got, ref, dtype, "conv")
    mean, rstd = S.stats_finalize(part, slots, d * h * w)
    src_stats = ref if impl == 0 else got.cpu()   # MFMA path takes stats from the f32 accumulators
    rm = src_stats.mean(dim=(2, 3, 4))
    rv = src_stats.var(dim=(2, 3, 4), unbiased=False)
    np.testing.assert_allclose(mean.cpu().numpy(), rm.numpy(), atol=2e-5 if dtype == "fp32" else 3e-3)
    np.testing.assert_allclose(rstd.cpu().numpy(), (rv + 1e-5).rsqrt().numpy(), rtol=2e-4 if dtype == "fp32" else 2e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("impl", IMPL)
@pytest.mark.parametrize("split,cout", [([32, 8, 16], 32), ([64, 64, 64], 64), ([32, 64], 32), ([8], 32)])
def test_conv1x1x1_forward(S, dtype, impl, split, cout):
    n, d, h, w = 2, 5, 7, 33
    cin = sum(split) if split != [8] else 2
    x = rnd(dtype, gen(n, sum(split), d, h, w, seed=5))
    if cin < sum(split):
        x[:, cin:] = 0
    wt = rnd(dtype, gen(cout, cin, 1, 1, 1, seed=6, scale=cin ** -0.5))
    ref = F.conv3d(x[:, :cin], wt)
    srcs, o = [], 0
    for c in split:
        srcs.append(S.to_cl(x[:, o:o + c].cuda(), dtype))
        o += c
    (raw,), _, _ = S.conv3d(srcs, wt.cuda(), None, 1, impl, cin=cin)
    assert_close(S.from_cl(raw, cout), ref, dtype, "conv1x1")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("impl", IMPL)
@pytest.mark.parametrize("case", [([32, 32], 32, 1, 27), ([16], 32, 2, 27), ([64, 64], 64, 1, 27),
                                  ([32, 8, 16], 32, 1, 1), ([32, 64], 32, 1, 1),
                                  # narrow destinations only: the store path that goes straight from the accumulators
                                  ([8], 16, 1, 27), ([16], 32, 2, 27), ([8, 16], 32, 1, 27), ([16, 8, 8], 64, 1, 27),
                                  ([8], 32, 1, 1), ([16, 16], 16, 1, 1)])
def test_conv_data_gradient(S, dtype, impl, case):
    """dgrad = same kernel on flipped/transposed weights, split over the concatenated inputs, with +=."""
    split, cout, dil, taps = case
    k = 3 if taps == 27 else 1
    n, d, h, w = 2, 6, 6, 36
    cin = sum(split)
    wt = rnd(dtype, gen(cout, cin, k, k, k, seed=7, scale=(taps * cin) ** -0.5))
    dy = rnd(dtype, gen(n, cout, d, h, w, seed=8))
    x = torch.zeros(n, cin, d, h, w, requires_grad=True)
    F.conv3d(x, wt, padding=dil if k == 3 else 0, dilation=dil if k == 3 else 1).backward(dy)
    ref = x.grad
    prev = rnd(dtype, gen(n, split[0], d, h, w, seed=9))          # first destination accumulates
    dsts = [S.to_cl(prev.cuda(), dtype)] + [torch.empty((n, d, h, w, c), dtype=S._tdtype(S._lib.dtype_code(dtype)), device="cuda")
                                            for c in split[1:]]
    S.conv3d([S.to_cl(dy.cuda(), dtype)], wt.cuda(), None, dil, impl, transpose_flip=True, dsts=dsts,
             accumulate=[1] + [0] * (len(split) - 1))
    o = 0
    for i, c in enumerate(split):
        want = ref[:, o:o + c] + (prev if i == 0 else 0)
        assert_close(S.from_cl(dsts[i]), want if dtype == "fp32" else rnd(dtype, want), dtype, f"dgrad dst{i}")
        o += c


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("impl", IMPL)
@pytest.mark.parametrize("case", [([8], 2, 8, 1, 27), ([16], 16, 32, 2, 27), ([32, 32], 64, 32, 1, 27),
                                  ([64], 64, 64, 2, 27), ([32, 8, 16], 56, 32, 1, 1), ([8], 2, 32, 1, 1)])
def test_conv_weight_gradient(S, dtype, impl, case):
    split, cin, cout, dil, taps = case
    k = 3 if taps == 27 else 1
    n, d, h, w = 2, 5, 6, 40
    x = rnd(dtype, gen(n, sum(split), d, h, w, seed=10))
    dy = rnd(dtype, gen(n, cout, d, h, w, seed=11))
    wt = torch.zeros(cout, cin, k, k, k, requires_grad=True)
    F.conv3d(x[:, :cin], wt, padding=dil if k == 3 else 0, dilation=dil if k == 3 else 1).backward(dy)
    srcs, o = [], 0
    for c in split:
        srcs.append(S.to_cl(x[:, o:o + c].cuda(), dtype))
        o += c
    dw = S.conv3d_wgrad(srcs, S.to_cl(dy.cuda(), dtype), cin, cout, taps, dil, impl)
    assert_close(dw, wt.grad, "fp32", "wgrad")   # accumulation is f32 in both modes


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [([32, 32], 32, 1, 37), ([64], 32, 1, 5), ([32], 64, 2, 37), ([32], 32, 1, 9), ([32], 32, 2, 6),
                                  ([64, 64], 64, 1, 4), ([64], 64, 2, 37), ([32], 64, 1, 3)])
def test_conv_weight_gradient_march(S, dtype, case):
    """csrc/wgrad_march.hip (forced: the dispatcher only picks it for large volumes) against F.conv3d's weight gradient:
    ragged x blocks (w = 40), ragged patches (h = 6), one or two z segments per parity class, both dilations, every
    (input, output) channel layout of the kernel."""
    split, cout, dil, d = case
    cin = sum(split)
    n, h, w = 2, 6, 40
    x = rnd(dtype, gen(n, cin, d, h, w, seed=12))
    dy = rnd(dtype, gen(n, cout, d, h, w, seed=13))
    wt = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x, wt, padding=dil, dilation=dil).backward(dy)
    srcs, o = [], 0
    for c in split:
        srcs.append(S.to_cl(x[:, o:o + c].cuda(), dtype))
        o += c
    dyc = S.to_cl(dy.cuda(), dtype)
    dw = S.conv3d_wgrad(srcs, dyc, cin, cout, 27, dil, S._lib.CONV_MARCH)
    assert_close(dw, wt.grad, "fp32", "wgrad march")
    # deterministic (fixed-order slab sum), and the tiled kernel agrees to f32 summation order
    assert torch.equal(dw, S.conv3d_wgrad(srcs, dyc, cin, cout, 27, dil, S._lib.CONV_MARCH))
    assert_close(dw, S.conv3d_wgrad(srcs, dyc, cin, cout, 27, dil, S._lib.CONV_TILED).cpu(), "fp32", "march vs tiled")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [([64, 32, 32], 64, (2, 5, 6, 40)),      # ec63: 2 input blocks per wave, 128-voxel chunks, ragged tail
                                  ([64, 64, 64], 128, (1, 9, 8, 24)),     # ec93: 3 blocks per wave, 64-voxel chunks
                                  ([32, 32], 32, (2, 3, 5, 7)),           # dc42: 210 voxels = one full chunk + a partial one
                                  ([64, 64], 64, (1, 16, 16, 16))])       # dc22
def test_conv_weight_gradient_1x1(S, dtype, case):
    """csrc/wgrad_1x1.hip (forced: the dispatcher only picks it for >= 16384 voxels) against F.conv3d's weight gradient, and
    against the tiled kernel."""
    split, cout, (n, d, h, w) = case
    cin = sum(split)
    x = rnd(dtype, gen(n, cin, d, h, w, seed=14))
    dy = rnd(dtype, gen(n, cout, d, h, w, seed=15))
    wt = torch.zeros(cout, cin, 1, 1, 1, requires_grad=True)
    F.conv3d(x, wt).backward(dy)
    srcs, o = [], 0
    for c in split:
        srcs.append(S.to_cl(x[:, o:o + c].cuda(), dtype))
        o += c
    dyc = S.to_cl(dy.cuda(), dtype)
    dw = S.conv3d_wgrad(srcs, dyc, cin, cout, 1, 1, S._lib.CONV_MARCH)
    assert_close(dw, wt.grad, "fp32", "wgrad 1x1")
    assert torch.equal(dw, S.conv3d_wgrad(srcs, dyc, cin, cout, 1, 1, S._lib.CONV_MARCH))
    assert_close(dw, S.conv3d_wgrad(srcs, dyc, cin, cout, 1, 1, S._lib.CONV_TILED).cpu(), "fp32", "1x1 vs tiled")


def test_conv_weight_gradient_march_refuses_unserved_layers(S):
    x = S.to_cl(torch.zeros(1, 16, 4, 4, 32).cuda(), "bf16")
    dy = S.to_cl(torch.zeros(1, 32, 4, 4, 32).cuda(), "bf16")
    with pytest.raises(RuntimeError, match="wgrad_march"):
        S.conv3d_wgrad([x], dy, 16, 32, 27, 1, S._lib.CONV_MARCH)


def _block_ref(raw, w_se, w_se2, w_side, b_side, slope=0.01):
    e = F.leaky_relu(F.instance_norm(raw), slope)
    e = e * torch.sigmoid(F.conv3d(e, w_se))
    if w_se2 is not None:
        e = e * torch.sigmoid(F.conv3d(e, w_se2))
    return e, F.conv3d(e, w_side, b_side)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,gates", [(8, 1), (16, 1), (32, 2), (64, 2)])
def test_gate_epilogue_forward_backward(S, dtype, c, gates):
    n, d, h, w = 2, 6, 10, 12
    raw = rnd(dtype, gen(n, c, d, h, w, seed=12) * 2 + 0.3).requires_grad_(True)
    w_se, w_se2 = gen(1, c, 1, 1, 1, seed=13, scale=0.5), (gen(1, c, 1, 1, 1, seed=14, scale=0.5) if gates == 2 else None)
    w_side, b_side = gen(2, c, 1, 1, 1, seed=15, scale=0.5), gen(2, seed=16, scale=0.1)
    prm = [t.requires_grad_(True) for t in (w_se, w_se2, w_side, b_side) if t is not None]
    e_ref, s_ref = _block_ref(raw, w_se, w_se2, w_side, b_side)
    g_e = rnd(dtype, gen(n, c, d, h, w, seed=17))
    g_s = gen(n, 2, d, h, w, seed=18)
    (e_ref * g_e).sum().add((s_ref * g_s).sum()).backward()

    raw_cl = S.to_cl(raw.detach().cuda(), dtype)
    part, slots = S.channel_stats(raw_cl)
    mean, rstd = S.stats_finalize(part, slots, d * h * w)
    cu = lambda t: None if t is None else t.detach().cuda()
    e, side = S.gate_epilogue_fwd(raw_cl, mean, rstd, cu(w_se), cu(w_se2), cu(w_side), cu(b_side))
    assert_close(S.from_cl(e), e_ref, dtype, "e")
    assert_close(side.permute(0, 4, 1, 2, 3), s_ref, "fp32" if dtype == "fp32" else "bf16", "side")
    out = S.gate_epilogue_bwd(raw_cl, mean, rstd, cu(w_se), cu(w_se2), cu(w_side), cu(b_side), g_e=S.to_cl(g_e.cuda(), dtype),
                              g_side=g_s.permute(0, 2, 3, 4, 1).contiguous().cuda())
    assert_close(S.from_cl(out["draw"]), raw.grad, dtype, "draw")
    looser = "fp32" if dtype == "fp32" else "bf16"
    assert_close(out["dw_se"], w_se.grad.reshape(-1), looser, "dw_se")
    if gates == 2:
        assert_close(out["dw_se2"], w_se2.grad.reshape(-1), looser, "dw_se2")
    assert_close(out["dw_side"], w_side.grad.reshape(-1), looser, "dw_side")
    assert_close(out["db_side"], b_side.grad, looser, "db_side")


@pytest.mark.parametrize("dtype", DT)
def test_gate_epilogue_level_map_and_head_grad(S, dtype):
    """The head weight / DropLayer scale applied at native resolution (SE_UNet.py:232-233 is linear)."""
    n, c, d, h, w = 2, 16, 4, 6, 8
    raw = rnd(dtype, gen(n, c, d, h, w, seed=19))
    w_se, w_side, b_side = gen(1, c, 1, 1, 1, seed=20), gen(2, c, 1, 1, 1, seed=21), gen(2, seed=22)
    head_w = gen(2, seed=23).requires_grad_(True)
    drop = torch.tensor([[0.0, 1.3, 9, 9], [0.7, 0.7, 9, 9]])
    _, s_ref = _block_ref(raw, w_se, None, w_side, b_side)
    lvl_ref = (s_ref * (head_w.view(1, 2, 1, 1, 1) * drop[:, :2].view(n, 2, 1, 1, 1))).sum(1)
    g_lvl = gen(n, d, h, w, seed=24)
    (lvl_ref * g_lvl).sum().backward()
    raw_cl = S.to_cl(raw.cuda(), dtype)
    part, slots = S.channel_stats(raw_cl)
    mean, rstd = S.stats_finalize(part, slots, d * h * w)
    lvl = torch.full((n, d, h, w), 5.0, device="cuda")
    S.gate_epilogue_fwd(raw_cl, mean, rstd, w_se.cuda(), None, w_side.cuda(), b_side.cuda(), level_map=lvl, level_accumulate=1,
                        head_w=head_w.detach().cuda(), drop=drop.cuda(), drop_stride=4, want_side=False)
    assert_close(lvl - 5.0, lvl_ref, dtype, "level map")
    out = S.gate_epilogue_bwd(raw_cl, mean, rstd, w_se.cuda(), None, w_side.cuda(), b_side.cuda(), g_level=g_lvl.cuda(),
                              head_w=head_w.detach().cuda(), drop=drop.cuda(), drop_stride=4)
    assert_close(out["dhead_w"], head_w.grad, "fp32" if dtype == "fp32" else "bf16", "dhead_w")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("two", [False, True])
def test_cat_epilogue_forward_backward(S, dtype, two):
    n, c, d, h, w = 2, 32, 5, 6, 12
    raw = rnd(dtype, gen(n, c, d, h, w, seed=25) + 0.2).requires_grad_(True)
    raw2 = rnd(dtype, gen(n, c, d, h, w, seed=26) * 3).requires_grad_(True) if two else None
    ref = F.leaky_relu(F.instance_norm(raw), 0.01)
    if two:
        ref = ref + F.leaky_relu(F.instance_norm(raw2), 0.01)
    g = rnd(dtype, gen(n, c, d, h, w, seed=27))
    (ref * g).sum().backward()

    def stats(t):
        cl = S.to_cl(t.detach().cuda(), dtype)
        p, s = S.channel_stats(cl)
        return (cl,) + S.stats_finalize(p, s, d * h * w)
    a = stats(raw)
    b = stats(raw2) if two else (None, None, None)
    out = S.cat_epilogue_fwd(*a, *b)
    assert_close(S.from_cl(out), ref, dtype, "cat out")
    dx, dx2 = S.cat_epilogue_bwd(S.to_cl(g.cuda(), dtype), *a, *b)
    assert_close(S.from_cl(dx), raw.grad, dtype, "cat draw")
    if two:
        assert_close(S.from_cl(dx2), raw2.grad, dtype, "cat draw2")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("inch,c", [(2, 32), (1, 32), (2, 64), (2, 128)])
def test_cat_epilogue_with_recomputed_x_branch(S, dtype, inch, c):
    """x-branch (x33 / x63 / x93): raw2 = conv1x1(x) is never stored -- statistics from the input moments, values
    recomputed in every pass; the weight gradient is formed in f64 from pass-A sums and the input's moments (it is what is left
    of terms that cancel, see cat_bwd_kernel XW): in fp32 mode it must agree with a FLOAT64 evaluation of the same graph to 2e-6
    of its largest element -- the f32 torch graph itself is only good to ~1e-5 here."""
    n, d, h, w = 2, 5, 6, 12
    x = rnd(dtype, gen(n, inch, d, h, w, seed=31) + 0.3)
    w2 = (gen(c, inch, 1, 1, 1, seed=32) * 0.7).requires_grad_(True)
    raw = rnd(dtype, gen(n, c, d, h, w, seed=33) + 0.2).requires_grad_(True)
    ref = F.leaky_relu(F.instance_norm(raw), 0.01) + F.leaky_relu(F.instance_norm(F.conv3d(x, w2)), 0.01)
    g = rnd(dtype, gen(n, c, d, h, w, seed=34))
    (ref * g).sum().backward()

    cl = S.to_cl(raw.detach().cuda(), dtype)
    p, sl = S.channel_stats(cl)
    mean, rstd = S.stats_finalize(p, sl, d * h * w)
    xin = torch.zeros((n, d, h, w, 8), dtype=S._tdtype(S._lib.dtype_code(dtype)), device="cuda")
    xin[..., :inch] = x.permute(0, 2, 3, 4, 1).cuda().to(xin.dtype)
    out, dx, dw = S.cat_epilogue_x(S.to_cl(g.cuda(), dtype), cl, mean, rstd, xin, w2.detach().cuda(), inch)
    assert_close(S.from_cl(out), ref.detach(), dtype, "cat out")
    assert_close(S.from_cl(dx), raw.grad, dtype, "cat draw")
    scale = float(w2.grad.abs().max())
    tol = 3e-5 if dtype == "fp32" else 2e-2
    assert float((dw.cpu() - w2.grad).abs().max()) <= tol * scale, (float((dw.cpu() - w2.grad).abs().max()), scale)
    if dtype == "fp32":
        w64 = w2.detach().double().requires_grad_(True)
        r64 = raw.detach().double()
        ref64 = F.leaky_relu(F.instance_norm(r64), 0.01) + F.leaky_relu(F.instance_norm(F.conv3d(x.double(), w64)), 0.01)
        (ref64 * g.double()).sum().backward()
        err = float((dw.cpu().double() - w64.grad).abs().max())
        assert err <= 2e-6 * float(w64.grad.abs().max()), (err, float(w64.grad.abs().max()))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c", [8, 32, 64, 128, 24])
def test_maxpool_ties_and_channel_counts(S, dtype, c):
    """Max-pool forward / backward on inputs FULL of ties (small integers): the gradient goes to the first maximum of the
    window in (z, y, x) scan order, like PyTorch's CPU kernel."""
    n, d, h, w = 2, 6, 4, 10
    g0 = torch.Generator().manual_seed(50 + c)
    x = torch.randint(0, 3, (n, c, d, h, w), generator=g0).float().requires_grad_(True)
    p_ref = F.max_pool3d(x, 2, 2)
    g = rnd(dtype, gen(*p_ref.shape, seed=51))
    p_ref.backward(g)
    xc = S.to_cl(x.detach().cuda(), dtype)
    assert torch.equal(S.from_cl(S.maxpool_fwd(xc)).cpu().float(), p_ref.detach())
    gi = S.from_cl(S.maxpool_bwd(xc, S.to_cl(g.cuda(), dtype))).cpu().float()
    assert torch.equal(gi, x.grad), float((gi - x.grad).abs().max())


@pytest.mark.parametrize("dtype", DT)
def test_maxpool_and_upsample(S, dtype):
    n, c, d, h, w = 2, 16, 4, 6, 10
    x = rnd(dtype, gen(n, c, d, h, w, seed=28)).requires_grad_(True)
    p_ref = F.max_pool3d(x, 2, 2)
    g = rnd(dtype, gen(*p_ref.shape, seed=29))
    p_ref.backward(g)
    xc = S.to_cl(x.detach().cuda(), dtype)
    assert_close(S.from_cl(S.maxpool_fwd(xc)), p_ref, dtype, "maxpool")
    prev = rnd(dtype, gen(n, c, d, h, w, seed=30))
    gi = S.maxpool_bwd(xc, S.to_cl(g.cuda(), dtype), S.to_cl(prev.cuda(), dtype))
    assert_close(S.from_cl(gi), rnd(dtype, x.grad + prev), dtype, "maxpool bwd (+=)")

    y = rnd(dtype, gen(n, c, d, h, w, seed=31)).requires_grad_(True)
    u_ref = F.interpolate(y, scale_factor=2, mode="trilinear", align_corners=True)
    gu = rnd(dtype, gen(*u_ref.shape, seed=32))
    u_ref.backward(gu)
    assert_close(S.from_cl(S.upsample2_fwd(S.to_cl(y.detach().cuda(), dtype))), u_ref, dtype, "upsample")
    assert_close(S.from_cl(S.upsample2_bwd(S.to_cl(gu.cuda(), dtype))), y.grad, dtype, "upsample bwd")
    # accumulate form (the gradient buffer already holds another consumer's contribution)
    init = rnd(dtype, gen(n, c, d, h, w, seed=43))
    got = S.from_cl(S.upsample2_bwd(S.to_cl(gu.cuda(), dtype), g_in=S.to_cl(init.cuda(), dtype)))
    assert_close(got, y.grad + init, dtype, "upsample bwd +=")


def test_maxpool_tie_goes_to_first(S):
    x = torch.zeros(1, 8, 2, 2, 2)
    x[0, :, 0, 1, 1] = 1.0
    x[0, :, 1, 0, 0] = 1.0        # tie: first in (z,y,x) scan order wins
    xr = x.clone().requires_grad_(True)
    F.max_pool3d(xr, 2, 2).backward(torch.ones(1, 8, 1, 1, 1))
    gi = S.maxpool_bwd(S.to_cl(x.cuda(), "bf16"), S.to_cl(torch.ones(1, 8, 1, 1, 1).cuda(), "bf16"))
    assert torch.equal(S.from_cl(gi).cpu(), xr.grad)


@pytest.mark.parametrize("case", [(2, 16, 8, 24, 4), (1, 8, 8, 136, 4), (1, 8, 4, 20, 3), (2, 8, 16, 320, 3)])
def test_heads_forward_backward(S, case):
    """row form (W % 8 == 0; 136: more than one pass of a wave over the row), the per-voxel form (W = 20) and the x pass of
    the backward without its row prefetch (W = 320), against F.interpolate(align_corners=True) and autograd."""
    n, d, h, w, nl = case
    maps = [gen(n, d >> l, h >> l, w >> l, seed=33 + l).requires_grad_(True) for l in range(nl)]
    bias = gen(1, seed=40)
    ref = bias.view(1, 1, 1, 1, 1) + maps[0].unsqueeze(1)
    for l in range(1, nl):
        ref = ref + F.interpolate(maps[l].unsqueeze(1), scale_factor=2 ** l, mode="trilinear", align_corners=True)
    g = gen(*ref.shape, seed=41)
    ref.backward(g)
    pred = S.head_fwd([m.detach().cuda() for m in maps], bias.cuda())
    assert_close(pred, ref, "fp32", "head fwd")
    levels, gb = S.head_bwd(g.cuda(), nl)
    for l in range(1, nl):
        assert_close(levels[l], maps[l].grad, "fp32", f"head bwd level {l}")
    assert abs(float(gb.cpu()) - float(g.sum())) < 1e-3


def test_side_upsample(S):
    side = gen(2, 3, 4, 5, 2, seed=42)
    ref = F.interpolate(side.permute(0, 4, 1, 2, 3), scale_factor=4, mode="trilinear", align_corners=True)
    assert_close(S.side_upsample(side.cuda(), 4), ref, "fp32", "side upsample")
    assert_close(S.side_upsample(side.cuda(), 1), side.permute(0, 4, 1, 2, 3), "fp32", "side upsample x1")


def test_losses_match_oracle(S):
    import seunet_oracle as orc
    import seunet_amd as A
    g = torch.Generator().manual_seed(5)
    logit = torch.randn(2, 1, 12, 12, 12, generator=g)
    t = (torch.rand(2, 1, 12, 12, 12, generator=g) > 0.9).float()
    wt = 1 + torch.rand(2, 1, 12, 12, 12, generator=g)
    sk = t * (torch.rand(2, 1, 12, 12, 12, generator=g) > 0.5).float()
    for name, args in (("dice_loss", (t,)), ("general_union_loss_lib", (t, wt)), ("atr_loss", (t, sk, wt))):
        p = torch.sigmoid(logit).requires_grad_(True)
        l_ref = getattr(orc, name)(p, *args)
        l_ref.backward()
        pg = torch.sigmoid(logit).cuda().requires_grad_(True)
        l = getattr(A, name)(pg, *[a.cuda() for a in args])
        (3.0 * l).backward()
        assert abs(float(l) - float(l_ref)) < 2e-6, name
        np.testing.assert_allclose(pg.grad.cpu().numpy() / 3.0, p.grad.numpy(), rtol=2e-4, atol=1e-9, err_msg=name)
    # the one-launch value kernel == the scalar arithmetic it replaces, to the bit
    from seunet_amd.losses import _value, _value_dev
    sums = (torch.rand(2, 7, dtype=torch.float64, generator=g) * 1000).cuda()
    for c0, c1 in (((1.0, 0.0, 0.0), (1.0, 0.0, 0.0)), ((0.0, 1.0, 0.5), (0.0, 0.5, 0.5)), ((0.3, 0.0, 2.0), (0.0, 0.0, 0.0))):
        assert torch.equal(_value_dev(sums[0], c0), _value(sums[0], *c0))
        assert torch.equal(_value_dev(sums[0], c0, sums[1], c1), _value(sums[0], *c0) + _value(sums[1], *c1))
    for stage in (1, 2, 3):
        a = logit.clone().requires_grad_(True)
        b = (logit * 0.5 + 0.1).clone().requires_grad_(True)
        l_ref = orc.stage_loss(stage, a, b, t, wt, sk)
        l_ref.backward()
        ag, bg = a.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
        l = A.fused_stage_loss(stage, ag, bg, t.cuda(), wt.cuda(), sk.cuda())
        l.backward()
        assert abs(float(l) - float(l_ref)) < 5e-6, stage
        np.testing.assert_allclose(ag.grad.cpu().numpy(), a.grad.numpy(), rtol=3e-4, atol=1e-9)
        np.testing.assert_allclose(bg.grad.cpu().numpy(), b.grad.numpy(), rtol=3e-4, atol=1e-9)


# ---- streaming small-channel convolution (csrc/conv_stream.hip): the full-resolution layers ec1 / ec2 / ec3 / dc6 ----
STREAM_FWD = [  # (padded source channels, logical cin, cout, dilation)
    (8, 2, 8, 1),      # ec1: packed network input, x-taps folded into K
    (8, 8, 16, 1),     # ec2
    (16, 16, 32, 2),   # ec3
    (32, 32, 16, 1),   # dc6
    (16, 16, 32, 1), (32, 32, 16, 2), (16, 16, 8, 2),
]


@pytest.mark.parametrize("shape", [(2, 6, 9, 40), (1, 37, 16, 32), (1, 5, 8, 31), (1, 80, 8, 64)])
@pytest.mark.parametrize("case", STREAM_FWD)
def test_conv_stream_forward_and_stats(S, case, shape):
    """Forward + InstanceNorm partial sums against F.conv3d on bf16-rounded operands; shapes with ragged patches (y, x not
    multiples of 8 / 32), marches longer than one segment (80 planes) and both z-parity classes of dilation 2."""
    src_c, cin, cout, dil = case
    n, d, h, w = shape
    x = rnd("bf16", gen(n, src_c, d, h, w, seed=2))
    if cin < src_c:
        x[:, cin:] = 0
    wt = rnd("bf16", gen(cout, cin, 3, 3, 3, seed=3, scale=(27 * cin) ** -0.5))
    b = gen(cout, seed=4, scale=0.1)
    ref = F.conv3d(x[:, :cin], wt, b, padding=dil, dilation=dil)
    raw, part, slots = S.conv3d_stream(S.to_cl(x.cuda(), "bf16"), wt.cuda(), b.cuda(), dil, want_stats=True)
    got = S.from_cl(raw, cout)
    assert_close(got, ref, "bf16", "conv_stream")
    mean, rstd = S.stats_finalize(part, slots, d * h * w)
    rm, rv = ref.mean(dim=(2, 3, 4)), ref.var(dim=(2, 3, 4), unbiased=False)
    np.testing.assert_allclose(mean.cpu().numpy()[:, :cout], rm.numpy(), atol=3e-3)
    np.testing.assert_allclose(rstd.cpu().numpy()[:, :cout], (rv + 1e-5).rsqrt().numpy(), rtol=2e-2)
    # the same numbers as the tiled implicit-GEMM kernel to bf16 rounding of the stored tensor
    (raw2,), _, _ = S.conv3d([S.to_cl(x.cuda(), "bf16")], wt.cuda(), b.cuda(), dil, 0, cin=cin)
    assert float((S.from_cl(raw2, cout) - got).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("case", [(16, 16, 32, 2), (32, 32, 16, 1), (8, 8, 16, 1)])
def test_conv_stream_result_of_a_sample_does_not_depend_on_its_batch(S, case):
    """The z axis is cut into marches by the SAMPLE's extents only (csrc/conv_stream.hip stream_zsteps), so a sample's raw output
    and its InstanceNorm partial records are the same bits alone and inside a batch of three -- what the data-parallel
    equivalence tests and the window loop build on.  72 planes: more than one march at dilation 1."""
    src_c, cin, cout, dil = case
    x = rnd("bf16", gen(3, src_c, 72, 16, 64, seed=12))
    wt = rnd("bf16", gen(cout, cin, 3, 3, 3, seed=13, scale=(27 * cin) ** -0.5))
    b = gen(cout, seed=14, scale=0.1)
    raw3, part3, slots3 = S.conv3d_stream(S.to_cl(x.cuda(), "bf16"), wt.cuda(), b.cuda(), dil, want_stats=True)
    raw1, part1, slots1 = S.conv3d_stream(S.to_cl(x[1:2].cuda(), "bf16"), wt.cuda(), b.cuda(), dil, want_stats=True)
    assert slots1 == slots3
    assert torch.equal(raw3[1:2], raw1)
    assert torch.equal(part3[1:2], part1)


@pytest.mark.parametrize("acc", [False, True])
@pytest.mark.parametrize("case", [(16, 8, 8, 1), (32, 16, 16, 2), (16, 32, 32, 1), (16, 8, 16, 1)])   # (dy channels, dx channels, dx tensor channels, dil)
def test_conv_stream_data_gradient(S, case, acc):
    """dgrad of ec2 (dy 16 -> dx 8), ec3 (dy 32 -> dx 16, dilation 2), dc6 (dy 16 -> dx 32): the forward weight
    (cout = dy channels, cin = dx channels) applied transposed / mirrored, optionally += into an existing gradient."""
    dyc, dxc, dst_c, dil = case
    n, d, h, w = 2, 7, 10, 36
    wt = rnd("bf16", gen(dyc, dxc, 3, 3, 3, seed=5, scale=(27 * dxc) ** -0.5))
    dy = rnd("bf16", gen(n, dyc, d, h, w, seed=6))
    xx = torch.zeros(n, dxc, d, h, w, requires_grad=True)
    F.conv3d(xx, wt, padding=dil, dilation=dil).backward(dy)
    ref = xx.grad
    old = rnd("bf16", gen(n, dst_c, d, h, w, seed=7)) if acc else None
    dst = S.to_cl(old.cuda(), "bf16") if acc else None
    got, _, _ = S.conv3d_stream(S.to_cl(dy.cuda(), "bf16"), wt.cuda(), None, dil, transpose_flip=True, dst=dst, dst_channels=dst_c,
                                accumulate=acc)
    g = S.from_cl(got, dxc)
    want = ref + (old[:, :dxc] if acc else 0)
    assert_close(g, want, "bf16", "conv_stream dgrad")
    if dst_c > dxc:
        assert float(S.from_cl(got)[:, dxc:].abs().max()) == 0.0 or acc


@pytest.mark.parametrize("shape", [(2, 6, 9, 40), (1, 37, 16, 32), (1, 5, 8, 31), (1, 80, 8, 64)])
@pytest.mark.parametrize("case", [(8, 2, 8, 1), (8, 8, 16, 1), (16, 16, 32, 2), (32, 32, 16, 1), (16, 16, 32, 1), (16, 16, 16, 2), (8, 8, 16, 2)])
def test_conv_stream_weight_gradient(S, case, shape):
    """dW of ec1 (2 of 8 packed channels -> 8), ec2 (8 -> 16), ec3 (16 -> 32, dilation 2), dc6 (32 -> 16) on the streaming
    kernel against autograd on bf16-rounded operands, ragged patches and marches longer than one segment included."""
    x_c, cin, cout, dil = case
    n, d, h, w = shape
    x = rnd("bf16", gen(n, x_c, d, h, w, seed=8))
    if cin < x_c:
        x[:, cin:] = 0
    dy = rnd("bf16", gen(n, cout, d, h, w, seed=9))
    wt = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x[:, :cin], wt, padding=dil, dilation=dil).backward(dy)
    got = S.conv3d_wgrad_stream(S.to_cl(x.cuda(), "bf16"), S.to_cl(dy.cuda(), "bf16"), cin, cout, dil)
    assert got.shape == wt.grad.shape
    err = float((got.cpu() - wt.grad).abs().max())
    assert err <= 2e-3 * float(wt.grad.abs().max()) + 1e-4, (err, float(wt.grad.abs().max()))
    # and the tiled kernel's answer on the same operands
    ref2 = S.conv3d_wgrad([S.to_cl(x.cuda(), "bf16")], S.to_cl(dy.cuda(), "bf16"), cin, cout, 27, dil)
    assert float((got - ref2).abs().max()) <= 2e-3 * float(wt.grad.abs().max()) + 1e-4


# ---- marching convolution (csrc/conv_march.hip): the 32 / 64-input-channel layers of the fine levels ----
MARCH_FWD = [  # (source channel split, cout, dilation)
    ([32, 32], 32, 1),   # dc5 (fused cat): 2 K-steps, 2 N groups x 2 row groups
    ([64], 32, 1),       # dc4
    ([32], 32, 1),       # ec4
    ([32], 32, 2),       # ec5
    ([32], 64, 2),       # ec6: 4 N groups, 8 rows per wave
    ([32], 64, 1),
    ([64], 64, 1),       # ec7 / dc2 shape
    ([64], 64, 2),       # ec8 / ec9 shape
    ([64], 32, 2),
    ([32, 32], 128, 1),  # two N blocks
]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 6, 9, 40), (1, 13, 16, 32), (1, 5, 8, 31), (1, 41, 8, 64)])
@pytest.mark.parametrize("case", MARCH_FWD)
def test_conv_march_forward_and_stats(S, case, shape, dtype):
    """Forward + InstanceNorm partial sums against F.conv3d on rounded operands; ragged patches (y, x not multiples of the
    patch), marches split into segments (41 planes on few workgroups), both z-parity classes of dilation 2, fused
    concatenation of two sources, several N blocks."""
    split, cout, dil = case
    n, d, h, w = shape
    cin = sum(split)
    x = rnd(dtype, gen(n, cin, d, h, w, seed=2))
    wt = rnd(dtype, gen(cout, cin, 3, 3, 3, seed=3, scale=(27 * cin) ** -0.5))
    b = gen(cout, seed=4, scale=0.1)
    ref = F.conv3d(x, wt, b, padding=dil, dilation=dil)
    srcs, o = [], 0
    for c in split:
        srcs.append(S.to_cl(x[:, o:o + c].cuda(), dtype))
        o += c
    (raw,), part, slots = S.conv3d_march(srcs, wt.cuda(), b.cuda(), dil, want_stats=True)
    got = S.from_cl(raw, cout)
    assert_close(got, ref, dtype, "conv_march")
    mean, rstd = S.stats_finalize(part, slots, d * h * w)
    rm, rv = ref.mean(dim=(2, 3, 4)), ref.var(dim=(2, 3, 4), unbiased=False)
    np.testing.assert_allclose(mean.cpu().numpy(), rm.numpy(), atol=3e-3)
    np.testing.assert_allclose(rstd.cpu().numpy(), (rv + 1e-5).rsqrt().numpy(), rtol=2e-2)
    # the same numbers as the tiled implicit-GEMM kernel to rounding of the stored tensor
    (raw2,), _, _ = S.conv3d(srcs, wt.cuda(), b.cuda(), dil, 0)
    assert float((S.from_cl(raw2, cout) - got).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("acc", [[0, 0], [1, 0], [1, 1]])
@pytest.mark.parametrize("case", [(32, [32, 32], 1),   # dc5: dy 32 -> dx 32 + 32 (two destinations)
                                  (32, [64], 1),       # dc4
                                  (64, [32], 2),       # ec6
                                  (32, [32], 2),       # ec5
                                  (32, [32], 1),       # ec4
                                  (64, [64, 64], 1),   # dc3: two N blocks, one per destination
                                  (64, [64], 2)])      # ec8 / ec9 shape
def test_conv_march_data_gradient(S, case, acc, dtype):
    """dgrad: the forward weight (cout = dy channels, cin = dx channels) applied transposed / mirrored, split over the
    concatenated inputs, each destination optionally += into an existing gradient."""
    dyc, split, dil = case
    acc = acc[:len(split)]
    n, d, h, w = 2, 7, 10, 36
    dxc = sum(split)
    wt = rnd(dtype, gen(dyc, dxc, 3, 3, 3, seed=5, scale=(27 * dxc) ** -0.5))
    dy = rnd(dtype, gen(n, dyc, d, h, w, seed=6))
    xx = torch.zeros(n, dxc, d, h, w, requires_grad=True)
    F.conv3d(xx, wt, padding=dil, dilation=dil).backward(dy)
    ref = xx.grad
    olds = [rnd(dtype, gen(n, c, d, h, w, seed=7 + i)) for i, c in enumerate(split)]
    dsts = [S.to_cl(o.cuda(), dtype) for o in olds]
    got, _, _ = S.conv3d_march([S.to_cl(dy.cuda(), dtype)], wt.cuda(), None, dil, transpose_flip=True, dsts=dsts, accumulate=acc)
    o = 0
    for i, c in enumerate(split):
        want = ref[:, o:o + c] + (olds[i] if acc[i] else 0)
        assert_close(S.from_cl(got[i], c), want, dtype, f"conv_march dgrad dst {i}")
        o += c


def test_conv_march_dropped_destination_and_long_march(S):
    """A null destination drops its channels (data gradient towards the network input side); 70 planes in one march."""
    n, d, h, w = 1, 70, 8, 32
    wt = rnd("bf16", gen(32, 64, 3, 3, 3, seed=11, scale=(27 * 64) ** -0.5))
    dy = rnd("bf16", gen(n, 32, d, h, w, seed=12))
    xx = torch.zeros(n, 64, d, h, w, requires_grad=True)
    F.conv3d(xx, wt, padding=1).backward(dy)
    d1 = torch.full((n, d, h, w, 32), 7.0, dtype=torch.bfloat16, device="cuda")
    got, _, _ = S.conv3d_march([S.to_cl(dy.cuda(), "bf16")], wt.cuda(), None, 1, transpose_flip=True, dsts=[None, d1], dst_channels=[32, 32])
    assert_close(S.from_cl(got[1], 32), xx.grad[:, 32:], "bf16", "conv_march dropped dst")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(1, 32, 5, 7, 70), (2, 64, 3, 4, 33), (1, 8, 1, 1, 2), (1, 16, 2, 9, 64), (1, 128, 2, 3, 5),
                                   (2, 64, 9, 4, 33), (1, 128, 4, 6, 20), (1, 32, 40, 9, 17), (1, 32, 33, 16, 16), (1, 64, 4, 4, 4)])
def test_upsample2_forward_backward_shapes(S, dtype, shape):
    """x2 trilinear (align_corners=True) on the tiled kernels: more than one 128-voxel x-chunk, odd extents, single rows,
    the 128-channel case (width x2); the backward takes the z-marching kernel when C is 32 / 64 / 128 and every coarse
    extent is >= 4 (ragged y / x tiles, several z segments, a ragged last segment), the per-plane tiled kernel or the gather
    kernel otherwise."""
    n, c, d, h, w = shape
    y = rnd(dtype, gen(n, c, d, h, w, seed=41)).requires_grad_(True)
    u_ref = F.interpolate(y, scale_factor=2, mode="trilinear", align_corners=True)
    gu = rnd(dtype, gen(*u_ref.shape, seed=42))
    u_ref.backward(gu)
    assert_close(S.from_cl(S.upsample2_fwd(S.to_cl(y.detach().cuda(), dtype))), u_ref, dtype, "upsample")
    assert_close(S.from_cl(S.upsample2_bwd(S.to_cl(gu.cuda(), dtype))), y.grad, dtype, "upsample bwd")
