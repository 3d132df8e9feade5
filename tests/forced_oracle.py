"""Test helper: the float64 oracle made to take the SAME discrete choices as a path under test.

SE-UNet contains discrete choices -- the sign of every LeakyReLU input (SE_UNet.py:18,44,60) and the arg-max of every 2x2x2
max-pool window (SE_UNet.py:131-133).  A forward pass that rounds differently than float64 flips a few of them (a handful
out of 12 million at 2 x 32^3), and ONE flipped element moves the gradient of every tensor upstream of it in the backward
pass by 1e-4 ... 1e-3 relative: the fp32 reference itself is 1.8e-3 away from its own float64 run on half of its tensors
(tests/flip_census.py, profiles/r03_flip_census_*.md).  Comparing a path with the plain float64 oracle therefore measures
where its flips happened to fall, not its arithmetic.

``forced_step`` runs the float64 oracle with the LeakyReLU signs and max-pool arg-maxes of the path under test imposed
(taken from the path's own raw conv outputs / statistics / pooled tensors): what float64 arithmetic gives for the same
choices.  Against that, a correct implementation agrees to rounding (1e-5 in fp32) on EVERY tensor -- a flip-free,
network-level parity gate.
"""
import torch
import torch.nn.functional as F

# order of the F.leaky_relu calls in OracleSEUNet.forward (one per block; x-branches have no materialised raw tensor in the HIP
# path: forward_with_intermediates recomputes them with the epilogue.s device function)
LRELU_ORDER = ["ec1", "ec2", "ec3", "ec33", "x33", "ec4", "ec5", "ec6", "ec63", "x63", "ec7", "ec8", "ec9", "ec93", "x93",
               "ec10", "ec11", "ec12", "ec123", "dc1", "dc2", "dc22", "dc3", "dc4", "dc42", "dc5", "dc6"]
# F.max_pool3d calls: pool(e1), pool(x), pool(e3), pool(x1), pool(e5); the feature pools consume the outputs of these blocks
POOL_ORDER = ["ec33", None, "ec63", None, "ec93"]
BLOCKS = [n for n in LRELU_ORDER if not n.startswith("x")]


def path_choices(inter):
    """{block: bool sign mask}, {block: arg-max index tensor} from ``SE_UNet.forward_with_intermediates`` output."""
    signs, pools = {}, {}
    for n, rec in inter.items():
        raw = rec["raw"].double().cpu()
        mean = rec["mean"].double().cpu()[:, :, None, None, None]
        signs[n] = (raw - mean) > 0
        if n in POOL_ORDER and "out" in rec:
            _, idx = F.max_pool3d(rec["out"].double().cpu(), 2, 2, return_indices=True)
            pools[n] = idx
    return signs, pools


def oracle_choices(orc, model, image):
    """The same records from an oracle module (any dtype): used to impose the fp32 reference's choices on float64."""
    raws, pool_in, hooks = {}, [], []
    for n in LRELU_ORDER:
        hooks.append(getattr(model, n).conv1.register_forward_hook(lambda m, i, out, n=n: raws.__setitem__(n, out.detach())))
    real = F.max_pool3d

    def spy(t, *a, **k):
        pool_in.append(t.detach())
        return real(t, *a, **k)
    orc.F.max_pool3d = spy
    try:
        with torch.no_grad():
            model(image)
    finally:
        orc.F.max_pool3d = real
        for h in hooks:
            h.remove()
    signs = {n: (r.double() - r.double().mean(dim=(2, 3, 4), keepdim=True)) > 0 for n, r in raws.items()}
    pools = {}
    for name, t in zip(POOL_ORDER, pool_in):
        if name is not None:
            pools[name] = F.max_pool3d(t.double(), 2, 2, return_indices=True)[1]
    return signs, pools


def forced_step(orc, batch, stage, signs, pools, slope=0.01, width_mult=1, drops=None, n_classes=1):
    """float64 oracle forward + stage loss + backward with the given choices imposed.  Returns (model with .grad, pred0,
    pred1, loss, number of sign choices that differ from float64's own, number of pool choices that differ).
    ``drops``: the two DropLayer scale tensors of a train-mode step (injected into the oracle like into the HIP path)."""
    o = orc.build_oracle(batch["image"].shape[1], n_classes, width_mult, seed=0, train=drops is not None).double()
    calls = {"lrelu": 0, "pool": 0, "sign_flips": 0, "pool_flips": 0}
    real_lrelu, real_pool = F.leaky_relu, F.max_pool3d

    def lrelu(t, negative_slope=0.01, inplace=False):
        name = LRELU_ORDER[calls["lrelu"]]
        calls["lrelu"] += 1
        if name not in signs:
            return real_lrelu(t, negative_slope)
        m = signs[name]
        calls["sign_flips"] += int((m != (t.detach() > 0)).sum())
        return torch.where(m, t, t * negative_slope)

    def pool(t, *a, **k):
        name = POOL_ORDER[calls["pool"]]
        calls["pool"] += 1
        if name is None or name not in pools:
            return real_pool(t, *a, **k)
        idx = pools[name]
        own = real_pool(t.detach(), 2, 2, return_indices=True)[1]
        calls["pool_flips"] += int((own != idx).sum())
        n, c = t.shape[:2]
        return t.flatten(2).gather(2, idx.flatten(2)).reshape(n, c, *idx.shape[2:])
    orc.F.leaky_relu, orc.F.max_pool3d = lrelu, pool
    try:
        if drops is None:
            pe, pd = o(batch["image"].double())
        else:
            pe, pd = o(batch["image"].double(), drops[0].double(), drops[1].double())
    finally:
        orc.F.leaky_relu, orc.F.max_pool3d = real_lrelu, real_pool
    assert calls["lrelu"] == len(LRELU_ORDER) and calls["pool"] == len(POOL_ORDER), calls
    loss = orc.stage_loss(stage, pe, pd, batch["label"].double(), batch["weight"].double(), batch["skel"].double())
    loss.backward()
    return o, pe.detach(), pd.detach(), float(loss.detach()), calls["sign_flips"], calls["pool_flips"]
