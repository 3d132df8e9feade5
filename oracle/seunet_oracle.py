"""CPU oracle for the SE-UNet hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU *restatement* of what the reference computes on
the path named in BASELINE.json.  It is the checker: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  The product path (``se-unet-airseg_amd``) never imports it and
fails loudly when the HIP library is missing.

Pinning: the reference ships no tests, golden vectors or trained weights
(SURVEY.md section 4), so this restatement is pinned against the reference
module itself: ``oracle/make_golden.py`` imports ``/root/reference/SE_UNet.py``
(pure torch) in the build container, runs it on the deterministic weights and
inputs defined *here*, and commits the outputs under ``tests/golden/``.
``tests/test_oracle_golden.py`` re-checks this file against those fixtures on
every run (also on the GPU box, where /root/reference does not exist).

Everything is written from the reference's behaviour, each function cites the
file:line it follows.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# network topology (reference SE_UNet.py:108-153).  One table drives both the
# oracle module below and the parameter registry checks in the tests.
# (name, kind, cin, cout, dilation, side-map upsample factor)
#   kind 'g1' = one spatial gate   (reference class SSEConv,  SE_UNet.py:9-35)
#   kind 'g2' = two spatial gates  (reference class SSEConv2, SE_UNet.py:51-82)
#   kind 'cat' = 1x1x1 conv + IN + LeakyReLU (reference CATConv, SE_UNet.py:37-49)
# Channel counts are for width_mult == 1 and are multiplied by width_mult,
# except entries written as 'in' (= in_channel).
# --------------------------------------------------------------------------
TOPOLOGY = [
    ("ec1", "g1", "in", 8, 1, 1),
    ("ec2", "g1", 8, 16, 1, 1),
    ("ec3", "g1", 16, 32, 2, 1),
    ("ec33", "cat", 56, 32, 0, 0),
    ("x33", "cat", "in", 32, 0, 0),
    ("ec4", "g2", 32, 32, 1, 2),
    ("ec5", "g2", 32, 32, 2, 2),
    ("ec6", "g2", 32, 64, 2, 2),
    ("ec63", "cat", 128, 64, 0, 0),
    ("x63", "cat", "in", 64, 0, 0),
    ("ec7", "g2", 64, 64, 1, 4),
    ("ec8", "g2", 64, 64, 2, 4),
    ("ec9", "g2", 64, 64, 2, 4),
    ("ec93", "cat", 192, 64, 0, 0),
    ("x93", "cat", "in", 64, 0, 0),
    ("ec10", "g2", 64, 64, 1, 8),
    ("ec11", "g2", 64, 64, 1, 8),
    ("ec12", "g2", 64, 64, 1, 8),
    ("ec123", "cat", 192, 64, 0, 0),
    ("dc1", "g2", 128, 64, 1, 4),
    ("dc2", "g2", 64, 64, 1, 4),
    ("dc22", "cat", 128, 64, 0, 0),
    ("dc3", "g2", 128, 64, 1, 2),
    ("dc4", "g2", 64, 32, 1, 2),
    ("dc42", "cat", 96, 32, 0, 0),
    ("dc5", "g1", 64, 32, 1, 1),
    ("dc6", "g1", 32, 16, 1, 1),
    ("dc62", "cat", 48, 16, 0, 0),
]
ENCODER_SIDE_BLOCKS = ["ec%d" % i for i in range(1, 13)]   # s0..s11, SE_UNet.py:232
DECODER_SIDE_BLOCKS = ["dc%d" % i for i in range(1, 7)]     # s12..s17, SE_UNet.py:233
DROP_THRESHOLD = 0.3                                         # SE_UNet.py:152-153


def _ch(v, in_channel: int, width_mult: int) -> int:
    return in_channel if v == "in" else int(v) * width_mult


def parameter_registry(in_channel: int = 1, n_classes: int = 1, width_mult: int = 1
                       ) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (name, shape) list = the reference's ``state_dict()`` contract
    (registration order of SE_UNet.py:15-21,42,57-65,108-151)."""
    reg: List[Tuple[str, Tuple[int, ...]]] = []
    for name, kind, cin, cout, _dil, _up in TOPOLOGY:
        ci, co = _ch(cin, in_channel, width_mult), _ch(cout, in_channel, width_mult)
        if kind == "cat":
            reg.append((f"{name}.conv1.weight", (co, ci, 1, 1, 1)))
            continue
        reg.append((f"{name}.conv1.weight", (co, ci, 3, 3, 3)))
        reg.append((f"{name}.conv1.bias", (co,)))
        reg.append((f"{name}.conv2.weight", (2, co, 1, 1, 1)))
        reg.append((f"{name}.conv2.bias", (2,)))
        reg.append((f"{name}.conv_se.weight", (1, co, 1, 1, 1)))
        if kind == "g2":
            reg.append((f"{name}.conv_se2.weight", (1, co, 1, 1, 1)))
    reg.append(("dc0_0.weight", (n_classes, 24, 1, 1, 1)))
    reg.append(("dc0_0.bias", (n_classes,)))
    reg.append(("dc0_1.weight", (n_classes, 12, 1, 1, 1)))
    reg.append(("dc0_1.bias", (n_classes,)))
    return reg


# --------------------------------------------------------------------------
# deterministic weights / inputs (build-owned; a pure function of name + seed so
# that no 6 MB weight fixture has to be committed -- SURVEY.md section 7 step 1)
# --------------------------------------------------------------------------
def _gen(tag: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(tag.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def deterministic_state_dict(in_channel: int = 2, n_classes: int = 1, width_mult: int = 1,
                             seed: int = 0, bias_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """PyTorch-default-style init U(+-1/sqrt(fan_in)) from a
    seeded CPU generator keyed by the tensor name."""
    sd = {}
    for name, shape in parameter_registry(in_channel, n_classes, width_mult):
        if name.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
        else:  # bias: fan_in of the matching weight
            wshape = dict(parameter_registry(in_channel, n_classes, width_mult))[
                name[:-4] + "weight"]
            fan_in = int(np.prod(wshape[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        u = torch.rand(shape, generator=_gen(name, seed), dtype=torch.float32)
        t = (2.0 * u - 1.0) * bound
        if name.endswith("bias"):
            t = t * bias_scale
        sd[name] = t
    return sd


def synthetic_batch(batch: int, size: Sequence[int], in_channel: int = 2, seed: int = 0
                    ) -> Dict[str, torch.Tensor]:
    """Synthetic patch batch: image channels ~ U[0,1) (the
    two HU windows of prediction.py:39-49 both land in [0,1]); label = Bernoulli(0.03)
    mask; weight = 1 + U[0,1)*label; skeleton = label * Bernoulli(0.1)."""
    d, h, w = size
    g = _gen("batch", seed)
    image = torch.rand((batch, in_channel, d, h, w), generator=g)
    label = (torch.rand((batch, 1, d, h, w), generator=g) < 0.03).float()
    weight = 1.0 + torch.rand((batch, 1, d, h, w), generator=g) * label
    skel = label * (torch.rand((batch, 1, d, h, w), generator=g) < 0.1).float()
    return {"image": image, "label": label, "weight": weight, "skel": skel}


def drop_scale_from_uniform(r: torch.Tensor, channel_num: int, thr: float = DROP_THRESHOLD
                            ) -> torch.Tensor:
    """DropLayer scale tensor from the uniform draw ``r`` of shape (B, C, 1, 1, 1)
    (reference SE_UNet.py:89-95): keep-mask = r >= thr, then multiplied by
    C / (sum over batch AND channels of the mask + 0.01)."""
    keep = (r >= thr).to(torch.float32)
    return keep * channel_num / (keep.sum() + 0.01)


# --------------------------------------------------------------------------
# the network, restated functionally
# --------------------------------------------------------------------------
def _inorm(t: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    # nn.InstanceNorm3d defaults: affine=False, track_running_stats=False, biased var
    # (reference SE_UNet.py:17,43,59)
    return F.instance_norm(t, eps=eps)


def _up(t: torch.Tensor, factor: int) -> torch.Tensor:
    # nn.Upsample(scale_factor, mode='trilinear', align_corners=True); factor 1 is the
    # identity but is still executed by the reference (SE_UNet.py:19,34)
    return F.interpolate(t, scale_factor=factor, mode="trilinear", align_corners=True)


class OracleSEUNet(nn.Module):
    """Restatement of reference ``SE_UNet`` (SE_UNet.py:99-238) with identical
    ``state_dict`` keys/shapes/order.  Extras the reference lacks, all defaulting to
    the reference behaviour: ``width_mult``, ``negative_slope`` (D1),
    a device-agnostic DropLayer with injectable scale tensors (Q6)."""

    def __init__(self, in_channel: int = 1, n_classes: int = 1, width_mult: int = 1,
                 negative_slope: float = 0.01):
        super().__init__()
        self.in_channel, self.n_classes = in_channel, n_classes
        self.width_mult, self.negative_slope = width_mult, negative_slope
        for name, kind, cin, cout, dil, _upf in TOPOLOGY:
            ci, co = _ch(cin, in_channel, width_mult), _ch(cout, in_channel, width_mult)
            blk = nn.Module()
            if kind == "cat":
                blk.conv1 = nn.Conv3d(ci, co, 1, bias=False)
            else:
                blk.conv1 = nn.Conv3d(ci, co, 3, padding=dil, dilation=dil, bias=True)
                blk.conv2 = nn.Conv3d(co, 2, 1, bias=True)
                blk.conv_se = nn.Conv3d(co, 1, 1, bias=False)
                if kind == "g2":
                    blk.conv_se2 = nn.Conv3d(co, 1, 1, bias=False)
            setattr(self, name, blk)
        self.dc0_0 = nn.Conv3d(24, n_classes, 1, bias=True)
        self.dc0_1 = nn.Conv3d(12, n_classes, 1, bias=True)
        self._kinds = {n: (k, d, u) for n, k, _ci, _co, d, u in TOPOLOGY}

    # -- blocks -----------------------------------------------------------
    def _gated(self, name: str, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """SSEConv.forward / SSEConv2.forward (SE_UNet.py:24-35, 68-82)."""
        kind, _dil, upf = self._kinds[name]
        blk = getattr(self, name)
        e = F.leaky_relu(_inorm(blk.conv1(t)), self.negative_slope)
        e = e * torch.sigmoid(blk.conv_se(e))
        if kind == "g2":   # second gate is computed from the already gated tensor
            e = e * torch.sigmoid(blk.conv_se2(e))
        side = _up(blk.conv2(e), upf)
        return e, side

    def _cat(self, name: str, t: torch.Tensor) -> torch.Tensor:
        """CATConv.forward (SE_UNet.py:45-49)."""
        return F.leaky_relu(_inorm(getattr(self, name).conv1(t)), self.negative_slope)

    # -- whole graph --------------------------------------------------------
    def forward(self, x: torch.Tensor, drop1: Optional[torch.Tensor] = None,
                drop2: Optional[torch.Tensor] = None):
        """SE_UNet.forward (SE_UNet.py:181-238).  ``drop1``/``drop2`` are DropLayer
        scale tensors of shape (B,24,1,1,1)/(B,12,1,1,1); None means: eval mode ->
        identity, train mode -> drawn from the CPU generator exactly in the
        reference's order (dropout1 then dropout2, SE_UNet.py:91,232-233)."""
        sides = {}
        pool = lambda t: F.max_pool3d(t, 2, 2)

        e0, sides["ec1"] = self._gated("ec1", x)
        e1, sides["ec2"] = self._gated("ec2", e0)
        e1_1, sides["ec3"] = self._gated("ec3", e1)
        e1 = self._cat("ec33", torch.cat((e1_1, e0, e1), 1)) + self._cat("x33", x)
        e2, x1 = pool(e1), pool(x)

        e2, sides["ec4"] = self._gated("ec4", e2)
        e3, sides["ec5"] = self._gated("ec5", e2)
        e3_1, sides["ec6"] = self._gated("ec6", e3)
        e3 = self._cat("ec63", torch.cat((e3_1, e2, e3), 1)) + self._cat("x63", x1)
        e4, x2 = pool(e3), pool(x1)

        e4, sides["ec7"] = self._gated("ec7", e4)
        e5, sides["ec8"] = self._gated("ec8", e4)
        e5_1, sides["ec9"] = self._gated("ec9", e5)
        e5 = self._cat("ec93", torch.cat((e5_1, e4, e5), 1)) + self._cat("x93", x2)
        e6 = pool(e5)

        e6, sides["ec10"] = self._gated("ec10", e6)
        e7, sides["ec11"] = self._gated("ec11", e6)
        e7_1, sides["ec12"] = self._gated("ec12", e7)
        e7 = self._cat("ec123", torch.cat((e7_1, e6, e7), 1))

        d0, sides["dc1"] = self._gated("dc1", torch.cat((_up(e7, 2), e5), 1))
        d0_1, sides["dc2"] = self._gated("dc2", d0)
        d0 = self._cat("dc22", torch.cat((d0_1, d0), 1))

        d1, sides["dc3"] = self._gated("dc3", torch.cat((_up(d0, 2), e3), 1))
        d1_1, sides["dc4"] = self._gated("dc4", d1)
        d1 = self._cat("dc42", torch.cat((d1_1, d1), 1))

        d2, sides["dc5"] = self._gated("dc5", torch.cat((_up(d1, 2), e1), 1))
        d2_1, sides["dc6"] = self._gated("dc6", d2)
        # dc62 (SE_UNet.py:230) is dead: its output feeds nothing; it is
        # not evaluated here, and its weight receives no gradient, like the reference.

        enc = torch.cat([sides[n] for n in ENCODER_SIDE_BLOCKS], 1)
        dec = torch.cat([sides[n] for n in DECODER_SIDE_BLOCKS], 1)
        b = x.shape[0]
        if self.training:
            if drop1 is None:
                drop1 = drop_scale_from_uniform(torch.rand(b, 24, 1, 1, 1), 24)
            if drop2 is None:
                drop2 = drop_scale_from_uniform(torch.rand(b, 12, 1, 1, 1), 12)
        if drop1 is not None:
            enc = enc * drop1.to(enc.dtype)
        if drop2 is not None:
            dec = dec * drop2.to(dec.dtype)
        return self.dc0_0(enc), self.dc0_1(dec)


def get_model():
    """reference SE_UNet.py:240-242: ``({}, SE_UNet(in_channel=2))``."""
    return {}, OracleSEUNet(in_channel=2)


# --------------------------------------------------------------------------
# losses (reference train.py:51-76)
# --------------------------------------------------------------------------
def dice_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """train.py:51-57: soft Dice over the whole flattened batch, smooth = 1."""
    p, t = pred.reshape(-1), target.reshape(-1)
    inter = (p * t).sum()
    return 1 - (2.0 * inter + 1.0) / (p.sum() + t.sum() + 1.0)


def general_union_loss_lib(pred: torch.Tensor, target: torch.Tensor, weight: torch.Tensor
                           ) -> torch.Tensor:
    """train.py:59-68: alpha = 0.2, exponent 0.7, sigma 1e-4 added to pred for both
    classes, voxel weight map multiplies numerator and denominator."""
    alpha, sigma = 0.2, 1e-4
    num = (weight * (pred + (target * sigma + (1 - target) * sigma)) ** 0.7 * target).sum()
    den = (weight * (alpha * pred + (1 - alpha) * target)).sum()
    return 1 - (num + 1.0) / (den + 1.0)


def atr_loss(pred: torch.Tensor, target: torch.Tensor, skel: torch.Tensor,
             weight: torch.Tensor) -> torch.Tensor:
    """train.py:70-76: ``target`` is ignored (overwritten by ``skel``), prediction is
    masked by the skeleton."""
    p = pred * skel
    num = (weight * p * skel).sum()
    den = (weight * (p + skel)).sum()
    return 1 - (num + 1.0) / (den + 1.0)


def stage_loss(stage: int, pred_en, pred_de, label, weight=None, skel=None):
    """Loss combination of the three training stages; inputs are the raw logits.
    stage 1: train.py:595-599 ; stage 2: train.py:429-435 ; stage 3: train.py:235-243."""
    pe, pd = torch.sigmoid(pred_en), torch.sigmoid(pred_de)
    if stage == 1:
        return dice_loss(pd, label) + dice_loss(pe, label)
    gul = general_union_loss_lib(pd, label, weight) + 0.5 * general_union_loss_lib(pe, label, weight)
    if stage == 2:
        return gul
    return gul + 0.5 * (atr_loss(pe, label, skel, weight) + atr_loss(pd, label, skel, weight))


# --------------------------------------------------------------------------
# whole-volume sliding-window inference (reference prediction.py:39-49, 65-109)
# --------------------------------------------------------------------------
def two_channel(hu: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """prediction.py:39-49: channel 0 = clip(HU,-1024,1024) -> [0,1];
    channel 1 = clip(HU,-1000,500) -> [0,1] (float64 like the reference)."""
    hu = hu.astype(float)
    c1 = (np.clip(hu, -1000, 500) + 1000) / 1500
    c0 = (np.clip(hu, -1024, 1024) + 1024) / 2048
    return c0, c1


def window_starts(dim: int, cube: int = 128, step: int = 64) -> List[int]:
    """prediction.py:80-100 / data.py:738-761: starts step*i; the last window is
    shifted back to end at ``dim`` instead of being padded."""
    if dim < cube:
        raise ValueError("axis shorter than the window is unsupported (SURVEY Q9)")
    n = (dim - cube) // step + 1 if (dim - cube) % step == 0 else (dim - cube) // step + 2
    out = []
    for i in range(n):
        lo = step * i
        if lo + cube > dim:
            lo = dim - cube
        out.append(lo)
    return out


def sliding_window_predict(model: nn.Module, x: torch.Tensor, cube: int = 128, step: int = 64, sigmoid: bool = True
                           ) -> np.ndarray:
    """prediction.py:78-109: per window ``sigmoid(pred1)`` accumulated into float64
    host buffers with an overlap count, then divided.  ``x`` is (1,C,X,Y,Z).
    ``sigmoid=False``: save_gradients.py:129-137 / weight_br.py:95-102, the same loop on the RAW
    logits (``pred += p_numpy``; those scripts run under ``case_net.train()``: the caller sets the mode)."""
    _, _, X, Y, Z = x.shape
    acc = np.zeros((1, 1, X, Y, Z))
    cnt = np.zeros((1, 1, X, Y, Z))
    with torch.no_grad():
        for xl in window_starts(X, cube, step):
            for yl in window_starts(Y, cube, step):
                for zl in window_starts(Z, cube, step):
                    _, p = model(x[:, :, xl:xl + cube, yl:yl + cube, zl:zl + cube])
                    p = (torch.sigmoid(p) if sigmoid else p).cpu().numpy()
                    acc[:, :, xl:xl + cube, yl:yl + cube, zl:zl + cube] += p
                    cnt[:, :, xl:xl + cube, yl:yl + cube, zl:zl + cube] += 1
    return np.squeeze(acc / cnt)


def validation_window_table(shape: Sequence[int], batch: int, cube: int = 128, step: int = 64
                            ) -> List[Tuple[int, int, int, int, int, int]]:
    """SegValCropData.crop_pos (data.py:731-773): [xl, xr, yl, yr, zl, zr] for every window, x outer / z inner, then
    copies of the FIRST window appended until the count is a multiple of the batch size (data.py:764-765)."""
    tmp = []
    for xl in window_starts(shape[0], cube, step):
        for yl in window_starts(shape[1], cube, step):
            for zl in window_starts(shape[2], cube, step):
                tmp.append((xl, xl + cube, yl, yl + cube, zl, zl + cube))
    while len(tmp) % batch != 0:
        tmp.append(tmp[0])
    return tmp


def sliding_window_validate(model: nn.Module, x: torch.Tensor, batch: int, cube: int = 128, step: int = 64) -> np.ndarray:
    """The validation / test form of the loop (train.py:682-693; test.py:151-161): batches of ``batch`` windows in table
    order (DataLoader(batch_size=batch, shuffle=False), train.py:181-185), ``pred[...] += sigmoid(p)[i]`` and
    ``pred_num[...] += 1`` for every entry of the padded table -- the copies of window 0 included -- then the division.
    The caller sets the mode; the reference runs it under ``model.train()`` (train.py:632), DropLayer active."""
    _, _, X, Y, Z = x.shape
    pos = validation_window_table((X, Y, Z), batch, cube, step)
    pred = np.zeros((1, 1, X, Y, Z))
    pred_num = np.zeros(pred.shape)
    with torch.no_grad():
        for i in range(0, len(pos), batch):
            chunk = pos[i:i + batch]
            xb = torch.cat([x[:, :, xl:xr, yl:yr, zl:zr] for xl, xr, yl, yr, zl, zr in chunk], 0)
            _, p = model(xb)
            p = torch.sigmoid(p).cpu().numpy()
            for k, (xl, xr, yl, yr, zl, zr) in enumerate(chunk):
                pred[0, :, xl:xr, yl:yr, zl:zr] += p[k]
                pred_num[0, :, xl:xr, yl:yr, zl:zr] += 1
    return np.squeeze(pred / pred_num)


def build_oracle(in_channel=2, n_classes=1, width_mult=1, seed=0, train=False) -> OracleSEUNet:
    m = OracleSEUNet(in_channel, n_classes, width_mult)
    m.load_state_dict(deterministic_state_dict(in_channel, n_classes, width_mult, seed))
    return m.train(train)
