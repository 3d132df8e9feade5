"""Host-side mirror of the reference's ``SE_UNet.py`` module surface, backed by the gfx950 HIP
library (``libseunet_hip.so``) through its C ABI.

Same constructor, attribute names, ``state_dict`` (117 tensors), ``forward(x) ->
(pred0, pred1)`` logits contract and ``get_model()`` as the reference (SE_UNet.py:99-153,181-242),
so ``train.py`` / ``prediction.py`` style callers work unchanged:

    model = SE_UNet(in_channel=2, n_classes=1).cuda()
    pred_en, pred_de = model(data)              # HIP kernels, differentiable
    loss = dice_loss(torch.sigmoid(pred_de), label) + dice_loss(torch.sigmoid(pred_en), label)
    loss.backward(); optimizer.step()

The whole forward (and the whole backward) is ONE call into the library
(``seunet_net_forward`` / ``seunet_net_backward``): the graph walk, workspace layout and all kernel
launches are native.  There is no PyTorch or CPU fallback: a CPU tensor or a missing ``.so`` raises.

``n_classes > 1`` (SE_UNet.py:100,150-151; no reference caller uses it) runs the heads on a general path (csrc/classes.hip: side
maps materialised, level maps per class) -- same results, more bytes than the fused single-class form.

Extras over the reference, all defaulting to its behaviour: ``width_mult``,
``negative_slope`` (D1), ``act_dtype`` ('bf16' performance mode / 'fp16' / 'fp32' 1e-3-parity mode) and a
device-agnostic ``DropLayer`` with the reference's CPU-generator RNG order (Q6).

fp16 storage (BASELINE configs[4]) needs what the reference never had: the Dice gradients at initialisation are ~1e-8
per voxel, below half precision's normal range, so the activation gradients are carried multiplied by a STATIC
``loss_scale`` (default 65536 for fp16, 1 otherwise; a build-side extension, attribute ``model.loss_scale``): the incoming
logit gradients are multiplied by it, the parameter gradients divided by it, nothing else changes.  A backward pass whose
scaled gradients overflow (inf / NaN in the flat gradient buffer) yields zero gradients and increments the device counter
``model.overflow_steps`` (no host synchronisation; lower ``model.loss_scale`` when it is ever non-zero).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib

config = {}   # reference SE_UNet.py:7

# Test hook: byte value every freshly allocated workspace / output / gradient buffer is filled with before the library is called
# (None = leave torch.empty's contents).  The library must never read a byte it has not written in the same pass: a run over
# 0xFF-filled buffers (NaN patterns in every storage type) has to give the bits of a run over zero-filled ones
# (tests/test_net_gpu.py::test_results_do_not_depend_on_the_prior_contents_of_the_workspace).
_DEBUG_FILL: Optional[int] = None


def _fresh(shape, dtype, device):
    t = torch.empty(shape, dtype=dtype, device=device)
    if _DEBUG_FILL is not None:
        t.view(torch.uint8).fill_(_DEBUG_FILL)
    return t


def _default_dtype() -> str:
    return os.environ.get("SEUNET_DTYPE", "bf16")


def _default_conv_impl() -> int:
    return _lib.CONV_NAIVE if os.environ.get("SEUNET_CONV_IMPL", "mfma").lower() == "naive" else _lib.CONV_MFMA


# ----------------------------------------------------------------------------------------------
# parameter containers with the reference's attribute names (SE_UNet.py:9-82)
# ----------------------------------------------------------------------------------------------
class SSEConv(nn.Module):
    """3x3x3 conv -> InstanceNorm -> LeakyReLU -> one spatial gate; side conv C->2 (SE_UNet.py:9-35)."""
    n_gates = 1

    def __init__(self, in_channel=1, out_channel1=1, out_channel2=2, stride=1, kernel_size=3,
                 padding=1, dilation=1, down_sample=1, bias=True):
        super().__init__()
        if stride != 1 or kernel_size != 3 or padding != 1 or out_channel2 != 2 or not bias:
            raise NotImplementedError("HIP path implements the configuration SE_UNet uses: k=3, stride 1, "
                                      "padding=dilation, 2 side channels, bias")
        self.in_channel, self.out_channel = in_channel, out_channel1
        self.dilation, self.down_sample = dilation, down_sample
        self.conv1 = nn.Conv3d(in_channel, out_channel1, 3, padding=dilation, dilation=dilation, bias=True)
        self.conv2 = nn.Conv3d(out_channel1, 2, 1, bias=True)
        self.conv_se = nn.Conv3d(out_channel1, 1, 1, bias=False)
        self.act = nn.LeakyReLU(inplace=True)     # (reference attribute SE_UNet.py:18; its slope is what the fused epilogue uses)
        if self.n_gates == 2:
            self.conv_se2 = nn.Conv3d(out_channel1, 1, 1, bias=False)

    def forward(self, x):
        from .ops import gated_block_forward
        return gated_block_forward(self, x)


class SSEConv2(SSEConv):
    """As SSEConv with two sequential gates (SE_UNet.py:51-82)."""
    n_gates = 2


class CATConv(nn.Module):
    """1x1x1 conv (no bias) -> InstanceNorm -> LeakyReLU (SE_UNet.py:37-49)."""

    def __init__(self, in_channel=1, out_channel1=1):
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel1
        self.conv1 = nn.Conv3d(in_channel, out_channel1, 1, bias=False)
        self.act = nn.LeakyReLU(inplace=True)     # (SE_UNet.py:44)

    def forward(self, x):
        from .ops import cat_block_forward
        return cat_block_forward(self, x)


class DropLayer(nn.Module):
    """Per-(sample, channel) keep mask with the batch-coupled rescale of SE_UNet.py:84-97.

    ``scale(batch)`` draws ``torch.rand(B, C, 1, 1, 1)`` from the CPU generator exactly like the
    reference (which then hard-codes ``.cuda()``); the scale tensor is applied inside the fused head
    kernel, so this module never touches the big activation."""

    def __init__(self, channel_num=1, thr=0.3):
        super().__init__()
        self.channel_num, self.threshold = channel_num, thr

    def scale(self, batch: int) -> torch.Tensor:
        r = torch.rand(batch, self.channel_num, 1, 1, 1)
        keep = (r >= self.threshold).to(torch.float32)
        return keep * self.channel_num / (keep.sum() + 0.01)

    def forward(self, x):
        if not self.training:
            return x
        return x * self.scale(x.shape[0]).to(x.device, x.dtype)


# ----------------------------------------------------------------------------------------------
# whole-network autograd function
# ----------------------------------------------------------------------------------------------
def make_desc(batch, in_channel, n_classes, d, h, w, width_mult, dtype_code, conv_impl, slope, eps=1e-5):
    return _lib.NetDesc(batch, in_channel, n_classes, d, h, w, width_mult, dtype_code, conv_impl, slope, eps)


def registry(desc: _lib.NetDesc) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) list as the native library lays the parameters out."""
    lib = _lib.load()
    out = []
    n = lib.seunet_net_param_count(C.byref(desc))
    buf = C.create_string_buffer(64)
    shape = (C.c_int * 5)()
    nd = C.c_int()
    for i in range(n):
        _lib.check(lib.seunet_net_param_info(C.byref(desc), i, buf, 64, shape, C.byref(nd)), "param_info")
        out.append((buf.value.decode(), tuple(shape[k] for k in range(nd.value))))
    return out


def _is_decoder(name: str) -> bool:
    return name.startswith("dc") and not name.startswith("dc0_")


def alloc_flat_grads(params, dead, device, names=None):
    """One contiguous f32 buffer for the gradients of all LIVE parameters plus per-parameter views of it (``None`` for the
    dead ``dc62`` block, which takes no space).  Layout: the encoder blocks and the two heads first, the decoder blocks
    (dc1 .. dc6, dc22, dc42) last -- the backward pass finishes the decoder's gradients first, so a data-parallel step
    reduces the tail of the buffer while the encoder is still being differentiated and the head of it afterwards
    (``ddp.GradSync``); without names: registry order.  Returns (flat, views in parameter order, first decoder element)."""
    idx = list(range(len(params)))
    if names is not None:
        idx = [i for i in idx if not _is_decoder(names[i])] + [i for i in idx if _is_decoder(names[i])]
    sizes = {i: (0 if dead[i] else params[i].numel()) for i in idx}
    flat = _fresh(sum(sizes.values()), torch.float32, device)
    grads, off, split = [None] * len(params), 0, None
    for i in idx:
        if names is not None and split is None and _is_decoder(names[i]):
            split = off
        if not dead[i]:
            grads[i] = flat[off:off + sizes[i]].view(params[i].shape)
        off += sizes[i]
    return flat, grads, (off if split is None else split)


class _SEUNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop1, drop2, meta, *params):
        lib = _lib.load()
        b, _, d, h, w = x.shape
        desc = make_desc(b, meta["in_channel"], meta["n_classes"], d, h, w, meta["width_mult"],
                         meta["dtype"], meta["conv_impl"], meta["negative_slope"])
        with torch.cuda.device(x.device):       # launch on x's GPU and stream even when it is not the current device
            ws_bytes = lib.seunet_net_workspace_bytes(C.byref(desc))
            if ws_bytes == 0:
                raise RuntimeError("libseunet_hip net_workspace_bytes: " + _lib.last_error())
            ws = _fresh(ws_bytes, torch.uint8, x.device)
            pred0 = _fresh((b, meta["n_classes"], d, h, w), torch.float32, x.device)
            pred1 = _fresh((b, meta["n_classes"], d, h, w), torch.float32, x.device)
            plist = [p.detach().contiguous() for p in params]
            for p in plist:
                if p.device != x.device:
                    raise RuntimeError(f"SE_UNet parameters are on {p.device} but the input is on {x.device}")
            parr = _lib.ptr_array(plist)
            _lib.check(lib.seunet_net_forward(C.byref(desc), parr, x.data_ptr(), _lib.ptr(drop1), _lib.ptr(drop2),
                                              pred0.data_ptr(), pred1.data_ptr(), ws.data_ptr(), ws_bytes,
                                              _lib.stream_ptr()), "net_forward")
        ctx.desc, ctx.ws, ctx.ws_bytes = desc, ws, ws_bytes
        ctx.plist, ctx.drop = plist, (drop1, drop2)
        ctx.dead = meta["dead"]
        ctx.loss_scale = float(meta.get("loss_scale", 1.0))
        ctx.overflow = meta.get("overflow")
        ctx.names, ctx.grad_sync = meta.get("names"), meta.get("grad_sync")
        return pred0, pred1

    @staticmethod
    def backward(ctx, g0, g1):
        if ctx.ws is None:
            raise RuntimeError("SE_UNet (HIP path): the backward pass of this forward call has already run; it consumes "
                               "the saved workspace, so a second backward (retain_graph=True, or two losses "
                               "back-propagated separately) is not supported -- sum the losses and call backward once")
        lib = _lib.load()
        dev = ctx.ws.device
        shape = (ctx.desc.batch, ctx.desc.n_classes, ctx.desc.d, ctx.desc.h, ctx.desc.w)
        with torch.cuda.device(dev):
            g0 = torch.zeros(shape, dtype=torch.float32, device=dev) if g0 is None else g0.contiguous().float()
            g1 = torch.zeros(shape, dtype=torch.float32, device=dev) if g1 is None else g1.contiguous().float()
            if ctx.loss_scale != 1.0:       # fp16 storage: keep the activation gradients inside half precision's range
                g0, g1 = g0 * ctx.loss_scale, g1 * ctx.loss_scale
            # all live parameter gradients are views of ONE flat buffer (one RCCL all-reduce under data parallelism)
            flat, grads, split = alloc_flat_grads(ctx.plist, ctx.dead, dev, ctx.names)
            garr = _lib.ptr_array(grads)
            parr = _lib.ptr_array(ctx.plist)
            sync = ctx.grad_sync
            ev = sync.decoder_event() if sync is not None else None
            _lib.check(lib.seunet_net_backward_ev(C.byref(ctx.desc), parr, g0.data_ptr(), g1.data_ptr(),
                                                  _lib.ptr(ctx.drop[0]), _lib.ptr(ctx.drop[1]), garr,
                                                  ctx.ws.data_ptr(), ctx.ws_bytes, _lib.stream_ptr(),
                                                  None if ev is None else ev.cuda_event), "net_backward")
            if sync is not None:      # data parallel: decoder bucket on the side stream (already under way), the rest here
                # (with a loss scale the SCALED buffer is what is summed over the ranks; the finiteness test below then runs on the
                # global sum: an inf / NaN on any one rank is an inf / NaN in every rank's sum, so all ranks drop the step together)
                sync.exchange(flat, split, ev)
            if ctx.loss_scale != 1.0:
                # a scaled activation gradient beyond half precision's range turns into inf / NaN in the flat buffer: such a step
                # is dropped (zero gradients) and counted on the device, with no host synchronisation -- the caller reads
                # ``model.overflow_steps`` when it wants to (and lowers ``model.loss_scale`` if it ever becomes non-zero).
                # Without ``grad_sync`` (``ddp.allreduce_gradients`` after the backward) the decision is per rank: a rank that
                # overflowed contributes zeros to the sum; the ranks still apply the same update.
                ok = torch.isfinite(flat).all()
                flat.copy_(torch.where(ok, flat * (1.0 / ctx.loss_scale), torch.zeros((), dtype=flat.dtype, device=dev)))
                if ctx.overflow is not None:
                    ctx.overflow.add_((~ok).to(ctx.overflow.dtype))
        ctx.ws = None
        return (None, None, None, None) + tuple(grads)


class SE_UNet(nn.Module):
    """Drop-in for the reference ``SE_UNet`` (SE_UNet.py:99-238) on MI355X."""

    def __init__(self, in_channel=1, n_classes=1, width_mult=1, negative_slope=0.01,
                 act_dtype: Optional[str] = None, conv_impl: Optional[int] = None):
        super().__init__()
        self.in_channel, self.n_classes = in_channel, n_classes
        self.width_mult, self.negative_slope = width_mult, negative_slope
        self.act_dtype = act_dtype or _default_dtype()
        self.conv_impl = _default_conv_impl() if conv_impl is None else conv_impl
        self.loss_scale = 65536.0 if _lib.dtype_code(self.act_dtype) == _lib.F16 else 1.0
        self.overflow_steps = None    # device counter of backward passes dropped because a scaled gradient left half precision's range
        self.batchnorm, self.bias, self.out_channel2, self.sigmoid_output = False, True, 2, 0
        m = width_mult
        # registration order == reference SE_UNet.py:108-153 (state_dict / parameters() order)
        self.ec1 = SSEConv(in_channel, 8 * m)
        self.ec2 = SSEConv(8 * m, 16 * m)
        self.ec3 = SSEConv(16 * m, 32 * m, dilation=2)
        self.ec33 = CATConv(56 * m, 32 * m)
        self.x33 = CATConv(in_channel, 32 * m)
        self.ec4 = SSEConv2(32 * m, 32 * m, down_sample=2)
        self.ec5 = SSEConv2(32 * m, 32 * m, dilation=2, down_sample=2)
        self.ec6 = SSEConv2(32 * m, 64 * m, dilation=2, down_sample=2)
        self.ec63 = CATConv(128 * m, 64 * m)
        self.x63 = CATConv(in_channel, 64 * m)
        self.ec7 = SSEConv2(64 * m, 64 * m, down_sample=4)
        self.ec8 = SSEConv2(64 * m, 64 * m, dilation=2, down_sample=4)
        self.ec9 = SSEConv2(64 * m, 64 * m, dilation=2, down_sample=4)
        self.ec93 = CATConv(192 * m, 64 * m)
        self.x93 = CATConv(in_channel, 64 * m)
        self.ec10 = SSEConv2(64 * m, 64 * m, down_sample=8)
        self.ec11 = SSEConv2(64 * m, 64 * m, down_sample=8)
        self.ec12 = SSEConv2(64 * m, 64 * m, down_sample=8)
        self.ec123 = CATConv(192 * m, 64 * m)
        self.dc1 = SSEConv2(128 * m, 64 * m, down_sample=4)
        self.dc2 = SSEConv2(64 * m, 64 * m, down_sample=4)
        self.dc22 = CATConv(128 * m, 64 * m)
        self.dc3 = SSEConv2(128 * m, 64 * m, down_sample=2)
        self.dc4 = SSEConv2(64 * m, 32 * m, down_sample=2)
        self.dc42 = CATConv(96 * m, 32 * m)
        self.dc5 = SSEConv(64 * m, 32 * m, down_sample=1)
        self.dc6 = SSEConv(32 * m, 16 * m, down_sample=1)
        self.dc62 = CATConv(48 * m, 16 * m)     # dead in forward (SE_UNet.py:230); kept for state_dict
        self.dc0_0 = nn.Conv3d(24, n_classes, 1, bias=True)
        self.dc0_1 = nn.Conv3d(12, n_classes, 1, bias=True)
        self.dropout1 = DropLayer(channel_num=24, thr=0.3)
        self.dropout2 = DropLayer(channel_num=12, thr=0.3)
        for m in self.modules():                 # the blocks' own activation modules carry this network's slope
            if isinstance(getattr(m, "act", None), nn.LeakyReLU):
                m.act.negative_slope = float(negative_slope)
        self._names = [n for n, _ in self.named_parameters()]
        self._dead = [n.startswith("dc62.") for n in self._names]
        self._registry_checked = False
        self._arena = {}              # inference workspaces by (shape, dtype, device): predict_logits()

    def _check_registry(self, desc):
        if self._registry_checked:
            return
        native = registry(desc)
        mine = [(n, tuple(p.shape)) for n, p in self.named_parameters()]
        if native != mine:
            raise RuntimeError("parameter registry of libseunet_hip differs from the nn.Module's state_dict")
        self._registry_checked = True

    @torch.no_grad()
    def forward_with_intermediates(self, x, blocks):
        """Diagnostic (eval-mode forward, no autograd): ``(pred0, pred1, {block: {"raw", "mean", "rstd", "out"}})`` for the
        named blocks -- the raw conv output, its InstanceNorm statistics and the block's output tensor as the kernels left
        them in the workspace (``seunet_net_read_tensor``).  tests/flip_census.py counts LeakyReLU-sign and max-pool-argmax
        disagreements with the float64 oracle from these."""
        lib = _lib.load()
        x = x.contiguous().float()
        b, _, d, h, w = x.shape
        desc = make_desc(b, self.in_channel, self.n_classes, d, h, w, self.width_mult, _lib.dtype_code(self.act_dtype),
                         self.conv_impl, self.negative_slope)
        level = {"ec1": 0, "ec2": 0, "ec3": 0, "ec33": 0, "dc5": 0, "dc6": 0, "ec4": 1, "ec5": 1, "ec6": 1, "ec63": 1, "dc3": 1,
                 "dc4": 1, "dc42": 1, "ec7": 2, "ec8": 2, "ec9": 2, "ec93": 2, "dc1": 2, "dc2": 2, "dc22": 2, "ec10": 3, "ec11": 3,
                 "ec12": 3, "ec123": 3, "x33": 0, "x63": 1, "x93": 2}   # (x-branches: raw / mean / rstd only)
        with torch.cuda.device(x.device):
            nbytes = lib.seunet_net_workspace_bytes(C.byref(desc))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            pred0 = torch.empty((b, self.n_classes, d, h, w), dtype=torch.float32, device=x.device)
            pred1 = torch.empty_like(pred0)
            plist = [p.detach().contiguous() for p in self.parameters()]
            parr = _lib.ptr_array(plist)
            _lib.check(lib.seunet_net_forward(C.byref(desc), parr, x.data_ptr(), None, None, pred0.data_ptr(),
                                              pred1.data_ptr(), ws.data_ptr(), nbytes, _lib.stream_ptr()), "net_forward")

            def read(name, which, buf, ch=None):
                _lib.check(lib.seunet_net_read_tensor(C.byref(desc), parr, ws.data_ptr(), nbytes, name.encode(), which, buf.data_ptr(),
                                                      ch, _lib.stream_ptr()), "net_read_tensor")
            out = {}
            for name in blocks:
                ch = C.c_int(0)
                tmp = torch.empty((b, 512), dtype=torch.float32, device=x.device)
                read(name, 1, tmp, C.byref(ch))
                c = ch.value
                rec = {"mean": tmp.reshape(-1)[:b * c].reshape(b, c).clone(), "rstd": torch.empty((b, c), dtype=torch.float32, device=x.device)}
                read(name, 2, rec["rstd"])
                lv = level[name]
                for key, which in (("raw", 0), ("out", 3)):
                    if which == 3 and name.startswith("x"):
                        continue
                    rec[key] = torch.empty((b, c, d >> lv, h >> lv, w >> lv), dtype=torch.float32, device=x.device)
                    read(name, which, rec[key])
                out[name] = rec
        return pred0, pred1, out

    @torch.no_grad()
    def predict_logits(self, x, out: Optional[torch.Tensor] = None):
        """Inference form of the forward (prediction.py:102-103 ``p0, p = model(x)`` keeps only ``p``): the decoder head's logits
        (B, n_classes, D, H, W), bit-identical to ``forward(x)[1]``; the encoder head, its twelve side convs and level maps are not
        evaluated (``seunet_net_forward`` with pred0 = NULL).  No autograd graph.  The workspace is a per-shape arena owned by
        the module and reused by every call (not a fresh 3-12 GB allocation per forward); ``release_arena()`` frees it."""
        if not x.is_cuda:
            raise RuntimeError("SE_UNet (HIP path) needs a tensor on an MI355X device; there is no CPU fallback")
        if x.dim() != 5 or x.shape[1] != self.in_channel:
            raise ValueError(f"expected input (B,{self.in_channel},D,H,W), got {tuple(x.shape)}")
        lib = _lib.load()
        x = x.contiguous().float()
        b, _, d, h, w = x.shape
        d1 = d2 = None
        if self.training:                   # RNG order: dropout1 then dropout2 (SE_UNet.py:232-233); the first draw is still made
            d1, d2 = self.dropout1.scale(b), self.dropout2.scale(b)
            d1 = d1.reshape(b, 24).to(x.device, torch.float32).contiguous()
            d2 = d2.reshape(b, 12).to(x.device, torch.float32).contiguous()
        desc = make_desc(b, self.in_channel, self.n_classes, d, h, w, self.width_mult, _lib.dtype_code(self.act_dtype),
                         self.conv_impl, self.negative_slope)
        self._check_registry(desc)
        key = (b, d, h, w, self.act_dtype, self.conv_impl, str(x.device))
        with torch.cuda.device(x.device):
            arena = self._arena.get(key)
            if arena is None:
                nbytes = lib.seunet_net_workspace_bytes(C.byref(desc))
                if nbytes == 0:
                    raise RuntimeError("libseunet_hip net_workspace_bytes: " + _lib.last_error())
                self._arena.clear()         # one shape at a time: a window loop uses one (plus a ragged last batch)
                arena = self._arena[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            if out is None:
                out = torch.empty((b, self.n_classes, d, h, w), dtype=torch.float32, device=x.device)
            plist = [p.detach().contiguous() for p in self.parameters()]
            _lib.check(lib.seunet_net_forward(C.byref(desc), _lib.ptr_array(plist), x.data_ptr(), _lib.ptr(d1), _lib.ptr(d2), None,
                                              out.data_ptr(), arena.data_ptr(), arena.numel(), _lib.stream_ptr()), "net_forward")
            arena.record_stream(torch.cuda.current_stream(x.device))
        return out

    def release_arena(self):
        self._arena.clear()

    def forward(self, x, drop_scales: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """x: (B, in_channel, D, H, W) float, D/H/W multiples of 8 -> (pred0, pred1) logits.
        ``drop_scales`` optionally injects the two DropLayer scale tensors (B,24,1,1,1)/(B,12,1,1,1)."""
        if not x.is_cuda:
            raise RuntimeError("SE_UNet (HIP path) needs a tensor on an MI355X device; there is no CPU fallback "
                               "(the CPU oracle lives in oracle/seunet_oracle.py and is test infrastructure).")
        if x.dim() != 5 or x.shape[1] != self.in_channel:
            raise ValueError(f"expected input (B,{self.in_channel},D,H,W), got {tuple(x.shape)}")
        if x.requires_grad:
            raise NotImplementedError("SE_UNet (HIP path) computes no gradient with respect to its input (no reference "
                                      "caller asks for one); detach() the input")
        x = x.contiguous().float()          # callers pass strided views
        b = x.shape[0]
        if drop_scales is not None:
            d1, d2 = drop_scales
        elif self.training:                 # RNG order: dropout1 then dropout2 (SE_UNet.py:232-233)
            d1, d2 = self.dropout1.scale(b), self.dropout2.scale(b)
        else:
            d1 = d2 = None
        if d1 is not None:
            d1 = d1.reshape(b, 24).to(x.device, torch.float32).contiguous()
            d2 = d2.reshape(b, 12).to(x.device, torch.float32).contiguous()
        meta = {"in_channel": self.in_channel, "n_classes": self.n_classes, "width_mult": self.width_mult,
                "dtype": _lib.dtype_code(self.act_dtype), "conv_impl": self.conv_impl,
                "negative_slope": float(self.negative_slope), "dead": self._dead, "loss_scale": float(self.loss_scale)}
        meta["names"], meta["grad_sync"] = self._names, getattr(self, "grad_sync", None)
        if self.loss_scale != 1.0:
            if self.overflow_steps is None or self.overflow_steps.device != x.device:
                self.overflow_steps = torch.zeros((), dtype=torch.int64, device=x.device)
            meta["overflow"] = self.overflow_steps
        if not self._registry_checked:
            self._check_registry(make_desc(b, self.in_channel, self.n_classes, x.shape[2], x.shape[3], x.shape[4],
                                           self.width_mult, meta["dtype"], self.conv_impl, self.negative_slope))
        return _SEUNetFunction.apply(x, d1, d2, meta, *self.parameters())


class CapturedForward:
    """The inference forward of ``model`` on FIXED buffers, recorded once as a HIP graph (``seunet_net_forward_capture``)
    and replayed per call: what the whole-volume window loops use (prediction.py:78-109 calls the network 343 times on a
    512^3 case).  Fill ``self.x`` (B, in_channel, D, H, W) -- and, under ``model.train()``, nothing else: ``__call__``
    draws the DropLayer scales like ``SE_UNet.forward`` -- then call; the logits land in ``self.pred0`` / ``self.pred1``
    (overwritten by the next call).  No autograd: the workspace is reused, so there is no backward of a replay.

    The graph holds raw pointers: the buffers live in this object, the parameters in the model.  In-place updates of the
    weights (``optimizer.step()``, ``load_state_dict``) are seen by the next replay; if a parameter tensor is REPLACED
    (``model.half()``, ``.to()``), the next call notices the changed pointers and records the graph again."""

    def __init__(self, model: "SE_UNet", batch: int, spatial: Tuple[int, int, int], decoder_only: bool = False):
        lib = _lib.load()
        self.model = model
        self.decoder_only = decoder_only    # inference form: the encoder head is not evaluated (pred0 stays unwritten)
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("CapturedForward needs the model on an MI355X device (no CPU fallback)")
        self.device = dev
        d, h, w = spatial
        self.desc = make_desc(batch, model.in_channel, model.n_classes, d, h, w, model.width_mult,
                              _lib.dtype_code(model.act_dtype), model.conv_impl, model.negative_slope)
        model._check_registry(self.desc)
        with torch.cuda.device(dev):
            self.ws_bytes = lib.seunet_net_workspace_bytes(C.byref(self.desc))
            if self.ws_bytes == 0:
                raise RuntimeError("libseunet_hip net_workspace_bytes: " + _lib.last_error())
            self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
            self.x = torch.zeros((batch, model.in_channel, d, h, w), dtype=torch.float32, device=dev)
            self.pred0 = torch.empty((batch, model.n_classes, d, h, w), dtype=torch.float32, device=dev)
            self.pred1 = torch.empty_like(self.pred0)
            self.drop1 = torch.ones((batch, 24), dtype=torch.float32, device=dev)
            self.drop2 = torch.ones((batch, 12), dtype=torch.float32, device=dev)
            self._stream = torch.cuda.Stream(device=dev)
            self._last = torch.cuda.Event()
        self._graphs = {}           # training flag -> (handle, parameter pointers, kept-alive tensors)

    def _record(self, training: bool):
        lib = _lib.load()
        plist = [p.detach().contiguous() for p in self.model.parameters()]
        parr = _lib.ptr_array(plist)
        d1 = self.drop1.data_ptr() if training else None
        d2 = self.drop2.data_ptr() if training else None
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        _lib.check(lib.seunet_init(self.device.index if self.device.index is not None else torch.cuda.current_device()), "init")
        with torch.cuda.stream(self._stream):
            args = (C.byref(self.desc), parr, self.x.data_ptr(), d1, d2, None if self.decoder_only else self.pred0.data_ptr(),
                    self.pred1.data_ptr(), self.ws.data_ptr(), self.ws_bytes, _lib.stream_ptr())
            _lib.check(lib.seunet_net_forward(*args), "net_forward")        # eager once: per-kernel one-time setup
            self._stream.synchronize()
            handle = C.c_void_p()
            _lib.check(lib.seunet_net_forward_capture(*args, C.byref(handle)), "net_forward_capture")
        cur.wait_stream(self._stream)
        old = self._graphs.get(training)
        if old is not None:
            self._destroy(old[0])
        self._graphs[training] = (handle, [p.data_ptr() for p in plist], plist)

    def __call__(self):
        lib = _lib.load()
        training = bool(self.model.training)
        with torch.cuda.device(self.device):
            if training:            # RNG order: dropout1 then dropout2 (SE_UNet.py:232-233)
                b = self.x.shape[0]
                d1, d2 = self.model.dropout1.scale(b), self.model.dropout2.scale(b)
                self.drop1.copy_(d1.reshape(b, 24), non_blocking=False)
                self.drop2.copy_(d2.reshape(b, 12), non_blocking=False)
            g = self._graphs.get(training)
            if g is None or g[1] != [p.data_ptr() for p in self.model.parameters()]:
                self._record(training)
                g = self._graphs[training]
            _lib.check(lib.seunet_graph_launch(g[0], _lib.stream_ptr()), "graph_launch")
            self._last.record(torch.cuda.current_stream(self.device))   # (a graph is destroyed only after its last replay has run)
        return self.pred0, self.pred1

    def _destroy(self, handle):
        # the host runs several replays ahead of the GPU: wait for the last launched replay before the executable graph goes
        self._last.synchronize()
        _lib.load().seunet_graph_destroy(handle)

    def __del__(self):
        try:
            for g in self._graphs.values():
                self._destroy(g[0])
        except Exception:
            pass


def load_reference_checkpoint(model: "SE_UNet", ckpt, strict: bool = False):
    """Load a checkpoint written by the reference's training loops into this model (train.py:322-324 saves
    ``model.module.state_dict()``; train.py:194-196 / test.py load with ``strict=False``).  ``ckpt``: a path or an already loaded
    state_dict.  Keys are the reference's own (same 117 tensors); a ``module.`` prefix -- a checkpoint saved from the
    ``DataParallel`` wrapper itself -- is stripped.  Returns torch's ``(missing_keys, unexpected_keys)`` record."""
    sd = torch.load(ckpt, map_location="cpu") if isinstance(ckpt, (str, bytes, os.PathLike)) else ckpt
    if isinstance(sd, dict) and "state_dict" in sd and all(not torch.is_tensor(v) for v in sd.values() if v is not sd["state_dict"]):
        sd = sd["state_dict"]
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)


def get_model():
    """reference SE_UNet.py:240-242."""
    net = SE_UNet(in_channel=2)
    return config, net
