"""Per-op Python wrappers over the C ABI (``include/seunet_hip.h``).

Used by the op/block-level parity tests and by the stand-alone ``SSEConv`` / ``SSEConv2`` /
``CATConv`` modules.  Tensors named ``*_cl`` are channels-last ``(N, D, H, W, C)`` torch tensors of
dtype float32 or bfloat16 (C a multiple of 8); everything else is the reference's NCDHW float32.
These wrappers run the HIP kernels only -- they are inference-style (no autograd graph); the
differentiable path is ``SE_UNet`` (one native forward / backward call for the whole network).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import Dims


def _tdtype(code: int):
    return {_lib.BF16: torch.bfloat16, _lib.F16: torch.float16}.get(code, torch.float32)


def _code(t: torch.Tensor) -> int:
    return {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}.get(t.dtype, _lib.F32)


def _dims_cl(t: torch.Tensor) -> Dims:
    n, d, h, w, _ = t.shape
    return Dims(n, d, h, w)


def _s():
    return _lib.stream_ptr()


# ---- layout ---------------------------------------------------------------------------------
def to_cl(x: torch.Tensor, dtype="bf16", c_pad: Optional[int] = None) -> torch.Tensor:
    code = dtype if isinstance(dtype, int) else _lib.dtype_code(dtype)
    x = x.contiguous().float()
    n, c, d, h, w = x.shape
    c_pad = c_pad or (c + 7) // 8 * 8
    out = torch.empty((n, d, h, w, c_pad), dtype=_tdtype(code), device=x.device)
    _lib.check(_lib.load().seunet_pack_cl(code, x.data_ptr(), c, out.data_ptr(), c_pad, Dims(n, d, h, w), _s()), "pack_cl")
    return out


def from_cl(t: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    n, d, h, w, cp = t.shape
    out = torch.empty((n, cp, d, h, w), dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().seunet_unpack_cl(_code(t), t.data_ptr(), cp, out.data_ptr(), Dims(n, d, h, w), _s()), "unpack_cl")
    return out if c is None else out[:, :c].contiguous()


# ---- convolution ------------------------------------------------------------------------------
def pack_weights(w: torch.Tensor, code: int, transpose_flip: bool = False) -> torch.Tensor:
    lib = _lib.load()
    co, ci = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3] * w.shape[4]
    ce_in, ce_out = (co, ci) if transpose_flip else (ci, co)
    nbytes = lib.seunet_conv_wpack_bytes(code, taps, ce_in, ce_out)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    w = w.contiguous().float()
    _lib.check(lib.seunet_conv_pack_weights(code, w.data_ptr(), taps, ci, co, int(transpose_flip), buf.data_ptr(), _s()),
               "conv_pack_weights")
    return buf


def conv3d(srcs: Sequence[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor] = None, dilation: int = 1,
           impl: int = _lib.CONV_MFMA, transpose_flip: bool = False, cin: Optional[int] = None,
           dsts: Optional[Sequence[Optional[torch.Tensor]]] = None, dst_channels: Optional[Sequence[int]] = None,
           accumulate: Optional[Sequence[int]] = None, want_stats: bool = False):
    """Convolution (or, with transpose_flip, its data gradient) of the channel concatenation of ``srcs``.
    Returns (list of destination tensors, stats_partial or None, slots)."""
    lib = _lib.load()
    code = _code(srcs[0])
    dims = _dims_cl(srcs[0])
    taps = weight.shape[2] * weight.shape[3] * weight.shape[4]
    co_w, ci_w = weight.shape[0], weight.shape[1]
    cin_e, cout_e = (co_w, ci_w) if transpose_flip else (ci_w, co_w)
    cin = cin or cin_e
    if dsts is None:
        dst_channels = dst_channels or [(cout_e + 7) // 8 * 8]
        dsts = [torch.empty(tuple(srcs[0].shape[:4]) + (c,), dtype=srcs[0].dtype, device=srcs[0].device) for c in dst_channels]
        accumulate = [0] * len(dsts)
    else:
        dst_channels = dst_channels or [t.shape[4] for t in dsts]
        accumulate = list(accumulate or [0] * len(dsts))
    wbuf = weight.contiguous().float() if impl == _lib.CONV_NAIVE else pack_weights(weight, code, transpose_flip)
    stats, slots = None, 0
    if want_stats:
        slots = lib.seunet_conv_stats_slots(impl, taps, dilation, dims)
        stats = torch.zeros((dims.n, slots, sum(dst_channels), 2), dtype=torch.float64, device=srcs[0].device)
    b = None if bias is None else bias.contiguous().float()
    _lib.check(lib.seunet_conv3d_fwd(code, impl, taps, dilation, len(srcs), _lib.ptr_array(list(srcs)),
                                     _lib.int_array([t.shape[4] for t in srcs]), cin, wbuf.data_ptr(), int(transpose_flip),
                                     _lib.ptr(b), len(dsts), _lib.ptr_array(list(dsts)), _lib.int_array(list(dst_channels)),
                                     _lib.int_array(list(accumulate)), _lib.ptr(stats), dims, _s()), "conv3d_fwd")
    return list(dsts), stats, slots


def conv3d_stream(src: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, dilation: int = 1,
                  transpose_flip: bool = False, dst: Optional[torch.Tensor] = None, dst_channels: Optional[int] = None,
                  accumulate: bool = False, want_stats: bool = False):
    """The streaming small-channel 3x3x3 convolution (csrc/conv_stream.hip): one bf16 source of 8 / 16 / 32 channels.
    Returns (destination, stats_partial or None, slots)."""
    lib = _lib.load()
    code, dims = _code(src), _dims_cl(src)
    co_w, ci_w = weight.shape[0], weight.shape[1]
    cout_e = ci_w if transpose_flip else co_w
    src_c = src.shape[4]
    dst_c = dst.shape[4] if dst is not None else (dst_channels or (cout_e + 7) // 8 * 8)
    if not lib.seunet_conv3d_stream_supported(code, dilation, src_c, dst_c):
        raise RuntimeError(f"conv3d_stream: {src_c} -> {dst_c} channels, dilation {dilation}, dtype {src.dtype} is not served by this kernel")
    wbuf = torch.empty(lib.seunet_conv3d_stream_wpack_bytes(src_c), dtype=torch.uint8, device=src.device)
    w = weight.contiguous().float()
    _lib.check(lib.seunet_conv3d_stream_pack(code, w.data_ptr(), ci_w, co_w, int(transpose_flip), src_c, dst_c, wbuf.data_ptr(), _s()),
               "conv3d_stream_pack")
    if dst is None:
        dst = torch.empty(tuple(src.shape[:4]) + (dst_c,), dtype=src.dtype, device=src.device)
    stats, slots = None, 0
    if want_stats:
        slots = lib.seunet_conv3d_stream_slots(dilation, dims)
        stats = torch.zeros((dims.n, slots, dst_c, 2), dtype=torch.float64, device=src.device)
    b = None
    if bias is not None:
        b = torch.zeros(dst_c, dtype=torch.float32, device=src.device)
        b[:bias.numel()] = bias.float()
    _lib.check(lib.seunet_conv3d_stream(code, dilation, src.data_ptr(), src_c, wbuf.data_ptr(), _lib.ptr(b), dst.data_ptr(), dst_c,
                                        int(accumulate), _lib.ptr(stats), dims, _s()), "conv3d_stream")
    return dst, stats, slots


def conv3d_march_supported(srcs: Sequence[torch.Tensor], dst_channels: Sequence[int], dilation: int = 1) -> bool:
    return bool(_lib.load().seunet_conv3d_march_supported(_code(srcs[0]), dilation, len(srcs), _lib.int_array([t.shape[4] for t in srcs]),
                                                          len(dst_channels), _lib.int_array(list(dst_channels))))


def conv3d_march(srcs: Sequence[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor] = None, dilation: int = 1,
                 transpose_flip: bool = False, dsts: Optional[Sequence[Optional[torch.Tensor]]] = None,
                 dst_channels: Optional[Sequence[int]] = None, accumulate: Optional[Sequence[int]] = None, want_stats: bool = False):
    """The marching 3x3x3 convolution (csrc/conv_march.hip): 32 or 64 source channels (one or two tensors), bf16 / fp16.
    Returns (list of destination tensors, stats_partial or None, slots)."""
    lib = _lib.load()
    code, dims = _code(srcs[0]), _dims_cl(srcs[0])
    co_w, ci_w = weight.shape[0], weight.shape[1]
    cout_e = ci_w if transpose_flip else co_w
    cin = sum(t.shape[4] for t in srcs)
    if dsts is None:
        dst_channels = list(dst_channels or [cout_e])
        dsts = [torch.empty(tuple(srcs[0].shape[:4]) + (c,), dtype=srcs[0].dtype, device=srcs[0].device) for c in dst_channels]
        accumulate = [0] * len(dsts)
    else:
        dst_channels = list(dst_channels or [t.shape[4] for t in dsts])
        accumulate = list(accumulate or [0] * len(dsts))
    if not conv3d_march_supported(srcs, dst_channels, dilation):
        raise RuntimeError(f"conv3d_march: {[t.shape[4] for t in srcs]} -> {dst_channels} channels, dilation {dilation}, "
                           f"dtype {srcs[0].dtype} is not served by this kernel")
    cout = sum(dst_channels)
    wbuf = torch.empty(lib.seunet_conv3d_march_wpack_bytes(cin, cout), dtype=torch.uint8, device=srcs[0].device)
    w = weight.contiguous().float()
    _lib.check(lib.seunet_conv3d_march_pack(code, w.data_ptr(), ci_w, co_w, int(transpose_flip), cin, cout, wbuf.data_ptr(), _s()),
               "conv3d_march_pack")
    stats, slots = None, 0
    if want_stats:
        slots = lib.seunet_conv3d_march_slots(dilation, cin, cout, dims)
        stats = torch.zeros((dims.n, slots, cout, 2), dtype=torch.float64, device=srcs[0].device)
    b = None
    if bias is not None:
        b = torch.zeros(cout, dtype=torch.float32, device=srcs[0].device)
        b[:bias.numel()] = bias.float()
    _lib.check(lib.seunet_conv3d_march(code, dilation, len(srcs), _lib.ptr_array(list(srcs)), _lib.int_array([t.shape[4] for t in srcs]),
                                       wbuf.data_ptr(), _lib.ptr(b), len(dsts), _lib.ptr_array(list(dsts)),
                                       _lib.int_array(dst_channels), _lib.int_array(accumulate), _lib.ptr(stats), dims, _s()),
               "conv3d_march")
    return list(dsts), stats, slots


def conv3d_wgrad(srcs: Sequence[torch.Tensor], dy: torch.Tensor, cin: int, cout: int, taps: int, dilation: int = 1,
                 impl: int = _lib.CONV_MFMA) -> torch.Tensor:
    lib = _lib.load()
    code = _code(dy)
    k = 3 if taps == 27 else 1
    dw = torch.empty((cout, cin, k, k, k), dtype=torch.float32, device=dy.device)
    nbytes = lib.seunet_conv3d_wgrad_workspace_bytes(taps, cin, cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    _lib.check(lib.seunet_conv3d_wgrad(code, impl, taps, dilation, len(srcs), _lib.ptr_array(list(srcs)),
                                       _lib.int_array([t.shape[4] for t in srcs]), cin, dy.data_ptr(), dy.shape[4],
                                       dw.data_ptr(), ws.data_ptr(), nbytes, _dims_cl(dy), _s()), "conv3d_wgrad")
    return dw


def conv3d_wgrad_stream(x: torch.Tensor, dy: torch.Tensor, cin: int, cout: int, dilation: int = 1) -> torch.Tensor:
    """Weight gradient on the streaming kernel (csrc/wgrad_stream.hip): x (N,D,H,W,8|16|32) bf16, dy (N,D,H,W,C) bf16."""
    lib = _lib.load()
    code, dims = _code(dy), _dims_cl(dy)
    if not lib.seunet_conv3d_wgrad_stream_supported(code, dilation, x.shape[4], dy.shape[4]):
        raise RuntimeError(f"conv3d_wgrad_stream: {x.shape[4]} x {dy.shape[4]} channels, dilation {dilation} is not served by this kernel")
    dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device=dy.device)
    nbytes = lib.seunet_conv3d_wgrad_stream_workspace_bytes(x.shape[4], dy.shape[4], dilation, dims)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    _lib.check(lib.seunet_conv3d_wgrad_stream(code, dilation, x.data_ptr(), x.shape[4], cin, dy.data_ptr(), dy.shape[4], cout,
                                              dw.data_ptr(), ws.data_ptr(), nbytes, dims, _s()), "conv3d_wgrad_stream")
    return dw


# ---- statistics -----------------------------------------------------------------------------------
def channel_stats(t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    lib = _lib.load()
    dims = _dims_cl(t)
    slots = lib.seunet_epilogue_slots(dims)
    part = torch.zeros((dims.n, slots, t.shape[4], 2), dtype=torch.float64, device=t.device)
    _lib.check(lib.seunet_channel_stats(_code(t), t.data_ptr(), t.shape[4], part.data_ptr(), dims, _s()), "channel_stats")
    return part, slots


def stats_finalize(partial: torch.Tensor, slots: int, count: int, eps: float = 1e-5, mode: int = 0):
    n, _, c, _ = partial.shape
    a = torch.empty((n, c), dtype=torch.float32, device=partial.device)
    b = torch.empty_like(a)
    _lib.check(_lib.load().seunet_stats_finalize(partial.data_ptr(), slots, c, n, count, eps, mode, a.data_ptr(),
                                                 b.data_ptr(), _s()), "stats_finalize")
    return a, b


# ---- epilogues --------------------------------------------------------------------------------------
def gate_epilogue_fwd(raw, mean, rstd, w_se, w_se2, w_side, b_side, slope=0.01, level_map=None, level_accumulate=0,
                      head_w=None, drop=None, drop_stride=0, want_side=True):
    n, d, h, w, c = raw.shape
    e = torch.empty_like(raw)
    side = torch.empty((n, d, h, w, 2), dtype=torch.float32, device=raw.device) if want_side else None
    f = lambda t: None if t is None else t.contiguous().float().reshape(-1)
    ws, ws2, wsd, bsd, hw = f(w_se), f(w_se2), f(w_side), f(b_side), f(head_w)
    _lib.check(_lib.load().seunet_gate_epilogue_fwd(_code(raw), raw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), c,
                                                    _lib.ptr(ws), _lib.ptr(ws2), _lib.ptr(wsd), _lib.ptr(bsd), slope,
                                                    e.data_ptr(), _lib.ptr(side), _lib.ptr(level_map), level_accumulate,
                                                    _lib.ptr(hw), _lib.ptr(drop), drop_stride, _dims_cl(raw), _s()),
               "gate_epilogue_fwd")
    return e, side


def gate_epilogue_bwd(raw, mean, rstd, w_se, w_se2, w_side, b_side, slope=0.01, g_e=None, g_side=None, g_level=None,
                      head_w=None, drop=None, drop_stride=0):
    """Both backward passes.  Returns dict with draw (gradient w.r.t. the raw conv output) and the parameter gradients."""
    lib = _lib.load()
    n, d, h, w, c = raw.shape
    dims = _dims_cl(raw)
    slots = lib.seunet_epilogue_slots(dims)
    dx = torch.empty_like(raw)
    stat = torch.zeros((n, slots, c, 2), dtype=torch.float64, device=raw.device)
    pg = torch.zeros((n * slots, 4 * c + 4), dtype=torch.float32, device=raw.device)
    f = lambda t: None if t is None else t.contiguous().float().reshape(-1)
    ws, ws2, wsd, bsd, hw = f(w_se), f(w_se2), f(w_side), f(b_side), f(head_w)

    def call(m1, m2, out, st, pgp):
        _lib.check(lib.seunet_gate_epilogue_bwd(_code(raw), raw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), c, _lib.ptr(ws),
                                                _lib.ptr(ws2), _lib.ptr(wsd), _lib.ptr(bsd), slope, _lib.ptr(g_e),
                                                _lib.ptr(g_side), _lib.ptr(g_level), _lib.ptr(hw), _lib.ptr(drop), drop_stride,
                                                _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(out), _lib.ptr(st), _lib.ptr(pgp), dims, _s()),
                   "gate_epilogue_bwd")
    call(None, None, None, stat, pg)
    m1, m2 = stats_finalize(stat, slots, d * h * w, 0.0, 1)
    dev = raw.device
    out = {"dw_se": torch.empty(c, device=dev), "dw_se2": torch.empty(c, device=dev), "dw_side": torch.empty(2 * c, device=dev),
           "db_side": torch.empty(2, device=dev), "dhead_w": torch.empty(2, device=dev)}
    _lib.check(lib.seunet_pgrad_reduce(pg.data_ptr(), n * slots, c, out["dw_se"].data_ptr(), out["dw_se2"].data_ptr(),
                                       out["dw_side"].data_ptr(), out["db_side"].data_ptr(), out["dhead_w"].data_ptr(), _s()),
               "pgrad_reduce")
    call(m1, m2, dx, None, None)
    out["draw"] = dx
    return out


def cat_epilogue_fwd(raw, mean, rstd, raw2=None, mean2=None, rstd2=None, slope=0.01):
    out = torch.empty_like(raw)
    _lib.check(_lib.load().seunet_cat_epilogue_fwd(_code(raw), raw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _lib.ptr(raw2),
                                                   _lib.ptr(mean2), _lib.ptr(rstd2), raw.shape[4], slope, out.data_ptr(),
                                                   _dims_cl(raw), _s()), "cat_epilogue_fwd")
    return out


def cat_epilogue_bwd(g_out, raw, mean, rstd, raw2=None, mean2=None, rstd2=None, slope=0.01):
    lib = _lib.load()
    n, d, h, w, c = raw.shape
    dims = _dims_cl(raw)
    slots = lib.seunet_epilogue_slots(dims)
    dx = torch.empty_like(raw)
    dx2 = torch.empty_like(raw) if raw2 is not None else None
    st = torch.zeros((n, slots, c, 2), dtype=torch.float64, device=raw.device)
    st2 = torch.zeros_like(st) if raw2 is not None else None

    def call(m1, m2, m1b, m2b, o1, o2, s1, s2):
        _lib.check(lib.seunet_cat_epilogue_bwd(_code(raw), g_out.data_ptr(), raw.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               _lib.ptr(raw2), _lib.ptr(mean2), _lib.ptr(rstd2), c, slope, _lib.ptr(m1), _lib.ptr(m2),
                                               _lib.ptr(m1b), _lib.ptr(m2b), _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(s1), _lib.ptr(s2),
                                               dims, _s()), "cat_epilogue_bwd")
    call(None, None, None, None, None, None, st, st2)
    m1, m2 = stats_finalize(st, slots, d * h * w, 0.0, 1)
    m1b, m2b = stats_finalize(st2, slots, d * h * w, 0.0, 1) if raw2 is not None else (None, None)
    call(m1, m2, m1b, m2b, dx, dx2, None, None)
    return dx, dx2


def cat_epilogue_x(g_out, raw, mean, rstd, x_in, w2, in_channel, slope=0.01, eps=1e-5):
    """Two-branch aggregation block whose second branch (the 1x1x1 conv ``w2`` of the <= 2-channel network input) is
    recomputed instead of stored (``seunet_xbranch_*``, ``seunet_cat_epilogue_fwd_x / bwd_x``).
    x_in: packed 8-channel input [N, D, H, W, 8]; w2: (C, in_channel, 1, 1, 1) f32.
    Returns (out, dx, dW2): forward output, gradient w.r.t. ``raw`` for the upstream ``g_out``, gradient of ``w2``."""
    lib = _lib.load()
    n, d, h, w, c = raw.shape
    dims = _dims_cl(raw)
    code = _code(raw)
    w2f = w2.contiguous().float().reshape(c, in_channel)
    mslots = lib.seunet_xbranch_moment_slots(dims)
    mom = torch.zeros((n, mslots, 5), dtype=torch.float64, device=raw.device)
    _lib.check(lib.seunet_xbranch_moments(code, x_in.data_ptr(), mom.data_ptr(), dims, _s()), "xbranch_moments")
    mean2 = torch.empty((n, c), dtype=torch.float32, device=raw.device)
    rstd2 = torch.empty_like(mean2)
    tot = torch.empty((n, 5), dtype=torch.float64, device=raw.device)
    _lib.check(lib.seunet_xbranch_stats(mom.data_ptr(), mslots, w2f.data_ptr(), c, in_channel, n, d * h * w, eps, mean2.data_ptr(),
                                        rstd2.data_ptr(), tot.data_ptr(), _s()), "xbranch_stats")
    out = torch.empty_like(raw)
    _lib.check(lib.seunet_cat_epilogue_fwd_x(code, raw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), x_in.data_ptr(), w2f.data_ptr(),
                                             in_channel, mean2.data_ptr(), rstd2.data_ptr(), c, slope, out.data_ptr(), dims, _s()),
               "cat_epilogue_fwd_x")
    slots = lib.seunet_epilogue_slots(dims)
    st = torch.zeros((n, slots, c, 2), dtype=torch.float64, device=raw.device)
    st2 = torch.zeros_like(st)
    part = torch.empty((n, slots, c, 2), dtype=torch.float64, device=raw.device)
    dx = torch.empty_like(raw)

    def bwd(m1, m2, m1b, m2b, o, s1, s2, xp):
        _lib.check(lib.seunet_cat_epilogue_bwd_x(code, g_out.data_ptr(), raw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), x_in.data_ptr(),
                                                 w2f.data_ptr(), in_channel, mean2.data_ptr(), rstd2.data_ptr(), c, slope, _lib.ptr(m1),
                                                 _lib.ptr(m2), _lib.ptr(m1b), _lib.ptr(m2b), _lib.ptr(o), _lib.ptr(s1), _lib.ptr(s2),
                                                 _lib.ptr(xp), dims, _s()), "cat_epilogue_bwd_x")
    bwd(None, None, None, None, None, st, st2, part)
    m1, m2 = stats_finalize(st, slots, d * h * w, 0.0, 1)
    m1b, m2b = stats_finalize(st2, slots, d * h * w, 0.0, 1)
    bwd(m1, m2, m1b, m2b, dx, None, None, None)
    dw = torch.zeros((c, in_channel, 1, 1, 1), dtype=torch.float32, device=raw.device)
    _lib.check(lib.seunet_cat_xgrad_finalize(part.data_ptr(), st2.data_ptr(), slots, tot.data_ptr(), w2f.data_ptr(), c, in_channel, n,
                                             eps, dw.data_ptr(), _s()), "cat_xgrad_finalize")
    return out, dx, dw


# ---- pooling / interpolation / heads ---------------------------------------------------------------------
def maxpool_fwd(t):
    n, d, h, w, c = t.shape
    out = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=t.dtype, device=t.device)
    _lib.check(_lib.load().seunet_maxpool_fwd(_code(t), t.data_ptr(), c, out.data_ptr(), _dims_cl(t), _s()), "maxpool_fwd")
    return out


def maxpool_bwd(t, g_out, g_in=None):
    acc = 0 if g_in is None else 1
    g_in = torch.empty_like(t) if g_in is None else g_in
    _lib.check(_lib.load().seunet_maxpool_bwd(_code(t), t.data_ptr(), g_out.data_ptr(), t.shape[4], g_in.data_ptr(), acc,
                                              _dims_cl(t), _s()), "maxpool_bwd")
    return g_in


def upsample2_fwd(t):
    n, d, h, w, c = t.shape
    out = torch.empty((n, 2 * d, 2 * h, 2 * w, c), dtype=t.dtype, device=t.device)
    _lib.check(_lib.load().seunet_upsample2_fwd(_code(t), t.data_ptr(), c, out.data_ptr(), _dims_cl(t), _s()), "upsample2_fwd")
    return out


def upsample2_bwd(g_out, g_in=None):
    n, d2, h2, w2, c = g_out.shape
    acc = 0 if g_in is None else 1
    if g_in is None:
        g_in = torch.empty((n, d2 // 2, h2 // 2, w2 // 2, c), dtype=g_out.dtype, device=g_out.device)
    _lib.check(_lib.load().seunet_upsample2_bwd(_code(g_out), g_out.data_ptr(), c, g_in.data_ptr(), acc,
                                                Dims(n, d2 // 2, h2 // 2, w2 // 2), _s()), "upsample2_bwd")
    return g_in


def side_upsample(side: torch.Tensor, scale: int) -> torch.Tensor:
    """side: (N, d, h, w, C) float32 -> (N, C, d*scale, h*scale, w*scale) float32, trilinear align_corners=True."""
    n, d, h, w, c = side.shape
    out = torch.empty((n, c, d * scale, h * scale, w * scale), dtype=torch.float32, device=side.device)
    _lib.check(_lib.load().seunet_side_upsample(side.data_ptr(), c, scale, out.data_ptr(), c, 0, Dims(n, d, h, w), _s()),
               "side_upsample")
    return out


def head_fwd(level_maps: Sequence[torch.Tensor], bias: torch.Tensor) -> torch.Tensor:
    n, d, h, w = level_maps[0].shape
    pred = torch.empty((n, 1, d, h, w), dtype=torch.float32, device=bias.device)
    b = bias.contiguous().float()
    _lib.check(_lib.load().seunet_head_fwd(_lib.ptr_array(list(level_maps)), len(level_maps), b.data_ptr(), pred.data_ptr(),
                                           Dims(n, d, h, w), _s()), "head_fwd")
    return pred


def head_bwd(g_pred: torch.Tensor, nlevels: int):
    lib = _lib.load()
    n, _, d, h, w = g_pred.shape
    dims = Dims(n, d, h, w)
    g_pred = g_pred.contiguous().float()
    levels: List[Optional[torch.Tensor]] = [None]
    for l in range(1, nlevels):
        levels.append(torch.empty((n, d >> l, h >> l, w >> l), dtype=torch.float32, device=g_pred.device))
    tmp = torch.empty(lib.seunet_head_bwd_tmp_floats(dims), dtype=torch.float32, device=g_pred.device)
    gb = torch.empty(1, dtype=torch.float32, device=g_pred.device)
    _lib.check(lib.seunet_head_bwd(g_pred.data_ptr(), _lib.ptr_array(levels), nlevels, tmp.data_ptr(), gb.data_ptr(), dims, _s()),
               "head_bwd")
    return levels, gb


# ---- stand-alone block modules (reference SSEConv / SSEConv2 / CATConv forward) -------------------------
def _block_dtype():
    from .SE_UNet import _default_dtype
    return _lib.dtype_code(_default_dtype())


def _conv_stats(srcs, weight, bias, dilation, impl, cin=None):
    (raw,), part, slots = conv3d(srcs, weight, bias, dilation, impl, cin=cin, want_stats=True)
    n, d, h, w, _ = raw.shape
    mean, rstd = stats_finalize(part, slots, d * h * w)
    return raw, mean, rstd


def _side_grad_to_block(g_s: torch.Tensor, down_sample: int) -> torch.Tensor:
    """Transpose of ``side_upsample``: (N, 2, D*s, H*s, W*s) -> (N, D, H, W, 2) float32 (the transposed align_corners
    interpolation is the one the heads use: ``seunet_head_bwd``, one call per side channel)."""
    if down_sample == 1:
        return g_s.permute(0, 2, 3, 4, 1).contiguous().float()
    lvl = {2: 1, 4: 2, 8: 3}[down_sample]
    parts = [head_bwd(g_s[:, c:c + 1].contiguous(), lvl + 1)[0][lvl] for c in range(g_s.shape[1])]
    return torch.stack(parts, dim=-1).contiguous()


class _GatedBlockFn(torch.autograd.Function):
    """SSEConv / SSEConv2 (SE_UNet.py:9-35, 51-82) as a differentiable unit: the network's kernels called block-wise
    (conv + statistics, fused epilogue; backward = the two InstanceNorm passes, weight gradient, data gradient)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w_se, w_se2, w2, b2, dilation, down_sample, slope):
        from .SE_UNet import _default_conv_impl
        impl = _default_conv_impl()
        xc = to_cl(x, _block_dtype())
        raw, mean, rstd = _conv_stats([xc], w1, b1, dilation, impl, cin=x.shape[1])
        e, side = gate_epilogue_fwd(raw, mean, rstd, w_se, w_se2, w2, b2, slope)
        ctx.save_for_backward(xc, raw, mean, rstd, w1, w_se, w_se2 if w_se2 is not None else w_se, w2, b2)
        ctx.meta = (dilation, down_sample, slope, impl, x.shape[1], w_se2 is not None)
        return from_cl(e), side_upsample(side, down_sample)

    @staticmethod
    def backward(ctx, g_e, g_s):
        xc, raw, mean, rstd, w1, w_se, w_se2, w2, b2 = ctx.saved_tensors
        dilation, down_sample, slope, impl, cin, two = ctx.meta
        code = _code(raw)
        g_ec = None if g_e is None else to_cl(g_e.contiguous().float(), code)
        g_side = None if g_s is None else _side_grad_to_block(g_s.contiguous().float(), down_sample)
        out = gate_epilogue_bwd(raw, mean, rstd, w_se, w_se2 if two else None, w2, b2, slope, g_e=g_ec, g_side=g_side)
        cout = w1.shape[0]
        dw1 = conv3d_wgrad([xc], out["draw"], cin, cout, 27, dilation, impl)
        gx = None
        if ctx.needs_input_grad[0]:
            (g,), _, _ = conv3d([out["draw"]], w1, None, dilation, impl, transpose_flip=True)
            gx = from_cl(g, cin)
        # conv1.bias feeds an affine-less InstanceNorm: its gradient is identically zero
        return (gx, dw1, torch.zeros(cout, device=raw.device), out["dw_se"].view_as(w_se),
                out["dw_se2"].view_as(w_se) if two else None, out["dw_side"].view_as(w2), out["db_side"].view_as(b2),
                None, None, None)


class _CatBlockFn(torch.autograd.Function):
    """CATConv (SE_UNet.py:37-49) as a differentiable unit."""

    @staticmethod
    def forward(ctx, x, w1, slope):
        from .SE_UNet import _default_conv_impl
        impl = _default_conv_impl()
        xc = to_cl(x, _block_dtype())
        raw, mean, rstd = _conv_stats([xc], w1, None, 1, impl, cin=x.shape[1])
        ctx.save_for_backward(xc, raw, mean, rstd, w1)
        ctx.meta = (slope, impl, x.shape[1])
        return from_cl(cat_epilogue_fwd(raw, mean, rstd, slope=slope))

    @staticmethod
    def backward(ctx, g):
        xc, raw, mean, rstd, w1 = ctx.saved_tensors
        slope, impl, cin = ctx.meta
        dx, _ = cat_epilogue_bwd(to_cl(g.contiguous().float(), _code(raw)), raw, mean, rstd, slope=slope)
        dw1 = conv3d_wgrad([xc], dx, cin, w1.shape[0], 1, 1, impl)
        gx = None
        if ctx.needs_input_grad[0]:
            (gg,), _, _ = conv3d([dx], w1, None, 1, impl, transpose_flip=True)
            gx = from_cl(gg, cin)
        return gx, dw1, None


def gated_block_forward(mod, x: torch.Tensor):
    """(e0, e1) of SE_UNet.py:24-35 / 68-82 for an NCDHW input on the GPU; differentiable w.r.t. the input and the block's
    parameters (the whole-network path does not go through here: it is ONE native call per direction)."""
    if not x.is_cuda:
        raise RuntimeError("HIP path needs a GPU tensor (no CPU fallback)")
    w_se2 = mod.conv_se2.weight if getattr(mod, "conv_se2", None) is not None else None
    return _GatedBlockFn.apply(x.contiguous().float(), mod.conv1.weight, mod.conv1.bias, mod.conv_se.weight, w_se2,
                               mod.conv2.weight, mod.conv2.bias, mod.dilation, mod.down_sample, float(mod.act.negative_slope))


def cat_block_forward(mod, x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError("HIP path needs a GPU tensor (no CPU fallback)")
    return _CatBlockFn.apply(x.contiguous().float(), mod.conv1.weight, float(mod.act.negative_slope))
