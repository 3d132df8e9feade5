"""ctypes binding of ``libseunet_hip.so`` (C ABI: ``include/seunet_hip.h``).

The product path has no fallback: if the shared library is missing or an entry point
returns non-zero, a ``RuntimeError`` is raised (message from ``seunet_last_error``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEUNET_LIB") or os.path.join(_HERE, "libseunet_hip.so")   # SEUNET_LIB: diagnostic builds only

F32, BF16, F16 = 0, 1, 2
CONV_MFMA, CONV_NAIVE, CONV_MARCH, CONV_TILED = 0, 1, 2, 3
LOSS_NSUMS = 7
DTI_F64, DTI_F32 = 0, 1
CC_EVALUATION, CC_MAXIMUM_3D = 0, 1


class Dims(C.Structure):
    _fields_ = [("n", C.c_int), ("d", C.c_int), ("h", C.c_int), ("w", C.c_int)]


class NetDesc(C.Structure):
    _fields_ = [("batch", C.c_int), ("in_channel", C.c_int), ("n_classes", C.c_int),
                ("d", C.c_int), ("h", C.c_int), ("w", C.c_int), ("width_mult", C.c_int),
                ("dtype", C.c_int), ("conv_impl", C.c_int),
                ("negative_slope", C.c_float), ("eps", C.c_float)]


_vp, _i, _f, _ll, _sz, _d = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t, C.c_double
_pp = C.POINTER(C.c_void_p)
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes).  Must list every symbol declared in include/seunet_hip.h
# (tests/test_abi.py checks the header against this table and against the .so).
PROTOTYPES = {
    "seunet_version": (_i, []),
    "seunet_last_error": (C.c_char_p, []),
    "seunet_pack_cl": (_i, [_i, _vp, _i, _vp, _i, Dims, _vp]),
    "seunet_unpack_cl": (_i, [_i, _vp, _i, _vp, Dims, _vp]),
    "seunet_conv_wpack_bytes": (_sz, [_i, _i, _i, _i]),
    "seunet_conv_pack_weights": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "seunet_conv_stats_slots": (_i, [_i, _i, _i, Dims]),
    "seunet_conv3d_fwd": (_i, [_i, _i, _i, _i, _i, _pp, _ip, _i, _vp, _i, _vp, _i, _pp, _ip, _ip, _vp, Dims, _vp]),
    "seunet_conv3d_stream_supported": (_i, [_i, _i, _i, _i]),
    "seunet_conv3d_stream_wpack_bytes": (_sz, [_i]),
    "seunet_conv3d_stream_slots": (_i, [_i, Dims]),
    "seunet_conv3d_stream_pack": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "seunet_conv3d_stream": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, Dims, _vp]),
    "seunet_conv3d_march_supported": (_i, [_i, _i, _i, _ip, _i, _ip]),
    "seunet_conv3d_march_wpack_bytes": (_sz, [_i, _i]),
    "seunet_conv3d_march_slots": (_i, [_i, _i, _i, Dims]),
    "seunet_conv3d_march_pack": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "seunet_conv3d_march": (_i, [_i, _i, _i, _pp, _ip, _vp, _vp, _i, _pp, _ip, _ip, _vp, Dims, _vp]),
    "seunet_conv3d_wgrad_stream_supported": (_i, [_i, _i, _i, _i]),
    "seunet_conv3d_wgrad_stream_workspace_bytes": (_sz, [_i, _i, _i, Dims]),
    "seunet_conv3d_wgrad_stream": (_i, [_i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, Dims, _vp]),
    "seunet_conv3d_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "seunet_conv3d_wgrad": (_i, [_i, _i, _i, _i, _i, _pp, _ip, _i, _vp, _i, _vp, _vp, _sz, Dims, _vp]),
    "seunet_epilogue_slots": (_i, [Dims]),
    "seunet_channel_stats": (_i, [_i, _vp, _i, _vp, Dims, _vp]),
    "seunet_stats_finalize": (_i, [_vp, _i, _i, _i, _ll, _f, _i, _vp, _vp, _vp]),
    "seunet_gate_epilogue_fwd": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp, _vp, _i, Dims, _vp]),
    "seunet_gate_epilogue_bwd": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, Dims, _vp]),
    "seunet_pgrad_reduce": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "seunet_cat_epilogue_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, Dims, _vp]),
    "seunet_cat_epilogue_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, Dims, _vp]),
    "seunet_maxpool_fwd": (_i, [_i, _vp, _i, _vp, Dims, _vp]),
    "seunet_maxpool_bwd": (_i, [_i, _vp, _vp, _i, _vp, _i, Dims, _vp]),
    "seunet_upsample2_fwd": (_i, [_i, _vp, _i, _vp, Dims, _vp]),
    "seunet_upsample2_bwd": (_i, [_i, _vp, _i, _vp, _i, Dims, _vp]),
    "seunet_side_upsample": (_i, [_vp, _i, _i, _vp, _i, _i, Dims, _vp]),
    "seunet_head_fwd": (_i, [_pp, _i, _vp, _vp, Dims, _vp]),
    "seunet_head_bwd_tmp_floats": (_sz, [Dims]),
    "seunet_head_bwd": (_i, [_vp, _pp, _i, _vp, _vp, Dims, _vp]),
    "seunet_loss_partial_floats": (_i, []),
    "seunet_loss_sums": (_i, [_vp, _i, _vp, _vp, _vp, _ll, _vp, _vp, _i, _vp]),
    "seunet_loss_value": (_i, [_vp, _d, _d, _d, _vp, _d, _d, _d, _vp, _vp]),
    "seunet_loss_grad": (_i, [_vp, _i, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _vp, _vp, _vp]),
    "seunet_xbranch_moment_slots": (_i, [Dims]),
    "seunet_xbranch_moments": (_i, [_i, _vp, _vp, Dims, _vp]),
    "seunet_xbranch_stats": (_i, [_vp, _i, _vp, _i, _i, _i, _ll, _f, _vp, _vp, _vp, _vp]),
    "seunet_cat_epilogue_fwd_x": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, Dims, _vp]),
    "seunet_cat_epilogue_bwd_x": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, Dims, _vp]),
    "seunet_cat_xgrad_finalize": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "seunet_cc_workspace_bytes": (_sz, [_i, _i, _i]),
    "seunet_largest_component": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "seunet_metric_out_bytes": (_sz, [_i]),
    "seunet_metric_sums": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _vp, _sz, _vp]),
    "seunet_crop_batch": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _ip, _ip, C.c_double, _i, _vp, _vp, _vp, _vp, _vp]),
    "seunet_hu_two_channel": (_i, [_vp, _i, _ll, _i, _vp, _vp]),
    "seunet_window_gather": (_i, [_vp, _i, _i, _i, _i, _i, _i, _ip, _vp, _vp]),
    "seunet_window_accumulate": (_i, [_vp, _i, _i, _ip, _i, _vp, _i, _i, _i, _vp]),
    "seunet_window_finalize": (_i, [_vp, _i, _i, _i, _i, _i, _ip, _i, _ip, _i, _ip, _i, _vp, _vp]),
    "seunet_dti_workspace_bytes": (_sz, [_i, _i, _i]),
    "seunet_dti": (_i, [_vp, _i, _i, _i, C.c_double, C.c_double, _i, _vp, _vp, _sz, _vp]),
    "seunet_adamw_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i, _i, _vp]),
    "seunet_net_param_count": (_i, [C.POINTER(NetDesc)]),
    "seunet_net_param_info": (_i, [C.POINTER(NetDesc), _i, C.c_char_p, _i, _ip, _ip]),
    "seunet_net_workspace_bytes": (_sz, [C.POINTER(NetDesc)]),
    "seunet_net_forward": (_i, [C.POINTER(NetDesc), _pp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "seunet_prof_enable": (_i, [_i]),
    "seunet_prof_enable_filtered": (_i, [C.c_char_p]),
    "seunet_prof_report": (_i, [C.c_char_p, _sz]),
    "seunet_init": (_i, [_i]),
    "seunet_net_forward_capture": (_i, [C.POINTER(NetDesc), _pp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, C.POINTER(C.c_void_p)]),
    "seunet_graph_launch": (_i, [_vp, _vp]),
    "seunet_graph_destroy": (_i, [_vp]),
    "seunet_net_read_tensor": (_i, [C.POINTER(NetDesc), _pp, _vp, _sz, C.c_char_p, _i, _vp, _ip, _vp]),
    "seunet_net_backward": (_i, [C.POINTER(NetDesc), _pp, _vp, _vp, _vp, _vp, _pp, _vp, _sz, _vp]),
    "seunet_net_backward_ev": (_i, [C.POINTER(NetDesc), _pp, _vp, _vp, _vp, _vp, _pp, _vp, _sz, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C se-unet-airseg_amd/csrc`).  There is no CPU or PyTorch fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def last_error() -> str:
    return load().seunet_last_error().decode(errors="replace")


def check(status: int, what: str = "") -> None:
    if status != 0:
        raise RuntimeError(f"libseunet_hip {what}: {last_error()}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors: Sequence) -> C.Array:
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else (t if isinstance(t, int) else t.data_ptr())
    return arr


def int_array(vals: Sequence[int]) -> C.Array:
    return (C.c_int * len(vals))(*vals)


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def dtype_code(name: str) -> int:
    name = str(name).lower().replace("torch.", "")
    if name in ("bf16", "bfloat16"):
        return BF16
    if name in ("fp32", "f32", "float32", "float"):
        return F32
    if name in ("fp16", "f16", "float16", "half"):
        return F16
    raise ValueError(f"unsupported activation dtype {name!r} (fp32, bf16 or fp16)")
