"""se-unet-airseg_amd: MI355X-native (gfx950) SE-UNet training / inference hot path.

Drop-in for the reference's ``SE_UNet.py`` module surface and ``train.py`` loss functions, backed by
hand-written HIP kernels behind a C ABI (``include/seunet_hip.h``).  The directory name is not a valid
Python identifier; import it as ``seunet_amd`` (alias module at the repo root) or via
``importlib.import_module("se-unet-airseg_amd")``.
"""
from . import _lib, ddp, ops, optim, pipeline, postprocess
from .SE_UNet import CapturedForward, CATConv, DropLayer, SE_UNet, SSEConv, SSEConv2, get_model, load_reference_checkpoint
from .optim import AdamW
from .pipeline import (AirwayHMData3GPU, AirwayHMDataGPU, CropSegDataGPU, aug_code, crop_batch, draw_stage1_plan, draw_stage2_plan,
                       draw_stage3_plan, two_channel_volume)
from .postprocess import (MetricSums, double_threshold_iteration, evaluation_case, largest_component, maximum_3d,
                          postprocess_prediction, zero_borders)
from .losses import atr_loss, dice_loss, fused_logit_loss, fused_stage_loss, general_union_loss_lib
from .sliding_window import sliding_window_predict, sliding_window_validate, two_channel, window_starts, window_table

__all__ = ["SE_UNet", "SSEConv", "SSEConv2", "CATConv", "DropLayer", "get_model", "load_reference_checkpoint", "dice_loss",
           "general_union_loss_lib", "atr_loss", "fused_logit_loss", "fused_stage_loss",
           "sliding_window_predict", "sliding_window_validate", "two_channel", "window_starts", "window_table", "AdamW", "CropSegDataGPU", "AirwayHMDataGPU", "AirwayHMData3GPU", "aug_code", "crop_batch", "draw_stage1_plan", "draw_stage2_plan", "draw_stage3_plan", "two_channel_volume", "double_threshold_iteration", "postprocess_prediction", "zero_borders", "maximum_3d", "largest_component",
           "evaluation_case", "MetricSums"]
