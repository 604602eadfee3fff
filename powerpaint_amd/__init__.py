"""powerpaint_amd -- MI355X-native (gfx950) implementation of PowerPaint's denoising hot path.

Mirrors the reference package layout for the path it replaces:
    powerpaint.models.{UNet2DConditionModel, BrushNetModel}      -> powerpaint_amd.models
    powerpaint.pipelines.{StableDiffusionInpaintPipeline, StableDiffusionPowerPaintBrushNetPipeline,
                          StableDiffusionControlNetInpaintPipeline} -> powerpaint_amd.pipelines
All device work goes through the C ABI of libpp_hip.so (include/pp_hip.h); there is no PyTorch/CPU fallback.
"""
__version__ = "0.1.0"
