from .pipeline_PowerPaint import StableDiffusionInpaintPipeline
from .pipeline_PowerPaint_Brushnet_CA import StableDiffusionPowerPaintBrushNetPipeline
from .pipeline_PowerPaint_ControlNet import StableDiffusionControlNetInpaintPipeline

__all__ = [
    "StableDiffusionInpaintPipeline",
    "StableDiffusionControlNetInpaintPipeline",
    "StableDiffusionPowerPaintBrushNetPipeline",
]
