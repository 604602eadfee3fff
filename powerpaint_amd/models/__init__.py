from .autoencoder_kl import AutoencoderKL
from .BrushNet_CA import BrushNetModel
from .clip_text import CLIPTextModel
from .controlnet import ControlNetModel
from .unet_2d_condition import UNet2DConditionModel

__all__ = ["BrushNetModel", "UNet2DConditionModel", "ControlNetModel", "AutoencoderKL", "CLIPTextModel"]
