"""VaeImageProcessor: the PIL / NumPy <-> tensor conversions at the two ends of the pipelines.

Host-side mirror of the diffusers class the reference pipelines instantiate
(/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:284 `VaeImageProcessor(vae_scale_factor=...)`, :1058
`image_processor.postprocess(image, output_type=..., do_denormalize=...)`; pipeline_PowerPaint_Brushnet_CA.py:1305-1311
`image_processor.preprocess(image / mask, height, width)`; pipeline_PowerPaint_ControlNet.py:281-283
`VaeImageProcessor(do_convert_rgb=True, do_normalize=False)` for the control image).  diffusers itself is not
installable here: these are the documented conversions (uint8 / 255, 2x - 1, Lanczos resize to multiples of the VAE
factor, x / 2 + 0.5 clamped, round to uint8), restated; plain data-format glue, no tensor math of the hot path.
"""
from typing import List, Optional

import numpy as np
import torch


class VaeImageProcessor:
    def __init__(self, do_resize: bool = True, vae_scale_factor: int = 8, resample: str = "lanczos",
                 do_normalize: bool = True, do_binarize: bool = False, do_convert_rgb: bool = False,
                 do_convert_grayscale: bool = False):
        from types import SimpleNamespace
        self.config = SimpleNamespace(do_resize=do_resize, vae_scale_factor=vae_scale_factor, resample=resample,
                                      do_normalize=do_normalize, do_binarize=do_binarize,
                                      do_convert_rgb=do_convert_rgb, do_convert_grayscale=do_convert_grayscale)

    # ---- elementary conversions
    @staticmethod
    def numpy_to_pil(images: np.ndarray) -> list:
        import PIL.Image
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        if images.shape[-1] == 1:
            return [PIL.Image.fromarray(i.squeeze(), mode="L") for i in images]
        return [PIL.Image.fromarray(i) for i in images]

    @staticmethod
    def pil_to_numpy(images) -> np.ndarray:
        if not isinstance(images, list):
            images = [images]
        return np.stack([np.array(i).astype(np.float32) / 255.0 for i in images], axis=0)

    @staticmethod
    def numpy_to_pt(images: np.ndarray) -> torch.Tensor:
        if images.ndim == 3:
            images = images[..., None]
        return torch.from_numpy(images.transpose(0, 3, 1, 2))

    @staticmethod
    def pt_to_numpy(images: torch.Tensor) -> np.ndarray:
        return images.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def normalize(images):
        return 2.0 * images - 1.0

    @staticmethod
    def denormalize(images):
        return (images / 2 + 0.5).clamp(0, 1)

    def get_default_height_width(self, image, height: Optional[int] = None, width: Optional[int] = None):
        import PIL.Image
        if height is None:
            height = image.height if isinstance(image, PIL.Image.Image) else image.shape[2 if torch.is_tensor(image) else 1]
        if width is None:
            width = image.width if isinstance(image, PIL.Image.Image) else image.shape[3 if torch.is_tensor(image) else 2]
        f = self.config.vae_scale_factor
        return height - height % f, width - width % f

    def resize(self, image, height: int, width: int):
        import PIL.Image
        if isinstance(image, PIL.Image.Image):
            res = {"lanczos": PIL.Image.LANCZOS, "bilinear": PIL.Image.BILINEAR, "bicubic": PIL.Image.BICUBIC,
                   "nearest": PIL.Image.NEAREST}[self.config.resample]
            return image.resize((width, height), resample=res)
        if torch.is_tensor(image):
            return torch.nn.functional.interpolate(image, size=(height, width))
        return self.pt_to_numpy(torch.nn.functional.interpolate(self.numpy_to_pt(image), size=(height, width)))

    # ---- the two entry points the pipelines use
    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        import PIL.Image
        if isinstance(image, (PIL.Image.Image, np.ndarray, torch.Tensor)):
            image = [image]
        if not isinstance(image, list) or not image:
            raise ValueError("image must be a PIL image, an ndarray, a tensor or a list of those")
        first = image[0]
        if isinstance(first, PIL.Image.Image):
            if self.config.do_convert_rgb:
                image = [i.convert("RGB") for i in image]
            elif self.config.do_convert_grayscale:
                image = [i.convert("L") for i in image]
            if self.config.do_resize:
                height, width = self.get_default_height_width(image[0], height, width)
                image = [self.resize(i, height, width) for i in image]
            t = self.numpy_to_pt(self.pil_to_numpy(image))
        elif isinstance(first, np.ndarray):
            arr = np.concatenate(image, axis=0) if first.ndim == 4 else np.stack(image, axis=0)
            t = self.numpy_to_pt(arr.astype(np.float32))
            if self.config.do_resize:
                height, width = self.get_default_height_width(t, height, width)
                t = self.resize(t, height, width)
        elif torch.is_tensor(first):
            t = torch.cat(image, dim=0) if first.ndim == 4 else torch.stack(image, dim=0)
            if t.ndim == 3:                                   # [B, H, W] masks
                t = t.unsqueeze(1)
            if self.config.do_resize:
                height, width = self.get_default_height_width(t, height, width)
                t = self.resize(t.float(), height, width)
        else:
            raise ValueError(f"unsupported image type {type(first)}")
        if self.config.do_normalize and not (torch.is_tensor(first) and t.min() < 0):
            t = self.normalize(t)                             # tensors already in [-1, 1] are left alone
        if self.config.do_binarize:
            t = (t >= 0.5).to(t.dtype)
        return t

    def postprocess(self, image: torch.Tensor, output_type: str = "pil",
                    do_denormalize: Optional[List[bool]] = None):
        if not torch.is_tensor(image):
            raise ValueError("postprocess expects the decoded image tensor [B, C, H, W]")
        if output_type == "latent":
            return image
        if output_type not in ("pt", "np", "pil"):
            output_type = "np"
        if do_denormalize is None:
            do_denormalize = [self.config.do_normalize] * image.shape[0]
        image = torch.stack([self.denormalize(image[i]) if do_denormalize[i] else image[i]
                             for i in range(image.shape[0])])
        if output_type == "pt":
            return image
        arr = self.pt_to_numpy(image)
        return arr if output_type == "np" else self.numpy_to_pil(arr)
