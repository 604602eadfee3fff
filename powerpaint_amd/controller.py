"""Controller pre / post-processing around the pipelines: what `PowerPaintController.predict` of
/root/reference/app.py:245-387 does to a user's image and mask before it calls the pipeline, and to the result after.

Pure host-side image plumbing (PIL / NumPy) -- nothing here touches the GPU; the pipeline it drives does.  The Gradio
UI, the model download / construction (`powerpaint_amd.loaders` covers the loading calls) and the ControlNet annotators
(Canny / OpenPose / HED / depth: third-party models) stay out of scope: `predict_controlnet` takes the already computed
control image.

    fit_short_side      app.py:258-268   resize so that the short side is 640 (512 for outpainting), aspect kept
    outpaint_canvas     app.py:270-304   grey canvas enlarged by the expansion ratios + the mask that keeps the original
                                         (shrunk by a 10-pixel "blurry gap" along every expanded axis)
    snap_to_eight       app.py:314-318   both sides rounded down to multiples of 8, image and mask resized to it
    overlay_mask        app.py:366-377   the red-tinted preview of the masked region returned next to the result
"""
from typing import Callable, Optional

import numpy as np

from .utils.utils import add_task

BLURRY_GAP = 10


def fit_short_side(image, outpainting: bool):
    """RGB copy of `image` whose shorter side is 640 pixels (512 when outpainting); the longer side is truncated, not
    rounded (app.py:258-268)."""
    image = image.convert("RGB")
    w, h = image.size
    target = 512 if outpainting else 640
    if w < h:
        return image.resize((target, int(h / w * target)))
    return image.resize((int(w / h * target), target))


def outpaint_canvas(image, vertical_expansion_ratio: float, horizontal_expansion_ratio: float):
    """(canvas image, canvas mask) for outpainting (app.py:270-304).  The original is centred on a grey (127) canvas of
    int(ratio * size); the mask is white (255 = repaint) except over the original, where it is black on the part that
    stays -- the whole original along an axis that is not expanded, the original minus a 10-pixel rim along one that is.
    With both ratios equal to 1 the mask stays all white, as in the reference."""
    import PIL.Image
    src = np.array(image)
    oh, ow = src.shape[:2]
    cw, ch = int(horizontal_expansion_ratio * ow), int(vertical_expansion_ratio * oh)
    top, left = int((ch - oh) / 2.0), int((cw - ow) / 2.0)
    canvas = np.full((ch, cw, 3), 127, dtype=np.uint8)
    canvas[top:top + oh, left:left + ow, :] = src
    mask = np.full((ch, cw, 3), 255, dtype=np.uint8)
    grow_v, grow_h = vertical_expansion_ratio != 1, horizontal_expansion_ratio != 1
    if grow_v or grow_h:
        gv, gh = (BLURRY_GAP if grow_v else 0), (BLURRY_GAP if grow_h else 0)
        mask[top + gv:top + oh - gv, left + gh:left + ow - gh, :] = 0
    return PIL.Image.fromarray(canvas), PIL.Image.fromarray(mask)


def snap_to_eight(image, mask):
    """Resize image and mask to (width, height) rounded down to multiples of 8 (app.py:314-318).  Returns them and
    (width, height)."""
    w, h = image.size
    w, h = w - w % 8, h - h % 8
    return image.resize((w, h)), mask.resize((w, h)), (w, h)


def overlay_mask(result, mask):
    """Preview: result with the masked region tinted red, result * (1 - m/512) + m/512 * (180, 0, 0) (app.py:366-377)."""
    import PIL.Image
    m = np.array(mask.convert("RGB")).astype("float")
    res = np.array(result).astype("float")
    red = np.zeros_like(res)
    red[:, :, 0] = 180.0
    return PIL.Image.fromarray((res * (1 - m / 512.0) + m / 512.0 * red).astype("uint8"))


def paste_back(result, image, mask, radius: int):
    """Result blended over the input through the Gaussian-blurred mask (app.py:378-382 / 468-474)."""
    import PIL.Image
    import PIL.ImageFilter
    soft = np.asarray(mask.convert("RGB").filter(PIL.ImageFilter.GaussianBlur(radius=radius))) / 255.0
    img = np.asarray(image.convert("RGB")) / 255.0
    out = np.asarray(result) / 255.0
    return PIL.Image.fromarray(np.uint8((out * soft + (1 - soft) * img) * 255))


class PowerPaintController:
    """`predict` / `predict_controlnet` / `infer` of the reference controller over pipelines the caller built (e.g.
    with `powerpaint_amd.models.*.from_pretrained` + `powerpaint_amd.pipelines.*`)."""

    def __init__(self, pipe, version: str = "ppt-v1", control_pipe=None, seed_fn: Optional[Callable[[int], None]] = None,
                 generator_device: str = "cuda"):
        self.pipe, self.version, self.control_pipe = pipe, version, control_pipe
        self.seed_fn = seed_fn or _set_seed
        self.generator_device = generator_device

    def predict(self, input_image: dict, prompt: str, fitting_degree: float, ddim_steps: int, scale: float, seed: int,
                negative_prompt: str, task: str, vertical_expansion_ratio: Optional[float] = None,
                horizontal_expansion_ratio: Optional[float] = None):
        """app.py:245-387.  `input_image` = {"image": PIL, "mask": PIL}; returns ([result], [mask, preview])."""
        image = fit_short_side(input_image["image"], task == "image-outpainting")
        mask = input_image["mask"]
        if vertical_expansion_ratio is not None and horizontal_expansion_ratio is not None:
            image, mask = outpaint_canvas(image, vertical_expansion_ratio, horizontal_expansion_ratio)
        if self.version != "ppt-v1":                                           # :306-310
            prompt = prompt + {"image-outpainting": " empty scene", "object-removal": " empty scene blur"}.get(task, "")
        pA, pB, nA, nB = add_task(prompt, negative_prompt, task, self.version)
        image, mask, (w, h) = snap_to_eight(image, mask)
        input_image["image"], input_image["mask"] = image, mask                # the reference updates the dict in place
        self.seed_fn(seed)
        common = dict(promptA=pA, promptB=pB, tradoff=fitting_degree, tradoff_nag=fitting_degree, negative_promptA=nA,
                      negative_promptB=nB, mask=mask.convert("RGB"), width=w, height=h, guidance_scale=scale,
                      num_inference_steps=ddim_steps)
        if self.version == "ppt-v1":
            result = self.pipe(image=image.convert("RGB"), **common).images[0]
        else:
            import PIL.Image
            import torch
            keep = np.array(image) * (1 - np.array(mask) / 255.0)                # :339-342: the BrushNet input is pre-masked
            image = PIL.Image.fromarray(keep.astype(np.uint8)).convert("RGB")
            input_image["image"] = image
            gen = torch.Generator(self.generator_device).manual_seed(seed)
            result = self.pipe(image=image, promptU=prompt, negative_promptU=negative_prompt, generator=gen,
                               brushnet_conditioning_scale=1.0, **common).images[0]
        return [result], [mask.convert("RGB"), overlay_mask(result, mask)]

    def predict_controlnet(self, input_image: dict, control_image, prompt: str, ddim_steps: int, scale: float, seed: int,
                           negative_prompt: str, controlnet_conditioning_scale: float):
        """app.py:389-475 with the annotator output (`control_image`, PIL) supplied by the caller."""
        if self.control_pipe is None:
            raise ValueError("no ControlNet pipeline registered")
        pA = pB = prompt + " P_obj"
        image = fit_short_side(input_image["image"], False)
        image, mask, (w, h) = snap_to_eight(image, input_image["mask"])
        input_image["image"], input_image["mask"] = image, mask
        control_image = control_image.resize((w, h))
        self.seed_fn(seed)
        result = self.control_pipe(promptA=pB, promptB=pA, tradoff=1.0, tradoff_nag=1.0, negative_promptA=negative_prompt,
                                   negative_promptB=negative_prompt, image=image.convert("RGB"),
                                   mask=mask.convert("RGB"), control_image=control_image, width=w, height=h,
                                   guidance_scale=scale, controlnet_conditioning_scale=controlnet_conditioning_scale,
                                   num_inference_steps=ddim_steps).images[0]
        return [image.convert("RGB"), paste_back(result, image, mask, 4)], [control_image, overlay_mask(result, mask)]

    def infer(self, input_image, text_guided_prompt, text_guided_negative_prompt, shape_guided_prompt,
              shape_guided_negative_prompt, fitting_degree, ddim_steps, scale, seed, task, vertical_expansion_ratio,
              horizontal_expansion_ratio, outpaint_prompt, outpaint_negative_prompt, removal_prompt,
              removal_negative_prompt, enable_control=False, input_control_image=None, control_type="canny",
              controlnet_conditioning_scale=None):
        """app.py:477-560: pick the prompt pair of the tab; expansion ratios only reach `predict` for outpainting."""
        prompts = {"text-guided": (text_guided_prompt, text_guided_negative_prompt),
                   "shape-guided": (shape_guided_prompt, shape_guided_negative_prompt),
                   "object-removal": (removal_prompt, removal_negative_prompt),
                   "image-outpainting": (outpaint_prompt, outpaint_negative_prompt)}
        if task not in prompts:
            task = "text-guided"
            prompts[task] = (text_guided_prompt, text_guided_negative_prompt)
        prompt, negative = prompts[task]
        if task == "image-outpainting":
            return self.predict(input_image, prompt, fitting_degree, ddim_steps, scale, seed, negative, task,
                                vertical_expansion_ratio, horizontal_expansion_ratio)
        if enable_control and task == "text-guided" and self.version == "ppt-v1":
            return self.predict_controlnet(input_image, input_control_image, prompt, ddim_steps, scale, seed, negative,
                                           controlnet_conditioning_scale)
        return self.predict(input_image, prompt, fitting_degree, ddim_steps, scale, seed, negative, task, None, None)


def _set_seed(seed: int):
    """app.py:29-35."""
    import random
    import torch
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
