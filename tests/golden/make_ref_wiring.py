#!/usr/bin/env python
"""Generate tests/golden/ref_wiring.pt from the REFERENCE'S OWN source files (run in the build container only).

The reference model code (/root/reference/powerpaint/models/{unet_2d_condition,unet_2d_blocks,BrushNet_CA}.py) is
imported unmodified through oracle/ref_shim.py (a `diffusers` stand-in whose leaf modules are the oracle's).  Weights
come from the ORACLE's seeded constructors (so tests can regenerate them without the reference), matrix weights rounded
to bf16 (the precision the HIP path stores them in).  Stored: inputs and the reference's outputs for
  (1) the 9-channel inpainting UNet forward,
  (2) BrushNetModel.from_unet(...) [weight copy checked], its 8 + 1 + 11 residuals (zero-convs randomised), and
  (3) the 4-channel UNet forward consuming those residuals through down/mid/up_block_add_samples.
Architecture: SD-1.5 topology (4 levels, cross-attention in the first three) at reduced widths, 1 layer per block.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim, sd_modules as OM  # noqa: E402

CFG = dict(block_out_channels=(320, 320, 640, 640), layers_per_block=1, cross_attention_dim=768, attention_head_dim=8)
OCFG = dict(CFG, down_block_types=OM.SD15["down_block_types"], up_block_types=OM.SD15["up_block_types"])


def bf16_(m):
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(p.to(torch.bfloat16).float())
    return m


def oracle_models():
    torch.manual_seed(20240101)
    u9 = bf16_(OM.UNet2DConditionModel(in_channels=9, **OCFG)).eval()
    torch.manual_seed(20240102)
    u4 = bf16_(OM.UNet2DConditionModel(in_channels=4, **OCFG)).eval()
    return u9, u4


def inputs():
    g = torch.Generator("cpu").manual_seed(7)
    return dict(x9=torch.randn(2, 9, 16, 16, generator=g), x4=torch.randn(2, 4, 16, 16, generator=g),
                ehs=torch.randn(2, 77, 768, generator=g), ehs_b=torch.randn(2, 77, 768, generator=g),
                cond=torch.randn(2, 5, 16, 16, generator=g), t=torch.tensor(681), scale=0.75)


def summary(t):
    f = t.flatten()
    return torch.cat([torch.stack([f.mean(), f.std(), f.abs().max()]), f[:32]])


def main():
    RU, RB = ref_shim.load_reference_models()
    u9, u4 = oracle_models()
    inp = inputs()
    out = {"inputs": inp, "cfg": CFG}
    with torch.no_grad():
        r9 = RU(in_channels=9, **CFG).eval()
        r9.load_state_dict(u9.state_dict())
        out["eps9"] = r9(inp["x9"], inp["t"], inp["ehs"], return_dict=False)[0]

        r4 = RU(in_channels=4, **CFG).eval()
        r4.load_state_dict(u4.state_dict())
        rb = RB.from_unet(r4).eval()                               # reference weight-copy logic (BrushNet_CA.py:456-542)
        ob = OM.BrushNetModel.from_unet(u4).eval()
        for (k, a), (k2, b) in zip(sorted(rb.state_dict().items()), sorted(ob.state_dict().items())):
            assert k == k2 and torch.equal(a, b), f"from_unet mismatch at {k}"
        OM.randomize_zero_convs(ob, seed=11)
        rb.load_state_dict(ob.state_dict())
        dn, md, up = rb(inp["x4"], inp["t"], encoder_hidden_states=inp["ehs_b"], brushnet_cond=inp["cond"],
                        conditioning_scale=inp["scale"], return_dict=False)
        out["n_down"], out["n_up"] = len(dn), len(up)
        out["res_shapes"] = [tuple(t.shape) for t in list(dn) + [md] + list(up)]
        out["res_summary"] = torch.stack([summary(t) for t in list(dn) + [md] + list(up)])
        out["eps4_brush"] = r4(inp["x4"], inp["t"], inp["ehs"], down_block_add_samples=list(dn), mid_block_add_sample=md,
                               up_block_add_samples=list(up), return_dict=False)[0]
        out["eps4_plain"] = r4(inp["x4"], inp["t"], inp["ehs"], return_dict=False)[0]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_wiring.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes;", out["n_down"], "down /", out["n_up"], "up residuals")


if __name__ == "__main__":
    main()
