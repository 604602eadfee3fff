"""CPU oracle for the PowerPaint denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch CPU fp32 restatement of the reference algorithm
(structure from /root/reference/powerpaint/..., leaf-module math from the pinned,
un-vendored dependency diffusers==0.27.0 -- see SURVEY.md section 8c / Appendix B).

PARITY STATUS: *parity unpinned by reference golden vectors* -- the reference ships no
tests, fixtures or golden tensors.  What pins this oracle instead:
  * tests/golden/ref_wiring_*.pt : outputs of the REFERENCE'S OWN python files
    (powerpaint/models/unet_2d_condition.py, unet_2d_blocks.py, BrushNet_CA.py)
    imported in the build container through a `diffusers` shim whose leaf modules
    are this oracle's leaves (oracle/ref_shim.py, tests/golden/make_ref_wiring.py).
    This pins every fork-specific behaviour (BrushNet residual routing, pop order,
    "first skip excludes the residual", zero-conv placement, from_unet weight copy).
  * torch.nn.functional as op-level truth for the leaf modules,
  * float64 NumPy re-derivations of the DDIM / DPM-Solver++(2M) closed forms,
  * parameter-count checks (UNet 859.5 M = the well-known SD-1.5 figure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (powerpaint_amd/) never does.
"""
