"""-m gpu: `GroupNorm -> SiLU -> conv3x3` as ONE launch (csrc/conv_gn.hip, PPGemmArgs.gn_in_*) through the C ABI.

Two checkers per case: (1) the two launches it replaces -- pp_groupnorm_apply_acc -> pp_gemm_bf16(PP_X_CONV3X3) -- whose
normalised activation is rounded to 16 bits at the same point, so only the fp32 summation order (chunk-major instead of
tap-major) and the reciprocal of the SiLU differ: at most one 16-bit ulp on a small fraction of the outputs; (2) plain
fp32 torch (F.group_norm / F.silu / F.conv2d) on the same 16-bit inputs.  gamma / beta are random (the synthetic network
weights use (1, 0), which would hide a channel-indexing bug).

These cases are also the guard of a compiler hazard in the shipping kernel: the per-chunk (scale, shift) table is stored by
ONE lane per k-slot and read by all lanes of the wave -- a cross-lane hand-off through LDS that is ordered only by the
`asm volatile("s_waitcnt lgkmcnt(0)")` every lane executes behind the storing lanes' branch (csrc/conv_gn.hip:204-208).
Without that fence hipcc ran the other lanes' table reads first (stale coefficients: 40 of these 50 cases failed, round 4).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from powerpaint_amd import _lib as L  # noqa: E402
from powerpaint_amd import ops  # noqa: E402

DEV = "cuda"
T256, T128, T64 = L.PP_TILE_256x160, L.PP_TILE_128x160, L.PP_TILE_64x160


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def gn_acc(x, groups=32):
    """int64 [B][groups][2] fixed-point (sum, sum of squares) of x [B,H,W,C], as the producers' epilogues leave them."""
    B, H, W, C = x.shape
    xf = x.double().reshape(B, H * W, groups, C // groups)
    return torch.stack([(xf.sum((1, 3)) * 2 ** 24).round().long(), ((xf * xf).sum((1, 3)) * 2 ** 20).round().long()],
                       -1).contiguous()


def close(out, ref, atol, rtol, what):
    out, ref = out.float(), ref.float()
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    err = (out - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} off; max abs err {float(err.max()):.4g} "
                           f"(ref max {float(ref.abs().max()):.4g})")


def run_case(B, H, W, C1, C2, Cout, dtype=torch.bfloat16, tile=0, splitk=0, tail=(0, 0), epi=False, gn_out=False, seed=0):
    groups, eps = 32, 1e-5
    x1 = (rnd(B, H, W, C1, seed=seed + 1, scale=1.5) + 0.3).to(dtype)
    x2 = (rnd(B, H, W, C2, seed=seed + 2, scale=0.7) - 0.2).to(dtype) if C2 else None
    xc = torch.cat([x1, x2], -1) if C2 else x1
    Ct = C1 + C2
    acc = gn_acc(xc, groups)
    g, b = rnd(Ct, seed=seed + 3) * 0.3 + 1.0, rnd(Ct, seed=seed + 4) * 0.3
    C3, C4 = tail
    x3 = rnd(B, H, W, C3, seed=seed + 5).to(dtype) if C3 else None
    x4 = rnd(B, H, W, C4, seed=seed + 6).to(dtype) if C4 else None
    K = 9 * Ct + C3 + C4
    w = rnd(Cout, K, seed=seed + 7, scale=K ** -0.5).to(dtype).contiguous()
    kw = {}
    if epi:
        kw = dict(rowvec=rnd(B, Cout, seed=seed + 9), res1=rnd(B, H, W, Cout, seed=seed + 10).to(dtype),
                  res2=rnd(B, H, W, Cout, seed=seed + 11).to(dtype))
    bias = rnd(Cout, seed=seed + 8)
    assert ops.conv_gn_supported(x1, Cout, x2=x2, x3=x3, x4=x4)
    acc_new = acc_old = None
    if gn_out:      # the statistics of the OUTPUT for two downstream norms (plain, and as channels 64.. of a wider concat)
        acc_new = [torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV) for _ in range(2)]
        acc_old = [torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV) for _ in range(2)]
        sub = lambda A: [(A[0], Cout // 32, 0, 32), (A[1], (Cout + 64) // 32, 64, 32)]   # noqa: E731
    out = ops.conv3x3(x1, w, bias, x2=x2, x3=x3, x4=x4, tile=tile, splitk=splitk, gn_in=(acc, ops.gn_gamma_beta(g, b), groups, eps),
                      gn=sub(acc_new) if gn_out else None, **kw)
    out2 = ops.conv3x3(x1, w, bias, x2=x2, x3=x3, x4=x4, tile=tile, splitk=splitk, gn_in=(acc, ops.gn_gamma_beta(g, b), groups, eps),
                       gn=sub([torch.zeros_like(a) for a in acc_new]) if gn_out else None, **kw)
    assert torch.equal(out, out2), "fused conv: not deterministic"
    # (1) the two launches it replaces
    y = ops.groupnorm_apply_acc(x1, acc, g, b, eps, True, x2=x2)
    old = ops.conv3x3(y, w, bias, x3=x3, x4=x4, gn=sub(acc_old) if gn_out else None, **kw)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    close(out, old, 2 * ulp, 1.5 * ulp, "fused vs groupnorm_apply_acc + conv3x3")
    frac = (out != old).float().mean().item()
    assert frac < 0.08, f"fused vs two-launch: {frac:.3f} of the outputs differ (expected rare one-ulp flips)"
    # (2) fp32 torch
    yn = F.silu(F.group_norm(xc.float().permute(0, 3, 1, 2), groups, g, b, eps))
    ref = F.conv2d(yn, w[:, :9 * Ct].float().reshape(Cout, 3, 3, Ct).permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    if C3:
        xt = torch.cat([x3, x4], -1) if C4 else x3
        ref = ref + xt.float() @ w[:, 9 * Ct:].float().t()
    if epi:
        ref = ref + kw["rowvec"].view(B, 1, 1, Cout) + kw["res1"].float() + kw["res2"].float()
    tol = 1.0 if dtype == torch.bfloat16 else 0.25
    close(out, ref, 3e-2 * tol, 1e-2 * tol, "fused vs fp32 torch")
    if gn_out:
        for k in range(2):
            a, o = acc_new[k].double(), acc_old[k].double()
            rel = ((a - o).abs() / (o.abs() + 2.0 ** 20)).max().item()
            assert rel < 2e-3, f"output GroupNorm statistics (consumer {k}) differ from the two-launch path: {rel:.3g}"
    return out


# the shapes of the SD-1.5 levels (reduced batch), every tile size the chooser can pick, with and without split-K
@pytest.mark.parametrize("B,H,W,C,Cout,tile,splitk", [
    (2, 64, 64, 320, 320, 0, 0),          # 64x64 level: 256-row tiles = 4 image rows
    (1, 64, 64, 64, 320, T256, 1),        # a single chunk (no next halo tile at all)
    (1, 64, 64, 128, 160, T256, 2),       # one chunk per split
    (2, 32, 32, 640, 640, 0, 0),          # 32x32: 128-row tiles
    (2, 32, 32, 640, 640, T256, 2),       # ... or 256-row tiles (8 image rows) split over the chunks
    (2, 16, 16, 1280, 1280, 0, 0),        # 16x16: one image per tile, split-K 4
    (2, 16, 16, 320, 320, T128, 1),
    (2, 8, 8, 1280, 1280, 0, 0),          # 8x8: 64-row tiles, two image rows per 16-row fragment
    (3, 8, 8, 256, 160, T64, 1),
    (1, 48, 32, 192, 320, 0, 0),          # non-square, 6 tiles per image
    (1, 96, 64, 128, 320, T256, 1),
    (1, 16, 128, 64, 160, T128, 1),       # W = 128 (config 5): one image row per tile
])
def test_conv_gn_levels(B, H, W, C, Cout, tile, splitk):
    run_case(B, H, W, C, 0, Cout, tile=tile, splitk=splitk)


@pytest.mark.parametrize("C1,C2", [(320, 320), (1280, 640), (640, 320), (64, 64)])
def test_conv_gn_concat_sources(C1, C2):
    """conv1 of the up blocks: GroupNorm over concat(hidden, skip) -- group boundaries that straddle the two tensors
    ((1280 + 640) / 32 = 60 channels per group), chunks from either source."""
    run_case(2, 16, 16, C1, C2, 320, seed=10)
    run_case(1, 32, 32, C1, C2, 160, tile=T128, splitk=1, seed=20)


@pytest.mark.parametrize("tile,splitk", [(0, 0), (T256, 1), (T256, 3), (T128, 1), (T128, 2), (T64, 1)])
@pytest.mark.parametrize("tail", [(640, 0), (640, 320), (64, 0)])
def test_conv_gn_with_1x1_tail(tile, splitk, tail):
    """conv2 with ResnetBlock2D.conv_shortcut merged in: the (un-normalised) block input rides as a 1x1 K tail."""
    run_case(2, 16, 16, 320, 0, 320, tile=tile, splitk=splitk, tail=tail, seed=30)


@pytest.mark.parametrize("tile,splitk", [(0, 0), (T256, 1), (T128, 1), (T256, 4), (T64, 2)])
def test_conv_gn_epilogue_and_output_statistics(tile, splitk):
    """bias + time-embedding row vector + two residuals, and the GroupNorm statistics of the output for two consumers
    (from the epilogue, or from the split-K combine)."""
    run_case(2, 16, 16, 320, 0, 320, tile=tile, splitk=splitk, epi=True, gn_out=True, seed=40)
    run_case(2, 32, 32, 128, 64, 640, tile=tile, splitk=min(splitk, 3), epi=True, gn_out=True, tail=(128, 0), seed=50)


@pytest.mark.parametrize("B,H,W,C,Cout", [(2, 32, 32, 320, 320), (1, 64, 64, 128, 160), (2, 8, 8, 640, 320)])
def test_conv_gn_fp16(B, H, W, C, Cout):
    run_case(B, H, W, C, 0, Cout, dtype=torch.float16, epi=True, seed=60)


def test_conv_gn_padding_is_of_the_normalised_tensor():
    """gamma = 0, beta with silu(beta) = 1: the normalised activation is exactly 1 inside the image, so with all-ones
    weights every output equals (valid taps) * C -- the zero padding must apply AFTER the normalisation, on every border,
    for every tile size (catches halo-row, out-of-row-lane and tap-offset errors exactly)."""
    beta0 = 1.2784645427610738      # silu(beta0) = 1
    for (H, W, tile) in [(64, 64, T256), (32, 32, T128), (16, 16, T256), (8, 8, T64), (16, 16, T64), (32, 32, T256)]:
        B, C, Cout = 2, 128, 160
        x = rnd(B, H, W, C, seed=3).to(torch.bfloat16)
        g, b = torch.zeros(C, device=DEV), torch.full((C,), beta0, device=DEV)
        w = torch.full((Cout, 9 * C), 1.0 / 128, dtype=torch.bfloat16, device=DEV)
        out = ops.conv3x3(x, w, None, tile=tile, splitk=1, gn_in=(gn_acc(x), ops.gn_gamma_beta(g, b), 32, 1e-5)).float()
        cnt = F.conv2d(torch.ones(1, 1, H, W, device=DEV), torch.ones(1, 1, 3, 3, device=DEV), padding=1)[0, 0]
        for bi in range(B):
            for ch in (0, Cout - 1):
                assert torch.equal(out[bi, :, :, ch], cnt), (H, W, tile, bi, ch, (out[bi, :, :, ch] - cnt).abs().max())


def test_conv_gn_unsupported_shapes_are_refused():
    x = rnd(1, 24, 40, 64).to(torch.bfloat16)          # W = 40: no tile of whole image rows divides 24 x 40
    assert not ops.conv_gn_supported(x, 320)
    w = rnd(320, 9 * 64).to(torch.bfloat16)
    with pytest.raises(L.PPError):
        ops.conv3x3(x, w, None, gn_in=(gn_acc(x), ops.gn_gamma_beta(torch.ones(64, device=DEV), torch.zeros(64, device=DEV)), 32, 1e-5))


def test_conv_gn_benchmark_shape_full_batch():
    """The launch configuration of the headline benchmark: batch 8 at 64x64, C = 320 (256 workgroups, one per CU)."""
    run_case(8, 64, 64, 320, 0, 320, epi=True, gn_out=True, seed=70)
    run_case(8, 64, 64, 640, 320, 320, seed=80)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,C,Cout,silu,fused_in", [(8, 1280, 1280, True, False), (16, 640, 1280, False, True), (16, 320, 320, True, True),
                                                   (8, 256, 640, False, False), (8, 320, 320, True, True)])
def test_groupnorm_apply_inside_the_splitk_combine(H, C, Cout, silu, fused_in, dtype):
    """PPGemmArgs.gn_next_* (ABI v17): the GroupNorm (+ SiLU) that consumes a split-K conv's output, applied by the combine
    at the levels where one workgroup owns a whole (batch item, 160-column tile).  Bit-identical to pp_groupnorm_apply_acc on
    the raw output with the accumulators the same launch filled; the raw output and the accumulators themselves are those of
    the plain split-K launch.  Producer = the tap-major conv (8x8 level) or the fused norm -> SiLU -> conv (16x16)."""
    B, groups = 3, 32
    x = (rnd(B, H, H, C, seed=1, scale=1.3) + 0.2).to(dtype)
    K = 9 * C
    w = rnd(Cout, K, seed=2, scale=K ** -0.5).to(dtype).contiguous()
    bias, rv = rnd(Cout, seed=3), rnd(B, Cout, seed=4)
    res = rnd(B, H, H, Cout, seed=5).to(dtype)
    g2, b2 = rnd(Cout, seed=6) * 0.3 + 1.0, rnd(Cout, seed=7) * 0.3
    kw = dict(rowvec=rv, res1=res, splitk=2)
    if fused_in:
        gi, bi = rnd(C, seed=8) * 0.3 + 1.0, rnd(C, seed=9) * 0.3
        kw["gn_in"] = (gn_acc(x), ops.gn_gamma_beta(gi, bi), groups, 1e-5)
    acc_a = [torch.zeros(B, groups, 2, dtype=torch.int64, device=DEV) for _ in range(2)]
    acc_b = [torch.zeros(B, groups, 2, dtype=torch.int64, device=DEV) for _ in range(2)]
    sub = lambda A: [(A[0], (Cout + 64) // groups, 64, groups), (A[1], Cout // groups, 0, groups)]   # noqa: E731
    out, y = ops.conv3x3(x, w, bias, gn=sub(acc_a), gn_next=(g2, b2, 1e-5, silu, 1), **kw)
    old = ops.conv3x3(x, w, bias, gn=sub(acc_b), **kw)
    assert torch.equal(out, old), "raw output differs from the plain split-K launch"
    for k in range(2):
        assert torch.equal(acc_a[k], acc_b[k]), f"accumulators of subscription {k} differ"
    y_ref = ops.groupnorm_apply_acc(old, acc_b[1], g2, b2, 1e-5, silu)
    assert torch.equal(y, y_ref), f"normalised tensor differs: max {float((y.float() - y_ref.float()).abs().max()):.4g}"
    yt = F.group_norm(old.float().permute(0, 3, 1, 2), groups, g2, b2, 1e-5)
    yt = (F.silu(yt) if silu else yt).permute(0, 2, 3, 1)
    tol = 1.0 if dtype == torch.bfloat16 else 0.25
    close(y, yt, 3e-2 * tol, 1.2e-2 * tol, "combine + apply vs fp32 torch")


def test_groupnorm_apply_in_combine_is_refused_where_it_cannot_run():
    x = rnd(2, 32, 32, 128).to(torch.bfloat16)          # 1024 rows per batch item: no workgroup owns a group's population
    w = rnd(320, 9 * 128, scale=0.03).to(torch.bfloat16)
    acc = torch.zeros(2, 32, 2, dtype=torch.int64, device=DEV)
    with pytest.raises(L.PPError):
        ops.conv3x3(x, w, None, splitk=2, gn=[(acc, 10, 0, 32)],
                    gn_next=(torch.ones(320, device=DEV), torch.zeros(320, device=DEV), 1e-5, True, 0))


# ---- round 6: a PLAIN conv3x3 on the halo-tile loop without the normalisation (pp_conv_gn_supported() == 2) ---------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,C1,C2,Cout,tail,splitk", [
    (2, 64, 320, 0, 320, (0, 0), 0),          # 64x64 level
    (2, 64, 640, 320, 320, (0, 0), 0),        # ... on a concatenated input (up block)
    (2, 64, 320, 0, 320, (640, 320), 0),      # ... with conv_shortcut as a 1x1 K tail
    (8, 32, 640, 0, 640, (0, 0), 0),          # 32x32 level, 128-row tiles
    (8, 32, 1280, 640, 640, (0, 0), 0),       # long K: 256-row tiles x 2 splits
    (2, 32, 640, 0, 640, (320, 0), 2),
    (8, 16, 1280, 0, 1280, (0, 0), 0),        # 16x16 level: 256-row tiles x 4 splits
    (8, 16, 1280, 640, 1280, (1280, 640), 0),
])
def test_plain_conv_on_the_halo_tile_loop(B, H, C1, C2, Cout, tail, splitk, dtype):
    """ResnetBlock2D's conv behind a separate GroupNorm apply (the 32x32 level since round 6; unet_2d_blocks.py:1274-1285):
    the library runs it on conv_gn.hip's loop with the normalisation compiled out.  Same request, same contract: against the
    tap-major implicit GEMM it replaces (an explicit tile code keeps that kernel) and against fp32 torch, with epilogue
    operands and the statistics of the output."""
    x1 = rnd(B, H, H, C1, seed=1).to(dtype)
    x2 = rnd(B, H, H, C2, seed=2).to(dtype) if C2 else None
    C3, C4 = tail
    x3 = rnd(B, H, H, C3, seed=3).to(dtype) if C3 else None
    x4 = rnd(B, H, H, C4, seed=4).to(dtype) if C4 else None
    Ct = C1 + C2
    K = 9 * Ct + C3 + C4
    w = rnd(Cout, K, seed=5, scale=K ** -0.5).to(dtype).contiguous()
    bias, rv = rnd(Cout, seed=6), rnd(B, Cout, seed=7)
    res = rnd(B, H, H, Cout, seed=8).to(dtype)
    assert ops.conv_halo_routed(x1, Cout, x2=x2, x3=x3, x4=x4)
    assert not ops.conv_halo_routed(x1, Cout, x2=x2, x3=x3, x4=x4, tile=T128)          # an explicit tile keeps the tap-major kernel
    A = [torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV) for _ in range(2)]
    out = ops.conv3x3(x1, w, bias, x2=x2, x3=x3, x4=x4, rowvec=rv, res1=res, splitk=splitk, gn=[(A[0], Cout // 32, 0, 32)])
    again = ops.conv3x3(x1, w, bias, x2=x2, x3=x3, x4=x4, rowvec=rv, res1=res, splitk=splitk)
    assert torch.equal(out, again), "not deterministic"
    old = ops.conv3x3(x1, w, bias, x2=x2, x3=x3, x4=x4, rowvec=rv, res1=res, tile=54, gn=[(A[1], Cout // 32, 0, 32)])
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    close(out, old, 2 * ulp, 1.5 * ulp, "halo-tile loop vs tap-major implicit GEMM")
    rel = ((A[0].double() - A[1].double()).abs() / (A[1].double().abs() + 2.0 ** 20)).max().item()
    assert rel < 2e-3, f"output GroupNorm statistics differ from the tap-major path: {rel:.3g}"
    xc = torch.cat([x1, x2], -1) if C2 else x1
    ref = F.conv2d(xc.float().permute(0, 3, 1, 2), w[:, :9 * Ct].float().reshape(Cout, 3, 3, Ct).permute(0, 3, 1, 2), bias,
                   padding=1).permute(0, 2, 3, 1)
    if C3:
        xt = torch.cat([x3, x4], -1) if C4 else x3
        ref = ref + xt.float() @ w[:, 9 * Ct:].float().t()
    ref = ref + rv.view(B, 1, 1, Cout) + res.float()
    tol = 1.0 if dtype == torch.bfloat16 else 0.25
    close(out, ref, 3e-2 * tol, 1e-2 * tol, "halo-tile loop vs fp32 torch")


def test_plain_convs_that_stay_on_the_tap_major_kernel():
    """The 8x8 level (split-K weight streams: the tap-major kernel's 128-row tiles in N-major order win), strided convs,
    channel counts off the 64 grid, the CFG twin store: not the halo-tile loop's -- routed as before, results as before."""
    x8 = rnd(8, 8, 8, 1280, seed=1).to(torch.bfloat16)
    assert not ops.conv_halo_routed(x8, 1280)
    x64 = rnd(1, 64, 64, 320, seed=2).to(torch.bfloat16)
    assert not ops.conv_halo_routed(x64, 320, stride=2)
    assert not ops.conv_halo_routed(rnd(1, 64, 64, 96, seed=3).to(torch.bfloat16), 320)
    w = rnd(320, 9 * 320, seed=4, scale=0.02).to(torch.bfloat16)
    out = ops.conv3x3(x64, w, None, dup=True)                                           # out_dup_rows: single-pass v2 epilogue
    ref = ops.conv3x3(x64, w, None, tile=54)
    assert torch.equal(out[0], out[1])
    close(out[:1], ref, 2 * 2.0 ** -8, 1.5 * 2.0 ** -8, "twin-store conv vs plain")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,C,Cout", [(2, 32, 320, 320), (8, 16, 640, 640), (8, 8, 1280, 1280), (1, 16, 64, 160)])
def test_upsampling_conv_on_the_halo_tile_loop(B, H, C, Cout, dtype):
    """Upsample2D: `F.interpolate(x, scale_factor=2.0, mode="nearest")` then `Conv2d(3x3)` (diffusers 0.27; ctor sites
    /root/reference/powerpaint/models/unet_2d_blocks.py:2406, 2542).  The halo tile of the 2H x 2W image is gathered from
    the H x W source (halo pixel (y, x) = source pixel (y / 2, x / 2)); zero padding applies to the upsampled image."""
    x = rnd(B, H, H, C, seed=1).to(dtype)
    w = rnd(Cout, 9 * C, seed=2, scale=(9 * C) ** -0.5).to(dtype).contiguous()
    bias = rnd(Cout, seed=3)
    assert ops.conv_halo_routed(x, Cout, up=True)
    out = ops.conv3x3(x, w, bias, up=True)
    assert out.shape == (B, 2 * H, 2 * H, Cout) and torch.equal(out, ops.conv3x3(x, w, bias, up=True))
    old = ops.conv3x3(x, w, bias, up=True, tile=54)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    close(out, old, 2 * ulp, 1.5 * ulp, "halo-tile loop vs tap-major implicit GEMM (upsampling)")
    xu = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xu, w.float().reshape(Cout, 3, 3, C).permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    tol = 1.0 if dtype == torch.bfloat16 else 0.25
    close(out, ref, 3e-2 * tol, 1e-2 * tol, "halo-tile loop (upsampling) vs fp32 torch")
