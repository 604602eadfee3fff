"""-m gpu: the split-K combine INSIDE the producing kernel (ABI v21, csrc/gemm_combine.h).

The S splits of an output tile wait for each other at the tile's counter (XCD-local, bounded) and each sums 1 / S of the
tile's rows over the fp32 slabs, in slab order, and runs the combine's epilogue on them.  Contract checked here, launch by
launch, on the launch shapes of the headline step (batch 8, so that tiles % 8 == 0 and every split of a tile shares an
XCD): the raw output, the GroupNorm accumulators and the normalised tensor of `gn_next` are BIT-IDENTICAL to the separate
combine kernels (`pp_splitk_reduce_kernel<true>`, `pp_splitk_reduce_gn_kernel`, `pp_splitk_reduce_gn_apply_kernel`) -- which
are themselves checked against fp32 torch in tests/test_ops_gpu.py / test_conv_gn_gpu.py -- the round-local fields of the
tile counters are zero again after the launch, no share was left abandoned, and a launch whose splits are NOT all resident
at once (512 workgroups: a split gives up waiting, the last arriver combines its share) is still bit-identical.  Reference ops: the ResnetBlock2D convs and FeedForward / proj_out Linears of the 16x16 / 8x8 levels,
/root/reference/powerpaint/models/unet_2d_blocks.py:1457-1500, 850-899, 2696-2770.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from powerpaint_amd import _lib as L  # noqa: E402
from powerpaint_amd import ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def gn_acc(x, groups=32):
    """fixed-point (sum, sumsq) accumulators of an NHWC tensor, as a producer's epilogue leaves them"""
    B, H, W, Cc = x.shape
    v = x.float().reshape(B, H * W, groups, Cc // groups)
    s = (v.sum((1, 3)).double() * 16777216.0).round().long()
    q = ((v * v).sum((1, 3)).double() * 1048576.0).round().long()
    return torch.stack([s, q], -1).contiguous()


def test_workgroup_placement_is_what_the_combine_rests_on():
    assert L.lib().pp_xcd_placement_ok() == 1


def _after(fused_expected=True):
    torch.cuda.synchronize()
    assert ops.last_combine["fused"] == fused_expected, ops.last_combine
    if ops.last_combine["ctr"] is not None:
        # (a counter word: bits 40.. = arrivals so far, monotonic; bits 0..39 = the round's per-XCC arrivals and abandoned
        # shares, taken out again by the last arriver)
        assert int((ops.last_combine["ctr"] & ((1 << 40) - 1)).abs().sum()) == 0, "tile counters not re-armed"
    assert ops.combine_faults(DEV) == 0


# (B, H, C1, C2, Cout, stride, up, tail, gn_in): the split-K conv launches of the headline UNet step
CONVS = [
    ("8x8 resnet conv, 128 rows x 8 splits", 8, 8, 1280, 0, 1280, 1, False, 0, False),
    ("8x8 up-block conv on a concat + 1x1 tail", 8, 8, 1280, 1280, 1280, 1, False, 2560, False),
    ("16x16 fused-norm conv, 256 rows x 4 splits", 8, 16, 1280, 0, 1280, 1, False, 0, True),
    ("16x16 fused-norm conv on a concat + tail", 8, 16, 1280, 640, 1280, 1, False, 1920, True),
    ("16x16 fused-norm conv, short K: 128 rows x 2 splits", 8, 16, 640, 0, 1280, 1, False, 0, True),
    ("32 -> 16 downsample conv, 256 rows x 8 splits", 8, 32, 640, 0, 640, 2, False, 0, False),
    ("8 -> 16 upsample conv", 8, 8, 1280, 0, 1280, 1, True, 0, False),
    ("32x32 fused-norm conv, long K: 256 rows x 2 splits", 8, 32, 1280, 640, 640, 1, False, 0, True),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CONVS, ids=[c[0] for c in CONVS])
def test_conv_combined_in_kernel_equals_the_separate_combine(case, dtype):
    name, B, H, C1, C2, Cout, stride, up, tail, fused_in = case
    groups = 32
    x1 = (rnd(B, H, H, C1, seed=1, scale=1.2) + 0.1).to(dtype)
    x2 = rnd(B, H, H, C2, seed=2).to(dtype) if C2 else None
    ho = (2 * H if up else H) // stride
    x3 = rnd(B, ho, ho, tail, seed=3).to(dtype) if tail else None
    K = 9 * (C1 + C2) + tail
    w = rnd(Cout, K, seed=4, scale=K ** -0.5).to(dtype).contiguous()
    bias, rv = rnd(Cout, seed=5), rnd(B, Cout, seed=6)
    res = rnd(B, ho, ho, Cout, seed=7).to(dtype)
    kw = dict(x2=x2, x3=x3, stride=stride, up=up, rowvec=rv, res1=res)
    if fused_in:
        xc = torch.cat([x1, x2], 3) if C2 else x1
        gi, bi = rnd(C1 + C2, seed=8) * 0.3 + 1.0, rnd(C1 + C2, seed=9) * 0.3
        kw["gn_in"] = (gn_acc(xc), ops.gn_gamma_beta(gi, bi), groups, 1e-5)

    def subs():
        A = [torch.zeros(B, groups, 2, dtype=torch.int64, device=DEV) for _ in range(2)]
        return A, [(A[0], (Cout + 320) // groups, 320, groups), (A[1], Cout // groups, 0, groups)]

    # (1) lean combine: no statistics
    ref = ops.conv3x3(x1, w, bias, **kw)
    out = ops.conv3x3(x1, w, bias, fuse_combine="force", **kw)
    _after()
    assert torch.equal(out, ref), f"{name}: lean combine differs"
    # (2) with the output's GroupNorm statistics (two subscriptions, one of them a concatenated consumer)
    Ar, sr = subs()
    ref = ops.conv3x3(x1, w, bias, gn=sr, **kw)
    Af, sf = subs()
    out = ops.conv3x3(x1, w, bias, gn=sf, fuse_combine="force", **kw)
    _after()
    assert torch.equal(out, ref), f"{name}: output differs (statistics form)"
    for k in range(2):
        assert torch.equal(Af[k], Ar[k]), f"{name}: accumulators of subscription {k} differ"
    # (3) with the consumer norm applied by the combine, where a split-K combine can own whole (item, group) populations
    if ho * ho <= 256:
        g2, b2 = rnd(Cout, seed=10) * 0.3 + 1.0, rnd(Cout, seed=11) * 0.3
        Ar, sr = subs()
        ref, yref = ops.conv3x3(x1, w, bias, gn=sr, gn_next=(g2, b2, 1e-5, True, 1), **kw)
        Af, sf = subs()
        out, y = ops.conv3x3(x1, w, bias, gn=sf, gn_next=(g2, b2, 1e-5, True, 1), fuse_combine="force", **kw)
        # a 128-row tile holds half a 16x16 image: that launch keeps the separate combine + apply (never a wrong one)
        whole = not (ho * ho == 256 and "128 rows" in name)
        _after(fused_expected=whole)
        assert torch.equal(out, ref) and torch.equal(y, yref), f"{name}: combine + apply differs"
        for k in range(2):
            assert torch.equal(Af[k], Ar[k])
        yt = F.silu(F.group_norm(ref.float().permute(0, 3, 1, 2), groups, g2, b2, 1e-5)).permute(0, 2, 3, 1)
        assert float((y.float() - yt).abs().max()) < (6e-2 if dtype == torch.bfloat16 else 1.5e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,hw", [(2048, 1280, 6400, 256), (2048, 1280, 5120, 256), (512, 1280, 10240, 64)])
def test_linear_combined_in_kernel_equals_the_separate_combine(M, N, K, hw, dtype):
    """FeedForward.net[2] . proj_out of the 16x16 level (K = 6400: 128-row tiles x 2 splits) and friends, with the residual
    and the statistics of the resnet norm behind the transformer."""
    x = rnd(M, K, seed=1).to(dtype)
    w = rnd(N, K, seed=2, scale=K ** -0.5).to(dtype)
    bias = rnd(N, seed=3)
    res = rnd(M, N, seed=4).to(dtype)
    B = M // hw
    kw = dict(bias=bias, res1=res, splitk=2 if K >= 6400 else 4, tile=54)
    ref = ops.gemm(x, w, **kw)
    out = ops.gemm(x, w, fuse_combine="force", **kw)
    _after()
    assert torch.equal(out, ref)
    Ar = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    Af = torch.zeros_like(Ar)
    ref = ops.gemm(x, w, gn=[(Ar, N // 32, 0, 32)], rows_per_batch=hw, **kw)
    out = ops.gemm(x, w, gn=[(Af, N // 32, 0, 32)], rows_per_batch=hw, fuse_combine="force", **kw)
    _after()
    assert torch.equal(out, ref) and torch.equal(Af, Ar)
    close = (out.float() - (x.float() @ w.float().t() + bias + res.float())).abs().max()
    assert float(close) < (6e-2 if dtype == torch.bfloat16 else 1.5e-2)


def test_counters_are_re_armed_and_results_reproducible_under_load():
    """The same counters serve launch after launch (every launch leaves them zero); fifty launches interleaved with a
    memory-heavy neighbour on a second stream (uneven load, other tenants in the L2s) give fifty bit-identical results."""
    B, H, C = 8, 8, 1280
    x = rnd(B, H, H, C, seed=1).to(torch.bfloat16)
    w = rnd(C, 9 * C, seed=2, scale=(9 * C) ** -0.5).to(torch.bfloat16)
    res = rnd(B, H, H, C, seed=3).to(torch.bfloat16)
    ref = ops.conv3x3(x, w, None, res1=res)
    first = ops.conv3x3(x, w, None, res1=res, fuse_combine="force")
    _after()
    ctr = ops.last_combine["ctr"]
    assert torch.equal(first, ref)
    junk = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()
    for i in range(50):
        if i % 2:
            with torch.cuda.stream(side):
                junk.add_(1.0)
        out = ops.conv3x3(x, w, None, res1=res, fuse_combine=ctr)
        assert ops.last_combine["fused"]
        assert torch.equal(out, ref), f"launch {i} differs"
    torch.cuda.synchronize()
    assert int((ctr & ((1 << 40) - 1)).abs().sum()) == 0 and ops.combine_faults(DEV) == 0
    assert int((ctr[0::2] >> 40).min()) == int((ctr[0::2] >> 40).max()) == 51 * 8      # 51 launches x 8 splits at every tile


def test_launches_that_cannot_combine_in_kernel_keep_the_separate_combine():
    """tiles % 8 != 0 (batch 3 at 8x8: 2 x 8 tiles... a 5-column-tile width), a 4-wave tile, an activation epilogue: the
    counters are a permission -- pp_gemm_combine_ctr_bytes() says 0 or pp_gemm_bf16 ignores them -- never a wrong sum."""
    x = rnd(3, 8, 8, 640, seed=1).to(torch.bfloat16)
    w = rnd(800, 9 * 640, seed=2, scale=0.02).to(torch.bfloat16)          # 2 m-tiles x 5 n-tiles = 10 tiles
    ref = ops.conv3x3(x, w, None)
    out = ops.conv3x3(x, w, None, fuse_combine="force")
    _after(fused_expected=False)
    assert torch.equal(out, ref)
    xs = rnd(512, 2560, seed=3).to(torch.bfloat16)
    ws = rnd(320, 2560, seed=4, scale=0.02).to(torch.bfloat16)
    ref = ops.gemm(xs, ws, tile=32, splitk=2)                             # 64-row 4-wave tile
    out = ops.gemm(xs, ws, tile=32, splitk=2, fuse_combine="force")
    _after(fused_expected=False)
    assert torch.equal(out, ref)
    ref = ops.gemm(xs, ws, tile=54, splitk=2, act=L.PP_ACT_SILU)          # not the lean epilogue
    out = ops.gemm(xs, ws, tile=54, splitk=2, act=L.PP_ACT_SILU, fuse_combine="force")
    _after(fused_expected=False)
    assert torch.equal(out, ref)


def test_where_the_library_advises_the_in_kernel_combine():
    """pp_gemm_combine_ctr_bytes() = what a planner is told.  Measured kernel by kernel inside the headline step's graph
    (profiles/r06_fused_combine.txt): launches of 2 and 4 splits win 0.5 .. 8 us against the separate combine, launches of 8
    splits (the 8x8 level, the 32 -> 16 downsample) lose 2 .. 3 us: advised for the former only, and only where every split
    of the launch is resident at once.  Counters handed over anyway are honoured (every test above)."""
    x = rnd(8, 8, 8, 1280, seed=1).to(torch.bfloat16)
    w = rnd(1280, 9 * 1280, seed=2, scale=0.01).to(torch.bfloat16)
    ops.conv3x3(x, w, None, fuse_combine=True)                             # 128 rows x 8 splits
    _after(fused_expected=False)
    xs, ws = rnd(2048, 6400, seed=3).to(torch.bfloat16), rnd(1280, 6400, seed=4, scale=0.01).to(torch.bfloat16)
    ref = ops.gemm(xs, ws)
    out = ops.gemm(xs, ws, fuse_combine=True)                              # 128 rows x 2 splits, 256 workgroups
    _after(fused_expected=True)
    assert torch.equal(out, ref)
    ops.gemm(xs, ws, tile=54, splitk=4, fuse_combine=True)                 # 512 workgroups: the splits are not co-resident
    _after(fused_expected=False)


def test_host_predicates_agree():
    """pp_gemm_combine_ctr_bytes / pp_gemm_combine_fused are pure host logic over the request (plus the cached probe)."""
    lib = L.lib()
    a = L.PPGemmArgs()
    a.M, a.N, a.K, a.x_mode, a.c1, a.ldx1 = 2048, 1280, 6400, L.PP_X_PLAIN, 6400, 6400
    a.x1 = a.w = a.out = 4096
    a.ldo = a.ldres1 = a.ldres2 = 1280
    a.scale, a.dtype = 1.0, L.PP_DT_BF16
    # the automatic choice: 128 rows x 2 splits = 128 tiles; two fp32 slabs + 6 KB of statistics scratch per tile
    assert lib.pp_gemm_workspace_bytes(C.byref(a)) == 2 * 2048 * 1280 * 4 + 128 * 6144
    assert lib.pp_gemm_combine_ctr_bytes(C.byref(a)) == 128 * 16               # advised: two counter words per tile
    assert lib.pp_gemm_combine_fused(C.byref(a)) == 0                          # no counters given
    a.tile_ctr = 4096
    assert lib.pp_gemm_combine_fused(C.byref(a)) == 1
    a.splitk, a.tile = 8, 54
    assert lib.pp_gemm_combine_ctr_bytes(C.byref(a)) == 0                      # 8 splits: not advised ...
    assert lib.pp_gemm_combine_fused(C.byref(a)) == 1                          # ... but honoured where handed over
    a.splitk, a.tile = 0, 0
    a.act = L.PP_ACT_SILU
    assert lib.pp_gemm_combine_fused(C.byref(a)) == 0                          # not the lean epilogue


def test_a_plan_reports_combine_faults_without_synchronising_its_caller():
    """NetRuntime.check_faults(blocking=False): the fault word travels to pinned host memory behind the queued work and is
    examined by a LATER check -- a pipeline call never ends in a device synchronisation -- while blocking=True (what
    DenoiseLoop.flush_faults and bench.py use where the host waits anyway) reads it on the spot."""
    from powerpaint_amd.engine import SDNet
    from powerpaint_amd.runtime import NetRuntime
    SMALL = dict(block_out_channels=(320, 640), layers_per_block=1,
                 down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))
    net = SDNet("unet", 9, **SMALL)
    net.load_state_dict(net.synthetic_state_dict(seed=3), DEV)
    rt = NetRuntime(net, DEV)
    rt.ensure(2, 16, 16, 77, 9, ("plain",))
    rt.check_faults(blocking=False)
    rt.check_faults(blocking=True)                                   # clean
    word = rt.arena.view(rt.lay["fault"], (1,), torch.int32)
    word.fill_(3)                                                    # what three abandoned, never recovered shares leave
    rt.check_faults(blocking=False)                                  # queued, not examined: must not raise, must not wait
    torch.cuda.synchronize()
    with pytest.raises(L.PPError, match="split-K"):
        rt.check_faults(blocking=False)                              # the copy has landed: the next check sees it
    assert int(word.item()) == 0                                     # ... and re-arms the word
    word.fill_(1)
    with pytest.raises(L.PPError):
        rt.check_faults(blocking=True)
    rt.check_faults(blocking=True)
