"""CPU: host logic of the product (plan compilation, parameter packing, schedulers' coefficient tables, sharding and
the world-size-2 weight broadcast over gloo).  No kernel is launched here."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import schedulers as OS
from oracle import sd_modules as OM
from powerpaint_amd import dist as ppdist
from powerpaint_amd import schedulers as PS
from powerpaint_amd.engine import SDNet, _geglu_interleave
from powerpaint_amd.runtime import NetRuntime

TINY = dict(block_out_channels=(320, 640), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))


def test_state_dict_spec_matches_oracle_architecture():
    for kind, cls, kw, nk in (("unet", OM.UNet2DConditionModel, dict(in_channels=9), {}),
                              ("brushnet", OM.BrushNetModel, {}, dict(conditioning_channels=5)),
                              ("controlnet", OM.ControlNetModel, {}, dict(conditioning_channels=3))):
        with torch.device("meta"):
            ref = {k: tuple(v.shape) for k, v in cls(**kw).state_dict().items()}
        assert SDNet(kind, kw.get("in_channels", 4), **nk).state_dict_spec() == ref


def test_full_size_plans_and_flop_accounting():
    """Launch plans of the real SD-1.5 shapes compile on the host; algorithmic FLOPs match SURVEY.md section 8d
    (minus the cross-attention K/V projections hoisted out of the step)."""
    # launches per forward: 3 LayerNorms per transformer (16 / 16 / 7) are folded into the neighbouring GEMMs
    # ... and FF2 + proj_out are one GEMM (one launch less per transformer)
    ln = (0 if SDNet.fold_ln else 3) - (1 if SDNet.merge_ff2_proj_out else 0)
    # ... and the 1x1 conv_shortcut of the channel-changing resnets rides in conv2 as a K tail (14 / 14 / 2 of them)
    sc = 1 if SDNet._merge_shortcut_env else 0
    exp = {"unet": (803.4, 352 + 16 * ln - 14 * sc), "brushnet": (826.2, 377 + 16 * ln - 14 * sc),
           "controlnet": (283.3 - 16.1, 167 + 7 * ln - 2 * sc)}
    for kind, cin, tot, nk in (("unet", 9, 9, {}), ("brushnet", 4, 9, dict(conditioning_channels=5)),
                               ("controlnet", 4, 4, dict(conditioning_channels=3))):
        net = SDNet(kind, cin, **nk)
        net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
        rt = NetRuntime(net, "cpu")
        rt.ensure(8, 64, 64, 77, tot, ("plain",), cond_hw=(512, 512))
        gf = rt.step_plan.flops / 8 / 1e9
        assert abs(gf - exp[kind][0]) / exp[kind][0] < 0.01, (kind, gf)
        names = [c[2] for c in rt.step_plan.calls]
        n_gn = {"unet": 61, "brushnet": 60, "controlnet": 27}[kind]
        # conv_norm_out's apply rides in the conv_out launch (pp_gn_conv3x3_smallcout) when its statistics come from the
        # producer's epilogue: one GroupNorm launch less in the UNet, and one launch less overall
        from powerpaint_amd.engine import GN_STATS_IN_EPILOGUE
        fused_out = 1 if (kind == "unet" and SDNet.fuse_conv_out and GN_STATS_IN_EPILOGUE) else 0
        # ... and the two norms of every ResnetBlock2D (22 / 22 / 10 blocks) run in the loader of the conv that consumes them
        # (csrc/conv_gn.hip, PPGemmArgs.gn_in_*) wherever pp_conv_gn_preferred says the fused launch is the faster one: NOWHERE
        # since round 6 -- the apply (a launch, or inside the producer's split-K combine) + the plain conv on the same loop
        # without the normalisation wins at every level down to 16x16 (profiles/r06_conv_raw.txt), at 8x8 the tap-major weight
        # stream (round 4)
        from powerpaint_amd.engine import FUSE_GN_CONV, GN_NEXT_IN_COMBINE
        n_cg = 0
        assert sum(1 for a in rt.step_plan.keep if getattr(a, "gn_in_acc", None)) == n_cg
        # ... and a single-tensor norm right behind a split-K launch at the 16x16 / 8x8 levels is applied by that launch's
        # combine (PPGemmArgs.gn_next_*): the non-concatenated resnet norms of the 8x8 level and (round 6: no fused-norm conv
        # any more) of the 16x16 level, the transformer norms of the 16x16 level and of the mid block
        n_next = ({"unet": 24, "brushnet": 24, "controlnet": 15}[kind]
                  if (GN_NEXT_IN_COMBINE and FUSE_GN_CONV and GN_STATS_IN_EPILOGUE) else 0)
        assert sum(1 for a in rt.step_plan.keep if getattr(a, "gn_next_out", None)) == n_next
        # ... and the front end of every C = 320 transformer (norm -> proj_in -> LayerNorm1-folded QKV) is one pp_tfront launch
        n_front = ({"unet": 5, "brushnet": 5, "controlnet": 2}[kind]
                   if (SDNet.fuse_tfront and SDNet.fold_ln and GN_STATS_IN_EPILOGUE) else 0)
        assert names.count("tfront") == n_front
        assert names.count("groupnorm_apply") == n_gn - fused_out - n_cg - n_next - n_front
        if fused_out:
            assert names.count("conv_out") == 1
        # GroupNorm statistics come out of the producing GEMMs' epilogues -- every producer is a GEMM-family launch
        # since conv_in runs on the implicit-GEMM kernel too; one zeroing launch per step instead
        n_stats = names.count("groupnorm_stats")
        if GN_STATS_IN_EPILOGUE:
            assert n_stats == 0 and names.count("zero_u64") == 1 and "conv3x3_direct" not in names
        else:
            assert n_stats == n_gn
        # ... and the C = 320 and (round 4) C = 640 cross-attention sub-blocks (to_q -> attention -> to_out) are one
        # pp_xattn_block launch each, their K / V folded into the projections by pp_xattn_fold in the setup plan
        n_x = names.count("xattn_block")
        per_width = {"unet": 5, "brushnet": 5, "controlnet": 2}[kind]
        assert n_x == ((per_width * (2 if SDNet.fuse_xattn_wide else 1)) if SDNet.fuse_xattn else 0)
        # ... and (round 5) the C = 1280 sub-blocks (16x16 level + mid block: 6 / 6 / 3 of them) run folded too, as TWO GEMMs
        # with per-prompt weights (PP_ACT_SOFTMAX80 + w_batch_stride): one launch less per block than the chain
        n_2g = {"unet": 6, "brushnet": 6, "controlnet": 3}[kind] if (SDNet.fuse_xattn and SDNet.fuse_xattn_2g) else 0
        assert sum(1 for a in rt.step_plan.keep if getattr(a, "act", 0) == 3) == n_2g      # (PP_ACT_SOFTMAX80)
        assert sum(1 for a in rt.step_plan.keep if getattr(a, "w_batch_stride", 0) > 0) == 2 * n_2g
        assert [c[2] for c in rt.setup_plan.calls].count("xattn_fold") == n_x + n_2g
        # ... and (round 5) the feed-forward of every C = 320 transformer (FF1 + GEGLU, FF2 . proj_out) is one pp_ff_fused launch
        n_ff = ({"unet": 5, "brushnet": 5, "controlnet": 2}[kind]
                if (SDNet.fuse_ff and SDNet.fold_ln and SDNet.merge_ff2_proj_out) else 0)
        assert names.count("ff_fused") == n_ff
        # ... and attn1.to_out (+ residual) of the C = 320 transformers rides in front of their fused cross-attention block
        n_pre = ({"unet": 5, "brushnet": 5, "controlnet": 2}[kind]
                 if (SDNet.fuse_xattn and SDNet.fuse_xattn_pre and SDNet.fold_ln) else 0)
        assert sum(1 for c in rt.step_plan.calls if c[2] == "xattn_block" and c[1][19]) == n_pre
        assert len(names) == exp[kind][1] - (n_gn - n_stats) + names.count("zero_u64") - fused_out - 2 * n_x - n_cg - n_next \
            - 2 * n_front - n_ff - n_pre - n_2g
        assert len(rt.setup_plan.calls) >= 15


def test_folded_cross_attention_forms_follow_the_step_geometry():
    """Round 5: which form a cross-attention sub-block takes is decided per step geometry -- the block kernels need 128-row
    (C = 320) / 64-row (C = 640) tiles inside one batch item, the two-GEMM form of C = 1280 whole 64-row GEMM tiles per item;
    where none fits the chain stays and NOTHING is folded for that block in the setup plan."""
    for (H, n_fold, n_2g, n_blk) in ((64, 16, 6, 10),     # headline: 64^2 / 32^2 block kernels, 16^2 + mid two GEMMs
                                     (32, 15, 5, 10),     # 32^2 / 16^2 block kernels (1024 / 256 rows per item), 8^2 = 64 rows:
                                     #                       two GEMMs at the lowest level, the 4x4 mid block keeps the chain
                                     (16, 10, 0, 10)):    # 16^2 / 8^2 block kernels; 4x4 and 2x2 at C = 1280: the chain, no fold
        net = SDNet("unet", 9)
        net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
        rt = NetRuntime(net, "cpu")
        rt.ensure(2, H, H, 77, 9, ("plain",), cond_hw=(8 * H, 8 * H))
        names = [c[2] for c in rt.step_plan.calls]
        folds = [c[2] for c in rt.setup_plan.calls].count("xattn_fold")
        two = sum(1 for a in rt.step_plan.keep if getattr(a, "act", 0) == 3)              # (PP_ACT_SOFTMAX80)
        assert (folds, two, names.count("xattn_block")) == (n_fold, n_2g, n_blk), (H, folds, two, names.count("xattn_block"))
        assert sum(1 for a in rt.step_plan.keep if getattr(a, "w_batch_stride", 0) > 0) == 2 * two


def test_brushnet_wiring_changes_the_unet_plan():
    net = SDNet("unet", 4, **TINY)
    net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
    rt = NetRuntime(net, "cpu")
    rt.ensure(2, 16, 16, 77, 4, ("plain",))
    n_plain = len(rt.step_plan.calls)
    shapes = rt._residual_shapes(2, 16, 16, True)
    assert [len(shapes[k]) for k in ("down", "mid", "up")] == [4, 1, 5]
    wiring = ("brushnet", {k: [0] * len(v) for k, v in shapes.items()})
    rt.ensure(2, 16, 16, 77, 4, wiring)
    assert len(rt.step_plan.calls) == n_plain + 1          # the second conv_in launch (skip captured before the add)
    res2 = [a for a in rt.step_plan.keep if a.res2]
    assert len(res2) == 4 + 1 + 5                          # every residual rides a GEMM epilogue (conv_in's too)
    slots = {s.ptr for g in rt.lay["slots"].values() for s in g}
    assert {a.res2 for a in res2} <= slots


def test_guess_mode_output_layout_of_the_side_networks():
    """pad_uncond (the pipelines' guess mode): residual tensors with twice the batch, the zero convs write the second
    half, one extra zeroing launch covers all of them; the UNet refuses the flag."""
    for kind, tot, nk in (("brushnet", 9, dict(conditioning_channels=5)), ("controlnet", 4, dict(conditioning_channels=3))):
        kw = {k: v for k, v in TINY.items() if not (kind == "controlnet" and k == "up_block_types")}
        net = SDNet(kind, 4, **kw, **nk)
        net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
        rt = NetRuntime(net, "cpu")
        rt.ensure(2, 16, 16, 77, tot, ("plain",), cond_hw=(128, 128))
        plain = [c[2] for c in rt.step_plan.calls]
        rt.ensure(2, 16, 16, 77, tot, ("plain",), cond_hw=(128, 128), pad_uncond=True)
        names = [c[2] for c in rt.step_plan.calls]
        assert len(names) == len(plain) + 1 and names.count("zero_u64") == plain.count("zero_u64") + 1
        outs = rt.outputs["down"] + [rt.outputs["mid"]] + rt.outputs.get("up", [])
        zc = [a for a in rt.step_plan.keep if a.scale is not None and any(a is c[1][0]._obj for c in rt.step_plan.calls
                                                                         if c[2] == "zero_conv")]
        assert len(zc) == len(outs) and all(o.B == 4 for o in outs)
        for a, o in zip(zc, outs):
            assert a.M == 2 * o.H * o.W and a.out == o.ptr + a.M * o.C * 2          # second half of the padded tensor
        zi = names.index("zero_u64", 1)
        fn, args, _ = rt.step_plan.calls[zi]
        end = outs[-1].ptr + outs[-1].rows * outs[-1].C * 2
        assert args[0] == outs[0].ptr and args[1] * 8 == end - outs[0].ptr
        assert zi < names.index("zero_conv")
    net = SDNet("unet", 4, **TINY)
    net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
    with pytest.raises(Exception):
        NetRuntime(net, "cpu").ensure(2, 16, 16, 77, 4, ("plain",), pad_uncond=True)


def test_conv_in_is_packed_for_the_implicit_gemm_over_a_padded_input():
    """conv_in (4 / 9 input channels) runs on the implicit-GEMM kernel: weights [Cout][tap][64] with zeros in the pad
    channels, the network input buffer 64 channels wide, no direct-conv launch in the plan."""
    torch.manual_seed(0)
    o = OM.UNet2DConditionModel(in_channels=9, **TINY)
    net = SDNet("unet", 9, **TINY).load_state_dict(o.state_dict(), "cpu")
    assert (net.cin0, net.cin_pad) == (9, 64)
    w = net.params.tensor("conv_in.weight")
    assert w.shape == (320, 9 * 64) and w.dtype == torch.bfloat16
    w = w.view(320, 3, 3, 64)
    assert torch.equal(w[..., :9], o.conv_in.weight.permute(0, 2, 3, 1).to(torch.bfloat16))
    assert torch.count_nonzero(w[..., 9:]) == 0
    rt = NetRuntime(net, "cpu")
    rt.ensure(2, 16, 16, 77, 9, ("plain",))
    assert rt.lay["x_in"].C == 64
    first = next(a for a in rt.step_plan.keep if a.x_mode == 1)
    assert (first.c1, first.N, first.K, first.M) == (64, 320, 9 * 64, 2 * 16 * 16) and first.x1 == rt.lay["x_in"].ptr
    assert first.gn_acc[0]                                    # ... and its epilogue carries resnets.0.norm1's statistics
    with pytest.raises(Exception):
        rt.ensure(2, 16, 16, 77, 4, ("plain",))               # a 4-channel input for the 9-channel network


def test_geglu_interleave_is_a_row_permutation():
    w = torch.arange(16 * 3, dtype=torch.float32).reshape(16, 3)
    p = _geglu_interleave(w)
    assert torch.equal(p[0], w[0]) and torch.equal(p[1], w[1]) and torch.equal(p[2], w[8]) and torch.equal(p[3], w[9])
    assert torch.equal(p[4], w[2]) and torch.equal(p[6], w[10])
    assert sorted(p[:, 0].tolist()) == sorted(w[:, 0].tolist())


def test_param_pack_roundtrip_and_single_buffer():
    torch.manual_seed(0)
    o = OM.UNet2DConditionModel(in_channels=9, **TINY)
    net = SDNet("unet", 9, **TINY).load_state_dict(o.state_dict(), "cpu")
    pk = net.params
    w = o.down_blocks[0].resnets[0].conv1.weight
    got = pk.tensor("down_blocks.0.resnets.0.conv1.weight")
    assert got.shape == (320, 9 * 320) and got.dtype == torch.bfloat16
    assert torch.equal(got, w.permute(0, 2, 3, 1).reshape(320, -1).to(torch.bfloat16))
    qkv = pk.tensor("down_blocks.0.attentions.0.transformer_blocks.0.attn1.qkv.weight")
    a = o.down_blocks[0].attentions[0].transformer_blocks[0].attn1
    assert torch.equal(qkv[640:], a.to_v.weight.to(torch.bfloat16))
    assert pk.tensor("temb_all.weight").shape == (net.temb_total, 1280)
    lo, hi = min(pk.ptr.values()), max(pk.ptr.values())
    assert pk.buf.data_ptr() <= lo and hi < pk.buf.data_ptr() + pk.buf.numel()      # everything in ONE buffer


@pytest.mark.parametrize("N", [10, 50])
def test_product_scheduler_tables_reproduce_the_oracle(N):
    """Emulate pp_cfg_sched_step on the host with the product's coefficient tables; compare with the oracle classes."""
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 8, 8, generator=g)
    eps = [torch.randn(1, 4, 8, 8, generator=g) for _ in range(N)]
    for P, O in ((PS.DDIMScheduler, OS.DDIMScheduler), (PS.DPMSolverMultistepScheduler, OS.DPMSolverMultistepScheduler)):
        p, o = P(), O()
        p.set_timesteps(N)
        o.set_timesteps(N)
        assert torch.equal(p.timesteps, o.timesteps)
        x, ref, m = x0.clone(), x0.clone(), torch.zeros_like(x0)
        for i, t in enumerate(o.timesteps):
            c = p._coef[i]
            if p.kind == 0:
                x = c[2] * ((x - c[0] * eps[i]) / c[1]) + c[3] * eps[i]
            else:
                pred = (x - c[0] * eps[i]) / c[1]
                x = c[2] * x - c[3] * pred - c[4] * (c[5] * (pred - m))
                m = pred
            ref = o.step(eps[i], t, ref)[0]
        assert torch.allclose(x, ref, rtol=1e-5, atol=1e-5), (P.__name__, (x - ref).abs().max())


def test_scheduler_duck_type_surface():
    s = PS.DDIMScheduler()
    s.set_timesteps(50)
    assert s.order == 1 and s.init_noise_sigma == 1.0 and s.config.steps_offset == 1
    assert s.timesteps[0] == 981 and len(s.timesteps) == 50
    x = torch.randn(2, 4, 8, 8)
    assert s.scale_model_input(x, s.timesteps[0]) is x
    import inspect
    assert {"eta", "generator"} <= set(inspect.signature(s.step).parameters)
    with pytest.raises(Exception):
        s.step(x, 981, x)           # CPU tensors: the step only exists as a HIP kernel -> loud failure, no fallback


def test_shard_ranges_cover_the_batch_once():
    for gb, world in ((32, 8), (16, 8), (5, 2), (4, 1), (3, 4)):
        seen = []
        for r in range(world):
            seen += list(ppdist.shard_range(gb, r, world))
        assert seen == list(range(gb))
    g1, g2 = ppdist.image_generator(7), ppdist.image_generator(7)
    assert torch.equal(torch.randn(4, generator=g1), torch.randn(4, generator=g2))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r, w, _ = ppdist.init_from_env("gloo")
    net = SDNet("unet", 9, **TINY)
    if r == 0:
        net.load_state_dict(net.synthetic_state_dict(seed=3), "cpu")
    else:
        net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
        assert int(net.params.buf.count_nonzero()) == 0
    ppdist.broadcast_params([net.params.buf], src=0)                   # the one collective of the path
    chk = float(net.params.buf.view(torch.int16).double().sum())
    # the other networks of a pipeline: packed VAE buffer + the CLIP tower (an nn.Module, flattened per dtype)
    from powerpaint_amd.models import AutoencoderKL, CLIPTextModel
    vae = AutoencoderKL(device="cpu", block_out_channels=(64, 64, 64, 64), layers_per_block=1)
    vae.load_state_dict(vae.net.synthetic_state_dict(seed=5 + r))        # different values per rank before the broadcast
    torch.manual_seed(100 + r)
    enc = CLIPTextModel(device="cpu", vocab_size=64, num_hidden_layers=1)
    ppdist.broadcast_models([vae, enc], src=0)
    chk += float(vae.param_buffer().view(torch.int16).double().sum())
    chk += float(sum(p.double().sum() for p in enc.parameters()))
    # image shards: rank-count-invariant inputs, gather back in global order
    idx = list(ppdist.shard_range(4, r, w))
    local = torch.stack([torch.randn(4, 2, 2, generator=ppdist.image_generator(i)) for i in idx])
    allv = ppdist.gather_latents(local, 4)
    t = ppdist.max_over_ranks(float(r + 1), "cpu")
    ppdist.barrier()
    q.put((r, chk, allv.tolist(), t))       # plain lists: a tensor's shared-memory handle can die with this process
    torch.distributed.destroy_process_group()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=240) for _ in procs], key=lambda x: x[0])
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()                     # (exact process objects we started)
                p.join(10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return res


def test_two_rank_broadcast_and_sharding_gloo():
    try:
        res = _run_two_ranks()
    except Exception:                        # a rendezvous port can be grabbed between probing and binding: one retry
        res = _run_two_ranks()
    assert res[0][1] == res[1][1] != 0.0                                 # identical parameter bytes on both ranks
    ref = torch.stack([torch.randn(4, 2, 2, generator=ppdist.image_generator(i)) for i in range(4)])
    assert res[0][2] == ref.tolist() and res[1][2] == ref.tolist()      # global order, independent of rank count
    assert res[0][3] == res[1][3] == 2.0


def _bench_worker(rank, world, port, q):
    """bench.py's own start-up path on a CPU arena: rank 0 materialises weights, rank 1 takes the `meta=True` /
    `materialize=False` branch and receives them in `broadcast_params`."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    r, w, _ = ppdist.init_from_env("gloo", timeout_s=120.0)        # (the watchdog + first all-reduce path of bench.py)
    assert ppdist.gather_strings(f"rank {r}") == ["rank 0", "rank 1"]
    out = []
    for cfg in ("v1", "v2", "controlnet"):
        pipe, nets, bcast_s = bench.build_pipeline(cfg, "cpu", r, w, net_kw=TINY)
        assert len(nets) == (1 if cfg == "v1" else 2) and bcast_s >= 0.0
        if r != 0:
            assert all(int(m.param_buffer().count_nonzero()) > 0 for m in nets)       # received, not generated
        out.append([float(m.param_buffer().view(torch.int16).double().sum()) for m in nets])
        kw = bench.synthetic_inputs(cfg, "cpu", r, 2, 8)       # rank-local inputs keyed by the global image index
        out.append(float(kw["latents"].double().sum()))
    ppdist.barrier()
    q.put((r, out))
    torch.distributed.destroy_process_group()


def test_bench_startup_path_two_ranks_gloo():
    """The N > 1 branch of bench.py (never exercised on hardware with one GPU per lease): both ranks end up with the
    same parameter bytes for every network of every config, and their synthetic inputs are the two halves of the
    global batch (seeded by global image index)."""
    ctx = mp.get_context("spawn")

    def run():
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
        finally:
            for p in procs:
                p.join(60)
                if p.is_alive():
                    p.kill()
                    p.join(10)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        return res

    try:
        res = run()
    except Exception:
        res = run()
    a, b = res[0][1], res[1][1]
    for i in (0, 2, 4):
        assert a[i] == b[i] and all(v != 0.0 for v in a[i])              # identical packed buffers on both ranks
    import bench
    for i, cfg in ((1, "v1"), (3, "v2"), (5, "controlnet")):
        both = bench.synthetic_inputs(cfg, "cpu", 0, 4, 8)["latents"]    # world = 1 with the whole batch
        assert a[i] == float(both[:2].double().sum()) and b[i] == float(both[2:].double().sum())


def test_distributed_watchdog_names_the_stuck_stage_and_exits():
    """bench.py --gpus N must fail fast, not hang: a rank that waits for a peer that never arrives leaves within the
    time-out with a message that names the stage."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, time; sys.path.insert(0, %r)\n"
            "from powerpaint_amd import dist as d\n"
            "with d.Watchdog(0.5, 'first all-reduce over nccl'):\n"
            "    time.sleep(30)\n" % root)
    env = dict(os.environ, RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    t0 = __import__("time").time()
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and __import__("time").time() - t0 < 60
    assert "rank 1/2" in p.stderr and "first all-reduce over nccl" in p.stderr and "MASTER_ADDR=127.0.0.1" in p.stderr
    with ppdist.Watchdog(30.0, "fast block"):      # a block that finishes cancels the timer
        pass
    with ppdist.Watchdog(0.0, "disabled"):
        pass


def test_encode_prompt_matches_reference_method():
    """Row a21 (blend) / a18 (CFG batch order): `PipelineBase._encode_prompt` against the reference's own
    `StableDiffusionInpaintPipeline._encode_prompt` (tests/golden/ref_encode_prompt.pt, lifted out of
    pipeline_PowerPaint.py:317-518 by AST).  Host logic over a torch text encoder: runs on CPU, compared exactly."""
    import json
    transformers = pytest.importorskip("transformers")
    from powerpaint_amd.pipelines._base import PipelineBase
    here = os.path.dirname(os.path.abspath(__file__))
    G = torch.load(os.path.join(here, "golden", "ref_encode_prompt.pt"), weights_only=False)
    with open(os.path.join(here, "golden", "ref_task_tokens.json")) as f:
        T = json.load(f)
    tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(T["vocab"])},
                                     merges=[tuple(m) for m in T["merges"]], model_max_length=77)
    n = G["vocab_size"]
    enc = transformers.CLIPTextModel(transformers.CLIPTextConfig(vocab_size=n, bos_token_id=n - 2, eos_token_id=n - 1,
                                                                 pad_token_id=n - 1, **G["cfg"])).eval()
    enc.load_state_dict(G["state_dict"])
    pipe = PipelineBase()
    pipe.register_modules(tokenizer=tok, text_encoder=enc)
    dev = torch.device("cpu")
    with torch.no_grad():
        for c, want in zip(G["cases"], G["outs"]):
            got = pipe._encode_prompt(c["promptA"], c["promptB"], c["t"], dev, c["n"], c["cfg"],
                                      negative_promptA=c["nA"], negative_promptB=c["nB"], t_nag=c["tn"])
            assert got.shape == want.shape and torch.equal(got, want), c
        got = pipe._encode_prompt(None, None, 0.5, dev, 2, True, t_nag=0.5, prompt_embeds=G["pe"],
                                  negative_prompt_embeds=G["ne"])
        assert torch.equal(got, G["outs"][-1])
        # the BrushNet pipeline's plain promptU encoder (pipeline_PowerPaint_Brushnet_CA.py:442-629)
        from powerpaint_amd.pipelines import StableDiffusionPowerPaintBrushNetPipeline
        v2 = StableDiffusionPowerPaintBrushNetPipeline(text_encoder=enc, tokenizer=tok)
        for c, want in zip(G["u_cases"], G["outs_u"]):
            got = v2.encode_prompt(c["prompt"], dev, c["n"], True, c["neg"])
            assert got.shape == want.shape and torch.equal(got, want), c
        got = v2.encode_prompt(None, dev, 2, True, None, prompt_embeds=G["pe"], negative_prompt_embeds=G["ne"])
        assert torch.equal(got, G["outs_u"][-1])


def test_check_inputs_error_behaviour_matches_reference():
    """Same inputs rejected, same exception type, as the reference's `check_inputs` (pipeline_PowerPaint.py:553-602;
    tests/golden/ref_check_inputs.json holds its verdict on 1728 argument combinations)."""
    import json
    from powerpaint_amd.pipelines import StableDiffusionInpaintPipeline
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_check_inputs.json")) as f:
        rows = json.load(f)
    pipe = StableDiffusionInpaintPipeline()
    emb = {"E": torch.zeros(1, 77, 8), "E2": torch.zeros(2, 77, 8), None: None}
    n_ok = 0
    for p, h, w, s, cb, ng, pe, ne, want in rows:
        try:
            pipe.check_inputs(p, h, w, s, cb, ng, emb[pe], emb[ne])
            got = "ok"
        except Exception as e:
            got = type(e).__name__
        assert got == want, (p, h, w, s, cb, ng, pe, ne, got, want)
        n_ok += got == "ok"
    assert 0 < n_ok < len(rows)


def test_retrieve_timesteps_follows_the_reference():
    """pipeline_PowerPaint_Brushnet_CA.py:87-128: a custom list reaches schedulers whose `set_timesteps` names
    `timesteps`; the four fused schedulers (like diffusers 0.27's) do not -> the reference's ValueError."""
    from powerpaint_amd import schedulers as PS
    from powerpaint_amd.pipelines.pipeline_PowerPaint_Brushnet_CA import retrieve_timesteps
    for cls in (PS.DDIMScheduler, PS.DPMSolverMultistepScheduler, PS.PNDMScheduler, PS.UniPCMultistepScheduler):
        s = cls()
        ts, n = retrieve_timesteps(s, 5, None)
        assert n == 5 and len(ts) == len(s.timesteps)
        with pytest.raises(ValueError, match="does not support custom"):
            retrieve_timesteps(s, None, None, timesteps=[999, 500, 1])

    class Custom:
        def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
            self.timesteps = torch.tensor(timesteps)

    ts, n = retrieve_timesteps(Custom(), None, None, timesteps=[9, 5, 1])
    assert n == 3 and ts.tolist() == [9, 5, 1]


def test_model_to_refuses_what_it_cannot_honour_and_wiring_is_dtype_safe():
    """`.to(other dtype)` on a packed network used to be silently ignored, and the zero-copy residual
    hand-off reinterpreted a side network's bits when the two networks stored different 16-bit formats."""
    from powerpaint_amd import _lib as L
    from powerpaint_amd import models as PM
    u = PM.UNet2DConditionModel(in_channels=4, device="cpu", dtype=torch.float16, **TINY)
    assert u.to("cpu") is u and u.to(torch.float16) is u and u.to(dtype=torch.float16, device="cpu") is u
    assert u.to(torch.zeros(1, dtype=torch.float16)) is u
    with pytest.raises(L.PPError, match="packed as"):
        u.to(torch.bfloat16)
    with pytest.raises(L.PPError, match="packed as"):
        u.to(dtype=torch.float32)
    with pytest.raises(L.PPError, match="live on"):
        u.to("cuda")
    # wiring: a residual tensor stored in another format is never taken by pointer (it goes through the converting copy)
    t16 = torch.zeros(1, 4, 2, 2, dtype=torch.float16)
    tbf = torch.zeros(1, 4, 2, 2, dtype=torch.bfloat16)
    t16._pp_nhwc_ptr, tbf._pp_nhwc_ptr = 1234, 5678
    kind, ptrs = u._wiring([t16, tbf], t16, [tbf], None, None)
    assert kind == "brushnet" and ptrs == {"down": [1234, 0], "mid": [1234], "up": [0]}


def test_runtime_scale_schedule_representation_is_stable():
    """guess mode hands `ensure` a LIST of per-residual scales; the stored form is a tuple, so an unchanged schedule must
    compare equal (it used to re-patch every launch record and drop the captured graph on every bind)."""
    from powerpaint_amd.runtime import NetRuntime
    calls = []

    class RT(NetRuntime):
        def __init__(self):
            self.key, self._scale, self.gemm_tile, self.gemm_splitk = None, 1.0, 0, 0

        def _patch_scale(self, scale):
            calls.append(scale)
            self._scale = tuple(float(v) for v in scale) if isinstance(scale, (list, tuple)) else scale

    rt = RT()
    rt.key = (2, 8, 8, 77, 9, ("plain",), None, 0, 0, False, False)
    rt.ensure(2, 8, 8, 77, 9, scale=[0.1, 0.5, 1.0])
    rt.ensure(2, 8, 8, 77, 9, scale=[0.1, 0.5, 1.0])
    rt.ensure(2, 8, 8, 77, 9, scale=(0.1, 0.5, 1.0))
    assert len(calls) == 1
    rt.ensure(2, 8, 8, 77, 9, scale=0.5)
    rt.ensure(2, 8, 8, 77, 9, scale=0.5)
    assert len(calls) == 2


def test_time_embedding_chain_is_where_the_loop_expects_it():
    """DenoiseLoop replaces the four t-only launches of a network's step plan (sinusoid, time_embedding.linear_1/2, all
    time_emb_proj rows) by one table-row copy: the plan compiler must keep emitting them as four consecutive calls whose
    last one writes `temb_total` floats (else the loop silently falls back to running the chain every step)."""
    import types
    from powerpaint_amd.pipelines._loop import DenoiseLoop
    for kind, cin, nk in (("unet", 9, {}), ("brushnet", 4, dict(conditioning_channels=5)),
                          ("controlnet", 4, dict(conditioning_channels=3))):
        net = SDNet(kind, cin, **TINY, **nk)
        net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
        rt = NetRuntime(net, "cpu")
        rt.ensure(2, 16, 16, 77, 9 if kind != "controlnet" else 4, ("plain",), cond_hw=(128, 128))
        info = DenoiseLoop._temb_split(types.SimpleNamespace(), rt, torch.zeros(7))
        assert info is not None, kind
        names = [rt.step_plan.calls[i][2] for i in info["idx"]]
        assert names == ["timestep_embedding", "linear_skinny", "linear_skinny", "linear_skinny"], (kind, names)
        assert info["total"] == net.temb_total and tuple(info["table"].shape) == (7, net.temb_total)
        assert sum(c[2] in ("timestep_embedding", "linear_skinny") for c in rt.step_plan.calls) == 4


def test_apply_inside_the_combine_never_aliases_what_the_producer_still_reads():
    """PPGemmArgs.gn_next_out is written by the producer's split-K combine WHILE that kernel reads the split-K slabs and the
    residual operands -- scratch the launch-plan compiler has released by the time the consuming norm is compiled.  The
    engine therefore places the normalised tensor above everything that was live when the producer was recorded
    (Builder._apply_in_producer_combine); this walks the full-size plans of all three networks and checks that no such
    output overlaps an operand of its own launch, and that it is consumed (as a conv / linear input) later in the plan."""
    from powerpaint_amd.engine import GN_NEXT_IN_COMBINE
    if not GN_NEXT_IN_COMBINE:
        pytest.skip("PP_GN_NEXT=0")

    def span(ptr, nbytes):
        return (int(ptr), int(ptr) + int(nbytes)) if ptr else None

    def overlap(a, b):
        return a is not None and b is not None and a[0] < b[1] and b[0] < a[1]

    for kind, cin, tot, nk in (("unet", 9, 9, {}), ("brushnet", 4, 9, dict(conditioning_channels=5)),
                               ("controlnet", 4, 4, dict(conditioning_channels=3))):
        net = SDNet(kind, cin, **nk)
        net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
        rt = NetRuntime(net, "cpu")
        rt.ensure(8, 64, 64, 77, tot, ("plain",), cond_hw=(512, 512))
        keep = list(rt.step_plan.keep)
        n = 0
        for i, a in enumerate(keep):
            if not a.gn_next_out:
                continue
            n += 1
            out = span(a.gn_next_out, a.M * a.N * 2)
            from powerpaint_amd import _lib as L_
            sk = L_.lib().pp_gemm_workspace_bytes(C.byref(a))
            assert sk > 0, "gn_next_out on a launch without a split-K combine"
            hw_in = a.batch * a.hin * a.win if a.x_mode else a.M
            operands = {"workspace": span(a.workspace, sk), "out": span(a.out, a.M * a.ldo * 2),
                        "res1": span(a.res1, a.M * a.ldres1 * 2), "res2": span(a.res2, a.M * a.ldres2 * 2),
                        "x1": span(a.x1, hw_in * a.c1 * 2), "x2": span(a.x2, hw_in * a.c2 * 2),
                        "x3": span(a.x3, a.M * a.c3 * 2), "x4": span(a.x4, a.M * a.c4 * 2)}
            for name, sp in operands.items():
                assert not overlap(out, sp), (kind, i, name, out, sp)
            # ... and somebody reads it: the next launches' x1
            assert any(b.x1 == a.gn_next_out for b in keep[i + 1:i + 4]), (kind, i, "normalised tensor is never consumed")
        assert n == {"unet": 24, "brushnet": 24, "controlnet": 15}[kind]


def test_twin_prefix_plans_at_the_headline_shapes():
    """Round 5: with `twin` (the loop vouches for a CFG pair built from one latents tensor) the launches in front of the
    first cross-attention run on half the rows -- conv_in (which stores its rows twice: out_dup_rows, and feeds the
    up-block concat norm's statistics for both halves), down_blocks.0.resnets.0, the first transformer's front end,
    self-attention and to_out -- the fused cross-attention block reads the half batch with wrap addressing and the
    transformer's last GEMM its half-batch residual.  Same launch count, fewer executed FLOPs; BrushNet adds inside the
    down path switch it off."""
    from powerpaint_amd.engine import TWIN_PREFIX
    if not (TWIN_PREFIX and SDNet.fuse_xattn):
        pytest.skip("lab switches")
    for kind, cin, tot, nk in (("unet", 9, 9, {}), ("brushnet", 4, 9, dict(conditioning_channels=5)),
                               ("controlnet", 4, 4, dict(conditioning_channels=3))):
        net = SDNet(kind, cin, **nk)
        net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
        rt = NetRuntime(net, "cpu")
        rt.ensure(8, 64, 64, 77, tot, ("plain",), cond_hw=(512, 512))
        f0, names0 = rt.step_plan.flops, [c[2] for c in rt.step_plan.calls]
        rt.ensure(8, 64, 64, 77, tot, ("plain",), cond_hw=(512, 512), twin=True)
        f1, names1 = rt.step_plan.flops, [c[2] for c in rt.step_plan.calls]
        assert names0 == names1
        # conv_in (real channels only) + 2 resnet convs + proj_in + QKV + self-attention + to_out, on 4 x 4096 rows
        rows, hw, C = 4 * 4096, 4096, 320
        # (attn1.to_out rides in the full-batch cross-attention launch when SDNet.fuse_xattn_pre: not halved then)
        to_out = 0 if (SDNet.fuse_xattn_pre and SDNet.fold_ln) else C
        saved = 2.0 * rows * C * (9 * tot + 2 * 9 * C + C + 3 * C + to_out) + 4.0 * 4 * 8 * hw * hw * 40
        assert abs((f0 - f1) - saved) / saved < 1e-6, (kind, f0 - f1, saved)
        dup = [a for a in rt.step_plan.keep if a.out_dup_rows]
        assert len(dup) == 1 and (dup[0].M, dup[0].N, dup[0].K, dup[0].out_dup_rows) == (rows, C, 576, rows)
        # the skip tensor's second consumer (the last up-block resnet's concat norm) gets both halves' sums
        assert (dup[0].gn_dup_batch, dup[0].gn_dup_mask) == ((4, 2) if kind != "controlnet" else (0, 0))
        wrap = [a for a in rt.step_plan.keep if a.res1_wrap_rows]     # (the second GEMM of the feed-forward, fused or not)
        assert len(wrap) == 1 and (wrap[0].M, wrap[0].K, wrap[0].res1_wrap_rows) == (2 * rows, 1600, rows)
        i = names1.index("xattn_block")
        args = rt.step_plan.calls[i][1]
        assert args[15] == 2 * rows and args[18] == rows                  # M, src_wrap_rows
        assert all(rt.step_plan.calls[j][1][18] == 0 for j, n in enumerate(names1) if n == "xattn_block" and j != i)
    net = SDNet("unet", 4)
    net.load_state_dict(net.synthetic_state_dict(meta=True), "cpu", materialize=False)
    rt = NetRuntime(net, "cpu")
    shapes = rt._residual_shapes(8, 64, 64, True)
    rt.ensure(8, 64, 64, 77, 4, ("brushnet", {k: [0] * len(v) for k, v in shapes.items()}), twin=True)
    assert not [a for a in rt.step_plan.keep if a.out_dup_rows or a.res1_wrap_rows]
