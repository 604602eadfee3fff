"""Per-network runtime: static arena, compiled launch plans, step-invariant setup, optional hipGraph replay."""
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .engine import Act, Arena, Builder, Plan, SDNet, _align

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class NetRuntime:
    """Owns the device state of one SDNet for one (batch, latent size, wiring)."""

    def __init__(self, net: SDNet, device):
        self.net = net
        self.device = torch.device(device)
        self.key = None
        self.arena: Optional[Arena] = None
        self.step_plan: Optional[Plan] = None
        self.setup_plan: Optional[Plan] = None
        self.outputs: Dict[str, object] = {}
        self._ctx_id = None
        self._cond_id = None
        self._ctx_keep = None
        self._cond_keep = None
        self.graph = None
        self.lib = L.lib()
        self.gemm_tile = 0
        self.gemm_splitk = 0

    # ------------------------------------------------------------------ build
    def _build(self, arena: Arena, B, H, W, nctx, cin_total, wiring, cond_hw, scale, pad_uncond=False, twin=False):
        net = self.net
        pb_setup = Builder(arena, dtype=net.dtype)
        pb_setup.gemm_tile, pb_setup.gemm_splitk = self.gemm_tile, self.gemm_splitk
        lay = {}
        lay["t_dev"] = arena.alloc(256)
        # int64 accumulators of the GroupNorm statistics the GEMM epilogues produce (<= 96 norms per network); zeroed
        # by the first launch of every step
        gn_cap = 96 * B * 32 * 2 * 8
        lay["gn_acc"] = arena.alloc(gn_cap)
        lay["fault"] = arena.alloc(256)          # placement violations seen by the in-kernel split-K combines (check_faults)
        # network input, NHWC; the channels are zero-padded to one 64-deep K chunk (conv_in runs on the implicit-GEMM
        # kernel).  The arena is zero-filled and nothing else ever writes the pad channels.
        if cin_total != net.cin0:
            raise L.PPError(f"{net.kind} takes {net.cin0} input channels, got {cin_total}")
        lay["x_in"] = Act(arena.alloc(B * H * W * net.cin_pad * 2), B, H, W, net.cin_pad)
        lay["ehs"] = arena.alloc(B * nctx * net.ctx_dim * 2)
        cond = None
        if net.kind == "controlnet":
            ch, cw = cond_hw
            cond = Act(arena.alloc(B * ch * cw * net.conditioning_channels * 2), B, ch, cw, net.conditioning_channels)
            lay["cond"] = cond
        # slots for foreign residual tensors (copied in at forward time)
        kind, n_down, n_up = wiring[0], 0, 0
        slots: Dict[str, List[Act]] = {}
        if kind in ("brushnet", "controlnet") and net.kind == "unet":
            shapes = self._residual_shapes(B, H, W, with_up=(kind == "brushnet"))
            slots = {k: [Act(arena.alloc(b * h * w * c * 2), b, h, w, c) for (b, c, h, w) in v]
                     for k, v in shapes.items()}
        lay["slots"] = slots
        net.build_setup(pb_setup, B, nctx, lay["ehs"], cond, hw0=(H, W))
        pb = Builder(arena, dtype=net.dtype)
        pb.gemm_tile, pb.gemm_splitk = self.gemm_tile, self.gemm_splitk
        pb.gn_acc_base, pb.gn_acc_cap = lay["gn_acc"], gn_cap
        pb.fault_ptr = lay["fault"]
        kw = {}
        if net.kind == "unet" and kind == "brushnet":
            ptrs = wiring[1]
            dn = [p if p else s.ptr for p, s in zip(ptrs["down"], slots["down"])]
            up = [p if p else s.ptr for p, s in zip(ptrs["up"], slots["up"])]
            md = ptrs["mid"][0] if ptrs["mid"][0] else slots["mid"][0].ptr
            kw = dict(add_down=dn, add_mid=md, add_up=up)
        elif net.kind == "unet" and kind == "controlnet":
            ptrs = wiring[1]
            dn = [p if p else s.ptr for p, s in zip(ptrs["down"], slots["down"])]
            md = ptrs["mid"][0] if ptrs["mid"][0] else slots["mid"][0].ptr
            kw = dict(ctrl_down=dn, ctrl_mid=md)
        if pad_uncond:
            if net.kind == "unet":
                raise L.PPError("pad_uncond is an output layout of the side networks (BrushNet / ControlNet)")
            kw["pad_uncond"] = True
        outs = net.build_step(pb, lay["x_in"], lay["t_dev"], scale=1.0, twin=twin, **kw)
        if pb.gn_acc_used:
            pb.plan.calls.insert(0, (L.lib().pp_zero_u64, (lay["gn_acc"], pb.gn_acc_used // 8), "zero_u64"))
        return lay, pb_setup.plan, pb.plan, outs

    def _residual_shapes(self, B, H, W, with_up: bool):
        """(batch, C, h, w) of the residual tensors the UNet accepts, in reference order."""
        net = self.net
        boc = net.boc
        down = [(B, boc[0], H, W)]
        h, w = H, W
        for i, c in enumerate(boc):
            for _ in range(net.L):
                down.append((B, c, h, w))
            if i != len(boc) - 1:
                h, w = h // 2, w // 2
                down.append((B, c, h, w))
        out = {"down": down, "mid": [(B, boc[-1], h, w)]}
        if with_up:
            up = []
            for i, c in enumerate(reversed(boc)):
                for _ in range(net.L + 1):
                    up.append((B, c, h, w))
                if i != len(boc) - 1:
                    h, w = h * 2, w * 2
                    up.append((B, c, h, w))
            out["up"] = up
        return out

    def ensure(self, B: int, H: int, W: int, nctx: int, cin_total: int, wiring=("plain",), cond_hw=None,
               scale: float = 1.0, pad_uncond: bool = False, twin: bool = False):
        """twin: the caller vouches that the second half of every network input (x_in, ControlNet conditioning) equals the
        first -- a CFG pair built from one tensor; the step plan then runs the prompt-independent prefix on one half
        (SDNet.build_step)."""
        def freeze(w):
            if len(w) == 1:
                return w
            return (w[0], tuple((k, tuple(v)) for k, v in sorted(w[1].items())))

        key = (B, H, W, nctx, cin_total, freeze(wiring), cond_hw, self.gemm_tile, self.gemm_splitk, bool(pad_uncond),
               bool(twin))
        if isinstance(scale, (list, tuple)):
            scale = tuple(float(v) for v in scale)       # (one representation: a list never equals the stored tuple)
        if key == self.key:
            if scale != self._scale:
                self._patch_scale(scale)
            return
        if H % (2 ** (len(self.net.boc) - 1)) or W % (2 ** (len(self.net.boc) - 1)):
            raise L.PPError(f"latent size {H}x{W} must be divisible by {2 ** (len(self.net.boc) - 1)}")
        # (the one synchronising probe of the library runs here, at plan-build time and never under a stream capture: are
        #  workgroups placed on the XCDs round-robin by linear id?  Plans combine split-K in-kernel only if so.)
        self.xcd_placement_ok = bool(self.lib.pp_xcd_placement_ok())
        dry = Arena()
        self._build(dry, B, H, W, nctx, cin_total, wiring, cond_hw, scale, pad_uncond, twin)
        self.arena = Arena(_align(dry.peak, 4096), self.device)
        self.lay, self.setup_plan, self.step_plan, self.outputs = self._build(
            self.arena, B, H, W, nctx, cin_total, wiring, cond_hw, scale, pad_uncond, twin)
        self.twin = bool(twin)
        self.key = key
        self._scale = 1.0
        if scale != 1.0:
            self._patch_scale(scale)
        self._ctx_id = None
        self._cond_id = None
        self._ctx_keep = self._cond_keep = None
        self.graph = None
        self.B, self.H, self.W, self.nctx = B, H, W, nctx

    def _patch_scale(self, scale):
        """conditioning_scale is baked into the zero-conv GEMM launches; patch it in place (float or per-output list)."""
        zc = [args[0]._obj for fn, args, name in self.step_plan.calls if name == "zero_conv"]
        vals = list(scale) if isinstance(scale, (list, tuple)) else [scale] * len(zc)
        for a, v in zip(zc, vals):
            a.scale = float(v)
        self._scale = tuple(float(v) for v in scale) if isinstance(scale, (list, tuple)) else scale
        self.graph = None

    # ------------------------------------------------------------------ inputs
    def set_timestep(self, t):
        tv = self.arena.view(self.lay["t_dev"], (1,), torch.float32)
        if torch.is_tensor(t):
            tv.copy_(t.reshape(-1)[:1].to(torch.float32), non_blocking=True)
        else:
            tv.fill_(float(t))

    def set_context(self, ehs: torch.Tensor, force: bool = False):
        """encoder_hidden_states [B, nctx, ctx_dim]; recomputes the hoisted cross-attention K/V^T only on change."""
        ident = (ehs.data_ptr(), ehs._version, tuple(ehs.shape), getattr(self.net.params, "version", 0))
        if not force and ident == self._ctx_id:
            return
        dst = self.arena.view(self.lay["ehs"], (self.B, self.nctx, self.net.ctx_dim), self.net.dtype)
        dst.copy_(ehs.to(self.device))
        if self.net.kind != "controlnet" or self._cond_id is not None:
            self.setup_plan.run(_stream())
        self._ctx_id = ident
        self._ctx_keep = ehs      # the identity is only an identity while the tensor lives: a freed temporary's address
        #                           (and version 0) is handed to the next temporary by the caching allocator

    def set_cond(self, cond: torch.Tensor):
        """ControlNet conditioning image [B,3,8H,8W] (NCHW, any float dtype)."""
        ident = (cond.data_ptr(), cond._version, tuple(cond.shape))
        if ident == self._cond_id:
            return
        c = self.lay["cond"]
        # the whole batch, or one CFG half (copied to both halves, as load_input does); anything else would leave rows stale
        if cond.shape[0] == c.B:
            self.load_nchw(cond, c.ptr, c.C, 0, c.H * c.W)
        elif 2 * cond.shape[0] == c.B:
            self.load_nchw(cond, c.ptr, c.C, 0, c.H * c.W, batch=c.B, batch_mod=cond.shape[0])
        else:
            raise L.PPError(f"controlnet_cond has batch {cond.shape[0]}, the network runs batch {c.B}")
        self._cond_id = ident
        self._cond_keep = cond    # (same reason as _ctx_keep)
        if self._ctx_id is not None:
            self.setup_plan.run(_stream())

    def load_nchw(self, src: torch.Tensor, dst_ptr: int, ldc: int, c0: int, hw: int, batch: Optional[int] = None,
                  batch_mod: int = 0):
        src = src.to(self.device)
        if src.dtype not in _DT:
            src = src.float()
        src = src.contiguous()
        nb = batch if batch is not None else src.shape[0]
        L.check(self.lib.pp_nchw_to_nhwc(src.data_ptr(), _DT[src.dtype], nb, src.shape[1], hw, batch_mod, dst_ptr, ldc,
                                         c0, L.dtype_code(self.net.dtype), _stream()), "pp_nchw_to_nhwc")

    def load_input(self, parts: Sequence[Tuple[torch.Tensor, int]]):
        """parts: (NCHW tensor, channel offset); a tensor with half the batch is CFG-duplicated."""
        x = self.lay["x_in"]
        for t, c0 in parts:
            mod = t.shape[0] if t.shape[0] != x.B else 0
            self.load_nchw(t, x.ptr, x.C, c0, x.H * x.W, batch=x.B, batch_mod=mod)

    def load_residual(self, group: str, idx: int, t: torch.Tensor):
        s = self.lay["slots"][group][idx]
        if tuple(t.shape) != (s.B, s.C, s.H, s.W):
            raise L.PPError(f"residual {group}[{idx}] has shape {tuple(t.shape)}, expected {(s.B, s.C, s.H, s.W)}")
        self.load_nchw(t, s.ptr, s.C, 0, s.H * s.W)

    # ------------------------------------------------------------------ run
    def run_step(self, use_graph: bool = False):
        if use_graph:
            if self.graph is None:
                self.capture()
            self.graph.replay()
        else:
            self.step_plan.run(_stream())

    def capture(self):
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.step_plan.run(s.cuda_stream)      # warm-up (sets func attributes outside capture)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.step_plan.run(_stream())
        self.graph = g

    def combines_in_kernel(self) -> bool:
        """Does the step plan hand tile counters to any split-K launch (PPGemmArgs.tile_ctr)?"""
        return any(getattr(a, "tile_ctr", None) for a in getattr(self.step_plan, "keep", []))

    def check_faults(self, blocking: bool = True):
        """The in-kernel split-K combine (csrc/gemm_combine.h) sums a tile's slabs through ONE XCD's L2.  Splits of a tile on
        different XCDs would count their arrivals in different L2s, never complete, give up, and leave their abandoned shares
        counted in the plan's fault word; so does a second-arrival wait that hit its bound.  A non-zero word means results since
        the last check may hold stale or missing partial sums: refuse.
        blocking=True synchronises and reads the word.  blocking=False costs the stream nothing: the word is copied to pinned
        host memory behind the work queued so far and examined by a LATER call (any check_faults) once that copy has landed --
        a pipeline call must not end in a device synchronisation (it would serialise the host's preparation of the next call
        behind this one's GPU work, +0.6 % at 50 steps), and a violation is a property of the box / plan, not of one call."""
        if self.arena is None or "fault" not in getattr(self, "lay", {}):
            return
        word = self.arena.view(self.lay["fault"], (1,), torch.int32)
        pend = self.__dict__.setdefault("_fault_pending", [])
        bad = 0
        if blocking:
            bad = int(word.item())
            pend.clear()
        else:
            while pend and pend[0][1].query():
                bad = max(bad, int(pend.pop(0)[0].item()))
            if len(pend) < 4:                                    # (bounded: the host may run many calls ahead)
                host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                host.copy_(word, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pend.append((host, ev))
        if bad:
            word.zero_()
            pend.clear()
            raise L.PPError(f"{bad} split-K shares were left uncombined or tiles combined across XCDs (workgroup placement "
                            f"changed under the plan): results are not trustworthy; rebuild the plan with PP_LAB=1 PP_FUSED_COMBINE=0")

    # ------------------------------------------------------------------ outputs
    def act_as_nchw(self, a: Act) -> torch.Tensor:
        """Zero-copy NCHW-logical (channels_last strides) bf16 torch view of an arena activation."""
        t = self.arena.view(a.ptr, (a.B, a.C, a.H, a.W), self.net.dtype,
                            strides=(a.H * a.W * a.C, 1, a.W * a.C, a.C))
        t._pp_nhwc_ptr = a.ptr
        return t

    def eps_tensor(self) -> torch.Tensor:
        return self.arena.view(self.outputs["eps"], (self.B, self.net.out_channels, self.H, self.W), torch.float32)
