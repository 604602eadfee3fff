"""Host-side execution engine for the MI355X denoising hot path.

Design (MI355X-first):
  * parameters live in ONE contiguous device buffer per network (`ParamPack`) in kernel-ready layouts
    ([Cout][ky][kx][Cin] bf16 conv weights, fused QKV / KV matrices, GEGLU-interleaved FF weights, fp32 biases and
    norm affine) -> a single RCCL broadcast moves a whole network over xGMI;
  * activations are NHWC bf16 inside one static `Arena` (stack-scoped bump allocator: temporaries of a resnet /
    transformer block are released at block exit so consecutive blocks reuse the same, Infinity-Cache-hot addresses);
  * a forward pass is compiled ONCE per (shape, wiring) into a `Plan` -- a flat list of C-ABI launches with baked
    pointers -- which is replayed eagerly (ctypes, ~2 us / launch) or captured into a hipGraph;
  * everything that varies per denoising step is read from device memory by the kernels (timestep, scheduler
    coefficients, step counter), so a captured step replays with zero host->device traffic.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L

ALIGN = 256


def _align(x: int, a: int = ALIGN) -> int:
    return (x + a - 1) // a * a


# ------------------------------------------------------------------------------------------------------------------
def _lab_switch(name: str) -> bool:
    """A/B switches of the launch-plan compiler (measurement scripts, bisecting).  They are honoured only when PP_LAB=1
    is set as well: the product has one code path, and a stray PP_* variable in a user's environment changes nothing."""
    return not (os.environ.get("PP_LAB") == "1" and os.environ.get(name, "1") == "0")


# (lab) PP_GN_EPILOGUE=0: every GroupNorm keeps its own statistics launch
GN_STATS_IN_EPILOGUE = _lab_switch("PP_GN_EPILOGUE")
# A GroupNorm apply behind a split-K launch at the 16x16 / 8x8 levels rides in that launch's combine (PPGemmArgs.gn_next_*).
# (lab) PP_GN_NEXT=0: the separate apply launch
GN_NEXT_IN_COMBINE = _lab_switch("PP_GN_NEXT")
# ResnetBlock2D's norm -> SiLU -> conv3x3 as ONE launch (csrc/conv_gn.hip).  (lab) PP_FUSE_GN_CONV=0: apply launch + conv
FUSE_GN_CONV = _lab_switch("PP_FUSE_GN_CONV")
# round 5: the CFG-identical prefix of a forward pass (conv_in .. first self-attention) on ONE half of the batch where the
# caller vouches for identical halves (SDNet.build_step(twin=True)).  (lab) PP_TWIN=0: the whole batch everywhere
TWIN_PREFIX = _lab_switch("PP_TWIN")
# round 6: the split-K combine inside the producing kernel, by the workgroup that arrives last at its tile
# (csrc/gemm_combine.h, PPGemmArgs.tile_ctr).  (lab) PP_FUSED_COMBINE=0: the separate combine launches
FUSED_COMBINE = _lab_switch("PP_FUSED_COMBINE")


class Arena:
    """Stack-scoped bump allocator over one device buffer (or a dry counting arena when device is None)."""

    def __init__(self, nbytes: int = 0, device=None):
        self.dry = device is None
        self.size = nbytes
        if not self.dry:
            self.buf = torch.zeros(max(nbytes, ALIGN) + ALIGN, dtype=torch.uint8, device=device)
            skew = (-self.buf.data_ptr()) % ALIGN
            self.buf = self.buf[skew:skew + max(nbytes, ALIGN)]
            self.base = self.buf.data_ptr()
        else:
            self.buf = None
            self.base = 1 << 20  # fake non-null base for dry runs
        self.off = 0
        self.peak = 0

    def alloc(self, nbytes: int) -> int:
        off = _align(self.off)
        self.off = off + nbytes
        self.peak = max(self.peak, self.off)
        if not self.dry and self.off > self.size:
            raise L.PPError(f"arena overflow: need {self.off} > {self.size}")
        return self.base + off

    def mark(self) -> int:
        return self.off

    def release(self, m: int):
        self.off = m

    def view(self, ptr: int, shape: Sequence[int], dtype: torch.dtype, strides: Optional[Sequence[int]] = None):
        """torch view of arena memory (element strides)."""
        esize = torch.empty((), dtype=dtype).element_size()
        off = ptr - self.base
        assert off % esize == 0
        typed = self.buf.view(dtype)
        if strides is None:
            n = 1
            for s in shape:
                n *= s
            return typed[off // esize: off // esize + n].view(*shape)
        return torch.as_strided(typed, tuple(shape), tuple(strides), off // esize)


class ParamPack:
    """All parameters of one network in one contiguous device buffer (single-collective broadcast unit)."""

    def __init__(self):
        self.items: List[Tuple[str, torch.Tensor]] = []
        self.offsets: Dict[str, int] = {}
        self.shapes: Dict[str, Tuple[torch.Size, torch.dtype]] = {}
        self.total = 0
        self.buf = None
        self.ptr: Dict[str, int] = {}
        self.version = 0      # bumped whenever the VALUES in `buf` change in place (re-broadcast, param_buffer() writes):
        #                       caches derived from the weights (time-embedding table, folded cross-attention operands) key on it

    def touch(self):
        self.version += 1

    def add(self, name: str, t: torch.Tensor, dtype: torch.dtype):
        assert name not in self.offsets, name
        t = t.detach().to(dtype).contiguous()
        off = _align(self.total)
        self.offsets[name] = off
        self.shapes[name] = (t.shape, dtype)
        self.total = off + t.numel() * t.element_size()
        self.items.append((name, t))

    def to_device(self, device, materialize: bool = True):
        self.buf = torch.zeros(_align(self.total) + ALIGN, dtype=torch.uint8, device=device)
        skew = (-self.buf.data_ptr()) % ALIGN
        self.buf = self.buf[skew:skew + _align(self.total)]
        base = self.buf.data_ptr()
        for name, t in self.items:
            off = self.offsets[name]
            if materialize and t.device.type != "meta":
                nb = t.numel() * t.element_size()
                self.buf[off:off + nb].copy_(t.view(-1).view(torch.uint8))
            self.ptr[name] = base + off
        self.items = []  # drop host copies
        return self

    def tensor(self, name: str) -> torch.Tensor:
        shape, dtype = self.shapes[name]
        off = self.offsets[name]
        n = 1
        for s in shape:
            n *= s
        es = torch.empty((), dtype=dtype).element_size()
        return self.buf[off:off + n * es].view(dtype).view(shape)


@dataclass
class Act:
    """NHWC bf16 activation [B][H][W][C] living in an arena."""
    ptr: int
    B: int
    H: int
    W: int
    C: int
    producer: object = None   # PPGemmArgs of the launch that writes this tensor (GroupNorm statistics subscribe to it)
    dup_half: int = 0         # > 0: the producer computed batch items [0, dup_half) and stored every row twice (the CFG twin
    #                           prefix, PPGemmArgs.out_dup_rows); this Act is the full-batch view of that tensor

    @property
    def rows(self) -> int:
        return self.B * self.H * self.W

    @property
    def nbytes(self) -> int:
        return self.rows * self.C * 2


class Plan:
    """Flat list of C-ABI launches with baked arguments."""

    def __init__(self):
        self.calls: List[Tuple] = []     # (fn, args_tuple, name)
        self.keep: List = []             # keep ctypes structs alive
        self.flops = 0.0                 # algorithmic FLOPs (2*MAC of conv/linear/attention matmuls)
        self.flops_by_kind: Dict[str, float] = {}

    def add(self, name: str, fn, *args):
        self.calls.append((fn, args, name))

    def count(self, kind: str, flops: float):
        self.flops += flops
        self.flops_by_kind[kind] = self.flops_by_kind.get(kind, 0.0) + flops

    def run(self, stream: int):
        for fn, args, name in self.calls:
            rc = fn(*args, stream)
            if rc != 0:
                L.check(rc, name)

    def run_timed(self, stream_obj) -> Dict[str, float]:
        """Eager replay with a HIP event pair around every launch (on the launch stream); ms per kernel name."""
        evs = []
        s = stream_obj.cuda_stream
        for fn, args, name in self.calls:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream_obj)
            rc = fn(*args, s)
            e1.record(stream_obj)
            if rc != 0:
                L.check(rc, name)
            evs.append((name, e0, e1))
        stream_obj.synchronize()
        out: Dict[str, float] = {}
        self.last_launch_ms = []
        for name, e0, e1 in evs:
            dt = e0.elapsed_time(e1)
            out[name] = out.get(name, 0.0) + dt
            self.last_launch_ms.append(dt)
        return out

    def describe(self, i: int) -> str:
        """Human-readable shape of launch i (GEMM-family launches carry a PPGemmArgs)."""
        fn, args, name = self.calls[i]
        a = getattr(args[0], "_obj", None) if args else None
        if isinstance(a, L.PPGemmArgs):
            return (f"{name} M={a.M} N={a.N} K={a.K} {'conv' if a.x_mode else 'lin'}"
                    f"{' s2' if a.stride == 2 else ''}{' up' if a.up else ''}{' cat' if a.c2 else ''}")
        return name


class Builder:
    """Appends launches to a Plan, allocating outputs / workspaces from an Arena."""

    def __init__(self, arena: Arena, plan: Optional[Plan] = None, dtype: torch.dtype = torch.bfloat16):
        self.arena = arena
        self.plan = plan or Plan()
        self.lib = L.lib()
        self.dtype = dtype                # 16-bit storage format of activations / matrix weights
        self.dt = L.dtype_code(dtype)     # ... as the C ABI's dtype code
        self.gemm_tile = 0       # tuning overrides (0 = auto)
        self.gemm_splitk = 0
        self.last_gemm = None    # PPGemmArgs of the most recent GEMM launch
        # GroupNorm statistics accumulated by the producers' epilogues: [gn_acc_base, +gn_acc_cap) bytes of persistent
        # arena memory for the int64 accumulators (0 = every GroupNorm keeps its own statistics launch)
        self.gn_acc_base = 0
        self.gn_acc_cap = 0
        self.gn_acc_used = 0
        # device word the in-kernel split-K combine counts placement violations in (NetRuntime.check_faults); 0 = none given
        self.fault_ptr = 0

    # -- memory
    def alloc(self, nbytes: int) -> int:
        return self.arena.alloc(nbytes)

    def new_act(self, B, H, W, Cc) -> Act:
        return Act(self.alloc(B * H * W * Cc * 2), B, H, W, Cc)

    def mark(self):
        return self.arena.mark()

    def release(self, m):
        self.arena.release(m)

    # -- ops
    def _gemm(self, a: L.PPGemmArgs, name: str):
        if a.tile == 0:
            a.tile = self.gemm_tile
        if a.splitk == 0:
            a.splitk = self.gemm_splitk
        a.dtype = self.dt
        ws = self.lib.pp_gemm_workspace_bytes(C.byref(a))
        if ws:
            a.workspace = self.alloc(ws)
            # tile counters of the in-kernel combine: persistent words of the pool the step's first launch zeroes (private to
            # this launch; every launch also leaves them zero).  A permission: pp_gemm_bf16 decides per launch.
            if FUSED_COMBINE and self.gn_acc_cap:
                nctr = _align(self.lib.pp_gemm_combine_ctr_bytes(C.byref(a)), 8)
                if nctr and self.gn_acc_used + nctr <= self.gn_acc_cap:
                    a.tile_ctr = self.gn_acc_base + self.gn_acc_used
                    a.combine_fault = self.fault_ptr or None
                    self.gn_acc_used += nctr
        self.plan.keep.append(a)
        self.last_gemm = a
        a._arena_top = self.arena.off       # everything this launch reads or scratches lies below (see _apply_in_producer_combine)
        self.plan.add(name, self.lib.pp_gemm_bf16, C.byref(a))
        self.plan.count(name, 2.0 * a.M * a.N * a.K)

    def linear(self, x: int, rows: int, K: int, w: int, N: int, bias: int = 0, ldx: Optional[int] = None,
               x2: int = 0, K2: int = 0, ldx2: int = 0, res1: int = 0, ldres1: int = 0, res2: int = 0,
               ldres2: int = 0, scale: float = 1.0, act: int = 0, out: int = 0, ldo: Optional[int] = None,
               rowvec: int = 0, ld_rowvec: int = 0, rows_per_batch: int = 0, out_vt: int = 0, vt_col0: int = 0,
               vt_ld: int = 0, out_f32: bool = False, row_stats_out: int = 0, ln_stats: int = 0, ln_colsum: int = 0,
               ln_tiles: int = 0, ln_dim: int = 0, ln_eps: float = 1e-5, name: str = "gemm", res1_wrap: int = 0,
               w_batch_stride: int = 0, vec_batch_stride: int = 0) -> int:
        """out[rows][N] = epilogue(X[rows][K(+K2)] @ W[N][K+K2]^T).  Returns the output pointer.
        row_stats_out / ln_*: LayerNorm folded across two GEMMs (see include/pp_hip.h).  res1_wrap: res1 holds only that
        many rows (one half of a CFG pair), row m adds row m mod res1_wrap.  w_batch_stride / vec_batch_stride: one weight
        matrix (and bias / column-sum vector) per batch item of rows_per_batch rows."""
        n_out = N // 2 if act == L.PP_ACT_GEGLU else (vt_col0 if out_vt else N)
        if ldo is None:
            ldo = n_out
        if not out:
            out = self.alloc(rows * ldo * (4 if out_f32 else 2))
        m = self.mark()
        a = L.PPGemmArgs()
        a.M, a.N, a.K, a.x_mode = rows, N, K + K2, L.PP_X_PLAIN
        a.x1, a.x2, a.c1, a.c2 = x, x2 or None, K, K2
        a.ldx1, a.ldx2 = (ldx if ldx is not None else K), (ldx2 or K2)
        a.w, a.bias = w, bias or None
        a.w_batch_stride, a.vec_batch_stride = w_batch_stride, vec_batch_stride
        a.rowvec, a.ld_rowvec, a.rows_per_batch = rowvec or None, ld_rowvec, rows_per_batch
        a.res1, a.ldres1 = res1 or None, ldres1 or N
        a.res1_wrap_rows = res1_wrap
        a.res2, a.ldres2 = res2 or None, ldres2 or N
        a.scale, a.act = scale, act
        a.out, a.ldo, a.out_f32 = out, ldo, int(out_f32)
        a.out_vt, a.vt_col0, a.vt_ld = out_vt or None, vt_col0, vt_ld
        a.row_stats_out = row_stats_out or None
        if ln_stats:
            a.ln_stats, a.ln_colsum, a.ln_tiles, a.ln_dim, a.ln_eps = ln_stats, ln_colsum, ln_tiles, ln_dim, ln_eps
        self._gemm(a, name)
        self.release(m)
        return out

    def conv3x3(self, x: Act, w: int, cout: int, bias: int = 0, stride: int = 1, up: bool = False,
                x2: Optional[Act] = None, rowvec: int = 0, res1: int = 0, res2: int = 0, scale: float = 1.0,
                out: Optional[Act] = None, x3: Optional[Act] = None, x4: Optional[Act] = None,
                name: str = "conv3x3", gn_in: Optional[Tuple[int, int, int, float, int]] = None, dup: bool = False) -> Act:
        """gn_in = (gamma, beta, gamma_beta_interleaved, eps, groups): the conv input is SiLU(GroupNorm(concat(x, x2))).
        Where the statistics arrive from the producers' epilogues and the shape suits csrc/conv_gn.hip the norm runs in
        the conv's loader (ONE launch, the normalised activation is never written); else pp_groupnorm_apply(_acc) + conv.
        dup: x is ONE half of a CFG pair whose halves are identical; the launch stores every output row twice and the
        returned Act is the full batch (PPGemmArgs.out_dup_rows, Act.dup_half)."""
        if gn_in is not None:
            gamma, beta, gb, eps, groups = gn_in
            acc = self._subscribe_gn_stats(x, x2, groups) if FUSE_GN_CONV else 0
            fused = None
            if acc:
                fused = (acc, gb, groups, eps)
            else:
                x, x2 = self.groupnorm(x, gamma, beta, eps, True, x2=x2, groups=groups), None
        else:
            fused = None
        hv, wv = (x.H * 2, x.W * 2) if up else (x.H, x.W)
        ho, wo = (hv + 2 - 3) // stride + 1, (wv + 2 - 3) // stride + 1
        if dup:
            assert out is None and gn_in is None
            out = self.new_act(2 * x.B, ho, wo, cout)
            out.dup_half = x.B
        if out is None:
            out = self.new_act(x.B, ho, wo, cout)
        m = self.mark()
        a = L.PPGemmArgs()
        c2 = x2.C if x2 is not None else 0
        c3 = x3.C if x3 is not None else 0
        c4 = x4.C if x4 is not None else 0
        a.M, a.N, a.K, a.x_mode = x.B * ho * wo, cout, 9 * (x.C + c2) + c3 + c4, L.PP_X_CONV3X3
        a.x1, a.x2, a.c1, a.c2 = x.ptr, (x2.ptr if x2 is not None else None), x.C, c2
        if x3 is not None:          # 1x1 tail over concat(x3, x4) at the output pixel (merged conv_shortcut)
            a.x3, a.c3 = x3.ptr, c3
            if x4 is not None:
                a.x4, a.c4 = x4.ptr, c4
        a.batch, a.hin, a.win, a.hout, a.wout, a.stride, a.up = x.B, x.H, x.W, ho, wo, stride, int(up)
        a.w, a.bias = w, bias or None
        a.rowvec, a.ld_rowvec, a.rows_per_batch = rowvec or None, 0, ho * wo
        a.res1, a.ldres1, a.res2, a.ldres2 = res1 or None, cout, res2 or None, cout
        a.scale, a.act = scale, 0
        a.out, a.ldo, a.out_f32 = out.ptr, cout, 0
        if dup:
            a.out_dup_rows = a.M
            a.dtype = self.dt
            if self.lib.pp_gemm_workspace_bytes(C.byref(a)):      # (split-K: the combine does not write twins)
                raise L.PPError("twin-prefix conv would run split-K; the caller must not request dup for this shape")
        if fused is not None:
            a.gn_in_acc, a.gn_in_gb, a.gn_in_groups, a.gn_in_eps, a.gn_in_silu = fused[0], fused[1], fused[2], fused[3], 1
            a.dtype = self.dt
            if not self.lib.pp_conv_gn_preferred(C.byref(a)):
                # (the statistics subscription stays: the apply launch reads the same accumulators)
                a.gn_in_acc, a.gn_in_gb, a.gn_in_groups, a.gn_in_silu = None, None, 0, 0
                gamma, beta, gb, eps, groups = gn_in
                self.release(m)
                xn = self._apply_in_producer_combine(x, fused[0], gamma, beta, eps, True, None) if x2 is None else None
                if xn is None:
                    xn = self.new_act(x.B, x.H, x.W, x.C + c2)
                    self.plan.add("groupnorm_apply", self.lib.pp_groupnorm_apply_acc, x.ptr, x.C,
                                  x2.ptr if x2 is not None else None, c2, x.B, x.H * x.W, groups, eps, gamma, beta, fused[0],
                                  1, xn.ptr, self.dt)
                m = self.mark()
                a.x1, a.x2, a.c1, a.c2 = xn.ptr, None, x.C + c2, 0
        self._gemm(a, name)
        out.producer = a
        self.release(m)
        return out

    def groupnorm(self, x: Act, gamma: int, beta: int, eps: float, silu: bool, x2: Optional[Act] = None,
                  groups: int = 32, out: Optional[Act] = None) -> Act:
        c2 = x2.C if x2 is not None else 0
        Ct = x.C + c2
        hw = x.H * x.W
        out_given = out
        x2p = x2.ptr if x2 is not None else None
        acc = self._subscribe_gn_stats(x, x2, groups)
        if acc and x2 is None:
            fused = self._apply_in_producer_combine(x, acc, gamma, beta, eps, silu, out_given)
            if fused is not None:
                return fused
        if out is None:
            out = self.new_act(x.B, x.H, x.W, Ct)
        if acc:
            # the statistics arrive from the epilogues of the launches that produced x (and x2): no stats launch
            self.plan.add("groupnorm_apply", self.lib.pp_groupnorm_apply_acc, x.ptr, x.C, x2p, c2, x.B, hw, groups, eps,
                          gamma, beta, acc, int(silu), out.ptr, self.dt)
            return out
        m = self.mark()
        ws = self.alloc(self.lib.pp_groupnorm_workspace_bytes(x.B, hw, Ct))
        self.plan.add("groupnorm_stats", self.lib.pp_groupnorm_stats, x.ptr, x.C, x2p, c2, x.B, hw, groups, ws,
                      self.dt)
        self.plan.add("groupnorm_apply", self.lib.pp_groupnorm_apply, x.ptr, x.C, x2p, c2, x.B, hw, groups, eps,
                      gamma, beta, ws, int(silu), out.ptr, self.dt)
        self.release(m)
        return out

    def _apply_in_producer_combine(self, x: Act, acc: int, gamma: int, beta: int, eps: float, silu: bool,
                                   out: Optional[Act]) -> Optional[Act]:
        """The apply of a single-tensor GroupNorm whose input was written by the launch right in front of it in the plan: if
        that launch ends in the split-K combine that owns whole (batch item, group) populations (pp_gemm_gn_next_ok: the
        16x16 and 8x8 levels), the combine writes the normalised tensor as well and no apply launch is added.  `acc` = the
        statistics subscription of this norm (its accumulators are still filled)."""
        a = x.producer
        if not GN_NEXT_IN_COMBINE or a is None or not acc or not self.plan.calls or a.gn_next_out or \
                getattr(a, "_fused_ff", False):
            return None
        last = self.plan.calls[-1][1]
        if not last or getattr(last[0], "_obj", None) is not a:      # something ran in between: its scratch may alias
            return None
        sub = 0 if a.gn_acc[0] == acc else 1 if a.gn_acc[1] == acc else -1
        if sub < 0 or not self.lib.pp_gemm_gn_next_ok(C.byref(a), sub):
            return None
        if out is None:
            # the producer's combine writes this tensor while it still reads its split-K slabs and residuals -- scratch the
            # caller has RELEASED by now: allocate above everything that was live when the producer was recorded
            self.arena.off = max(self.arena.off, getattr(a, "_arena_top", self.arena.off))
            out = self.new_act(x.B, x.H, x.W, x.C)
        elif out.ptr < self.arena.base + getattr(a, "_arena_top", 0):
            return None                                     # (a caller-provided buffer that may alias that scratch)
        a.gn_next_out, a.gn_next_gamma, a.gn_next_beta = out.ptr, gamma, beta
        a.gn_next_eps, a.gn_next_silu, a.gn_next_sub = eps, int(silu), sub
        return out

    def _subscribe_gn_stats(self, x: Act, x2: Optional[Act], groups: int) -> int:
        """If every input tensor of this GroupNorm is written by a GEMM launch whose epilogue can accumulate the
        statistics, patch those launches (PPGemmArgs.gn_acc slot) and return the accumulator pointer, else 0."""
        if not (GN_STATS_IN_EPILOGUE and self.gn_acc_cap):
            return 0
        hw = x.H * x.W
        parts = [(x, 0)] + ([(x2, x.C)] if x2 is not None else [])
        Ct = x.C + (x2.C if x2 is not None else 0)
        if hw % 64 or Ct % groups or Ct // groups < 8:     # (>= 8 channels per group: LDS group slots per 160-col tile)
            return 0
        for t, _ in parts:
            a = t.producer
            if a is None or (a.gn_acc[0] and a.gn_acc[1]):
                return 0
            rpb = a.rows_per_batch
            a.rows_per_batch = hw
            ok = self.lib.pp_gemm_gn_stats_ok(C.byref(a))
            a.rows_per_batch = rpb
            if not ok or (rpb not in (0, hw)):
                return 0
        nbytes = x.B * groups * 2 * 8
        if self.gn_acc_used + nbytes > self.gn_acc_cap:
            return 0
        acc = self.gn_acc_base + self.gn_acc_used
        self.gn_acc_used += nbytes
        for t, c0 in parts:
            a = t.producer
            k = 0 if not a.gn_acc[0] else 1
            a.rows_per_batch = hw
            a.gn_acc[k], a.gn_cg[k], a.gn_c0[k], a.gn_groups[k] = acc, Ct // groups, c0, groups
            if t.dup_half:       # this consumer sees the full (twice-stored) tensor: both halves' accumulators get the sums
                a.gn_dup_batch, a.gn_dup_mask = t.dup_half, a.gn_dup_mask | (1 << k)
        return acc

    def layernorm(self, x: int, rows: int, Cc: int, gamma: int, beta: int, eps: float = 1e-5) -> int:
        out = self.alloc(rows * Cc * 2)
        self.plan.add("layernorm", self.lib.pp_layernorm, x, rows, Cc, gamma, beta, eps, out, self.dt)
        return out

    def attention(self, q: int, ldq: int, k: int, ldk: int, vt: int, ldvt: int, B: int, heads: int, nq: int,
                  nk: int, d: int, out: int = 0, ldo: int = 0, log2q: bool = False) -> int:
        """log2q: q already holds Q * d^-0.5 * log2(e) (pp_tfront q_scale) -> the PP_ATTN_PIPE_LOG2 kernel."""
        Cc = heads * d
        if not out:
            out, ldo = self.alloc(B * nq * Cc * 2), Cc
        if log2q:
            self.plan.add("attention", self.lib.pp_attention_fwd_variant, q, ldq, k, ldk, vt, ldvt, out, ldo, B, heads, nq,
                          nk, d, float(d) ** -0.5, self.dt, L.PP_ATTN_PIPE_LOG2)
        else:
            self.plan.add("attention", self.lib.pp_attention_fwd, q, ldq, k, ldk, vt, ldvt, out, ldo, B, heads, nq, nk,
                          d, float(d) ** -0.5, self.dt)
        self.plan.count("attention", 4.0 * B * heads * nq * nk * d)
        return out

    def ff_fused(self, hs: int, rows: int, Cc: int, hw: int, w1: int, b1: int, cs1: int, ln_stats: int, ln_tiles: int,
                 w2: int, bias2: int, out: int, res1: int = 0, res1_wrap: int = 0, res2: int = 0, w2_kperm: bool = False):
        """FeedForward (GEGLU) + FF2 . proj_out of a C = 320 transformer as ONE launch (csrc/ff_fused.hip, pp_ff_fused): the
        [rows][4C] GEGLU tensor is never written.  The launch record is the PPGemmArgs of the SECOND GEMM (what the two-launch
        plan hands pp_gemm_bf16 for `[g | hs] [W_po W_ff2 | W_po]^T`), so GroupNorm-statistics subscriptions of the consumer
        patch it exactly as they patch a GEMM.  Returns that record."""
        a = L.PPGemmArgs()
        a.M, a.N, a.K, a.x_mode = rows, Cc, 5 * Cc, L.PP_X_PLAIN
        a.x1, a.x2, a.c1, a.c2 = hs, hs, 4 * Cc, Cc          # (x1 is never read: the tensor it would name does not exist)
        a.ldx1, a.ldx2 = 4 * Cc, Cc
        a.w, a.bias = w2, bias2 or None
        a.res1, a.ldres1, a.res1_wrap_rows = res1 or None, Cc, res1_wrap
        a.res2, a.ldres2 = res2 or None, Cc
        a.scale, a.act = 1.0, 0
        a.out, a.ldo, a.out_f32 = out, Cc, 0
        a.rows_per_batch = hw
        a.splitk, a.tile = 1, 0
        a.dtype = self.dt
        a._fused_ff = True                  # (no split-K combine behind this launch: _apply_in_producer_combine keeps off)
        self.plan.keep.append(a)
        self.last_gemm = a
        a._arena_top = self.arena.off
        self.plan.add("ff_fused", self.lib.pp_ff_fused, C.byref(a), w1, b1, cs1 or None, ln_stats or None, ln_tiles, 1e-5,
                      int(w2_kperm))
        self.plan.count("ff_fused", 2.0 * rows * 8 * Cc * Cc + 2.0 * rows * Cc * 5 * Cc)   # (FF1 + [g | hs] W2'^T)
        return a

    def add(self, a: int, b: int, out: int, n: int):
        self.plan.add("add", self.lib.pp_add_bf16, a, b, out, n, self.dt)


# ------------------------------------------------------------------------------------------------------------------
# parameter preparation (diffusers state-dict keys -> kernel layouts)
# ------------------------------------------------------------------------------------------------------------------
def _conv_igemm(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [Cout, 9*Cin] with k = (ky*3+kx)*Cin + ci."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def _conv_direct(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [3,3,Cin,Cout]."""
    return w.permute(2, 3, 1, 0)


CIN_CHUNK = 64   # K chunk of the implicit-GEMM conv: one (tap, 64 channels) slice per main-loop step


def _conv_igemm_cpad(w: torch.Tensor, cpad: int) -> torch.Tensor:
    """conv_in ([Cout, 4|9, 3, 3]) for the implicit-GEMM kernel: input channels zero-padded to `cpad` (one 64-deep K
    chunk per tap; the network input buffer carries the same zero channels), then the igemm layout."""
    pad = cpad - w.shape[1]
    if pad:
        w = torch.cat([w, torch.zeros(w.shape[0], pad, 3, 3, dtype=w.dtype, device=w.device)], 1)
    return _conv_igemm(w)


def _kperm(w: torch.Tensor) -> torch.Tensor:
    """[N][K] -> the input index permuted inside every group of 32: storage position 8 kg + j holds index
    16 (j >> 2) + 4 kg + (j & 3).  A lane's 16-byte A fragment (k-group kg) of a GEMM whose B operand comes straight out of
    the previous GEMM's MFMA accumulators (four consecutive columns per register quad) then lines up with it: the layout
    csrc/tfront.hip and csrc/xattn_fused.hip chain two GEMMs with."""
    K = w.shape[-1]
    assert K % 32 == 0
    kp = torch.arange(K)
    s32, kg, j = kp // 32, (kp // 8) % 4, kp % 8
    kk = 32 * s32 + 16 * (j // 4) + 4 * kg + (j % 4)
    return w[..., kk]


def _kperm_geglu(w: torch.Tensor) -> torch.Tensor:
    """[N][K] -> the input (hidden-unit) index permuted inside every group of 32: storage position 8 kg + 2 q + e holds unit
    8 q + 2 kg + e.  The GEGLU of an MFMA accumulator quad (h0, h1, g0, g1) is two hidden units (2 kg, 2 kg + 1 of the
    16-column block q), so a lane's four quads of a 64-column chunk ARE its 16-byte B fragment of a GEMM whose A operand is
    packed like this: how csrc/ff_fused.hip chains FF1 -> GEGLU -> FF2 in registers."""
    K = w.shape[-1]
    assert K % 32 == 0
    sp = torch.arange(K)
    s32, r = sp // 32, sp % 32
    kg, q, e = r // 8, (r // 2) % 4, r % 2
    return w[..., 32 * s32 + 8 * q + 2 * kg + e]


def _geglu_interleave(w: torch.Tensor) -> torch.Tensor:
    """GEGLU proj rows [h(0..F) ; g(0..F)] -> groups of four rows (h_{2q}, h_{2q+1}, g_{2q}, g_{2q+1})."""
    F2 = w.shape[0]
    F = F2 // 2
    h, g = w[:F], w[F:]
    rest = w.shape[1:]
    h = h.reshape(F // 2, 2, *rest)
    g = g.reshape(F // 2, 2, *rest)
    return torch.cat([h, g], dim=1).reshape(F2, *rest)


class SDNet:
    """One network of the SD-1.5 family ("unet" | "brushnet" | "controlnet") compiled to HIP launch plans.

    Mirrors the wiring of /root/reference/powerpaint/models/unet_2d_condition.py:1040-1363 (UNet incl. the fork's
    *_add_samples), /root/reference/powerpaint/models/BrushNet_CA.py:690-952 (BrushNet) and the diffusers-0.27.0
    ControlNetModel used at /root/reference/powerpaint/pipelines/pipeline_PowerPaint_ControlNet.py:1686-1694.
    """

    # BasicTransformerBlock.norm1/2/3 folded into the GEMMs on either side (no LayerNorm launch, no normalised copy in
    # HBM).  (lab) PP_FOLD_LN=0 keeps the stand-alone pp_layernorm launches (A/B measurements, bisecting).
    fold_ln = _lab_switch("PP_FOLD_LN")
    # FeedForward.net[2] and Transformer2DModel.proj_out composed into one GEMM ((lab) PP_MERGE_FF2=0: two launches)
    merge_ff2_proj_out = _lab_switch("PP_MERGE_FF2")
    fuse_conv_out = _lab_switch("PP_FUSE_CONV_OUT")      # (lab) =0: conv_norm_out apply and conv_out as two launches
    # the C = 320 cross-attention sub-blocks (norm2 -> to_q -> 77-key attention -> to_out + residual) as ONE pp_xattn_block
    # launch each, K / V folded into the projections once per prompt (pp_xattn_fold in the setup plan): 55 against 66 us
    # per block at 64x64, step -0.55 % (profiles/r03_xattn_fused_ab.txt).  (lab) PP_XATTN_FUSED=0: the three-launch chain
    fuse_xattn = _lab_switch("PP_XATTN_FUSED")
    # round 4: Transformer2DModel.norm -> proj_in -> LayerNorm1-folded QKV at C = 320 as ONE launch (csrc/tfront.hip;
    # three launches and two activation round trips before).  (lab) PP_TFRONT=0: the chain
    fuse_tfront = _lab_switch("PP_TFRONT")
    # round 5: BasicTransformerBlock.attn1.to_out (+ residual) rides in front of the fused cross-attention block at C = 320
    # (five more weight slabs in pp_xattn_block, one launch and one hidden-state round trip less).  (lab) PP_XATTN_PRE=0
    fuse_xattn_pre = _lab_switch("PP_XATTN_PRE")
    # round 5: FeedForward (GEGLU) + FF2 . proj_out at C = 320 as ONE launch with the hidden dimension streamed
    # (csrc/ff_fused.hip; two launches and the [M][4C] GEGLU round trip before).  (lab) PP_FF_FUSED=0: the two launches
    fuse_ff = _lab_switch("PP_FF_FUSED")
    # ... as the 8-wave kernel (two waves per SIMD, GEGLU values exchanged through LDS, W2' in natural hidden order).
    # (lab) PP_FF_W8=0: the 4-wave kernel (one wave per SIMD, GEGLU chained in registers, W2' hidden index permuted)
    ff_w8 = _lab_switch("PP_FF_W8")
    # round 5: the fused front end hands Q over pre-multiplied by d^-0.5 * log2(e) and self-attention runs the
    # PP_ATTN_PIPE_LOG2 form of the pipelined kernel (no multiply-add per score).  (lab) PP_ATTN_LOG2=0: plain Q
    attn_log2 = _lab_switch("PP_ATTN_LOG2")
    # round 4: the same for C = 640 / 1280 (64-row tiles x 320-column groups, xattn_wide_kernel).  (lab) PP_XATTN_WIDE=0
    # keeps the chain at those levels
    fuse_xattn_wide = _lab_switch("PP_XATTN_WIDE")
    # ... up to this width: at C = 1280 the kernel loses (59 against 47 us per block: four column groups recompute the
    # logits, and a 4-wave workgroup pays ~150 cycles of issue per LDS-DMA piece with no partner wave to hide it);
    # (lab) PP_XATTN_WIDE_C=1280 runs it there too
    xattn_wide_max_c = int(os.environ.get("PP_XATTN_WIDE_C", "640")) if os.environ.get("PP_LAB") == "1" else 640
    # round 5: above that width (C = 1280: the 16x16 level and the mid block) the folded sub-block runs as TWO plain GEMMs with
    # per-prompt weights -- logits + per-head softmax in the first one's epilogue (PP_ACT_SOFTMAX80), probabilities x H^T +
    # residual in the second -- instead of to_q -> 77-key attention -> to_out: one launch less and half the multiplications
    # per block.  (lab) PP_XATTN_2G=0: the three-launch chain
    fuse_xattn_2g = _lab_switch("PP_XATTN_2G")
    # ResnetBlock2D.conv_shortcut merged into conv2 as a 1x1 K tail (needs 64-channel multiples; (lab) PP_MERGE_SHORTCUT=0 off)
    _merge_shortcut_env = _lab_switch("PP_MERGE_SHORTCUT")

    def __init__(self, kind: str, in_channels: int, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 heads=8, cross_attention_dim=768, groups=32, eps=1e-5,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 conditioning_channels: int = 0, cond_embed_channels=(16, 32, 96, 256), out_channels: int = 4,
                 dtype: torch.dtype = torch.bfloat16):
        assert kind in ("unet", "brushnet", "controlnet")
        L.dtype_code(dtype)              # bf16 | fp16 (raises otherwise)
        self.dtype = dtype
        self.kind = kind
        self.in_channels = in_channels
        self.conditioning_channels = conditioning_channels
        # channels of the network input as conv_in sees it, and as the input buffer stores it (zero-padded: see build_step)
        self.cin0 = in_channels + (conditioning_channels if kind == "brushnet" else 0)
        self.cin_pad = _align(self.cin0, CIN_CHUNK)
        self.boc = tuple(block_out_channels)
        self.merge_shortcut = self._merge_shortcut_env and all(c % 64 == 0 for c in self.boc)
        self.L = layers_per_block
        self.heads = heads
        self.ctx_dim = cross_attention_dim
        self.groups, self.eps = groups, eps
        self.down_types, self.up_types = tuple(down_block_types), tuple(up_block_types)
        self.cond_embed_channels = tuple(cond_embed_channels)
        self.out_channels = out_channels
        self.params: Optional[ParamPack] = None
        self.P: Dict[str, int] = {}
        self.temb_off: Dict[str, int] = {}
        self.temb_total = 0
        for c in self.boc:
            if c % 64 or (c // heads) not in (40, 80, 160):
                raise L.PPError(f"unsupported channel width {c} (need multiple of 64 and head_dim in 40/80/160)")

    # ---------------------------------------------------------------- structure walk (shared by params and plans)
    def _resnet_specs(self) -> List[Tuple[str, int, int]]:
        """(prefix, cin, cout) of every ResnetBlock2D in execution order."""
        out = []
        boc, Lp = self.boc, self.L
        oc = boc[0]
        for i in range(len(boc)):
            ic, oc = oc, boc[i]
            for j in range(Lp):
                out.append((f"down_blocks.{i}.resnets.{j}", ic if j == 0 else oc, oc))
        out.append(("mid_block.resnets.0", boc[-1], boc[-1]))
        out.append(("mid_block.resnets.1", boc[-1], boc[-1]))
        if self.kind != "controlnet":
            rev = list(reversed(boc))
            oc = rev[0]
            for i in range(len(boc)):
                prev, oc = oc, rev[i]
                ic = rev[min(i + 1, len(boc) - 1)]
                for j in range(Lp + 1):
                    skip = ic if j == Lp else oc
                    rin = prev if j == 0 else oc
                    out.append((f"up_blocks.{i}.resnets.{j}", rin + skip, oc))
        return out

    def _attn_specs(self) -> List[Tuple[str, int]]:
        out = []
        for i, t in enumerate(self.down_types):
            if t == "CrossAttnDownBlock2D":
                out += [(f"down_blocks.{i}.attentions.{j}", self.boc[i]) for j in range(self.L)]
        out.append(("mid_block.attentions.0", self.boc[-1]))
        if self.kind != "controlnet":
            rev = list(reversed(self.boc))
            for i, t in enumerate(self.up_types):
                if t == "CrossAttnUpBlock2D":
                    out += [(f"up_blocks.{i}.attentions.{j}", rev[i]) for j in range(self.L + 1)]
        return out

    def _zero_conv_specs(self) -> List[Tuple[str, int]]:
        """(state-dict prefix, channels) of the 1x1 output convs (BrushNet 12+1+15, ControlNet 12+1)."""
        if self.kind == "unet":
            return []
        nm = "brushnet" if self.kind == "brushnet" else "controlnet"
        out = [(f"{nm}_down_blocks.0", self.boc[0])]
        k = 1
        for i, c in enumerate(self.boc):
            for _ in range(self.L):
                out.append((f"{nm}_down_blocks.{k}", c)); k += 1
            if i != len(self.boc) - 1:
                out.append((f"{nm}_down_blocks.{k}", c)); k += 1
        out.append((f"{nm}_mid_block", self.boc[-1]))
        if self.kind == "brushnet":
            k = 0
            for i, c in enumerate(reversed(self.boc)):
                for _ in range(self.L + 1):
                    out.append((f"brushnet_up_blocks.{k}", c)); k += 1
                if i != len(self.boc) - 1:
                    out.append((f"brushnet_up_blocks.{k}", c)); k += 1
        return out

    # ---------------------------------------------------------------- parameters
    def state_dict_spec(self) -> Dict[str, Tuple[int, ...]]:
        """diffusers-format parameter names -> shapes for this architecture (SURVEY.md section 8b "Weight naming")."""
        sp: Dict[str, Tuple[int, ...]] = {}
        boc, te = self.boc, self.boc[0] * 4

        def conv(name, cout, cin, k):
            sp[name + ".weight"] = (cout, cin, k, k)
            sp[name + ".bias"] = (cout,)

        def lin(name, cout, cin, bias=True):
            sp[name + ".weight"] = (cout, cin)
            if bias:
                sp[name + ".bias"] = (cout,)

        def norm(name, c):
            sp[name + ".weight"] = (c,)
            sp[name + ".bias"] = (c,)

        conv("conv_in_condition" if self.kind == "brushnet" else "conv_in", boc[0], self.cin0, 3)
        lin("time_embedding.linear_1", te, boc[0])
        lin("time_embedding.linear_2", te, te)
        for pre, cin, cout in self._resnet_specs():
            norm(pre + ".norm1", cin); conv(pre + ".conv1", cout, cin, 3); lin(pre + ".time_emb_proj", cout, te)
            norm(pre + ".norm2", cout); conv(pre + ".conv2", cout, cout, 3)
            if cin != cout:
                conv(pre + ".conv_shortcut", cout, cin, 1)
        for i in range(len(boc) - 1):
            conv(f"down_blocks.{i}.downsamplers.0.conv", boc[i], boc[i], 3)
            if self.kind != "controlnet":
                c = list(reversed(boc))[i]
                conv(f"up_blocks.{i}.upsamplers.0.conv", c, c, 3)
        for pre, c in self._attn_specs():
            norm(pre + ".norm", c); conv(pre + ".proj_in", c, c, 1); conv(pre + ".proj_out", c, c, 1)
            tb = pre + ".transformer_blocks.0"
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{tb}.{n}", c)
            for a, kd in (("attn1", c), ("attn2", self.ctx_dim)):
                lin(f"{tb}.{a}.to_q", c, c, False); lin(f"{tb}.{a}.to_k", c, kd, False); lin(f"{tb}.{a}.to_v", c, kd, False)
                lin(f"{tb}.{a}.to_out.0", c, c)
            lin(f"{tb}.ff.net.0.proj", 8 * c, c); lin(f"{tb}.ff.net.2", c, 4 * c)
        if self.kind == "unet":
            norm("conv_norm_out", boc[0]); conv("conv_out", self.out_channels, boc[0], 3)
        for pre, c in self._zero_conv_specs():
            conv(pre, c, c, 1)
        if self.kind == "controlnet":
            ce, ch = "controlnet_cond_embedding", self.cond_embed_channels
            conv(f"{ce}.conv_in", ch[0], self.conditioning_channels, 3)
            for i in range(len(ch) - 1):
                conv(f"{ce}.blocks.{2 * i}", ch[i], ch[i], 3)
                conv(f"{ce}.blocks.{2 * i + 1}", ch[i + 1], ch[i], 3)
            conv(f"{ce}.conv_out", boc[0], ch[-1], 3)
        return sp

    def synthetic_state_dict(self, device="cpu", seed: int = 0, meta: bool = False) -> Dict[str, torch.Tensor]:
        """Random-init weights of this architecture (there is no network for checkpoints): fan-in scaled normal for
        matrices, (1, 0) for norm affine, N(0, 0.02) for the zero-convs so that routing bugs cannot hide."""
        sp = self.state_dict_spec()
        if meta:
            return {k: torch.empty(v, device="meta") for k, v in sp.items()}
        g = torch.Generator(device=device).manual_seed(seed)
        zc = tuple(p for p, _ in self._zero_conv_specs()) + ("controlnet_cond_embedding.conv_out",)
        sd = {}
        for k, shp in sp.items():
            if k.endswith(".weight") and len(shp) == 1:
                sd[k] = torch.ones(shp, device=device)
            elif len(shp) == 1:
                sd[k] = torch.randn(shp, generator=g, device=device) * 0.02
                if ".norm" in k or k.startswith("conv_norm_out"):
                    sd[k] = torch.zeros(shp, device=device)
            else:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
                std = 0.02 if k.startswith(zc) else (1.0 / fan_in) ** 0.5
                sd[k] = torch.randn(shp, generator=g, device=device) * std
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device, materialize: bool = True):
        """sd: diffusers-format state dict (fp32/any float, CPU or meta).  Packs into kernel layouts on `device`."""
        pk = ParamPack()
        bf, f32 = self.dtype, torch.float32      # `bf`: the 16-bit storage format of matrix weights (bf16 or fp16)

        def W(k):
            return sd[k].float() if sd[k].device.type != "meta" else sd[k]

        conv_in = "conv_in_condition" if self.kind == "brushnet" else "conv_in"
        pk.add("conv_in.weight", _conv_igemm_cpad(W(conv_in + ".weight"), self.cin_pad), bf)
        pk.add("conv_in.bias", W(conv_in + ".bias"), f32)
        for n in ("linear_1", "linear_2"):
            pk.add(f"time_embedding.{n}.weight", W(f"time_embedding.{n}.weight"), bf)
            pk.add(f"time_embedding.{n}.bias", W(f"time_embedding.{n}.bias"), f32)
        # resnets
        tw, tb, off = [], [], 0
        for pre, cin, cout in self._resnet_specs():
            for nrm in ("norm1", "norm2"):
                pk.add(f"{pre}.{nrm}.weight", W(f"{pre}.{nrm}.weight"), f32)
                pk.add(f"{pre}.{nrm}.bias", W(f"{pre}.{nrm}.bias"), f32)
                # (gamma, beta) interleaved per channel: what the fused norm -> SiLU -> conv loader DMAs per 64-channel chunk
                pk.add(f"{pre}.{nrm}.gb", torch.stack([W(f"{pre}.{nrm}.weight"), W(f"{pre}.{nrm}.bias")], 1), f32)
            pk.add(f"{pre}.conv1.weight", _conv_igemm(W(f"{pre}.conv1.weight")), bf)
            pk.add(f"{pre}.conv1.bias", W(f"{pre}.conv1.bias"), f32)
            if cin != cout and self.merge_shortcut:
                # conv2(h) + conv_shortcut(x) = one implicit GEMM: the 1x1 shortcut is a K tail over the block input
                pk.add(f"{pre}.conv2.weight", torch.cat([_conv_igemm(W(f"{pre}.conv2.weight")),
                                                          W(f"{pre}.conv_shortcut.weight").reshape(cout, cin)], 1), bf)
                pk.add(f"{pre}.conv2.bias", W(f"{pre}.conv2.bias") + W(f"{pre}.conv_shortcut.bias"), f32)
            else:
                pk.add(f"{pre}.conv2.weight", _conv_igemm(W(f"{pre}.conv2.weight")), bf)
                pk.add(f"{pre}.conv2.bias", W(f"{pre}.conv2.bias"), f32)
                if cin != cout:
                    pk.add(f"{pre}.conv_shortcut.weight", W(f"{pre}.conv_shortcut.weight").reshape(cout, cin), bf)
                    pk.add(f"{pre}.conv_shortcut.bias", W(f"{pre}.conv_shortcut.bias"), f32)
            tw.append(W(f"{pre}.time_emb_proj.weight"))
            tb.append(W(f"{pre}.time_emb_proj.bias"))
            self.temb_off[pre] = off
            off += cout
        self.temb_total = off
        pk.add("temb_all.weight", torch.cat(tw, 0), bf)
        pk.add("temb_all.bias", torch.cat(tb, 0), f32)
        # samplers
        for i in range(len(self.boc) - 1):
            pre = f"down_blocks.{i}.downsamplers.0.conv"
            pk.add(pre + ".weight", _conv_igemm(W(pre + ".weight")), bf)
            pk.add(pre + ".bias", W(pre + ".bias"), f32)
            if self.kind != "controlnet":
                pre = f"up_blocks.{i}.upsamplers.0.conv"
                pk.add(pre + ".weight", _conv_igemm(W(pre + ".weight")), bf)
                pk.add(pre + ".bias", W(pre + ".bias"), f32)
        # transformers
        for pre, c in self._attn_specs():
            pk.add(f"{pre}.norm.weight", W(f"{pre}.norm.weight"), f32)
            pk.add(f"{pre}.norm.bias", W(f"{pre}.norm.bias"), f32)
            pk.add(f"{pre}.proj_in.weight", W(f"{pre}.proj_in.weight").reshape(c, c), bf)
            pk.add(f"{pre}.proj_in.bias", W(f"{pre}.proj_in.bias"), f32)
            w_po, b_po = W(f"{pre}.proj_out.weight").reshape(c, c), W(f"{pre}.proj_out.bias")
            if self.merge_ff2_proj_out:
                # FF2 and proj_out are two linear maps with only a residual add between them:
                #   proj_out(FF2(g) + hs) = [g | hs] [W_po W_ff2 | W_po]^T + (W_po b_ff2 + b_po)
                # -> ONE GEMM over the K-concatenation of g and hs (same FLOPs, one launch and one hidden-state round
                # trip less per transformer); composed in fp32 at pack time.
                w_f2, b_f2 = W(f"{pre}.transformer_blocks.0.ff.net.2.weight"), W(f"{pre}.transformer_blocks.0.ff.net.2.bias")
                pk.add(f"{pre}.ff2_proj_out.weight", torch.cat([w_po @ w_f2, w_po], 1), bf)
                pk.add(f"{pre}.ff2_proj_out.bias", w_po @ b_f2 + b_po, f32)
                if c == 320 and self.fuse_ff and self.fold_ln and not self.ff_w8:
                    # the same matrix with its hidden index permuted: second GEMM of the 4-wave fused feed-forward
                    pk.add(f"{pre}.ff2_proj_out.weight_kp", torch.cat([_kperm_geglu(w_po @ w_f2), w_po], 1), bf)
            else:
                pk.add(f"{pre}.proj_out.weight", w_po, bf)
                pk.add(f"{pre}.proj_out.bias", b_po, f32)
            tb_ = f"{pre}.transformer_blocks.0"
            wqkv = torch.cat([W(f"{tb_}.attn1.to_q.weight"), W(f"{tb_}.attn1.to_k.weight"),
                              W(f"{tb_}.attn1.to_v.weight")], 0)
            wff1, bff1 = W(f"{tb_}.ff.net.0.proj.weight"), W(f"{tb_}.ff.net.0.proj.bias")
            if self.fold_ln:
                # LayerNorm folded into the Linear that consumes it:  LN(x) W^T = rstd (x (g.W)^T - mean colsum) + W b
                def fold(name, w, nrm, bias=None, il=False):
                    g_, b_ = W(f"{tb_}.{nrm}.weight"), W(f"{tb_}.{nrm}.bias")
                    wf = w * g_[None, :]
                    cs = wf.to(bf).float().sum(1)          # of the weights as the MFMA sees them
                    t = w @ b_ if bias is None else w @ b_ + bias
                    if il:
                        wf, cs, t = _geglu_interleave(wf), _geglu_interleave(cs), _geglu_interleave(t)
                    pk.add(f"{name}.weight", wf, bf)
                    pk.add(f"{name}.colsum", cs, f32)
                    pk.add(f"{name}.bias", t, f32)

                fold(f"{tb_}.attn1.qkv", wqkv, "norm1")
                if wqkv.shape[1] == 320 and self.fuse_tfront:
                    # the same folded weight with its input index permuted: second GEMM of the fused front end (csrc/tfront.hip)
                    g_ = W(f"{tb_}.norm1.weight")
                    pk.add(f"{tb_}.attn1.qkv.weight_kp", _kperm(wqkv * g_[None, :]), bf)
                fold(f"{tb_}.attn2.to_q", W(f"{tb_}.attn2.to_q.weight"), "norm2")
                fold(f"{tb_}.ff1", wff1, "norm3", bff1, il=True)
            else:
                for nrm in ("norm1", "norm2", "norm3"):
                    pk.add(f"{tb_}.{nrm}.weight", W(f"{tb_}.{nrm}.weight"), f32)
                    pk.add(f"{tb_}.{nrm}.bias", W(f"{tb_}.{nrm}.bias"), f32)
                pk.add(f"{tb_}.attn1.qkv.weight", wqkv, bf)
                pk.add(f"{tb_}.attn2.to_q.weight", W(f"{tb_}.attn2.to_q.weight"), bf)
                pk.add(f"{tb_}.ff1.weight", _geglu_interleave(wff1), bf)
                pk.add(f"{tb_}.ff1.bias", _geglu_interleave(bff1), f32)
            pk.add(f"{tb_}.attn1.to_out.weight", W(f"{tb_}.attn1.to_out.0.weight"), bf)
            pk.add(f"{tb_}.attn1.to_out.bias", W(f"{tb_}.attn1.to_out.0.bias"), f32)
            pk.add(f"{tb_}.attn2.kv.weight", torch.cat([W(f"{tb_}.attn2.to_k.weight"), W(f"{tb_}.attn2.to_v.weight")], 0), bf)
            pk.add(f"{tb_}.attn2.to_out.weight", W(f"{tb_}.attn2.to_out.0.weight"), bf)
            pk.add(f"{tb_}.attn2.to_out.bias", W(f"{tb_}.attn2.to_out.0.bias"), f32)
            if not self.merge_ff2_proj_out:
                pk.add(f"{tb_}.ff2.weight", W(f"{tb_}.ff.net.2.weight"), bf)
                pk.add(f"{tb_}.ff2.bias", W(f"{tb_}.ff.net.2.bias"), f32)
        if self.kind == "unet":
            pk.add("conv_norm_out.weight", W("conv_norm_out.weight"), f32)
            pk.add("conv_norm_out.bias", W("conv_norm_out.bias"), f32)
            pk.add("conv_out.weight", _conv_igemm(W("conv_out.weight")), bf)
            pk.add("conv_out.bias", W("conv_out.bias"), f32)
        for pre, c in self._zero_conv_specs():
            pk.add(pre + ".weight", W(pre + ".weight").reshape(c, c), bf)
            pk.add(pre + ".bias", W(pre + ".bias"), f32)
        if self.kind == "controlnet":
            ce = "controlnet_cond_embedding"
            names = ["conv_in"] + [f"blocks.{i}" for i in range(2 * (len(self.cond_embed_channels) - 1))] + ["conv_out"]
            for n in names:
                pk.add(f"{ce}.{n}.weight", _conv_direct(W(f"{ce}.{n}.weight")), bf)
                pk.add(f"{ce}.{n}.bias", W(f"{ce}.{n}.bias"), f32)
        pk.to_device(device, materialize)
        self.params, self.P = pk, pk.ptr
        return self

    # ---------------------------------------------------------------- plan pieces
    def _resnet(self, pb: Builder, pre: str, x: Act, cout: int, temb_all: int, x2: Optional[Act] = None,
                res2: int = 0) -> Act:
        P = self.P
        cin = x.C + (x2.C if x2 is not None else 0)
        out = pb.new_act(x.B, x.H, x.W, cout)
        m = pb.mark()
        def gn(nrm):
            return (P[f"{pre}.{nrm}.weight"], P[f"{pre}.{nrm}.bias"], P[f"{pre}.{nrm}.gb"], self.eps, self.groups)

        h = pb.conv3x3(x, P[f"{pre}.conv1.weight"], cout, P[f"{pre}.conv1.bias"], x2=x2,
                       rowvec=temb_all + 4 * self.temb_off[pre], name="conv3x3", gn_in=gn("norm1"))
        if cin != cout and self.merge_shortcut:
            pb.conv3x3(h, P[f"{pre}.conv2.weight"], cout, P[f"{pre}.conv2.bias"], res2=res2, out=out, x3=x, x4=x2,
                       name="conv3x3", gn_in=gn("norm2"))
            pb.release(m)
            return out
        if cin != cout:
            sc = pb.linear(x.ptr, x.rows, x.C, P[f"{pre}.conv_shortcut.weight"], cout, P[f"{pre}.conv_shortcut.bias"],
                           x2=(x2.ptr if x2 is not None else 0), K2=(x2.C if x2 is not None else 0), name="conv1x1")
        else:
            assert x2 is None
            sc = x.ptr
        pb.conv3x3(h, P[f"{pre}.conv2.weight"], cout, P[f"{pre}.conv2.bias"], res1=sc, res2=res2, out=out,
                   name="conv3x3", gn_in=gn("norm2"))
        pb.release(m)
        return out

    def _transformer(self, pb: Builder, pre: str, x: Act, kv: Tuple[int, int, int, int], res2: int = 0,
                     twin: bool = False) -> Act:
        """kv = (k_ptr, ldk, vt_ptr, ldvt) of the hoisted cross-attention K / V^T; nk = ctx tokens in self._nctx.
        twin: x is ONE half of a CFG pair whose halves are identical up to here (build_step): norm / proj_in / attn1 run on
        that half, the fused cross-attention block -- the first consumer that mixes the prompt in -- reads it with batch-wrap
        addressing, and everything behind it (and the returned Act) is the full batch."""
        P = self.P
        Cc, rows, hw = x.C, x.rows, x.H * x.W
        d = Cc // self.heads
        tb = f"{pre}.transformer_blocks.0"
        Bo = 2 * x.B if twin else x.B          # batch / rows behind the first cross-attention
        rows_o = Bo * hw
        out = pb.new_act(Bo, x.H, x.W, Cc)
        m = pb.mark()
        fold = self.fold_ln
        tiles = (Cc + 159) // 160

        def producer(r=None):     # row-moment buffer written by the GEMM that produces the next LayerNorm's input
            return pb.alloc((r or rows) * tiles * 8) if fold else 0

        def normed(h, st, nrm, lin):
            """-> (x pointer, kwargs) for the Linear `lin` applied to LayerNorm_nrm(h)."""
            if fold:
                return h, dict(bias=P[f"{tb}.{lin}.bias"], ln_stats=st, ln_colsum=P[f"{tb}.{lin}.colsum"],
                               ln_tiles=tiles, ln_dim=Cc, ln_eps=1e-5)
            xn = pb.layernorm(h, rows, Cc, P[f"{tb}.{nrm}.weight"], P[f"{tb}.{nrm}.bias"])
            return xn, (dict(bias=P[f"{tb}.{lin}.bias"]) if lin == "ff1" else {})

        front = None
        if fold and self.fuse_tfront and f"{tb}.attn1.qkv.weight_kp" in P and \
                pb.lib.pp_tfront_supported(rows, Cc, hw, self.groups):
            acc = pb._subscribe_gn_stats(x, None, self.groups)
            if acc:
                hs = pb.alloc(rows * Cc * 2)
                vt = pb.alloc(x.B * Cc * hw * 2)
                qk = pb.alloc(rows * 2 * Cc * 2)
                # (round 5) Q leaves the front end as Q * d^-0.5 * log2(e) where the pipelined attention kernel covers the
                # shape: its scores then come out of the MFMA as exp2 arguments (PP_ATTN_PIPE_LOG2)
                log2q = bool(self.attn_log2 and pb.lib.pp_attention_log2_ok(hw, hw, d))
                pb.plan.add("tfront", pb.lib.pp_tfront, x.ptr, Cc, acc, P[f"{pre}.norm.weight"], P[f"{pre}.norm.bias"], 1e-6,
                            self.groups, P[f"{pre}.proj_in.weight"], P[f"{pre}.proj_in.bias"],
                            P[f"{tb}.attn1.qkv.weight_kp"], P[f"{tb}.attn1.qkv.colsum"], P[f"{tb}.attn1.qkv.bias"], 1e-5, hs, Cc,
                            qk, 2 * Cc, vt, hw, rows, Cc, hw, (float(d) ** -0.5) * 1.4426950408889634 if log2q else 1.0, pb.dt)
                pb.plan.count("tfront", 2.0 * rows * Cc * Cc + 2.0 * rows * 3 * Cc * Cc)     # (proj_in + QKV)
                front = pb.attention(qk, 2 * Cc, qk + 2 * Cc, 2 * Cc, vt, hw, x.B, self.heads, hw, hw, d, log2q=log2q)
        if front is not None:
            a = front
        else:
            st = producer()
            n = pb.groupnorm(x, P[f"{pre}.norm.weight"], P[f"{pre}.norm.bias"], 1e-6, False, groups=self.groups)
            hs = pb.linear(n.ptr, rows, Cc, P[f"{pre}.proj_in.weight"], Cc, P[f"{pre}.proj_in.bias"], row_stats_out=st,
                           name="conv1x1")
            # self-attention: fused QKV GEMM, V written transposed by the epilogue
            ln, kw = normed(hs, st, "norm1", "attn1.qkv")
        if front is not None:
            pass
        elif hw % 8 == 0:
            vt = pb.alloc(x.B * Cc * hw * 2)
            qk = pb.linear(ln, rows, Cc, P[f"{tb}.attn1.qkv.weight"], 3 * Cc, out_vt=vt, vt_col0=2 * Cc, vt_ld=hw,
                           rows_per_batch=hw, name="linear", **kw)
            a = pb.attention(qk, 2 * Cc, qk + 2 * Cc, 2 * Cc, vt, hw, x.B, self.heads, hw, hw, d)
        else:
            # tiny latents (hw < 8, e.g. 2x2): V^T rows must be 16-byte aligned and zero padded -> unfused transpose
            ldvt = _align(hw, 8)
            vt = pb.alloc(x.B * Cc * ldvt * 2)
            qkv = pb.linear(ln, rows, Cc, P[f"{tb}.attn1.qkv.weight"], 3 * Cc, name="linear", **kw)
            pb.plan.add("transpose_v", pb.lib.pp_transpose_v, qkv + 4 * Cc, 3 * Cc, x.B, hw, Cc, vt, ldvt)
            a = pb.attention(qkv, 3 * Cc, qkv + 2 * Cc, 3 * Cc, vt, ldvt, x.B, self.heads, hw, hw, d)
        xa = getattr(self, "xa", {}).get(pre)
        if xa is not None and len(xa) > 4 and xa[4] == 1 and fold and \
                pb.lib.pp_xattn_block_supported(rows_o, Cc, hw, self._nctx, self.heads):
            # attn1.to_out + residual in front of the cross-attention sub-block, same launch (G^T was folded with kperm = 1)
            st2 = producer(rows_o)
            o = pb.alloc(rows_o * Cc * 2)
            pb.plan.add("xattn_block", pb.lib.pp_xattn_block, a, Cc, hs, Cc, None, tiles, 1e-5, xa[0], xa[1], xa[2], xa[3],
                        P[f"{tb}.attn2.to_out.bias"], o, Cc, st2 or None, rows_o, Cc, hw, rows if twin else 0,
                        P[f"{tb}.attn1.to_out.weight"], P[f"{tb}.attn1.to_out.bias"], pb.dt)
            pb.plan.count("xattn_block", 6.0 * rows_o * Cc * Cc + 4.0 * rows_o * self._nctx * Cc)   # (+ attn1.to_out)
            hs, st = o, st2
            xa = "done"
        else:
            st = producer()
            hs = pb.linear(a, rows, Cc, P[f"{tb}.attn1.to_out.weight"], Cc, P[f"{tb}.attn1.to_out.bias"], res1=hs,
                           row_stats_out=st, name="linear")
        # cross-attention (K / V^T hoisted out of the step: encoder_hidden_states are step-invariant)
        if xa == "done":
            pass
        elif xa is not None and len(xa) > 4 and xa[4] == 2 and hw % 64 == 0 and not twin:
            # the folded sub-block as two GEMMs with one (G^T, H^T) per batch item (setup plan: pp_xattn_fold(kperm = 2))
            S = self.heads * 80
            ln, kw = normed(hs, st, "norm2", "attn2.to_q")
            stats = dict(ln_stats=kw["ln_stats"], ln_colsum=xa[1], ln_tiles=tiles, ln_dim=Cc, ln_eps=1e-5) if fold else {}
            prob = pb.linear(ln, rows, Cc, xa[0], S, bias=xa[2], act=L.PP_ACT_SOFTMAX80, rows_per_batch=hw,
                             w_batch_stride=S * Cc, vec_batch_stride=S, name="linear", **stats)
            st = producer()
            hs = pb.linear(prob, rows, S, xa[3], Cc, P[f"{tb}.attn2.to_out.bias"], res1=hs, row_stats_out=st,
                           rows_per_batch=hw, w_batch_stride=Cc * S, name="linear")
        elif xa is not None and len(xa) > 4 and xa[4] == 2:
            raise L.PPError("H^T was folded for the two-GEMM cross-attention, which this shape does not take (build_setup "
                            "folds it only for step geometries with whole 64-row tiles per batch item)")
        elif xa is not None and len(xa) > 4 and xa[4] == 1:
            raise L.PPError("G^T was folded for the chained cross-attention block, which this shape does not take")
        elif xa is not None and pb.lib.pp_xattn_block_supported(rows_o, Cc, hw, self._nctx, self.heads):
            ln, kw = normed(hs, st, "norm2", "attn2.to_q")
            st2 = producer(rows_o)
            o = pb.alloc(rows_o * Cc * 2)
            pb.plan.add("xattn_block", pb.lib.pp_xattn_block, ln, Cc, hs, Cc, kw.get("ln_stats"), tiles if fold else 0,
                        1e-5, xa[0], xa[1], xa[2], xa[3], P[f"{tb}.attn2.to_out.bias"], o, Cc, st2 or None, rows_o, Cc, hw,
                        rows if twin else 0, None, None, pb.dt)
            # (the FLOPs of the chain it stands for: the folded form multiplies twice as much)
            pb.plan.count("xattn_block", 4.0 * rows_o * Cc * Cc + 4.0 * rows_o * self._nctx * Cc)
            hs, st = o, st2
        elif twin:
            raise L.PPError("twin prefix without the fused cross-attention block (build_step checks _twin_ok first)")
        else:
            ln, kw = normed(hs, st, "norm2", "attn2.to_q")
            q = pb.linear(ln, rows, Cc, P[f"{tb}.attn2.to_q.weight"], Cc, name="linear", **kw)
            a = pb.attention(q, Cc, kv[0], kv[1], kv[2], kv[3], x.B, self.heads, hw, self._nctx, d)
            st = producer()
            hs = pb.linear(a, rows, Cc, P[f"{tb}.attn2.to_out.weight"], Cc, P[f"{tb}.attn2.to_out.bias"], res1=hs,
                           row_stats_out=st, name="linear")
        # feed-forward: GEGLU fused into the first GEMM's epilogue
        rows_h, rows = rows, rows_o           # (from here on: the full batch)
        wrap = rows_h if twin else 0          # the transformer's input (the proj_out residual) holds one half only
        w2n = f"{pre}.ff2_proj_out.weight" + ("" if self.ff_w8 else "_kp")
        if fold and self.fuse_ff and self.merge_ff2_proj_out and w2n in P and pb.lib.pp_ff_fused_supported(rows, Cc, hw):
            out.producer = pb.ff_fused(hs, rows, Cc, hw, P[f"{tb}.ff1.weight"], P[f"{tb}.ff1.bias"], P[f"{tb}.ff1.colsum"],
                                       st, tiles, P[w2n], P[f"{pre}.ff2_proj_out.bias"], out.ptr,
                                       res1=x.ptr, res1_wrap=wrap, res2=res2, w2_kperm=not self.ff_w8)
            pb.release(m)
            return out
        ln, kw = normed(hs, st, "norm3", "ff1")
        g = pb.linear(ln, rows, Cc, P[f"{tb}.ff1.weight"], 8 * Cc, act=L.PP_ACT_GEGLU, name="linear_geglu", **kw)
        if self.merge_ff2_proj_out:
            pb.linear(g, rows, 4 * Cc, P[f"{pre}.ff2_proj_out.weight"], Cc, P[f"{pre}.ff2_proj_out.bias"], x2=hs, K2=Cc,
                      ldx2=Cc, res1=x.ptr, res1_wrap=wrap, res2=res2, out=out.ptr, name="linear")
            out.producer = pb.last_gemm
        else:
            hs = pb.linear(g, rows, 4 * Cc, P[f"{tb}.ff2.weight"], Cc, P[f"{tb}.ff2.bias"], res1=hs, name="linear")
            pb.linear(hs, rows, Cc, P[f"{pre}.proj_out.weight"], Cc, P[f"{pre}.proj_out.bias"], res1=x.ptr, res1_wrap=wrap,
                      res2=res2, out=out.ptr, name="conv1x1")
            out.producer = pb.last_gemm
        pb.release(m)
        return out

    # ---------------------------------------------------------------- setup plan: step-invariant work
    def build_setup(self, pb: Builder, B: int, nctx: int, ehs: int, cond: Optional[Act] = None,
                    hw0: Optional[Tuple[int, int]] = None):
        """Cross-attention K and V^T for every transformer from encoder_hidden_states (bf16 [B*nctx][ctx_dim] at
        `ehs`); for ControlNet also the conditioning embedding of `cond` (NHWC bf16 image).  hw0 = (H, W) of the latents the
        step plan will be built for (the folded cross-attention operands of a transformer depend on which kernel runs it)."""
        self._nctx = nctx
        ldvt = _align(nctx, 8)
        pre_hw: Dict[str, bool] = {}        # transformer -> the fused block kernel takes its step geometry
        pre_hw64: Dict[str, bool] = {}      # transformer -> whole 64-row GEMM tiles per batch item (the two-GEMM form)
        if hw0 is not None:
            nl = len(self.boc)
            for pre, c in self._attn_specs():
                part = pre.split(".")
                lvl = int(part[1]) if part[0] == "down_blocks" else (nl - 1 if part[0] == "mid_block" else nl - 1 - int(part[1]))
                hw = (hw0[0] >> lvl) * (hw0[1] >> lvl)
                pre_hw[pre] = hw > 0 and bool(pb.lib.pp_xattn_block_supported(B * hw, c, hw, nctx, self.heads))
                pre_hw64[pre] = hw > 0 and hw % 64 == 0
        self.kv: Dict[str, Tuple[int, int, int, int]] = {}
        self.xa: Dict[str, Tuple[int, int, int, int]] = {}       # folded cross-attention operands (pp_xattn_fold)
        for pre, c in self._attn_specs():
            tb = f"{pre}.transformer_blocks.0"
            vt = pb.alloc(B * c * ldvt * 2)
            k = pb.linear(ehs, B * nctx, self.ctx_dim, self.P[f"{tb}.attn2.kv.weight"], 2 * c, out_vt=vt,
                          vt_col0=c, vt_ld=ldvt, rows_per_batch=nctx, name="linear")
            self.kv[pre] = (k, c, vt, ldvt)
            tbq = f"{tb}.attn2.to_q"
            two_gemm = bool(self.fuse_xattn and self.fuse_xattn_2g and c > 320 and pre_hw64.get(pre) and
                            not (self.fuse_xattn_wide and c <= self.xattn_wide_max_c))
            # (no folded operands for a step geometry none of the folded forms takes -- they were dead weight
            #  in the arena and in the setup plan)
            block_ok = bool(pre_hw.get(pre)) if hw0 is not None else True
            if self.fuse_xattn and (two_gemm or (block_ok and (c == 320 or (self.fuse_xattn_wide and c <= self.xattn_wide_max_c)))) and \
                    pb.lib.pp_xattn_block_supported(128, c, 128, nctx, self.heads):
                S = self.heads * 80
                gt, ht = pb.alloc(B * S * c * 2), pb.alloc(B * c * S * 2)
                gcs, gb = pb.alloc(B * S * 4), pb.alloc(B * S * 4)
                fold = self.fold_ln
                # C = 320: attn1.to_out runs in front of the sub-block in the same launch, whose logits then take their B
                # operand from that GEMM's accumulators -> G^T with its channel index permuted (pre_hw: the step geometry
                # must be one the block kernel takes, else the step plan keeps the chain and the plain layout)
                kperm = 2 if two_gemm else int(bool(c == 320 and fold and self.fuse_xattn_pre and pre_hw.get(pre)))
                pb.plan.add("xattn_fold", pb.lib.pp_xattn_fold, k, c, vt, ldvt, B, nctx, self.heads, c,
                            self.P[f"{tbq}.weight"], self.P[f"{tbq}.colsum"] if fold else None,
                            self.P[f"{tbq}.bias"] if fold else None, self.P[f"{tb}.attn2.to_out.weight"],
                            float(c // self.heads) ** -0.5, gt, gcs, gb, ht, int(kperm), pb.dt)
                self.xa[pre] = (gt, gcs, gb, ht, kperm)
        self.cond_emb = None
        if self.kind == "controlnet":
            assert cond is not None
            ce = "controlnet_cond_embedding"
            chans = self.cond_embed_channels
            lib = pb.lib

            def dconv(x: Act, name: str, cout: int, stride: int, silu: int) -> Act:
                ho, wo = (x.H - 1) // stride + 1, (x.W - 1) // stride + 1
                o = pb.new_act(x.B, ho, wo, cout)
                pb.plan.add("conv3x3_direct", lib.pp_conv3x3_direct, x.ptr, x.B, x.H, x.W, x.C,
                            self.P[f"{ce}.{name}.weight"], self.P[f"{ce}.{name}.bias"], cout, stride, silu, None, o.ptr,
                            pb.dt)
                return o

            e = dconv(cond, "conv_in", chans[0], 1, 1)
            for i in range(len(chans) - 1):
                e = dconv(e, f"blocks.{2 * i}", chans[i], 1, 1)
                e = dconv(e, f"blocks.{2 * i + 1}", chans[i + 1], 2, 1)
            self.cond_emb = dconv(e, "conv_out", self.boc[0], 1, 0)

    # ---------------------------------------------------------------- step plan
    def twin_prefix_ok(self, lib, B: int, H: int, W: int, nctx: int, brush: bool = False) -> bool:
        """Can build_step(twin=True) run the CFG-identical prefix of this network on one half of the batch?  Needs the
        fused cross-attention block at the first level (it is the launch that reads the half batch with wrap addressing)
        and an input that nothing prompt-dependent touches before it: no BrushNet adds inside the down path."""
        pre = "down_blocks.0.attentions.0"
        return (TWIN_PREFIX and not brush and B >= 2 and B % 2 == 0 and self.L >= 1 and self.boc[0] == 320 and
                self.down_types[0] == "CrossAttnDownBlock2D" and pre in getattr(self, "xa", {}) and (H * W) % 128 == 0 and
                bool(lib.pp_xattn_block_supported(B * H * W, 320, H * W, nctx, self.heads)))

    def build_step(self, pb: Builder, x_in: Act, t_dev: int, add_down: Optional[List[int]] = None,
                   add_mid: int = 0, add_up: Optional[List[int]] = None, ctrl_down: Optional[List[int]] = None,
                   ctrl_mid: int = 0, scale: float = 1.0, pad_uncond: bool = False, twin: bool = False) -> Dict[str, object]:
        """Append one forward pass.  x_in: NHWC bf16 input (already channel-concatenated).  Returns outputs:
        unet -> {"eps": ptr fp32 NCHW}; brushnet -> {"down": [Act], "mid": Act, "up": [Act]}; controlnet likewise.
        pad_uncond (side networks): the pipelines' guess mode runs the side branch on the conditional half of a CFG pair
        only and hands the UNet `cat([zeros_like(d), d])` (pipeline_PowerPaint_Brushnet_CA.py:1421-1425,
        pipeline_PowerPaint_ControlNet.py:1697-1702): the residuals are written into the second half of tensors with
        twice the batch whose first half one launch per step zeroes.
        twin: the caller GUARANTEES that batch items [B/2, B) of x_in (and of the ControlNet conditioning) equal items
        [0, B/2) -- the loop builds the pair as `torch.cat([latents] * 2)` (pipeline_PowerPaint.py:990-996) over CFG-duplicated
        mask / masked-image latents.  The two halves of the forward pass are then bit-identical until the prompt enters at the
        first cross-attention (unet_2d_condition.py:1183-1236): conv_in, down_blocks.0.resnets.0 and the first transformer's
        norm / proj_in / self-attention run on ONE half (conv_in stores its rows twice -- the skip tensor's consumers in the
        up path see the full batch -- and the fused cross-attention block and the transformer's last GEMM read the half
        batch with wrap addressing).  Where twin_prefix_ok() says no, the flag changes nothing."""
        P, lib = self.P, pb.lib
        B, H, W = x_in.B, x_in.H, x_in.W
        boc = self.boc
        brush = add_down is not None
        add_down = list(add_down) if brush else None
        add_up = list(add_up) if brush else None
        # (a forced tile / split-K factor -- NetRuntime.gemm_tile / gemm_splitk, a measurement knob -- may send conv_in off
        #  the single-pass staged epilogue that stores the twin copy: the flag then changes nothing, as everywhere it cannot run)
        twin = bool(twin) and not pad_uncond and not (pb.gemm_tile or pb.gemm_splitk) and \
            self.twin_prefix_ok(lib, B, H, W, self._nctx, brush)

        def pop(lst):
            return lst.pop(0) if lst is not None else 0

        # 1. time embedding (identical for every batch row: timesteps.expand(B))
        tsin = pb.alloc(boc[0] * 4)
        t1 = pb.alloc(boc[0] * 4 * 4)
        temb = pb.alloc(boc[0] * 4 * 4)
        temb_all = pb.alloc(self.temb_total * 4)
        te = boc[0] * 4
        pb.plan.add("timestep_embedding", lib.pp_timestep_embedding, t_dev, 1, boc[0], tsin)
        pb.plan.add("linear_skinny", lib.pp_linear_skinny, tsin, 1, boc[0], P["time_embedding.linear_1.weight"],
                    P["time_embedding.linear_1.bias"], te, t1, te, 0, L.PP_ACT_SILU, pb.dt)
        pb.plan.add("linear_skinny", lib.pp_linear_skinny, t1, 1, te, P["time_embedding.linear_2.weight"],
                    P["time_embedding.linear_2.bias"], te, temb, te, 0, 0, pb.dt)
        pb.plan.add("linear_skinny", lib.pp_linear_skinny, temb, 1, te, P["temb_all.weight"], P["temb_all.bias"],
                    self.temb_total, temb_all, self.temb_total, L.PP_ACT_SILU, 0, pb.dt)

        # 2. conv_in: the implicit-GEMM conv over the zero-padded input (4 / 9 real channels in one 64-deep K chunk per
        # tap: 6x the algorithmic MACs, still 4x faster than the scalar direct conv at 64x64 -- 9 short K steps on the
        # matrix cores -- and its epilogue carries the GroupNorm statistics of resnets.0.norm1 and the side-branch add)
        if x_in.C != self.cin_pad:
            raise L.PPError(f"network input buffer has {x_in.C} channels, conv_in expects {self.cin_pad} (zero-padded)")

        def conv_in(add_ptr) -> Act:
            x = Act(x_in.ptr, B // 2, H, W, x_in.C) if twin else x_in
            o = pb.conv3x3(x, P["conv_in.weight"], boc[0], P["conv_in.bias"], res2=add_ptr or 0, name="conv3x3", dup=twin)
            pb.plan.count("conv3x3", -2.0 * x.B * H * W * boc[0] * 9 * (x_in.C - self.cin0))   # (count the real MACs)
            return o

        if self.kind == "controlnet":
            s = conv_in(self.cond_emb.ptr)
            skips = [s]
        else:
            s = conv_in(None)
            skips = [s]                       # captured BEFORE the BrushNet add (unet_2d_condition.py:1220-1223)
            if brush:
                s = conv_in(pop(add_down))
        if twin:                              # the half the prefix goes on with (skips[0] stays the full, twice-stored tensor)
            s = Act(s.ptr, B // 2, s.H, s.W, s.C, producer=s.producer)

        # 3. down
        for i, typ in enumerate(self.down_types):
            for j in range(self.L):
                pre = f"down_blocks.{i}.resnets.{j}"
                has_attn = typ == "CrossAttnDownBlock2D"
                r2 = pop(add_down)
                s = self._resnet(pb, pre, s, boc[i], temb_all, res2=0 if has_attn else r2)
                if has_attn:
                    ap = f"down_blocks.{i}.attentions.{j}"
                    s = self._transformer(pb, ap, s, self.kv[ap], res2=r2, twin=twin and i == 0 and j == 0)
                skips.append(s)
            if i != len(boc) - 1:
                pre = f"down_blocks.{i}.downsamplers.0.conv"
                s = pb.conv3x3(s, P[pre + ".weight"], boc[i], P[pre + ".bias"], stride=2, res2=pop(add_down),
                               name="conv3x3")
                skips.append(s)

        # 4. mid
        s = self._resnet(pb, "mid_block.resnets.0", s, boc[-1], temb_all)
        s = self._transformer(pb, "mid_block.attentions.0", s, self.kv["mid_block.attentions.0"])
        mid_res = add_mid if brush else (ctrl_mid if ctrl_down is not None else 0)
        s = self._resnet(pb, "mid_block.resnets.1", s, boc[-1], temb_all, res2=mid_res)

        def zero_convs(feats: List[Act]) -> List[Act]:
            specs = self._zero_conv_specs()
            mult = 2 if pad_uncond else 1
            outs = [pb.new_act(mult * f.B, f.H, f.W, c) for (_, c), f in zip(specs, feats)]
            if pad_uncond:           # one zeroing launch over the whole (contiguous) block of padded outputs
                end = outs[-1].ptr + outs[-1].rows * outs[-1].C * 2
                pb.plan.add("zero_u64", lib.pp_zero_u64, outs[0].ptr, (end - outs[0].ptr + 7) // 8)
            for (pre, c), f, o in zip(specs, feats, outs):
                pb.linear(f.ptr, f.rows, c, P[pre + ".weight"], c, P[pre + ".bias"], scale=scale,
                          out=o.ptr + (f.rows * c * 2 if pad_uncond else 0), name="zero_conv")
            return outs

        if self.kind == "controlnet":
            outs = zero_convs(skips + [s])
            return {"down": outs[:-1], "mid": outs[-1]}

        if ctrl_down is not None:  # stock-UNet ControlNet residuals on the skip tensors (unet_2d_condition.py:1263-1272)
            new = []
            for sk, cp in zip(skips, ctrl_down):
                o = pb.new_act(sk.B, sk.H, sk.W, sk.C)
                pb.add(sk.ptr, cp, o.ptr, sk.rows * sk.C)
                new.append(o)
            skips = new

        brush_down = list(skips)
        brush_mid = s
        brush_up: List[Act] = []

        # 5. up
        rev = list(reversed(boc))
        for i, typ in enumerate(self.up_types):
            has_attn = typ == "CrossAttnUpBlock2D"
            for j in range(self.L + 1):
                sk = skips.pop()
                assert sk.H == s.H and sk.W == s.W, "skip / hidden size mismatch (odd latent size?)"
                r2 = pop(add_up)
                s = self._resnet(pb, f"up_blocks.{i}.resnets.{j}", s, rev[i], temb_all, x2=sk,
                                 res2=0 if has_attn else r2)
                if has_attn:
                    ap = f"up_blocks.{i}.attentions.{j}"
                    s = self._transformer(pb, ap, s, self.kv[ap], res2=r2)
                brush_up.append(s)
            if i != len(boc) - 1:
                pre = f"up_blocks.{i}.upsamplers.0.conv"
                s = pb.conv3x3(s, P[pre + ".weight"], rev[i], P[pre + ".bias"], up=True, res2=pop(add_up),
                               name="conv3x3")
                brush_up.append(s)

        if self.kind == "brushnet":
            outs = zero_convs(brush_down + [brush_mid] + brush_up)
            nd = len(brush_down)
            return {"down": outs[:nd], "mid": outs[nd], "up": outs[nd + 1:]}

        # 6. out: conv_norm_out + SiLU + conv_out -- one launch when the GroupNorm statistics arrive from the producer's
        # epilogue (the normalised 64x64x320 activation is then never written), else norm launch(es) + conv
        eps = pb.alloc(B * self.out_channels * H * W * 4)
        acc = 0
        if self.fuse_conv_out and lib.pp_gn_conv3x3_smallcout_supported(boc[0], self.out_channels, self.groups):
            acc = pb._subscribe_gn_stats(s, None, self.groups)
        if acc:
            pb.plan.add("conv_out", lib.pp_gn_conv3x3_smallcout, s.ptr, B, H, W, boc[0], self.groups, self.eps,
                        P["conv_norm_out.weight"], P["conv_norm_out.bias"], acc, P["conv_out.weight"],
                        P["conv_out.bias"], self.out_channels, eps, pb.dt)
        else:
            h = pb.groupnorm(s, P["conv_norm_out.weight"], P["conv_norm_out.bias"], self.eps, True, groups=self.groups)
            pb.plan.add("conv_out", lib.pp_conv3x3_smallcout, h.ptr, B, H, W, boc[0], P["conv_out.weight"],
                        P["conv_out.bias"], self.out_channels, eps, pb.dt)
        pb.plan.count("conv_out", 2.0 * B * H * W * self.out_channels * 9 * boc[0])
        return {"eps": eps}
