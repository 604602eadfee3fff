"""Per-launch time of the split-K launches of the headline step with the combine in-kernel (PPGemmArgs.tile_ctr) against the
separate combine launch: hot (back to back) and cold (behind a 256 MiB memset), lean / + statistics / + consumer-norm apply.
  python tools/fc_time.py  -> table on stdout (profiles/r06_fused_combine.txt)"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from powerpaint_amd import _lib as L, ops

DEV = "cuda"
lib = L.lib()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(torch.bfloat16)


def conv_args(B, H, C1, C2, Cout, stride=1, up=False, tail=0, gn_in=False, form="lean"):
    """a PPGemmArgs + the tensors that keep its pointers alive"""
    x1 = rnd(B, H, H, C1, seed=1)
    x2 = rnd(B, H, H, C2, seed=2) if C2 else None
    ho = (2 * H if up else H) // stride
    x3 = rnd(B, ho, ho, tail, seed=3) if tail else None
    K = 9 * (C1 + C2) + tail
    w = rnd(Cout, K, seed=4, scale=K ** -0.5)
    bias = torch.zeros(Cout, device=DEV)
    rv = torch.zeros(B, Cout, device=DEV)
    res = rnd(B, ho, ho, Cout, seed=7)
    out = torch.empty(B, ho, ho, Cout, dtype=torch.bfloat16, device=DEV)
    a = L.PPGemmArgs()
    a.dtype = L.PP_DT_BF16
    a.M, a.N, a.K, a.x_mode = B * ho * ho, Cout, K, L.PP_X_CONV3X3
    a.x1, a.x2, a.c1, a.c2 = x1.data_ptr(), (x2.data_ptr() if C2 else None), C1, C2
    a.x3, a.c3 = (x3.data_ptr() if tail else None), tail
    a.batch, a.hin, a.win, a.hout, a.wout, a.stride, a.up = B, H, H, ho, ho, stride, int(up)
    a.w, a.bias, a.rowvec, a.ld_rowvec, a.rows_per_batch = w.data_ptr(), bias.data_ptr(), rv.data_ptr(), Cout, ho * ho
    a.res1, a.ldres1, a.ldres2, a.scale, a.out, a.ldo = res.data_ptr(), Cout, Cout, 1.0, out.data_ptr(), Cout
    keep = [x1, x2, x3, w, bias, rv, res, out]
    if gn_in:
        acc_in = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
        acc_in[:, :, 1] = 1 << 24
        gb = torch.ones(C1 + C2, 2, device=DEV)
        a.gn_in_acc, a.gn_in_gb, a.gn_in_groups, a.gn_in_silu, a.gn_in_eps = acc_in.data_ptr(), gb.data_ptr(), 32, 1, 1e-5
        keep += [acc_in, gb]
    if form in ("gn", "apply"):
        acc = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
        a.gn_acc[0], a.gn_cg[0], a.gn_c0[0], a.gn_groups[0] = acc.data_ptr(), Cout // 32, 0, 32
        keep.append(acc)
    if form == "apply":
        y = torch.empty_like(out)
        g2, b2 = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
        a.gn_next_out, a.gn_next_gamma, a.gn_next_beta, a.gn_next_eps, a.gn_next_silu, a.gn_next_sub = \
            y.data_ptr(), g2.data_ptr(), b2.data_ptr(), 1e-5, 1, 0
        keep += [y, g2, b2]
    ws = lib.pp_gemm_workspace_bytes(C.byref(a))
    wsb = torch.empty((max(ws, 8) + 7) // 8 * 2, dtype=torch.float32, device=DEV)
    a.workspace = wsb.data_ptr()
    a._wsb = wsb
    keep.append(wsb)
    return a, keep


def timeit(a, fused, cold, iters=30):
    n = ((a.M + 127) // 128) * ((a.N + 159) // 160) * 8       # (also where the library advises the separate combine)
    ctr = torch.zeros(n // 4, dtype=torch.int64, device=DEV)       # 16 bytes per tile
    a.tile_ctr = ctr.data_ptr() if fused else None
    a.dbg = int(os.environ.get("PP_FC_DBG", "0"), 0) if fused else 0          # (lab build: phase ablations, gemm_combine.h)
    is_fused = bool(lib.pp_gemm_combine_fused(C.byref(a)))
    s = torch.cuda.current_stream().cuda_stream
    junk = torch.empty(64 << 20, dtype=torch.float32, device=DEV) if cold else None
    for _ in range(3):
        L.check(lib.pp_gemm_bf16(C.byref(a), s), "gemm")
    tot = 0.0
    evs = []
    for _ in range(iters):
        if cold:
            junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.pp_gemm_bf16(C.byref(a), s), "gemm")
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    if fused and is_fused and os.environ.get("PP_FC_STAMPS") == "1":
        stamps(a, cold, junk)
    return ts[len(ts) // 2], is_fused


def stamps(a, cold, junk):
    """(lab build) s_memrealtime stamps of every split's thread 0 in the tail of ONE launch (gemm_combine.h, FC_STAMP):
    where the tail's time goes.  10 ns ticks -> us."""
    ws = lib.pp_gemm_workspace_bytes(C.byref(a))
    tiles = lib.pp_gemm_combine_ctr_bytes(C.byref(a)) // 16
    scr = 16 * 24 * 16 + 512
    slab_bytes = ws - tiles * scr
    splits = slab_bytes // (a.M * a.N * 4)
    a.dbg |= 0x2000
    s = torch.cuda.current_stream().cuda_stream
    if cold:
        junk.zero_()
    L.check(lib.pp_gemm_bf16(C.byref(a), s), "gemm")
    torch.cuda.synchronize()
    a.dbg &= ~0x2000
    raw = a._wsb.view(torch.int64).cpu()
    st = []
    for t in range(tiles):
        base = (slab_bytes + t * scr + 16 * 24 * 16) // 8
        for sh in range(splits):
            st.append(raw[base + sh * 8: base + sh * 8 + 8].tolist())
    st = torch.tensor(st, dtype=torch.float64) * 0.01
    t0min = st[:, 0].min()
    d = lambda i, j: float((st[:, i] - st[:, j]).mean())
    reach = lambda i: float(st[:, i].max() - t0min)
    print(f"    stamps ({'cold' if cold else 'hot'}, {tiles} tiles x {splits}): mean per split  drain {d(1, 0):5.2f}  atomic {d(2, 1):5.2f}  "
          f"wait {d(3, 2):5.2f}  inputs {d(7, 3):5.2f}  slabs {d(4, 7):5.2f}  rest {d(5, 4):5.2f}  acks {d(6, 5):5.2f} us | last split reaches: entry {reach(0):5.2f}  "
          f"drained {reach(1):5.2f}  arrived {reach(2):5.2f}  released {reach(3):5.2f}  combined {reach(5):5.2f}  acked {reach(6):5.2f} us")


CASES = [
    ("8x8 conv 1280->1280 (128 x 8)", dict(B=8, H=8, C1=1280, C2=0, Cout=1280)),
    ("8x8 conv 2560->1280 + tail (128 x 8)", dict(B=8, H=8, C1=1280, C2=1280, Cout=1280, tail=2560)),
    ("16x16 fused-norm conv 1280->1280 (256 x 4)", dict(B=8, H=16, C1=1280, C2=0, Cout=1280, gn_in=True)),
    ("16x16 fused-norm conv 1920->1280 + tail (256 x 4)", dict(B=8, H=16, C1=1280, C2=640, Cout=1280, tail=1920, gn_in=True)),
    ("16x16 fused-norm conv 640->1280 (128 x 2)", dict(B=8, H=16, C1=640, C2=0, Cout=1280, gn_in=True)),
    ("32->16 downsample conv 640 (256 x 8)", dict(B=8, H=32, C1=640, C2=0, Cout=640, stride=2)),
    ("32x32 fused-norm conv 1920->640 (256 x 2)", dict(B=8, H=32, C1=1280, C2=640, Cout=640, gn_in=True)),
]

print(f"# tools/fc_time.py on {torch.cuda.get_device_name(0)}, lib {lib.pp_build_id().decode()}: median us per C-ABI call "
      f"(kernel + separate combine, or kernel with the combine inside), 30 launches")
print(f"{'launch':52s} {'form':6s} {'hot sep':>8s} {'hot fused':>9s} {'cold sep':>9s} {'cold fused':>10s}  fused?")
for name, kw in CASES:
    if os.environ.get("PP_FC_CASES") and not any(t in name for t in os.environ["PP_FC_CASES"].split(",")):
        continue
    for form in ("lean", "gn", "apply"):
        ho = (2 * kw["H"] if kw.get("up") else kw["H"]) // kw.get("stride", 1)
        if form == "apply" and ho * ho > 256:
            continue
        a, keep = conv_args(form=form, **kw)
        if form == "apply" and not lib.pp_gemm_gn_next_ok(C.byref(a), 0):
            continue
        r = {}
        for cold in (False, True):
            for fused in (False, True):
                r[(cold, fused)] = timeit(a, fused, cold)
        print(f"{name:52s} {form:6s} {r[(False, False)][0]:8.1f} {r[(False, True)][0]:9.1f} {r[(True, False)][0]:9.1f} "
              f"{r[(True, True)][0]:10.1f}  {r[(False, True)][1]}")
