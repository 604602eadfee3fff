"""CPU: the C-ABI library loads and exports every symbol include/pp_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "pp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from powerpaint_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libpp_hip.so does not export {s}"


def test_python_binding_covers_header():
    from powerpaint_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    assert _lib.lib().pp_abi_version() == _lib.ABI_VERSION


def test_every_exported_symbol_is_called_by_the_product():
    """The public header describes the PRODUCT: every entry point it declares is called from powerpaint_amd/ (the launch-plan
    compiler, the pipelines, the tensor-level wrappers) -- rejected experiments and measurement hooks do not live in it
    (VERDICT round 3, item 6)."""
    import glob
    src = ""
    for f in glob.glob(os.path.join(ROOT, "powerpaint_amd", "**", "*.py"), recursive=True):
        if os.path.basename(f) != "_lib.py":
            src += open(f).read()
    plumbing = {"pp_abi_version", "pp_last_error", "pp_build_id"}          # used by _lib.py itself
    unused = [s for s in header_symbols() if s not in plumbing and not re.search(r"\b%s\b" % s, src)]
    assert not unused, f"declared in include/pp_hip.h but never called by powerpaint_amd/: {unused}"


def test_gemm_args_struct_layout_matches_header():
    """sizeof(PPGemmArgs) from the C compiler == ctypes.sizeof (guards against silent ABI drift)."""
    import subprocess
    import tempfile
    from powerpaint_amd import _lib
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "pp_hip.h"\nint main(){printf("%zu %zu %zu %zu",'
                           'sizeof(PPGemmArgs),offsetof(PPGemmArgs,w),offsetof(PPGemmArgs,out),offsetof(PPGemmArgs,workspace));}')
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sz, ow, oo, ows = map(int, subprocess.check_output([exe]).split())
    A = _lib.PPGemmArgs
    assert (sz, ow, oo, ows) == (ctypes.sizeof(A), A.w.offset, A.out.offset, A.workspace.offset)


def test_bad_args_are_rejected_without_a_gpu():
    from powerpaint_amd import _lib
    lib = _lib.lib()
    a = _lib.PPGemmArgs()
    assert lib.pp_gemm_bf16(ctypes.byref(a), None) == -1            # PP_ERR_BAD_ARG
    assert lib.pp_gemm_workspace_bytes(ctypes.byref(a)) == 0
    assert lib.pp_attention_fwd(None, 0, None, 0, None, 0, None, 0, 1, 8, 64, 64, 40, 1.0, 1, None) == -1
    assert lib.pp_layernorm(None, 1, 320, None, None, 1e-5, None, 1, None) == -1
    assert lib.pp_ddim_variance_noise(None, None, 16, None, None, None) == -1
    assert lib.pp_latent_blend(None, None, None, None, None, None, 1, 4, 256, None) == -1
    b = _lib.PPGemmArgs()
    b.M = b.N = b.K = 64
    b.dtype = 7                                                      # not bf16 / fp16
    assert lib.pp_gemm_bf16(ctypes.byref(b), None) == -1


def test_build_id_names_the_sources_the_library_was_built_from():
    """pp_build_id() = the digest the Makefile computes over the sources, pp_common.h and include/pp_hip.h: stable across
    rebuilds and checkout paths (bench.py and the committed rocprof summaries name builds by it) -- and a library that
    is STALE against the tree (sources edited, `make` not run) fails here instead of being measured."""
    import hashlib
    import re
    from powerpaint_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "powerpaint_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^SRCS := (.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    hdrs = re.search(r"^SRC_ID := \$\(shell cat \$\(SRCS\) (.*?) \$\(ROOT\)/include/pp_hip.h", mk, re.M).group(1).split()
    assert "pp_common.h" in hdrs
    for f in [os.path.join(csrc, s) for s in srcs + hdrs] + [os.path.join(root, "include", "pp_hip.h")]:
        h.update(open(f, "rb").read())
    bid = _lib.build_id()
    assert re.fullmatch(r"[0-9a-f]{12}", bid), bid
    assert bid == h.hexdigest()[:12], "libpp_hip.so is stale against powerpaint_amd/csrc: run `make -C powerpaint_amd/csrc`"


def _conv_args(B, H, W, c1, c2, cout, c3=0, c4=0, gn_in=True, splitk=0):
    """A PP_X_CONV3X3 request with dummy (non-null) pointers: the host-side queries only look at shapes and flags."""
    from powerpaint_amd import _lib as L
    a = L.PPGemmArgs()
    a.dtype = L.PP_DT_BF16
    a.M, a.N, a.K, a.x_mode = B * H * W, cout, 9 * (c1 + c2) + c3 + c4, L.PP_X_CONV3X3
    a.x1, a.c1 = 0x1000, c1
    if c2:
        a.x2, a.c2 = 0x2000, c2
    if c3:
        a.x3, a.c3 = 0x3000, c3
    if c4:
        a.x4, a.c4 = 0x4000, c4
    a.batch, a.hin, a.win, a.hout, a.wout, a.stride, a.up = B, H, W, H, W, 1, 0
    a.w, a.out, a.ldo, a.ldres1, a.ldres2 = 0x5000, 0x6000, cout, cout, cout
    a.rows_per_batch, a.scale, a.splitk = H * W, 1.0, splitk
    if gn_in:
        a.gn_in_acc, a.gn_in_gb, a.gn_in_groups, a.gn_in_silu, a.gn_in_eps = 0x7000, 0x8000, 32, 1, 1e-5
    return a


def test_fused_conv_routing_and_combine_apply_queries_on_the_host():
    """pp_conv_gn_supported / pp_conv_gn_preferred (ABI v16) and pp_gemm_gn_next_ok (ABI v17) are pure host logic: the
    routing classes of DESIGN section 4 at the SD-1.5 shapes (batch 8), without a GPU."""
    import ctypes as C
    from powerpaint_amd import _lib as L
    lib = L.lib()
    sup = lambda a: lib.pp_conv_gn_supported(C.byref(a))       # noqa: E731
    pref = lambda a: lib.pp_conv_gn_preferred(C.byref(a))      # noqa: E731
    # every level has a tile of whole image rows; preferred NOWHERE since round 6: the apply launch (or the apply in the
    # producer's combine) + the plain conv on the halo-tile loop without the normalisation is faster down to 16x16, at 8x8
    # the tap-major weight stream
    for (H, c1, c2, cout, tail, want) in [(64, 320, 0, 320, (0, 0), 0), (64, 640, 320, 320, (0, 0), 0), (64, 320, 0, 320, (640, 320), 0),
                                          (32, 320, 0, 640, (0, 0), 0), (32, 1280, 640, 640, (0, 0), 0), (32, 640, 0, 640, (640, 320), 0),
                                          (16, 1280, 1280, 1280, (0, 0), 0), (16, 640, 0, 1280, (0, 0), 0),
                                          (8, 1280, 0, 1280, (0, 0), 0), (8, 1280, 1280, 1280, (0, 0), 0)]:
        a = _conv_args(8, H, H, c1, c2, cout, *tail)
        assert sup(a) == 1, (H, c1, c2)
        assert pref(a) == want, (H, c1, c2)
    # not the fused kernel's geometry: stride 2, upsampling, channel counts off the 64 grid, no statistics
    a = _conv_args(8, 64, 64, 320, 0, 320)
    a.stride, a.hout, a.wout, a.M = 2, 32, 32, 8 * 32 * 32
    assert sup(a) == 0 and pref(a) == 0
    assert sup(_conv_args(8, 64, 64, 96, 0, 320)) == 0
    # a PLAIN conv (no statistics of its input): 2 = routed to the halo-tile loop without the normalisation, from 32 pixels of
    # width up and only where the automatic tile choice is asked for
    assert sup(_conv_args(8, 64, 64, 320, 0, 320, gn_in=False)) == 2 and sup(_conv_args(8, 32, 32, 640, 0, 640, gn_in=False)) == 2
    assert sup(_conv_args(8, 16, 16, 1280, 0, 1280, gn_in=False)) == 2 and sup(_conv_args(8, 8, 8, 1280, 0, 1280, gn_in=False)) == 0
    a = _conv_args(8, 64, 64, 320, 0, 320, gn_in=False)
    a.tile = L.PP_TILE_128x160
    assert sup(a) == 0
    # the apply of the CONSUMER norm inside the split-K combine: whole (batch item, group) populations per workgroup only
    def next_ok(H, cout, splitk, cg=None, c0=0, rows=None):
        a = _conv_args(8, H, H, 1280, 0, cout, gn_in=False, splitk=splitk)
        a.workspace = 0x9000
        a.gn_acc[0], a.gn_cg[0], a.gn_c0[0], a.gn_groups[0] = 0xa000, cg or cout // 32, c0, 32
        if rows:
            a.rows_per_batch = rows
        return lib.pp_gemm_gn_next_ok(C.byref(a), 0)
    assert next_ok(8, 1280, 8) == 1 and next_ok(16, 1280, 4) == 1      # 64 / 256 rows per batch item, split-K launches
    assert next_ok(8, 1280, 1) == 0                                    # no combine behind the launch
    assert next_ok(32, 640, 2) == 0                                    # 1024 rows per batch item
    assert next_ok(8, 1280, 8, c0=64) == 0                             # a concatenated consumer (channel offset)
    assert next_ok(8, 1280, 8, cg=42) == 0                             # groups that straddle the 40-column tiles
    a = _conv_args(8, 8, 8, 1280, 0, 1280, gn_in=False, splitk=8)
    assert lib.pp_gemm_gn_next_ok(C.byref(a), 0) == 0                  # no statistics subscription
    # row-local fused kernels: C = 320 in 128-row tiles, wider in 64-row tiles
    assert lib.pp_tfront_supported(32768, 320, 4096, 32) == 1 and lib.pp_tfront_supported(8192, 640, 1024, 32) == 0
    assert lib.pp_tfront_supported(320, 320, 64, 32) == 0
    assert lib.pp_xattn_block_supported(32768, 320, 4096, 77, 8) == 1 and lib.pp_xattn_block_supported(8192, 640, 1024, 77, 8) == 1
    assert lib.pp_xattn_block_supported(512, 1280, 64, 77, 8) == 1 and lib.pp_xattn_block_supported(512, 960, 64, 77, 8) == 0
    assert lib.pp_xattn_block_supported(32768, 320, 4096, 81, 8) == 0


def test_kperm_is_the_accumulator_to_operand_layout_of_the_chained_gemms():
    """engine._kperm (the input-index permutation of the second GEMM's weights in csrc/tfront.hip; the same map packs H^T in
    csrc/xattn_fused.hip): lane (row m, k-group g) of a 16x16x32 MFMA holds, as accumulator quads of the FIRST GEMM, the
    columns 16 nb + 4 g + {0..3} of its row; quads of blocks 2 s and 2 s + 1 back to back are the lane's eight B-operand
    values of k-block s.  With the weights permuted by _kperm the second GEMM then contracts matching indices."""
    import torch
    from powerpaint_amd.engine import _kperm
    K, N = 320, 48
    g_ = torch.Generator().manual_seed(0)
    hs, w = torch.randn(16, K, generator=g_), torch.randn(N, K, generator=g_)
    perm = _kperm(torch.arange(K).float()[None, :])[0].long()
    assert sorted(perm.tolist()) == list(range(K))
    # what the lanes hand to the MFMA: position 32 s + 8 g + j of row m <- accumulator element (block 2 s + (j >> 2), quad index j & 3)
    frag = torch.empty(16, K)
    for s in range(K // 32):
        for g in range(4):
            for j in range(8):
                nb, i = 2 * s + (j >> 2), j & 3
                frag[:, 32 * s + 8 * g + j] = hs[:, 16 * nb + 4 * g + i]
    assert torch.equal(frag, hs[:, perm])
    assert torch.allclose(frag @ _kperm(w).t(), hs @ w.t(), atol=1e-4)
