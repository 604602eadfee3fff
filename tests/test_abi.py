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
