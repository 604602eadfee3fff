"""CPU: static checks on the gfx950 code objects inside libpp_hip.so (no GPU needed).

The hot kernels must not touch scratch (private) memory: twice in this code base a by-reference lambda capture or an
index-driven select made LLVM keep a closure / a lookup table in scratch and the kernel 2-3x slower without any
functional symptom.  The metadata of every kernel (`.private_segment_fixed_size`, `.vgpr_count`) is read from the
embedded fat binary with the ROCm LLVM tools."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# kernels known to spill a little (bytes): the legacy register-staged GEMM (tiles 1-3: tests and fallback only) and the
# GroupNorm-statistics epilogue of the 8-wave 128x160x2 tile under its 128-VGPR budget (one launch per UNet step)
ALLOWED = [(r"pp_gemm_kernelILi", 32), (r"pp_gemm_kernel_v2ILi128ELi160ELi4ELi2ELi[01]ELi2ELi4ELb0ELb0E", 128)]


def kernels():
    so = os.path.join(ROOT, "powerpaint_amd", "libpp_hip.so")
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(so) or not all(os.path.exists(t) for t in tools):
        pytest.skip("libpp_hip.so or the ROCm LLVM tools are not available")
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([tools[0], f"--dump-section=.hip_fatbin={fat}", so, os.path.join(d, "unused")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            piece = os.path.join(d, f"b{i}.fat")
            open(piece, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, f"b{i}.co")
            subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={piece}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes = subprocess.run([tools[2], "--notes", co], capture_output=True, text=True).stdout
            name = None
            rec = {}
            for line in notes.splitlines():
                m = re.match(r"\s*\.(name|private_segment_fixed_size|vgpr_count|agpr_count):\s+(\S+)", line)
                if not m:
                    continue
                if m.group(1) == "name":
                    if name:
                        out.append((name, rec))
                    name, rec = m.group(2), {}
                else:
                    rec[m.group(1)] = int(m.group(2))
            if name:
                out.append((name, rec))
    return out


def test_no_hot_kernel_uses_scratch_memory():
    ks = kernels()
    assert len(ks) >= 200, len(ks)                      # five translation units, every template instantiation
    names = [n for n, _ in ks]
    for must in ("attn_pipe_kernel", "pp_gemm_kernel_v2", "gn_apply_kernel", "conv3x3_cout4_mfma_kernel",
                 "cfg_sched_step_kernel", "ddim_variance_noise_kernel", "latent_blend_kernel"):
        assert any(must in n for n in names), must
    bad = []
    for n, r in ks:
        scratch = r.get("private_segment_fixed_size", 0)
        limit = max([lim for pat, lim in ALLOWED if re.search(pat, n)] + [0])
        if scratch > limit:
            bad.append((n, scratch))
    assert not bad, bad


def test_register_budgets_of_the_shipping_configurations():
    """Occupancy-defining register counts: the ping-pong GEMM tiles and the attention kernels must keep two waves per
    SIMD (<= 256 VGPRs + AGPRs), the 64-query attention waves in particular (210 measured)."""
    ks = dict(kernels())
    for n, r in ks.items():
        total = r.get("vgpr_count", 0) + r.get("agpr_count", 0)
        if "attn_pipe_kernelILi40ELi32ELi0E" in n or re.search(r"pp_gemm_kernel_v2ILi(256|128)ELi160ELi4ELi2ELi[01]ELi[34]ELi[04]ELb0ELb1E", n):
            assert total <= 256, (n, total)
