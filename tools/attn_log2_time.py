"""The shipping self-attention launch of the 64x64 level (PP_ATTN_PIPE_LOG2, d = 40, N = 4096, batch 8 x 8 heads), hot.
  python tools/attn_log2_time.py     (lab: PP_ATTN_NOSTORE=1 skips the output store)"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops, _lib as L
d, n, B, H = 40, 4096, 8, 8
C = H * d
q = (torch.randn(B * n, 2 * C, device="cuda") * (d ** -0.5 * math.log2(math.e))).to(torch.bfloat16)
v = torch.randn(B * n, C, device="cuda").to(torch.bfloat16)
vt = ops.transpose_v(v, B, n)
f = lambda: ops.attention(q[:, :C], q[:, C:], vt, B, H, n, n, d, variant=L.PP_ATTN_PIPE_LOG2)
for _ in range(4):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record(); torch.cuda.synchronize()
print(f"attention LOG2 d=40 N=4096 B=8: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us  NOSTORE={os.environ.get('PP_ATTN_NOSTORE')}")
