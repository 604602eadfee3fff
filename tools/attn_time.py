import os, sys, torch
sys.path.insert(0, "/root/repo")
from powerpaint_amd import ops
d, nq, nk = 40, 4096, 4096
B, H = 8, 8
C = H * d
q = torch.randn(B * nq, 2 * C, device="cuda").to(torch.bfloat16)
k = torch.randn(B * nk, 2 * C, device="cuda").to(torch.bfloat16)
v = torch.randn(B * nk, C, device="cuda").to(torch.bfloat16)
vt = ops.transpose_v(v, B, nk)
for _ in range(4):
    ops.attention(q[:, :C], k[:, C:], vt, B, H, nq, nk, d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention(q[:, :C], k[:, C:], vt, B, H, nq, nk, d)
e1.record(); torch.cuda.synchronize()
print("attention d=40 N=4096 B=8:", e0.elapsed_time(e1) * 1e3 / 20, "us")
