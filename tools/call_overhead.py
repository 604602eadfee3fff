"""What a pipeline call costs OUTSIDE its denoising steps (the bench's images/s includes it): wall time of pipe(**kw) at
several step counts -> slope (ms per step) and intercept (ms per call); then a cProfile of one call's host side.
  python tools/call_overhead.py [--config v1]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "v1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pipe, nets, _ = bench.build_pipeline(cfg, dev, 0, 1, dtype=torch.bfloat16, scheduler=None, bcast_timeout=0)
pipe.use_graph = True
pts = []
for n in (50, 10, 25, 50, 100):
    kw = bench.synthetic_inputs(cfg, dev, 0, 4, 64, n)
    for _ in range(2):
        pipe(**kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        pipe(**kw)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    pts.append((n, ts[1]))
    print(f"{n:4d} denoise steps: {ts[1]:9.3f} ms per call  ({ts[1] / n:7.4f} ms per step)")
(n1, t1), (n2, t2) = pts[1], pts[-1]
slope = (t2 - t1) / (n2 - n1)
print(f"slope {slope:.4f} ms per step, intercept {t1 - slope * n1:.3f} ms per call (from {n1} and {n2} steps)")
kw = bench.synthetic_inputs(cfg, dev, 0, 4, 64, 50)
pipe(**kw)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
pipe(**kw)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
