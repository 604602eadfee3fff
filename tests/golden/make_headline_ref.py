#!/usr/bin/env python
"""Generates tests/golden/headline_ref.pt: what the CPU oracle (oracle/loops.py, oracle/sd_modules.py -- the restatement
of the reference's denoising loops, see tests/headline_cases.py for the file:line it follows) produces for the two
headline-parity cases, so that the GPU test compares the HIP path with it in seconds.  ~10 minutes on 8 cores:

    python tests/golden/make_headline_ref.py

The oracle is deterministic for a given torch build up to the summation order of the CPU BLAS / conv kernels (1e-6
relative, four orders below the gates).  `PP_HEADLINE_LIVE=1 pytest tests/test_headline_parity_gpu.py -m gpu` runs the
oracle live instead and also checks it against this file.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import headline_cases as HC  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count())
    t0 = time.time()
    with torch.no_grad():
        lat_after = HC.config2_oracle_run(HC.config2_oracle_model(), HC.config2_inputs())
        print(f"config 2: {len(lat_after)} steps in {time.time() - t0:.0f} s", flush=True)
        t1 = time.time()
        ou, ob = HC.config3_oracle_models()
        eps, ts, lat3 = HC.config3_oracle_run(ou, ob, HC.config3_inputs())
        print(f"config 3: {len(eps)} steps in {time.time() - t1:.0f} s", flush=True)
    keep = (9, 19, 29, 39, 49)
    out = dict(torch_version=torch.__version__,
               cfg2_lat_after={k: lat_after[k].clone() for k in keep},
               cfg2_scale=[float(l.abs().max()) for l in lat_after],
               cfg3_eps=torch.stack(eps), cfg3_t=ts, cfg3_lat_after=torch.stack(lat3))
    torch.save(out, os.path.join(ROOT, "tests", "golden", "headline_ref.pt"))
    print("wrote tests/golden/headline_ref.pt", flush=True)


if __name__ == "__main__":
    main()
