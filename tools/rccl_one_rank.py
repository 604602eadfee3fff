#!/usr/bin/env python
"""RCCL on the GPU box with THIS code's collectives, world size 1 (the pool's boxes have one GPU; N > 1 has never run on
hardware -- DESIGN.md section 6).  `powerpaint_amd.dist` short-circuits every helper at world size 1, so this script makes the
same torch.distributed calls directly, with the tensors the product hands over: the packed parameter buffer of the full SD-1.5
UNet as one uint8 broadcast, the float64 MAX all-reduce of bench.py's timing, all_gather_object of the rank log, all_gather
of final latents, barrier.  What it can show: the nccl (= RCCL) backend initialises on the box under the launcher's environment
(HSA_ENABLE_IPC_MODE_LEGACY=0, 127.0.0.1 rendezvous), accepts these dtypes / sizes, and leaves the buffers intact.  What it
cannot: any inter-GPU transport.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_one_rank.py
-> profiles/r06_rccl_one_rank.txt"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import dist as ppdist  # noqa: E402
from powerpaint_amd import models as PM  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    t0 = time.perf_counter()
    with ppdist.Watchdog(120, "init_process_group(nccl)"):
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    print(f"init_process_group(nccl) world {world}: {time.perf_counter() - t0:.2f} s; HSA_ENABLE_IPC_MODE_LEGACY="
          f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')} MASTER_ADDR={os.environ.get('MASTER_ADDR')}")
    print(ppdist.device_identity(local))
    unet = PM.UNet2DConditionModel(in_channels=9, device=dev, dtype=torch.bfloat16)
    unet.load_state_dict(unet.net.synthetic_state_dict(device=dev, seed=1))
    buf = unet.param_buffer()
    before = int(buf.view(torch.int64)[: buf.numel() // 8].sum().item())
    with ppdist.Watchdog(120, "broadcast of the packed parameter buffer"):
        t0 = time.perf_counter()
        dist.broadcast(buf, src=0)
        torch.cuda.synchronize()
    after = int(buf.view(torch.int64)[: buf.numel() // 8].sum().item())
    print(f"broadcast {buf.dtype} x {buf.numel() / 2 ** 20:.0f} MiB (full SD-1.5 9-channel UNet, one collective): "
          f"{time.perf_counter() - t0:.3f} s, checksum {'intact' if before == after else 'CHANGED'}")
    t = torch.tensor([1.234567], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    print(f"all_reduce(MAX, float64): {float(t.item()):.6f}")
    out = [None] * world
    dist.all_gather_object(out, f"rank {rank}: {ppdist.device_identity(local)}")
    print(f"all_gather_object: {len(out)} string(s)")
    lat = torch.randn(4, 4, 64, 64, device=dev)
    parts = [torch.empty_like(lat) for _ in range(world)]
    dist.all_gather(parts, lat)
    print(f"all_gather of final latents {tuple(lat.shape)}: equal = {bool(torch.equal(parts[0], lat))}")
    dist.barrier()
    torch.cuda.synchronize()
    print("barrier: ok")
    dist.destroy_process_group()
    assert before == after


if __name__ == "__main__":
    main()
