"""Multi-GPU data parallelism of the denoising path: one process per GPU, images sharded across ranks, ONE collective.

The path shards naturally: every image (and its CFG twin) is independent -- GroupNorm, LayerNorm and
attention are per sample -- so rank r owns images [r*per_rank, (r+1)*per_rank) and nothing crosses ranks inside the
step loop.  The only traffic is a start-up broadcast of each network's packed parameter buffer from rank 0 (RCCL over
xGMI when the backend is "nccl"; one flat contiguous buffer per network so the library can split it across all links),
plus an optional all-gather of the final latents.  Per-image seeds are derived from the GLOBAL image index so results
do not depend on the rank count.
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class Watchdog:
    """`with Watchdog(seconds, what):` -- if the block has not finished after `seconds`, print a clear message naming
    the stuck stage and the rank, and end the process (exit code 3).  A hung RCCL bootstrap or first collective (wrong
    MASTER_ADDR, a peer that died, IPC handles refused) otherwise blocks every rank forever and the launcher's own
    timeout says nothing about where.  seconds <= 0 disables it."""

    def __init__(self, seconds: float, what: str):
        self.seconds, self.what, self.timer = seconds, what, None

    def _fire(self):
        import sys
        rank, world = os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1")
        sys.stderr.write(f"[powerpaint_amd.dist] rank {rank}/{world}: '{self.what}' did not finish within "
                         f"{self.seconds:.0f} s -- giving up (MASTER_ADDR={os.environ.get('MASTER_ADDR')}, "
                         f"MASTER_PORT={os.environ.get('MASTER_PORT')}, "
                         f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})\n")
        sys.stderr.flush()
        sys.stdout.flush()
        # hard exit with the documented code 3 (both streams flushed above): a rank wedged inside a collective in native
        # code acts on neither an exception raised in this timer thread nor, reliably, on SIGTERM
        os._exit(3)

    def __enter__(self):
        if self.seconds and self.seconds > 0:
            import threading
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


def init_from_env(backend: Optional[str] = None, timeout_s: float = 0.0) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank); initialises torch.distributed when WORLD_SIZE > 1.  `timeout_s` > 0: the
    rendezvous and a first 4-byte all-reduce must finish within that time or the process exits with a message."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        kw = {}
        if timeout_s and timeout_s > 0:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        with Watchdog(timeout_s, f"torch.distributed init_process_group(backend={backend})"):
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        with Watchdog(timeout_s, f"first all-reduce over {backend}"):
            t = torch.ones(1, device=torch.device("cuda", local) if backend == "nccl" else "cpu")
            dist.all_reduce(t)
            if backend == "nccl":
                torch.cuda.synchronize()
            if int(t.item()) != world:
                raise RuntimeError(f"first all-reduce returned {t.item()} on a world of {world}")
    return rank, world, local


def device_identity(local: int) -> str:
    """'name | pci <bus id> | <GB> GB' of this rank's GPU (bench.py logs it per rank for N > 1)."""
    if not torch.cuda.is_available():
        return "cpu"
    p = torch.cuda.get_device_properties(local)
    bus = ""
    for attr in ("pci_bus_id", "pci_device_id", "pci_domain_id"):
        if hasattr(p, attr):
            bus += f"{attr.split('_')[1]}={getattr(p, attr)} "
    uuid = getattr(p, "uuid", "")
    return f"cuda:{local} {p.name} | pci {bus.strip() or '?'} | uuid {uuid} | {p.total_memory / 2 ** 30:.0f} GiB"


def gather_strings(s: str) -> List[str]:
    """Every rank's string on every rank (start-up logging)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [s]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, s)
    return out


def shard_range(global_batch: int, rank: int, world: int) -> range:
    """Contiguous block partition of the image batch; remainder images go to the lowest ranks."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def image_generator(global_index: int, base_seed: int = 1234) -> torch.Generator:
    """CPU generator keyed by the global image index (rank-count invariant inputs)."""
    return torch.Generator("cpu").manual_seed(base_seed + int(global_index))


def broadcast_params(buffers: List[torch.Tensor], src: int = 0):
    """One broadcast per network parameter buffer (uint8 view of the packed bf16/fp32 arena)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for b in buffers:
        dist.broadcast(b, src=src)


def broadcast_models(models, src: int = 0):
    """Start-up weight broadcast for any mix of the networks: the packed-buffer models (UNet / BrushNet / ControlNet /
    AutoencoderKL: one collective each, their `param_buffer()`), and nn.Modules (CLIPTextModel: parameters and buffers
    flattened into one contiguous tensor per dtype, broadcast, scattered back -- a handful of collectives instead of
    one per tensor).  Ranks other than `src` only need the same architecture; their values are overwritten."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for m in models:
        if hasattr(m, "param_buffer"):
            dist.broadcast(m.param_buffer(), src=src)
            if hasattr(m, "params_changed"):
                m.params_changed()      # (stale time-embedding rows / folded cross-attention operands otherwise)
            continue
        if not isinstance(m, torch.nn.Module):
            raise TypeError(f"cannot broadcast {type(m).__name__}: neither param_buffer() nor an nn.Module")
        by_dtype = {}
        for t in list(m.parameters()) + list(m.buffers()):
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
        for ts in by_dtype.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src)
            off = 0
            with torch.no_grad():
                for t in ts:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n


def gather_latents(local: torch.Tensor, global_batch: int) -> Optional[torch.Tensor]:
    """All-gather the per-rank final latents into global image order (equal shards required)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if global_batch % world:
        raise ValueError("gather_latents needs equal shards")
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local.contiguous())
    return torch.cat(out, dim=0)


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
