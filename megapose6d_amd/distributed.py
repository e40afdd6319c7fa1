"""Row sharding + gathers for the multi-GPU path (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

The hot path shards naturally: every (object, hypothesis) row is independent through coarse
scoring, each refiner chain and re-scoring; the only coupling is the per-detection top-K / arg-max.  So: rank r owns
rows r, r+W, r+2W, ... (interleaved: every rank touches every object, meshes stay balanced), results are exchanged with
ONE all-gather per stage of a packed [rows, k] fp32 tensor (<= 2.5 MB at 64 x 576 x 17 floats -- latency-bound, far below
the per-link xGMI budget, so no bucketing or ring tuning is warranted), and every rank then holds the full table.
The reference has no in-pipeline collective at all (its eval gathers through the filesystem,
src/megapose/utils/tensor_collection.py:165-186).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


# One-rank emulation (a MEASUREMENT RIG, bench.py --emulate-rank-of N; never a result): the process behaves like rank `r` of `world` --
# it owns rows r::world of every stage table -- without a process group; a "gather" fills the other ranks' rows with copies of its own
# block (right shapes, plausible values, wrong poses).  What it times is exactly one rank's share of an N-GPU call minus the RCCL
# all-gathers (3 per call, a few MB: tens of microseconds over xGMI), on a box that has one GPU.
_emulated: Optional[tuple] = None


def emulate(r: Optional[int], world: int = 1) -> None:
    """switch the one-rank emulation on (r, world) or off (None)"""
    global _emulated
    if r is None:
        _emulated = None
        return
    assert not is_initialized(), "emulate() is for a process WITHOUT a process group"
    assert 0 <= r < world and world >= 1
    _emulated = (int(r), int(world))


def emulated() -> bool:
    return _emulated is not None


def rank() -> int:
    if _emulated is not None:
        return _emulated[0]
    return dist.get_rank() if is_initialized() else 0


def world_size() -> int:
    if _emulated is not None:
        return _emulated[1]
    return dist.get_world_size() if is_initialized() else 1


def init_from_env(backend: Optional[str] = None) -> None:
    """Initialise the default process group from torchrun's environment (RANK / WORLD_SIZE / MASTER_*)."""
    if is_initialized() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC; only effective if the HSA runtime has not started yet
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")


def shard_indices(n: int, r: int, world: int) -> np.ndarray:
    """Rows owned by rank r: r, r+world, ..."""
    return np.arange(r, n, world)


def shard_size(n: int, r: int, world: int) -> int:
    return len(range(r, n, world))


# Per-process gather statistics (bench.py reports them): number of all-gathers, payload bytes and -- when `timing` is switched
# on -- the device time of each RCCL all-gather measured with HIP events on the stream it is issued on.
class GatherStats:
    def __init__(self) -> None:
        self.timing = False
        self.reset()

    def reset(self) -> None:
        self.calls = 0
        self.bytes = 0
        self.backend = None
        self._events = []
        self.host_s = 0.0   # host time of the replicated table work (pandas top-K / arg-max): does not shrink with the world size

    def ms(self) -> float:
        """sum of the recorded all-gather intervals (synchronises)"""
        if not self._events:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in self._events))


stats = GatherStats()


def gather_rows(local: torch.Tensor, n: int, r: int, world: int, group=None) -> torch.Tensor:
    """Inverse of shard_indices: every rank passes its [shard_size, k] rows and receives the full [n, k] table in
    original row order.  One all_gather of equally sized (padded) blocks."""
    assert local.dim() == 2 and local.shape[0] == shard_size(n, r, world)
    per = (n + world - 1) // world
    k = local.shape[1]
    block = torch.zeros(per, k, dtype=local.dtype, device=local.device)
    block[: local.shape[0]] = local
    if _emulated is not None:   # measurement rig: every other rank's block := a copy of ours (see `emulate`)
        stats.calls += 1
        stats.bytes += world * per * k * local.element_size()
        stats.backend = "emulated"
        if local.shape[0] < per and local.shape[0] > 0:
            block[local.shape[0]:] = local[-1]
        return block.unsqueeze(1).expand(per, world, k).reshape(per * world, k)[:n].contiguous()
    out = torch.empty(world * per, k, dtype=local.dtype, device=local.device)
    backend = dist.get_backend(group)
    stats.calls += 1
    stats.bytes += out.numel() * out.element_size()
    stats.backend = backend
    if backend == "nccl":  # RCCL over xGMI: one fused all-gather
        ev = None
        if stats.timing:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_gather_into_tensor(out, block, group=group)
        if ev is not None:
            ev[1].record()
            stats._events.append(ev)
    else:  # gloo (CPU tests, or several ranks sharing one GPU in tests): gather through host memory
        host = torch.empty(world * per, k, dtype=local.dtype)
        _all_gather_cpu(host, block.cpu(), world, group)
        out.copy_(host)
    # out[q*per + j] is row q + j*world
    full = out.view(world, per, k).transpose(0, 1).reshape(per * world, k)
    return full[:n].contiguous()


def _all_gather_cpu(out: torch.Tensor, block: torch.Tensor, world: int, group=None) -> None:
    parts = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(parts, block, group=group)
    out.copy_(torch.cat(parts, dim=0))
