"""Multi-GPU: tile rows sharded across ranks, one process per GPU, no data-path
collective while rendering; ONE framebuffer gather at the end.

The reference has no multi-device code at all (single MTLDevice,
TestApp/ViewController.m:16).  Tiles are independent given the read-only scene,
so rank r renders the contiguous band of tile rows workloads.band_rows(...) gives
it and the bands are gathered to rank 0 over RCCL/xGMI (`nccl` backend) -- 7
direct links into the root, one band per link.  torch.distributed is used purely
as the collective transport; on CPU test boxes the same code runs over `gloo`.
"""
from __future__ import annotations

import os

from .workloads import band_rows


def env_rank_world() -> tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (idempotent)."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def band_layout(height: int, world: int) -> list[tuple[int, int, int]]:
    """[(tile_row0, tile_row1, pixel_rows)] per rank for a viewport of `height` px."""
    tiles_y = (height + 15) // 16
    out = []
    for r in range(world):
        r0, r1 = band_rows(tiles_y, world, r)
        out.append((r0, r1, max(0, min(r1 * 16, height) - r0 * 16)))
    return out


def padded_band_rows(height: int, world: int) -> int:
    return max(p for _, _, p in band_layout(height, world))


def gather_framebuffer(band, height: int, dst: int = 0, full=None):
    """Gather every rank's band ([padded_rows, width, 4] uint8 tensor, same shape on
    all ranks) into the full [height, width, 4] image on rank `dst`.

    Returns the assembled tensor on `dst`, None elsewhere.  With world == 1 the band
    is the image."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    layout = band_layout(height, world)
    if world == 1:
        return band[:height]
    if rank == dst:
        gathered = torch.empty((world,) + tuple(band.shape), dtype=band.dtype, device=band.device)
        dist.gather(band, gather_list=list(gathered.unbind(0)), dst=dst)
        if full is None:
            full = torch.empty((height, band.shape[1], 4), dtype=band.dtype, device=band.device)
        for r, (r0, _r1, rows) in enumerate(layout):
            if rows:
                full[r0 * 16 : r0 * 16 + rows] = gathered[r, :rows]
        return full
    dist.gather(band, gather_list=None, dst=dst)
    return None
