"""Multi-GPU: tile rows sharded across ranks, one process per GPU, no data-path
collective while rendering; ONE framebuffer gather at the end.

The reference has no multi-device code at all (single MTLDevice,
TestApp/ViewController.m:16).  Tiles are independent given the read-only scene,
so rank r renders a contiguous band of tile rows and the bands are gathered to
rank 0 over RCCL/xGMI (`nccl` backend): one grouped send/recv, every band lands
directly in its rows of the final image (no padded staging, no second copy) and
the root's 7 links each carry one band.  torch.distributed is used purely as the
transport; on CPU test boxes the same code runs over `gloo`.  The same exchange
exists behind the C ABI for non-Python hosts (pm_comm_create / pm_gather).
"""
from __future__ import annotations

import bisect
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


def _layout_from_cuts(cuts: list[int], height: int) -> list[tuple[int, int, int]]:
    return [(a, b, max(0, min(b * 16, height) - a * 16)) for a, b in zip(cuts[:-1], cuts[1:])]


def band_layout(height: int, world: int, cuts: list[int] | None = None) -> list[tuple[int, int, int]]:
    """[(tile_row0, tile_row1, pixel_rows)] per rank for a viewport of `height` px: the
    near-equal split, or the split at the given tile-row `cuts` (world + 1 entries)."""
    tiles_y = (height + 15) // 16
    if cuts is None:
        cuts = [band_rows(tiles_y, world, r)[0] for r in range(world)] + [tiles_y]
    assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == tiles_y
    return _layout_from_cuts(list(cuts), height)


def sub_band_layouts(layout: list[tuple[int, int, int]], height: int, chunks: int) -> list[list[tuple[int, int, int]]]:
    """The gather pipelined under the render: every rank's band cut into `chunks` runs of whole tile rows (as equal
    as the rows allow; a band with fewer rows than chunks gets empty ones), chunk k of all ranks forming one band table
    of the shape `band_layout` returns.  A rank renders its sub-band k, posts ITS gather on a second stream and renders
    sub-band k + 1 meanwhile: at 8 GPUs the 8192^2 frame is 0.07 ms of rendering in front of >= 0.22 ms of wire time
    (DESIGN.md 6), so what overlaps is nearly all of a rank's rendering.  Deterministic in (layout, chunks): every rank
    computes every rank's rows."""
    assert chunks >= 1
    out = []
    for k in range(chunks):
        cuts = []
        for a, b, _rows in layout:
            s0, s1 = a + (b - a) * k // chunks, a + (b - a) * (k + 1) // chunks
            cuts.append((s0, s1, max(0, min(s1 * 16, height) - s0 * 16)))
        out.append(cuts)
    return out


def balanced_cuts(cuts: list[int], band_ms: list[float]) -> list[int]:
    """Cost-balanced tile-row split ("Tiger rows are uneven").

    Input: the current split and the render time every rank measured for its band.  The cost
    per tile row is taken as constant inside each band; the new cuts are the equal-cost
    quantiles of that piecewise-constant density.  Every band keeps at least one tile row.
    Pure host arithmetic -- iterate it (measure, re-cut) two or three times."""
    world = len(cuts) - 1
    tiles_y = cuts[-1]
    if world == 1 or tiles_y <= world:
        return list(cuts)
    dens = []
    for r in range(world):
        rows = max(1, cuts[r + 1] - cuts[r])
        dens += [max(band_ms[r], 1e-9) / rows] * (cuts[r + 1] - cuts[r])
    cum = [0.0]
    for d in dens:
        cum.append(cum[-1] + d)
    out = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        c = bisect.bisect_left(cum, target)
        if c > 0 and target - cum[c - 1] < cum[min(c, tiles_y)] - target:
            c -= 1
        out.append(min(max(c, out[-1] + 1), tiles_y - (world - r)))
    out.append(tiles_y)
    return out


def padded_band_rows(height: int, world: int, cuts: list[int] | None = None) -> int:
    return max(p for _, _, p in band_layout(height, world, cuts))


def gather_bands(band, layout, height: int, dst: int = 0, full=None):
    """ONE grouped send/recv: every rank's band ([>= rows, width, 4] uint8, its first `rows`
    pixel rows valid) goes straight into rows [row0*16, row0*16 + rows) of `full`
    ([height, width, 4]) on rank `dst`.  The root's own band is copied locally unless `band`
    already is that slice of `full` (render straight into the final image).

    Returns `full` on `dst`, None elsewhere.  With world == 1 the band is the image."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return band[:height]
    r0, _r1, rows = layout[rank]
    ops = []
    if rank == dst:
        if full is None:
            full = torch.empty((height, band.shape[1], 4), dtype=band.dtype, device=band.device)
        for k, (k0, _k1, krows) in enumerate(layout):
            if krows == 0:
                continue
            view = full[k0 * 16 : k0 * 16 + krows]
            if k == dst:
                if view.data_ptr() != band.data_ptr():
                    view.copy_(band[:krows])
            else:
                ops.append(dist.P2POp(dist.irecv, view, k))
    elif rows:
        ops.append(dist.P2POp(dist.isend, band[:rows], dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()  # (nccl: orders the current stream behind the transfer, does not block the host)
    return full if rank == dst else None


def allgather_bands(band_padded, layout, height: int, full=None, scratch=None):
    """The all-gather alternative (every rank ends up with the image): bands padded to a common
    size, ncclAllGather, then one strided copy per band.  Kept to compare schedules on xGMI
    (a ring all-gather is per-link bound at 7 steps; the grouped gather is not)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return band_padded[:height]
    if scratch is None:
        scratch = torch.empty((world,) + tuple(band_padded.shape), dtype=band_padded.dtype, device=band_padded.device)
    dist.all_gather(list(scratch.unbind(0)), band_padded)  # one ncclAllGather on the nccl backend
    if full is None:
        full = torch.empty((height, band_padded.shape[1], 4), dtype=band_padded.dtype, device=band_padded.device)
    for k, (k0, _k1, krows) in enumerate(layout):
        if krows:
            full[k0 * 16 : k0 * 16 + krows] = scratch[k, :krows]
    return full


def gather_framebuffer(band, height: int, dst: int = 0, full=None):
    """Equal-split convenience form of gather_bands (kept for callers of the round-1 API)."""
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    return gather_bands(band, band_layout(height, world), height, dst=dst, full=full)


def agree_on_c_abi_gather(requested: str, world: int, make_id, broadcast_id, make_comm, try_exchange, all_min):
    """Which exchange every rank of a job uses for the band gather: the product's own collective behind the C ABI
    (pm_comm_create / pm_gather, "cabi") or torch.distributed's grouped send/recv ("sendrecv") -- decided so that a job ALWAYS ends
    with one answer on every rank, whatever fails where (bench.py prints its line either way):

      1. every rank loads RCCL (make_id() makes an id, which binds the library); the ranks agree (all_min) that all could --
         ncclCommInitRank is a collective, a rank that cannot even load the library must be found before the others enter it;
      2. rank 0's id travels (broadcast_id), every rank creates its communicator (make_comm(id)) and runs ONE exchange
         (try_exchange(comm)); the ranks agree that all of that worked everywhere.

    requested: "auto" (fall back to "sendrecv" when a step fails anywhere), "cabi" (raise instead), anything else is returned as is.
    Returns (impl, comm or None, why) -- why: the first failure this rank saw, "" if none."""
    if requested not in ("auto", "cabi"):
        return requested, None, ""
    ok, why, comm = 1, "", None
    try:
        make_id()
    except Exception as e:  # noqa: BLE001
        ok, why = 0, repr(e)
    if all_min(ok) != 1:
        if requested == "cabi":
            raise RuntimeError(f"--gather-impl cabi: RCCL cannot be loaded on some rank ({why or 'another rank'})")
        return "sendrecv", None, why or "RCCL not loadable on another rank"
    try:
        uid = broadcast_id(make_id)
        comm = make_comm(uid)
        try_exchange(comm)
    except Exception as e:  # noqa: BLE001 -- any failure means: not on this stack
        ok, why = 0, repr(e)
    if all_min(ok) == 1:
        return "cabi", comm, ""
    if requested == "cabi":
        raise RuntimeError(f"--gather-impl cabi: pm_comm_create / pm_gather failed on some rank ({why or 'another rank'})")
    return "sendrecv", None, why or "pm_comm_create / pm_gather failed on another rank"


def gather_wire_floor_ms(layout, width: int, world: int, link_gbs: float = 153.0) -> float:
    """What the band gather cannot beat on one MI355X node: every non-root band travels into the root over its own xGMI link
    (point to point, ~153 GB/s each, seven links per GPU), so the exchange takes at least the LARGEST non-root band's bytes over one
    link -- launch and protocol latencies (tens of microseconds) come on top."""
    if world <= 1:
        return 0.0
    worst = max((rows for k, (_r0, _r1, rows) in enumerate(layout) if k != 0), default=0)
    return worst * width * 4 / (link_gbs * 1e9) * 1e3
