#!/usr/bin/env python
"""bench.py -- the metric BASELINE.json names: Mpixels/s on the Ghostscript Tiger at
3840x2160 (fills + strokes, BASELINE config 3), 1/2/4/8 GPUs, with the HBM-roofline
fraction of the dominant kernel and the CPU oracle timed beside it.

    python bench.py --gpus N --steps K --warmup W

A "step" is one frame of the hot path -- pm_bin_kernel, pm_coarse_kernel, pm_fine_kernel
(tileKernel + renderKernel + composite of the reference) -- over the scene already
resident in HBM (flatten/encode happens once per scene, like the reference encodes once
per resize, PietRenderer.m:145).  Frames are submitted back to back without waiting,
as the reference commits command buffers (PietRenderer.m:102): up to four frames are in
flight on four in-order streams, so `value` is frames completed per second x pixels;
the latency of one frame alone is reported next to it (roofline.frame_latency_ms).

N = 1 : one 3840x2160 Tiger frame per step.
N > 1 : weak scaling -- the viewport is 3840 x (2160*N) with one Tiger per 2160-row
        band; rank r renders the tile rows of band r (the scene is replicated; the path
        has no exchange step, so there is no collective in the timed region and every
        band stays in its GPU's HBM, exactly as the frame does at N = 1).
        value = all pixels of all ranks / time of the slowest rank.
        The one collective of the design -- gathering the bands to rank 0 over RCCL/xGMI
        for presentation -- is timed separately after the run and reported as
        config.gather_ms (--gather puts it into every step instead).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def stacked_tigers(pm, n: int):
    """n Tigers at the config-3 scale (10.8), one per 2160-row band."""
    wl = pm.workloads.tiger(3840, 2160)
    if n == 1:
        return wl
    scale = wl.width_scale
    sets = [wl.paths.transformed((1.0, 0.0, 0.0, 1.0, 0.0, 2160.0 * k / scale)) for k in range(n)]
    wl.paths = pm.PathSet.concat(sets)
    wl.height = 2160 * n
    wl.name = f"tiger_3840x2160_x{n}"
    return wl


def cpu_baseline(pm, wl_single, seconds_budget: float = 12.0):
    """The oracle (a scalar C port of the reference algorithm) on this host, rank 0 only."""
    from oracle import pmo

    scene, _ = pmo.scene_from_paths(pmo.scaled_paths(wl_single.paths.paths, wl_single.width_scale), wl_single.paths.els, wl_single.affine)
    w, h = wl_single.width, wl_single.height
    frames, t0 = 0, time.perf_counter()
    while True:
        pmo.render(scene, w, h)
        frames += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or frames >= 8:
            break
    mpix = w * h * frames / el / 1e6
    out = {
        "value": round(mpix, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": f"{frames} full 3840x2160 Tiger frames (tileKernel+renderKernel restatement, oracle/), {el:.1f} s on 1 of {os.cpu_count()} host cores",
    }
    # the same port with renderKernel's tile rows spread over the host cores (ctypes drops
    # the GIL); the tileKernel restatement stays on one thread
    from concurrent.futures import ThreadPoolExecutor

    cores = max(1, min(os.cpu_count() or 1, 64))
    tiles_y = (h + 15) // 16
    cuts = [tiles_y * i // (4 * cores) for i in range(4 * cores + 1)]
    frames, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while True:
            P = pmo.Ptcl(scene, w, h)
            list(ex.map(lambda ab: P.render_rows(ab[0], ab[1]), [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]))
            P.close()
            frames += 1
            el = time.perf_counter() - t0
            if el > seconds_budget / 2 or frames >= 16:
                break
    out["all_cores"] = {"value": round(w * h * frames / el / 1e6, 3), "unit": "Mpix/s", "cores": cores,
                        "sample": f"{frames} frames, render rows on {cores} threads, tile pass on 1, {el:.1f} s"}
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--gather", action="store_true", help="N>1: gather the bands to rank 0 inside every timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump", default=None, help="rank 0 saves the last gathered frame as .npy (tests)")
    args = ap.parse_args()

    import torch

    import piet_metal_amd as pm
    from piet_metal_amd import dist as pmd

    rank, world, local = pmd.env_rank_world()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run", file=sys.stderr)
            return 2
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: there is no CPU fallback for the product path", file=sys.stderr)
        return 2
    # Rehearsal of the N>1 path on a 1-GPU box (tests/test_gpu_parity.py): every rank on
    # device 0 and gloo as the transport.  The driver's multi-GPU runs never set these.
    if os.environ.get("PM_BENCH_SHARE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("PM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            pmd.init_process_group("nccl")
        else:
            os.environ["LOCAL_RANK"] = str(local)
            pmd.init_process_group(backend)
        import torch.distributed as dist

    wl = stacked_tigers(pm, world)
    r = pm.Renderer(local)
    r.resize(wl.width, wl.height)
    scene_bytes, n_items = r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    layout = pmd.band_layout(wl.height, world)
    r0, r1, rows = layout[rank]
    if world > 1:
        r.set_band(r0, r1)
    pad_rows = pmd.padded_band_rows(wl.height, world)
    band = torch.zeros((pad_rows, wl.width, 4), dtype=torch.uint8, device=f"cuda:{local}")
    full = torch.empty((wl.height, wl.width, 4), dtype=torch.uint8, device=f"cuda:{local}") if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream()
    do_gather = world > 1 and args.gather

    def gather_step():
        r.render_to(band, stream)  # caller-owned band on torch's stream: the gather follows in stream order
        pmd.gather_framebuffer(band, wl.height, dst=0, full=full)

    def step():
        if do_gather:
            gather_step()
        else:
            r.render()

    def fence():
        if world > 1:
            dist.barrier()
        r.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_ms = None
    if world > 1:  # the presentation gather, on its own: render into the band tensor + gather
        for _ in range(2):
            gather_step()
        fence()
        tg = time.perf_counter()
        n_g = 10
        for _ in range(n_g):
            gather_step()
        fence()
        t = torch.tensor([(time.perf_counter() - tg) / n_g * 1e3], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather_ms = float(t.item())

    # per-kernel durations with HIP events on the stream the kernels run on (ctx stream)
    # (a) inside the pipelined batch, i.e. under the conditions of the timed region;
    # (b) each kernel alone on the GPU (frames serialized on one stream)
    tm = r.time_frames(max(10, min(args.steps, 2000)), pipelined=True)
    alone = r.time_frames(20)
    lat = r.frame_latency(100)
    n_overlapped, n_serial = args.warmup + args.steps + tm["iters"], alone["iters"] + lat["iters"]
    st = r.stats()
    band_px = wl.width * rows
    total_px = wl.width * wl.height
    ms_per_step = elapsed / args.steps * 1e3
    value = total_px / (elapsed / args.steps) / 1e6

    if rank == 0:
        # algorithmic bytes of one launch of the dominant kernel = one frame of this
        # rank's band: scene read once + every RGBA8 pixel written once
        b_alg = scene_bytes + 4 * band_px
        kernels = {"pm_bin_kernel": tm["bin_ms"], "pm_coarse_kernel": tm["coarse_ms"], "pm_fine_kernel": tm["fine_ms"], "pm_clear_kernel": tm["clear_ms"]}
        dom = max(kernels, key=kernels.get)
        dom_ms = kernels[dom]
        achieved = b_alg / (dom_ms * 1e-3) / 1e9
        alone_ms = {"pm_bin_kernel": alone["bin_ms"], "pm_coarse_kernel": alone["coarse_ms"], "pm_fine_kernel": alone["fine_ms"], "pm_clear_kernel": alone["clear_ms"]}
        if kernels["pm_clear_kernel"] == 0:  # folded into pm_fine_kernel's launch
            kernels.pop("pm_clear_kernel")
            alone_ms.pop("pm_clear_kernel")
        latency_ms = lat["median_ms"]
        pipelined_ms = tm["total_ms"] / tm["iters"]
        traffic, issue = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                prof = json.load(open(tpath))
                traffic = prof.get(dom, {}).get("hbm_bytes_per_launch")
                # The bound this path actually runs against: VALU issue.  A CDNA4 SIMD issues one
                # wave64 VALU instruction per 4 cycles; instruction counts are the SQ counters of
                # the committed rocprofv3 PMC passes (profiles/), one launch of each kernel = one frame.
                insts = sum(prof[k]["valu_insts_per_launch"] for k in kernels if "valu_insts_per_launch" in prof.get(k, {}))
                props = torch.cuda.get_device_properties(local)
                n_simd = props.multi_processor_count * 4
                clock_ghz = getattr(props, "clock_rate", 2400000) / 1e6
                if insts:
                    floor_ms = insts * 4 / n_simd / (clock_ghz * 1e9) * 1e3
                    issue = {"bound": "valu-issue", "valu_wave_insts_per_frame": insts, "simds": n_simd, "cycles_per_inst": 4,
                             "clock_ghz": round(clock_ghz, 3), "floor_ms": round(floor_ms, 5), "frac": round(floor_ms / ms_per_step, 4),
                             "source": prof.get("_source")}
            except Exception:
                traffic, issue = None, None
        out = {
            "metric": "Mpixels/s, Ghostscript Tiger 3840x2160 (fills+strokes)",
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 geometry + f16 accumulators (as the reference)", "data": "synthetic: embedded Ghostscript_Tiger.svg, scale 10.8, flattened on device",
            "config": {
                "workload": "BASELINE config 3: Ghostscript Tiger 3840x2160, fills + strokes" + ("" if world == 1 else f", one Tiger per 2160-row band x {world} (weak scaling)"),
                "viewport": [wl.width, wl.height], "items": n_items, "scene_bytes": scene_bytes,
                "parallelism": "1 GPU" if world == 1 else f"tile-row bands x{world}, scene replicated, " + ("RCCL gather of bands to rank 0 every step" if do_gather else "no collective in the timed region (bands stay resident, as the frame does at N=1)"),
                "gather_ms": None if gather_ms is None else round(gather_ms, 4),
                "queued_tiles_rank0": st["queued_tiles"],
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": b_alg, "kernel_ms": round(dom_ms, 5),
                "kernels_ms": {k: round(v, 5) for k, v in kernels.items()},
                "kernels_alone_ms": {k: round(v, 5) for k, v in alone_ms.items()},
                # what `rocprofv3 --kernel-trace --stats` of this command averages per kernel: the
                # overlapping launches (warm-up + timed steps + the timed batch) at the in-flight
                # duration, the serialized ones (alone pass + latency pass) at the alone duration
                "trace_average_ms": {k: round((n_overlapped * kernels[k] + n_serial * alone_ms[k]) / (n_overlapped + n_serial), 5) for k in kernels},
                "alone_frac": round(b_alg / (max(alone_ms.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "frame_latency_ms": round(latency_ms, 5), "frame_latency_min_ms": round(lat["min_ms"], 5),
                "frame_latency_frac": round(b_alg / (latency_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "frame_pipelined_ms": round(pipelined_ms, 5),
                "frame_pipelined_frac": round(b_alg / (pipelined_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "kernels_ms / kernel_ms: per-launch durations inside the overlapping batch (four frames in flight on four streams, as in the timed region), from events carried by the dispatches -- what rocprofv3 --kernel-trace shows; kernels_alone_ms: the same kernels with frames serialized on one stream; frame_latency_ms: one frame with nothing else in flight, first kernel begin to last kernel end (median of 100; SURVEY 8d's t_frame); the path is latency/VALU bound, not HBM bound (DESIGN.md)",
            },
        }
        if issue is not None:
            out["issue_roofline"] = issue
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pm, pm.workloads.tiger(3840, 2160))
        print(json.dumps(out), flush=True)
    if args.dump and rank == 0:
        import numpy as np

        img = full if world > 1 else band[: wl.height]
        if world > 1 and not do_gather:
            pass  # `full` holds the frame of the separate gather pass above
        if world == 1:
            r.render_to(band, stream)
            torch.cuda.synchronize()
        np.save(args.dump, img.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
