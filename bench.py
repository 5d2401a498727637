#!/usr/bin/env python
"""bench.py -- the metric BASELINE.json names: Mpixels/s on the Ghostscript Tiger at
3840x2160 (fills + strokes, BASELINE config 3) on 1/2/4/8 GPUs, with the HBM-roofline
fraction of the dominant kernel and the CPU oracle timed beside it.

    python bench.py --gpus N --steps K --warmup W

A "step" is one frame of the hot path -- pm_bin_kernel, then pm_fine_kernel (list building fused in)
(tileKernel + renderKernel + composite of the reference) -- over the scene already
resident in HBM (flatten/encode happens once per scene, like the reference encodes once
per resize, PietRenderer.m:145; its cost is reported as scene.*).

Two figures, both top level and both named:
  value               W*H / t_frame, t_frame = ONE frame alone, first kernel begin to last
                      kernel end: two HIP events on the frame's stream around the plain launches
                      pm_render makes (pm_frame_latency), median over the K timed steps' worth of
                      frames (>= 100).  The contract metric.
  sustained_mpix_s    W*H / (wall time of the K timed steps / K): frames submitted back to
                      back without waiting, as the reference commits command buffers
                      (PietRenderer.m:102); up to four frames overlap on four in-order streams.
                      ms_per_step is this wall time per step.

N > 1 (one process per GPU, torch.distributed.run): STRONG scaling of the same 3840x2160
frame -- rank r renders a band of tile rows (cost-balanced cuts from measured band times),
and every step ends with the one collective of the design: the bands are gathered straight
into the final image on rank 0 (one grouped send/recv over RCCL/xGMI).  value is then
W*H / t_frame with t_frame = one step alone (band render + gather) between two events on the
frame's stream -- the GPU's clock, as at N = 1 --, MAX over ranks; the same step host-timed
(t_frame_e2e), t_render (slowest rank's band), t_gather and the sustained end-to-end rate are
reported next to it; config.rccl_lib / rccl_ranks name the RCCL the product's collective is
bound to.  A "config5" block carries the same measurements for BASELINE config 5 (8192^2 grid
of 25 Tigers), where the frame is big enough to shard; `--workload config5` makes it the main
line (also with --gpus N).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


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
    # the same port with the frame cut into slices of tile-group rows spread over the host cores
    # (ctypes drops the GIL): every slice runs the tileKernel restatement for its threadgroups and
    # then renderKernel for their pixels, as independent as the reference's threadgroups are
    from concurrent.futures import ThreadPoolExecutor

    cores = max(1, min(os.cpu_count() or 1, 64))
    groups_y = ((h + 15) // 16 + 1) // 2
    cuts = [groups_y * i // (4 * cores) for i in range(4 * cores + 1)]
    tiles_y = (h + 15) // 16

    def slice_(ab):
        P = pmo.Ptcl(scene, w, h, group_rows=ab)
        P.render_rows(2 * ab[0], min(2 * ab[1], tiles_y))
        P.close()

    frames, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while True:
            list(ex.map(slice_, [(a, b_) for a, b_ in zip(cuts[:-1], cuts[1:]) if b_ > a]))
            frames += 1
            el = time.perf_counter() - t0
            if el > seconds_budget / 2 or frames >= 32:
                break
    out["all_cores"] = {"value": round(w * h * frames / el / 1e6, 3), "unit": "Mpix/s", "cores": cores,
                        "sample": f"{frames} frames, tile pass + render in {4 * cores} slices of tile-group rows on {cores} threads, {el:.1f} s"}
    return out


class Job:
    """One workload on this rank's GPU: scene resident, band set, buffers for the gather."""

    def __init__(self, pm, pmd, torch, dist, r, wl, rank, world, local, args):
        self.pm, self.pmd, self.torch, self.dist = pm, pmd, torch, dist
        self.r, self.wl, self.rank, self.world, self.local, self.args = r, wl, rank, world, local, args
        t0 = time.perf_counter()
        r.resize(wl.width, wl.height)
        t1 = time.perf_counter()
        self.scene_bytes, self.n_items = r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        t2 = time.perf_counter()
        r.render()
        t3 = time.perf_counter()
        r.sync()
        t4 = time.perf_counter()
        self.first_frame_ms = (t4 - t0) * 1e3  # fused: flatten + encode + index + arena + frame 1
        self.scene_t = r.scene_timings()
        # ... and where it went (host wall clock, the four calls one after the other: they sum to first_frame_ms)
        self.first_frame_parts = {
            "resize_ms": round((t1 - t0) * 1e3, 4), "flatten_and_encode_ms": round((t2 - t1) * 1e3, 4),
            "submit_ms": round((t3 - t2) * 1e3, 4), "of_submit_binning_plan_ms": round(self.scene_t["arena_setup_ms"], 4),
            "frame_until_synced_ms": round((t4 - t3) * 1e3, 4),
        }
        self.cuts = None
        self.layout = pmd.band_layout(wl.height, world)
        self.band = self.full = self.pad = self.scratch = None
        # The step's stream: torch's current stream -- made a stream of its own first.  The legacy default stream's handle is 0,
        # which pm_render_to / pm_gather read as "the context's own stream": the frame would then run beside torch's collectives
        # and copies instead of in front of them (found when the dumped frame of the gloo rehearsal was zeroed before its step).
        if world > 1 and torch.cuda.current_stream().cuda_stream == 0:
            torch.cuda.set_stream(torch.cuda.Stream(device=f"cuda:{local}"))
        self.stream = torch.cuda.current_stream()
        self.balance_log = []
        self.comm, self.gather_impl = None, args.gather_impl
        self.chunks = []
        if world > 1:
            self._set_cuts([b[0] for b in self.layout] + [self.layout[-1][1]])
            if not args.equal_bands:
                for _ in range(3):
                    ms = self._all_band_ms()
                    self.balance_log.append({"cuts": list(self.cuts), "band_ms": [round(v, 4) for v in ms]})
                    new = pmd.balanced_cuts(self.cuts, ms)
                    if new == self.cuts:
                        break
                    self._set_cuts(new)
            self._alloc()
            self._setup_cabi()
            self._setup_chunks()

    # ---- the product's own collective (pm_comm_* / pm_gather): id from rank 0 through the torch store ----
    def _setup_cabi(self):
        """(the agreement itself: piet_metal_amd.dist.agree_on_c_abi_gather -- every rank ends with one answer, and the line is
        printed either way; tests/test_dist_cpu.py runs it across three processes with a RCCL whose ncclCommInitRank fails)"""
        torch, dist = self.torch, self.dist

        def all_min(v):
            t = torch.tensor([v], dtype=torch.int32, device=self.band.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())

        def broadcast_id(make_id):
            box = [make_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        def try_exchange(comm):
            self.r.render_to(self.band, self.stream)
            comm.gather(self.layout, root=0, full=self.full, band=self.band, stream=self.stream)  # one exchange: it works or it does not
            torch.cuda.synchronize()

        self.gather_impl, self.comm, why = self.pmd.agree_on_c_abi_gather(
            self.args.gather_impl, self.world, self.pm.Comm.unique_id, broadcast_id,
            lambda uid: self.pm.Comm(self.r, uid, self.rank, self.world), try_exchange, all_min)
        self.gather_fallback = why
        if why and self.rank == 0:
            print(f"bench.py: pm_gather not usable here ({why}): falling back to torch.distributed send/recv", file=sys.stderr)

    # ---- the gather pipelined under the render (--gather-chunks K) ---------------------------
    def _setup_chunks(self):
        """K sub-bands per rank, a renderer context each (a context's band is part of its binning plan: changing it per step
        would re-plan per step); sub-band k is rendered on the step's stream, its gather goes to a second stream behind an
        event, sub-band k + 1 is rendered meanwhile.  The step ends when both streams have: the step's stream waits for the
        exchange stream."""
        self.chunks = []
        k_req = max(1, int(getattr(self.args, "gather_chunks", 1)))
        if self.world == 1 or k_req == 1 or self.gather_impl == "allgather":
            return
        torch = self.torch
        self.xstream = torch.cuda.Stream(device=self.band.device)
        r0 = self.layout[self.rank][0]
        scene_bytes = None
        for lay in self.pmd.sub_band_layouts(self.layout, self.wl.height, k_req):
            s0, s1, srows = lay[self.rank]
            q = None
            if s1 > s0:
                # A sub-band's context is LEAN: nothing reserved up front (a context's reservations are for a whole frame: 4.4 GB;
                # K of them would multiply a rank's memory by K -- these allocate what their band needs when it is first
                # rendered), and it takes the scene the rank's own context flattened instead of flattening it again.
                keep = os.environ.get("PM_PREALLOC")
                os.environ["PM_PREALLOC"] = "0"
                try:
                    q = self.pm.Renderer(self.local)
                finally:
                    if keep is None:
                        os.environ.pop("PM_PREALLOC", None)
                    else:
                        os.environ["PM_PREALLOC"] = keep
                q.resize(self.wl.width, self.wl.height)
                if scene_bytes is None:
                    scene_bytes = self.r.download_scene()
                q.set_scene_bytes(scene_bytes)
                q.set_band(s0, s1)
            view = self.full[s0 * 16 : s0 * 16 + srows] if self.rank == 0 else self.band[(s0 - r0) * 16 : (s0 - r0) * 16 + srows]
            self.chunks.append((q, lay, view, torch.cuda.Event()))

    def _step_chunked(self):
        torch = self.torch
        self.xstream.wait_stream(self.stream)  # (the previous step's frame has been handed on)
        for q, lay, view, ev in self.chunks:
            if q is not None:
                q.render_to(view, self.stream)
            ev.record(self.stream)
            self.xstream.wait_event(ev)
            if self.gather_impl == "cabi":
                self.comm.gather(lay, root=0, full=self.full, band=view if q is not None else self.band, stream=self.xstream, renderer=q or self.r)
            else:
                with torch.cuda.stream(self.xstream):
                    self.pmd.gather_bands(view, lay, self.wl.height, dst=0, full=self.full)
        self.stream.wait_stream(self.xstream)

    # ---- bands -------------------------------------------------------------------------
    def _set_cuts(self, cuts):
        self.cuts = list(cuts)
        self.layout = self.pmd.band_layout(self.wl.height, self.world, self.cuts)
        r0, r1, _ = self.layout[self.rank]
        self.r.set_band(r0, r1)

    def _all_band_ms(self):
        self.r.render()
        self.r.sync()
        mine = self.r.frame_latency(30)["median_ms"]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=f"cuda:{self.local}")
        t[self.rank] = mine
        self.dist.all_reduce(t)
        return [float(v) for v in t.tolist()]

    def _alloc(self):
        torch, wl = self.torch, self.wl
        dev = f"cuda:{self.local}"
        r0, _r1, rows = self.layout[self.rank]
        if self.rank == 0:
            self.full = torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device=dev)
            self.band = self.full[r0 * 16 : r0 * 16 + rows]  # the root renders straight into the final image
        else:
            self.band = torch.zeros((max(rows, 1), wl.width, 4), dtype=torch.uint8, device=dev)

    # ---- one step ----------------------------------------------------------------------
    def step(self):
        if self.world == 1:
            self.r.render()
            return
        if self.chunks:
            self._step_chunked()
            return
        self.r.render_to(self.band, self.stream)  # caller-owned band on torch's stream: the gather follows in stream order
        self.gather()

    def gather(self):
        if self.gather_impl == "cabi":
            self.comm.gather(self.layout, root=0, full=self.full, band=self.band, stream=self.stream)
        elif self.gather_impl == "allgather":
            if self.pad is None:
                rows = self.pmd.padded_band_rows(self.wl.height, self.world, self.cuts)
                self.pad = self.torch.zeros((rows, self.wl.width, 4), dtype=self.torch.uint8, device=self.band.device)
                self.scratch = self.torch.empty((self.world,) + tuple(self.pad.shape), dtype=self.torch.uint8, device=self.band.device)
                if self.full is None:
                    self.full = self.torch.zeros((self.wl.height, self.wl.width, 4), dtype=self.torch.uint8, device=self.band.device)
            n = self.layout[self.rank][2]
            self.pad[:n] = self.band[:n]
            self.pmd.allgather_bands(self.pad, self.layout, self.wl.height, full=self.full, scratch=self.scratch)
        else:
            self.pmd.gather_bands(self.band, self.layout, self.wl.height, dst=0, full=self.full)

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        for q, *_ in self.chunks:
            if q is not None:
                q.sync()
        self.r.sync()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, v: float) -> float:
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=f"cuda:{self.local}")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    # ---- measurements --------------------------------------------------------------------
    def sustained(self, steps: int, warmup: int, precondition: int):
        """(elapsed seconds of `steps` steps, MAX over ranks).  `precondition` untimed steps
        bring clocks, queues and caches to the state a long run has before the counted warm-up."""
        for _ in range(precondition):
            self.step()
        self.fence()
        for _ in range(warmup):
            self.step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        return self.max_over_ranks(time.perf_counter() - t0)

    def lone_frame_host_ms(self, n: int) -> float:
        """One step alone, host-timed around submit + device sync: median, MAX over ranks."""
        ts = []
        for _ in range(n):
            self.fence()
            t0 = time.perf_counter()
            self.step()
            self.r.sync()
            self.torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return self.max_over_ranks(statistics.median(ts))

    def lone_frame_event_ms(self, n: int) -> float:
        """One step alone on the GPU's own clock: two events on the frame's stream around band render +
        gather (N > 1: both run on torch's current stream) -- the clock pm_frame_latency uses at N = 1, so
        that `value` means the same at every N.  Median, MAX over ranks."""
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(n):
            self.fence()
            e0.record(self.stream)
            self.step()
            e1.record(self.stream)
            self.r.sync()
            self.torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return self.max_over_ranks(statistics.median(ts))

    def rccl_info(self) -> dict:
        """Which RCCL the product's collective is bound to, and the size of its communicator.  At N = 1 a
        one-rank communicator is made for the purpose (outside every timed region)."""
        out = {"rccl_lib": None, "rccl_ranks": None}
        try:
            if self.comm is not None:
                out.update(self.comm.info())
            elif self.world == 1:
                comm = self.pm.Comm(self.r, self.pm.Comm.unique_id(), 0, 1)
                out.update(comm.info())
                comm.close()
            else:
                out["rccl_lib"] = self.pm.Comm.library_path()
        except Exception as e:  # noqa: BLE001 -- a report field, never a reason to lose the line
            out["rccl_error"] = repr(e)[:200]
        return out

    def gather_alone_ms(self, n: int) -> float:
        if self.world == 1:
            return 0.0
        ts = []
        for _ in range(n):
            self.fence()
            t0 = time.perf_counter()
            self.gather()
            self.torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return self.max_over_ranks(statistics.median(ts))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--equal-bands", action="store_true", help="N>1: near-equal tile-row split instead of cost-balanced cuts")
    ap.add_argument("--gather-impl", choices=["auto", "cabi", "sendrecv", "allgather"], default="auto",
                    help="N>1: cabi = the product's own collective behind the C ABI (pm_comm_create / pm_gather: grouped RCCL send/recv "
                         "into the final image); sendrecv = the same exchange through torch.distributed; allgather = padded all-gather; "
                         "auto (default) = cabi, falling back to sendrecv on every rank if any rank cannot set it up")
    ap.add_argument("--gather-chunks", type=int, default=1,
                    help="N>1: every rank renders its band in this many sub-bands (a context each) and posts each sub-band's gather on a "
                         "second stream while it renders the next: the exchange runs under the render instead of behind it")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE config 5 block")
    ap.add_argument("--workload", choices=["config2", "config3", "config4", "config5", "held1", "held2", "held3"], default="config3",
                    help="the main line's workload (default: BASELINE config 3, the one the metric is quoted on; the others are for profiles/)")
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
    dist = None
    if world > 1:
        if backend == "nccl":
            pmd.init_process_group("nccl")
        else:
            os.environ["LOCAL_RANK"] = str(local)
            pmd.init_process_group(backend)
        import torch.distributed as dist
    staged = backend != "nccl" and world > 1  # gloo rehearsal: bands travel through host tensors
    if staged:
        _patch_for_host_transport(pmd, torch)
        if args.gather_impl in ("auto", "cabi"):
            args.gather_impl = "sendrecv"  # (the rehearsal's ranks share one GPU: no RCCL communicator between them)

    r = pm.Renderer(local)
    held = pm.workloads.heldout_workloads()
    wl = {"config2": lambda: pm.workloads.tiger(1920, 1080, fills_only=True), "config3": lambda: pm.workloads.tiger(3840, 2160),
          "config4": pm.workloads.config4_blobs, "config5": pm.workloads.config5_tiger_grid,
          "held1": lambda: held["held1"], "held2": lambda: held["held2"], "held3": lambda: held["held3"]}[args.workload]()
    workload_name = {"config2": "BASELINE config 2: Ghostscript Tiger 1920x1080, solid fills only",
                     "config3": "BASELINE config 3: Ghostscript Tiger 3840x2160, fills + strokes",
                     "config4": "BASELINE config 4: 10k overlapping cubic-Bezier paths, 4096x4096",
                     "config5": "BASELINE config 5: 5x5 Tigers at scale 8, 8192x8192",
                     "held1": "held-out 1: Ghostscript Tiger 2560x1440, fills + strokes",
                     "held2": "held-out 2: 2k overlapping cubic-Bezier paths, 2048x2048",
                     "held3": "held-out 3: 20k glyph-like closed paths, 3840x2160"}[args.workload]
    if args.workload != "config3":
        args.no_config5 = True
        args.no_cpu_baseline = True
    job = Job(pm, pmd, torch, dist, r, wl, rank, world, local, args)
    W, H = wl.width, wl.height
    px = W * H

    precondition = max(0, min(300, 3000 // max(1, world * world)))
    elapsed = job.sustained(args.steps, args.warmup, precondition)
    ms_per_step = elapsed / args.steps * 1e3
    sustained = px / (elapsed / args.steps) / 1e6

    n_lat = max(100, min(args.steps, 400))
    t_frame_host = job.lone_frame_host_ms(min(n_lat, 100))
    if world == 1:
        lat = r.frame_latency(n_lat)
        t_frame = lat["median_ms"]
        t_render, t_gather = t_frame, 0.0
        tm = r.time_frames(max(10, min(args.steps, 2000)), pipelined=True)
        alone = r.time_frames(20)
        n_overlapped, n_serial = precondition + args.warmup + args.steps + tm["iters"], alone["iters"] + lat["iters"] + min(n_lat, 100) + 1
    else:
        # band render alone (kernel timestamps), slowest rank; the gather alone; one whole step alone
        job.r.render_to(job.band, job.stream)
        job.fence()
        lat = r.frame_latency(n_lat)
        t_render = job.max_over_ranks(lat["median_ms"])
        t_gather = job.gather_alone_ms(30)
        t_frame = job.lone_frame_event_ms(min(n_lat, 100))
        tm = r.time_frames(max(10, min(args.steps, 500)), pipelined=True)
        alone = r.time_frames(20)
        n_overlapped = n_serial = 1
    st = r.stats()
    value = px / (t_frame * 1e-3) / 1e6
    rccl = job.rccl_info()

    cfg5 = None
    if not args.no_config5:
        cfg5 = config5_block(pm, pmd, torch, dist, r, rank, world, local, args)

    if rank == 0:
        rows = job.layout[rank][2]
        b_alg = job.scene_bytes + 4 * W * rows  # this rank's launch: scene read once + its pixels written once
        b_alg_frame = world * job.scene_bytes + 4 * px
        kernels = {"pm_bin_kernel": tm["bin_ms"], "pm_coarse_kernel": tm["coarse_ms"], "pm_fine_kernel": tm["fine_ms"], "pm_clear_kernel": tm["clear_ms"]}
        alone_ms = {"pm_bin_kernel": alone["bin_ms"], "pm_coarse_kernel": alone["coarse_ms"], "pm_fine_kernel": alone["fine_ms"], "pm_clear_kernel": alone["clear_ms"]}
        if kernels["pm_clear_kernel"] == 0:  # folded into pm_fine_kernel's launch
            kernels.pop("pm_clear_kernel")
        if alone_ms["pm_clear_kernel"] == 0:  # (a frame alone folds it by default, frames behind others do not)
            alone_ms.pop("pm_clear_kernel")
        if kernels["pm_coarse_kernel"] == 0:  # fused: pm_fine_kernel<true> builds each tile's list itself
            kernels.pop("pm_coarse_kernel")
            alone_ms.pop("pm_coarse_kernel")
        # the dominant kernel and its duration: kernels serialized on one stream, every launch alone on the
        # GPU (a duration that fits inside a step; with four frames in flight the launches stretch each
        # other and their in-flight durations exceed ms_per_step: kept below as *_inflight)
        dom = max(alone_ms, key=alone_ms.get)
        dom_ms = alone_ms[dom]
        achieved = b_alg / (dom_ms * 1e-3) / 1e9
        dom_inflight_ms = kernels[dom]
        pipelined_ms = tm["total_ms"] / tm["iters"]
        traffic, traffic_frame, issue, traffic_meta = None, None, None, None
        # (one committed PMC digest per workload: the Tiger's is hbm_traffic.json, the others' hbm_traffic_<workload>.json)
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json" if args.workload == "config3" else f"hbm_traffic_{args.workload}.json")
        if os.path.exists(tpath) and world == 1:
            try:
                prof = json.load(open(tpath))
                # the PMC figures are copied from a committed profile, not measured in this run: say which
                # kernels they were taken on, and whether those are the kernels running now
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from make_traffic import kernel_sources_sha16

                now = kernel_sources_sha16()
                traffic_meta = {"source": prof.get("_source"), "measured_at_commit": prof.get("_commit"),
                                "kernel_sources_sha16": prof.get("_kernel_sources_sha16"), "kernel_sources_sha16_now": now,
                                "stale": prof.get("_kernel_sources_sha16") != now}
                if traffic_meta["stale"]:
                    print(f"bench.py: profiles/hbm_traffic.json was measured on other kernel sources ({prof.get('_kernel_sources_sha16')} at commit "
                          f"{prof.get('_commit')}, now {now}): roofline.traffic / issue_roofline are from that profile", file=sys.stderr)
                traffic = prof.get(dom, {}).get("hbm_bytes_per_launch")
                # (the counted runs pin PM_FOLD_CLEAR=1, tools/prof_round.sh: a frame is the launches a frame ALONE
                #  makes, clearing inside the tile kernel's -- a separate pm_clear_kernel moves the same bytes)
                counted = [k for k in alone_ms if k != "pm_clear_kernel"]
                per_kernel = [prof[k]["hbm_bytes_per_launch"] for k in counted if "hbm_bytes_per_launch" in prof.get(k, {})]
                if len(per_kernel) == len(counted):
                    traffic_frame = {"hbm_bytes_per_frame": int(sum(per_kernel)), "ratio_to_algorithmic": round(sum(per_kernel) / b_alg, 3)}
                insts = sum(prof[k]["valu_insts_per_launch"] for k in counted if "valu_insts_per_launch" in prof.get(k, {}))
                props = torch.cuda.get_device_properties(local)
                n_simd = props.multi_processor_count * 4
                clock_ghz = getattr(props, "clock_rate", 2400000) / 1e6
                if insts:
                    # CDNA4 SIMDs are 32 lanes wide: a wave64 VALU instruction issues in 2 cycles
                    # (MI355X_MICROARCH.md, per-instruction constants); counts = SQ_INSTS_VALU of
                    # the committed PMC passes, one launch of each kernel = one frame
                    floor_ms = insts * 2 / n_simd / (clock_ghz * 1e9) * 1e3
                    issue = {"bound": "valu-issue", "valu_wave_insts_per_frame": insts, "simds": n_simd, "cycles_per_inst": 2,
                             "clock_ghz": round(clock_ghz, 3), "floor_ms": round(floor_ms, 5),
                             "frac_sustained": round(floor_ms / ms_per_step, 4), "frac_lone_frame": round(floor_ms / t_frame, 4),
                             "source": prof.get("_source")}
            except Exception:
                traffic, traffic_frame, issue = None, None, None
        out = {
            "metric": "Mpixels/s, Ghostscript Tiger 3840x2160 (fills+strokes): W*H / t_frame" if args.workload == "config3" else f"Mpixels/s, {workload_name}: W*H / t_frame",
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "t_frame_ms": round(t_frame, 5),
            "value_definition": ("W*H / t_frame; t_frame = one frame alone, first kernel begin to last kernel end: two HIP events on the frame's stream "
                                 f"around the plain launches pm_render makes (pm_frame_latency), median of {lat['iters']} (SURVEY.md 8d)") if world == 1 else
                                ("W*H / t_frame; t_frame = one step alone -- band render + gather into the final image on rank 0 -- between two events on "
                                 "the frame's stream (the GPU clock of the N=1 figure), median, MAX over ranks; t_frame_host_ms / config.t_frame_e2e_ms is the "
                                 "same step host-timed incl. launch latency and the device sync"),
            "sustained_mpix_s": round(sustained, 1), "ms_per_step": round(ms_per_step, 5),
            "sustained_definition": "W*H / (wall time of the K timed steps / K), steps submitted back to back, barrier + device sync on both sides, MAX over ranks"
                                    + ("" if world == 1 else "; every step ends with the gather to rank 0"),
            "t_frame_host_ms": round(t_frame_host, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 geometry + f16 accumulators (as the reference)", "data": {"config2": "synthetic: embedded Ghostscript_Tiger.svg, scale 5.4, fills only, flattened on device",
                                                                                "config3": "synthetic: embedded Ghostscript_Tiger.svg, scale 10.8, flattened on device",
                                                                                "config4": "synthetic: 10 000 random closed cubic paths (SplitMix64 seed 0x5EED0004), flattened on device",
                                                                                "config5": "synthetic: 5x5 grid of the embedded Ghostscript_Tiger.svg at scale 8, flattened on device",
                                                                                "held1": "synthetic: embedded Ghostscript_Tiger.svg, scale 7.2, flattened on device",
                                                                                "held2": "synthetic: 2 000 random closed cubic paths (SplitMix64 seed 0x5EED0008), flattened on device",
                                                                                "held3": "synthetic: 20 000 glyph-like closed paths (SplitMix64 seed 0x5EED0007), flattened on device"}[args.workload],
            "config": {
                "workload": workload_name,
                "viewport": [W, H], "items": job.n_items, "scene_bytes": job.scene_bytes,
                "parallelism": "1 GPU" if world == 1 else f"tile-row bands x{world} ({'near-equal' if args.equal_bands else 'cost-balanced'} cuts), scene replicated, "
                               f"{ {'cabi': 'pm_gather (C ABI: grouped RCCL send/recv)', 'sendrecv': 'grouped send/recv (torch.distributed)', 'allgather': 'padded all-gather'}.get(job.gather_impl, job.gather_impl) } "
                               "of the bands into the final image on rank 0 in every step",
                "gather_impl": None if world == 1 else job.gather_impl,
                "gather_chunks": None if world == 1 else max(1, len(job.chunks)),
                "rccl_lib": rccl.get("rccl_lib"), "rccl_ranks": rccl.get("rccl_ranks"), **({"rccl_error": rccl["rccl_error"]} if "rccl_error" in rccl else {}),
                "band_cuts": job.cuts, "balance": job.balance_log or None,
                "t_render_ms": round(t_render, 5), "t_gather_ms": round(t_gather, 5), "t_frame_e2e_ms": round(t_frame_host, 5),
                # (what t_gather cannot beat: the largest non-root band over one xGMI link at ~153 GB/s; launch latency comes on top)
                "t_gather_wire_floor_ms": None if world == 1 else round(pmd.gather_wire_floor_ms(job.layout, W, world), 5),
                **({"gather_fallback": job.gather_fallback} if world > 1 and getattr(job, "gather_fallback", "") else {}),
                "queued_tiles_rank0": st["queued_tiles"], "precondition_steps": precondition,
            },
            "scene": {
                "flatten_encode_ms": round(job.scene_t["flatten_encode_ms"], 4), "scene_index_host_ms": round(job.scene_t["scene_index_ms"], 4),
                "arena_setup_host_ms": round(job.scene_t["arena_setup_ms"], 4), "first_frame_ms": round(job.first_frame_ms, 4),
                "first_frame_parts": job.first_frame_parts,
                "note": "once per scene, like the reference encodes once per resize (PietRenderer.m:145): flatten_encode = pm_flatten_and_encode's four kernels + "
                        "their read-backs; scene_index = header/item read-back, validation, pm_index_kernel; arena_setup = per-(scene, viewport) sizing on the host, "
                        "paid by the first frame; first_frame = all of it fused, resize -> first frame complete (host wall clock)",
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_frame": traffic_frame, "traffic_profile": traffic_meta,
                "algorithmic_bytes_per_launch": b_alg, "kernel_ms": round(dom_ms, 5),
                "kernels_alone_ms": {k: round(v, 5) for k, v in alone_ms.items()},
                "kernels_inflight_ms": {k: round(v, 5) for k, v in kernels.items()},
                "frac_inflight": round(b_alg / (dom_inflight_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "frac_serial_frame": round(b_alg / (sum(alone_ms.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "trace_average_ms": {k: round((n_overlapped * kernels[k] + n_serial * alone_ms.get(k, 0.0)) / (n_overlapped + (n_serial if k in alone_ms else 0)), 5) for k in kernels},
                "trace_average_note": "what the AverageNs column of a kernel trace of THIS command shows: the mix of in-flight and serialized launches it makes",
                "frame_latency_ms": round(t_render, 5),
                "frac_frame": round(b_alg_frame / world / (t_render * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "frame_pipelined_ms": round(pipelined_ms, 5),
                "frac_frame_pipelined": round(b_alg / (pipelined_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "frac credits the whole frame's bytes to ONE of the frame's kernels; frac_serial_frame = the same bytes / the SUM of the kernels' serialized "
                        "durations (the frame's kernels back to back, no gaps) and frac_frame = / the lone frame: read those as the path's fraction of HBM peak. "
                        "achieved/frac: algorithmic bytes of one launch / the dominant kernel's average launch duration with the kernels serialized on one "
                        "stream, events carried by the dispatches (what a rocprofv3 --kernel-trace of PM_FRAME_STREAMS=1 PM_SLOTS=1 shows: "
                        "profiles/*_serial_kernel_stats.csv); frac_inflight / kernels_inflight_ms: the same launches inside the overlapping batch of the "
                        "timed steps (four frames in flight: the durations stretch each other and exceed ms_per_step; what the default kernel trace "
                        "shows); frac_frame: B_alg(N)/N / the slowest rank's lone-frame time (SURVEY 8d multi-GPU roofline); frac_frame_pipelined: "
                        "B_alg / ms_per_step; the path is latency bound, not HBM bound (DESIGN.md)",
            },
        }
        if issue is not None:
            out["issue_roofline"] = issue
        if cfg5 is not None:
            out["config5"] = cfg5
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pm, pm.workloads.tiger(3840, 2160))
        # (RCCL writes a version banner to C stdout when a communicator is made; through a pipe it would come out
        #  at exit, BEHIND the line: everything buffered so far goes first, the JSON line is the last one printed)
        sys.stdout.flush()
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if args.dump:
        # the frame of one more step of the main workload (config 5 may have run in between)
        import numpy as np

        job2 = job if args.no_config5 else Job(pm, pmd, torch, dist, r, wl, rank, world, local, args)
        if world > 1:
            job2.fence()
            if rank == 0:
                job2.full.zero_()  # (what is saved is what THIS step gathered)
            job2.fence()
        job2.step()
        job2.fence()
        if rank == 0:
            img = job2.full if world > 1 else torch.from_numpy(r.read_pixels())
            np.save(args.dump, img.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()
    return 0


def config5_block(pm, pmd, torch, dist, r, rank, world, local, args):
    """BASELINE config 5: 8192^2 grid of 25 Tigers, tile rows sharded across the ranks, RCCL
    framebuffer gather -- the same measurements as the main line, fewer iterations."""
    wl = pm.workloads.config5_tiger_grid()
    job = Job(pm, pmd, torch, dist, r, wl, rank, world, local, args)
    px = wl.width * wl.height
    steps = 40
    elapsed = job.sustained(steps, 5, 10)
    if world > 1:
        job.r.render_to(job.band, job.stream)
        job.fence()
    lat = r.frame_latency(30)
    t_render = job.max_over_ranks(lat["median_ms"])
    t_gather = job.gather_alone_ms(10)
    t_e2e = job.lone_frame_host_ms(20)
    t_frame = t_render if world == 1 else job.lone_frame_event_ms(20)  # (the GPU's clock at every N, like the main line)
    b_alg = world * job.scene_bytes + 4 * px
    return {
        "workload": "BASELINE config 5: 5x5 Tigers at scale 8, 8192x8192" + ("" if world == 1 else f", tile rows on {world} GPUs + gather to rank 0"),
        "items": job.n_items, "scene_bytes": job.scene_bytes, "band_cuts": job.cuts,
        "value": round(px / (t_frame * 1e-3) / 1e6, 1), "t_frame_ms": round(t_frame, 4),
        "t_render_ms": round(t_render, 4), "t_gather_ms": round(t_gather, 4), "t_frame_e2e_ms": round(t_e2e, 4),
        "t_gather_wire_floor_ms": None if world == 1 else round(pmd.gather_wire_floor_ms(job.layout, job.wl.width, world), 4),
        "sustained_mpix_s": round(px / (elapsed / steps) / 1e6, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
        "frac_frame": round(b_alg / world / (t_render * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "gather_GBs_into_root": None if world == 1 or t_gather <= 0 else round(4 * px * (world - 1) / world / (t_gather * 1e-3) / 1e9, 1),
        "flatten_encode_ms": round(job.scene_t["flatten_encode_ms"], 3), "scene_index_host_ms": round(job.scene_t["scene_index_ms"], 3),
        "arena_setup_host_ms": round(job.scene_t["arena_setup_ms"], 3), "first_frame_ms": round(job.first_frame_ms, 3),
        "first_frame_parts": job.first_frame_parts,
    }


def _patch_for_host_transport(pmd, torch):
    """gloo rehearsal on one GPU (tests only): the collectives take CPU tensors, so bands are
    staged through the host around the very same gather_bands / allgather_bands calls."""
    g0, a0 = pmd.gather_bands, pmd.allgather_bands

    def gather_bands(band, layout, height, dst=0, full=None):
        out = g0(band.cpu(), layout, height, dst=dst, full=None if full is None else full.cpu())
        if out is not None and full is not None:
            full.copy_(out)
        return full if out is not None else None

    def allgather_bands(pad, layout, height, full=None, scratch=None):
        out = a0(pad.cpu(), layout, height)
        if full is not None:
            full.copy_(out)
        return full

    pmd.gather_bands, pmd.allgather_bands = gather_bands, allgather_bands


if __name__ == "__main__":
    sys.exit(main())
