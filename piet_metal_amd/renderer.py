"""Renderer: the host-side mirror of PietRenderer (TestApp/PietRenderer.{h,m}).

    PietRenderer                          piet_metal_amd.Renderer
    -initWithMetalKitView:                Renderer(device)
    -mtkView:drawableSizeWillChange:      resize(w, h)          (+ set_band for multi-GPU)
    initScene / init_test_scene           scene_buffer() + upload_scene(), or
                                          flatten_and_encode(paths, affine)
    -drawInMTKView:                       render() / render_to(tensor)

All work happens in libpiet_metal_amd.so (HIP kernels); this file only moves
pointers.  torch is optional here and only used as a device-memory / stream
provider (render_to) -- plumbing, not the product.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .encoder import PathSet


_warned_default_stream = False


def _warn_default_stream() -> None:
    global _warned_default_stream
    if not _warned_default_stream:
        _warned_default_stream = True
        import warnings

        warnings.warn("piet_metal_amd: torch's default stream (handle 0) cannot be named through the C ABI -- the frame ran on the context's "
                      "own stream and was waited for; pass a torch.cuda.Stream() to keep the submission asynchronous", RuntimeWarning, stacklevel=3)


class Renderer:
    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        err = C.c_int(0)
        self._h = self._lib.pm_create(device, C.byref(err))
        if not self._h:
            raise _lib.PietMetalError(err.value, "pm_create")
        self.device = device
        self.width = self.height = 0
        self.row0 = self.row1 = 0

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- viewport ---------------------------------------------------------------
    def resize(self, width: int, height: int) -> None:
        _lib.check(self._lib.pm_resize(self._h, width, height), "pm_resize")
        self.width, self.height = width, height
        self.row0, self.row1 = 0, (height + 15) // 16

    def set_band(self, tile_row0: int, tile_row1: int) -> None:
        _lib.check(self._lib.pm_set_band(self._h, tile_row0, tile_row1), "pm_set_band")
        self.row0, self.row1 = tile_row0, tile_row1

    @property
    def band_pixel_rows(self) -> int:
        return min(self.row1 * 16, self.height) - self.row0 * 16

    # ---- scene --------------------------------------------------------------------
    def scene_buffer(self) -> np.ndarray:
        """Pinned host staging buffer (the reference's _sceneBuf.contents)."""
        cap = C.c_size_t(0)
        p = self._lib.pm_scene_buffer(self._h, C.byref(cap))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cap.value,))

    def reserve_scene(self, nbytes: int) -> None:
        _lib.check(self._lib.pm_scene_reserve(self._h, nbytes), "pm_scene_reserve")

    def upload_scene(self, nbytes: int) -> None:
        _lib.check(self._lib.pm_upload_scene(self._h, nbytes), "pm_upload_scene")

    def set_scene_bytes(self, data: bytes | np.ndarray) -> None:
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        if a.size > self.scene_buffer().size:
            self.reserve_scene(int(a.size))
        self.scene_buffer()[: a.size] = a
        self.upload_scene(int(a.size))

    def flatten_and_encode(self, paths: PathSet, affine, width_scale: float) -> tuple[int, int]:
        """On-device flatten + encode; returns (scene_bytes, n_items)."""
        aff = (C.c_double * 6)(*[float(v) for v in affine])
        nbytes, nitems = C.c_size_t(0), C.c_uint32(0)
        _lib.check(
            self._lib.pm_flatten_and_encode(
                self._h, paths.paths.ctypes.data, len(paths.paths), paths.els.ctypes.data, len(paths.els), aff,
                float(width_scale), C.byref(nbytes), C.byref(nitems),
            ),
            "pm_flatten_and_encode",
        )
        return nbytes.value, nitems.value

    def reflatten(self, affine, width_scale: float) -> tuple[int, int]:
        """Re-flatten the resident paths of the last flatten_and_encode under a new affine
        (animation: no upload, no allocation); returns (scene_bytes, n_items)."""
        aff = (C.c_double * 6)(*[float(v) for v in affine])
        nbytes, nitems = C.c_size_t(0), C.c_uint32(0)
        _lib.check(self._lib.pm_reflatten(self._h, aff, float(width_scale), C.byref(nbytes), C.byref(nitems)), "pm_reflatten")
        return nbytes.value, nitems.value

    def download_scene(self) -> np.ndarray:
        nbytes = C.c_size_t(0)
        self._lib.pm_scene_device_ptr(self._h, C.byref(nbytes))
        out = np.zeros(max(nbytes.value, 8), np.uint8)
        _lib.check(self._lib.pm_download_scene(self._h, out.ctypes.data, out.size, C.byref(nbytes)), "pm_download_scene")
        return out[: nbytes.value]

    # ---- frames ---------------------------------------------------------------------
    def render(self) -> None:
        _lib.check(self._lib.pm_render(self._h), "pm_render")

    def render_to(self, tensor, stream=None) -> None:
        """Render into a torch uint8 CUDA tensor of shape [band_rows, width, 4], on `stream` (a torch stream) -- None, or
        torch's legacy default stream (handle 0), means the context's OWN stream: order later work on torch's side with
        sync(), or make the current stream a torch.cuda.Stream() first (bench.py does)."""
        if tensor.dtype.__str__() != "torch.uint8" or not tensor.is_cuda or tensor.dim() != 3 or tensor.shape[2] != 4:
            raise TypeError("render_to needs a CUDA uint8 tensor [rows, width, 4]")
        if tensor.shape[1] != self.width or tensor.shape[0] < self.band_pixel_rows or tensor.stride(1) != 4 or tensor.stride(2) != 1:
            raise ValueError("tensor does not match the viewport band")
        s = stream.cuda_stream if stream is not None else None
        _lib.check(self._lib.pm_render_to(self._h, tensor.data_ptr(), tensor.stride(0), s), "pm_render_to")
        if stream is not None and not s:
            # torch's legacy default stream has handle 0, which the C ABI reads as "the context's own stream": the frame would run
            # BESIDE the torch work queued on that tensor, not behind it (round-5 advisor).  Not silently: the frame is waited for
            # here, so that whatever the caller queues next on any stream sees it complete, and the caller is told once.
            _warn_default_stream()
            self.sync()

    def sync(self) -> None:
        _lib.check(self._lib.pm_sync(self._h), "pm_sync")

    def set_target_format(self, bgra: bool) -> None:
        """Byte order the kernels store pixels in from now on: RGBA8 (default) or BGRA8, the
        reference drawable's MTLPixelFormatBGRA8Unorm (PietRenderer.m:29)."""
        _lib.check(self._lib.pm_set_target_format(self._h, _lib.PM_FMT_BGRA8 if bgra else _lib.PM_FMT_RGBA8), "pm_set_target_format")

    def read_pixels(self, bgra: bool = False) -> np.ndarray:
        out = np.zeros((self.band_pixel_rows, self.width, 4), np.uint8)
        _lib.check(
            self._lib.pm_read_pixels(self._h, out.ctypes.data, self.width * 4, _lib.PM_FMT_BGRA8 if bgra else _lib.PM_FMT_RGBA8),
            "pm_read_pixels",
        )
        return out

    def time_frames(self, iters: int, per_kernel: bool = True, pipelined: bool = False) -> dict:
        """total_ms: `iters` frames through the frame pipeline.  bin/coarse/fine_ms: average
        launch duration of the three kernels -- each alone on the GPU (default), or inside
        the overlapping pipelined batch (pipelined=True)."""
        tot, k1, k2, k3, k4 = C.c_float(0), C.c_float(0), C.c_float(0), C.c_float(0), C.c_float(0)
        pk = per_kernel
        fn = self._lib.pm_time_frames_pipelined if pipelined else self._lib.pm_time_frames
        _lib.check(
            fn(self._h, iters, C.byref(tot), C.byref(k1) if pk else None, C.byref(k2) if pk else None, C.byref(k3) if pk else None,
               C.byref(k4) if pk else None),
            "pm_time_frames",
        )
        return {"total_ms": tot.value, "bin_ms": k1.value, "coarse_ms": k2.value, "fine_ms": k3.value, "clear_ms": k4.value, "iters": iters}

    def frame_latency(self, iters: int = 100) -> dict:
        """One frame at a time: first kernel's begin to last kernel's end (median / min, ms)."""
        med, mn = C.c_float(0), C.c_float(0)
        _lib.check(self._lib.pm_frame_latency(self._h, iters, C.byref(med), C.byref(mn)), "pm_frame_latency")
        return {"median_ms": med.value, "min_ms": mn.value, "iters": iters}

    def dense_kernel_frames(self) -> int:
        """Frames rendered with the one-wave-per-tile instantiation of the tile kernel (dense scenes)."""
        n = C.c_uint32(0)
        _lib.check(self._lib.pm_tile_kernel_info(self._h, C.byref(n)), "pm_tile_kernel_info")
        return int(n.value)

    def binning_info(self) -> dict:
        """Frames binned with a wave per strip row since pm_create (all / only because they ran behind other frames / of
        those, with a wave for every row of a plan that chains rows)."""
        out = (C.c_uint32 * 3)()
        _lib.check(self._lib.pm_binning_info(self._h, out), "pm_binning_info")
        return {"wave_per_row": int(out[0]), "inflight_only": int(out[1]), "no_chains": int(out[2])}

    def binning_plan_info(self) -> dict:
        """The binning plan in force: entries of the strip-row work list, strip rows cut in two, plans remade from the frames'
        own report since pm_create, and whether the plan in force is such a plan."""
        out = (C.c_uint32 * 4)()
        _lib.check(self._lib.pm_binning_plan_info(self._h, out), "pm_binning_plan_info")
        return {"entries": int(out[0]), "rows_cut": int(out[1]), "plans_fed_back": int(out[2]), "fed_back": bool(out[3])}

    def one_launch_info(self) -> dict:
        """Frames rendered as one launch so far, and whether a lone frame of the resident scene would be."""
        n = C.c_uint32(0)
        a = C.c_int(0)
        _lib.check(self._lib.pm_one_launch_info(self._h, C.byref(n), C.byref(a)), "pm_one_launch_info")
        return {"frames": int(n.value), "applies": bool(a.value)}

    def time_one_launch(self, iters: int = 100) -> float:
        """Average duration (ms) of pm_frame_kernel, frame alone (dispatch-attached events)."""
        ms = C.c_float(0)
        _lib.check(self._lib.pm_time_one_launch(self._h, iters, C.byref(ms)), "pm_time_one_launch")
        return float(ms.value)

    def fill_coverage(self, item_ix: int) -> np.ndarray:
        """f32-accumulated winding coverage of one Fill item over the viewport band (validation)."""
        out = np.zeros((self.band_pixel_rows, self.width), np.float32)
        _lib.check(self._lib.pm_fill_coverage(self._h, item_ix, out.ctypes.data, self.width), "pm_fill_coverage")
        return out

    def scene_timings(self) -> dict:
        """Host wall-clock cost of the last scene replacement (flatten+encode, index, arena)."""
        t = _lib.SceneTimings()
        _lib.check(self._lib.pm_get_scene_timings(self._h, C.byref(t)), "pm_get_scene_timings")
        out = {name: getattr(t, name) for name, _ in t._fields_}
        n = C.c_uint32(0)
        _lib.check(self._lib.pm_get_binning_plans(self._h, C.byref(n)), "pm_get_binning_plans")
        out["binning_plans"] = int(n.value)
        return out

    def frame_timeline(self, iters: int = 100) -> dict:
        """The lone frame taken apart (ms, medians): kernels and the gaps between them."""
        v = (C.c_float * 6)()
        _lib.check(self._lib.pm_debug_frame_timeline(self._h, iters, v), "pm_debug_frame_timeline")
        return dict(zip(("bin_ms", "gap1_ms", "coarse_ms", "gap2_ms", "fine_ms", "total_ms"), [float(x) for x in v]))

    def stats(self) -> dict:
        s = _lib.Stats()
        _lib.check(self._lib.pm_get_stats(self._h, C.byref(s)), "pm_get_stats")
        return {name: getattr(s, name) for name, _ in s._fields_}

    def capture_ptcl(self, max_cmds_per_tile: int = 512):
        """Per-tile command lists of the last frame in the reference's 24-byte layout.
        Returns (counts[rows, tiles_x], solid[rows, tiles_x], cmds[rows, tiles_x, max, 6])."""
        st = self.stats()
        rows, tx = st["band_row1"] - st["band_row0"], st["tiles_x"]
        counts = np.zeros((rows, tx), np.uint32)
        solid = np.zeros((rows, tx), np.uint32)
        cmds = np.zeros((rows, tx, max_cmds_per_tile, 6), np.uint32)
        _lib.check(
            self._lib.pm_debug_capture_ptcl(self._h, max_cmds_per_tile, counts.ctypes.data, solid.ctypes.data, cmds.ctypes.data),
            "pm_debug_capture_ptcl",
        )
        return counts, solid, cmds


class Comm:
    """pm_comm_* / pm_gather: the C-ABI form of the band gather (RCCL bound at run time).
    The 128-byte id comes from rank 0 (Comm.unique_id()) and reaches the other ranks through
    whatever the host has (here: any picklable channel)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().pm_comm_unique_id(buf), "pm_comm_unique_id")
        return bytes(buf)

    def __init__(self, renderer: "Renderer", uid: bytes, rank: int, world: int):
        self._lib = _lib.load()
        self._r = renderer
        self.rank, self.world = rank, world
        err = C.c_int(0)
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._h = self._lib.pm_comm_create(renderer._h, buf, rank, world, C.byref(err))
        if not self._h:
            raise _lib.PietMetalError(err.value, "pm_comm_create")

    def gather(self, layout, root: int = 0, full=None, band=None, stream=None, renderer=None) -> None:
        """layout = [(tile_row0, tile_row1, ...)] per rank; full = torch uint8 [H, W, 4] on the root;
        band = this rank's band tensor (None: the renderer's last frame).  renderer: another context of the same
        device whose band the table names (a rank that renders its band in sub-bands, one context each, gathers every
        sub-band with the one communicator)."""
        rows = (C.c_uint32 * (2 * self.world))(*[v for b in layout for v in (b[0], b[1])])
        _lib.check(
            self._lib.pm_gather(
                (renderer or self._r)._h, self._h, band.data_ptr() if band is not None else None, band.stride(0) if band is not None else 0, rows, root,
                full.data_ptr() if full is not None else None, full.stride(0) if full is not None else 0,
                stream.cuda_stream if stream is not None else None,
            ),
            "pm_gather",
        )
        if stream is not None and not stream.cuda_stream:  # (torch's default stream: see Renderer.render_to)
            _warn_default_stream()
            (renderer or self._r).sync()

    @staticmethod
    def library_path() -> str:
        """The RCCL shared object the C ABI bound (loads it if it has not yet)."""
        buf = C.create_string_buffer(1024)
        _lib.check(_lib.load().pm_comm_info(None, buf, len(buf), None), "pm_comm_info")
        return buf.value.decode()

    def info(self) -> dict:
        """{"rccl_lib": path of the bound library, "rccl_ranks": ncclCommCount of this communicator}"""
        buf = C.create_string_buffer(1024)
        n = C.c_int(0)
        _lib.check(self._lib.pm_comm_info(self._h, buf, len(buf), C.byref(n)), "pm_comm_info")
        return {"rccl_lib": buf.value.decode(), "rccl_ranks": int(n.value)}

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pm_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _time_tiles(self, max_slots: int = 1 << 20) -> np.ndarray:
    """Developer profiling: per-slot timeline of pm_fine_kernel, rows =
    (start, end, tile | quarter << 31, wave << 32 | commands, phase A ticks, phase B ticks,
    list complete (fused kernel), own-item ticks, 4 x packed list-building stages)."""
    out = np.zeros((max_slots, 12), np.uint64)
    n = C.c_size_t(0)
    _lib.check(self._lib.pm_debug_time_tiles(self._h, out.ctypes.data, max_slots, C.byref(n)), "pm_debug_time_tiles")
    return out[: n.value]


Renderer.time_tiles = _time_tiles


def _time_bins(self, max_rows: int = 1 << 20) -> np.ndarray:
    out = np.zeros((max_rows, 16), np.uint64)
    n = C.c_size_t(0)
    _lib.check(self._lib.pm_debug_time_bins(self._h, out.ctypes.data, max_rows, C.byref(n)), "pm_debug_time_bins")
    return out[: n.value]


Renderer.time_bins = _time_bins


def init_test_scene(buf: np.ndarray) -> None:
    """The reference's one FFI symbol (include/piet_metal.h:3): Tiger at scale 8."""
    lib = _lib.load()
    lib.init_test_scene(buf.ctypes.data, buf.size)
