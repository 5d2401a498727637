"""ctypes wrapper of the CPU ORACLE (oracle/libpmo_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from piet_metal_amd/.  It restates the
reference's algorithm (see oracle/pmo.h); it is the checker, not the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpmo_oracle.so")

MODE_HALF, MODE_F32 = 0, 1
FMT_RGBA8, FMT_BGRA8 = 0, 4

PATH_DTYPE = np.dtype(
    [("el_begin", "<u4"), ("el_end", "<u4"), ("flags", "<u4"), ("fill_rgba", "<u4"), ("stroke_rgba", "<u4"), ("stroke_width", "<f4")]
)
EL_DTYPE = np.dtype([("tag", "<u4"), ("pad", "<u4"), ("p", "<f8", (6,))])

_lib = None


def build() -> None:
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    lib.pmo_scene_cardioid.restype = C.c_int64
    lib.pmo_scene_cardioid.argtypes = [C.c_void_p, C.c_size_t]
    lib.pmo_scene_path_test.restype = C.c_int64
    lib.pmo_scene_path_test.argtypes = [C.c_void_p, C.c_size_t]
    lib.pmo_scene_from_paths.restype = C.c_int64
    lib.pmo_scene_from_paths.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    lib.pmo_ptcl_build.restype = C.c_void_p
    lib.pmo_ptcl_build.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
    lib.pmo_ptcl_build_rows.restype = C.c_void_p
    lib.pmo_ptcl_build_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.pmo_ptcl_free.argtypes = [C.c_void_p]
    lib.pmo_ptcl_tiles_x.restype = C.c_uint32
    lib.pmo_ptcl_tiles_x.argtypes = [C.c_void_p]
    lib.pmo_ptcl_tiles_y.restype = C.c_uint32
    lib.pmo_ptcl_tiles_y.argtypes = [C.c_void_p]
    lib.pmo_ptcl_count.restype = C.c_uint32
    lib.pmo_ptcl_count.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.pmo_ptcl_cmds.restype = C.c_void_p
    lib.pmo_ptcl_cmds.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.pmo_ptcl_solid.restype = C.c_uint32
    lib.pmo_ptcl_solid.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.pmo_ptcl_total_cmds.restype = C.c_uint64
    lib.pmo_ptcl_total_cmds.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.pmo_render_rows.restype = C.c_int
    lib.pmo_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pmo_render.restype = C.c_int
    lib.pmo_render.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pmo_fill_coverage.restype = C.c_int
    lib.pmo_fill_coverage.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pmo_lut_srgb_to_linear_half.argtypes = [C.c_void_p]
    lib.pmo_lut_unorm_to_half.argtypes = [C.c_void_p]
    lib.pmo_lut_linear_half_to_srgb8.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def _scene(fn, cap: int) -> np.ndarray:
    buf = np.zeros(cap, np.uint8)
    n = fn(buf.ctypes.data, buf.size)
    if n < 0:
        raise RuntimeError("oracle scene encoder failed")
    return buf[:n].copy()


def scene_cardioid() -> np.ndarray:
    return _scene(load().pmo_scene_cardioid, 1 << 16)


def scene_path_test() -> np.ndarray:
    return _scene(load().pmo_scene_path_test, 1 << 12)


def scene_from_paths(paths: np.ndarray, els: np.ndarray, affine, cap: int | None = None) -> tuple[np.ndarray, int]:
    """make_tiger's two passes (flatten.rs + Encoder) on parsed paths.  `paths`
    must hold stroke widths ALREADY multiplied by the scale (src/lib.rs:320)."""
    paths = np.ascontiguousarray(paths, dtype=PATH_DTYPE)
    els = np.ascontiguousarray(els, dtype=EL_DTYPE)
    aff = (C.c_double * 6)(*[float(v) for v in affine])
    cap = cap or (1 << 24)
    while True:
        buf = np.zeros(cap, np.uint8)
        n_items = C.c_uint32(0)
        n = load().pmo_scene_from_paths(buf.ctypes.data, buf.size, paths.ctypes.data, len(paths), els.ctypes.data, len(els), aff, C.byref(n_items))
        if n >= 0:
            return buf[:n].copy(), n_items.value
        if cap >= (1 << 31):
            raise RuntimeError("oracle scene_from_paths failed")
        cap *= 4


def scaled_paths(paths: np.ndarray, width_scale: float) -> np.ndarray:
    """width * (scale as f32), src/lib.rs:320, in f32."""
    p = np.ascontiguousarray(paths, dtype=PATH_DTYPE).copy()
    p["stroke_width"] = (p["stroke_width"].astype(np.float32) * np.float32(width_scale)).astype(np.float32)
    return p


class Ptcl:
    """Per-tile command lists (tileKernel output) for one viewport."""

    def __init__(self, scene: np.ndarray, width: int, height: int, group_rows: tuple[int, int] | None = None):
        self._lib = load()
        scene = np.ascontiguousarray(scene, dtype=np.uint8)
        if group_rows is None:
            self._h = self._lib.pmo_ptcl_build(scene.ctypes.data, scene.size, width, height)
        else:  # a slice of the tile pass: tile-group rows [a, b) = tile rows [2a, 2b)
            self._h = self._lib.pmo_ptcl_build_rows(scene.ctypes.data, scene.size, width, height, group_rows[0], group_rows[1])
        if not self._h:
            raise RuntimeError("oracle tileKernel failed (scene out of bounds?)")
        self.width, self.height = width, height
        self.tiles_x = self._lib.pmo_ptcl_tiles_x(self._h)
        self.tiles_y = self._lib.pmo_ptcl_tiles_y(self._h)

    def close(self):
        if self._h:
            self._lib.pmo_ptcl_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count(self, tx: int, ty: int) -> int:
        return self._lib.pmo_ptcl_count(self._h, tx, ty)

    def solid(self, tx: int, ty: int) -> int:
        return self._lib.pmo_ptcl_solid(self._h, tx, ty)

    def cmds(self, tx: int, ty: int) -> np.ndarray:
        n = self.count(tx, ty)
        p = self._lib.pmo_ptcl_cmds(self._h, tx, ty)
        return np.frombuffer(C.string_at(p, n * 24), dtype=np.uint32).reshape(n, 6).copy()

    def total_cmds(self) -> tuple[int, int]:
        mx = C.c_uint32(0)
        tot = self._lib.pmo_ptcl_total_cmds(self._h, C.byref(mx))
        return int(tot), mx.value

    def render_rows(self, ty0: int, ty1: int, flags: int = 0) -> np.ndarray:
        ty1 = min(ty1, self.tiles_y)
        rows = min(ty1 * 16, self.height) - ty0 * 16
        out = np.zeros((rows, self.width, 4), np.uint8)
        r = self._lib.pmo_render_rows(self._h, self.width, self.height, ty0, ty1, flags, out.ctypes.data)
        if r != 0:
            raise RuntimeError("oracle renderKernel failed")
        return out

    def render(self, flags: int = 0) -> np.ndarray:
        return self.render_rows(0, self.tiles_y, flags)


def render(scene: np.ndarray, width: int, height: int, flags: int = 0) -> np.ndarray:
    p = Ptcl(scene, width, height)
    try:
        return p.render(flags)
    finally:
        p.close()


def fill_coverage(scene: np.ndarray, item_ix: int, width: int, height: int) -> np.ndarray:
    scene = np.ascontiguousarray(scene, dtype=np.uint8)
    out = np.zeros((height, width), np.float32)
    r = load().pmo_fill_coverage(scene.ctypes.data, scene.size, item_ix, width, height, out.ctypes.data)
    if r != 0:
        raise RuntimeError(f"oracle fill_coverage failed ({r})")
    return out


def luts():
    a = np.zeros(256, np.uint16)
    b = np.zeros(256, np.uint16)
    c = np.zeros(65536, np.uint8)
    lib = load()
    lib.pmo_lut_srgb_to_linear_half(a.ctypes.data)
    lib.pmo_lut_unorm_to_half(b.ctypes.data)
    lib.pmo_lut_linear_half_to_srgb8(c.ctypes.data)
    return a, b, c
