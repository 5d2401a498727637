"""Scene encoder: the piet-metal `Encoder` API (src/lib.rs:79-254) over the C ABI.

Method names, argument order and error behaviour follow the Rust type: a misuse
that panics there (assert!/unwrap) raises PietMetalError here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class Encoder:
    """`Encoder::new(buf)`: writes the scene into a caller-owned byte buffer."""

    def __init__(self, buf: np.ndarray):
        if buf.dtype != np.uint8 or not buf.flags["C_CONTIGUOUS"]:
            raise TypeError("Encoder needs a contiguous uint8 buffer")
        self._lib = _lib.load()
        self._buf = buf  # keep alive
        self._h = self._lib.pm_encoder_new(buf.ctypes.data, buf.size)
        if not self._h:
            raise MemoryError("pm_encoder_new failed")

    def close(self) -> None:
        if self._h:
            self._lib.pm_encoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self, size: int) -> int:
        return int(self._lib.pm_encoder_alloc(self._h, size))

    def write_struct(self, ix: int, data: bytes) -> None:
        """Encoder::write_struct (src/lib.rs:122): raw bytes of a #[repr(C)] value at offset ix."""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        _lib.check(self._lib.pm_encoder_write_struct(self._h, ix, buf, len(data)), "pm_encoder_write_struct")

    def encode_points(self, points):
        """Encoder::encode_points (src/lib.rs:224) -> (points_ix, (x0, y0, x1, y1))."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        ix = C.c_size_t(0)
        bb = (C.c_double * 4)()
        _lib.check(self._lib.pm_encoder_encode_points(self._h, pts.ctypes.data, len(pts), C.byref(ix), bb), "pm_encoder_encode_points")
        return int(ix.value), tuple(bb)

    def begin_group(self, n_items: int) -> None:
        """`Encoder::begin_group`; inside an open group it starts a nested group (extension)."""
        _lib.check(self._lib.pm_encoder_begin_group(self._h, n_items), "begin_group")

    def end_group(self) -> None:
        _lib.check(self._lib.pm_encoder_end_group(self._h), "end_group")

    def circle(self, center, radius: float) -> None:
        _lib.check(self._lib.pm_encoder_circle(self._h, center[0], center[1], radius), "circle")

    def fill_compound(self, subpaths, rgba: int, even_odd: bool = False) -> None:
        """Beyond the reference: ONE Fill item made of several closed sub-paths that share a winding
        sum (holes); `subpaths` is a sequence of (n, 2) point arrays."""
        subs = [np.ascontiguousarray(s, np.float64).reshape(-1, 2) for s in subpaths]
        counts = np.asarray([len(s) for s in subs], np.uint32)
        pts = np.concatenate(subs) if subs else np.zeros((0, 2), np.float64)
        _lib.check(self._lib.pm_encoder_fill_compound(self._h, pts.ctypes.data, counts.ctypes.data, len(subs), rgba,
                                                     _lib.PM_FILL_EVEN_ODD if even_odd else 0), "fill_compound")

    def fill_path(self, els: np.ndarray, rgba: int, even_odd: bool = False, compound: bool = False) -> None:
        """Beyond the reference: a path with curves and sub-paths (PathSet.EL_DTYPE elements), flattened
        on the host like make_tiger's encode_path; compound=True: ONE Fill item, holes are holes."""
        els = np.ascontiguousarray(els, dtype=_EL_DTYPE)
        flags = (_lib.PM_FILL_EVEN_ODD if even_odd else 0) | (_lib.PM_FILL_COMPOUND if compound else 0)
        _lib.check(self._lib.pm_encoder_fill_path(self._h, els.ctypes.data, len(els), rgba & 0xFFFFFFFF, flags), "fill_path")

    def stroke_path(self, els: np.ndarray, rgba: int, width: float) -> None:
        """A path's sub-paths as poly-lines, with the thin-line rule of encode_path_stroke."""
        els = np.ascontiguousarray(els, dtype=_EL_DTYPE)
        _lib.check(self._lib.pm_encoder_stroke_path(self._h, els.ctypes.data, len(els), rgba & 0xFFFFFFFF, float(width)), "stroke_path")

    def ellipse(self, center, rx: float, ry: float) -> None:
        """Beyond the reference: the ellipse inscribed in the item's bbox (a Circle item with the
        ellipse bit), shaded as PietRender.metal:488-489 says it should be."""
        _lib.check(self._lib.pm_encoder_ellipse(self._h, center[0], center[1], rx, ry), "ellipse")

    def stroke_line(self, p0, p1, width: float, rgba: int) -> None:
        _lib.check(
            self._lib.pm_encoder_stroke_line(self._h, p0[0], p0[1], p1[0], p1[1], width, rgba & 0xFFFFFFFF),
            "stroke_line",
        )

    @staticmethod
    def _pts(points) -> np.ndarray:
        a = np.ascontiguousarray(points, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != 2:
            raise ValueError("points must be (n, 2)")
        return a

    def fill(self, points, rgba: int, even_odd: bool = False) -> None:
        """`Encoder::fill`; even_odd sets the winding-rule bit of PietFill.flags (an extension
        the reference reserves the field for, src/lib.rs:54)."""
        a = self._pts(points)
        if even_odd:
            _lib.check(self._lib.pm_encoder_fill_rule(self._h, a.ctypes.data, a.shape[0], rgba & 0xFFFFFFFF, _lib.PM_FILL_EVEN_ODD), "fill")
        else:
            _lib.check(self._lib.pm_encoder_fill(self._h, a.ctypes.data, a.shape[0], rgba & 0xFFFFFFFF), "fill")

    def polyline(self, points, rgba: int, width: float) -> None:
        a = self._pts(points)
        _lib.check(
            self._lib.pm_encoder_polyline(self._h, a.ctypes.data, a.shape[0], rgba & 0xFFFFFFFF, width), "polyline"
        )

    @property
    def bytes_used(self) -> int:
        return int(self._lib.pm_encoder_bytes_used(self._h))


def scene_cardioid(buf: np.ndarray) -> int:
    """make_cardioid (src/lib.rs:257-270); returns bytes written."""
    n = _lib.load().pm_scene_cardioid(buf.ctypes.data, buf.size)
    if n < 0:
        raise _lib.PietMetalError(int(n), "pm_scene_cardioid")
    return int(n)


def scene_path_test(buf: np.ndarray) -> int:
    """make_path_test (src/lib.rs:273-284); returns bytes written."""
    n = _lib.load().pm_scene_path_test(buf.ctypes.data, buf.size)
    if n < 0:
        raise _lib.PietMetalError(int(n), "pm_scene_path_test")
    return int(n)


def parse_color(s: str) -> int:
    """parse_color (src/lib.rs:375-385)."""
    return int(_lib.load().pm_parse_color(s.encode()))


_PATH_DTYPE = np.dtype(
    [("el_begin", "<u4"), ("el_end", "<u4"), ("flags", "<u4"), ("fill_rgba", "<u4"), ("stroke_rgba", "<u4"), ("stroke_width", "<f4")]
)
_EL_DTYPE = np.dtype([("tag", "<u4"), ("pad", "<u4"), ("p", "<f8", (6,))])
assert _PATH_DTYPE.itemsize == 24 and _EL_DTYPE.itemsize == 56


class PathSet:
    """Parsed paths: `paths` (structured, 24 B) and `els` (structured, 56 B) arrays
    in the layout of pm_path / pm_path_el."""

    PATH_DTYPE = _PATH_DTYPE
    EL_DTYPE = _EL_DTYPE

    def __init__(self, paths: np.ndarray, els: np.ndarray):
        self.paths = np.ascontiguousarray(paths, dtype=_PATH_DTYPE)
        self.els = np.ascontiguousarray(els, dtype=_EL_DTYPE)

    @classmethod
    def _from_handle(cls, lib, h) -> "PathSet":
        try:
            npaths, nels = lib.pm_svg_n_paths(h), lib.pm_svg_n_els(h)
            paths = np.frombuffer(C.string_at(lib.pm_svg_paths(h), npaths * 24), dtype=_PATH_DTYPE).copy() if npaths else np.zeros(0, _PATH_DTYPE)
            els = np.frombuffer(C.string_at(lib.pm_svg_els(h), nels * 56), dtype=_EL_DTYPE).copy() if nels else np.zeros(0, _EL_DTYPE)
            vb, w, hh = (C.c_double * 4)(), C.c_double(0), C.c_double(0)
            has_vb = lib.pm_svg_viewbox(h, vb, C.byref(w), C.byref(hh))
        finally:
            lib.pm_svg_free(h)
        ps = cls(paths, els)
        ps.viewbox = tuple(vb) if has_vb else None  # the outermost <svg>'s viewBox (user units)
        ps.size = (w.value, hh.value)                # its width / height in px (0: not given)
        return ps

    def fit_affine(self, width: int, height: int):
        """(affine, scale) that shows the document's viewBox (or its width x height) centred in a
        width x height viewport, aspect ratio kept (SVG's default preserveAspectRatio xMidYMid meet);
        None when the document gives neither."""
        vb = getattr(self, "viewbox", None)
        if vb is None:
            w, h = getattr(self, "size", (0.0, 0.0))
            if not (w > 0 and h > 0):
                return None
            vb = (0.0, 0.0, w, h)
        s = min(width / vb[2], height / vb[3])
        return (s, 0.0, 0.0, s, (width - s * vb[2]) / 2.0 - s * vb[0], (height - s * vb[3]) / 2.0 - s * vb[1]), s

    @classmethod
    def from_svg(cls, text: bytes | str, reject_arc_paths: bool = False, spec_defaults: bool = False, flat_gradients: bool = False) -> "PathSet":
        """Parse an SVG document.  spec_defaults: SVG's initial `fill: black` instead of the
        reference's rule that only a fill property fills (src/lib.rs:299); flat_gradients: a
        url(#gradient) paint becomes the mean colour of the gradient's stops instead of `none`."""
        lib = _lib.load()
        data = text.encode() if isinstance(text, str) else bytes(text)
        err = C.c_int(0)
        flags = (_lib.PM_SVG_REJECT_ARC_PATHS if reject_arc_paths else 0) | (_lib.PM_SVG_SPEC_DEFAULTS if spec_defaults else 0)
        flags |= _lib.PM_SVG_FLAT_GRADIENTS if flat_gradients else 0
        h = lib.pm_svg_parse(data, len(data), flags, C.byref(err))
        if not h:
            raise _lib.PietMetalError(err.value, "pm_svg_parse")
        return cls._from_handle(lib, h)

    @classmethod
    def tiger(cls, reject_arc_paths: bool = False) -> "PathSet":
        """The embedded Ghostscript_Tiger.svg (src/lib.rs:288)."""
        lib = _lib.load()
        err = C.c_int(0)
        h = lib.pm_svg_tiger(_lib.PM_SVG_REJECT_ARC_PATHS if reject_arc_paths else 0, C.byref(err))
        if not h:
            raise _lib.PietMetalError(err.value, "pm_svg_tiger")
        return cls._from_handle(lib, h)

    def fills_only(self) -> "PathSet":
        p = self.paths.copy()
        p["flags"] &= _lib.PM_PATH_FILL
        return PathSet(p, self.els)

    @staticmethod
    def concat(sets: list["PathSet"]) -> "PathSet":
        paths, els, base = [], [], 0
        for s in sets:
            p = s.paths.copy()
            p["el_begin"] += base
            p["el_end"] += base
            base += len(s.els)
            paths.append(p)
            els.append(s.els)
        return PathSet(np.concatenate(paths), np.concatenate(els))

    def transformed(self, affine) -> "PathSet":
        """Apply an affine [a b c d e f] to the element coordinates on the host
        (used to lay out multi-copy scenes before the device flatten)."""
        a, b, c, d, e, f = [float(v) for v in affine]
        els = self.els.copy()
        p = els["p"]
        x, y = p[:, 0::2].copy(), p[:, 1::2].copy()
        p[:, 0::2] = a * x + c * y + e
        p[:, 1::2] = b * x + d * y + f
        return PathSet(self.paths, els)
