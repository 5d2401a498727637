"""The workloads BASELINE.json names, as (paths, affine, width_scale, viewport).

The reference hard-codes Tiger * 8.0 regardless of viewport (src/lib.rs:287); the
BASELINE configs need a scale it never defines, so it is stated here
(SURVEY.md F7): scale = height / 200 (the SVG viewBox is 200 x 200), centred
horizontally.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _lib
from .encoder import PathSet


@dataclass
class Workload:
    name: str
    width: int
    height: int
    paths: PathSet | None  # None => scene_bytes is given directly
    affine: tuple
    width_scale: float
    scene_bytes: bytes | None = None


def tiger(width: int, height: int, fills_only: bool = False, reject_arc_paths: bool = False) -> Workload:
    ps = PathSet.tiger(reject_arc_paths)
    if fills_only:
        ps = ps.fills_only()
    scale = height / 200.0
    off_x = (width - height) / 2.0
    return Workload(
        name=f"tiger_{width}x{height}" + ("_fills" if fills_only else ""),
        width=width, height=height, paths=ps,
        affine=(scale, 0.0, 0.0, scale, off_x, 0.0), width_scale=scale,
    )


def tiger_reference() -> Workload:
    """What init_test_scene builds: Tiger * 8 (src/lib.rs:286-328), 1600 x 1600 px."""
    return Workload("tiger_x8", 1600, 1600, PathSet.tiger(), (8.0, 0.0, 0.0, 8.0, 0.0, 0.0), 8.0)


def config1_rect(rotated: bool = False) -> Workload:
    """BASELINE config 1: one filled 256 x 256 rect in a 512 x 512 viewport."""
    x0, y0, s = 72.0, 40.0, 256.0
    pts = np.array([[x0, y0], [x0 + s, y0], [x0 + s, y0 + s], [x0, y0 + s]], np.float64)
    if rotated:
        th = np.deg2rad(0.37)
        c, si = np.cos(th), np.sin(th)
        ctr = pts.mean(axis=0)
        d = pts - ctr
        pts = np.stack([ctr[0] + c * d[:, 0] - si * d[:, 1] + 40.3 - 72.0, ctr[1] + si * d[:, 0] + c * d[:, 1] + 37.7 - 40.0], axis=1)
    els = np.zeros(5, PathSet.EL_DTYPE)
    els["tag"] = [_lib.PM_EL_MOVE, _lib.PM_EL_LINE, _lib.PM_EL_LINE, _lib.PM_EL_LINE, _lib.PM_EL_CLOSE]
    for i in range(4):
        els["p"][i, 0:2] = pts[i]
    paths = np.zeros(1, PathSet.PATH_DTYPE)
    paths[0] = (0, 5, _lib.PM_PATH_FILL, 0x1F4FA0FF, 0, 0.0)
    return Workload("rect_rot" if rotated else "rect", 512, 512, PathSet(paths, els), (1.0, 0.0, 0.0, 1.0, 0.0, 0.0), 1.0)


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self) -> float:  # [0, 1)
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)


def config4_blobs(n_paths: int = 10000, size: int = 4096, seed: int = 0x5EED0004) -> Workload:
    """BASELINE config 4: n closed 4-cubic blobs (perturbed circles).

    Draw order per path: cx, cy, r, then 12 radial jitters (p0 c0a c0b p1 c1a c1b
    ... in path order), then one 64-bit draw for the colour (RGB = low 24 bits,
    alpha = 0x40 + (bits 24..31 mod 0xC0))."""
    rng = SplitMix64(seed)
    k = 0.5522847498307936
    # unit-circle control polygon, 4 arcs starting at angle 0 (y down)
    unit = []
    for q in range(4):
        a0 = q * np.pi / 2
        c0, s0 = np.cos(a0), np.sin(a0)
        c1, s1 = np.cos(a0 + np.pi / 2), np.sin(a0 + np.pi / 2)
        unit.append([(c0, s0), (c0 - k * s0, s0 + k * c0), (c1 + k * s1, s1 - k * c1)])
    els = np.zeros(n_paths * 6, PathSet.EL_DTYPE)
    paths = np.zeros(n_paths, PathSet.PATH_DTYPE)
    for i in range(n_paths):
        cx, cy = rng.uniform() * size, rng.uniform() * size
        r = 32.0 + rng.uniform() * (256.0 - 32.0)
        pts = []
        for q in range(4):
            for (ux, uy) in unit[q]:
                j = 1.0 + (rng.uniform() - 0.5) * 0.5
                pts.append((cx + ux * r * j, cy + uy * r * j))
        bits = rng.next()
        rgb = bits & 0xFFFFFF
        alpha = 0x40 + ((bits >> 24) & 0xFF) % 0xC0
        e = i * 6
        els["tag"][e] = _lib.PM_EL_MOVE
        els["p"][e, 0:2] = pts[0]
        for q in range(4):
            els["tag"][e + 1 + q] = _lib.PM_EL_CURVE
            end = pts[(3 * (q + 1)) % 12]
            els["p"][e + 1 + q, 0:6] = (*pts[3 * q + 1], *pts[3 * q + 2], *end)
        els["tag"][e + 5] = _lib.PM_EL_CLOSE
        paths[i] = (e, e + 6, _lib.PM_PATH_FILL, (rgb << 8) | alpha, 0, 0.0)
    return Workload(f"blobs_{n_paths}_{size}", size, size, PathSet(paths, els), (1.0, 0.0, 0.0, 1.0, 0.0, 0.0), 1.0)


def config5_tiger_grid(size: int = 8192, copies: int = 5, scale: float = 8.0) -> Workload:
    """BASELINE config 5: copies x copies Tigers at `scale`, pitch size/copies px."""
    base = PathSet.tiger()
    pitch = size // copies  # 1638 for 8192 / 5
    sets = []
    for gy in range(copies):
        for gx in range(copies):
            # translate in user units so that one global affine (scale) lays the grid out
            sets.append(base.transformed((1.0, 0.0, 0.0, 1.0, gx * pitch / scale, gy * pitch / scale)))
    return Workload(f"tiger_grid_{size}", size, size, PathSet.concat(sets), (scale, 0.0, 0.0, scale, 0.0, 0.0), scale)


def heldout_glyphs(n_paths: int = 20000, width: int = 3840, height: int = 2160, seed: int = 0x5EED0007) -> Workload:
    """Held-out workload 3 (no threshold of the frame path was chosen on it): ~20 k tiny closed paths, text-like -- lines of
    "glyphs" 7-15 px high, each a closed outline of two cubics and two or three line segments, opaque dark on white, one
    in eight translucent and coloured.  Many items, few segments each, nearly every tile touched lightly."""
    rng = SplitMix64(seed)
    els = np.zeros(n_paths * 7, PathSet.EL_DTYPE)
    paths = np.zeros(n_paths, PathSet.PATH_DTYPE)
    x, y, line_h = 24.0, 30.0, 22.0
    e = 0
    for i in range(n_paths):
        w = 6.0 + rng.uniform() * 8.0
        h = 7.0 + rng.uniform() * 8.0
        if x + w > width - 24.0:
            x = 24.0 + rng.uniform() * 12.0
            y += line_h
        if y + h > height - 10.0:
            y = 30.0 + rng.uniform() * 6.0
        j = lambda a: (rng.uniform() - 0.5) * a  # noqa: E731
        x0, y0, x1, y1 = x, y - h, x + w, y
        start = e
        els["tag"][e] = _lib.PM_EL_MOVE
        els["p"][e, 0:2] = (x0 + j(1.5), y1)
        e += 1
        els["tag"][e] = _lib.PM_EL_CURVE  # left side up, bowed
        els["p"][e, 0:6] = (x0 - 1.0 + j(2), y1 - h * 0.4, x0 + j(2), y0 + h * 0.3, x0 + w * 0.3 + j(2), y0 + j(1.5))
        e += 1
        els["tag"][e] = _lib.PM_EL_LINE
        els["p"][e, 0:2] = (x1 - w * 0.25 + j(2), y0 + j(1.5))
        e += 1
        els["tag"][e] = _lib.PM_EL_CURVE  # right side down
        els["p"][e, 0:6] = (x1 + 1.0 + j(2), y0 + h * 0.35, x1 + j(2), y1 - h * 0.3, x1 - w * 0.2 + j(2), y1 + j(1.0))
        e += 1
        if rng.uniform() < 0.5:  # a notch in the base line
            els["tag"][e] = _lib.PM_EL_LINE
            els["p"][e, 0:2] = (x0 + w * 0.5 + j(1), y1 - h * 0.25 + j(1))
            e += 1
        els["tag"][e] = _lib.PM_EL_CLOSE
        e += 1
        bits = rng.next()
        rgba = 0x101018FF if bits & 7 else ((bits >> 8) & 0xFFFFFF) << 8 | (0x60 + ((bits >> 32) & 0x7F))
        paths[i] = (start, e, _lib.PM_PATH_FILL, rgba, 0, 0.0)
        x += w + 1.5 + (6.0 if (bits >> 40) % 6 == 0 else 0.0)
    return Workload(f"glyphs_{n_paths}_{width}x{height}", width, height, PathSet(paths, els[:e].copy()), (1.0, 0.0, 0.0, 1.0, 0.0, 0.0), 1.0)


def heldout_workloads() -> dict:
    """Three scenes that drove no threshold or switch of the frame path: what the policy
    switches are checked against in tools/heldout_policy.py, with full-size oracle goldens (tests/golden, `--held`)."""
    return {"held1": tiger(2560, 1440), "held2": config4_blobs(2000, 2048, seed=0x5EED0008), "held3": heldout_glyphs()}


def band_rows(tiles_y: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous tile-row band of `rank`: near-equal split."""
    base, rem = divmod(tiles_y, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)
