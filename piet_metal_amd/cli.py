"""Headless presentation for the renderer: SVG (or the embedded Tiger) -> PNG.

    python -m piet_metal_amd.cli tiger out.png --width 3840 --height 2160
    python -m piet_metal_amd.cli drawing.svg out.png --scale 4 --width 1024 --height 1024

    python -m piet_metal_amd.cli tiger spin.png --frames 60 --spin 360     (spin-000.png ... spin-059.png)

Replaces the reference's MTKView shell (TestApp/ViewController.m, PietRenderer.m:90-101) for a
machine without a display: the frame is rendered on the MI355X by the same three kernels as
bench.py and read back once.  Files are read with the SVG front-end's full document layer (groups,
transforms, style, opacity, fill-rule, basic shapes; SVG's initial `fill: black`); `tiger` is the
embedded asset read as make_tiger reads it (src/lib.rs:286-328).  --frames renders an animation the
way the reference's view does on every change (PietRenderer.m:90-101, :145): the scene is encoded
again for each frame -- here by re-flattening the resident paths on the device (pm_reflatten).
"""
from __future__ import annotations

import argparse
import struct
import sys
import zlib

import numpy as np


def write_png(path: str, rgba: np.ndarray) -> None:
    """RGBA8 [H, W, 4] -> PNG (colour type 6, no interlace); stdlib only."""
    h, w, c = rgba.shape
    if c != 4 or rgba.dtype != np.uint8:
        raise ValueError("write_png needs an [H, W, 4] uint8 array")
    raw = np.empty((h, 1 + 4 * w), np.uint8)
    raw[:, 0] = 0  # filter type None
    raw[:, 1:] = rgba.reshape(h, 4 * w)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def read_png_rgba(path: str) -> np.ndarray:
    """Inverse of write_png (only the subset write_png produces); used by the tests."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        (n,), tag = struct.unpack(">I", data[pos : pos + 4]), data[pos + 4 : pos + 8]
        body = data[pos + 8 : pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 4 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 4).copy()


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m piet_metal_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("input", help="an .svg file, or 'tiger' for the embedded Ghostscript Tiger")
    ap.add_argument("output", help="PNG file to write")
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1600)
    ap.add_argument("--scale", type=float, default=None, help="user units -> pixels (default: fit the file's viewBox into the viewport; the Tiger: height / 200)")
    ap.add_argument("--offset", type=float, nargs=2, default=None, metavar=("X", "Y"), help="translation in pixels (default: centre horizontally)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--reject-arc-paths", action="store_true", help="skip <path>s that use the arc command (kurbo 0.5.6 question, SURVEY F6)")
    ap.add_argument("--reference-fill-rule", action="store_true", help="files: only a fill property fills (make_tiger, src/lib.rs:299) instead of SVG's initial black")
    ap.add_argument("--no-flat-gradients", action="store_true", help="files: do not draw gradient paints at all (default: as the mean colour of their stops)")
    ap.add_argument("--frames", type=int, default=1, help="render an animation of this many frames (output NAME-###.png)")
    ap.add_argument("--spin", type=float, default=360.0, help="--frames: total rotation about the viewport centre, degrees")
    args = ap.parse_args(argv)

    from . import PathSet, Renderer

    if args.input == "tiger":
        paths = PathSet.tiger(args.reject_arc_paths)
    else:
        with open(args.input, "rb") as f:
            paths = PathSet.from_svg(f.read(), args.reject_arc_paths, spec_defaults=not args.reference_fill_rule, flat_gradients=not args.no_flat_gradients)
    scale = args.scale if args.scale is not None else args.height / 200.0
    off = args.offset if args.offset is not None else ((args.width - args.height) / 2.0 if args.scale is None else 0.0, 0.0)
    base = (scale, 0.0, 0.0, scale, float(off[0]), float(off[1]))
    fit = paths.fit_affine(args.width, args.height) if (args.input != "tiger" and args.scale is None and args.offset is None) else None
    if fit is not None:  # a file that says where its picture is: show that, centred (xMidYMid meet)
        base, scale = fit
    with Renderer(args.device) as r:
        r.resize(args.width, args.height)
        nbytes, nitems = r.flatten_and_encode(paths, base, scale)
        r.render()
        img = r.read_pixels()
        if args.frames <= 1:
            write_png(args.output, img)
            print(f"{args.output}: {args.width}x{args.height}, {nitems} items, scene {nbytes} bytes", file=sys.stderr)
            return 0
        import math
        import time

        stem = args.output[:-4] if args.output.lower().endswith(".png") else args.output
        cx, cy = args.width / 2.0, args.height / 2.0
        t_gpu = 0.0
        for k in range(args.frames):
            aff = spin_affine(base, math.radians(args.spin * k / args.frames), cx, cy)
            t0 = time.perf_counter()
            nbytes, nitems = r.reflatten(aff, scale)  # the per-frame re-encode, on the device
            r.render()
            r.sync()
            t_gpu += time.perf_counter() - t0
            write_png(f"{stem}-{k:03d}.png", r.read_pixels())
        print(f"{stem}-###.png: {args.frames} frames {args.width}x{args.height}, re-encode + render {t_gpu / args.frames * 1e3:.2f} ms per frame", file=sys.stderr)
    return 0


def spin_affine(base, theta: float, cx: float, cy: float):
    """`base` followed by a rotation by theta about (cx, cy) (kurbo Affine coefficients [a b c d e f])."""
    import math

    a, b, c, d, e, f = base
    cs, sn = math.cos(theta), math.sin(theta)
    # R * base, R = translate(cx, cy) rotate(theta) translate(-cx, -cy)
    re, rf = cx - cs * cx + sn * cy, cy - sn * cx - cs * cy
    return (cs * a - sn * b, sn * a + cs * b, cs * c - sn * d, sn * c + cs * d, cs * e - sn * f + re, sn * e + cs * f + rf)


if __name__ == "__main__":
    sys.exit(main())
