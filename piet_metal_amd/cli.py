"""Headless presentation for the renderer: SVG (or the embedded Tiger) -> PNG.

    python -m piet_metal_amd.cli tiger out.png --width 3840 --height 2160
    python -m piet_metal_amd.cli drawing.svg out.png --scale 4 --width 1024 --height 1024

Replaces the reference's MTKView shell (TestApp/ViewController.m, PietRenderer.m:90-101) for a
machine without a display: the frame is rendered on the MI355X by the same three kernels as
bench.py and read back once.  The SVG subset is what src/lib.rs:286-385 understands: <path>
elements with d / fill / stroke / stroke-width, 3- and 6-digit hex colours.
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
    ap.add_argument("--scale", type=float, default=None, help="user units -> pixels (default: height / 200, the Tiger's viewBox)")
    ap.add_argument("--offset", type=float, nargs=2, default=None, metavar=("X", "Y"), help="translation in pixels (default: centre horizontally)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--reject-arc-paths", action="store_true", help="skip <path>s that use the arc command (kurbo 0.5.6 question, SURVEY F6)")
    args = ap.parse_args(argv)

    from . import PathSet, Renderer

    if args.input == "tiger":
        paths = PathSet.tiger(args.reject_arc_paths)
    else:
        with open(args.input, "rb") as f:
            paths = PathSet.from_svg(f.read(), args.reject_arc_paths)
    scale = args.scale if args.scale is not None else args.height / 200.0
    off = args.offset if args.offset is not None else ((args.width - args.height) / 2.0 if args.scale is None else 0.0, 0.0)
    with Renderer(args.device) as r:
        r.resize(args.width, args.height)
        nbytes, nitems = r.flatten_and_encode(paths, (scale, 0.0, 0.0, scale, float(off[0]), float(off[1])), scale)
        r.render()
        img = r.read_pixels()
    write_png(args.output, img)
    print(f"{args.output}: {args.width}x{args.height}, {nitems} items, scene {nbytes} bytes", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
