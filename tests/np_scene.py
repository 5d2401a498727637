"""Independent Python restatement of the scene producer: make_tiger's two passes on parsed paths
(src/lib.rs:286-367) = Affine * path, flatten.rs:10-47 (kurbo 0.5.6 to_quads / eval as published:
parity unpinned), the thin-line rule and the Encoder's byte layout
(src/lib.rs:15-254) -- written from the Rust source with Python floats (f64) and struct.pack.
Cross-checks oracle/pmo_flatten.c + pmo_encoder.c.  Test infrastructure.
"""
import math


def subdivision_count(x):
    """kurbo to_quads' n = max(1, ceil(x^(1/6))) as an exact integer property of x: the smallest
    n >= 1 with n^6 >= x (Python integers against the exact rational value of the double x)."""
    from fractions import Fraction

    if not x > 1.0:
        return 1
    fx = Fraction(x)
    n = max(1, int(round(x ** (1.0 / 6.0))))
    while n > 1 and (n - 1) ** 6 >= fx:
        n -= 1
    while n ** 6 < fx:
        n += 1
    return n
import struct

import numpy as np

EL_MOVE, EL_LINE, EL_QUAD, EL_CURVE, EL_CLOSE = range(5)
TOLERANCE = 0.1
THIN_LINE = np.float32(0.7)


def flatten(els, affine):
    a, b, c, d, e, f = affine
    xf = lambda x, y: (a * x + c * y + e, b * x + d * y + f)
    subs, cur, last = [], None, (0.0, 0.0)
    for el in els:
        tag, p = int(el["tag"]), [float(v) for v in el["p"]]
        if tag == EL_MOVE:
            if cur is not None:
                subs.append(cur)
            q = xf(p[0], p[1])
            cur, last = [q], q
        elif tag == EL_LINE:
            q = xf(p[0], p[1])
            cur.append(q)
            last = q
        elif tag == EL_CURVE:
            p1, p2, p3 = xf(p[0], p[1]), xf(p[2], p[3]), xf(p[4], p[5])
            acc = TOLERANCE * 1e-2
            max_hypot2 = 432.0 * acc * acc
            ax, ay = p1[0] * 3.0 - last[0], p1[1] * 3.0 - last[1]
            bx, by = p2[0] * 3.0 - p3[0], p2[1] * 3.0 - p3[1]
            dx, dy = bx - ax, by - ay
            err = dx * dx + dy * dy
            nf = subdivision_count(err / max_hypot2)
            n = int(nf) if nf >= 1.0 else 1
            for k in range(n):
                t = (k + 1) / n
                mt = 1.0 - t
                ev = lambda q0, q1, q2, q3: q0 * (mt * mt * mt) + (q1 * (mt * mt * 3.0) + (q2 * (mt * 3.0) + q3 * t) * t) * t
                cur.append((ev(last[0], p1[0], p2[0], p3[0]), ev(last[1], p1[1], p2[1], p3[1])))
            last = p3
        # QuadTo, ClosePath: `_ => ()`
    if cur is not None:
        subs.append(cur)
    return subs


def _short_bbox(x0, y0, x1, y1):
    cl = lambda v: int(min(max(v, 0.0), 65535.0))
    return struct.pack("<4H", cl(math.floor(x0)), cl(math.floor(y0)), cl(math.ceil(x1)), cl(math.ceil(y1)))


def scene_from_paths(paths, els, affine) -> bytes:
    """paths: stroke widths ALREADY multiplied by the scale (src/lib.rs:320)."""
    flat = [flatten(els[int(p["el_begin"]) : int(p["el_end"])], affine) for p in paths]
    # (extensions: flags bit 2 = even-odd rule, bit 3 = the filled sub-paths form ONE compound item)
    n_items = sum((min(len(s), 1) if p["flags"] & 8 else len(s)) * bool(p["flags"] & 1) + len(s) * bool(p["flags"] & 2) for p, s in zip(paths, flat))
    item_start = 8 + 8 * n_items
    buf = bytearray(item_start + 32 * n_items)
    struct.pack_into("<II", buf, 0, n_items, item_start)
    state = {"ix": 0}

    def add_item(item: bytes, bbox: bytes):
        i = state["ix"]
        buf[8 + 8 * i : 16 + 8 * i] = bbox
        buf[item_start + 32 * i : item_start + 32 * i + len(item)] = item
        state["ix"] = i + 1

    def encode_points(pts):
        nonlocal buf
        points_ix = len(buf)
        x0 = x1 = pts[0][0]
        y0 = y1 = pts[0][1]
        for (x, y) in pts:
            x0, x1, y0, y1 = min(x0, x), max(x1, x), min(y0, y), max(y1, y)
            buf += struct.pack("<ff", np.float32(x), np.float32(y))
        return points_ix, (x0, y0, x1, y1)

    be = lambda v: struct.unpack("<I", struct.pack(">I", v & 0xFFFFFFFF))[0]
    for p, subs in zip(paths, flat):
        rule = 1 if p["flags"] & 4 else 0
        if p["flags"] & 1 and p["flags"] & 8 and subs:
            # compound fill: points and a separator {NaN, start index} after every sub-path
            pix = len(buf)
            allp = [q for pts in subs for q in pts]
            bb = (min(q[0] for q in allp), min(q[1] for q in allp), max(q[0] for q in allp), max(q[1] for q in allp))
            at = 0
            for pts in subs:
                for (x, y) in pts:
                    buf += struct.pack("<ff", np.float32(x), np.float32(y))
                buf += struct.pack("<II", 0x7FC00000, at)
                at += len(pts) + 1
            add_item(struct.pack("<5I", 3, 2 | rule, be(int(p["fill_rgba"])), at, pix), _short_bbox(*bb))
        elif p["flags"] & 1:
            for pts in subs:
                pix, bb = encode_points(pts)
                add_item(struct.pack("<5I", 3, rule, be(int(p["fill_rgba"])), len(pts), pix), _short_bbox(*bb))
        if p["flags"] & 2:
            width, rgba = np.float32(p["stroke_width"]), int(p["stroke_rgba"])
            if width < THIN_LINE:
                alpha = np.float32(rgba & 0xFF)
                alpha = np.float32(alpha * np.sqrt(np.float32(width / THIN_LINE)))
                rgba = (rgba & ~0xFF & 0xFFFFFFFF) | int(alpha)
                width = THIN_LINE
            hw = float(np.float32(width * np.float32(0.5)))
            for pts in subs:
                pix, bb = encode_points(pts)
                add_item(struct.pack("<IIfII", 4, be(rgba), width, len(pts), pix), _short_bbox(bb[0] - hw, bb[1] - hw, bb[2] + hw, bb[3] + hw))
    return bytes(buf)
