"""Independent numpy restatement of renderKernel (TestApp/PietRender.metal:457-566) and of the
colour tables of decisions D2-D4, written against the Metal source and DESIGN.md section 2 --
not against oracle/pmo_render.c -- to cross-check the C oracle's second half (test infra).

All per-pixel state is numpy float32 / float16 arrays of one tile (16 x 16); numpy rounds every
float16 operation once (it computes in float32, which is exact for +, -, * of two binary16
values, and rounds back), i.e. exactly the "one rounding per source-level operation" rule.
"""
import numpy as np

f32, f16 = np.float32, np.float16


def lut_srgb_to_linear_half():  # unpack_unorm4x8_srgb_to_half: exact EOTF, rounded once to binary16
    c = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    return lin.astype(f16)  # float64 -> float16 is a single correctly rounded conversion in numpy


def lut_unorm_to_half():
    return (np.arange(256, dtype=np.float64) / 255.0).astype(f16)


def lut_linear_half_to_srgb8():
    """select(1.055 * pow(rgb, 1/2.4) - 0.055, 12.92 * rgb, rgb < 0.0031308) in half (:563),
    then unorm8 = clamp * 255 rounded half-to-even."""
    h = np.arange(65536, dtype=np.uint16).view(f16)
    thr, k, s, o = f16(0.0031308), f16(12.92), f16(1.055), f16(0.055)
    e = np.float64(f16(f32(1.0) / f32(2.4)))
    with np.errstate(all="ignore"):
        x = h.astype(np.float64)
        p = np.power(np.where(x > 0, x, 0.0), e).astype(f16)  # pow correctly rounded to binary16
        hi = (s * p).astype(f16) - o
        lo = k * h
        y = np.where(h < thr, lo, hi).astype(f16)
        v = np.clip(y.astype(f32), f32(0), f32(1))
        out = np.rint(v * f32(255)).astype(np.uint8)
    out[np.isnan(h)] = 0
    return out


def sat(x):
    with np.errstate(invalid="ignore"):
        return np.fmin(np.fmax(x, f32(0)), f32(1))  # fmax(NaN, 0) = 0: saturate(NaN) = 0


def hmix(x, y, a):  # mix(x, y, a) = x + (y - x) * a, each operation rounded to binary16
    return (x + ((y - x).astype(f16) * a).astype(f16)).astype(f16)


def u2f(u):
    return np.array([u], dtype=np.uint32).view(f32)[0]


def render_tile(cmds, tx, ty, tables):
    """cmds: uint32 [n, 6] (tag, body[5]) of one tile -> RGBA8 [16, 16, 4], or None for Bail."""
    srgb2lin, unorm2h, lin2srgb = tables
    ys, xs = np.mgrid[0:16, 0:16]
    px = (xs + 16 * tx).astype(f32)
    py = (ys + 16 * ty).astype(f32)
    rgb = [np.full((16, 16), 1.0, f16) for _ in range(3)]
    df = np.full((16, 16), 1e9, f32)
    sa = np.zeros((16, 16), f16)

    def blend(rgba, alpha):
        fa = (unorm2h[rgba >> 24] * alpha).astype(f16)
        for k in range(3):
            rgb[k] = hmix(rgb[k], srgb2lin[(rgba >> (8 * k)) & 0xFF], fa)

    with np.errstate(all="ignore"):
        for c in cmds:
            tag, b = int(c[0]), [int(v) for v in c[1:]]
            if tag == 1:  # End
                break
            if tag == 9:  # Bail
                return None
            if tag == 2:  # Circle :481-494
                x0, y0, x1, y1 = f32(b[1] & 0xFFFF), f32(b[1] >> 16), f32(b[2] & 0xFFFF), f32(b[2] >> 16)
                cx, cy = x0 + (x1 - x0) * f32(0.5), y0 + (y1 - y0) * f32(0.5)
                dx, dy = px - cx, py - cy
                r = np.sqrt(dx * dx + dy * dy)
                alpha = sat(np.fmin(cx - x0, cy - y0) - r).astype(f16)
                if b[0] & 1:  # extension D10: the ellipse inscribed in the bbox, F / |grad F|
                    rx, ry = cx - x0, cy - y0
                    if rx > 0 and ry > 0:
                        ux, uy = dx / (rx * rx), dy / (ry * ry)
                        g = (dx * ux + dy * uy) - f32(1)
                        ln = f32(2) * np.sqrt(ux * ux + uy * uy)
                        alpha = sat(-(g / ln)).astype(f16)
                    else:
                        alpha = np.zeros((16, 16), f16)
                for k in range(3):
                    rgb[k] = hmix(rgb[k], f16(0), alpha)
            elif tag == 3:  # Line :495-499 + stroke() :49-55
                sx, sy, ex, ey = u2f(b[1]), u2f(b[2]), u2f(b[3]), u2f(b[4])
                lx, ly = ex - sx, ey - sy
                dx, dy = px - sx, py - sy
                t = sat((lx * dx + ly * dy) / (lx * lx + ly * ly))
                fx, fy = lx * t - dx, ly * t - dy
                df = np.fmin(df, np.sqrt(fx * fx + fy * fy))
            elif tag == 5:  # Stroke :500-507
                alpha = sat(u2f(b[0]) + f32(0.5) - df).astype(f16)
                blend(b[1], alpha)
                df = np.full((16, 16), 1e9, f32)
            elif tag == 4:  # Fill :508-529
                sx, sy = u2f(b[1]) - px, u2f(b[2]) - py
                ex, ey = u2f(b[3]) - px, u2f(b[4]) - py
                wx, wy = sat(sy), sat(ey)
                live = wx != wy
                tx_ = (wx - sy) / (ey - sy)
                ty_ = (wy - sy) / (ey - sy)
                xsx = sx + (ex - sx) * tx_
                xsy = sx + (ex - sx) * ty_
                xmin = np.fmin(np.fmin(xsx, xsy), f32(1)) - f32(1e-6)
                xmax = np.fmax(xsx, xsy)
                bb = np.fmin(xmax, f32(1))
                cc = np.fmax(bb, f32(0))
                dd = np.fmax(xmin, f32(0))
                area = (bb + f32(0.5) * (dd * dd - cc * cc) - xmin) / (xmax - xmin)
                contrib = (area * (wx - wy)).astype(f16)
                sa = np.where(live, (sa + contrib).astype(f16), sa)
            elif tag == 6:  # FillEdge :530-534: half + float is an f32 add, rounded once to half
                sgn = f32(np.int32(np.uint32(b[0]).view(np.int32)))
                v = sgn * sat(py - u2f(b[1]) + f32(1))
                sa = (sa.astype(f32) + v).astype(f16)
            elif tag == 7:  # DrawFill :535-545
                bd = f16(f32(np.uint32(b[0]).view(np.int32)))
                alpha = (sa + bd).astype(f16)
                if b[4] & 1:  # extension: even-odd, the formula in the reference's comment (:539), all in half
                    r = np.rint((f16(0.5) * alpha).astype(f16)).astype(f16)
                    alpha = np.abs((alpha - (f16(2) * r).astype(f16)).astype(f16))
                else:
                    alpha = np.fmin(np.abs(alpha.astype(f32)), f32(1)).astype(f16)
                blend(b[1], alpha)
                sa = np.zeros((16, 16), f16)
            elif tag == 8:  # Solid :546-551
                blend(b[0], f16(1))
            else:
                raise ValueError(tag)
    out = np.full((16, 16, 4), 255, np.uint8)
    for k in range(3):
        out[:, :, k] = lin2srgb[rgb[k].view(np.uint16)]
    return out
