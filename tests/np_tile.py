"""Independent Python restatement of tileKernel + TileEncoder (TestApp/PietRender.metal:69-454),
written from the Metal source -- one Python object per lane of a 16 x 2-tile threadgroup, numpy
float32 scalars for every float operation (one rounding per operation) -- to cross-check the C
oracle's lane simulation (oracle/pmo_tile.c).  Test infrastructure; slow, small scenes only.

Scene layout (src/lib.rs:15-77, TestApp/GenTypes.h:21-328): SimpleGroup {n_items, items_ix},
ShortBbox[n] at byte 8, 32-byte items:  Circle {tag}; Line {tag, flags, rgba, width, start, end};
Fill {tag, flags, rgba, n_points, points_ix}; Poly {tag, rgba, width, n_points, points_ix}.
Commands (24 bytes, GenTypes.h:330-495) as uint32 [tag, body0..body4]:
  Circle {_, bbox.xy|, bbox.zw|}  Line/Fill {_, start.x, start.y, end.x, end.y}
  Stroke {halfWidth, rgba}  FillEdge {int(sign), y}  DrawFill {int(backdrop), rgba}  Solid {rgba}
"""
import struct

import numpy as np

f32 = np.float32
TILE_W = TILE_H = 16
GROUP_W, GROUP_H = 16, 2  # tiles per threadgroup (PietShaderTypes.h:21-22)
END, CIRCLE, LINE, FILL, STROKE, FILL_EDGE, DRAW_FILL, SOLID, BAIL = range(1, 10)


def bits(x) -> int:
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def sign(x):
    return f32(1.0) if x > 0 else (f32(-1.0) if x < 0 else f32(0.0))


def straddles(s00, s01, s10, s11):
    return f32(f32(f32(s00 * s01) + f32(s00 * s10)) + f32(s00 * s11)) < f32(3.0)


class Encoder:  # TileEncoder :69-157
    def __init__(self):
        self.cmds = []
        self.solid = 0xFFFFFFFF

    def push(self, tag, *body, draws=True):
        self.cmds.append([tag] + list(body) + [0] * (5 - len(body)))
        if draws:
            self.solid = 0

    def solid_cmd(self, rgba):
        if (rgba & 0xFF000000) == 0xFF000000:
            self.solid = rgba
            self.cmds = []  # dst = tileBegin
        self.push(SOLID, rgba, draws=False)

    def end(self):
        if self.solid:
            return [[BAIL, 0, 0, 0, 0, 0]], self.solid
        return self.cmds + [[END, 0, 0, 0, 0, 0]], 0


def tile_lists(scene: bytes, width: int, height: int):
    """-> {(tx, ty): (uint32 [n, 6] commands, solid colour)} for the whole viewport."""
    scene = bytes(scene)
    u32 = lambda o: struct.unpack_from("<I", scene, o)[0]
    flt = lambda o: f32(struct.unpack_from("<f", scene, o)[0])
    pt = lambda base, i: (flt(base + 8 * i), flt(base + 8 * i + 4))
    # Extension (nested groups): an item of type 5 {tag, flags, group_ix} stands for the items of the
    # SimpleGroup at group_ix, in place and in order -- walked here by explicit recursion into
    # (bbox, item offset) pairs, nothing is copied or appended.
    def walk(group, depth=0):
        assert depth <= 32
        gn, gitems = u32(group), u32(group + 4)
        for i in range(gn):
            it = gitems + 32 * i
            if (u32(it) & 0xFFFF) == 5:
                yield from walk(u32(it + 8), depth + 1)
            else:
                yield struct.unpack_from("<4H", scene, group + 8 + 8 * i), it

    flat = list(walk(0))
    n = len(flat)
    bbox = [b for b, _ in flat]
    item_at = [it for _, it in flat]
    tiles_x, tiles_y = (width + 15) // 16, (height + 15) // 16
    out = {}
    for gy in range((tiles_y + GROUP_H - 1) // GROUP_H):
        for gx in range((tiles_x + GROUP_W - 1) // GROUP_W):
            lanes = []
            for tix in range(32):
                tx, ty = gx * GROUP_W + (tix & 15), gy * GROUP_H + (tix >> 4)
                lanes.append({"x0": tx * TILE_W, "y0": ty * TILE_H, "enc": Encoder(), "tx": tx, "ty": ty})
            sx0, sy0 = gx * GROUP_W * TILE_W, gy * GROUP_H * TILE_H
            stw, sth = GROUP_W * TILE_W, GROUP_H * TILE_H
            for ix in range(n):  # the per-32 bitmap only decides WHICH items every lane walks (:191-208)
                bx, by, bz, bw = bbox[ix]
                if not (bz >= sx0 and bx < sx0 + stw and bw >= sy0 and by < sy0 + sth):
                    continue
                item = item_at[ix]
                word = u32(item)
                tag = word & 0xFFFF
                hits = [bz >= L["x0"] and bx < L["x0"] + TILE_W and bw >= L["y0"] and by < L["y0"] + TILE_H for L in lanes]
                if tag == 1:  # Circle :218-222
                    for L, hit in zip(lanes, hits):
                        if hit:
                            L["enc"].push(CIRCLE, (word >> 16) & 1, bx | (by << 16), bz | (bw << 16))  # (+ the ellipse bit, extension D10)
                elif tag == 2:  # Line :223-247
                    rgba, width_ = u32(item + 8), flt(item + 12)
                    s, e = pt(item + 16, 0), pt(item + 16, 1)
                    for L, hit in zip(lanes, hits):
                        if not hit:
                            continue
                        x0, y0 = L["x0"], L["y0"]
                        a = f32(e[1] - s[1]); b = f32(s[0] - e[0])
                        c = f32(-f32(f32(a * s[0]) + f32(b * s[1])))
                        hw = f32(f32(f32(0.5) * width_) + f32(0.5))
                        left = f32(a * f32(f32(x0) - hw)); right = f32(a * f32(f32(x0 + TILE_W) + hw))
                        top = f32(b * f32(f32(y0) - hw)); bot = f32(b * f32(f32(y0 + TILE_H) + hw))
                        s00 = sign(f32(f32(top + left) + c)); s01 = sign(f32(f32(top + right) + c))
                        s10 = sign(f32(f32(bot + left) + c)); s11 = sign(f32(f32(bot + right) + c))
                        if straddles(s00, s01, s10, s11):
                            L["enc"].push(LINE, 0, bits(s[0]), bits(s[1]), bits(e[0]), bits(e[1]))
                            L["enc"].push(STROKE, bits(f32(f32(0.5) * width_)), rgba)
                elif tag == 3:  # Fill :248-362
                    rgba, npts, pix = u32(item + 8), u32(item + 12), u32(item + 16)
                    even_odd = u32(item + 4) & 1  # extension: PietFill.flags bit 0 (src/lib.rs:54)
                    P = lambda k: pt(pix, k)
                    compound = (u32(item + 4) >> 1) & 1  # extension D11: sub-paths, NaN separators carrying the start index

                    def seg_end(k):  # index of the end point of segment k, or None if entry k is a separator
                        if compound and np.isnan(P(k)[0]):
                            return None
                        n = 0 if k + 1 == npts else k + 1
                        if compound and np.isnan(P(n)[0]):
                            n = min(u32(pix + 8 * n + 4), npts - 1)
                        return n

                    backdrop = [f32(0.0)] * 32
                    any_fill = [False] * 32
                    for j in range(0, npts, 16):
                        vote = 0
                        for tix, L in enumerate(lanes):  # phase 1: lane tix looks at segment j + (tix & 15)
                            k = j + (tix & 15)
                            if k >= npts or seg_end(k) is None:
                                continue
                            st, en = P(k), P(seg_end(k))
                            xmin, ymin = min(st[0], en[0]), min(st[1], en[1])
                            xmax, ymax = max(st[0], en[0]), max(st[1], en[1])
                            y0 = L["y0"]
                            if ymax >= y0 and ymin < y0 + TILE_H and xmin < sx0 + stw:
                                a = f32(en[1] - st[1]); b = f32(st[0] - en[0])
                                c = f32(-f32(f32(a * st[0]) + f32(b * st[1])))
                                left = f32(a * f32(sx0)); right = f32(a * f32(sx0 + stw))
                                ytop = max(f32(y0), ymin); ybot = min(f32(y0 + TILE_H), ymax)
                                top = f32(b * ytop); bot = f32(b * ybot)
                                s_tl = sign(f32(f32(f32(right - f32(a * f32(TILE_W))) + f32(f32(y0) * b)) + c))
                                s00 = sign(f32(f32(top + left) + c)); s01 = sign(f32(f32(top + right) + c))
                                s10 = sign(f32(f32(bot + left) + c)); s11 = sign(f32(f32(bot + right) + c))
                                fill_hit = (s_tl == sign(a) and ymin <= y0) or (straddles(s00, s01, s10, s11) and xmax > sx0)
                                if fill_hit:
                                    vote |= 1 << tix
                        for tix, (L, hit) in enumerate(zip(lanes, hits)):  # phase 2: own row's 16 votes
                            fv = (vote >> (tix & 16)) & 0xFFFF
                            while fv:
                                sub = (fv & -fv).bit_length() - 1
                                fv &= fv - 1
                                if not hit:
                                    continue
                                k = j + sub
                                st, en = P(k), P(seg_end(k))
                                xmin, ymin = min(st[0], en[0]), min(st[1], en[1])
                                xmax, ymax = max(st[0], en[0]), max(st[1], en[1])
                                x0, y0 = L["x0"], L["y0"]
                                a = f32(en[1] - st[1]); b = f32(st[0] - en[0])
                                c = f32(-f32(f32(a * st[0]) + f32(b * st[1])))
                                left = f32(a * f32(x0)); right = f32(a * f32(x0 + TILE_W))
                                ytop = max(f32(y0), ymin); ybot = min(f32(y0 + TILE_H), ymax)
                                top = f32(b * ytop); bot = f32(b * ybot)
                                s_tl = sign(f32(f32(left + f32(f32(y0) * b)) + c))
                                s00 = sign(f32(f32(top + left) + c)); s01 = sign(f32(f32(top + right) + c))
                                s10 = sign(f32(f32(bot + left) + c)); s11 = sign(f32(f32(bot + right) + c))
                                if s_tl == sign(a) and ymin <= y0:
                                    backdrop[tix] = f32(backdrop[tix] - s00)
                                enc = L["enc"]
                                if xmin < x0 and xmax > x0:
                                    with np.errstate(all="ignore"):
                                        t = f32(f32(st[0] - f32(x0)) / b)
                                    y_edge = f32(st[1] + f32(f32(en[1] - st[1]) * t))  # mix()
                                    if y_edge >= y0 and y_edge < y0 + TILE_H:
                                        enc.push(FILL_EDGE, int(s00) & 0xFFFFFFFF, bits(y_edge), draws=False)
                                        if b > 0:
                                            enc.push(FILL, 0, bits(st[0]), bits(st[1]), bits(f32(x0)), bits(y_edge), draws=False)
                                        else:
                                            enc.push(FILL, 0, bits(f32(x0)), bits(y_edge), bits(en[0]), bits(en[1]), draws=False)
                                        any_fill[tix] = True
                                    elif straddles(s00, s01, s10, s11):
                                        enc.push(FILL, 0, bits(st[0]), bits(st[1]), bits(en[0]), bits(en[1]), draws=False)
                                        any_fill[tix] = True
                                elif straddles(s00, s01, s10, s11) and xmin < x0 + TILE_W and xmax > x0:
                                    enc.push(FILL, 0, bits(st[0]), bits(st[1]), bits(en[0]), bits(en[1]), draws=False)
                                    any_fill[tix] = True
                    for tix, L in enumerate(lanes):
                        if any_fill[tix]:
                            L["enc"].push(DRAW_FILL, int(backdrop[tix]) & 0xFFFFFFFF, rgba, 0, 0, even_odd)
                        elif (int(backdrop[tix]) % 2 != 0) if even_odd else (backdrop[tix] != 0):  # wholly inside
                            L["enc"].solid_cmd(rgba)
                elif tag == 4:  # Poly :363-446
                    rgba, width_, npts, pix = u32(item + 4), flt(item + 8), u32(item + 12), u32(item + 16)
                    nseg = (npts - 1) & 0xFFFFFFFF
                    if nseg >= 1 << 31:
                        nseg = 0
                    P = lambda k: pt(pix, k)
                    hw = f32(f32(f32(0.5) * width_) + f32(0.5))
                    any_stroke = [False] * 32
                    for j in range(0, nseg, 32):
                        vote = 0
                        for tix, L in enumerate(lanes):  # phase 1: lane tix looks at segment j + tix (its own y0: Q4)
                            k = j + tix
                            if k >= nseg:
                                continue
                            st, en = P(k), P(k + 1)
                            xmin, ymin = min(st[0], en[0]), min(st[1], en[1])
                            xmax, ymax = max(st[0], en[0]), max(st[1], en[1])
                            if ymax > f32(f32(sy0) - hw) and ymin < f32(f32(sy0 + sth) + hw) and xmax > f32(f32(sx0) - hw) and xmin < f32(f32(sx0 + stw) + hw):
                                y0 = L["y0"]
                                a = f32(en[1] - st[1]); b = f32(st[0] - en[0])
                                c = f32(-f32(f32(a * st[0]) + f32(b * st[1])))
                                left = f32(a * f32(f32(sx0) - hw)); right = f32(a * f32(f32(sx0 + stw) + hw))
                                top = f32(b * f32(f32(y0) - hw)); bot = f32(b * f32(f32(y0 + TILE_H) + hw))
                                s00 = sign(f32(f32(top + left) + c)); s01 = sign(f32(f32(top + right) + c))
                                s10 = sign(f32(f32(bot + left) + c)); s11 = sign(f32(f32(bot + right) + c))
                                if straddles(s00, s01, s10, s11):
                                    vote |= 1 << tix
                        for tix, (L, hit) in enumerate(zip(lanes, hits)):
                            pv = vote
                            while pv:
                                sub = (pv & -pv).bit_length() - 1
                                pv &= pv - 1
                                if not hit:
                                    continue
                                k = j + sub
                                st, en = P(k), P(k + 1)
                                xmin, ymin = min(st[0], en[0]), min(st[1], en[1])
                                xmax, ymax = max(st[0], en[0]), max(st[1], en[1])
                                x0, y0 = L["x0"], L["y0"]
                                if ymax > f32(f32(y0) - hw) and ymin < f32(f32(y0 + TILE_H) + hw) and xmax > f32(f32(x0) - hw) and xmin < f32(f32(x0 + TILE_W) + hw):
                                    a = f32(en[1] - st[1]); b = f32(st[0] - en[0])
                                    c = f32(-f32(f32(a * st[0]) + f32(b * st[1])))
                                    left = f32(a * f32(f32(x0) - hw)); right = f32(a * f32(f32(x0 + TILE_W) + hw))
                                    top = f32(b * f32(f32(y0) - hw)); bot = f32(b * f32(f32(y0 + TILE_H) + hw))
                                    s00 = sign(f32(f32(top + left) + c)); s01 = sign(f32(f32(top + right) + c))
                                    s10 = sign(f32(f32(bot + left) + c)); s11 = sign(f32(f32(bot + right) + c))
                                    if straddles(s00, s01, s10, s11):
                                        L["enc"].push(LINE, 0, bits(st[0]), bits(st[1]), bits(en[0]), bits(en[1]))
                                        any_stroke[tix] = True
                    for tix, L in enumerate(lanes):
                        if any_stroke[tix]:
                            L["enc"].push(STROKE, bits(f32(f32(0.5) * width_)), rgba)
            for L in lanes:
                if L["tx"] < tiles_x and L["ty"] < tiles_y:
                    cmds, solid = L["enc"].end()
                    out[(L["tx"], L["ty"])] = (np.array(cmds, dtype=np.uint32).reshape(-1, 6), solid)
    return out
