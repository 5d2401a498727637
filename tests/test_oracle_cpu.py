"""CPU tests of the ORACLE: pins against committed goldens + hand-derived
known-answer tests.  The reference ships no tests or vectors,
so these are the pins this repo creates ("parity unpinned" upstream)."""
import hashlib
import json
import os
import sys
import struct

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


def check_scene(pmo, golden, key, scene, with_image=True):
    g = golden[key]
    w, h = g["viewport"]
    assert scene.size == g["scene_bytes"]
    assert sha(scene) == g["scene_sha256"]
    P = pmo.Ptcl(scene, w, h)
    assert list(P.total_cmds()) == [g["total_cmds"], g["max_cmds_per_tile"]]
    hsh = hashlib.sha256()  # per-tile command lists, as tests/golden/make_golden.py hashes them
    for ty in range(P.tiles_y):
        for tx in range(P.tiles_x):
            c = np.ascontiguousarray(P.cmds(tx, ty), dtype=np.uint32)
            hsh.update(np.uint32(len(c)).tobytes())
            hsh.update(c.tobytes())
    assert hsh.hexdigest() == g["ptcl_sha256"]
    if with_image:
        img = P.render()
        assert sha(img) == g["rgba_sha256"]
        assert int(img.astype(np.uint64).sum()) == g["rgba_sum"]
    P.close()


def test_golden_path_test(pmo, golden):
    check_scene(pmo, golden, "path_test_512x832", pmo.scene_path_test())


def test_golden_cardioid(pmo, golden):
    check_scene(pmo, golden, "cardioid_2048x1536", pmo.scene_cardioid())


def test_golden_tiger_reference_scale(pm, pmo, golden):
    wl = pm.workloads.tiger_reference()
    scene, n_items = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    assert n_items == golden["tiger_x8"]["n_items"] == 304
    check_scene(pmo, golden, "tiger_x8", scene)
    img = pmo.render(scene, 1600, 1600)
    crop = np.load(os.path.join(GOLD, "tiger_x8_crop_704_496_64x64.npy"))
    assert np.array_equal(img[496:560, 704:768], crop)


@pytest.mark.parametrize("name,args", [("tiger_480x270", (480, 270, False)), ("tiger_1920x1080_fills", (1920, 1080, True))])
def test_golden_tiger_configs(pm, pmo, golden, name, args):
    wl = pm.workloads.tiger(args[0], args[1], fills_only=args[2])
    scene, n_items = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    assert n_items == golden[name]["n_items"]
    check_scene(pmo, golden, name, scene)


def test_golden_tiger_4k_scene_and_lists(pm, pmo, golden):
    # image hash of the 4K frame is checked on the GPU side; here scene + command stats only
    wl = pm.workloads.tiger(3840, 2160)
    scene, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    check_scene(pmo, golden, "tiger_3840x2160", scene, with_image=False)


def _extension_scenes(pm, pmo):
    sys.path.insert(0, GOLD)
    import make_golden

    make_golden.pm, make_golden.pmo = pm, pmo
    return make_golden.ext_scenes()


def test_golden_extension_scenes(pm, pmo, golden):
    """The committed pins of the encoder extensions (even-odd, nested groups, ellipses, compound fills:
    decisions D9-D11) and of the SVG document layer (shapes.svg under both rule sets): scene bytes,
    command lists and pixels of the oracle today equal what was committed -- so the oracle and the
    product cannot drift TOGETHER on semantics that have no reference to fall back on."""
    for name, (scene, w, h) in _extension_scenes(pm, pmo).items():
        assert golden[name]["viewport"] == [w, h]
        check_scene(pmo, golden, name, scene)


def test_luts_pinned(pmo, golden):
    a, b, c = pmo.luts()
    assert sha(a) == golden["luts"]["srgb2lin_sha256"]
    assert sha(b) == golden["luts"]["unorm2h_sha256"]
    assert sha(c) == golden["luts"]["lin2srgb_sha256"]
    # known answers: black, white, mid grey
    assert a[0] == 0 and a[255] == 0x3C00 and b[255] == 0x3C00 and b[0] == 0
    assert c[0] == 0 and c[0x3C00] == 255
    assert np.all(np.diff(c[: 0x3C00 + 1].astype(int)) >= 0)  # monotone on [0, 1]
    assert c[0x8001] == 0 and c[0x7C00] == 255  # negative -> 0, +inf -> 255
    # sRGB 0.5 (linear 0.2140) encodes back to 128 +- 1
    lin = a[128]
    assert abs(int(c[lin]) - 128) <= 1


def test_half_vs_f32_accumulator_modes(pmo, golden):
    # D7: the binary16 accumulators of the source vs an all-f32 interpreter differ by at
    # most a few 8-bit steps (1 on simple scenes, 3 where ~100 layers blend in the Tiger)
    for key, bound in (("path_test_512x832", 1), ("cardioid_2048x1536", 1), ("tiger_x8", 4)):
        assert golden[key]["half_vs_f32_max_lsb"] <= bound
    s = pmo.scene_cardioid()
    a = pmo.render(s, 2048, 1536, pmo.MODE_HALF).astype(int)
    b = pmo.render(s, 2048, 1536, pmo.MODE_F32).astype(int)
    assert np.abs(a - b).max() <= 1


# ---- hand-derived known-answer tests ------------------------------------------------


def one_fill_scene(pts, rgba_be_bytes=(0x10, 0x20, 0x30, 0xFF)):
    """A one-item scene written by hand from the layout tables (src/lib.rs:15-68)."""
    pts = np.asarray(pts, np.float32)
    n = len(pts)
    x0, y0 = np.floor(pts.min(0))
    x1, y1 = np.ceil(pts.max(0))
    buf = bytearray()
    buf += struct.pack("<II", 1, 16)                     # SimpleGroup: n_items, items_ix
    buf += struct.pack("<4H", int(x0), int(y0), int(x1), int(y1))
    buf += struct.pack("<II4BII", 3, 0, *rgba_be_bytes, n, 48) + bytes(12)  # PietFill padded to 32
    buf += pts.tobytes()
    return np.frombuffer(bytes(buf), np.uint8)


def test_kat_solid_interior_tile_is_bail(pmo):
    # big opaque triangle: a tile well inside gets {Bail} and the stored colour bytes
    scene = one_fill_scene([(8.5, 8.5), (400.25, 20.5), (30.5, 500.75)])
    P = pmo.Ptcl(scene, 512, 512)
    tx, ty = 4, 4  # pixel (64..80, 64..80) is inside the triangle
    assert P.count(tx, ty) == 1 and P.cmds(tx, ty)[0, 0] == 9  # Cmd_Bail
    assert P.solid(tx, ty) == 0xFF302010
    img = P.render()
    assert img[70, 70].tolist() == [0x10, 0x20, 0x30, 0xFF]
    # a tile nothing touches is untouched opaque white, also a Bail
    assert P.solid(31, 0) == 0xFFFFFFFF and img[5, 500].tolist() == [255, 255, 255, 255]
    P.close()


def test_kat_translucent_interior_solid_is_discarded(pmo):
    # reference quirk (PietRender.metal:127-151): a translucent Solid on an otherwise
    # untouched tile leaves solidColor = white, end() writes Bail, the fill vanishes
    scene = one_fill_scene([(8.5, 8.5), (400.25, 20.5), (30.5, 500.75)], (0, 0, 0x80, 0xE0))
    P = pmo.Ptcl(scene, 512, 512)
    assert P.solid(4, 4) == 0xFFFFFFFF
    assert P.render()[70, 70].tolist() == [255, 255, 255, 255]
    P.close()


def _clip_area(poly, x0, y0, x1, y1):
    """Exact area of polygon `poly` inside the box (Sutherland-Hodgman + shoelace)."""
    def clip(pts, inside, inter):
        out = []
        for i in range(len(pts)):
            a, b = pts[i - 1], pts[i]
            ia, ib = inside(a), inside(b)
            if ia and ib:
                out.append(b)
            elif ia and not ib:
                out.append(inter(a, b))
            elif not ia and ib:
                out.append(inter(a, b))
                out.append(b)
        return out
    def ix(xc):
        return lambda a, b: (xc, a[1] + (b[1] - a[1]) * (xc - a[0]) / (b[0] - a[0]))
    def iy(yc):
        return lambda a, b: (a[0] + (b[0] - a[0]) * (yc - a[1]) / (b[1] - a[1]), yc)
    pts = [tuple(map(float, p)) for p in poly]
    for inside, inter in ((lambda p: p[0] >= x0, ix(x0)), (lambda p: p[0] <= x1, ix(x1)),
                          (lambda p: p[1] >= y0, iy(y0)), (lambda p: p[1] <= y1, iy(y1))):
        pts = clip(pts, inside, inter)
        if not pts:
            return 0.0
    return abs(sum(pts[i - 1][0] * pts[i][1] - pts[i][0] * pts[i - 1][1] for i in range(len(pts)))) / 2.0


def test_kat_fill_area_matches_exact_polygon_area(pmo):
    # general-position convex quadrilateral (no axis-aligned edges: those hit quirk Q1 and
    # the 1e-6 fudge of PietRender.metal:518-520): every pixel's coverage must equal the
    # exact polygon/pixel intersection area
    quad = [(20.25, 20.5), (100.75, 23.5), (97.75, 100.25), (17.25, 96.5)]
    scene = one_fill_scene(quad)
    cov = pmo.fill_coverage(scene, 0, 128, 128)
    assert cov[60, 60] == 1.0 and cov[5, 5] == 0.0 and cov[120, 60] == 0.0 and cov[60, 120] == 0.0
    worst = 0.0
    for y in range(16, 106):
        for x in range(12, 106):
            worst = max(worst, abs(float(cov[y, x]) - _clip_area(quad, x, y, x + 1, y + 1)))
    assert worst < 2e-4, worst


def test_kat_rotated_rect_area(pm, pmo):
    # BASELINE config 1 (b): sum of coverage of the rotated 256x256 square = 65536 +- eps
    wl = pm.workloads.config1_rect(rotated=True)
    scene, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, 1.0), wl.paths.els, wl.affine)
    cov = pmo.fill_coverage(scene, 0, 512, 512)
    assert abs(float(cov.astype(np.float64).sum()) - 65536.0) < 2.0


def _slit_double_square(angle_deg=17.0, centre=(150.3, 140.7), outer=90.0, inner=40.0):
    """One closed polyline that walks a square and then, through a slit, a smaller concentric one
    in the SAME direction: winding 1 in the ring, 2 in the core (general position: rotated)."""
    th = np.deg2rad(angle_deg)
    rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    sq = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], np.float64)
    o, i = sq * outer, sq * inner
    pts = np.concatenate([o, o[:1], i, i[:1]])  # the closing edge walks the slit back: it cancels
    return pts @ rot.T + np.asarray(centre)


def test_kat_even_odd_rule(pm, pmo):
    """Extension: PietFill.flags bit 0 = even-odd (the formula in the reference's comment,
    PietRender.metal:539-540).  Known answers on a doubly wound core: non-zero fills it, even-odd
    leaves a hole; ring pixels are covered either way; total coverage = ring area resp. full
    square; and tiles wholly inside the core (backdrop 2) become nothing instead of Solid."""
    from test_host_cpu import encode_ops

    pts = _slit_double_square()
    nz = encode_ops(pm, [("fill", pts, 0x203040FF)])
    eo = encode_ops(pm, [("fill_eo", pts, 0x203040FF)])
    cov_nz, cov_eo = pmo.fill_coverage(nz, 0, 304, 288), pmo.fill_coverage(eo, 0, 304, 288)
    cy, cx = 141, 150
    assert cov_nz[cy, cx] == 1.0 and cov_eo[cy, cx] == 0.0            # the core
    assert cov_nz[cy, cx + 65] == 1.0 and cov_eo[cy, cx + 65] == 1.0  # the ring
    assert cov_nz[5, 5] == 0.0 and cov_eo[5, 5] == 0.0
    assert abs(float(cov_nz.astype(np.float64).sum()) - 180.0 ** 2) < 3.0
    assert abs(float(cov_eo.astype(np.float64).sum()) - (180.0 ** 2 - 80.0 ** 2)) < 6.0
    assert float(cov_eo.max()) <= 1.0 and float(cov_eo.min()) >= 0.0
    P_nz, P_eo = pmo.Ptcl(nz, 304, 288), pmo.Ptcl(eo, 304, 288)
    tx, ty = cx // 16, cy // 16  # a tile wholly inside the core: backdrop 2
    assert P_nz.solid(tx, ty) == 0xFF403020 and P_nz.cmds(tx, ty)[0, 0] == 9       # Solid(opaque) -> Bail
    assert P_eo.solid(tx, ty) == 0xFFFFFFFF and P_eo.cmds(tx, ty)[0, 0] == 9       # nothing drawn: background
    # the rule travels in the last word of DrawFill; everything else of the lists is identical
    seen = 0
    for yy in range(P_nz.tiles_y):
        for xx in range(P_nz.tiles_x):
            a, b = P_nz.cmds(xx, yy), P_eo.cmds(xx, yy)
            if len(a) == len(b) and len(a) > 1:
                d = a != b
                assert not d[:, :5].any() and set(b[d[:, 5], 5]) <= {1}
                assert all(b[k, 0] == 7 for k in np.nonzero(d[:, 5])[0])
                seen += int(d.any())
    assert seen > 10
    # in the rendered bytes: the hole shows the background, edge tiles of the ring the blended colour
    # (tiles wholly inside the ring hit the reference's translucent-Solid quirk and stay white)
    img = pmo.render(encode_ops(pm, [("fill_eo", pts, 0x20304080)]), 304, 288)
    assert tuple(img[cy, cx]) == (255, 255, 255, 255) and (img[:, :, :3] != 255).any()
    img_nz = pmo.render(encode_ops(pm, [("fill", pts, 0x20304080)]), 304, 288)
    assert (img_nz != img).any()


def test_kat_compound_fill(pm, pmo):
    """Extension D11 ("need to deal with subpaths", src/lib.rs:194): one Fill item made of several
    closed sub-paths.  Known answers: a reversed inner contour is a hole under the non-zero rule, an
    equally oriented one is not -- unless the rule is even-odd; the coverage sums are the exact
    areas; one sub-path alone renders like the plain Fill; two disjoint sub-paths render like two
    Fills; the separators carry the start index of their sub-path."""
    from test_host_cpu import encode_ops

    th = np.deg2rad(11.0)
    rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    sq = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], np.float64)
    outer = (sq * 50.0) @ rot.T + (120.3, 110.6)
    inner = (sq * 20.0) @ rot.T + (120.3, 110.6)
    W, H = 240, 224
    cov = lambda ops: pmo.fill_coverage(encode_ops(pm, ops), 0, W, H).astype(np.float64)
    hole = cov([("fill_cp", [outer, inner[::-1]], 0x203040FF)])
    same = cov([("fill_cp", [outer, inner], 0x203040FF)])
    same_eo = cov([("fill_cp_eo", [outer, inner], 0x203040FF)])
    assert abs(hole.sum() - (100.0 ** 2 - 40.0 ** 2)) < 4.0 and hole[110, 120] == 0.0 and hole[110, 120 + 35] == 1.0
    assert abs(same.sum() - 100.0 ** 2) < 3.0 and same[110, 120] == 1.0
    assert abs(same_eo.sum() - (100.0 ** 2 - 40.0 ** 2)) < 6.0 and same_eo[110, 120] == 0.0
    one = encode_ops(pm, [("fill_cp", [outer], 0x20304080)])
    plain = encode_ops(pm, [("fill", outer, 0x20304080)])
    assert not np.array_equal(one, plain) and np.array_equal(pmo.render(one, W, H), pmo.render(plain, W, H))
    far = outer + (0.0, 0.0)
    a, b = outer * 0.4 + (10.0, 5.0), outer * 0.4 + (120.0, 100.0)
    two = pmo.render(encode_ops(pm, [("fill_cp", [a, b], 0x203040FF)]), W, H)
    assert np.array_equal(two, pmo.render(encode_ops(pm, [("fill", a, 0x203040FF), ("fill", b, 0x203040FF)]), W, H))
    scene = encode_ops(pm, [("fill_cp", [outer, inner[::-1], far[:1]], 0x203040FF)])
    items = int(np.frombuffer(scene[4:8].tobytes(), "<u4")[0])
    flags, npt, pix = (int(v) for v in np.frombuffer(scene[items + 4 : items + 20].tobytes(), "<u4")[[0, 2, 3]])
    assert flags == 2 and npt == 4 + 1 + 4 + 1 + 1 + 1
    words = np.frombuffer(scene[pix : pix + 8 * npt].tobytes(), "<u4").reshape(-1, 2)
    seps = [k for k in range(npt) if words[k, 0] == 0x7FC00000]
    assert seps == [4, 9, 11] and [int(words[k, 1]) for k in seps] == [0, 5, 10]


def test_kat_ellipse(pm, pmo):
    """Extension D10: a Circle item with bit 16 of its tag word set is the ellipse inscribed in its
    bbox (PietRender.metal:488-489's TODO).  Known answers: centre black, just outside the axes
    white, mirror symmetry about both axes (the arithmetic is even in dx and dy), area between the
    ellipses of radii (r - 1) and r (the edge ramp lies inside, as the circle's does), rx == ry
    agrees with the circle to first order, a bbox without area draws nothing, and the flag travels
    in CmdCircle's padding word while plain circles keep a zero there."""
    from test_host_cpu import encode_ops

    cx, cy, rx, ry = 100, 90, 40, 20
    img = pmo.render(encode_ops(pm, [("ellipse", float(cx), float(cy), float(rx), float(ry))]), 208, 176)
    assert tuple(img[cy, cx]) == (0, 0, 0, 255)
    assert img[cy, cx + rx - 2, 0] == 0 and img[cy, cx + rx + 1, 0] == 255
    assert img[cy + ry - 2, cx, 0] == 0 and img[cy + ry + 1, cx, 0] == 255
    assert img[cy + ry - 2, cx + rx - 2, 0] == 255  # the bbox corner is outside
    win = img[cy - ry - 2 : cy + ry + 3, cx - rx - 2 : cx + rx + 3, 0]
    assert np.array_equal(win, win[::-1]) and np.array_equal(win, win[:, ::-1])
    lin = np.where(img[:, :, 0] <= 10, img[:, :, 0] / 255.0 / 12.92, ((img[:, :, 0] / 255.0 + 0.055) / 1.055) ** 2.4)
    area = float((1.0 - lin).sum())
    assert np.pi * (rx - 1) * (ry - 1) < area < np.pi * rx * ry, area
    # rx == ry: F / |grad F| = (r^2 - R^2) / 2r, within (r - R)^2 / 2r of the circle's r - R
    R = 30
    e = pmo.render(encode_ops(pm, [("ellipse", 64.0, 64.0, float(R), float(R))]), 128, 128)[:, :, 0]
    c = pmo.render(encode_ops(pm, [("circle", 64.0, 64.0, float(R))]), 128, 128)[:, :, 0]
    to_lin = lambda v: np.where(v <= 10, v / 255.0 / 12.92, ((v / 255.0 + 0.055) / 1.055) ** 2.4)
    assert np.abs(to_lin(e) - to_lin(c)).max() < 1.0 / (2 * R) + 0.01 and (e != c).any()
    assert (pmo.render(encode_ops(pm, [("ellipse", 64.0, 64.0, 30.0, 0.0)]), 128, 128)[:, :, :3] == 255).all()
    P = pmo.Ptcl(encode_ops(pm, [("ellipse", 40.0, 40.0, 20.0, 9.0), ("circle", 90.0, 40.0, 9.0)]), 128, 80)
    flags = {}
    for ty in range(P.tiles_y):
        for tx in range(P.tiles_x):
            for c_ in P.cmds(tx, ty):
                if c_[0] == 2:
                    flags.setdefault(int(c_[2]) & 0xFFFF, set()).add(int(c_[1]))
    P.close()
    assert flags == {20: {1}, 81: {0}}, flags  # keyed by bbox.x0: the ellipse's lists carry 1, the circle's 0


def test_nested_groups_render_like_the_inlined_items(pm, pmo):
    """Extension: a PietGroup item stands for its children, in place and in order.  Lists, solid
    colours and pixels of nested scenes equal those of the same items in one flat group, and
    the flat form the oracle builds keeps the original bytes in front."""
    from test_host_cpu import encode_ops, extend_ops, inline_ops, random_ops

    for seed in (21, 22, 23):
        tree = extend_ops(seed, random_ops(seed, 120, extent=400.0))
        assert any(op[0] == "group" for op in tree)
        nested, flat = encode_ops(pm, tree), encode_ops(pm, inline_ops(tree))
        assert np.array_equal(pmo.render(nested, 416, 400), pmo.render(flat, 416, 400))
        Pn, Pf = pmo.Ptcl(nested, 416, 400), pmo.Ptcl(flat, 416, 400)
        for ty in range(Pn.tiles_y):
            for tx in range(Pn.tiles_x):
                assert Pn.solid(tx, ty) == Pf.solid(tx, ty) and np.array_equal(Pn.cmds(tx, ty), Pf.cmds(tx, ty)), (seed, tx, ty)
    # malformed nesting is rejected, not followed: a group that contains itself
    bad = encode_ops(pm, [("group", [("circle", 50.0, 50.0, 9.0)])]).copy()
    items = int(np.frombuffer(bad[4:8].tobytes(), "<u4")[0])
    bad[items + 8 : items + 12] = np.frombuffer(np.uint32(0).tobytes(), np.uint8)  # group_ix -> the root
    with pytest.raises(RuntimeError):
        pmo.render(bad, 128, 128)


def test_quirk_q1_axis_aligned_rect_is_reproduced(pm, pmo):
    # BASELINE config 1 (a): horizontal edges crossing a tile's left boundary lose their
    # winding.  The oracle reproduces the source; only self-consistency
    # and the golden hash are asserted, not geometric correctness.
    wl = pm.workloads.config1_rect()
    scene, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, 1.0), wl.paths.els, wl.affine)
    img = pmo.render(scene, 512, 512)
    inside = img[41:296, 73:328]
    frac = float((inside[..., 0] == 0x1F).mean())
    assert frac > 0.5  # most of the rect is painted ...
    assert np.array_equal(img, pmo.render(scene, 512, 512))


def test_empty_and_degenerate_scenes(pmo):
    empty = np.frombuffer(struct.pack("<II", 0, 8), np.uint8)
    img = pmo.render(empty, 40, 24)
    assert img.shape == (24, 40, 4) and (img == 255).all()
    # single-point fill and two-point (zero-area) fill render nothing
    for pts in ([(10.5, 10.5)], [(10.5, 10.5), (30.5, 20.5)]):
        scene = one_fill_scene(pts)
        assert (pmo.render(scene, 48, 48) == 255).all()


def test_tile_pass_in_slices_equals_the_whole_frame(pm, pmo):
    """pmo_ptcl_build_rows: threadgroups are independent, so the tile pass of a frame can be cut into
    slices of tile-group rows (bench.py's all-cores CPU baseline does): lists and pixels of the
    slices equal the whole frame's."""
    wl = pm.workloads.tiger(700, 410)
    scene, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    want = pmo.render(scene, 700, 410)
    whole = pmo.Ptcl(scene, 700, 410)
    tiles_y = (410 + 15) // 16
    groups_y = (tiles_y + 1) // 2
    got = np.zeros_like(want)
    for a in range(0, groups_y, 3):
        b = min(a + 3, groups_y)
        P = pmo.Ptcl(scene, 700, 410, group_rows=(a, b))
        rows = P.render_rows(2 * a, min(2 * b, tiles_y))
        got[32 * a : 32 * a + rows.shape[0]] = rows
        for ty in range(2 * a, min(2 * b, tiles_y)):
            for tx in range(P.tiles_x):
                assert np.array_equal(P.cmds(tx, ty), whole.cmds(tx, ty)) and P.solid(tx, ty) == whole.solid(tx, ty)
        P.close()
    whole.close()
    assert np.array_equal(got, want)


def test_viewport_not_multiple_of_tile(pmo):
    s = pmo.scene_cardioid()
    big = pmo.render(s, 2048, 1536)
    odd = pmo.render(s, 1999, 1501)
    assert np.array_equal(odd, big[:1501, :1999])


def test_flatten_known_answers(pmo):
    # straight-line cubic: err = 0 -> n = 1 -> just the end point
    els = np.zeros(3, pmo.EL_DTYPE)
    els["tag"] = [0, 3, 1]
    els["p"][0, :2] = (1.0, 2.0)
    els["p"][1, :6] = (2.0, 3.0, 3.0, 4.0, 4.0, 5.0)
    els["p"][2, :2] = (9.0, 9.0)
    paths = np.zeros(1, pmo.PATH_DTYPE)
    paths[0] = (0, 3, 1, 0xFF, 0, 0.0)
    scene, n_items = pmo.scene_from_paths(paths, els, (1, 0, 0, 1, 0, 0))
    assert n_items == 1
    n_points, pix = struct.unpack_from("<II", scene.tobytes(), 16 + 12)
    assert n_points == 3
    pts = np.frombuffer(scene.tobytes()[pix : pix + 24], np.float32).reshape(3, 2)
    assert pts.tolist() == [[1, 2], [4, 5], [9, 9]]
    # a curved cubic: n = ceil((err / (432 * 1e-6)) ** (1/6)); last point is exactly p3
    els["p"][1, :6] = (1.0, 102.0, 101.0, 102.0, 101.0, 2.0)
    scene, _ = pmo.scene_from_paths(paths, els, (1, 0, 0, 1, 0, 0))
    n_points, pix = struct.unpack_from("<II", scene.tobytes(), 16 + 12)
    p0, p1, p2, p3 = np.array([1.0, 2.0]), np.array([1.0, 102.0]), np.array([101.0, 102.0]), np.array([101.0, 2.0])
    err = float((((3 * p2 - p3) - (3 * p1 - p0)) ** 2).sum())
    n = max(1, int(np.ceil((err / (432.0 * (0.1 * 1e-2) ** 2)) ** (1.0 / 6.0))))
    assert n_points == 1 + n + 1
    pts = np.frombuffer(scene.tobytes()[pix : pix + 8 * n_points], np.float32).reshape(-1, 2)
    assert pts[n].tolist() == [101.0, 2.0]
    mid = pts[n // 2] if n % 2 == 0 else None
    if mid is not None:  # t = 0.5: (p0 + 3p1 + 3p2 + p3) / 8
        assert np.allclose(mid, (p0 + 3 * p1 + 3 * p2 + p3) / 8, atol=1e-4)


def test_thin_line_rule(pmo):
    # src/lib.rs:353-362: width < 0.7 => alpha *= sqrt(w/0.7), width = 0.7
    els = np.zeros(2, pmo.EL_DTYPE)
    els["tag"] = [0, 1]
    els["p"][0, :2] = (5.0, 5.0)
    els["p"][1, :2] = (50.0, 30.0)
    paths = np.zeros(1, pmo.PATH_DTYPE)
    paths[0] = (0, 2, 2, 0, 0x000000FF, 0.175)
    scene, n_items = pmo.scene_from_paths(paths, els, (1, 0, 0, 1, 0, 0))
    assert n_items == 1
    tag, rgba_b0, rgba_b1, rgba_b2, rgba_b3, width = struct.unpack_from("<I4Bf", scene.tobytes(), 16)
    assert tag == 4 and abs(width - 0.7) < 1e-7
    assert rgba_b3 == int(255.0 * np.sqrt(np.float32(0.175) / np.float32(0.7)))  # 127


def test_render_half_matches_independent_numpy_restatement(pm, pmo):
    """oracle/pmo_render.c against tests/np_render.py, a second restatement of renderKernel and
    of the colour tables written from the Metal source with numpy float32 / float16 arrays:
    the three tables and every non-Bail tile of four scenes (incl. the Tiger), byte for byte."""
    import np_render
    from test_host_cpu import encode_ops, extend_ops, random_ops

    a, b, c = pmo.luts()
    tables = (np_render.lut_srgb_to_linear_half(), np_render.lut_unorm_to_half(), np_render.lut_linear_half_to_srgb8())
    assert np.array_equal(a, tables[0].view(np.uint16))
    assert np.array_equal(b, tables[1].view(np.uint16))
    assert np.array_equal(c, tables[2])
    scenes = [
        (pmo.scene_path_test(), 256, 320),
        (pmo.scene_cardioid(), 480, 352),
        (encode_ops(pm, random_ops(77, 150, extent=300.0)), 320, 304),
        (encode_ops(pm, extend_ops(81, random_ops(81, 150, extent=300.0))), 320, 304),  # even-odd fills, nested groups
    ]
    wl = pm.workloads.tiger(640, 360)
    scenes.append((pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)[0], 640, 360))
    checked = 0
    for scene, w, h in scenes:
        want = pmo.render(scene, w, h)
        P = pmo.Ptcl(scene, w, h)
        for ty in range(P.tiles_y):
            for tx in range(P.tiles_x):
                got = np_render.render_tile(P.cmds(tx, ty), tx, ty, tables)
                if got is None:
                    continue  # Bail: the composite's solid colour, not renderKernel's business
                ref = want[16 * ty : 16 * ty + 16, 16 * tx : 16 * tx + 16]
                assert np.array_equal(got[: ref.shape[0], : ref.shape[1]], ref), (tx, ty)
                checked += 1
        P.close()
    assert checked > 700


def test_tile_lists_match_independent_python_restatement(pm, pmo):
    """oracle/pmo_tile.c (lane simulation in C) against tests/np_tile.py, a second restatement of
    tileKernel + TileEncoder written from the Metal source as Python objects per lane with numpy
    float32 scalars: per-tile command lists word for word, solid colours, and -- through
    tests/np_render.py -- the pixels of the whole independent pipeline, byte for byte."""
    import np_render
    import np_tile
    from test_host_cpu import encode_ops, extend_ops, random_ops

    tables = (np_render.lut_srgb_to_linear_half(), np_render.lut_unorm_to_half(), np_render.lut_linear_half_to_srgb8())
    wl = pm.workloads.tiger(256, 144)
    tiger = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)[0]
    scenes = [
        ("path_test", pmo.scene_path_test(), 512, 832, False),
        ("cardioid", pmo.scene_cardioid(), 480, 352, True),
        ("random77", encode_ops(pm, random_ops(77, 150, extent=300.0)), 320, 304, True),
        ("random78", encode_ops(pm, random_ops(78, 200, extent=700.0)), 700, 500, False),
        ("random79", encode_ops(pm, random_ops(79, 120, extent=120.0)), 130, 100, True),
        ("extended82", encode_ops(pm, extend_ops(82, random_ops(82, 160, extent=300.0))), 320, 304, True),  # even-odd, nested groups
        ("tiger", tiger, 256, 144, True),
    ]
    for name, scene, w, h, pixels in scenes:
        got = np_tile.tile_lists(scene.tobytes(), w, h)
        P = pmo.Ptcl(scene, w, h)
        want_img = pmo.render(scene, w, h) if pixels else None
        for ty in range(P.tiles_y):
            for tx in range(P.tiles_x):
                oc = P.cmds(tx, ty)
                g, solid = got[(tx, ty)]
                assert solid == P.solid(tx, ty), (name, tx, ty)
                assert len(oc) == len(g), (name, tx, ty)
                if g[0, 0] == np_tile.BAIL:
                    assert oc[0, 0] == np_tile.BAIL, (name, tx, ty)
                else:
                    assert np.array_equal(oc, g), (name, tx, ty)
                if pixels and g[0, 0] != np_tile.BAIL:
                    img = np_render.render_tile(g, tx, ty, tables)
                    ref = want_img[16 * ty : 16 * ty + 16, 16 * tx : 16 * tx + 16]
                    assert np.array_equal(img[: ref.shape[0], : ref.shape[1]], ref), (name, tx, ty)
        P.close()


def test_scene_builder_matches_independent_python_restatement(pm, pmo):
    """oracle/pmo_flatten.c + pmo_encoder.c (make_tiger's two passes on parsed paths) against
    tests/np_scene.py, a second restatement from the Rust source: scene bytes, byte for byte."""
    import np_scene
    from piet_metal_amd import _lib

    cases = [pm.workloads.tiger_reference(), pm.workloads.tiger(1920, 1080, fills_only=True), pm.workloads.tiger(3840, 2160),
             pm.workloads.config4_blobs(80, 512)]
    for wl in cases:
        sp = pmo.scaled_paths(wl.paths.paths, wl.width_scale)
        ref, _ = pmo.scene_from_paths(sp, wl.paths.els, wl.affine)
        assert np.array_equal(np.frombuffer(np_scene.scene_from_paths(sp, wl.paths.els, wl.affine), np.uint8), ref), wl.name
    # random path sets: every element kind (quads and closes are ignored by flatten.rs), several
    # subpaths, fills / strokes / both, thin strokes, rotations
    rng = np.random.default_rng(5)
    for _case in range(25):
        els, paths = [], []
        for _ in range(int(rng.integers(1, 30))):
            e0 = len(els)
            for _sub in range(int(rng.integers(1, 4))):
                p = rng.uniform(0, 500, 2)
                els.append((_lib.PM_EL_MOVE, [p[0], p[1], 0, 0, 0, 0]))
                for _seg in range(int(rng.integers(1, 6))):
                    q = [p + rng.uniform(-80, 80, 2) for _ in range(3)]
                    kind = int(rng.integers(0, 4))
                    if kind == 0:
                        els.append((_lib.PM_EL_LINE, [q[0][0], q[0][1], 0, 0, 0, 0])); p = q[0]
                    elif kind == 1:
                        els.append((_lib.PM_EL_QUAD, [q[0][0], q[0][1], q[1][0], q[1][1], 0, 0]))
                    elif kind == 2:
                        els.append((_lib.PM_EL_CURVE, [q[0][0], q[0][1], q[1][0], q[1][1], q[2][0], q[2][1]])); p = q[2]
                    else:
                        els.append((_lib.PM_EL_CLOSE, [0] * 6))
            flags = int(rng.integers(1, 4)) | (4 if rng.random() < 0.3 else 0) | (8 if rng.random() < 0.4 else 0)  # (+ even-odd, compound)
            paths.append((e0, len(els), flags, int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32)),
                          float(rng.choice([0.05, 0.4, 1.0, 5.0]))))
        E = np.zeros(len(els), pm.PathSet.EL_DTYPE)
        for i, (t, p) in enumerate(els):
            E["tag"][i] = t
            E["p"][i] = p
        P = np.array(paths, dtype=pm.PathSet.PATH_DTYPE)
        s = float(rng.choice([0.5, 1.0, 3.3]))
        th = float(rng.uniform(0, 6.28))
        aff = (s * np.cos(th), s * np.sin(th), -s * np.sin(th), s * np.cos(th), float(rng.uniform(-20, 90)), float(rng.uniform(-20, 90)))
        sp = pmo.scaled_paths(P, s)
        ref, _ = pmo.scene_from_paths(sp, E, aff)
        assert np.array_equal(np.frombuffer(np_scene.scene_from_paths(sp, E, aff), np.uint8), ref), _case


def test_fill_pixel_classes():
    """pm_fine.hip FillPass2 evaluates the area integral of renderKernel's Fill (PietRender.metal:517-527) only for
    the pixels the segment's piece of a pixel row passes over; a pixel wholly right of it gets half(wx - wy)
    (area is exactly 1.0f), a pixel wholly left +0 (the numerator is exactly 0 in binary32).  Here the rule the kernel
    uses to tell the three classes apart -- the piece's x extent with 1/8 pixel of slack -- is held against the
    full evaluation of all 16 pixels in numpy binary32, every operation rounded once as in the kernel, over
    coordinates up to 60 000, integer and vertical edges, pieces far left of the tile."""
    f32 = np.float32
    rng = np.random.default_rng(20260927)

    def contrib(fsx, fex, px, tx, ty, wd):
        with np.errstate(all="ignore"):
            sx, ex = f32(fsx - px), f32(fex - px)
            xsx = f32(sx + f32(f32(ex - sx) * tx))
            xsy = f32(sx + f32(f32(ex - sx) * ty))
            xmin = f32(np.fmin(np.fmin(xsx, xsy), f32(1.0)) - f32(1e-6))
            xmax = np.fmax(xsx, xsy)
            b = np.fmin(xmax, f32(1.0))
            c = np.fmax(b, f32(0.0))
            d = np.fmax(xmin, f32(0.0))
            area = f32(f32(f32(b + f32(f32(0.5) * f32(f32(d * d) - f32(c * c)))) - xmin) / f32(xmax - xmin))
            return np.float16(f32(area * wd))

    n = 120000
    scale = rng.choice([16.0, 256.0, 4096.0, 60000.0], n).astype(f32)
    x0 = np.floor(rng.uniform(0, 1, n) * scale / 16).astype(f32) * f32(16)
    fsx = np.where(rng.uniform(0, 1, n) < 0.3, np.round(x0 + rng.uniform(-40, 60, n)), x0 + rng.uniform(-40, 60, n)).astype(f32)
    fex = np.where(rng.uniform(0, 1, n) < 0.2, fsx, fsx + rng.normal(0, 1, n) * rng.choice([0.01, 1, 5, 30, 200], n)).astype(f32)
    y0 = np.floor(rng.uniform(0, 1, n) * scale / 16).astype(f32) * 16
    fsy = (y0 + rng.uniform(-20, 36, n)).astype(f32)
    fey = (fsy + rng.normal(0, 1, n) * rng.choice([0.001, 0.3, 3, 20], n)).astype(f32)
    py = (y0 + rng.integers(0, 16, n)).astype(f32)
    with np.errstate(all="ignore"):
        sy, ey = (fsy - py).astype(f32), (fey - py).astype(f32)
        wx, wy = np.clip(sy, 0, 1).astype(f32), np.clip(ey, 0, 1).astype(f32)
        live = wx != wy
        tx = ((wx - sy).astype(f32) / (ey - sy).astype(f32)).astype(f32)
        ty = ((wy - sy).astype(f32) / (ey - sy).astype(f32)).astype(f32)
        wd = (wx - wy).astype(f32)
        xa = (fsx + ((fex - fsx).astype(f32) * tx).astype(f32)).astype(f32)
        xb = (fsx + ((fex - fsx).astype(f32) * ty).astype(f32)).astype(f32)
        lo, hi = np.fmin(xa, xb), np.fmax(xa, xb)
        ok = (lo >= f32(-1e30)) & (hi <= f32(1e30))
        n_left = np.where(ok, np.clip(np.floor((lo - x0).astype(f32) - f32(1.125)) + 1, 0, 16), 0).astype(int)
        far = ((x0 + f32(15)) - lo).astype(f32) >= f32(30.0)
        first_right = np.where(ok & ~far, np.clip(np.ceil((hi - x0).astype(f32) + f32(0.125)), 0, 16), 16).astype(int)
    first_right = np.maximum(first_right, n_left)
    checked = hot = 0
    for i in np.nonzero(live)[0][:6000]:
        for j in range(16):
            if n_left[i] <= j < first_right[i]:
                hot += 1
                continue
            want = contrib(fsx[i], fex[i], f32(x0[i] + j), tx[i], ty[i], wd[i])
            got = np.float16(0) if j < n_left[i] else np.float16(wd[i])
            assert want == got, (i, j, float(fsx[i]), float(fex[i]), float(tx[i]), float(ty[i]), float(wd[i]), want, got)
            checked += 1
    assert checked > 50000 and hot < 6 * 6000
