"""CPU tests of the product's HOST side (no GPU compute): the encoder against the
oracle's restatement of src/lib.rs, the SVG front-end against an independent
parser, the C-ABI surface, and the absence of any CPU rendering fallback."""
import ctypes as C
import math
import os
import re
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol(pm):
    """Every function include/piet_metal_amd.h declares must be exported by the .so
    and bound by the Python layer."""
    hdr = open(os.path.join(ROOT, "include", "piet_metal_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b((?:pm_[a-z0-9_]+|init_test_scene))\s*\(", hdr))
    names -= {n for n in names if re.search(r"typedef\s+struct[^;]*\b%s\b" % n, hdr)}
    assert len(names) >= 35
    lib = C.CDLL(pm._lib.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in pm._lib.SIGNATURES, f"{n} not bound in piet_metal_amd/_lib.py"
    assert set(pm._lib.SIGNATURES) == names


def test_reference_ffi_symbol_signature(pm):
    # include/piet_metal.h:3 -- void init_test_scene(uint8_t *buf, ssize_t buf_size)
    restype, argtypes = pm._lib.SIGNATURES["init_test_scene"]
    assert restype is None and argtypes == [C.c_void_p, C.c_ssize_t]


def test_no_cpu_fallback_without_gpu(pm):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pm.PietMetalError) as ei:
        pm.Renderer(0)
    assert ei.value.status == pm._lib.PM_ERR_NO_DEVICE
    # and the package never imports the oracle
    import sys

    for name, mod in list(sys.modules.items()):
        if name.startswith("piet_metal_amd"):
            src = getattr(mod, "__file__", "") or ""
            if src.endswith(".py"):
                text = open(src).read()
                assert "import pmo" not in text and "from oracle" not in text, src


def test_scene_cardioid_and_path_test_match_oracle(pm, pmo):
    buf = np.zeros(1 << 16, np.uint8)
    n = pm.scene_cardioid(buf)
    assert np.array_equal(buf[:n], pmo.scene_cardioid())
    assert struct.unpack_from("<II", buf.tobytes(), 0) == (192, 8 + 8 * 192)
    buf[:] = 0
    n = pm.scene_path_test(buf)
    assert np.array_equal(buf[:n], pmo.scene_path_test())
    # memory order of the colour is R,G,B,A (rgba.to_be(), src/lib.rs:200)
    assert buf[16 + 8 : 16 + 12].tolist() == [0x00, 0x00, 0x80, 0xE0]


def _oracle_encode(pmo, ops, cap=1 << 20):
    """Drive the oracle's C encoder with the same op list."""
    lib = pmo.load()

    class Enc(C.Structure):
        _fields_ = [("buf", C.c_void_p), ("cap", C.c_size_t), ("free_space", C.c_size_t), ("group_count", C.c_size_t),
                    ("group_ix", C.c_size_t), ("group_start", C.c_size_t), ("error", C.c_int), ("open", C.c_int),
                    ("depth", C.c_int), ("stack", C.c_size_t * 3 * 32)]

    buf = np.zeros(cap, np.uint8)
    e = Enc()
    lib.pmo_encoder_init(C.byref(e), C.c_void_p(buf.ctypes.data), C.c_size_t(cap))

    def emit(group):
        lib.pmo_encoder_begin_group(C.byref(e), C.c_size_t(len(group)))
        for op in group:
            if op[0] == "circle":
                lib.pmo_encoder_circle(C.byref(e), C.c_double(op[1]), C.c_double(op[2]), C.c_double(op[3]))
            elif op[0] == "ellipse":  # extension: Circle item + ellipse bit
                lib.pmo_encoder_ellipse(C.byref(e), *[C.c_double(v) for v in op[1:5]])
            elif op[0] == "line":
                lib.pmo_encoder_stroke_line(C.byref(e), *[C.c_double(v) for v in op[1:5]], C.c_float(op[5]), C.c_uint32(op[6]))
            elif op[0] == "fill":
                a = np.ascontiguousarray(op[1], np.float64)
                lib.pmo_encoder_fill(C.byref(e), C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), C.c_uint32(op[2]))
            elif op[0] == "fill_eo":  # extension: PietFill.flags bit 0
                a = np.ascontiguousarray(op[1], np.float64)
                lib.pmo_encoder_fill_rule(C.byref(e), C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), C.c_uint32(op[2]), C.c_uint32(1))
            elif op[0] in ("fill_cp", "fill_cp_eo"):  # extension: compound fill (sub-paths)
                subs = [np.ascontiguousarray(q, np.float64).reshape(-1, 2) for q in op[1]]
                a = np.concatenate(subs)
                cnt = np.asarray([len(q) for q in subs], np.uint32)
                lib.pmo_encoder_fill_compound(C.byref(e), C.c_void_p(a.ctypes.data), C.c_void_p(cnt.ctypes.data), C.c_size_t(len(subs)),
                                              C.c_uint32(op[2]), C.c_uint32(1 if op[0] == "fill_cp_eo" else 0))
            elif op[0] == "group":  # extension: nested group
                emit(op[1])
            else:
                a = np.ascontiguousarray(op[1], np.float64)
                lib.pmo_encoder_polyline(C.byref(e), C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), C.c_uint32(op[2]), C.c_float(op[3]))
        lib.pmo_encoder_end_group(C.byref(e))

    emit(ops)
    assert e.error == 0
    return buf[: e.free_space].copy()


def random_ops(seed, n, extent=700.0, opaque_mask=0xFF):
    """Random Encoder calls.  Colours are what the Encoder API takes, 0xRRGGBBAA (it stores them
    byte-swapped, `rgba.to_be()`, src/lib.rs:181/:200/:213): half of the items are opaque."""
    rng = np.random.default_rng(seed)
    ops = []
    for _ in range(n):
        k = rng.integers(0, 4)
        rgba = int(rng.integers(0, 1 << 32))
        if rng.random() < 0.5:
            rgba |= opaque_mask  # opaque: alpha is the LOW byte of the API colour
        if k == 0:
            ops.append(("circle", float(rng.uniform(0, extent)), float(rng.uniform(0, extent)), float(rng.uniform(1, 40))))
        elif k == 1:
            p = rng.uniform(-20, extent, 4)
            ops.append(("line", *[float(v) for v in p], float(rng.uniform(0.3, 30)), rgba))
        elif k == 2:
            m = int(rng.integers(1, 12))
            c = rng.uniform(0, extent, 2)
            pts = c + rng.uniform(-120, 120, (m, 2))
            if rng.random() < 0.3:
                pts = np.round(pts / 16) * 16  # vertices on tile boundaries / axis-aligned edges (quirks Q1-Q3)
            ops.append(("fill", pts, rgba))
        else:
            m = int(rng.integers(1, 40))
            c = rng.uniform(0, extent, 2)
            pts = c + np.cumsum(rng.uniform(-25, 25, (m, 2)), axis=0)
            ops.append(("poly", pts, rgba, float(rng.uniform(0.2, 12))))
    return ops


def encode_ops(pm, ops, cap=1 << 20):
    buf = np.zeros(cap, np.uint8)
    e = pm.Encoder(buf)

    def emit(group):
        e.begin_group(len(group))
        for op in group:
            if op[0] == "circle":
                e.circle((op[1], op[2]), op[3])
            elif op[0] == "ellipse":
                e.ellipse((op[1], op[2]), op[3], op[4])
            elif op[0] == "line":
                e.stroke_line((op[1], op[2]), (op[3], op[4]), op[5], op[6])
            elif op[0] == "fill":
                e.fill(op[1], op[2])
            elif op[0] == "fill_eo":
                e.fill(op[1], op[2], even_odd=True)
            elif op[0] in ("fill_cp", "fill_cp_eo"):
                e.fill_compound(op[1], op[2], even_odd=op[0] == "fill_cp_eo")
            elif op[0] == "group":
                emit(op[1])
            else:
                e.polyline(op[1], op[2], op[3])
        e.end_group()

    emit(ops)
    n = e.bytes_used
    e.close()
    return buf[:n].copy()


def extend_ops(seed, ops):
    """The extensions on top of a random op list: about a third of the fills take the even-odd
    rule (and self-overlap, so that the rule shows), half of the circles become ellipses (some
    degenerate), and runs of items move into nested groups (up to three levels).  inline_ops()
    gives the flat list a nested one must render like."""
    rng = np.random.default_rng(seed ^ 0xE0)
    out = []
    for op in ops:
        if op[0] == "fill" and len(op[1]) >= 3 and rng.random() < 0.3:
            # compound fill: the outline, a hole (the outline shrunk and reversed), sometimes a third
            # contour somewhere else and a one-point sub-path
            pts = np.asarray(op[1], np.float64)
            c = pts.mean(axis=0)
            subs = [pts, (c + (pts - c) * float(rng.uniform(0.2, 0.8)))[::-1]]
            if rng.random() < 0.4:
                subs.append(pts + rng.uniform(-60, 60, 2))
            if rng.random() < 0.2:
                subs.append(pts[:1] + 3.0)
            out.append(("fill_cp_eo" if rng.random() < 0.3 else "fill_cp", subs, op[2]))
            continue
        if op[0] == "circle" and rng.random() < 0.5:
            rx, ry = float(rng.uniform(0.2, 60)), float(rng.uniform(0.2, 60))
            if rng.random() < 0.1:
                ry = 0.0  # a bbox without height: nothing is drawn
            op = ("ellipse", op[1], op[2], rx, ry)
        if op[0] == "fill" and rng.random() < 0.35:
            pts = np.asarray(op[1], np.float64)
            if len(pts) >= 3 and rng.random() < 0.7:  # walk the outline twice, the second time shrunk: winding 2 inside
                c = pts.mean(axis=0)
                pts = np.concatenate([pts, pts[:1], c + (pts - c) * 0.55, c + (pts[:1] - c) * 0.55])
            op = ("fill_eo", pts, op[2])
        out.append(op)

    def nest(lst, depth):
        res, i = [], 0
        while i < len(lst):
            if depth < 3 and rng.random() < 0.15:
                k = int(rng.integers(0, 7))  # (empty groups too)
                res.append(("group", nest(lst[i : i + k], depth + 1)))
                i += k
            else:
                res.append(lst[i])
                i += 1
        return res

    return nest(out, 0)


def inline_ops(ops):
    flat = []
    for op in ops:
        flat.extend(inline_ops(op[1]) if op[0] == "group" else [op])
    return flat


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_encoder_random_scenes_match_oracle(pm, pmo, seed):
    ops = random_ops(seed, 60)
    assert np.array_equal(encode_ops(pm, ops), _oracle_encode(pmo, ops))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_encoder_nested_groups_and_fill_rule_match_oracle(pm, pmo, seed):
    """The two encoder extensions (nested groups, PietFill.flags) byte for byte against the
    oracle's encoder, and the layout itself: a group is an item {5, 0, group_ix} of its parent
    whose box is the union of its children's boxes."""
    ops = extend_ops(seed, random_ops(seed, 80))
    got = encode_ops(pm, ops)
    assert np.array_equal(got, _oracle_encode(pmo, ops))
    u32 = lambda o: int(np.frombuffer(got[o : o + 4].tobytes(), "<u4")[0])
    def check(group, ops_):
        assert u32(group) == len(ops_)
        items = u32(group + 4)
        assert items == group + 8 + 8 * len(ops_)
        for i, op in enumerate(ops_):
            it = items + 32 * i
            if op[0] == "group":
                assert u32(it) == 5 and u32(it + 4) == 0
                child = u32(it + 8)
                check(child, op[1])
                bb = np.frombuffer(got[group + 8 + 8 * i : group + 16 + 8 * i].tobytes(), "<u2")
                cb = np.frombuffer(got[child + 8 : child + 8 + 8 * len(op[1])].tobytes(), "<u2").reshape(-1, 4)
                if len(cb):
                    assert list(bb) == [cb[:, 0].min(), cb[:, 1].min(), cb[:, 2].max(), cb[:, 3].max()]
                else:
                    assert list(bb) == [0, 0, 0, 0]
            elif op[0] == "fill_eo":
                assert u32(it) == 3 and u32(it + 4) == 1
            elif op[0] == "fill":
                assert u32(it) == 3 and u32(it + 4) == 0
    check(0, ops)


def test_encoder_misuse_is_an_error_not_a_crash(pm):
    buf = np.zeros(4096, np.uint8)
    e = pm.Encoder(buf)
    e.begin_group(1)
    e.circle((10, 10), 3)
    with pytest.raises(pm.PietMetalError):  # assert!(group_ix < group_count), src/lib.rs:152
        e.circle((20, 20), 3)
    e2 = pm.Encoder(np.zeros(4096, np.uint8))
    e2.begin_group(2)
    e2.circle((10, 10), 3)
    with pytest.raises(pm.PietMetalError):  # assert_eq!(group_ix, group_count), src/lib.rs:147
        e2.end_group()
    e3 = pm.Encoder(np.zeros(64, np.uint8))
    with pytest.raises(pm.PietMetalError) as ei:  # slice index panic in write_struct, src/lib.rs:127
        e3.begin_group(100)
    assert ei.value.status == pm._lib.PM_ERR_CAPACITY
    e4 = pm.Encoder(np.zeros(4096, np.uint8))
    e4.begin_group(1)
    with pytest.raises(pm.PietMetalError):  # .expect("encoded empty points vector"), src/lib.rs:238
        e4.fill(np.zeros((0, 2)), 0xFF)


def test_encoder_write_struct_and_encode_points(pm):
    """The two remaining `pub` methods of Encoder (src/lib.rs:122 write_struct, :224 encode_points): a Fill
    item assembled by hand from them is byte for byte what Encoder::fill (:195-207) writes."""
    import struct

    pts = [(10.25, 20.5), (300.125, 25.0), (150.0, 270.75), (0.1, 99.9)]
    a = np.zeros(512, np.uint8)
    ea = pm.Encoder(a)
    ea.begin_group(1)
    ea.fill(pts, 0x11223344)
    ea.end_group()
    b = np.zeros(512, np.uint8)
    eb = pm.Encoder(b)
    eb.begin_group(1)
    ix, bb = eb.encode_points(pts)
    assert ix == 8 + 8 + 32 and bb == (0.1, 20.5, 300.125, 270.75)  # f64 box of the points (Rect::from_points / union_pt)
    # ShortBbox::from_rect (:88-97): floor / ceil;  PietFill {item_type 3, flags, rgba_color (byte-swapped, :200), n_points, points_ix}
    eb.write_struct(8, struct.pack("<4H", 0, 20, 301, 271))
    eb.write_struct(16, struct.pack("<5I", 3, 0, 0x44332211, len(pts), ix))
    assert eb.bytes_used == ea.bytes_used and a.tobytes() == b.tobytes()
    # the Rust slice index panics past the end of the buffer: an error code here, nothing written
    with pytest.raises(pm.PietMetalError) as ei:
        eb.write_struct(510, b"\x01\x02\x03\x04")
    assert ei.value.status == pm._lib.PM_ERR_CAPACITY and b[510] == 0
    e4 = pm.Encoder(np.zeros(64, np.uint8))
    with pytest.raises(pm.PietMetalError):  # .expect("encoded empty points vector"), :238
        e4.encode_points(np.zeros((0, 2)))


def test_bbox_rules(pm):
    # fill: floor/ceil of the point box; polyline/line: inflated by width/2; clamp to u16
    buf = np.zeros(4096, np.uint8)
    e = pm.Encoder(buf)
    e.begin_group(3)
    e.fill([(10.2, 20.7), (30.9, 25.1), (-5.0, 70000.0)], 0xFF)
    e.polyline([(10.2, 20.7), (30.9, 25.1)], 0xFF, 3.0)
    e.stroke_line((30.9, 25.1), (10.2, 20.7), 3.0, 0xFF)
    e.end_group()
    bb = np.frombuffer(buf[8:32].tobytes(), np.uint16).reshape(3, 4)
    assert bb[0].tolist() == [0, 20, 31, 65535]
    assert bb[1].tolist() == [8, 19, 33, 27]
    assert bb[2].tolist() == [8, 19, 33, 27]


def test_parse_color(pm):
    assert pm.parse_color("#fff") == 0xFFFFFFFF
    assert pm.parse_color("#FFC") == 0xFFFFCCFF
    assert pm.parse_color("#cc7226") == 0xCC7226FF
    assert pm.parse_color("none") == 0xFF00FF80  # src/lib.rs:383


# ---- independent SVG path-data parser (spec-level, pure Python) -----------------------

_NUM = re.compile(r"[+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?")


def py_parse_path(d):
    """Returns [(tag, [coords...])] with tags M L C Z only (H/V -> L, S -> C); raises on A/Q/T."""
    i, n = 0, len(d)
    out = []
    cur = start = ctrl = (0.0, 0.0)
    cmd = None

    def skip():
        nonlocal i
        while i < n and d[i] in " \t\r\n,":
            i += 1

    def num():
        nonlocal i
        skip()
        m = _NUM.match(d, i)
        assert m, (i, d[i : i + 20])
        i = m.end()
        return float(m.group())

    while True:
        skip()
        if i >= n:
            break
        if d[i].isalpha():
            cmd = d[i]
            i += 1
        elif cmd in "Mm":
            cmd = "L" if cmd == "M" else "l"
        rel = cmd.islower()
        C_ = cmd.upper()
        if C_ == "Z":
            out.append(("Z", []))
            cur = ctrl = start
            continue

        def pt():
            x, y = num(), num()
            return (x + cur[0], y + cur[1]) if rel else (x, y)

        if C_ == "M":
            p = pt(); out.append(("M", [*p])); cur = start = ctrl = p
        elif C_ == "L":
            p = pt(); out.append(("L", [*p])); cur = ctrl = p
        elif C_ == "H":
            x = num(); x = x + cur[0] if rel else x
            cur = ctrl = (x, cur[1]); out.append(("L", [*cur]))
        elif C_ == "V":
            y = num(); y = y + cur[1] if rel else y
            cur = ctrl = (cur[0], y); out.append(("L", [*cur]))
        elif C_ == "C":
            p1, p2, p3 = pt(), pt(), pt()
            out.append(("C", [*p1, *p2, *p3])); ctrl, cur = p2, p3
        elif C_ == "S":
            p1 = (2.0 * cur[0] - ctrl[0], 2.0 * cur[1] - ctrl[1])
            p2, p3 = pt(), pt()
            out.append(("C", [*p1, *p2, *p3])); ctrl, cur = p2, p3
        elif C_ == "Q":
            p1, p2 = pt(), pt()
            out.append(("Q", [*p1, *p2])); ctrl, cur = p1, p2
        elif C_ == "T":
            prev_quad = bool(out) and out[-1][0] == "Q"
            p1 = (2.0 * cur[0] - ctrl[0], 2.0 * cur[1] - ctrl[1]) if prev_quad else cur
            p2 = pt()
            out.append(("Q", [*p1, *p2])); ctrl, cur = p1, p2
        else:
            raise NotImplementedError(cmd)
    return out


def _svg_path_elements(d):
    tags = {"M": 0, "L": 1, "Q": 2, "C": 3, "Z": 4}
    return [(tags[t], coords) for t, coords in py_parse_path(d)]


def test_svg_front_end_matches_independent_parser(pm):
    svg = open(os.path.join(ROOT, "piet_metal_amd", "assets", "Ghostscript_Tiger.svg")).read()
    ps = pm.PathSet.from_svg(svg)
    tags = {"M": 0, "L": 1, "C": 3, "Z": 4}
    path_tags = re.findall(r"<path\b([^>]*)>", svg)
    assert len(path_tags) == len(ps.paths) == 138
    checked = 0
    for attrs, path in zip(path_tags, ps.paths):
        d = re.search(r'\bd="([^"]*)"', attrs).group(1)
        fill = re.search(r'\bfill="([^"]*)"', attrs)
        stroke = re.search(r'\bstroke="([^"]*)"', attrs)
        sw = re.search(r'\bstroke-width="([^"]*)"', attrs)
        assert bool(path["flags"] & 1) == bool(fill) and bool(path["flags"] & 2) == bool(stroke)
        if fill:
            assert path["fill_rgba"] == pm.parse_color(fill.group(1))
        if stroke:
            assert path["stroke_rgba"] == pm.parse_color(stroke.group(1))
            assert path["stroke_width"] == np.float32(float(sw.group(1)))
        if re.search(r"[aA]", d):
            continue  # arcs: product-defined conversion, covered by test_svg_arcs
        want = py_parse_path(d)
        got = ps.els[path["el_begin"] : path["el_end"]]
        assert len(got) == len(want)
        for g, (t, coords) in zip(got, want):
            assert g["tag"] == tags[t]
            assert g["p"][: len(coords)].tolist() == coords  # bit-exact f64
        checked += 1
    assert checked == 132


def test_svg_arcs(pm):
    # quarter circle, radius 10, centre (10, 0): ends exactly on the end point, stays on the circle
    ps = pm.PathSet.from_svg('<svg><path d="M0 0 a10 10 0 0 1 10 -10" fill="#000"/></svg>')
    els = ps.els
    assert els["tag"].tolist() == [0, 3]
    assert els["p"][1, 4:6].tolist() == [10.0, -10.0]
    p0 = np.array([0.0, 0.0]); p1, p2, p3 = els["p"][1, 0:2], els["p"][1, 2:4], els["p"][1, 4:6]
    for t in np.linspace(0, 1, 17):
        q = (1 - t) ** 3 * p0 + 3 * (1 - t) ** 2 * t * p1 + 3 * (1 - t) * t * t * p2 + t ** 3 * p3
        assert abs(np.hypot(q[0] - 10.0, q[1]) - 10.0) < 3e-3
    # packed flags ("a1 1 0 016 6") and the reject switch
    ps = pm.PathSet.from_svg('<svg><path d="M0 0a4 4 0 016 6z" fill="#000"/><path d="M1 1L2 2" stroke="#000" stroke-width="2"/></svg>')
    assert len(ps.paths) == 2
    ps = pm.PathSet.from_svg('<svg><path d="M0 0a4 4 0 016 6z" fill="#000"/><path d="M1 1L2 2" stroke="#000" stroke-width="2"/></svg>', reject_arc_paths=True)
    assert len(ps.paths) == 1 and ps.paths[0]["flags"] == 2 and ps.paths[0]["stroke_width"] == 2.0
    assert len(pm.PathSet.tiger(reject_arc_paths=True).paths) == 132  # 6 arc paths


def test_svg_syntax_edge_cases(pm):
    ps = pm.PathSet.from_svg("<svg><!-- c --><path d='M.5.5-1-1 1e1,2E-1 z m1 1h2v-3H0V0' fill='#abc'/></svg>")
    e = ps.els
    assert e["tag"].tolist() == [0, 1, 1, 4, 0, 1, 1, 1, 1]
    assert e["p"][:3, :2].tolist() == [[0.5, 0.5], [-1.0, -1.0], [10.0, 0.2]]
    assert e["p"][4, :2].tolist() == [1.5, 1.5]  # relative moveto after Z starts from the sub-path start
    assert e["p"][5:, :2].tolist() == [[3.5, 1.5], [3.5, -1.5], [0.0, -1.5], [0.0, 0.0]]
    bad = pm.PathSet.from_svg("<svg><path d='M0 0 L' fill='#000'/><path d='M0 0L1 1' fill='#000'/></svg>")
    assert len(bad.paths) == 1  # a path kurbo would reject is skipped (src/lib.rs:296), not fatal
    with pytest.raises(pm.PietMetalError):
        pm.PathSet.from_svg("<svg><path fill='#000'/></svg>")  # .attribute("d").unwrap(), src/lib.rs:295


# ---- independent SVG document walker (pure Python, from the SVG 1.1 spec) -----------------------
# Second witness for the document layer of the C++ front-end: element nesting, property inheritance,
# `style`, opacity, transforms, fill-rule, basic shapes.  It produces (flags, fill, stroke, width) per
# drawn element and the element's geometry as absolute path elements in root user space; arcs are
# compared through sample points (the conversion to cubics is product-defined).

def _svg_walk(svg_text, spec_defaults=False):
    import xml.etree.ElementTree as ET

    NAMES = {"black": 0x000000, "white": 0xFFFFFF, "red": 0xFF0000, "green": 0x008000, "blue": 0x0000FF, "teal": 0x008080,
             "purple": 0x800080, "yellow": 0xFFFF00, "orange": 0xFFA500, "gray": 0x808080}

    def paint(v, cur):
        v = v.strip()
        if v in ("none", "transparent") or v.startswith("url("):
            return None
        if v.startswith("#"):
            h = v[1:]
            if len(h) == 3:
                h = "".join(c * 2 for c in h)
            return int(h, 16)
        if v.startswith("rgb("):
            parts = [t.strip() for t in v[4:-1].split(",")]
            ch = [float(t[:-1]) * 255 / 100 if t.endswith("%") else float(t) for t in parts]
            return (int(round(min(255, max(0, ch[0])))) << 16) | (int(round(min(255, max(0, ch[1])))) << 8) | int(round(min(255, max(0, ch[2]))))
        return NAMES.get(v, cur)

    def mat_mul(A, B):  # A * B, B applied first; [a b c d e f]
        a, b, c, d, e, f = A
        g, h, i, j, k, l = B
        return [a * g + c * h, b * g + d * h, a * i + c * j, b * i + d * j, a * k + c * l + e, b * k + d * l + f]

    def transform(v):
        M = [1, 0, 0, 1, 0, 0]
        for name, args in re.findall(r"([a-zA-Z]+)\s*\(([^)]*)\)", v):
            x = [float(t) for t in re.split(r"[\s,]+", args.strip()) if t]
            if name == "matrix":
                T = x
            elif name == "translate":
                T = [1, 0, 0, 1, x[0], x[1] if len(x) > 1 else 0]
            elif name == "scale":
                T = [x[0], 0, 0, x[1] if len(x) > 1 else x[0], 0, 0]
            elif name == "rotate":
                c, s_ = math.cos(math.radians(x[0])), math.sin(math.radians(x[0]))
                T = [c, s_, -s_, c, 0, 0]
                if len(x) == 3:
                    T = mat_mul(mat_mul([1, 0, 0, 1, x[1], x[2]], T), [1, 0, 0, 1, -x[1], -x[2]])
            elif name == "skewX":
                T = [1, 0, math.tan(math.radians(x[0])), 1, 0, 0]
            elif name == "skewY":
                T = [1, math.tan(math.radians(x[0])), 0, 1, 0, 0]
            M = mat_mul(M, T)
        return M

    def opac(v):
        v = v.strip()
        return min(1.0, max(0.0, float(v[:-1]) / 100 if v.endswith("%") else float(v)))

    out = []

    def visit(el, st, used=0):
        tag = el.tag.split("}")[-1]
        if tag in ("defs", "clipPath", "mask", "pattern", "marker") or (tag == "symbol" and not used):
            return
        st = dict(st)
        props = {k: v for k, v in el.attrib.items()}

        def decls_of(text):
            d = {}
            for part in text.split(";"):
                if ":" in part:
                    k, v = part.split(":", 1)
                    d[k.strip()] = v.strip()
            return d

        # cascade: presentation attributes < element rules < class rules < id rules < style attribute
        sources = [props]
        classes = props.get("class", "").split()
        for kind in (0, 1, 2):
            for rk, rname, rdecl in sheet:
                if rk == kind and ((kind == 0 and rname == tag) or (kind == 1 and rname in classes) or (kind == 2 and rname == props.get("id"))):
                    sources.append(decls_of(rdecl))
        sources.append(decls_of(props.get("style", "")))
        own_opacity = None
        for src in sources:
            for k in ("fill", "stroke"):
                if k in src:
                    st[k] = paint(src[k], st[k])
            if "stroke-width" in src:
                st["width"] = float(np.float32(float(src["stroke-width"])))
            if "fill-rule" in src:
                st["evenodd"] = src["fill-rule"].strip() == "evenodd"
            if "fill-opacity" in src:
                st["fo"] = opac(src["fill-opacity"])
            if "stroke-opacity" in src:
                st["so"] = opac(src["stroke-opacity"])
            if "opacity" in src:
                own_opacity = opac(src["opacity"])
        if own_opacity is not None:
            st["op"] = st["op"] * own_opacity
        if "transform" in props:
            st["ctm"] = mat_mul(st["ctm"], transform(props["transform"]))
        if tag in ("g", "svg", "a", "switch", "symbol"):
            for ch in el:
                visit(ch, st)
            return
        if tag == "use":  # the referenced element, here, under this element's properties and transform * translate(x, y)
            num = lambda k: float(props[k]) if k in props else 0.0
            st["ctm"] = mat_mul(st["ctm"], [1, 0, 0, 1, num("x"), num("y")])
            href = props.get("href") or props.get("{http://www.w3.org/1999/xlink}href") or ""
            target = next((e for e in root.iter() if e.get("id") == href[1:]), None) if href.startswith("#") else None
            if target is not None and used < 8:
                visit(target, st, used + 1)
            return
        f = lambda k, dflt=0.0: float(re.match(r"\s*([-+0-9.eE]+)", props[k]).group(1)) if k in props else dflt
        geom, closed = None, True
        if tag == "path":
            geom = ("path", props["d"])
        elif tag == "rect":
            w, h = f("width"), f("height")
            if w > 0 and h > 0:
                rx, ry = f("rx", -1.0), f("ry", -1.0)
                if rx < 0 and ry < 0: rx = ry = 0.0
                elif rx < 0: rx = ry
                elif ry < 0: ry = rx
                geom = ("rect", f("x"), f("y"), w, h, min(rx, w / 2), min(ry, h / 2))
        elif tag in ("circle", "ellipse"):
            rx, ry = (f("r"), f("r")) if tag == "circle" else (f("rx"), f("ry"))
            if rx > 0 and ry > 0:
                geom = ("ellipse", f("cx"), f("cy"), rx, ry)
        elif tag == "line":
            geom, closed = ("poly", [(f("x1"), f("y1")), (f("x2"), f("y2"))], False), False
        elif tag in ("polyline", "polygon"):
            nums = [float(t) for t in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?", props.get("points", ""))]
            pts = list(zip(nums[0::2], nums[1::2]))
            if len(pts) >= 2:
                geom, closed = ("poly", pts, tag == "polygon"), tag == "polygon"
        if geom is None:
            return
        flags, fill_rgba, stroke_rgba, width = 0, 0, 0, 0.0
        if st["fill"] is not None and (closed or tag in ("path", "polyline")):
            flags |= 1 | (4 if st["evenodd"] else 0)
            if spec_defaults and tag == "path" and len(re.findall(r"[Mm]", props.get("d", ""))) > 1:
                flags |= 8  # SVG fills the sub-paths of a path together: one compound item
            fill_rgba = (st["fill"] << 8) | int(round(255 * min(1.0, max(0.0, st["op"] * st["fo"]))))
        if st["stroke"] is not None:
            flags |= 2
            stroke_rgba = (st["stroke"] << 8) | int(round(255 * min(1.0, max(0.0, st["op"] * st["so"]))))
            a, b, c, d, _, _ = st["ctm"]
            ident = st["ctm"] == [1, 0, 0, 1, 0, 0]
            width = st["width"] if ident else float(np.float32(st["width"] * math.sqrt(abs(a * d - b * c))))
        if flags & 3:
            out.append({"flags": flags, "fill": fill_rgba, "stroke": stroke_rgba, "width": width, "geom": geom, "ctm": st["ctm"]})

    root = ET.fromstring(svg_text)
    sheet = []  # (kind, name, declarations) of the <style> rules with a simple selector, in document order
    for el in root.iter():
        if el.tag.split("}")[-1] == "style":
            css = re.sub(r"/\*.*?\*/", "", el.text or "", flags=re.S)
            for sel, body in re.findall(r"([^{}]+)\{([^{}]*)\}", css):
                for one in sel.split(","):
                    one = one.strip()
                    if re.fullmatch(r"[.#]?[A-Za-z0-9_-]+", one):
                        sheet.append((1 if one[0] == "." else 2 if one[0] == "#" else 0, one.lstrip(".#"), body))
    visit(root, {"fill": 0 if spec_defaults else None, "stroke": None, "width": 1.0, "evenodd": False, "fo": 1.0, "so": 1.0, "op": 1.0,
                 "ctm": [1, 0, 0, 1, 0, 0]})
    return out


def _apply(M, p):
    return (M[0] * p[0] + M[2] * p[1] + M[4], M[1] * p[0] + M[3] * p[1] + M[5])


def _bez(p0, els_row, t):
    tag = int(els_row["tag"])
    P = els_row["p"]
    if tag == 1:
        return (p0[0] + (P[0] - p0[0]) * t, p0[1] + (P[1] - p0[1]) * t)
    if tag == 2:
        mt = 1 - t
        return (mt * mt * p0[0] + 2 * mt * t * P[0] + t * t * P[2], mt * mt * p0[1] + 2 * mt * t * P[1] + t * t * P[3])
    mt = 1 - t
    return (mt ** 3 * p0[0] + 3 * mt * mt * t * P[0] + 3 * mt * t * t * P[2] + t ** 3 * P[4],
            mt ** 3 * p0[1] + 3 * mt * mt * t * P[1] + 3 * mt * t * t * P[3] + t ** 3 * P[5])


def test_encoder_paths_with_curves_match_the_flatten_pipeline(pm, pmo):
    """pm_encoder_fill_path / pm_encoder_stroke_path (host flatten + encode of one path with curves
    and sub-paths: the signature src/lib.rs:194 anticipates) item by item against the oracle's
    make_tiger restatement on whole documents under the identity transform: the same scene bytes --
    the Tiger (fills, strokes, thin lines, arcs as cubics) and shapes.svg with SVG semantics
    (compound fills, even-odd)."""
    docs = [pm.PathSet.tiger(), pm.PathSet.from_svg(open(os.path.join(ROOT, "tests", "data", "shapes.svg")).read(), spec_defaults=True)]
    ident = (1.0, 0.0, 0.0, 1.0, 0.0, 0.0)
    for ps in docs:
        want, n_items = pmo.scene_from_paths(pmo.scaled_paths(ps.paths, 1.0), ps.els, ident)
        buf = np.zeros(len(want) + 4096, np.uint8)
        e = pm.Encoder(buf)
        e.begin_group(n_items)
        for p in ps.paths:
            els = ps.els[int(p["el_begin"]) : int(p["el_end"])]
            fl = int(p["flags"])
            if fl & pm._lib.PM_PATH_FILL:
                e.fill_path(els, int(p["fill_rgba"]), even_odd=bool(fl & pm._lib.PM_PATH_EVEN_ODD), compound=bool(fl & pm._lib.PM_PATH_COMPOUND))
            if fl & pm._lib.PM_PATH_STROKE:
                e.stroke_path(els, int(p["stroke_rgba"]), float(p["stroke_width"]))
        e.end_group()
        assert np.array_equal(buf[: e.bytes_used], want)
    # a LineTo before any MoveTo is an error (the reference panics), not a guess
    bad = np.zeros(1, pm.PathSet.EL_DTYPE)
    bad["tag"][0] = pm._lib.PM_EL_LINE
    e = pm.Encoder(np.zeros(1024, np.uint8))
    e.begin_group(1)
    with pytest.raises(pm.PietMetalError):
        e.fill_path(bad, 0xFF)


def test_svg_colour_keywords_match_an_independent_table(pm):
    """The 147 colour keywords of SVG 1.1 against Pillow's table (an independent source)."""
    ImageColor = pytest.importorskip("PIL.ImageColor")
    checked = 0
    for name in ImageColor.colormap:
        if name == "rebeccapurple":  # CSS4, not SVG 1.1: unknown keywords leave the inherited paint
            ps = pm.PathSet.from_svg('<svg><rect width="5" height="5" fill="%s"/></svg>' % name)
            assert len(ps.paths) == 0
            continue
        r, g, b = ImageColor.getrgb(name)[:3]
        ps = pm.PathSet.from_svg('<svg><rect width="5" height="5" fill="%s"/></svg>' % name)
        assert int(ps.paths[0]["fill_rgba"]) == ((r << 24) | (g << 16) | (b << 8) | 0xFF), name
        checked += 1
    assert checked == 147


def test_svg_parser_survives_mutated_documents(pm):
    """tests/dev/fuzz_svg.py: 1 500 mutated copies of shapes.svg and of the Tiger's head -- every one is
    either parsed or rejected with an error code (no crash, no hang)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "dev"))
    import fuzz_svg

    ok, err, _ = fuzz_svg.run(7, 1500)
    assert ok + err == 1500 and ok > 100 and err > 100


def test_svg_display_and_visibility(pm):
    """`display: none` removes an element and its subtree (descendants cannot undo it); `visibility`
    is inherited and a descendant may turn it back on; both through attributes, style and sheets."""
    svg = '''<svg><style>.off { display: none } .ghost { visibility: hidden }</style>
      <g display="none"><rect width="5" height="5" fill="red" display="inline"/></g>
      <g class="ghost"><rect width="5" height="5" fill="green"/><rect width="6" height="6" fill="blue" visibility="visible"/></g>
      <rect width="7" height="7" fill="black" style="display:none"/><rect class="off" width="7" height="7" fill="black"/>
      <rect width="8" height="8" fill="#123" stroke="#456" visibility="collapse"/><rect width="8" height="8" fill="#123"/></svg>'''
    ps = pm.PathSet.from_svg(svg)
    assert [hex(int(p["fill_rgba"])) for p in ps.paths] == ["0xffff", "0x112233ff"]


def test_svg_lengths_units_and_percentages(pm):
    """SVG 1.1 section 7.10: absolute units at 96 per inch, percentages of the outermost viewBox
    (width for x-like, height for y-like, sqrt((w^2 + h^2) / 2) for radii), on shapes and stroke widths."""
    ps = pm.PathSet.from_svg('<svg viewBox="0 0 200 100"><rect width="100%" height="50%" fill="red"/>'
                             '<circle cx="50%" cy="50%" r="10%" fill="blue"/>'
                             '<line x1="1in" y1="12pt" x2="10mm" y2="2pc" stroke="#000" stroke-width="1.5pt"/></svg>')
    els = lambda k: ps.els[int(ps.paths[k]["el_begin"]) : int(ps.paths[k]["el_end"])]
    assert [tuple(e["p"][:2]) for e in els(0)[:4]] == [(0.0, 0.0), (200.0, 0.0), (200.0, 50.0), (0.0, 50.0)]
    r = 0.1 * math.sqrt((200.0 ** 2 + 100.0 ** 2) / 2.0)
    assert tuple(els(1)[0]["p"][:2]) == pytest.approx((100.0 + r, 50.0))
    assert tuple(els(2)[0]["p"][:2]) == (96.0, 16.0) and tuple(els(2)[1]["p"][:2]) == pytest.approx((10 * 96 / 25.4, 32.0))
    assert float(ps.paths[2]["stroke_width"]) == 2.0
    # without a viewBox the document's own size is the reference; without either a percentage is nothing
    ps = pm.PathSet.from_svg('<svg width="300" height="150"><rect width="50%" height="100%" fill="red"/></svg>')
    assert tuple(ps.els[2]["p"][:2]) == (150.0, 150.0)
    assert len(pm.PathSet.from_svg('<svg><rect width="50%" height="100%" fill="red"/></svg>').paths) == 0


def test_svg_colour_functions(pm):
    """rgb() / rgba() / hsl() / hsla(): colours against Pillow's conversions, alpha channels folded
    into the item's alpha."""
    ImageColor = pytest.importorskip("PIL.ImageColor")
    rng = np.random.default_rng(3)
    cases = ["rgb(10%, 20%, 30%)", "rgb(12, 200, 255)", "hsl(0,100%,50%)", "hsl(120, 100%, 25%)"]
    cases += ["hsl(%d, %d%%, %d%%)" % (rng.integers(0, 360), rng.integers(0, 101), rng.integers(0, 101)) for _ in range(60)]
    for c in cases:
        ps = pm.PathSet.from_svg('<svg><rect width="5" height="5" fill="%s"/></svg>' % c)
        r, g, b = ImageColor.getrgb(c)[:3]
        assert int(ps.paths[0]["fill_rgba"]) == ((r << 24) | (g << 16) | (b << 8) | 0xFF), c
    ps = pm.PathSet.from_svg('<svg><rect width="5" height="5" fill="rgba(255, 0, 0, 0.5)" stroke="hsla(240,100%,50%,25%)" opacity="0.5"/></svg>')
    assert int(ps.paths[0]["fill_rgba"]) == 0xFF000040 and int(ps.paths[0]["stroke_rgba"]) == 0x0000FF20


def test_svg_gradient_paints_flattened_to_their_mean(pm):
    """PM_SVG_FLAT_GRADIENTS: url(#gradient) becomes the mean of the stops (colour per sRGB component,
    stop-opacity folded into the item's alpha); stops are inherited through href; without the flag,
    and for references that are not gradients, the paint is `none`."""
    svg = '''<svg xmlns="http://www.w3.org/2000/svg" xmlns:xlink="http://www.w3.org/1999/xlink">
      <defs>
        <linearGradient id="g1"><stop offset="0" stop-color="#ff0000"/><stop offset="1" style="stop-color:#0000ff;stop-opacity:0.5"/></linearGradient>
        <radialGradient id="g2" xlink:href="#g1" r="3"/>
        <linearGradient id="g3"><stop offset="0" stop-color="white"/><stop offset=".5" stop-color="rgb(0,0,0)"/><stop offset="1"/></linearGradient>
        <rect id="notagradient" width="1" height="1"/>
      </defs>
      <rect width="5" height="5" fill="url(#g1)"/>
      <rect width="5" height="5" fill="url(#g2)" stroke="url(#nope)" fill-opacity="0.5"/>
      <g fill="url('#g3')"><rect width="5" height="5"/><rect width="5" height="5" fill="#123456"/></g>
      <rect width="5" height="5" fill="url(#notagradient)"/>
    </svg>'''
    assert len(pm.PathSet.from_svg(svg).paths) == 1  # only the plain colour draws
    ps = pm.PathSet.from_svg(svg, flat_gradients=True)
    assert [int(p["flags"]) for p in ps.paths] == [1, 1, 1, 1]
    assert [hex(int(p["fill_rgba"])) for p in ps.paths] == ["0x800080bf", "0x80008060", "0x555555ff", "0x123456ff"]


def test_svg_viewbox_and_fit(pm):
    """The outermost <svg>'s viewBox and size come through the ABI; fit_affine maps the viewBox into a
    viewport like preserveAspectRatio="xMidYMid meet"; units of width / height are converted to px."""
    ps = pm.PathSet.from_svg(open(os.path.join(ROOT, "tests", "data", "shapes.svg")).read())
    assert ps.viewbox == (0.0, 0.0, 400.0, 300.0) and ps.size == (400.0, 300.0)
    aff, s = ps.fit_affine(1200, 600)
    assert s == 2.0 and aff == (2.0, 0.0, 0.0, 2.0, 200.0, 0.0)
    ps = pm.PathSet.from_svg('<svg viewBox="10,20 50 100" width="100%"><svg viewBox="0 0 1 1"><path d="M0 0h1v1z" fill="red"/></svg></svg>')
    assert ps.viewbox == (10.0, 20.0, 50.0, 100.0) and ps.size == (0.0, 0.0)  # the OUTERMOST element counts
    aff, s = ps.fit_affine(200, 200)
    assert s == 2.0 and aff == (2.0, 0.0, 0.0, 2.0, 50.0 - 20.0, -40.0)
    ps = pm.PathSet.from_svg('<svg width="2in" height="36pt" viewBox="0 0 0 5"><rect width="5" height="5" fill="red"/></svg>')
    assert ps.viewbox is None and ps.size == (192.0, 48.0)  # a viewBox without area is ignored
    assert ps.fit_affine(96, 96)[1] == 0.5
    assert pm.PathSet.from_svg('<svg><rect width="5" height="5" fill="red"/></svg>').fit_affine(10, 10) is None
    assert pm.PathSet.tiger().viewbox == (0.0, 0.0, 200.0, 200.0)


@pytest.mark.parametrize("spec_defaults", [False, True])
def test_svg_document_layer_matches_independent_walker(pm, spec_defaults):
    """tests/data/shapes.svg (nested groups, transforms, style, opacity, both fill rules, every basic
    shape) through the C++ front-end and through the independent Python walker: the same drawn
    elements in the same order with the same paints, rule bits and widths; straight geometry
    bit for bit after the transform, curved geometry (arcs -> cubics are product-defined) on the
    transformed outline it has to follow."""
    svg = open(os.path.join(ROOT, "tests", "data", "shapes.svg")).read()
    ps = pm.PathSet.from_svg(svg, spec_defaults=spec_defaults)
    want = _svg_walk(svg, spec_defaults)
    assert len(ps.paths) == len(want) == 21
    for p, w in zip(ps.paths, want):
        assert int(p["flags"]) == w["flags"], w
        if w["flags"] & 1:
            assert int(p["fill_rgba"]) == w["fill"], (hex(int(p["fill_rgba"])), hex(w["fill"]))
        if w["flags"] & 2:
            assert int(p["stroke_rgba"]) == w["stroke"] and float(p["stroke_width"]) == pytest.approx(w["width"], rel=1e-6)
        els = ps.els[int(p["el_begin"]) : int(p["el_end"])]
        M, g = w["ctm"], w["geom"]
        if g[0] == "poly":
            pts = [_apply(M, q) for q in g[1]]
            assert els["tag"].tolist() == [0] + [1] * (len(pts) - 1) + ([4] if g[2] else [])
            assert [tuple(r) for r in els["p"][: len(pts), :2].tolist()] == pts
        elif g[0] == "rect" and g[5] == 0:
            x, y, ww, hh = g[1:5]
            pts = [_apply(M, q) for q in ((x, y), (x + ww, y), (x + ww, y + hh), (x, y + hh))]
            assert els["tag"].tolist() == [0, 1, 1, 1, 4] and [tuple(r) for r in els["p"][:4, :2].tolist()] == pts
        elif g[0] == "path":
            ref = _svg_path_elements(g[1])
            assert [int(t) for t in els["tag"]] == [e[0] for e in ref]
            for row, e in zip(els, ref):
                for k in range(len(e[1]) // 2):
                    q = _apply(M, (e[1][2 * k], e[1][2 * k + 1]))
                    assert row["p"][2 * k] == pytest.approx(q[0], abs=1e-9) and row["p"][2 * k + 1] == pytest.approx(q[1], abs=1e-9)
        else:  # ellipse / rounded rect: sample the emitted curves; every sample lies on the transformed outline
            a, b, c, d, e_, f_ = M
            det = a * d - b * c
            inv = [d / det, -b / det, -c / det, a / det, (c * f_ - d * e_) / det, (b * e_ - a * f_) / det]
            cur = None
            n_curves = 0
            for row in els:
                tag = int(row["tag"])
                if tag == 0:
                    cur = (row["p"][0], row["p"][1])
                    continue
                if tag == 4:
                    continue
                for t in (0.0, 0.25, 0.5, 0.75, 1.0):
                    q = _apply(inv, _bez(cur, row, t))  # back in the shape's own user space
                    if g[0] == "ellipse":
                        r = math.hypot((q[0] - g[1]) / g[3], (q[1] - g[2]) / g[4])
                        assert abs(r - 1.0) < 3e-4
                    else:
                        x, y, ww, hh, rx, ry = g[1:]
                        dx = max(x + rx - q[0], 0.0, q[0] - (x + ww - rx)) / rx
                        dy = max(y + ry - q[1], 0.0, q[1] - (y + hh - ry)) / ry
                        inside_core = dx == 0.0 or dy == 0.0
                        if inside_core:
                            assert min(abs(q[0] - x), abs(q[0] - x - ww), abs(q[1] - y), abs(q[1] - y - hh)) < 1e-9
                        else:
                            assert abs(math.hypot(dx, dy) - 1.0) < 3e-4
                n_curves += tag == 3
                k = {1: 0, 2: 2, 3: 4}[tag]
                cur = (row["p"][k], row["p"][k + 1])
            assert n_curves == 4
    # the reference's reading is unchanged on the Tiger by all of this: see test_tiger_paths_are_pinned


def test_tiger_paths_are_pinned(pm):
    import hashlib
    import json

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["tiger_paths"]
    ps = pm.PathSet.tiger()
    assert (len(ps.paths), len(ps.els)) == (g["n_paths"], g["n_els"])
    assert hashlib.sha256(ps.paths.tobytes()).hexdigest() == g["paths_sha256"]
    assert hashlib.sha256(ps.els.tobytes()).hexdigest() == g["els_sha256"]


def test_workload_generators_are_deterministic(pm):
    a = pm.workloads.config4_blobs(50, 1024)
    b = pm.workloads.config4_blobs(50, 1024)
    assert np.array_equal(a.paths.els, b.paths.els) and np.array_equal(a.paths.paths, b.paths.paths)
    assert len(a.paths.els) == 300 and (a.paths.paths["flags"] == 1).all()
    assert ((a.paths.paths["fill_rgba"] & 0xFF) >= 0x40).all()
    rng = pm.workloads.SplitMix64(0x5EED0004)
    assert rng.next() == pm.workloads.SplitMix64(0x5EED0004).next()
    g = pm.workloads.config5_tiger_grid(2048, 2, 8.0)
    assert len(g.paths.paths) == 4 * 138
    assert pm.workloads.band_rows(135, 8, 0) == (0, 17) and pm.workloads.band_rows(135, 8, 7) == (119, 135)
    assert sum(b - a for a, b in (pm.workloads.band_rows(135, 8, r) for r in range(8))) == 135


def test_png_writer_round_trip(tmp_path):
    from piet_metal_amd import cli

    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    cli.write_png(p, img)
    assert np.array_equal(cli.read_png_rgba(p), img)
    with pytest.raises(ValueError):
        cli.write_png(p, img[:, :, :3])


@pytest.mark.parametrize("seed", range(6))
def test_svg_path_grammar_random(pm, seed):
    """Random path data in every spelling the grammar allows (absolute / relative, implicit
    repeats, implicit lineto after moveto, smooth cubics, H/V, numbers glued by signs and dots,
    exponents) through the C++ front-end and the independent Python parser: bit-exact f64."""
    rng = np.random.default_rng(1000 + seed)

    def fmt(v):
        style = rng.integers(0, 4)
        if style == 0:
            return repr(float(np.round(v, 3)))
        if style == 1:
            return f"{v:.2e}"
        if style == 2 and abs(v) < 1:
            s = f"{abs(v):.3f}"[1:]  # ".123"
            return ("-" if v < 0 else "") + s
        return f"{v:.4f}".rstrip("0").rstrip(".") or "0"

    def nums(k):
        vals = [float(rng.uniform(-200, 200)) if rng.random() < 0.8 else float(rng.uniform(-1, 1)) for _ in range(k)]
        out = ""
        for v in vals:
            s = fmt(v)
            sep = rng.choice([" ", ",", " , ", ""]) if out else ""
            if sep == "" and out and not (s[0] == "-" or (s[0] == "." and "." in out.split()[-1].split(",")[-1].lstrip("-") and "e" not in out.split()[-1].lower())):
                sep = " "
            out += sep + s
        return out

    tags = {"M": 0, "L": 1, "C": 3, "Z": 4}
    docs = []
    for _ in range(40):
        d = "M" + nums(2)
        if rng.random() < 0.4:
            d += " " + nums(2 * int(rng.integers(1, 3)))  # implicit lineto after moveto
        for _ in range(int(rng.integers(1, 9))):
            c = rng.choice(list("LlHhVvCcSsZzMm"))
            if c in "Zz":
                d += c
            else:
                per = {"L": 2, "H": 1, "V": 1, "C": 6, "S": 4, "M": 2}[c.upper()]
                rep_n = int(rng.integers(1, 3))
                d += rng.choice(["", " "]) + c + nums(per * rep_n)
        docs.append(d)
    svg = "<svg>" + "".join(f'<path d="{d}" fill="#123456"/>' for d in docs) + "</svg>"
    ps = pm.PathSet.from_svg(svg)
    assert len(ps.paths) == len(docs)
    for d, path in zip(docs, ps.paths):
        want = py_parse_path(d)
        got = ps.els[path["el_begin"] : path["el_end"]]
        assert len(got) == len(want), d
        for g, (t, coords) in zip(got, want):
            assert g["tag"] == tags[t], d
            assert g["p"][: len(coords)].tolist() == coords, d


def test_profile_stamp_digest_ignores_comments_but_not_code(tmp_path):
    """profiles/hbm_traffic.json is stamped with a digest of the kernel sources (tools/make_traffic.py) and
    bench.py calls the profile stale when the digest of the sources it runs differs: rewording a comment
    must not do that, changing a token must.  The committed stamp matches the committed sources."""
    import json
    import shutil
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_traffic as mt

    src = os.path.join(ROOT, "piet_metal_amd", "csrc")
    dst = tmp_path / "piet_metal_amd" / "csrc"
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns("*.o", "*.so"))
    base = mt.kernel_sources_sha16(str(tmp_path))
    assert base == mt.kernel_sources_sha16()
    f = dst / "pm_bin_rows.h"
    text = f.read_text()
    f.write_text("// a new remark\n" + text.replace("// ", "//   ", 5) + "\n/* and\n   another */\n")
    assert mt.kernel_sources_sha16(str(tmp_path)) == base
    f.write_text(text.replace("kSupCPL = 2", "kSupCPL = 3", 1))
    assert mt.kernel_sources_sha16(str(tmp_path)) != base
    stamp = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    assert stamp["_kernel_sources_sha16"] == mt.kernel_sources_sha16(), "profiles/hbm_traffic.json is older than the kernel sources: re-run tools/prof_round.sh"


def test_the_frame_path_kernels_do_not_spill(tmp_path):
    """pm_bin_kernel<false, 4> -- the frame path's binning kernel -- must fit its 96 VGPRs without scratch: a spilled value is reloaded
    with `s_waitcnt vmcnt(0)`, which also waits for the loads the vote loop keeps in flight for its next round and for its stores
    (round 6: three spills crept in with new code and cost every strip row's votes their software pipelining -- unnoticed for
    hours, because nothing fails).  The listing of the very flags the library is built with says so."""
    import re
    import shutil
    import subprocess

    hipcc = "/opt/rocm/bin/hipcc"
    if not shutil.which(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "piet_metal_amd", "csrc")
    mk = open(os.path.join(src, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS := (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").replace("$(HERE)", src + "/").replace("$(EXTRA)", "").split()
    # (kernel name fragment -> VGPR budget: five workgroups per CU, six for the one-wave tile kernel)
    want = {"pm_bin.hip": {"pm_bin_kernelILb0ELi4E": 96, "pm_bin_kernelILb1ELi4E": 96},
            # ... and the tile kernels of the frame path (general, one-wave, profiled): a reload at the top of a tile waits for the previous
            # tile's pixel stores -- the frame path's general kernel carried three such spills for two rounds
            "pm_fine.hip": {"pm_fine_kernelILb1ELb0ELb0ELb0E": 96, "pm_fine_kernelILb1ELb0ELb0ELb1E": 80, "pm_fine_kernelILb1ELb1ELb0ELb0E": 96}}
    for unit, kernels in want.items():
        out = str(tmp_path / (unit + ".s"))
        subprocess.check_call([hipcc, *flags, "-S", "--cuda-device-only", os.path.join(src, unit), "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read()
        found = 0
        for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.M | re.S):
            name, body = m.group(1), m.group(2)
            budget = next((v for k, v in kernels.items() if k in name), None)
            if budget is None:
                continue
            found += 1
            scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
            vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
            assert scratch == 0 and vgpr <= budget, (name, scratch, vgpr)
        assert found == len(kernels), unit
