"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, must be
byte-identical to the oracle on the same encoded scene -- RGBA8 pixels, per-tile
command lists, and the device-flattened scene bytes.  Full-size configurations are
covered through committed golden hashes and size-independent properties
(band-vs-full equality, idempotence, clip-invariance)."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from test_host_cpu import encode_ops, extend_ops, inline_ops, random_ops

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


def gpu_render(renderer, scene, w, h):
    renderer.resize(w, h)
    renderer.set_scene_bytes(scene)
    renderer.render()
    renderer.sync()
    return renderer.read_pixels()


def assert_ptcl_equal(renderer, pmo, scene, w, h, maxc=1024):
    P = pmo.Ptcl(scene, w, h)
    counts, solid, cmds = renderer.capture_ptcl(maxc)
    for ty in range(P.tiles_y):
        for tx in range(P.tiles_x):
            oc = P.cmds(tx, ty)
            assert counts[ty, tx] == len(oc), (tx, ty)
            assert solid[ty, tx] == P.solid(tx, ty), (tx, ty)
            assert np.array_equal(cmds[ty, tx, : len(oc)], oc), (tx, ty)
    P.close()


def test_extension_is_loaded_and_device_is_gfx950(pm, renderer):
    assert os.path.exists(pm._lib.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libpiet_metal_amd.so" in maps
    import torch

    arch = torch.cuda.get_device_properties(0).gcnArchName
    assert arch.split(":")[0] == "gfx950", arch


@pytest.mark.parametrize("name,w,h", [("path_test", 512, 832), ("cardioid", 2048, 1536), ("cardioid", 1999, 1501), ("cardioid", 300, 200)])
def test_reference_scenes_pixels_and_lists(pm, pmo, renderer, name, w, h):
    scene = pmo.scene_path_test() if name == "path_test" else pmo.scene_cardioid()
    got = gpu_render(renderer, scene, w, h)
    assert np.array_equal(got, pmo.render(scene, w, h))
    assert_ptcl_equal(renderer, pmo, scene, w, h)


def test_host_encoded_scene_through_pinned_buffer(pm, pmo, renderer):
    # the drop-in flow of PietRenderer.m:203-205: encode into the renderer's own buffer
    renderer.resize(640, 480)
    buf = renderer.scene_buffer()
    buf[:8192] = 0
    n = pm.scene_cardioid(buf)
    renderer.upload_scene(n)
    renderer.render()
    assert np.array_equal(renderer.read_pixels(), pmo.render(pmo.scene_cardioid(), 640, 480))
    bgra = renderer.read_pixels(bgra=True)
    assert np.array_equal(bgra[..., [2, 1, 0, 3]], renderer.read_pixels())


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_scenes(pm, pmo, renderer, seed):
    # circles, lines, fills (incl. vertices on tile boundaries / axis-aligned edges: quirks
    # Q1-Q3), polylines (quirk Q4), opaque and translucent colours
    scene = encode_ops(pm, random_ops(seed, 120, extent=900.0))
    w, h = 928, 912
    got = gpu_render(renderer, scene, w, h)
    want = pmo.render(scene, w, h)
    assert np.array_equal(got, want)
    assert_ptcl_equal(renderer, pmo, scene, w, h)


def test_many_items_multiple_batches_and_long_lists(pm, pmo, renderer):
    # > 256 items (several binning batches) stacked on few tiles: lists far beyond the
    # reference's 170-command tile buffer (quirk Q5)
    rng = np.random.default_rng(7)
    ops = []
    for i in range(700):
        c = rng.uniform(40, 200, 2)
        pts = c + rng.uniform(-60, 60, (int(rng.integers(3, 9)), 2))
        rgba = (int(rng.integers(0, 1 << 24)) << 8) | (0xFF if i % 97 == 0 else int(rng.integers(0x20, 0xFF)))
        ops.append(("fill", pts, rgba) if i % 3 else ("poly", pts, rgba, float(rng.uniform(0.5, 6))))
    scene = encode_ops(pm, ops, cap=1 << 22)
    got = gpu_render(renderer, scene, 256, 256)
    P = pmo.Ptcl(scene, 256, 256)
    assert P.total_cmds()[1] > 300
    assert np.array_equal(got, P.render())
    assert_ptcl_equal(renderer, pmo, scene, 256, 256, maxc=4096)


def test_one_strip_row_with_thousands_of_surviving_chunks(pm, pmo, renderer):
    """Three fills and a polyline of 7 000 points each zigzag inside a single strip row: every chunk of 8
    segments survives binning's box test there -- 3 500 chunks, far more than the LDS survivor list
    (1 024) holds: the spill path of pm_bin_kernel, and dozens of vote / scatter rounds per wave.
    (7 000 points keep an item's winding inside the packed per-tile counter)"""
    n = 7000
    xs = 3.0 + 245.0 * (np.arange(n) % 2) + 0.01 * np.arange(n)
    ys = 99.5 + 12.0 * ((np.arange(n) // 2) % 2) + 0.0003 * np.arange(n)
    pts = np.stack([xs, ys], axis=1)
    ops = [("fill", pts, 0x3060A0C0), ("fill", pts + np.array([0.5, -1.25]), 0xA0303080), ("circle", 120.0, 104.0, 40.0),
           ("poly", pts[::-1] + np.array([1.5, 2.25]), 0x20C040FF, 1.5), ("fill", pts * np.array([1.0, 1.01]), 0x10F01060)]
    scene = encode_ops(pm, ops, cap=1 << 20)
    w, h = 340, 160
    got = gpu_render(renderer, scene, w, h)
    assert np.array_equal(got, pmo.render(scene, w, h))
    assert_ptcl_equal(renderer, pmo, scene, w, h, maxc=1 << 16)


def test_list_longer_than_the_lds_command_buffer(pm, pmo, renderer):
    # 900 translucent polylines through one tile: > 2000 commands, so the per-tile kernel
    # must flush its 768-slot LDS list mid-stream (pixel state carried in registers); an
    # opaque fill in the middle exercises the "restart at tileBegin" rule after a flush
    rng = np.random.default_rng(21)
    ops = []
    for i in range(900):
        a = rng.uniform(67, 77, 2)  # everything inside tile (4, 4)
        pts = np.stack([a, a + rng.uniform(-3, 3, 2), a + rng.uniform(-3, 3, 2)])
        ops.append(("poly", pts, (int(rng.integers(0, 1 << 24)) << 8) | int(rng.integers(0x10, 0x60)), float(rng.uniform(0.3, 3))))
        if i == 600:
            ops.append(("fill", np.array([(70.5, 40.0), (160.0, 75.5), (70.5, 120.0), (20.0, 75.5)]), 0x336699FF))
    for variant in (ops, ops[:601] + ops[602:]):
        scene = encode_ops(pm, variant, cap=1 << 22)
        got = gpu_render(renderer, scene, 192, 160)
        P = pmo.Ptcl(scene, 192, 160)
        assert np.array_equal(got, P.render())
        assert_ptcl_equal(renderer, pmo, scene, 192, 160, maxc=4096)
    assert P.total_cmds()[1] > 1600  # the variant without the opaque fill


def test_workgroup_tile_fills_shared_fragments_and_their_fallback(pm, pmo, renderer):
    """A tile the workgroup renders together has the fragments of a chunk's Fill commands made by the four waves before the items are
    handed out, a quarter of the Fills per wave into a region of 64 fragments each (PrepareFillsShared).  Slanted slivers (a few rows
    per segment) fit; near-vertical ones cross all 16 rows of the tile -- 16 fragments per Fill, a wave's share overflows its region and
    the workgroup falls back to fragments per item; a mix does both from chunk to chunk.  Translucent, so no list restarts."""
    rng = np.random.default_rng(29)
    for kind in ("slanted", "vertical", "mixed"):
        ops = []
        for i in range(70):
            x = 66.0 + float(rng.uniform(0, 10))
            vertical = kind == "vertical" or (kind == "mixed" and (i // 12) % 2 == 1)
            if vertical:  # a sliver through the whole height of tile (4, 4) (and its neighbours above and below)
                w = float(rng.uniform(0.4, 2.5))
                lean = float(rng.uniform(-0.8, 0.8))
                pts = np.array([(x, 60.0), (x + w, 60.0), (x + w + lean, 84.0), (x + lean, 84.0)])
            else:
                y = 65.0 + float(rng.uniform(0, 10))
                pts = np.array([(x, y), (x + float(rng.uniform(1, 5)), y + float(rng.uniform(-1, 1))), (x + float(rng.uniform(0, 3)), y + float(rng.uniform(1, 4)))])
            ops.append(("fill", pts, (int(rng.integers(0, 1 << 24)) << 8) | int(rng.integers(0x20, 0x90))))
        scene = encode_ops(pm, ops, cap=1 << 20)
        got = gpu_render(renderer, scene, 192, 160)
        P = pmo.Ptcl(scene, 192, 160)
        assert np.array_equal(got, P.render()), kind
        assert_ptcl_equal(renderer, pmo, scene, 192, 160, maxc=4096)
        assert P.total_cmds()[1] > 190, (kind, P.total_cmds())  # longer than the single-wave limit: a workgroup tile, three chunks and more


def test_dense_fill_pairs_fill_the_fragment_region(pm, pmo, renderer):
    """Pass 1 of the row-sparse Fill evaluation takes nine and more Fill commands as (command, row) pairs, 64 per step (FillPairs).  Tall
    translucent triangles through one tile: every long edge is live in 10-12 rows, so the second step finds the wave's 64 fragment slots
    full in the middle of a command -- that command is taken back, the pass stops in front of it and the command loop comes back for the
    rest.  Lists of 24-36 stream elements: a single wave's tiles (and with more triangles a workgroup's items of nine Fills and more)."""
    rng = np.random.default_rng(31)
    for n_tri, rows in ((8, 11.0), (9, 12.5), (7, 15.5), (14, 11.0)):
        ops = []
        for i in range(n_tri):
            x = 65.0 + float(rng.uniform(0, 11))
            y = 64.05 + float(rng.uniform(0, 15.9 - rows))
            pts = np.array([(x, y), (x + float(rng.uniform(0.6, 3.0)), y + float(rng.uniform(0, 0.8))), (x + float(rng.uniform(-2.0, 2.0)), y + rows)])
            ops.append(("fill", pts, (int(rng.integers(0, 1 << 24)) << 8) | int(rng.integers(0x20, 0x90))))
        scene = encode_ops(pm, ops, cap=1 << 20)
        got = gpu_render(renderer, scene, 192, 160)
        P = pmo.Ptcl(scene, 192, 160)
        assert np.array_equal(got, P.render()), (n_tri, rows)
        assert_ptcl_equal(renderer, pmo, scene, 192, 160, maxc=4096)
        assert P.total_cmds()[1] >= 3 * n_tri, (n_tri, P.total_cmds())


def test_one_wave_kernel_lists_around_its_lds_chunks(pm, pmo, monkeypatch):
    """The tile kernel's one-wave-per-tile instantiation keeps the first TWO chunks of 64 commands of a wave's list in LDS and reads only
    what lies beyond back from the tile's list in HBM: lists that end just below, at and beyond both boundaries (three commands per
    polyline: 63, 66, 126, 129, 132 ... 900), with and without an opaque fill in the middle (the list restarts there).  PM_DENSE_FACTOR
    makes any frame with a long list a dense one, so the frames behind a scene's first run that kernel."""
    monkeypatch.setenv("PM_DENSE_FACTOR", "100000")
    r = pm.Renderer(0)
    try:
        r.resize(192, 160)
        rng = np.random.default_rng(23)
        for n in (21, 22, 42, 43, 44, 64, 90, 300):
            for with_fill in (False, True):
                ops = []
                for i in range(n):
                    a = rng.uniform(67, 77, 2)  # everything inside tile (4, 4)
                    pts = np.stack([a, a + rng.uniform(-3, 3, 2), a + rng.uniform(-3, 3, 2)])
                    ops.append(("poly", pts, (int(rng.integers(0, 1 << 24)) << 8) | int(rng.integers(0x10, 0x60)), float(rng.uniform(0.3, 3))))
                    if with_fill and i == n // 3:
                        ops.append(("fill", np.array([(70.5, 40.0), (160.0, 75.5), (70.5, 120.0), (20.0, 75.5)]), 0x336699FF))
                scene = encode_ops(pm, ops, cap=1 << 22)
                want = pmo.render(scene, 192, 160)
                r.set_scene_bytes(scene)
                before = r.dense_kernel_frames()
                for k in range(3):
                    r.render()
                    assert np.array_equal(r.read_pixels(), want), (n, with_fill, k)
                assert r.dense_kernel_frames() == before + 2, (n, with_fill)
    finally:
        r.close()


def test_empty_scene_and_tiny_viewports(pm, pmo, renderer):
    empty = np.frombuffer(struct.pack("<II", 0, 8), np.uint8)
    assert (gpu_render(renderer, empty, 40, 24) == 255).all()
    scene = pmo.scene_cardioid()
    for w, h in ((1, 1), (15, 17), (257, 33)):
        assert np.array_equal(gpu_render(renderer, scene, w, h), pmo.render(scene, w, h))


def test_malformed_scene_is_rejected(pm, renderer):
    renderer.resize(64, 64)
    bad = np.frombuffer(struct.pack("<II", 1000, 8), np.uint8)  # items run past the buffer
    with pytest.raises(pm.PietMetalError) as ei:
        renderer.set_scene_bytes(bad)
    assert ei.value.status == pm._lib.PM_ERR_SCENE


def _flatten_case(pm, pmo, renderer, wl):
    renderer.resize(wl.width, wl.height)
    nbytes, nitems = renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    dev = renderer.download_scene()
    ref, ref_items = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    assert nitems == ref_items and nbytes == ref.size
    assert np.array_equal(dev, ref)  # flatten.rs on device == CPU encoder, byte for byte
    return ref


def test_device_flatten_tiger_reference_scale_golden(pm, pmo, renderer, golden):
    wl = pm.workloads.tiger_reference()
    scene = _flatten_case(pm, pmo, renderer, wl)
    assert sha(scene) == golden["tiger_x8"]["scene_sha256"]
    renderer.render()
    got = renderer.read_pixels()
    assert sha(got) == golden["tiger_x8"]["rgba_sha256"]
    crop = np.load(os.path.join(ROOT, "tests", "golden", "tiger_x8_crop_704_496_64x64.npy"))
    assert np.array_equal(got[496:560, 704:768], crop)
    assert_ptcl_equal(renderer, pmo, scene, wl.width, wl.height)


def test_init_test_scene_drop_in(pm, pmo, golden):
    buf = np.zeros(16 << 20, np.uint8)  # the reference's 16 MiB shared buffer (PietRenderer.m:53)
    pm.init_test_scene(buf)
    n = golden["tiger_x8"]["scene_bytes"]
    assert sha(buf[:n]) == golden["tiger_x8"]["scene_sha256"]
    assert not buf[n:].any()


@pytest.mark.parametrize("cfg", ["config1", "config1_rot", "config2", "config3"])
def test_baseline_configs(pm, pmo, renderer, golden, cfg):
    wl = {
        "config1": lambda: pm.workloads.config1_rect(),
        "config1_rot": lambda: pm.workloads.config1_rect(True),
        "config2": lambda: pm.workloads.tiger(1920, 1080, fills_only=True),
        "config3": lambda: pm.workloads.tiger(3840, 2160),
    }[cfg]()
    scene = _flatten_case(pm, pmo, renderer, wl)
    assert sha(scene) == golden[wl.name]["scene_sha256"]
    renderer.render()
    got = renderer.read_pixels()
    assert sha(got) == golden[wl.name]["rgba_sha256"]          # committed pin
    assert np.array_equal(got, pmo.render(scene, wl.width, wl.height))  # live oracle
    # the GPU's per-tile command lists against the committed pin (24-byte reference layout)
    counts, _solid, cmds = renderer.capture_ptcl(golden[wl.name]["max_cmds_per_tile"])
    hsh = hashlib.sha256()
    for ty in range(counts.shape[0]):
        for tx in range(counts.shape[1]):
            n = int(counts[ty, tx])
            hsh.update(np.uint32(n).tobytes())
            hsh.update(np.ascontiguousarray(cmds[ty, tx, :n]).tobytes())
    assert hsh.hexdigest() == golden[wl.name]["ptcl_sha256"]
    st = renderer.stats()
    assert st["overflow"] == 0 and st["arena_used_dwords"] <= st["arena_cap_dwords"]


def test_4k_frames_behind_one_another_bin_with_a_wave_per_strip_row(pm, golden, monkeypatch):
    """A frame submitted while the previous one is still running bins with a wave per strip row when it has enough strip
    rows to fill the chip that way (>= 4 per CU and no chains: the 4K Tiger's 1 109), a lone frame with a workgroup
    per row: eight frames back to back, then one alone, every one of them the committed pin -- on the context's own
    streams and slots, and with the two ways pinned (PM_BIN_WAVES_INFLIGHT=4: never a wave per row)."""
    wl = pm.workloads.tiger(3840, 2160)
    for inflight in (None, "4"):
        if inflight is not None:
            monkeypatch.setenv("PM_BIN_WAVES_INFLIGHT", inflight)
        r = pm.Renderer(0)
        try:
            r.resize(wl.width, wl.height)
            r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
            for burst in (8, 1, 3):
                for _ in range(burst):
                    r.render()
                assert sha(r.read_pixels()) == golden[wl.name]["rgba_sha256"], (inflight, burst)
        finally:
            r.close()


def test_config4_blobs_reduced_vs_oracle_and_full_properties(pm, pmo, renderer):
    # oracle-sized: 600 blobs at 1024^2, byte-exact
    wl = pm.workloads.config4_blobs(600, 1024)
    scene = _flatten_case(pm, pmo, renderer, wl)
    renderer.render()
    assert np.array_equal(renderer.read_pixels(), pmo.render(scene, wl.width, wl.height))
    # full size (10k blobs, 4096^2): idempotence + band-vs-full + oracle on one band of rows
    wl = pm.workloads.config4_blobs()
    renderer.resize(wl.width, wl.height)
    renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    scene = renderer.download_scene()
    renderer.render()
    full = renderer.read_pixels()
    renderer.render()
    assert np.array_equal(full, renderer.read_pixels())
    renderer.set_band(100, 110)
    renderer.render()
    band = renderer.read_pixels()
    assert np.array_equal(band, full[1600:1760])
    P = pmo.Ptcl(scene, wl.width, wl.height)
    assert np.array_equal(band[:32], P.render_rows(100, 102))
    P.close()


def test_extension_scenes_against_committed_goldens(pm, pmo, renderer, golden):
    """The pins of the extension scenes (tests/golden/make_golden.py --ext): the GPU's pixels and
    per-tile command lists against the COMMITTED hashes, not only against today's oracle."""
    from test_oracle_cpu import _extension_scenes

    for name, (scene, w, h) in _extension_scenes(pm, pmo).items():
        got = gpu_render(renderer, scene, w, h)
        assert sha(got) == golden[name]["rgba_sha256"], name
        counts, _solid, cmds = renderer.capture_ptcl(golden[name]["max_cmds_per_tile"])
        hsh = hashlib.sha256()
        for ty in range(counts.shape[0]):
            for tx in range(counts.shape[1]):
                n = int(counts[ty, tx])
                hsh.update(np.uint32(n).tobytes())
                hsh.update(np.ascontiguousarray(cmds[ty, tx, :n]).tobytes())
        assert hsh.hexdigest() == golden[name]["ptcl_sha256"], name


@pytest.mark.parametrize("key", ["held1", "held2", "held3"])
def test_heldout_workloads_full_size_goldens(pm, renderer, golden, key):
    """The three held-out workloads (piet_metal_amd/workloads.py: Tiger 2560x1440 with strokes, 2 k blobs at 2048^2, 20 k
    glyph-like paths at 4K -- scenes no threshold of the frame path was chosen on) against the oracle's full-size pins
    (tests/golden/make_golden.py --held): device-flattened scene bytes, pixels, every tile's list; alone and four in flight."""
    wl = pm.workloads.heldout_workloads()[key]
    g = golden[wl.name]
    renderer.resize(wl.width, wl.height)
    nbytes, n_items = renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    assert n_items == g["n_items"] and nbytes == g["scene_bytes"]
    assert sha(renderer.download_scene()) == g["scene_sha256"]
    renderer.render()
    assert sha(renderer.read_pixels()) == g["rgba_sha256"]
    counts, _solid, cmds = renderer.capture_ptcl(g["max_cmds_per_tile"])
    hsh = hashlib.sha256()
    for ty in range(counts.shape[0]):
        for tx in range(counts.shape[1]):
            n = int(counts[ty, tx])
            hsh.update(np.uint32(n).tobytes())
            hsh.update(np.ascontiguousarray(cmds[ty, tx, :n]).tobytes())
    assert hsh.hexdigest() == g["ptcl_sha256"]
    before = renderer.binning_info()
    for _ in range(8):
        renderer.render()
    assert sha(renderer.read_pixels()) == g["rgba_sha256"]
    if key == "held3":
        # 2 025 light strip rows: alone a workgroup per row, chained over the plan's 1 280 workgroups; behind running frames a
        # wave for EVERY row (the chains are not walked) -- the policy check found that 14 % faster per frame
        after = renderer.binning_info()
        assert after["no_chains"] - before["no_chains"] >= 4
        assert after["wave_per_row"] - before["wave_per_row"] == after["inflight_only"] - before["inflight_only"]


@pytest.mark.parametrize("cfg", ["config4", "config5"])
def test_baseline_configs_4_and_5_full_size_goldens(pm, pmo, renderer, golden, cfg):
    """BASELINE configs 4 (10 k blobs, 4096^2) and 5 (25 Tigers, 8192^2) at FULL size against the
    committed pins the oracle produced in the build container (tests/golden/make_golden.py --big):
    device-flattened scene bytes, every pixel, every tile's command list and solid colour."""
    wl = pm.workloads.config4_blobs() if cfg == "config4" else pm.workloads.config5_tiger_grid()
    g = golden[wl.name]
    renderer.resize(wl.width, wl.height)
    nbytes, nitems = renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    assert (nbytes, nitems) == (g["scene_bytes"], g["n_items"])
    assert sha(renderer.download_scene()) == g["scene_sha256"]
    renderer.render()
    img = renderer.read_pixels()
    assert int(img.astype(np.uint64).sum()) == g["rgba_sum"]
    assert sha(img) == g["rgba_sha256"]
    del img
    # command lists band by band (the capture buffer of the whole frame would be > 1 GB)
    tiles_y = (wl.height + 15) // 16
    hsh, solid_all, total, mx = hashlib.sha256(), [], 0, 0
    for r0 in range(0, tiles_y, 32):
        renderer.set_band(r0, min(r0 + 32, tiles_y))
        renderer.render()
        counts, solid, cmds = renderer.capture_ptcl(g["max_cmds_per_tile"])
        solid_all.append(solid)
        total += int(counts.sum())
        mx = max(mx, int(counts.max()))
        flat_counts = counts.reshape(-1)
        flat_cmds = cmds.reshape(-1, cmds.shape[2], 6)
        for t in range(flat_counts.size):
            n = int(flat_counts[t])
            hsh.update(np.uint32(n).tobytes())
            hsh.update(flat_cmds[t, :n].tobytes())
    assert (total, mx) == (g["total_cmds"], g["max_cmds_per_tile"])
    assert sha(np.concatenate(solid_all, axis=0)) == g["solid_sha256"]
    assert hsh.hexdigest() == g["ptcl_sha256"]


def test_bands_reassemble_to_full_frame(pm, pmo, renderer):
    # the multi-GPU sharding on one GPU: every band of an 8-way split equals the
    # corresponding rows of the full frame
    wl = pm.workloads.tiger(1920, 1080)
    renderer.resize(wl.width, wl.height)
    renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    renderer.render()
    full = renderer.read_pixels()
    from piet_metal_amd import dist as pmd

    for world in (2, 8):
        for rank, (r0, r1, rows) in enumerate(pmd.band_layout(wl.height, world)):
            renderer.set_band(r0, r1)
            renderer.render()
            band = renderer.read_pixels()
            assert band.shape[0] == rows
            assert np.array_equal(band, full[r0 * 16 : r0 * 16 + rows]), (world, rank)


def test_config5_tiger_grid_reduced(pm, pmo, renderer):
    # 2x2 Tigers at scale 4 in 2048^2 (oracle-sized stand-in for the 8192^2 grid), full + bands
    wl = pm.workloads.config5_tiger_grid(2048, 2, 4.0)
    scene = _flatten_case(pm, pmo, renderer, wl)
    renderer.render()
    full = renderer.read_pixels()
    assert np.array_equal(full, pmo.render(scene, wl.width, wl.height))
    renderer.set_band(64, 128)
    renderer.render()
    assert np.array_equal(renderer.read_pixels(), full[1024:2048])


def test_render_to_torch_tensor_on_torch_stream(pm, pmo, renderer):
    import torch

    wl = pm.workloads.tiger(480, 270)
    renderer.resize(wl.width, wl.height)
    renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    scene = renderer.download_scene()
    t = torch.zeros((272, 480, 4), dtype=torch.uint8, device="cuda:0")
    renderer.render_to(t, torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert np.array_equal(t[:270].cpu().numpy(), pmo.render(scene, 480, 270))


def test_bgra8_target_is_the_rgba_frame_with_r_and_b_exchanged(pm, pmo, renderer):
    """The reference's renderKernel writes a BGRA8Unorm drawable (PietRenderer.m:29): with
    pm_set_target_format(BGRA8) the kernels store that byte order themselves -- interpreted tiles,
    tiles resolved to one colour by binning or by the list builder -- into the context's framebuffer
    and into a caller-owned one."""
    import torch

    wl = pm.workloads.tiger(640, 360)
    renderer.resize(wl.width, wl.height)
    renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    want = pmo.render(renderer.download_scene(), wl.width, wl.height)
    try:
        renderer.set_target_format(bgra=True)
        renderer.render()
        assert np.array_equal(renderer.read_pixels(bgra=True), want[:, :, [2, 1, 0, 3]])
        assert np.array_equal(renderer.read_pixels(), want)  # swizzled back on the host
        if torch.cuda.is_available():  # (always on the GPU box; the CPU emulation of tests/emu has no torch device)
            t = torch.zeros((368, wl.width, 4), dtype=torch.uint8, device="cuda:0")
            renderer.render_to(t, None)
            renderer.sync()
            assert np.array_equal(t[: wl.height].cpu().numpy(), want[:, :, [2, 1, 0, 3]])
    finally:
        renderer.set_target_format(bgra=False)
    renderer.render()
    assert np.array_equal(renderer.read_pixels(), want)
    assert np.array_equal(renderer.read_pixels(bgra=True), want[:, :, [2, 1, 0, 3]])


def test_back_to_back_frames_and_timing_api(pm, pmo, renderer):
    scene = pmo.scene_cardioid()
    renderer.resize(1024, 768)
    renderer.set_scene_bytes(scene)
    tm = renderer.time_frames(10)
    assert tm["total_ms"] > 0 and tm["bin_ms"] > 0 and tm["fine_ms"] > 0
    assert np.array_equal(renderer.read_pixels(), pmo.render(scene, 1024, 768))


def test_pipelined_frames_every_slot_and_scene_switch(pm, pmo, renderer):
    """Frames overlap across two streams and three frame slots: whichever slot the last
    frame landed in must hold the right picture, also right after the scene changed
    under frames still in flight."""
    a, b = pmo.scene_cardioid(), pmo.scene_path_test()
    renderer.resize(800, 832)
    want_a, want_b = pmo.render(a, 800, 832), pmo.render(b, 800, 832)
    renderer.set_scene_bytes(a)
    for n in (1, 2, 3, 4, 7):
        for _ in range(n):
            renderer.render()
        assert np.array_equal(renderer.read_pixels(), want_a), n
    for _ in range(5):
        renderer.render()
    renderer.set_scene_bytes(b)  # no sync by the caller
    for _ in range(4):
        renderer.render()
    assert np.array_equal(renderer.read_pixels(), want_b)
    assert_ptcl_equal(renderer, pmo, b, 800, 832)
    # a caller-owned framebuffer interleaved with pipelined frames
    import torch

    t = torch.zeros((832, 800, 4), dtype=torch.uint8, device="cuda:0")
    renderer.render()
    renderer.render_to(t, None)
    renderer.render()
    renderer.sync()
    assert np.array_equal(t.cpu().numpy(), want_b)
    assert np.array_equal(renderer.read_pixels(), want_b)


@pytest.mark.parametrize("chunks", [1, 3])
def test_bench_two_rank_path_rehearsal_on_one_gpu(pm, pmo, tmp_path, chunks):
    """(chunks = 3: --gather-chunks, every rank's band rendered in three sub-bands by three contexts, each sub-band's
    gather posted on a second stream while the next sub-band renders.)
    bench.py's N>1 path (cost-balanced bands via pm_set_band, pm_render_to on torch's stream, the
    grouped gather straight into the final image on rank 0 inside every step) with both ranks on
    this box's one GPU and gloo as transport: the gathered 3840x2160 frame must equal the
    oracle's render, and rank 0 prints the JSON line with the strong-scaling fields."""
    import subprocess
    import sys

    dump = tmp_path / "frame.npy"
    env = dict(os.environ, PM_BENCH_SHARE_DEVICE="1", PM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29517 + chunks), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-config5",
           "--gather-chunks", str(chunks), "--dump", str(dump)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    js = json.loads(line)
    assert js["n_gpus"] == 2 and js["scaling"] == "strong" and js["value"] > 0 and js["sustained_mpix_s"] > 0
    cfg = js["config"]
    assert cfg["viewport"] == [3840, 2160] and cfg["band_cuts"][0] == 0 and cfg["band_cuts"][-1] == 135 and len(cfg["band_cuts"]) == 3
    assert cfg["t_render_ms"] > 0 and cfg["t_gather_ms"] > 0 and cfg["t_frame_e2e_ms"] > 0 and cfg["gather_chunks"] == chunks
    assert "rccl_lib" in cfg and "rccl_ranks" in cfg and 0 < js["roofline"]["frac_serial_frame"] <= js["roofline"]["frac"]
    # (event-timed step and the same step host-timed, medians of two separate runs of the loop.  Two processes share this GPU and gloo carries the
    #  gather through the host: the two agree within a factor of a few on a quiet box, and a hiccup in either run is not a failure of the path)
    assert 0 < js["t_frame_ms"] < 1000 and 0 < cfg["t_frame_e2e_ms"] < 1000
    wl = pm.workloads.tiger(3840, 2160)
    scene, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    got = np.load(dump)
    assert got.shape == (2160, 3840, 4)
    want = pmo.render(scene, 3840, 2160)
    bad = np.nonzero(np.any(got != want, axis=2))
    assert len(bad[0]) == 0, (len(bad[0]), int(bad[0].min()), int(bad[0].max()), int(bad[1].min()), int(bad[1].max()), got[bad[0][0], bad[1][0]].tolist(), cfg["band_cuts"])


def test_c_abi_gather_single_rank(pm, pmo, renderer):
    """pm_comm_* / pm_gather (RCCL bound at run time): with one rank the grouped send/recv is a
    self transfer, which still runs the whole C-ABI path -- id, communicator, band bookkeeping,
    ncclSend/ncclRecv on the frame's stream -- and must land the frame in the final image."""
    import torch

    scene = pmo.scene_cardioid()
    renderer.resize(1000, 600)
    renderer.set_scene_bytes(scene)
    comm = pm.Comm(renderer, pm.Comm.unique_id(), 0, 1)
    try:
        # ONE RCCL per process: the copy torch mapped (its process group would run on it) is the one bound
        info = comm.info()
        assert info["rccl_ranks"] == 1 and "rccl" in os.path.basename(info["rccl_lib"]), info
        mapped = {l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l}
        assert len({os.path.realpath(m) for m in mapped}) == 1 and os.path.realpath(info["rccl_lib"]) in {os.path.realpath(m) for m in mapped}, (info, mapped)
        full = torch.zeros((600, 1000, 4), dtype=torch.uint8, device="cuda:0")
        renderer.render()
        comm.gather([(0, 38)], root=0, full=full)  # src = the context's last frame
        renderer.sync()
        torch.cuda.synchronize()
        assert np.array_equal(full.cpu().numpy(), pmo.render(scene, 1000, 600))
        band = torch.zeros((608, 1000, 4), dtype=torch.uint8, device="cuda:0")
        full.zero_()
        s = torch.cuda.current_stream()
        renderer.render_to(band, s)
        comm.gather([(0, 38)], root=0, full=full, band=band, stream=s)
        torch.cuda.synchronize()
        assert np.array_equal(full.cpu().numpy(), pmo.render(scene, 1000, 600))
        with pytest.raises(pm.PietMetalError):
            comm.gather([(0, 37)], root=0, full=full)  # not this context's band
    finally:
        comm.close()


def test_plain_c_consumer_of_the_c_abi(pm, golden, tmp_path):
    """tests/cabi_smoke.c: C11, built with gcc against include/piet_metal_amd.h alone (calling
    conventions and struct layouts as a C compiler sees them, not as ctypes was told), linked to the
    library, run as its own process: encoder -> pinned scene buffer -> upload -> render -> read back,
    RGBA8 and BGRA8 targets, hashes against the oracle's (tests/golden/make_golden.py --cabi)."""
    import subprocess

    from piet_metal_amd import _lib

    lib = _lib.load()._name  # the library this test session runs on
    exe = str(tmp_path / "cabi_smoke")
    rocm = "/opt/rocm/lib"
    subprocess.check_call(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi_smoke.c"), "-o", exe, lib, "-lm",
                           f"-Wl,-rpath,{os.path.dirname(lib)}", f"-Wl,-rpath,{rocm}", f"-Wl,-rpath-link,{rocm}"])
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=600).stdout
    got = dict(line.split()[:2] for line in out.splitlines() if line and not line.startswith("tiles"))
    g = golden["cabi_smoke"]
    assert int(got["scene_bytes"]) == g["scene_bytes"] and out.split()[3] == g["scene_fnv1a64"], out
    assert got["rgba_fnv1a64"] == g["rgba_fnv1a64"], out
    assert got["bgra_fnv1a64"] == g["bgra_fnv1a64"], out
    assert "overflow 0" in out


def test_command_list_arena_overflow_grows_and_rerenders(pm, pmo, monkeypatch):
    """Lists are sized from what binning finds, so the arena can run out: pm_sync must notice,
    grow it and render the frame again (also with several frames in flight)."""
    monkeypatch.setenv("PM_PTCL_INITIAL_CMDS", "2048")
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(960, 540)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        scene = r.download_scene()
        want = pmo.render(scene, 960, 540)
        for _ in range(5):
            r.render()
        assert np.array_equal(r.read_pixels(), want)
        st = r.stats()
        assert st["overflow"] == 0 and st["ptcl_used_cmds"] > 2048
        for _ in range(3):
            r.render()
        assert np.array_equal(r.read_pixels(), want)
    finally:
        r.close()


def test_overflow_is_not_hidden_by_calls_that_wait_for_the_device(pm, pmo, monkeypatch):
    """pm_get_stats / pm_frame_latency / ... wait for the device themselves; a frame whose
    command-list arena ran out must still be found and repaired by the next pm_sync / pm_read_pixels
    (round-2 advisor finding: render -> stats -> read_pixels returned a frame with holes)."""
    monkeypatch.setenv("PM_PTCL_INITIAL_CMDS", "2048")
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(640, 360)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        want = pmo.render(r.download_scene(), wl.width, wl.height)
        r.render()
        st = r.stats()  # waits for the device (SyncAll) before anybody looked at the overflow flag
        assert st["overflow"] == 1
        assert np.array_equal(r.read_pixels(), want)
        assert r.stats()["overflow"] == 0
    finally:
        r.close()


def test_overflow_in_every_in_flight_frame_is_repaired(pm, pmo, monkeypatch):
    """Up to four frames are in flight, and pm_render_to frames each own a caller buffer: when the
    command-list arena runs out, pm_sync must repair EVERY target, not just the last frame's
    (round-1 advisor finding)."""
    import torch

    monkeypatch.setenv("PM_PTCL_INITIAL_CMDS", "2048")
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(960, 540)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        want = pmo.render(r.download_scene(), 960, 540)
        bufs = [torch.zeros((544, 960, 4), dtype=torch.uint8, device="cuda:0") for _ in range(3)]
        side = torch.cuda.Stream()
        r.render_to(bufs[0], None)
        r.render_to(bufs[1], side)
        r.render()
        r.render_to(bufs[2], None)
        del side  # the caller's stream may be gone before pm_sync
        r.sync()
        for k, b in enumerate(bufs):
            assert np.array_equal(b[:540].cpu().numpy(), want), k
        assert r.stats()["overflow"] == 0
        r.render()
        assert np.array_equal(r.read_pixels(), want)
    finally:
        r.close()


def test_failed_scene_replacement_leaves_no_stale_state(pm, pmo, renderer):
    """A rejected scene must not leave the old item count / index paired with the new bytes
    (round-1 advisor finding): rendering is refused until a valid scene is resident again."""
    good = pmo.scene_cardioid()
    renderer.resize(512, 512)
    renderer.set_scene_bytes(good)
    renderer.render()
    renderer.sync()
    bad = good.copy()
    bad[0:4] = np.frombuffer(struct.pack("<I", 0x7fffffff), np.uint8)  # n_items out of range
    with pytest.raises(pm.PietMetalError):
        renderer.set_scene_bytes(bad)
    with pytest.raises(pm.PietMetalError):
        renderer.render()
    assert renderer.stats()["n_items"] == 0
    renderer.set_scene_bytes(good)
    renderer.render()
    assert np.array_equal(renderer.read_pixels(), pmo.render(good, 512, 512))


@pytest.mark.parametrize("split,heavy", [("0", "32"), ("1", "32"), ("1", "4")])
def test_both_fine_modes_agree_with_the_oracle(pm, pmo, monkeypatch, split, heavy):
    """pm_fine_kernel renders a tile with one wave (row-sparse Fill fragments, list order) or, for
    long lists, with a whole workgroup (items evaluated in parallel, blends in list order).
    PM_FINE_SPLIT=0 forces the first, PM_HEAVY_STREAM=4 pushes almost every tile into the
    second.  All must be byte-exact, also where lists exceed one LDS chunk."""
    monkeypatch.setenv("PM_FINE_SPLIT", split)
    monkeypatch.setenv("PM_HEAVY_STREAM", heavy)
    r = pm.Renderer(0)
    try:
        for (w, h, fills) in ((960, 540, False), (1920, 1080, True)):
            wl = pm.workloads.tiger(w, h, fills_only=fills)
            r.resize(w, h)
            r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
            scene = r.download_scene()
            r.render()
            assert np.array_equal(r.read_pixels(), pmo.render(scene, w, h)), (w, h)
        ops = random_ops(77, 700, 256.0, opaque_mask=0)  # translucent: long lists: several chunks, several PrepareFills calls
        scene = encode_ops(pm, ops)
        assert np.array_equal(gpu_render(r, scene, 256, 256), pmo.render(scene, 256, 256))
    finally:
        r.close()


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("fold_clear", ["1", "0"])
@pytest.mark.parametrize("handout", ["0", "1", "2"])
def test_every_frame_path_switch_agrees_with_the_oracle(pm, pmo, monkeypatch, fused, fold_clear, handout):
    """The frame path has switches (pm_create reads them): list building fused into the tile kernel or
    as its own launch (PM_FUSED), resolved tiles cleared by the tile kernel's launch or by
    pm_clear_kernel (PM_FOLD_CLEAR), tiles dealt statically or drawn (PM_HANDOUT).  Every combination
    renders the same bytes and builds the same lists -- captured from the kernel that built them."""
    monkeypatch.setenv("PM_FUSED", fused)
    monkeypatch.setenv("PM_FOLD_CLEAR", fold_clear)
    monkeypatch.setenv("PM_HANDOUT", handout)
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(480, 270)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        scene = r.download_scene()
        for _ in range(3):  # (frames in flight: the hand-out decision of PM_HANDOUT=0 depends on it)
            r.render()
        assert np.array_equal(r.read_pixels(), pmo.render(scene, wl.width, wl.height))
        assert_ptcl_equal(r, pmo, scene, wl.width, wl.height)
        ops = random_ops(77, 300, extent=500.0)
        scene2 = encode_ops(pm, ops)
        got = gpu_render(r, scene2, 517, 500)
        assert np.array_equal(got, pmo.render(scene2, 517, 500))
        assert_ptcl_equal(r, pmo, scene2, 517, 500)
        if os.environ.get("PM_EXPECT_DENSE") == "1" and fused == "1":  # (tests/test_emu_cpu.py: a device so small that these scenes are dense)
            assert r.dense_kernel_frames() > 0
    finally:
        r.close()


def test_dense_scenes_get_the_one_wave_per_tile_kernel(pm, pmo, monkeypatch):
    """A frame whose tile kernel finds it dense -- its long lists, at a workgroup each, would occupy every wave -- tells the host
    (a pinned word), and the frames that follow get the tile kernel's one-wave-per-tile instantiation (80 VGPRs, 19 KB of LDS: six
    workgroups per CU).  Same bytes from both instantiations, alone and in flight; a new scene starts undecided again; lists are
    captured from the general kernel; PM_DENSE_KERNEL=0 never switches."""
    wl = pm.workloads.config4_blobs(3000, 1024, seed=0x5EED0009)
    tig = pm.workloads.tiger(480, 270)
    for mode in ("1", "0"):
        monkeypatch.setenv("PM_DENSE_KERNEL", mode)
        r = pm.Renderer(0)
        try:
            r.resize(wl.width, wl.height)
            r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
            scene = r.download_scene()
            want = pmo.render(scene, wl.width, wl.height)
            r.render()
            assert np.array_equal(r.read_pixels(), want)  # (the scene's first frame: the general kernel)
            assert r.dense_kernel_frames() == 0
            for _ in range(3):  # frames alone
                r.render()
                assert np.array_equal(r.read_pixels(), want)
            alone = r.dense_kernel_frames()
            assert (alone == 3) if mode == "1" else (alone == 0)
            for _ in range(8):  # ... and behind one another
                r.render()
            assert np.array_equal(r.read_pixels(), want)
            assert_ptcl_equal(r, pmo, scene, wl.width, wl.height)
            assert np.array_equal(r.read_pixels(), want)
            before = r.dense_kernel_frames()
            # a sparse scene: undecided again, and its own frames say "not dense"
            r.resize(tig.width, tig.height)
            r.flatten_and_encode(tig.paths, tig.affine, tig.width_scale)
            tscene = r.download_scene()
            for _ in range(3):
                r.render()
                assert np.array_equal(r.read_pixels(), pmo.render(tscene, tig.width, tig.height))
            assert r.dense_kernel_frames() == before
            # a scene without a single long list is the one-wave kernel's too (nothing in it would get a workgroup)
            gl = pm.workloads.heldout_glyphs(500, 512, 512)
            r.resize(gl.width, gl.height)
            r.flatten_and_encode(gl.paths, gl.affine, gl.width_scale)
            gscene = r.download_scene()
            gwant = pmo.render(gscene, gl.width, gl.height)
            for k in range(4):
                r.render()
                assert np.array_equal(r.read_pixels(), gwant), k
            assert r.stats()["heavy_tiles"] == 0 and r.stats()["queued_tiles"] > 100
            assert r.dense_kernel_frames() == (before + 3 if mode == "1" else before)
        finally:
            r.close()


@pytest.mark.parametrize("grid_per_cu", [None, "1"])
def test_one_launch_frames_agree_with_the_oracle(pm, pmo, monkeypatch, golden, grid_per_cu):
    """PM_ONE_LAUNCH=1: a lone frame is ONE launch of pm_frame_kernel -- every workgroup bins a strip row, renders that
    row's first tiles from its own LDS hand-over and then takes tiles other rows left in the frame's FIFOs (pieces stored
    write-through, read with agent-scope loads; csrc/pm_frame.hip).  Same bytes and the same command lists as the oracle:
    lists captured from the one-launch kernel's own capture instantiation; scenes with every item type; strip rows with
    several records; the 4K Tiger, whose 1 109 strip rows are more than the 1 024 resident workgroups (chains of two);
    a band; frames submitted behind one another (those take two launches) -- and no frame ever gave up waiting.
    grid_per_cu = 1: a quarter of the grid, so that small scenes chain strip rows too and 4K falls back to two launches."""
    monkeypatch.setenv("PM_ONE_LAUNCH", "1")
    if grid_per_cu is not None:
        monkeypatch.setenv("PM_FRAME_WG_PER_CU", grid_per_cu)
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(960, 540)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        scene = r.download_scene()
        r.render()
        assert r.one_launch_info()["applies"]
        assert np.array_equal(r.read_pixels(), pmo.render(scene, wl.width, wl.height))
        assert_ptcl_equal(r, pmo, scene, wl.width, wl.height)
        before = r.one_launch_info()["frames"]
        for _ in range(6):  # back to back: only frames that find the device idle are one launch
            r.render()
        assert np.array_equal(r.read_pixels(), pmo.render(scene, wl.width, wl.height))
        assert r.one_launch_info()["frames"] >= before
        for seed, n, w, h in ((31, 900, 600, 400), (32, 300, 1100, 300), (33, 120, 928, 912)):
            scene = encode_ops(pm, random_ops(seed, n, extent=float(max(w, h))))
            got = gpu_render(r, scene, w, h)
            assert np.array_equal(got, pmo.render(scene, w, h)), seed
            assert_ptcl_equal(r, pmo, scene, w, h)
        r.set_band(3, 17)
        r.render()
        assert np.array_equal(r.read_pixels(), pmo.render(scene, 928, 912)[48:272])
        # full size: the 4K Tiger (chains of strip rows at the default grid) and Tiger 1080p, against the committed goldens
        for wl, name in ((pm.workloads.tiger(3840, 2160), "tiger_3840x2160"), (pm.workloads.tiger(1920, 1080, fills_only=True), "tiger_1920x1080_fills")):
            r.resize(wl.width, wl.height)
            r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
            for _ in range(3):
                r.render()
                r.sync()
            assert hashlib.sha256(r.read_pixels().tobytes()).hexdigest() == golden[name]["rgba_sha256"], name
        info = r.one_launch_info()
        assert info["frames"] > before
        assert info["applies"], "a frame gave up waiting inside its launch"
    finally:
        r.close()


@pytest.mark.parametrize("wg_per_cu", [None, "1", "2", "7", "0"])
def test_binning_launch_variants(pm, pmo, monkeypatch, wg_per_cu):
    """pm_bin_kernel runs as chains of strip rows over a grid the chip holds at once (PM_BIN_WG_PER_CU
    per CU, five by default; 0: a workgroup per row whatever their number).  Same bytes whatever the
    grid, for frames submitted behind one another and for frames waited for one by one."""
    if wg_per_cu is not None:
        monkeypatch.setenv("PM_BIN_WG_PER_CU", wg_per_cu)
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(960, 540)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        scene = r.download_scene()
        want = pmo.render(scene, wl.width, wl.height)
        for _ in range(3):  # behind one another: the per-frame decision sees frames in flight
            r.render()
        assert np.array_equal(r.read_pixels(), want)
        for _ in range(2):  # one by one: nothing in flight
            r.render()
            r.sync()
        assert np.array_equal(r.read_pixels(), want)
        assert_ptcl_equal(r, pmo, scene, wl.width, wl.height)
    finally:
        r.close()


@pytest.mark.parametrize("wg_per_cu,row_lists", [(None, False), ("1", False), (None, True)])
def test_a_wave_per_strip_row(pm, pmo, monkeypatch, wg_per_cu, row_lists):
    """pm_bin_kernel with ONE wave per strip row (what frames with many times more light strip rows than the chip
    holds workgroups get, config 5; PM_BIN_WAVES=1 forces it): records of 64 candidates instead of 256, so rows
    with more candidates chain pieces; 16 tiles per wave in the candidates pass; no workgroup barrier.  Same
    bytes and the same command lists as the oracle -- with chains of strip rows per wave (a small grid), with
    per-tile-row item lists, over fills, strokes, lines, circles and compound fills."""
    monkeypatch.setenv("PM_BIN_WAVES", "1")
    if wg_per_cu is not None:
        monkeypatch.setenv("PM_BIN_WG_PER_CU", wg_per_cu)
    if row_lists:
        monkeypatch.setenv("PM_ROW_LIST_MIN_ITEMS", "1")
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(960, 540)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        scene = r.download_scene()
        for _ in range(3):
            r.render()
        assert np.array_equal(r.read_pixels(), pmo.render(scene, wl.width, wl.height))
        assert_ptcl_equal(r, pmo, scene, wl.width, wl.height)
        # strip rows with well over 64 candidates (several records per row) and long items (survivor lists beyond LDS)
        for seed, n, w, h in ((31, 900, 600, 400), (32, 300, 1100, 300)):
            scene = encode_ops(pm, random_ops(seed, n, extent=float(max(w, h))))
            r.resize(w, h)
            r.set_scene_bytes(scene)
            r.render()
            assert np.array_equal(r.read_pixels(), pmo.render(scene, w, h)), seed
            assert_ptcl_equal(r, pmo, scene, w, h)
        r.set_band(3, 17)
        r.render()
        assert np.array_equal(r.read_pixels(), pmo.render(scene, 1100, 300)[48:272])
    finally:
        r.close()


def test_one_launch_frame_that_gives_up_leaves_its_slot_clean(pm, pmo, monkeypatch):
    """PM_ONE_LAUNCH=1 with a wait budget of one microsecond (PM_ONE_LAUNCH_SPIN_US=1): waves of the one launch give up waiting for
    tiles almost at once, the frame has holes and raises the slot's pinned word.  pm_sync puts the hand-over state of EVERY such slot
    back to all zero -- also of a frame that was waited for (pm_get_stats) and superseded before pm_sync ever looked at it (round-5
    advisor finding: its FIFO entries stayed behind, and a later one-launch frame of that slot could pop one) -- renders what has to
    be rendered again with two launches, and keeps to two launches from then on.  Every frame read back equals the oracle's."""
    monkeypatch.setenv("PM_ONE_LAUNCH", "1")
    monkeypatch.setenv("PM_ONE_LAUNCH_SPIN_US", "1")
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(1280, 720)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        want = pmo.render(r.download_scene(), wl.width, wl.height)
        r.render()
        r.stats()   # waits for the frame without looking at its flags
        r.render()  # the next slot: supersedes the first frame on the context's own target
        r.sync()
        assert np.array_equal(r.read_pixels(), want)
        one = r.one_launch_info()["frames"]
        for _ in range(9):  # every slot again, several times
            r.render()
            r.sync()
            assert np.array_equal(r.read_pixels(), want)
        for _ in range(6):
            r.render()
        assert np.array_equal(r.read_pixels(), want)
        if one >= 1 and r.one_launch_info()["frames"] == one:
            pass  # a frame gave up: two launches ever since (nothing to assert when, on a quiet box, no wave ever had to wait that long)
    finally:
        r.close()


@pytest.mark.parametrize("mode,waves", [("1", None), ("2", None), ("2", "1")])
def test_heavy_strip_rows_cut_in_two(pm, pmo, monkeypatch, mode, waves):
    """The host side of tileKernel's dispatch geometry (PietRenderer.m:63-77) gives the heaviest strip rows TWO workgroups, tiles
    0-7 and 8-15 of the strip (pm_binning_plan_info): which rows, the frames' own binning kernels say (segment slots per strip row,
    read back once, at the third frame of a plan -- PM_BIN_SPLIT_SLOTS lowers the bar so that a small frame has such rows);
    PM_BIN_SPLIT=2 cuts every row from the first frame on.  A half bins the candidates, chunks and segments that can matter to
    ITS tiles -- a fill's segments left of it still count towards its backdrops -- and the reference's own predicates keep the
    strip's geometry (the phase-1 votes, quirk Q4).  Same bytes and the same lists as the oracle: before and after the plan is
    remade, with frames in flight (which bin from the list WITHOUT the cuts), with a wave per strip row, on bands, and on scenes
    with every item type."""
    monkeypatch.setenv("PM_BIN_SPLIT", mode)
    monkeypatch.setenv("PM_BIN_SPLIT_SLOTS", "48")
    if waves:
        monkeypatch.setenv("PM_BIN_WAVES", waves)
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(1000, 620)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        scene = r.download_scene()
        want = pmo.render(scene, wl.width, wl.height)
        for k in range(6):
            r.render()
            r.sync()  # (frames alone: the list with the cuts)
            assert np.array_equal(r.read_pixels(), want), k
        info = r.binning_plan_info()
        assert info["rows_cut"] > 0 and info["entries"] > info["rows_cut"], info
        if mode == "1":
            assert info["fed_back"] and info["plans_fed_back"] == 1, info
        assert_ptcl_equal(r, pmo, scene, wl.width, wl.height)
        for _ in range(5):  # frames in flight
            r.render()
        assert np.array_equal(r.read_pixels(), want)
        r.set_band(7, 29)
        for _ in range(5):
            r.render()
            r.sync()
        assert np.array_equal(r.read_pixels(), want[112:464])
        for seed, n, w, h in ((41, 700, 900, 500), (42, 300, 1100, 300)):
            scene = encode_ops(pm, random_ops(seed, n, extent=float(max(w, h))))
            r.resize(w, h)
            r.set_scene_bytes(scene)
            for _ in range(5):
                r.render()
                r.sync()
            assert np.array_equal(r.read_pixels(), pmo.render(scene, w, h)), seed
            assert_ptcl_equal(r, pmo, scene, w, h)
    finally:
        r.close()


@pytest.mark.parametrize("coarse_wg,fine_wg,fused", [("1", "1", "1"), ("7", "3", "0"), ("2", "9", "1")])
def test_persistent_grid_sizes(pm, pmo, monkeypatch, coarse_wg, fine_wg, fused):
    """PM_COARSE_WG_PER_CU / PM_FINE_WG_PER_CU size the persistent grids; the hand-out must cover
    every queued tile whatever the grid (few workgroups: many passes; many: empty ones)."""
    monkeypatch.setenv("PM_COARSE_WG_PER_CU", coarse_wg)
    monkeypatch.setenv("PM_FINE_WG_PER_CU", fine_wg)
    monkeypatch.setenv("PM_FUSED", fused)
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(960, 540)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        scene = r.download_scene()
        r.render()
        assert np.array_equal(r.read_pixels(), pmo.render(scene, wl.width, wl.height))
        assert_ptcl_equal(r, pmo, scene, wl.width, wl.height)
    finally:
        r.close()


@pytest.mark.parametrize("seed,n,extent,w,h", [(31, 200, 500.0, 520, 500), (32, 400, 300.0, 330, 310), (33, 60, 1500.0, 1400, 900)])
def test_even_odd_fills_and_nested_groups(pm, pmo, renderer, seed, n, extent, w, h):
    """The encoder extensions end to end on the GPU: scenes with even-odd fills (PietFill.flags)
    inside nested groups -- pixels and per-tile command lists against the oracle, and the nested
    scene against its own inlined form."""
    tree = extend_ops(seed, random_ops(seed, n, extent=extent, opaque_mask=0xFF if seed != 32 else 0))
    nested, flat = encode_ops(pm, tree, cap=4 << 20), encode_ops(pm, inline_ops(tree), cap=4 << 20)
    got = gpu_render(renderer, nested, w, h)
    assert np.array_equal(got, pmo.render(nested, w, h))
    assert_ptcl_equal(renderer, pmo, nested, w, h, maxc=2048)
    assert np.array_equal(renderer.download_scene(), nested)  # the flat form stays the renderer's business
    assert np.array_equal(gpu_render(renderer, flat, w, h), got)
    st = renderer.stats()
    assert st["n_items"] == len(inline_ops(tree))


def test_compound_fills(pm, pmo, renderer):
    """Extension D11 on the GPU: Fill items made of several sub-paths (holes, islands, one-point
    sub-paths, hundreds of sub-paths in one item so that chunks hold nothing but separators),
    pixels and command lists against the oracle; a malformed separator index stays inside the array."""
    rng = np.random.default_rng(51)
    ops = []
    for _ in range(120):
        m = int(rng.integers(3, 9))
        c = rng.uniform(20, 480, 2)
        pts = c + rng.uniform(-90, 90, (m, 2))
        subs = [pts, (c + (pts - c) * 0.5)[::-1]]
        if rng.random() < 0.5:
            subs.append(pts[:2] + rng.uniform(-40, 40, 2))
        rgba = int(rng.integers(0, 1 << 32)) | (0xFF if rng.random() < 0.5 else 0)
        ops.append(("fill_cp_eo" if rng.random() < 0.3 else "fill_cp", subs, rgba))
    many = [np.array([[x, y], [x + 3.0, y + 0.5], [x + 2.5, y + 3.0]]) for x in np.arange(10.0, 500.0, 7.0) for y in (100.0, 300.0, 303.5)]
    many += [np.array([[float(x), 250.0]]) for x in range(30)]  # one-point sub-paths: chunks of separators and degenerate segments
    ops.append(("fill_cp", many, 0x102030FF))
    ops += random_ops(52, 60, extent=500.0)
    scene = encode_ops(pm, ops, cap=4 << 20)
    got = gpu_render(renderer, scene, 512, 500)
    assert np.array_equal(got, pmo.render(scene, 512, 500))
    assert_ptcl_equal(renderer, pmo, scene, 512, 500, maxc=2048)
    # a separator whose index points outside the array is clamped, not followed (both sides)
    bad = encode_ops(pm, [("fill_cp", [many[0], many[1]], 0x102030FF)]).copy()
    items = struct.unpack("<I", bad[4:8].tobytes())[0]
    pix = struct.unpack("<I", bad[items + 16 : items + 20].tobytes())[0]
    bad[pix + 8 * 3 + 4 : pix + 8 * 3 + 8] = np.frombuffer(struct.pack("<I", 0x7FFFFFFF), np.uint8)
    assert np.array_equal(gpu_render(renderer, bad, 128, 128), pmo.render(bad, 128, 128))


def test_ellipses(pm, pmo, renderer):
    """Extension D10 on the GPU: Circle items with the ellipse bit -- sparse (one wave per tile),
    piled up (lists longer than a chunk, rendered by a workgroup), degenerate and clipped by the
    viewport -- pixels and command lists against the oracle; plain circles next to them unchanged."""
    rng = np.random.default_rng(41)
    ops = []
    for _ in range(300):
        c = rng.uniform(-20, 520, 2)
        if rng.random() < 0.7:
            rx, ry = rng.uniform(0.3, 90, 2)
            if rng.random() < 0.08:
                rx = 0.0
            ops.append(("ellipse", float(c[0]), float(c[1]), float(rx), float(ry)))
        else:
            ops.append(("circle", float(c[0]), float(c[1]), float(rng.uniform(1, 60))))
    for k in range(260):  # a pile on a few tiles: > 4 chunks of commands
        ops.append(("ellipse", 250.0 + 0.1 * k, 240.0 - 0.07 * k, 30.0 + 0.2 * k, 12.0 + 0.05 * k))
    ops += random_ops(42, 80, extent=500.0)  # fills and strokes in between and on top
    scene = encode_ops(pm, ops, cap=4 << 20)
    got = gpu_render(renderer, scene, 512, 500)
    assert np.array_equal(got, pmo.render(scene, 512, 500))
    assert_ptcl_equal(renderer, pmo, scene, 512, 500, maxc=2048)
    only_circles = encode_ops(pm, [op for op in ops if op[0] != "ellipse"], cap=4 << 20)
    assert np.array_equal(gpu_render(renderer, only_circles, 512, 500), pmo.render(only_circles, 512, 500))


def test_even_odd_through_the_device_flatten(pm, pmo, renderer):
    """PM_PATH_EVEN_ODD on a path reaches PietFill.flags through the flatten kernels: scene bytes
    equal the oracle's encoder, pixels equal its render (Tiger, every fill under even-odd)."""
    wl = pm.workloads.tiger(960, 540)
    ps = pm.PathSet(wl.paths.paths.copy(), wl.paths.els)
    ps.paths["flags"] |= np.where(ps.paths["flags"] & pm._lib.PM_PATH_FILL, pm._lib.PM_PATH_EVEN_ODD, 0).astype(np.uint32)
    renderer.resize(wl.width, wl.height)
    renderer.flatten_and_encode(ps, wl.affine, wl.width_scale)
    scene = renderer.download_scene()
    want_scene, _ = pmo.scene_from_paths(pmo.scaled_paths(ps.paths, wl.width_scale), ps.els, wl.affine)
    assert np.array_equal(scene, want_scene)
    renderer.render()
    got = renderer.read_pixels()
    assert np.array_equal(got, pmo.render(scene, wl.width, wl.height))
    plain, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    assert (got != pmo.render(plain, wl.width, wl.height)).any()  # the Tiger has self-overlapping outlines


@pytest.mark.parametrize("split", [0, 700])
def test_flatten_block_parallel_prefix_sums(pm, pmo, renderer, monkeypatch, split):
    """Above 16 384 elements the flatten stage's two prefix sums run as blocks of 1 024 in parallel (KScanA / KScanTops / KScanD /
    KScanApply) instead of one workgroup walking the arrays; PM_SCAN_SPLIT moves the threshold so that a scene the oracle encodes in
    a second goes that way: 2 500 blobs = 2 500 paths (three blocks of paths), 15 000 elements (15 blocks), fills, strokes and
    compound fills.  Byte for byte the CPU encoder's scene, as every flatten test."""
    monkeypatch.setenv("PM_SCAN_SPLIT", str(split))
    wl = pm.workloads.config4_blobs(2500, 512)
    _flatten_case(pm, pmo, renderer, wl)
    wl = pm.workloads.heldout_glyphs(1500, 512, 512)
    _flatten_case(pm, pmo, renderer, wl)
    monkeypatch.delenv("PM_SCAN_SPLIT")
    wl = pm.workloads.config4_blobs(2500, 512)  # and the single-workgroup sums on the same input
    _flatten_case(pm, pmo, renderer, wl)


def test_compound_fills_through_the_device_flatten(pm, pmo, renderer):
    """PM_PATH_COMPOUND: the flatten kernels write ONE Fill item per path, its sub-paths separated
    in the point array -- scene bytes equal the oracle's encoder, pixels equal its render (Tiger,
    every filled path compound, with and without the even-odd rule, strokes untouched)."""
    wl = pm.workloads.tiger(960, 540)
    for extra in (0, pm._lib.PM_PATH_EVEN_ODD):
        ps = pm.PathSet(wl.paths.paths.copy(), wl.paths.els)
        fill = (ps.paths["flags"] & pm._lib.PM_PATH_FILL) != 0
        ps.paths["flags"] |= np.where(fill, pm._lib.PM_PATH_COMPOUND | extra, 0).astype(np.uint32)
        renderer.resize(wl.width, wl.height)
        renderer.flatten_and_encode(ps, wl.affine, wl.width_scale)
        scene = renderer.download_scene()
        want_scene, n_items = pmo.scene_from_paths(pmo.scaled_paths(ps.paths, wl.width_scale), ps.els, wl.affine)
        assert np.array_equal(scene, want_scene)
        assert renderer.stats()["n_items"] == n_items < 304  # the Tiger has paths with several sub-paths
        renderer.render()
        assert np.array_equal(renderer.read_pixels(), pmo.render(scene, wl.width, wl.height))
        # the view changes, the resident paths are flattened again (animation path)
        aff2 = tuple(v * 0.7 for v in wl.affine[:4]) + (40.0, 25.0)
        renderer.reflatten(aff2, wl.width_scale * 0.7)
        want2, _ = pmo.scene_from_paths(pmo.scaled_paths(ps.paths, wl.width_scale * 0.7), ps.els, aff2)
        assert np.array_equal(renderer.download_scene(), want2)


def test_malformed_nested_groups_are_rejected(pm, renderer):
    good = encode_ops(pm, [("group", [("circle", 50.0, 50.0, 9.0)]), ("circle", 20.0, 20.0, 5.0)])
    renderer.resize(128, 128)
    renderer.set_scene_bytes(good)
    renderer.render()
    renderer.sync()
    items = struct.unpack("<I", good[4:8].tobytes())[0]
    for group_ix in (0, 4, 0xFFFFFFF0):  # itself (a cycle), misaligned into the header, out of range
        bad = good.copy()
        bad[items + 8 : items + 12] = np.frombuffer(struct.pack("<I", group_ix), np.uint8)
        with pytest.raises(pm.PietMetalError):
            renderer.set_scene_bytes(bad)
        with pytest.raises(pm.PietMetalError):
            renderer.render()


def test_f32_coverage_matches_the_f32_reference(pm, pmo, renderer):
    """North star: "coverage within 1 ULP of the f32 reference".  pm_fill_coverage evaluates one
    Fill item's commands with an f32 signedArea (the frame path keeps the reference's `half`); the
    oracle's pmo_fill_coverage is the same arithmetic on the CPU.  Every operation is one IEEE
    rounding on both sides, so the bar here is tighter than the north star's: 0 ULP."""
    def ulps(a, b):
        ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
        return int(np.abs(ia - ib).max())

    from test_oracle_cpu import _slit_double_square

    cases = []
    quad = np.array([(20.25, 20.5), (100.75, 23.5), (97.75, 100.25), (17.25, 96.5)])
    cases.append((encode_ops(pm, [("fill", quad, 0x102030FF)]), 0, 128, 128))
    cases.append((encode_ops(pm, [("fill_eo", _slit_double_square(), 0x20304080)]), 0, 304, 288))
    ops = [op for op in random_ops(91, 200, extent=500.0) if op[0] in ("fill", "circle")]
    scene = encode_ops(pm, ops)
    fills = [i for i, op in enumerate(ops) if op[0] == "fill"]
    cases += [(scene, i, 520, 500) for i in fills[:6]]
    wl = pm.workloads.tiger(960, 540)
    renderer.resize(wl.width, wl.height)
    renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    tiger = renderer.download_scene()
    t_items = struct.unpack("<I", tiger[4:8].tobytes())[0]
    t_fills = [i for i in range(struct.unpack("<I", tiger[0:4].tobytes())[0]) if tiger[t_items + 32 * i] == 3]
    cases += [(tiger, i, 960, 540) for i in (t_fills[0], t_fills[7], t_fills[40], t_fills[-1])]
    worst = 0
    for scene, item, w, h in cases:
        renderer.resize(w, h)
        renderer.set_scene_bytes(scene)
        got = renderer.fill_coverage(item)
        want = pmo.fill_coverage(scene, item, w, h)
        worst = max(worst, ulps(got, want))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (item, w, h, ulps(got, want))
        # the scene is intact afterwards: a frame still renders as before
        renderer.render()
        assert np.array_equal(renderer.read_pixels(), pmo.render(scene, w, h))
    assert worst == 0
    with pytest.raises(pm.PietMetalError):
        renderer.fill_coverage(10 ** 6)


def test_scene_buffer_pointer_survives_device_side_growth(pm, pmo):
    """A C host caches pm_scene_buffer's pointer like the reference caches _sceneBuf.contents
    (PietRenderer.m:204).  Calls that need more DEVICE room than the buffer has (here pm_fill_coverage
    appending its one-item group behind a scene that fills the buffer) must grow the device copy only
    (round-2 advisor finding: the pinned buffer used to be reallocated under the caller)."""
    r = pm.Renderer(0)
    try:
        buf = r.scene_buffer()
        p0, cap = buf.ctypes.data, buf.size
        scene = pmo.scene_path_test()
        buf[: scene.size] = scene
        r.resize(128, 832)
        r.upload_scene(cap)  # (the bytes behind the scene are padding nothing refers to)
        got = r.fill_coverage(0)
        want = pmo.fill_coverage(scene, 0, 128, 832)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        again = r.scene_buffer()
        assert again.ctypes.data == p0 and again.size == cap
        buf[: scene.size] = scene  # the cached pointer is still writable memory of the context
        r.upload_scene(scene.size)
        r.render()
        assert np.array_equal(r.read_pixels(), pmo.render(scene, 128, 832))
    finally:
        r.close()


def test_cli_renders_svg_to_png(pm, pmo, tmp_path):
    from piet_metal_amd import cli

    out = str(tmp_path / "tiger.png")
    assert cli.main(["tiger", out, "--width", "640", "--height", "400"]) == 0
    wl = pm.workloads.tiger(640, 400)
    scene, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    assert np.array_equal(cli.read_png_rgba(out), pmo.render(scene, 640, 400))


def test_second_svg_document_end_to_end(pm, pmo, tmp_path):
    """tests/data/shapes.svg -- nested groups with transforms, style declarations, opacity, both
    fill rules, every basic shape -- through the CLI (SVG document layer, on-device flatten, the
    three frame kernels, PNG) against the oracle on the same parsed paths."""
    from piet_metal_amd import cli

    src = os.path.join(ROOT, "tests", "data", "shapes.svg")
    out = str(tmp_path / "shapes.png")
    assert cli.main([src, out, "--width", "1200", "--height", "900", "--scale", "3"]) == 0
    ps = pm.PathSet.from_svg(open(src).read(), spec_defaults=True)
    assert len(ps.paths) == 21 and (ps.paths["flags"] & pm._lib.PM_PATH_EVEN_ODD).any() and (ps.paths["flags"] & pm._lib.PM_PATH_COMPOUND).any()
    scene, n_items = pmo.scene_from_paths(pmo.scaled_paths(ps.paths, 3.0), ps.els, (3.0, 0.0, 0.0, 3.0, 0.0, 0.0))
    want = pmo.render(scene, 1200, 900)
    got = cli.read_png_rgba(out)
    assert np.array_equal(got, want)
    # sanity of the picture itself: the even-odd outline has a hole where the non-zero twin is filled
    page = tuple(got[3 * 295, 3 * 395][:3])
    assert page == (0xEE, 0xF4, 0xFA)
    assert tuple(got[3 * 212, 3 * 103][:3]) == page        # even-odd: the doubly wound core is a hole
    assert tuple(got[3 * 212, 3 * 50][:3]) == (0xC9, 0x2A, 0x2A)   # its ring is painted
    assert tuple(got[3 * 212, 3 * 273][:3]) == (0, 0x80, 0x80)   # the non-zero twin's core is painted (teal)
    assert tuple(got[3 * 10, 3 * 308][:3]) == page              # the compound path's inner sub-path is a hole
    assert tuple(got[3 * 4, 3 * 302][:3]) == (0x5F, 0x3D, 0xC4)  # its ring is painted
    assert tuple(got[3 * 290, 3 * 26][:3]) == (0x0B, 0x72, 0x85)  # <use href="#leaf">: the midrib of the group from <defs>
    assert tuple(got[3 * 291, 3 * 100][:3]) != page               # <use href="#dot">: the <symbol>, painted with the fill of its <use>
    # without --scale the CLI fits the document's viewBox (400 x 300) into the viewport, centred
    assert cli.main([src, out, "--width", "1000", "--height", "600"]) == 0
    scene2, _ = pmo.scene_from_paths(pmo.scaled_paths(ps.paths, 2.0), ps.els, (2.0, 0.0, 0.0, 2.0, 100.0, 0.0))
    assert np.array_equal(cli.read_png_rgba(out), pmo.render(scene2, 1000, 600))


def test_animation_reflatten_resident_paths(pm, pmo, renderer, tmp_path):
    """Per-frame re-encode (PietRenderer.m:90-101, :145) on the device: pm_reflatten flattens the
    resident paths again under a new affine -- scene bytes and pixels of every frame equal the
    oracle's for that affine, and equal a fresh pm_flatten_and_encode."""
    from piet_metal_amd import cli

    wl = pm.workloads.tiger(640, 400)
    renderer.resize(wl.width, wl.height)
    with pytest.raises(pm.PietMetalError):
        pm.Renderer(0).reflatten(wl.affine, wl.width_scale)  # nothing resident yet in a fresh context
    renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    for k in range(4):
        aff = cli.spin_affine(wl.affine, np.deg2rad(25.0 * k), 320.0, 200.0)
        nbytes, nitems = renderer.reflatten(aff, wl.width_scale)
        scene = renderer.download_scene()
        want_scene, want_items = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, aff)
        assert nitems == want_items and np.array_equal(scene, want_scene), k
        renderer.render()
        assert np.array_equal(renderer.read_pixels(), pmo.render(scene, wl.width, wl.height)), k
    t = renderer.scene_timings()
    assert t["flatten_encode_ms"] > 0
    out = str(tmp_path / "spin.png")
    assert cli.main(["tiger", out, "--width", "320", "--height", "200", "--frames", "3", "--spin", "90"]) == 0
    assert all(os.path.exists(str(tmp_path / f"spin-{k:03d}.png")) for k in range(3))
    a = cli.read_png_rgba(str(tmp_path / "spin-001.png"))
    scene1, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, 1.0), wl.paths.els, cli.spin_affine((1.0, 0.0, 0.0, 1.0, 60.0, 0.0), np.deg2rad(30.0), 160.0, 100.0))
    assert np.array_equal(a, pmo.render(scene1, 320, 200))


@pytest.mark.parametrize("part_items", [None, 37])
def test_per_row_item_lists_large_scene_path(pm, pmo, monkeypatch, part_items):
    """Scenes with thousands of items bin through per-tile-row item lists
    (pm_rowcull_kernel); forced on here for small scenes, full frame and bands -- a workgroup per tile row, and
    (PM_ROW_LIST_PART_ITEMS=37) a row's scan cut into a dozen workgroups with their own places in the row's list,
    as scenes of > 2 048 items get it."""
    monkeypatch.setenv("PM_ROW_LIST_MIN_ITEMS", "1")
    if part_items is not None:
        monkeypatch.setenv("PM_ROW_LIST_PART_ITEMS", str(part_items))
    r = pm.Renderer(0)
    try:
        wl = pm.workloads.tiger(1000, 700)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        scene = r.download_scene()
        want = pmo.render(scene, 1000, 700)
        for _ in range(3):
            r.render()
        assert np.array_equal(r.read_pixels(), want)
        assert_ptcl_equal(r, pmo, scene, 1000, 700)
        r.set_band(10, 31)
        r.render()
        assert np.array_equal(r.read_pixels(), want[160:496])
        for seed in (21, 22):
            scene = encode_ops(pm, random_ops(seed, 400, extent=700.0))
            r.resize(700, 500)
            r.set_scene_bytes(scene)
            r.render()
            assert np.array_equal(r.read_pixels(), pmo.render(scene, 700, 500)), seed
    finally:
        r.close()


def test_maximum_viewports(pm, pmo):
    """The reference caps the grid at 256 x 256 tiles (PietShaderTypes.h:24-32); here the grid
    is dynamic up to the u16 bbox range.  A huge viewport must show the same picture in the
    region a small one covers, and be background elsewhere."""
    r = pm.Renderer(0)
    try:
        scene = pmo.scene_cardioid()
        want = pmo.render(scene, 2048, 1536)
        r.resize(16384, 12288)  # 1024 x 768 tiles, 805 Mpix
        r.set_scene_bytes(scene)
        r.render()
        r.render()
        got = r.read_pixels()
        assert got.shape == (12288, 16384, 4)
        assert np.array_equal(got[:1536, :2048], want)
        assert (got[1536:] == 255).all() and (got[:1536, 2048:] == 255).all()
        del got
        # the widest viewport the u16 bboxes allow, one and a half tile rows tall
        r.resize(65535, 24)
        r.set_scene_bytes(scene)
        r.render()
        got = r.read_pixels()
        assert np.array_equal(got[:, :2048], want[:24])
        assert (got[:, 2048:] == 255).all()
    finally:
        r.close()


def test_frame_latency_api(pm, pmo, renderer):
    scene = pmo.scene_cardioid()
    renderer.resize(1024, 768)
    renderer.set_scene_bytes(scene)
    lat = renderer.frame_latency(20)
    assert 0 < lat["min_ms"] <= lat["median_ms"] < 50
    assert np.array_equal(renderer.read_pixels(), pmo.render(scene, 1024, 768))


@pytest.mark.parametrize("streams,slots", [(1, 1), (1, 3), (2, 5), (3, 2)])
def test_other_pipeline_depths(pm, pmo, monkeypatch, streams, slots):
    """PM_FRAME_STREAMS / PM_SLOTS other than the default 4 / 4: slot reuse across streams is
    ordered by events, the pictures do not change."""
    monkeypatch.setenv("PM_FRAME_STREAMS", str(streams))
    monkeypatch.setenv("PM_SLOTS", str(slots))
    r = pm.Renderer(0)
    try:
        scene = pmo.scene_cardioid()
        r.resize(640, 480)
        r.set_scene_bytes(scene)
        want = pmo.render(scene, 640, 480)
        for n in (1, 2, 5, 9):
            for _ in range(n):
                r.render()
            assert np.array_equal(r.read_pixels(), want), n
        tm = r.time_frames(5, pipelined=True)
        assert tm["total_ms"] > 0 and tm["fine_ms"] > 0
    finally:
        r.close()


def test_config5_full_size_properties(pm, pmo, renderer):
    """BASELINE config 5 at full size on one GPU (5 x 5 Tigers, 8192^2, 7 600 items: the
    per-row item lists are in use): the eight 64-row bands of the 8-GPU split reproduce the
    full frame, rendering is idempotent, and two tile rows are checked against the oracle."""
    wl = pm.workloads.config5_tiger_grid()
    renderer.resize(wl.width, wl.height)
    nbytes, nitems = renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    assert nitems == 7600
    scene = renderer.download_scene()
    renderer.render()
    full = renderer.read_pixels()
    renderer.render()
    renderer.render()
    assert np.array_equal(full, renderer.read_pixels())
    for rank in (0, 3, 7):
        r0, r1 = pm.workloads.band_rows(512, 8, rank)
        assert (r0, r1) == (64 * rank, 64 * rank + 64)
        renderer.set_band(r0, r1)
        renderer.render()
        assert np.array_equal(renderer.read_pixels(), full[r0 * 16 : r1 * 16]), rank
    P = pmo.Ptcl(scene, wl.width, wl.height)
    assert np.array_equal(full[200 * 16 : 202 * 16], P.render_rows(200, 202))
    P.close()


@pytest.mark.timeout(120)
def test_non_finite_coordinates_do_not_hang(pm, renderer):
    """The reference leaves NaN / infinite / huge coordinates undefined (float -> int casts in
    tileKernel); here they must neither hang nor fault, and the context must stay usable."""
    rng = np.random.default_rng(3)
    bad = [np.nan, np.inf, -np.inf, 3.0e38, -3.0e38, 1.0e9]
    buf = np.zeros(1 << 20, np.uint8)
    e = pm.Encoder(buf)
    n = 120
    e.begin_group(n)
    for i in range(n):
        pts = rng.uniform(0, 400, (int(rng.integers(2, 9)), 2))
        pts[int(rng.integers(0, len(pts))), int(rng.integers(0, 2))] = bad[i % len(bad)]
        if i % 2:
            e.polyline(pts, 0xFF3366CC, 3.0)
        else:
            e.fill(pts, 0xFF22AA44)
    e.end_group()
    renderer.resize(512, 512)
    renderer.set_scene_bytes(buf[: e.bytes_used])
    for _ in range(3):
        renderer.render()
    img = renderer.read_pixels()
    assert img.shape == (512, 512, 4)
    # still alive
    scene = buf[: e.bytes_used].copy()
    renderer.set_scene_bytes(scene)
    renderer.render()
    renderer.sync()


@pytest.mark.parametrize("seed,n,extent,w,h", [(20206, 279, 400.0, 1424, 398), (20740, 417, 900.0, 1173, 1384)])
def test_fuzz_regressions(pm, pmo, renderer, seed, n, extent, w, h):
    """Scenes a 2 000-scene fuzz run found one pixel off by one in: `half(area * wd)` had been
    fused into v_fma_mixlo_f16, which rounds once instead of twice (ToHalf in pm_fine.hip)."""
    # (generated with the colour mask the fuzz tool had that day: mostly translucent items)
    scene = encode_ops(pm, random_ops(seed, n, extent=extent, opaque_mask=0xFF000000), cap=4 << 20)
    assert np.array_equal(gpu_render(renderer, scene, w, h), pmo.render(scene, w, h))


def test_strict_barrier_build_renders_the_same_bytes(pm, golden):
    """LdsBarrier() (csrc/pm_kernels_common.h) leaves out the wait for outstanding global stores: global data
    written before it must not be read by another wave after it.  libpiet_metal_amd_strict.so is the same
    library with every such barrier a full __syncthreads(): BASELINE configs 2-4 at full size must come out
    byte for byte the same from both builds -- and equal to the committed goldens."""
    import subprocess
    import sys

    outs = {}
    for variant in ("", "strict"):
        env = dict(os.environ, PM_LIB_VARIANT=variant)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "render_hashes.py"), "config2", "config3", "config4"],
                           env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        lines = dict(l.split(" ", 1) for l in p.stdout.strip().splitlines())
        assert lines.pop("lib") == ("libpiet_metal_amd_strict.so" if variant else "libpiet_metal_amd.so")
        outs[variant] = lines
    assert outs[""] == outs["strict"]
    for cfg, name in (("config2", "tiger_1920x1080_fills"), ("config3", "tiger_3840x2160"), ("config4", "blobs_10000_4096")):
        assert outs[""][cfg] == golden[name]["rgba_sha256"], cfg


def test_view_changes_keep_their_plan_while_the_boxes_stay_inside(pm, pmo, renderer):
    """A view change (pm_reflatten) plans binning for item boxes widened by a tile; the scenes that follow keep that
    plan -- no host sizing, no uploads -- while their boxes stay inside the widened ones, and get a new one when an
    item leaves its box, changes its segment count, or the viewport changes.  Forty small steps of a drifting, slowly
    zooming Tiger, then a jump: every frame equals the oracle's render of the scene bytes the device flattened."""
    wl = pm.workloads.tiger(560, 352)
    renderer.resize(wl.width, wl.height)
    renderer.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    renderer.render()
    fed0 = renderer.binning_plan_info()["plans_fed_back"]
    a = list(wl.affine)
    plans = []
    for k in range(44):
        if k == 40:  # a jump: boxes leave the plan's
            a[4] += 90.0
            a[5] += 37.0
        a[4] += 0.37
        a[5] -= 0.21
        a[0] *= 1.0007
        a[3] *= 1.0007
        renderer.reflatten(tuple(a), wl.width_scale * a[0] / wl.affine[0])
        scene = renderer.download_scene()
        for _ in range(1 + k % 3):  # frames in flight on several slots
            renderer.render()
        assert np.array_equal(renderer.read_pixels(), pmo.render(scene, wl.width, wl.height)), k
        plans.append(renderer.scene_timings()["binning_plans"])
        if k == 20:
            assert_ptcl_equal(renderer, pmo, scene, wl.width, wl.height)
    # the first view change plans (wide boxes); the steps up to the jump keep that plan but for the few where the zoom has
    # pushed an item out of its box -- and for ONE plan remade from what the first frames' binning kernels reported (the heaviest
    # strip rows cut in two, pm_binning_plan_info); the jump plans again
    fed = renderer.binning_plan_info()["plans_fed_back"] - fed0
    assert fed <= 1 and plans[3] - plans[0] <= fed and plans[39] - plans[0] <= 4 + fed, plans
    assert plans[40] == plans[39] + 1, plans
    renderer.resize(wl.width + 64, wl.height)  # another viewport: the plan goes
    renderer.reflatten(tuple(a), wl.width_scale * a[0] / wl.affine[0])
    scene = renderer.download_scene()
    renderer.render()
    assert np.array_equal(renderer.read_pixels(), pmo.render(scene, wl.width + 64, wl.height))

