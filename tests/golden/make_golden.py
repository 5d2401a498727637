"""Generates tests/golden/golden.json: the pins this repo creates because the
reference has none (no tests, no golden images).

Everything here is produced by the CPU oracle (oracle/) in the build container
from inputs both sides share: the reference's three scene generators
(src/lib.rs:257-328) and the BASELINE configs.  Run from the repo root:

    python tests/golden/make_golden.py

Inputs that need the product's SVG front-end (Tiger path elements) are hashed too,
so a parser change shows up as a golden mismatch instead of silently moving both
sides.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import piet_metal_amd as pm  # noqa: E402
from oracle import pmo  # noqa: E402


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ptcl_sha(P) -> str:
    """Per-tile command lists: for every tile in row-major order, the command count (u32) followed
    by the 24-byte commands as the reference lays them out (TestApp/GenTypes.h:330-495)."""
    hsh = hashlib.sha256()
    for ty in range(P.tiles_y):
        for tx in range(P.tiles_x):
            c = np.ascontiguousarray(P.cmds(tx, ty), dtype=np.uint32)
            hsh.update(np.uint32(len(c)).tobytes())
            hsh.update(c.tobytes())
    return hsh.hexdigest()


def scene_entry(scene, w, h, with_image=True, with_f32=True):
    P = pmo.Ptcl(scene, w, h)
    tot, mx = P.total_cmds()
    e = {"scene_bytes": int(scene.size), "scene_sha256": sha(scene), "viewport": [w, h], "total_cmds": tot, "max_cmds_per_tile": mx}
    if with_image:
        img = P.render()
        e["rgba_sha256"] = sha(img)
        e["rgba_sum"] = int(img.astype(np.uint64).sum())
        if with_f32:
            f32 = P.render(pmo.MODE_F32)
            e["half_vs_f32_max_lsb"] = int(np.abs(img.astype(int) - f32.astype(int)).max())
    solid = np.array([[P.solid(tx, ty) for tx in range(P.tiles_x)] for ty in range(P.tiles_y)], np.uint32)
    e["solid_sha256"] = sha(solid)
    e["ptcl_sha256"] = ptcl_sha(P)
    P.close()
    return e


def big_main():
    """python tests/golden/make_golden.py --big : full-size pins of BASELINE configs 4 and 5
    (minutes of CPU each; merged into golden.json, everything else untouched)."""
    import time

    path = os.path.join(os.path.dirname(__file__), "golden.json")
    out = json.load(open(path))
    for wl in [pm.workloads.config4_blobs(), pm.workloads.config5_tiger_grid()]:
        t0 = time.time()
        scene, n_items = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
        e = scene_entry(scene, wl.width, wl.height, with_f32=False)
        e["n_items"] = n_items
        out[wl.name] = e
        print(wl.name, e, f"{time.time() - t0:.0f} s", flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


def held_main():
    """python tests/golden/make_golden.py --held : full-size pins of the three held-out workloads
    (piet_metal_amd/workloads.py, heldout_workloads; merged into golden.json)."""
    import time

    path = os.path.join(os.path.dirname(__file__), "golden.json")
    out = json.load(open(path))
    for key, wl in pm.workloads.heldout_workloads().items():
        t0 = time.time()
        scene, n_items = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
        e = scene_entry(scene, wl.width, wl.height, with_f32=False)
        e["n_items"] = n_items
        e["heldout"] = key
        out[wl.name] = e
        print(key, wl.name, e, f"{time.time() - t0:.0f} s", flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


def ext_scenes():
    """The scenes that pin the encoder extensions (even-odd, nested groups, ellipses, compound fills,
    DESIGN.md 2 decisions D9-D11) and the SVG document layer: name -> (scene bytes, width, height)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_cpu import _oracle_encode, extend_ops, random_ops

    out = {}
    for seed, n, extent, w, h in [(82, 160, 300.0, 320, 304), (91, 260, 600.0, 640, 576)]:
        out[f"extended_ops_{seed}"] = (_oracle_encode(pmo, extend_ops(seed, random_ops(seed, n, extent=extent))), w, h)
    svg = open(os.path.join(ROOT, "tests", "data", "shapes.svg")).read()
    for spec in (False, True):
        ps = pm.PathSet.from_svg(svg, spec_defaults=spec)
        scene, _ = pmo.scene_from_paths(pmo.scaled_paths(ps.paths, 3.0), ps.els, (3.0, 0.0, 0.0, 3.0, 0.0, 0.0))
        out["shapes_svg_x3_" + ("svg_rules" if spec else "tiger_rules")] = (scene, 1200, 900)
    return out


def ext_main():
    """python tests/golden/make_golden.py --ext : pins of the extension scenes (merged into golden.json)."""
    path = os.path.join(os.path.dirname(__file__), "golden.json")
    out = json.load(open(path))
    for name, (scene, w, h) in ext_scenes().items():
        out[name] = scene_entry(scene, w, h)
        print(name, out[name], flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def fnv1a64(a) -> str:
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def cabi_main():
    """python tests/golden/make_golden.py --cabi : what tests/cabi_smoke.c (a plain-C consumer of the
    C ABI) must print -- the cardioid scene rendered at 1024x768 by the oracle, hashed with the
    FNV-1a the C program carries (merged into golden.json)."""
    path = os.path.join(os.path.dirname(__file__), "golden.json")
    out = json.load(open(path))
    scene = pmo.scene_cardioid()
    img = pmo.render(scene, 1024, 768)
    out["cabi_smoke"] = {"viewport": [1024, 768], "scene_bytes": int(scene.size), "scene_fnv1a64": fnv1a64(scene),
                         "rgba_fnv1a64": fnv1a64(img), "bgra_fnv1a64": fnv1a64(img[:, :, [2, 1, 0, 3]]), "rgba_sha256": sha(img)}
    print(out["cabi_smoke"])
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def main():
    if "--cabi" in sys.argv:
        return cabi_main()
    if "--big" in sys.argv:
        return big_main()
    if "--ext" in sys.argv:
        return ext_main()
    if "--held" in sys.argv:
        return held_main()
    out = {}
    out["path_test_512x832"] = scene_entry(pmo.scene_path_test(), 512, 832)
    out["cardioid_2048x1536"] = scene_entry(pmo.scene_cardioid(), 2048, 1536)
    tig = pm.PathSet.tiger()
    out["tiger_paths"] = {"n_paths": len(tig.paths), "n_els": len(tig.els), "paths_sha256": sha(tig.paths), "els_sha256": sha(tig.els)}
    tig_na = pm.PathSet.tiger(reject_arc_paths=True)
    out["tiger_paths_no_arcs"] = {"n_paths": len(tig_na.paths), "n_els": len(tig_na.els)}
    for wl in [pm.workloads.tiger_reference(), pm.workloads.tiger(1920, 1080, fills_only=True), pm.workloads.tiger(3840, 2160),
               pm.workloads.tiger(480, 270), pm.workloads.config1_rect(), pm.workloads.config1_rect(True)]:
        scene, n_items = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
        e = scene_entry(scene, wl.width, wl.height)
        e["n_items"] = n_items
        out[wl.name] = e
    # a 64x64 px crop of the reference Tiger (eye region) as raw RGBA for eyeballing / exact compare
    wl = pm.workloads.tiger_reference()
    scene, _ = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    img = pmo.render(scene, wl.width, wl.height)
    np.save(os.path.join(os.path.dirname(__file__), "tiger_x8_crop_704_496_64x64.npy"), img[496:560, 704:768].copy())
    a, b, c = pmo.luts()
    out["luts"] = {"srgb2lin_sha256": sha(a), "unorm2h_sha256": sha(b), "lin2srgb_sha256": sha(c)}
    path = os.path.join(os.path.dirname(__file__), "golden.json")
    if os.path.exists(path):  # keep the full-size pins of `--big` runs
        for k, v in json.load(open(path)).items():
            out.setdefault(k, v)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
