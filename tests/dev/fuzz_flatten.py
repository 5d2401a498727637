"""Randomised check of the on-device flatten + encode (GPU box): random paths (lines, quads,
cubics, several subpaths, open and closed, fills and strokes incl. thin strokes) under random
affines -- device-built scene bytes against the oracle's CPU encoder, then pixels.

    python tests/dev/fuzz_flatten.py [first_seed] [count]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import piet_metal_amd as pm
from piet_metal_amd import _lib
from oracle import pmo


def random_pathset(rng, n_paths, extent):
    els, paths = [], []
    for _ in range(n_paths):
        e0 = len(els)
        for _sub in range(int(rng.integers(1, 4))):
            p = rng.uniform(0, extent, 2)
            els.append((_lib.PM_EL_MOVE, [p[0], p[1], 0, 0, 0, 0]))
            for _seg in range(int(rng.integers(1, 7))):
                kind = rng.integers(0, 3)
                step = float(rng.choice([3.0, 30.0, 150.0]))
                q = [p + rng.uniform(-step, step, 2) for _ in range(3)]
                if kind == 0:
                    els.append((_lib.PM_EL_LINE, [q[0][0], q[0][1], 0, 0, 0, 0])); p = q[0]
                elif kind == 1:
                    els.append((_lib.PM_EL_QUAD, [q[0][0], q[0][1], q[1][0], q[1][1], 0, 0])); p = q[1]
                else:
                    els.append((_lib.PM_EL_CURVE, [q[0][0], q[0][1], q[1][0], q[1][1], q[2][0], q[2][1]])); p = q[2]
            if rng.random() < 0.6:
                els.append((_lib.PM_EL_CLOSE, [0] * 6))
        flags = int(rng.integers(1, 4))  # fill, stroke or both
        flags |= (4 if rng.random() < 0.3 else 0) | (8 if rng.random() < 0.4 else 0)  # even-odd rule, compound fill
        rgba = lambda: (int(rng.integers(0, 1 << 24)) << 8 | (0xFF if rng.random() < 0.4 else int(rng.integers(1, 255)))) & 0xFFFFFFFF
        width = float(rng.choice([0.05, 0.3, 1.0, 4.0]))
        paths.append((e0, len(els), flags, rgba(), rgba(), width))
    E = np.zeros(len(els), pm.PathSet.EL_DTYPE)
    for i, (t, p) in enumerate(els):
        E["tag"][i] = t; E["p"][i] = p
    P = np.array(paths, dtype=pm.PathSet.PATH_DTYPE)
    return pm.PathSet(P, E)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    r = pm.Renderer(0)
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed * 104729 + 7)
        ps = random_pathset(rng, int(rng.integers(1, 120)), float(rng.choice([200.0, 800.0])))
        s = float(rng.choice([0.5, 1.0, 2.7, 8.0]))
        th = rng.uniform(0, 6.28) if rng.random() < 0.5 else 0.0
        aff = (s * np.cos(th), s * np.sin(th), -s * np.sin(th), s * np.cos(th), float(rng.uniform(-50, 200)), float(rng.uniform(-50, 200)))
        w, h = int(rng.integers(64, 1600)), int(rng.integers(64, 1200))
        r.resize(w, h)
        r.flatten_and_encode(ps, aff, s)
        dev = r.download_scene()
        ref, _ = pmo.scene_from_paths(pmo.scaled_paths(ps.paths, s), ps.els, aff)
        ok = np.array_equal(dev, ref)
        if ok:
            r.render()
            ok = np.array_equal(r.read_pixels(), pmo.render(ref, w, h))
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed}: bytes equal: {np.array_equal(dev, ref)} sizes {dev.size} {ref.size}", flush=True)
    print(f"flatten fuzz: {count} path sets from seed {first}: {bad} mismatches, {time.time() - t0:.1f} s")
    return 1 if bad else 0

sys.exit(main())
