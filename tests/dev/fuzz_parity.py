"""Randomised parity sweep (GPU box): random scenes x random viewports x random bands,
GPU pixels (and, for a subset, per-tile command lists) against the oracle.

    python tests/dev/fuzz_parity.py [first_seed] [count] [--ext]   (--ext: even-odd fills, nested groups)
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piet_metal_amd as pm
from oracle import pmo
from test_host_cpu import random_ops, encode_ops, extend_ops

def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ext = "--ext" in sys.argv
    r = pm.Renderer(0)
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed * 7919 + 13)
        n = int(rng.integers(1, 500))
        extent = float(rng.choice([120.0, 400.0, 900.0, 2000.0]))
        w, h = int(rng.integers(16, 1800)), int(rng.integers(16, 1400))
        ops = random_ops(seed, n, extent=extent)
        if ext:  # even-odd fills + nested groups
            ops = extend_ops(seed, ops)
        scene = encode_ops(pm, ops, cap=8 << 20)
        r.resize(w, h)
        r.set_scene_bytes(scene)
        frames = int(rng.integers(1, 4))
        for _ in range(int(os.environ.get("PM_FUZZ_FRAMES", "0")) or frames):  # (PM_FUZZ_FRAMES=5: past the frame at which a plan is remade from the frames' report)
            r.render()
        got = r.read_pixels()
        want = pmo.render(scene, w, h)
        ok = np.array_equal(got, want)
        if ok and seed % 5 == 0:  # command lists too
            P = pmo.Ptcl(scene, w, h)
            counts, solid, cmds = r.capture_ptcl(2048)
            for ty in range(P.tiles_y):
                for tx in range(P.tiles_x):
                    oc = P.cmds(tx, ty)
                    if counts[ty, tx] != len(oc) or solid[ty, tx] != P.solid(tx, ty) or not np.array_equal(cmds[ty, tx, : len(oc)], oc):
                        ok = False
            P.close()
        if ok and seed % 3 == 0 and h >= 64:  # a band of tile rows
            ty = (h + 15) // 16
            a = int(rng.integers(0, ty - 1)); b = int(rng.integers(a + 1, ty + 1))
            r.set_band(a, b)
            r.render()
            ok = np.array_equal(r.read_pixels(), want[a * 16 : min(b * 16, h)])
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed}: n={n} extent={extent} viewport {w}x{h}", flush=True)
    print(f"fuzz: {count} scenes from seed {first}: {bad} mismatches, {time.time() - t0:.1f} s")
    return 1 if bad else 0

sys.exit(main())
