"""Reproduce one seed of fuzz_parity.py with diagnostics:  python tests/dev/fuzz_one.py <seed>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piet_metal_amd as pm
from oracle import pmo
from test_host_cpu import random_ops, encode_ops

seed = int(sys.argv[1])
rng = np.random.default_rng(seed * 7919 + 13)
n = int(rng.integers(1, 500))
extent = float(rng.choice([120.0, 400.0, 900.0, 2000.0]))
w, h = int(rng.integers(16, 1800)), int(rng.integers(16, 1400))
ops = random_ops(seed, n, extent=extent)
scene = encode_ops(pm, ops, cap=4 << 20)
r = pm.Renderer(0)
r.resize(w, h)
r.set_scene_bytes(scene)
nrender = int(rng.integers(1, 4))
for _ in range(nrender): r.render()
got = r.read_pixels()
want = pmo.render(scene, w, h)
diff = (got != want).any(axis=2)
print(f"seed {seed}: n={n} extent={extent} {w}x{h} renders={nrender}: {int(diff.sum())} differing pixels")
if diff.any():
    ys, xs = np.nonzero(diff)
    tiles = sorted(set((int(x) // 16, int(y) // 16) for x, y in zip(xs, ys)))
    print("tiles:", tiles[:20])
    for x, y in list(zip(xs, ys))[:5]: print((int(x), int(y)), got[y, x].tolist(), want[y, x].tolist())
P = pmo.Ptcl(scene, w, h)
counts, solid, cmds = r.capture_ptcl(2048)
nbad = 0
for ty in range(P.tiles_y):
    for tx in range(P.tiles_x):
        oc = P.cmds(tx, ty)
        if counts[ty, tx] != len(oc) or solid[ty, tx] != P.solid(tx, ty) or not np.array_equal(cmds[ty, tx, : len(oc)], oc):
            nbad += 1
            if nbad <= 4:
                print(f"PTCL tile ({tx},{ty}): gpu n={counts[ty,tx]} solid={solid[ty,tx]:08x} | oracle n={len(oc)} solid={P.solid(tx,ty):08x}")
                m = min(len(oc), int(counts[ty, tx]))
                for i in range(m):
                    if not np.array_equal(cmds[ty, tx, i], oc[i]):
                        print("   first diff at", i, "gpu", cmds[ty, tx, i], "oracle", oc[i]); break
print("ptcl mismatching tiles:", nbad)
if h >= 64:
    ty = (h + 15) // 16
    a = int(rng.integers(0, ty - 1)); b = int(rng.integers(a + 1, ty + 1))
    r.set_band(a, b); r.render()
    bd = (r.read_pixels() != want[a * 16 : min(b * 16, h)]).any(axis=2)
    print(f"band [{a},{b}): {int(bd.sum())} differing pixels")
