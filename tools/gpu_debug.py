import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import piet_metal_amd as pm
from oracle import pmo
from test_host_cpu import encode_ops
from gpu_check import run
rng = np.random.default_rng(7)
ops = []
for i in range(700):
    c = rng.uniform(40, 200, 2)
    pts = c + rng.uniform(-60, 60, (int(rng.integers(3, 9)), 2))
    rgba = (int(rng.integers(0, 1 << 24)) << 8) | (0xFF if i % 97 == 0 else int(rng.integers(0x20, 0xFF)))
    ops.append(("fill", pts, rgba) if i % 3 else ("poly", pts, rgba, float(rng.uniform(0.5, 6))))
r = pm.Renderer(0)
for n in (700, 300, 256, 200, 100):
    scene = encode_ops(pm, ops[:n], cap=1 << 22)
    print("n =", n)
    run(f"many{n}", r, scene, 256, 256)
scene = encode_ops(pm, ops[:700], cap=1 << 22)
r.resize(256, 256); r.set_scene_bytes(scene); r.render(); got = r.read_pixels()
want = pmo.render(scene, 256, 256)
d = (got != want).any(axis=2)
for (tx, ty) in [(10, 8), (10, 9)]:
    print("tile", tx, ty)
    for y in range(16):
        print("".join("X" if d[ty*16+y, tx*16+x] else "." for x in range(16)))
P = pmo.Ptcl(scene, 256, 256)
for (tx, ty) in [(10, 8), (10, 9)]:
    c = P.cmds(tx, ty)
    print("tile", tx, ty, "n", len(c), "tags", "".join(str(t) for t in c[:, 0]))
