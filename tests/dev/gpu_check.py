"""Developer check (GPU box): product vs oracle on a few scenes, with diagnostics."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piet_metal_amd as pm
from oracle import pmo

def compare_ptcl(r, scene, w, h, maxc=1024, verbose=5):
    P = pmo.Ptcl(scene, w, h)
    counts, solid, cmds = r.capture_ptcl(maxc)
    nbad = 0
    for ty in range(P.tiles_y):
        for tx in range(P.tiles_x):
            oc = P.cmds(tx, ty); n = len(oc)
            ok = counts[ty, tx] == n and solid[ty, tx] == P.solid(tx, ty)
            if ok and n <= maxc:
                ok = np.array_equal(cmds[ty, tx, :n], oc)
            if not ok:
                nbad += 1
                if nbad <= verbose:
                    print(f"  PTCL mismatch tile ({tx},{ty}): gpu n={counts[ty,tx]} solid={solid[ty,tx]:08x} oracle n={n} solid={P.solid(tx,ty):08x}")
                    m = min(n, int(counts[ty, tx]), maxc)
                    for i in range(m):
                        if not np.array_equal(cmds[ty, tx, i], oc[i]):
                            print(f"    first diff at cmd {i}: gpu {cmds[ty,tx,i]} oracle {oc[i]}")
                            break
    print(f"  ptcl: {nbad} mismatching tiles of {P.tiles_x*P.tiles_y}; total cmds {P.total_cmds()}")
    return nbad

def run(name, r, scene, w, h, ptcl=True):
    r.resize(w, h)
    r.set_scene_bytes(scene)
    r.render(); r.sync()
    got = r.read_pixels()
    t = time.time(); want = pmo.render(scene, w, h); to = time.time() - t
    bad = int((got != want).any(axis=2).sum())
    print(f"{name}: {w}x{h} scene {scene.size} B: {bad} differing pixels (oracle {to:.2f}s) stats {r.stats()}")
    if bad:
        ys, xs = np.nonzero((got != want).any(axis=2))
        print("  first diffs:", [(int(x), int(y), got[y, x].tolist(), want[y, x].tolist()) for x, y in list(zip(xs, ys))[:5]])
        print("  tiles with diffs:", sorted(set((int(x)//16, int(y)//16) for x, y in zip(xs, ys)))[:20])
    if ptcl:
        compare_ptcl(r, scene, w, h)
    return bad

def main():
    r = pm.Renderer(0)
    bad = 0
    bad += run("path_test", r, pmo.scene_path_test(), 512, 832)
    bad += run("cardioid", r, pmo.scene_cardioid(), 2048, 1536)
    # device flatten vs oracle flatten
    for wl in [pm.workloads.tiger_reference(), pm.workloads.tiger(1920, 1080, fills_only=True), pm.workloads.tiger(3840, 2160),
               pm.workloads.config1_rect(), pm.workloads.config1_rect(True)]:
        r.resize(wl.width, wl.height)
        t = time.time(); nb, ni = r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale); tf = time.time() - t
        dev_scene = r.download_scene()
        ref_scene, ref_items = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
        same = np.array_equal(dev_scene, ref_scene)
        print(f"{wl.name}: device flatten {nb} B {ni} items in {tf*1e3:.1f} ms; identical to oracle: {same}")
        if not same:
            bad += 1
            if dev_scene.size == ref_scene.size:
                d = np.nonzero(dev_scene != ref_scene)[0]
                print("  first differing bytes at", d[:10], "of", d.size)
            else:
                print("  sizes", dev_scene.size, ref_scene.size)
        bad += run(wl.name, r, ref_scene, wl.width, wl.height, ptcl=wl.width <= 1920)
        tm = r.time_frames(20)
        print(f"  timing: {tm}  => {wl.width*wl.height/ (tm['total_ms']/tm['iters']*1e-3)/1e6:.0f} Mpix/s")
    print("TOTAL BAD", bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
