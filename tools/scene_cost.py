"""Developer profiling: what a scene replacement / a view change costs on the host clock."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
wl = {"config3": lambda: pm.workloads.tiger(3840, 2160), "config5": pm.workloads.config5_tiger_grid, "config4": pm.workloads.config4_blobs}[os.environ.get("PM_TL_WORKLOAD", "config3")]()
r = pm.Renderer(0)
r.resize(wl.width, wl.height)
def T(f, n=1):
    r.sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    r.sync(); return (time.perf_counter() - t0) / n * 1e3
print("workload", wl.name)
print("flatten_and_encode (first, allocs): %.3f ms" % T(lambda: r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)), r.scene_timings())
print("first render: %.3f ms" % T(r.render), r.scene_timings())
print("flatten_and_encode (again): %.3f ms" % T(lambda: r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)), r.scene_timings())
print("render after it: %.3f ms" % T(r.render), r.scene_timings())
a = list(wl.affine)
def anim():
    a[4] += 0.25
    r.reflatten(tuple(a), wl.width_scale)
    r.render()
for k in range(3):
    print("reflatten + render: %.3f ms" % T(anim, 20), r.scene_timings())
def refl():
    a[4] += 0.25
    r.reflatten(tuple(a), wl.width_scale)
print("reflatten alone: %.3f ms" % T(refl, 20), r.scene_timings())
print("render alone (scene unchanged): %.3f ms" % T(r.render, 20))
