"""How fast can the host submit frames?  (pm_render = 2 launches)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
r = pm.Renderer(0)
for wl in (pm.workloads.config1_rect(), pm.workloads.tiger(1920, 1080, fills_only=True), pm.workloads.tiger(3840, 2160)):
    r.resize(wl.width, wl.height)
    r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    for _ in range(50): r.render()
    r.sync()
    n = 3000
    t0 = time.perf_counter()
    for _ in range(n): r.render()
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    print(f"{wl.name}: submit {1e6*(t1-t0)/n:.1f} us/frame, until done {1e6*(t2-t0)/n:.1f} us/frame")
