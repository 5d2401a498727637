"""Developer profiling: the lone frame taken apart (kernels and the gaps between them)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
W = pm.workloads
cases = {"config3": W.tiger(3840, 2160), "config2": W.tiger(1920, 1080, fills_only=True)}
r = pm.Renderer(0)
for name, wl in cases.items():
    r.resize(wl.width, wl.height)
    r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
    for _ in range(50): r.render()
    r.sync()
    print(name, json.dumps({k: round(v * 1e3, 2) for k, v in r.frame_timeline(200).items()}), "(us)")
