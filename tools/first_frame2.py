"""Developer timing: config 5's first frame in a context that has rendered the 4K Tiger before (bench.py's order)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
r = pm.Renderer(0)
w3 = pm.workloads.tiger(3840, 2160)
r.resize(w3.width, w3.height); r.flatten_and_encode(w3.paths, w3.affine, w3.width_scale)
for _ in range(40): r.render()
r.sync()
wl = pm.workloads.config5_tiger_grid()
t0 = time.perf_counter(); r.resize(wl.width, wl.height); t1 = time.perf_counter()
r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale); t2 = time.perf_counter()
r.render(); t3 = time.perf_counter(); r.sync(); t4 = time.perf_counter()
st = r.scene_timings()
print(f"resize {1e3*(t1-t0):.3f} flatten+scene {1e3*(t2-t1):.3f} (kernels+readback {st['flatten_encode_ms']:.3f} index {st['scene_index_ms']:.3f}) render submit {1e3*(t3-t2):.3f} (arena {st['arena_setup_ms']:.3f}) sync {1e3*(t4-t3):.3f} | total {1e3*(t4-t0):.3f} ms")
