"""Developer timing: BASELINE config 1 (one rect, 512 x 512) -- lone frame, frames in flight, host cost per submitted frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
r = pm.Renderer(0)
wl = pm.workloads.config1_rect()
r.resize(wl.width, wl.height)
r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
for _ in range(50): r.render()
r.sync()
print("lone", r.time_frames(200, per_kernel=True, pipelined=False))
print("pipelined", r.time_frames(2000, per_kernel=False, pipelined=True))
n = 5000
t0 = time.perf_counter()
for _ in range(n): r.render()
t1 = time.perf_counter(); r.sync(); t2 = time.perf_counter()
print(f"host submit {1e6*(t1-t0)/n:.2f} us per frame; sustained {1e6*(t2-t0)/n:.2f} us per frame ({n} frames)")
print("latency", r.frame_latency(200))
