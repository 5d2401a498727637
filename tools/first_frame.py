"""Developer timing: where does a scene's first frame go (host wall clock)?  tools/first_frame.py [config5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
name = sys.argv[1] if len(sys.argv) > 1 else "config5"
wl = {"config2": lambda: pm.workloads.tiger(1920, 1080, fills_only=True), "config3": lambda: pm.workloads.tiger(3840, 2160),
      "config4": pm.workloads.config4_blobs, "config5": pm.workloads.config5_tiger_grid}[name]()
r = pm.Renderer(0)
for rep in range(3):
    t0 = time.perf_counter(); r.resize(wl.width, wl.height); t1 = time.perf_counter()
    r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale); t2 = time.perf_counter()
    r.render(); t3 = time.perf_counter(); r.sync(); t4 = time.perf_counter()
    r.render(); r.sync(); t5 = time.perf_counter()
    st = r.scene_timings()
    print(f"{name} rep {rep}: resize {1e3*(t1-t0):.3f} flatten+scene {1e3*(t2-t1):.3f} (kernels+readback {st['flatten_encode_ms']:.3f} index {st['scene_index_ms']:.3f}) render submit {1e3*(t3-t2):.3f} (arena {st['arena_setup_ms']:.3f}) sync {1e3*(t4-t3):.3f} second frame {1e3*(t5-t4):.3f} | total first {1e3*(t4-t0):.3f} ms")
    r.resize(64, 64)  # (forget the viewport: the next repetition resizes again)
