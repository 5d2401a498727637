"""Developer profiling: what a scene replacement costs, piece by piece (host wall clock around pm_flatten_and_encode, and -- under
rocprofv3 --kernel-trace --stats -- the flatten kernels' own durations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
W = pm.workloads
cases = {"config3": lambda: W.tiger(3840, 2160), "config4": W.config4_blobs, "config5": W.config5_tiger_grid, "held3": lambda: W.heldout_workloads()["held3"]}
for name in sys.argv[1:] or ["config4", "config5", "held3"]:
    wl = cases[name]()
    r = pm.Renderer(0)
    r.resize(wl.width, wl.height)
    ts, tr, tf = [], [], []
    for k in range(12):
        t0 = time.perf_counter(); r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale); t1 = time.perf_counter()
        r.render(); r.sync(); t2 = time.perf_counter()
        t3 = time.perf_counter(); r.reflatten(wl.affine, wl.width_scale); t4 = time.perf_counter()
        r.render(); r.sync()
        ts.append((t1 - t0) * 1e3); tf.append((t2 - t1) * 1e3); tr.append((t4 - t3) * 1e3)
    print(name, "paths", len(wl.paths.paths), "els", len(wl.paths.els), "flatten_and_encode ms: first %.3f min %.3f | first frame after it min %.3f | reflatten min %.3f" % (ts[0], min(ts[1:]), min(tf[1:]), min(tr)), r.scene_timings())
    r.close()
