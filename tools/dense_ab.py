"""Developer A/B on one box: the tile kernel's one-wave-per-tile instantiation (six workgroups per CU) against the general one on the
dense workloads -- lone frame, sustained, the tile kernel alone; PM_DENSE_KERNEL / PM_FINE_WG_PER_CU_DENSE[_INFLIGHT] variants."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
W = pm.workloads
cases = {"config4": W.config4_blobs, "config5": W.config5_tiger_grid, "held2": lambda: W.heldout_workloads()["held2"], "held3": lambda: W.heldout_workloads()["held3"],
         "config3": lambda: W.tiger(3840, 2160)}
variants = {"general": {"PM_DENSE_KERNEL": "0"}, "dense 6/6": {}, "dense 6/5": {"PM_FINE_WG_PER_CU_DENSE_INFLIGHT": "5"}, "dense 6/4": {"PM_FINE_WG_PER_CU_DENSE_INFLIGHT": "4"},
            "dense 5/5": {"PM_FINE_WG_PER_CU_DENSE": "5", "PM_FINE_WG_PER_CU_DENSE_INFLIGHT": "5"}}
names = sys.argv[1:] or ["config4", "config5", "held2"]
for name in names:
    wl = cases[name]()
    rs = {}
    for v, env in variants.items():
        os.environ.update(env)
        rs[v] = pm.Renderer(0)
        for k in env: os.environ.pop(k)
    dig = {}
    for v, r in rs.items():
        r.resize(wl.width, wl.height); r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        for _ in range(12): r.render()
        r.sync()
        dig[v] = hashlib.sha256(r.read_pixels().tobytes()).hexdigest()[:12]
    print(name, "same bytes:", len(set(dig.values())) == 1)
    n = 100 if wl.width * wl.height < 2e7 else 40
    best = {v: [1e9, 1e9] for v in rs}
    for rep in range(3):
        for v, r in rs.items():
            lone = r.frame_latency(20)["median_ms"] * 1e3
            t0 = time.perf_counter()
            for _ in range(n): r.render()
            r.sync()
            sus = (time.perf_counter() - t0) / n * 1e6
            best[v] = [min(best[v][0], lone), min(best[v][1], sus)]
    g = best["general"]
    for v, (lone, sus) in best.items():
        print("   %-10s lone %8.1f us (%+5.1f %%)   sustained %8.1f us (%+5.1f %%)   dense frames %d" % (v, lone, (lone / g[0] - 1) * 100, sus, (sus / g[1] - 1) * 100, rs[v].dense_kernel_frames()))
    for r in rs.values(): r.close()
