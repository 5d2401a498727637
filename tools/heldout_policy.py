"""Do the frame path's policy switches generalise?  Three held-out workloads (piet_metal_amd/workloads.py, heldout_workloads:
Tiger 2560x1440 with strokes, 2 k blobs at 2048^2, 20 k glyph-like paths at 4K -- none of them drove a threshold) plus, for
reference, the four the thresholds were chosen on.  For every switch of pm_create, the lone frame (pm_frame_latency, median)
and the sustained frame (four in flight) under the default and under each alternative, on one box, alternating.
   python tools/heldout_policy.py [--all] [out.json]
A default that is more than 5 % slower than an alternative on a held-out scene is flagged."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm

SWITCHES = {
    "PM_HANDOUT": ["1", "2"],                    # default 0: drawn for a lone frame, static behind other frames
    "PM_FOLD_CLEAR": ["0", "1"],                 # default 2: folded for a lone frame / small viewports
    "PM_BIN_WAVES": ["1", "4"],                  # default: by the number and weight of strip rows
    "PM_BIN_WAVES_INFLIGHT": ["4"],              # default 1
    "PM_FINE_WG_PER_CU_INFLIGHT": ["5"],         # default 3
    "PM_HEAVY_STREAM_LONE": ["24", "72"],        # default 40
    "PM_HEAVY_STREAM": ["40", "112"],            # default 72
    "PM_BIN_PRIO_SLOTS": ["0", "1000000"],       # default 320
    "PM_ROW_LIST_MIN_ITEMS": ["1", "1000000000"],  # default 2048
    "PM_FINE_SPLIT": ["0"],                      # default 1: long lists get a workgroup
    "PM_DENSE_FACTOR": ["1", "16"],              # default 4: one wave per tile once long lists x 4 fill the grid (1: the rule of rounds 2-4)
    "PM_DENSE_KERNEL": ["0"],                    # default 1: dense frames get the one-wave-per-tile instantiation of the tile kernel
    "PM_FINE_WG_PER_CU_DENSE_INFLIGHT": ["6"],   # default 4 (that kernel's grid behind other frames)
    "PM_ROW_LIST_PART_ITEMS": ["1000000000"],    # default: parts of >= 2 048 items, about four workgroups per CU (this: one workgroup per tile row)
    "PM_ONE_LAUNCH": ["1"],                      # default 0: two launches per frame
}


def measure(wl, env):
    # (the switches stay set for the whole measurement: pm_create reads most of them, but the ones of the binning plan --
    #  PM_ROW_LIST_MIN_ITEMS, PM_ROW_LIST_PART_ITEMS -- are read when a scene's first frame is planned)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    r = None
    try:
        r = pm.Renderer(0)
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        for _ in range(40):
            r.render()
        r.sync()
        lone = r.frame_latency(60)["median_ms"]
        n = 200
        for _ in range(40):
            r.render()
        r.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            r.render()
        r.sync()
        sus = (time.perf_counter() - t0) / n * 1e3
        return lone, sus
    finally:
        if r is not None:
            r.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    out_path = next((a for a in sys.argv[1:] if a.endswith(".json")), None)
    cases = dict(pm.workloads.heldout_workloads())
    if "--all" in sys.argv:
        cases.update({"config2": pm.workloads.tiger(1920, 1080, fills_only=True), "config3": pm.workloads.tiger(3840, 2160),
                      "config4": pm.workloads.config4_blobs(), "config5": pm.workloads.config5_tiger_grid()})
    report, flagged = {}, []
    for name, wl in cases.items():
        base = [measure(wl, {}) for _ in range(3)]
        b_lone, b_sus = min(b[0] for b in base), min(b[1] for b in base)
        report[name] = {"workload": wl.name, "default": {"lone_ms": round(b_lone, 5), "sustained_ms": round(b_sus, 5)}, "switches": {}}
        print(f"{name} ({wl.name}): default lone {b_lone * 1e3:.1f} us, sustained {b_sus * 1e3:.1f} us/frame", flush=True)
        for sw, alts in SWITCHES.items():
            for alt in alts:
                lone, sus = min(measure(wl, {sw: alt}) for _ in range(2))
                d_l, d_s = (b_lone / lone - 1) * 100, (b_sus / sus - 1) * 100  # > 0: the alternative is faster
                report[name]["switches"][f"{sw}={alt}"] = {"lone_ms": round(lone, 5), "sustained_ms": round(sus, 5),
                                                            "default_slower_lone_pct": round(d_l, 1), "default_slower_sustained_pct": round(d_s, 1)}
                mark = ""
                if name.startswith("held") and (d_l > 5 or d_s > 5):
                    mark = "   <-- default more than 5 % behind"
                    flagged.append((name, f"{sw}={alt}", round(d_l, 1), round(d_s, 1)))
                print(f"   {sw}={alt:<11} lone {lone * 1e3:8.1f} us ({-d_l:+5.1f} %)   sustained {sus * 1e3:8.1f} us ({-d_s:+5.1f} %){mark}", flush=True)
    report["_flagged"] = flagged
    print("flagged:", flagged)
    if out_path:
        json.dump(report, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
