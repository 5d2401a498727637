"""Developer A/B on one box: the lone frame as ONE launch (pm_frame_kernel) and as two (PM_ONE_LAUNCH=0), same process,
alternating contexts.  Prints t_frame (median / min of pm_frame_latency), the one-launch kernel's own duration and,
against the oracle-free cross-check, whether the two contexts' frames are the same bytes."""
import os, sys, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import piet_metal_amd as pm

W = pm.workloads
CASES = {"config3": lambda: W.tiger(3840, 2160), "config2": lambda: W.tiger(1920, 1080, fills_only=True),
         "tiger1440": lambda: W.tiger(2560, 1440), "config4": W.config4_blobs, "config5": W.config5_tiger_grid}
names = sys.argv[1:] or ["config3", "config2"]
rounds = int(os.environ.get("PM_AB_ROUNDS", "3"))


def make(mode):
    os.environ["PM_ONE_LAUNCH"] = str(mode)
    r = pm.Renderer(0)
    os.environ.pop("PM_ONE_LAUNCH")
    return r


for name in names:
    wl = CASES[name]()
    ctx = {m: make(m) for m in (1, 0)}
    digest = {}
    for m, r in ctx.items():
        r.resize(wl.width, wl.height)
        r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        for _ in range(30):
            r.render()
            r.sync()
        digest[m] = hashlib.sha256(r.read_pixels().tobytes()).hexdigest()[:16]
    info = ctx[1].one_launch_info()
    print(name, "one-launch applies:", info, "same bytes:", digest[1] == digest[0], digest)
    for k in range(rounds):
        for m, r in ctx.items():
            lat = r.frame_latency(200)
            extra = ""
            if m == 1 and r.one_launch_info()["applies"]:
                extra = " kernel_alone_us %.2f" % (r.time_one_launch(100) * 1e3)
            if m == 0:
                al = r.time_frames(20)
                extra = " kernels_alone_us " + json.dumps({kk: round(v * 1e3, 2) for kk, v in al.items() if kk.endswith("_ms") and v})
            print("  [%s] t_frame_us median %.2f min %.2f%s" % ("one" if m else "two", lat["median_ms"] * 1e3, lat["min_ms"] * 1e3, extra))
    print("  after:", ctx[1].one_launch_info())
    for r in ctx.values():
        r.close()
