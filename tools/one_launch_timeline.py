"""Developer profiling: ONE one-launch frame (pm_frame_kernel's profiling instantiation) taken apart per workgroup and wave."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import piet_metal_amd as pm

W = pm.workloads
CASES = {"config3": lambda: W.tiger(3840, 2160), "config2": lambda: W.tiger(1920, 1080, fills_only=True), "tiger1440": lambda: W.tiger(2560, 1440)}
name = os.environ.get("PM_TL_WORKLOAD", "config3")
wl = CASES[name]()
r = pm.Renderer(0)
r.resize(wl.width, wl.height)
r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
for _ in range(20):
    r.render(); r.sync()
best = None
for rep in range(int(os.environ.get("PM_TL_REPS", "5"))):
    buf = np.zeros((4096, 32), np.uint64)
    n = C.c_size_t(0)
    pm._lib.check(r._lib.pm_debug_time_frame(r._h, buf.ctypes.data, buf.shape[0], C.byref(n)), "pm_debug_time_frame")
    d = buf[: n.value].astype(np.int64)
    t0 = d[:, 0].min()
    span = (d[:, 8:32].reshape(-1, 4, 6)[:, :, 2].max() - t0) / 100.0
    if best is None or span < best[0]: best = (span, d)
span, d = best
t0 = d[:, 0].min()
us = lambda x: (x - t0) / 100.0
n_wg = len(d)
has_row = d[:, 1] - d[:, 0] > 50  # binned something (> 0.5 us)
w = d[:, 8:32].reshape(n_wg, 4, 6)
own, idle, ex, nl, tl, polls = [w[:, :, k] for k in range(6)]
print(f"{name}: workgroups {n_wg}, with a strip row {has_row.sum()}, span {span:.1f} us (best of the reps)")
print("entry: last workgroup starts at %.2f us" % us(d[:, 0]).max())
b = us(d[has_row, 1])
print("strip row binned + handed over: mean %.1f p50 %.1f p90 %.1f max %.1f us" % (b.mean(), np.median(b), np.percentile(b, 90), b.max()))
ko = own[has_row]
kept = ko > 0
if kept.sum():
    print("kept single-wave tiles: %d, done at mean %.1f p90 %.1f max %.1f us; duration after binning mean %.1f" % (
        kept.sum(), us(ko[kept]).mean(), np.percentile(us(ko[kept]), 90), us(ko[kept]).max(),
        ((ko - d[has_row, 1][:, None])[kept] / 100.0).mean()))
nwg, twg = d[:, 2], d[:, 3] / 100.0
print("workgroup tiles: %d by %d workgroups, mean %.1f us each, most per workgroup %d" % (nwg.sum(), (nwg > 0).sum(), twg.sum() / max(nwg.sum(), 1), nwg.max()))
print("FIFO single-wave tiles: %d, mean %.1f us each, per wave max %d" % (nl.sum(), tl.sum() / 100.0 / max(nl.sum(), 1), nl.max()))
fi = idle[idle > 0]
print("first idle moment per wave: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (us(fi).mean(), np.percentile(us(fi), 10), np.median(us(fi)), np.percentile(us(fi), 90), us(fi).max()))
e = us(ex)
print("wave exit: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (e.mean(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max()))
print("polls that found nothing: total %d, per wave mean %.1f max %d" % (polls.sum(), polls.mean(), polls.max()))
# the tail: which workgroups end last, and what they did
last = np.argsort(-ex.max(axis=1))[:8]
for g in last:
    print("  wg %4d row %s binned %.1f  wg-tiles %d (%.1f us)  light %s (%.1f us)  exit %.1f" % (
        g, bool(has_row[g]), us(d[g, 1]), nwg[g], twg[g], nl[g].tolist(), tl[g].sum() / 100.0, us(ex[g]).max()))
# how busy is the machine over time: waves inside a tile or binning, per 2 us bucket (approximate: binning until d[:,1])
r.close()
