"""Developer profiling: how well does binning's per-tile estimate (relevant segments + closing commands) order the tile kernel's work?
Per-slot durations (pm_debug_time_tiles) joined with the captured command lists; a least-squares cost model on the list's make-up; and a
replay of the hand-out (every wave takes the next slot when it is free) in four orders: as measured, longest first (the bound), by the
current estimate, by the fitted model."""
import heapq, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
W = pm.workloads
wl = {"config2": lambda: W.tiger(1920, 1080, fills_only=True), "config3": lambda: W.tiger(3840, 2160), "config4": W.config4_blobs,
      "config5": W.config5_tiger_grid, "held1": lambda: W.heldout_workloads()["held1"]}[os.environ.get("PM_TL_WORKLOAD", "config3")]()
r = pm.Renderer(0)
r.resize(wl.width, wl.height)
r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
for _ in range(3):
    r.render(); r.sync()
t = r.time_tiles()
counts, solid, cmds = r.capture_ptcl(int(os.environ.get("PM_TL_MAXCMDS", "512")))
t = t[t[:, 0] > 0]
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
tile = (t[:, 2] & 0x7fffffff).astype(np.int64); quarter = (t[:, 2] >> 31).astype(bool)
wave = (t[:, 3] >> 32).astype(np.int64)
us = 1e-2
dur = (end - start) * us
t0 = start.min()
n_waves = len(np.unique(wave))
print(wl.name, "slots", len(t), "waves", n_waves, "span %.1f us" % ((end.max() - t0) * us), "sum %.0f us -> %.1f per wave" % (dur.sum(), dur.sum() / n_waves))
tx = counts.shape[1]
tags = cmds[..., 0]
valid = np.arange(cmds.shape[2])[None, None, :] < counts[..., None]
feat = {}
for name, tg in (("fill_seg", 4), ("line_seg", 3), ("fill_edge", 6), ("draw_fill", 7), ("stroke", 5), ("solid", 8), ("circle", 2)):
    feat[name] = ((tags == tg) & valid).sum(axis=2).reshape(-1)
ty_of = tile // tx; tx_of = tile % tx
flat = tile  # (full frame: tile index == row * tiles_x + column)
m = ~quarter
X = np.stack([np.ones(m.sum())] + [feat[k][flat[m]].astype(np.float64) for k in feat], axis=1)
y = dur[m]
coef, *_ = np.linalg.lstsq(X, y, rcond=None)
pred = X @ coef
print("cost model (us): const %.2f  " % coef[0] + "  ".join("%s %.3f" % (k, c) for k, c in zip(feat, coef[1:])))
est_now = (feat["fill_seg"] + feat["line_seg"] + feat["draw_fill"] + feat["stroke"] + feat["solid"] + feat["circle"] + feat["fill_edge"])[flat[m]].astype(np.float64)
print("correlation with the measured duration: current estimate %.3f, fitted model %.3f" % (np.corrcoef(est_now, y)[0, 1], np.corrcoef(pred, y)[0, 1]))
res = y - pred
worst = np.argsort(-res)[:12]
idx = np.flatnonzero(m)
for i in worst:
    s = idx[i]
    print("  under-predicted: slot %5d tile (%3d,%3d) dur %.1f pred %.1f start %.1f  " % (s, ty_of[s], tx_of[s], y[i], pred[i], (start[s] - t0) * us) + " ".join("%s %d" % (k, feat[k][flat[s]]) for k in feat))

# the slow light tiles: where do they sit, what stage took the time, and who shares their workgroup?
lo = lambda c: (t[:, c] & 0xffffffff).astype(np.float64) * us
hi = lambda c: (t[:, c] >> 32).astype(np.float64) * us
coarse = np.where(t[:, 6] > 0, (t[:, 6].astype(np.int64) - start) * us, 0.0)
ncmd = (t[:, 3] & 0xffffffff).astype(np.int64)
if t.shape[1] >= 12:
    st = {"hdr": lo(8), "cand": hi(8), "own": lo(9), "scan": hi(9), "seg": lo(10), "emit": hi(10)}
    for i in worst:
        s = idx[i]
        wg = wave[s] // 4
        mates = np.flatnonzero((wave // 4 == wg) & (np.arange(len(t)) != s))
        print("  slot %5d wave %5d wg %4d (xcd %d) list %.1f [" % (s, wave[s], wg, wg % 8, coarse[s]) + " ".join("%s %.1f" % (k, v[s]) for k, v in st.items()) + "] render %.1f | workgroup mates: " % (dur[s] - coarse[s])
              + ", ".join("w%d s%d ncmd %d %.1f-%.1f" % (wave[j] % 4, j, ncmd[j], (start[j] - t0) * us, (end[j] - t0) * us) for j in mates[:8]))
    light = m & (ncmd <= 6)
    wgs = wave // 4
    print("light tiles (<= 6 commands): %d, duration by workgroup index range:" % light.sum())
    for a in range(0, int(wgs.max()) + 1, 128):
        mm = light & (wgs >= a) & (wgs < a + 128)
        if mm.any(): print("   wg %4d-%4d: n %4d mean %.1f p90 %.1f max %.1f" % (a, a + 127, mm.sum(), dur[mm].mean(), np.percentile(dur[mm], 90), dur[mm].max()))
    print("   by wave of the workgroup:", ["%.1f" % dur[light & (wave % 4 == k)].mean() for k in range(4)])
    slow = light & (dur > 15)
    print("   slow ones (> 15 us): %d; their start %.1f..%.1f; list %.1f render %.1f (all light: list %.1f render %.1f)" % (slow.sum(), (start[slow].min() - t0) * us, (start[slow].max() - t0) * us, coarse[slow].mean(), (dur - coarse)[slow].mean(), coarse[light].mean(), (dur - coarse)[light].mean()))

def replay(order, label):
    # the quarter slots (four waves of one workgroup each) first, as the kernel does; then the given order
    free = [0.0] * n_waves
    heapq.heapify(free)
    for d in dur[quarter]:
        heapq.heappush(free, heapq.heappop(free) + d)
    for i in order:
        heapq.heappush(free, heapq.heappop(free) + y[i])
    print("  replay %-32s makespan %.1f us" % (label, max(free)))

print("hand-out replayed with the measured durations (%d one-wave slots on %d waves):" % (m.sum(), n_waves))
replay(np.argsort(start[m], kind="stable"), "in the order the slots started")
replay(np.arange(m.sum()), "in slot (queue) order")
replay(np.argsort(-y), "longest first (bound)")
replay(np.argsort(-est_now, kind="stable"), "by the current estimate, exact")
replay(np.argsort(-pred, kind="stable"), "by the fitted model, exact")
for nq in (8, 16, 32):
    qs = np.quantile(pred, np.linspace(0, 1, nq + 1)[1:-1])
    cls = np.searchsorted(qs, pred)
    replay(np.argsort(-cls, kind="stable"), "fitted model in %d classes" % nq)
