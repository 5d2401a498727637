"""Developer profiling: where does pm_bin_kernel's time go at Tiger 4K?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
wl = {"config2": lambda: pm.workloads.tiger(1920, 1080, fills_only=True), "config3": lambda: pm.workloads.tiger(3840, 2160),
      "config4": pm.workloads.config4_blobs, "config5": pm.workloads.config5_tiger_grid}[os.environ.get("PM_TL_WORKLOAD", "config3")]()
print("workload", wl.name, wl.width, wl.height)
r = pm.Renderer(0)
r.resize(wl.width, wl.height)
r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
for _ in range(6): r.render()  # (past the frame at which the plan is remade from the frames' own report)
r.sync()
t = r.time_bins().astype(np.int64)
t = t[t[:, 0] > 0]  # strip rows no item reaches never get a workgroup
us = 1e-2
t0 = t[:, 0].min()
end = np.maximum(t[:, 7], t[:, 14]) if t.shape[1] >= 16 else t[:, 7]  # (the tail wave may outlive wave 0)
dur = (end - t[:, 0]) * us
print(f"WGs {len(t)} kernel span {(end.max()-t0)*us:.1f} us; WG duration mean {dur.mean():.2f} p50 {np.median(dur):.2f} p90 {np.percentile(dur,90):.2f} max {dur.max():.2f}")
print(f"start spread: last WG starts at {(t[:,0].max()-t0)*us:.1f} us")
has = t[:, 1] > 0
print(f"WGs with a record: {has.sum()}")
def ph(a, b, m): 
    d = (t[m, b] - t[m, a]) * us
    return f"mean {d.mean():.2f} p90 {np.percentile(d,90):.2f} max {d.max():.2f}"
m = has
print("item scan      :", ph(0, 1, m))
print("headers+scan   :", ph(1, 2, m))
print("segment stream :", ph(2, 3, m), "(last record)")
print("finalise       :", ph(3, 4, m))
if t.shape[1] >= 16 and (t[:, 12] > 0).any():
    print("   candidates pass (backdrops, hit bits, ballots) :", ph(3, 12, m))
    print("   entries + scatter (wave 0)                     :", ph(12, 4, m))
    if (t[:, 13] > 0).any():
        m13 = m & (t[:, 13] > 0)
        print("      entries :", ph(12, 13, m13), "| scatter :", ph(13, 4, m13))
print("wave 0 exit     :", ph(4, 7, m))
if t.shape[1] >= 16 and (t[:, 14] > 0).any():
    print("tail wave ends after wave 0's scatter by:", ph(4, 14, m))
    pass
e = ~has
d = (t[e, 7] - t[e, 0]) * us
if e.any(): print(f"WGs without candidates: {e.sum()}, duration mean {d.mean():.2f} max {d.max():.2f}")
ch = t[:, 6]
worst = np.argsort(-dur)[:6]
for i in worst: print(f"  WG {i} chunks {ch[i]} dur {dur[i]:.1f} start {(t[i,0]-t0)*us:.1f} phases {[(t[i,k+1]-t[i,k])*us if t[i,k+1]>0 and t[i,k]>0 else None for k in (0,1,2,3,4)]}")
print("total chunks", ch.sum())
print(f"sum of WG durations {dur.sum():.0f} us -> {dur.sum()/1024:.1f} us of the machine at 1024 resident WGs")
for lo, hi in [(0, 8), (8, 12), (12, 20), (20, 30), (30, 45), (45, 1000)]:
    m2 = (dur >= lo) & (dur < hi)
    print(f"  WGs with {lo}-{hi} us: {m2.sum()} -> {dur[m2].sum():.0f} us total")
print("round 0 of the segment stream (WGs with >= 512 elements in it):")
big = (t[:, 6] >= 512) & (t[:, 10] > 0)
if not big.any(): big = (t[:, 6] >= 192) & (t[:, 10] > 0)
for nm, a, b in [("headers done -> round start", 2, 8), ("chunk tests + scan (round 0)", 8, 9), ("other rounds + survivor list + barrier", 9, 10), ("votes (all rounds)", 10, 3)]:
    d = (t[big, b] - t[big, a]) * us
    print(f"   {nm:28s} mean {d.mean():.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f}")
el = t[big, 6]
d = (t[big, 3] - t[big, 10]) * us
print(f"   {big.sum()} WGs, elements mean {el.mean():.0f}; expansion per 256-element step: {(d / np.ceil(el / 256)).mean():.2f} us")
# finer picture of the rows that set the span: durations in 1 us bins, end times, and what the longest rows have in common
hist, edges = np.histogram(dur, bins=np.arange(0, 31, 1.0))
print("WG duration histogram (1 us bins):", " ".join(f"{int(e)}:{h}" for e, h in zip(edges[:-1], hist) if h))
endt = (end - t0) * us
hist, edges = np.histogram(endt, bins=np.arange(0, 41, 1.0))
print("WG end-time histogram (1 us bins):", " ".join(f"{int(e)}:{h}" for e, h in zip(edges[:-1], hist) if h))
for q in (50, 75, 90, 95, 98, 99, 100):
    print(f"   p{q}: dur {np.percentile(dur, q):.1f} end {np.percentile(endt, q):.1f}")
sl = t[:, 6]
for lo, hi in [(0, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 1 << 30)]:
    m2 = (sl >= lo) & (sl < hi)
    if m2.any(): print(f"  rows with {lo}-{hi} slots: {m2.sum()}, dur mean {dur[m2].mean():.2f} max {dur[m2].max():.2f}; stream {((t[m2,3]-t[m2,2])*us).mean():.2f} finalise {((t[m2,4]-t[m2,3])*us).mean():.2f}")

# the slowest rows' segment stream taken apart (first record): super-chunk tests, chunk tests + survivor list, votes
slow = np.argsort(-dur)[:12]
for i in slow:
    if t[i, 10] > 0 and t[i, 8] > 0:
        print(f"  slow WG {i}: slots {t[i,6]} dur {dur[i]:.1f} | to round 0 {(t[i,8]-t[i,2])*us:.2f} sup tests {(t[i,9]-t[i,8])*us:.2f} chunk rounds+list {(t[i,10]-t[i,9])*us:.2f} votes {(t[i,3]-t[i,10])*us:.2f} | cand pass {(t[i,12]-t[i,3])*us:.2f} entries {(t[i,13]-t[i,12])*us:.2f} scatter {(t[i,4]-t[i,13])*us:.2f} tail after {(end[i]-t[i,4])*us:.2f}")
m2 = (t[:, 10] > 0) & (t[:, 8] > 0)
for nm, a, b in (("to round 0", 2, 8), ("sup tests", 8, 9), ("chunk rounds + list", 9, 10), ("votes", 10, 3)):
    d = (t[m2, b] - t[m2, a]) * us
    print(f"  all rows: {nm:20s} mean {d.mean():.2f} p90 {np.percentile(d,90):.2f} max {d.max():.2f}")
