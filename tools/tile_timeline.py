"""Developer profiling: where does pm_fine_kernel's time go at Tiger 4K?"""
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
if len(sys.argv) > 2:  # only a band of tile rows: the tiles of the band without the rest of the frame around them
    r.set_band(int(sys.argv[1]), int(sys.argv[2]))
for _ in range(6):  # (past the frame at which the plan is remade from the frames' own report)
    r.render()
    if os.environ.get("PM_TIMELINE_LONE", "1") != "0":
        r.sync()  # every frame alone: the hand-out and the class thresholds of a lone frame
r.sync()
t = r.time_tiles()
t = t[t[:, 0] > 0]  # (rows of slots the replay did not reach stay zero)
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
tile = (t[:, 2] & 0x7fffffff).astype(np.int64); quarter = (t[:, 2] >> 31).astype(bool)
wave = (t[:, 3] >> 32).astype(np.int64); ncmd = (t[:, 3] & 0xffffffff).astype(np.int64)
t0 = start.min(); us = 1e-2  # 100 MHz wall clock -> 10 ns ticks
dur = (end - start) * us
print(f"slots {len(t)}  quarter slots {quarter.sum()}  kernel span {(end.max()-t0)*us:.1f} us")
print(f"slot duration us: mean {dur.mean():.2f} median {np.median(dur):.2f} p90 {np.percentile(dur,90):.2f} p99 {np.percentile(dur,99):.2f} max {dur.max():.2f}")
print(f"sum of slot durations {dur.sum():.0f} us over {len(np.unique(wave))} waves -> {dur.sum()/len(np.unique(wave)):.1f} us per wave")
# per-wave finish time
fin = {}
for w, e in zip(wave, end): fin[w] = max(fin.get(w, 0), e)
f = (np.array(list(fin.values())) - t0) * us
print(f"wave finish time us: mean {f.mean():.1f} p50 {np.median(f):.1f} p90 {np.percentile(f,90):.1f} max {f.max():.1f}")
print(f"first-slot start spread us: {((start[np.argsort(start)][:3000]-t0)*us).max():.1f}")
for lo, hi in [(0,0),(1,4),(5,16),(17,48),(49,100),(101,1000)]:
    m = (ncmd >= lo) & (ncmd <= hi)
    if m.any(): print(f"ncmd {lo:3d}-{hi:4d}: {m.sum():6d} slots, mean dur {dur[m].mean():6.2f} us, total {dur[m].sum():8.0f} us, us/cmd {dur[m].sum()/max(1,ncmd[m].sum()):.3f}")
coarse = np.where(t[:, 6] > 0, (t[:, 6].astype(np.int64) - start) * us, 0.0)
if (t[:, 6] > 0).any():
    print(f"fused: list building per slot us: mean {coarse.mean():.2f} p90 {np.percentile(coarse,90):.2f} max {coarse.max():.2f}, total {coarse.sum():.0f} us of {dur.sum():.0f}")
worst = np.argsort(-dur)[:8]
for i in worst: print(f"  worst slot {i} tile {tile[i]} q={quarter[i]} ncmd {ncmd[i]} dur {dur[i]:.1f} us start {(start[i]-t0)*us:.1f}  phase A {t[i,4]*us:.1f} us  phase B {t[i,5]*us:.1f} us  list {coarse[i]:.1f} us  own items {t[i,7]*us:.1f} us")
q = quarter & (ncmd > 0)
if q.any(): print(f"workgroup-mode slots: {q.sum()}, mean dur {dur[q].mean():.2f} us, phase A {t[q,4].mean()*us:.2f} us, phase B {t[q,5].mean()*us:.2f} us, other {(dur[q]-(t[q,4]+t[q,5])*us).mean():.2f} us")

if t.shape[1] >= 12 and (t[:, 11] > 0).any():
    lo = lambda c: (t[:, c] & 0xffffffff).astype(np.float64) * us
    hi = lambda c: (t[:, c] >> 32).astype(np.float64) * us
    stages = {"header+hits": lo(8), "candidates": hi(8), "owners": lo(9), "scan": hi(9), "segments": lo(10), "emission": hi(10)}
    rounds = (t[:, 11] & 0xffffffff).astype(np.int64); records = (t[:, 11] >> 32).astype(np.int64)
    built = records > 0
    print(f"list building by stage (tiles whose wave built a list: {built.sum()}; records/tile {records[built].mean():.2f}, rounds/tile {rounds[built].mean():.2f}):")
    for name, v in stages.items():
        print(f"   {name:12s} mean {v[built].mean():5.2f} us   p90 {np.percentile(v[built], 90):5.2f}   max {v[built].max():5.2f}")
    tot = sum(stages.values())
    print(f"   sum of stages mean {tot[built].mean():.2f} us (list building total mean {coarse[built].mean():.2f})")
    w = np.argsort(-coarse)[:3]
    for i in w:
        print(f"   slot {i} ncmd {ncmd[i]} list {coarse[i]:.1f} us: " + ", ".join(f"{k} {v[i]:.1f}" for k, v in stages.items()) + f", rounds {rounds[i]}, records {records[i]}")

last = np.argsort(-end)[:10]
print("slots that end last:")
for i in last: print(f"  last slot {i} tile {tile[i]} q={quarter[i]} ncmd {ncmd[i]} start {(start[i]-t0)*us:.1f} end {(end[i]-t0)*us:.1f} dur {dur[i]:.1f} list {coarse[i]:.1f}")

# first tile of a wave vs. the tiles after it: what the previous tile's pixel stores (in flight when the next tile's piece is
# requested) cost the next tile, and the gap between a wave's tiles
if t.shape[1] >= 12 and (t[:, 11] > 0).any():
    order = np.lexsort((start, wave))
    first = np.zeros(len(t), bool)
    prev_end = {}
    gap = np.zeros(len(t))
    for i in order:
        w = wave[i]
        if w not in prev_end: first[i] = True
        else: gap[i] = (start[i] - prev_end[w]) * us
        prev_end[w] = end[i]
    sw = built & ~quarter
    for name, m in (("first tile of its wave", sw & first), ("later tiles", sw & ~first)):
        if m.any():
            print(f"{name}: {m.sum()} slots, list building mean {coarse[m].mean():.2f} us (candidates stage {stages['candidates'][m].mean():.2f}), duration mean {dur[m].mean():.2f}, mean ncmd {ncmd[m].mean():.1f}, gap before it {gap[m].mean():.2f} us")
