"""Developer profiling: pm_bin_kernel strip rows by their position in a workgroup's chain (first = cold instruction cache)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm
wl = {"config2": lambda: pm.workloads.tiger(1920, 1080, fills_only=True), "config3": lambda: pm.workloads.tiger(3840, 2160),
      "config4": pm.workloads.config4_blobs, "config5": pm.workloads.config5_tiger_grid}[os.environ.get("PM_TL_WORKLOAD", "config3")]()
per_cu = int(os.environ.get("PM_BIN_WG_PER_CU", "5"))
r = pm.Renderer(0)
r.resize(wl.width, wl.height)
r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
for _ in range(3): r.render()
r.sync()
t = r.time_bins().astype(np.int64)
act = np.nonzero(t[:, 0] > 0)[0]
t = t[act]
us = 1e-2
grid = min(len(t), 256 * per_cu) if per_cu else len(t)
pos = np.arange(len(t)) // grid
end = np.maximum(t[:, 7], t[:, 14])
dur = (end - t[:, 0]) * us
slots = t[:, 6]
print(f"{wl.name} per_cu {per_cu} rows {len(t)} grid {grid} span {(end.max() - t[:, 0].min()) * us:.1f} us")
for p in range(min(pos.max() + 1, 6)):
    m = pos == p
    ph = lambda a, b: ((t[m, b] - t[m, a]) * us).mean()
    print(f"  chain position {p}: rows {m.sum():5d} slots mean {slots[m].mean():6.0f} dur mean {dur[m].mean():6.2f} p50 {np.median(dur[m]):6.2f} | scan {ph(0,1):.2f} hdr {ph(1,2):.2f} stream {ph(2,3):.2f} cand {ph(3,12):.2f} scatter {ph(12,4):.2f}")
# light rows only, by position (same work, cold vs warm)
for p in range(min(pos.max() + 1, 4)):
    m = (pos == p) & (slots <= 128)
    if m.sum(): print(f"  rows with <= 128 slots at position {p}: {m.sum():5d} dur mean {dur[m].mean():6.2f}")
