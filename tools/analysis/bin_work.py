"""Developer analysis (CPU, no GPU): how much of pm_bin_kernel's strip-row work is redundant?
Per strip row of a workload: chunk tests, surviving chunks (x8 = slots), and what the slots turn out to be
(fills only; polylines are culled on all four sides already)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pmo

def load_workload(name):
    import piet_metal_amd.workloads as w
    wl = {"config2": lambda: w.tiger(1920, 1080, fills_only=True), "config3": lambda: w.tiger(3840, 2160),
          "config4": w.config4_blobs, "config5": w.config5_tiger_grid}[name]()
    scene, n = pmo.scene_from_paths(pmo.scaled_paths(wl.paths.paths, wl.width_scale), wl.paths.els, wl.affine)
    return wl, scene

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config3"
    wl, scene = load_workload(name)
    W, H = wl.width, wl.height
    u32 = scene[: len(scene) // 4 * 4].view(np.uint32)
    n_items, items_ix = int(u32[0]), int(u32[1])
    bbox = scene[8: 8 + 8 * n_items].view(np.uint16).reshape(n_items, 4).astype(np.int64)
    items = scene[items_ix: items_ix + 32 * n_items].view(np.uint32).reshape(n_items, 8)
    tags = items[:, 0] & 0xffff
    strips_x = (W + 255) // 256
    rows = (H + 15) // 16
    print(name, W, H, "items", n_items, "tags", np.bincount(tags), "strip rows", strips_x * rows)
    T = np.zeros((rows, strips_x, 8), np.int64)  # tests, surv, surv_tight, slots voted, voted with tiles, backdrop-only voted, cand items, segs in strip-x-range
    for it in range(n_items):
        tag = int(tags[it])
        if tag not in (3, 4): continue
        npt = int(items[it, 3]); pix = int(items[it, 4])
        pts = scene[pix: pix + 8 * npt].view(np.float32).reshape(npt, 2).astype(np.float64)
        if tag == 3:
            a = pts; b = np.roll(pts, -1, axis=0)
        else:
            a = pts[:-1]; b = pts[1:]
        nseg = len(a)
        if nseg == 0: continue
        xmin = np.minimum(a[:, 0], b[:, 0]); xmax = np.maximum(a[:, 0], b[:, 0])
        ymin = np.minimum(a[:, 1], b[:, 1]); ymax = np.maximum(a[:, 1], b[:, 1])
        nch = (nseg + 7) // 8
        pad = nch * 8 - nseg
        def padr(v, fill): return np.concatenate([v, np.full(pad, fill)]).reshape(nch, 8)
        cx0 = padr(xmin, 1e30).min(1); cx1 = padr(xmax, -1e30).max(1)
        cy0 = padr(ymin, 1e30).min(1); cy1 = padr(ymax, -1e30).max(1)
        bx, by, bz, bw = bbox[it]
        s_lo, s_hi = max(0, bx // 256), min(strips_x - 1, bz // 256)
        r_lo, r_hi = max(0, by // 16), min(rows - 1, bw // 16)
        hw = 0.0
        if tag == 4: hw = 0.5 * float(items[it, 2:3].view(np.float32)[0]) + 0.5
        for r in range(r_lo, r_hi + 1):
            y0 = 16.0 * r; y1 = y0 + 16.0
            for s in range(s_lo, s_hi + 1):
                x0 = 256.0 * s; x1 = x0 + 256.0
                T[r, s, 6] += 1
                T[r, s, 0] += nch
                if tag == 3:
                    sv = (cy1 >= y0) & (cy0 < y1) & (cx0 < x1)
                    svt = sv & ((cx1 > x0) | (cy0 <= y0))
                    T[r, s, 1] += sv.sum(); T[r, s, 2] += svt.sum()
                    segm = np.repeat(sv, 8)[:nseg]
                    pre = (ymax >= y0) & (ymin < y1) & (xmin < x1) & segm
                    cross = pre & (ymin <= y0)            # can carry backdrop
                    instrip = pre & (xmax > x0)           # can have tiles
                    T[r, s, 3] += (cross | instrip).sum()  # (upper bound of voted)
                    T[r, s, 4] += instrip.sum()
                    T[r, s, 5] += (cross & ~instrip).sum()
                else:
                    sy0 = 32.0 * (r // 2)
                    sv = (cy1 > sy0 - hw) & (cy0 < sy0 + 32 + hw) & (cx1 > x0 - hw) & (cx0 < x1 + hw)
                    T[r, s, 1] += sv.sum(); T[r, s, 2] += sv.sum()
                    segm = np.repeat(sv, 8)[:nseg]
                    pre = (ymax > y0 - hw) & (ymin < y1 + hw) & (xmax > x0 - hw) & (xmin < x1 + hw) & segm
                    T[r, s, 3] += pre.sum(); T[r, s, 4] += pre.sum()
    act = T[:, :, 6] > 0
    print("active strip rows", act.sum())
    names = ["chunk tests", "surviving chunks", "surviving (tight cull)", "slots that can vote", "  of them with tiles in strip", "  backdrop-only (left of strip)", "candidates"]
    for k, nm in enumerate(names):
        v = T[:, :, k][act]
        print(f"{nm:34s} total {v.sum():9d} mean {v.mean():8.1f} p90 {np.percentile(v, 90):8.0f} max {v.max():6d}")
    slots = T[:, :, 1] * 8
    order = np.argsort(-slots.ravel())[:12]
    for o in order:
        r, s = divmod(o, strips_x)
        print(f"  row {r} strip {s}: tests {T[r,s,0]} surv {T[r,s,1]} (slots {T[r,s,1]*8}) tight {T[r,s,2]} (slots {T[r,s,2]*8}) in-strip segs {T[r,s,4]} bd-only {T[r,s,5]} cands {T[r,s,6]}")
    # per tile row totals: what a whole-row workgroup would see
    rowsum = T.sum(axis=1)
    print("per tile ROW: in-strip segs mean", rowsum[:, 4].mean(), "max", rowsum[:, 4].max())

if __name__ == "__main__":
    main()
