"""Developer analysis (CPU): strip-row slots under alternative chunk sizes / culls / a two-level chunk index."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bin_work import load_workload

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
    CS = [8, 4]       # chunk sizes
    SUP = 8           # chunks per super-chunk
    K = {}
    def acc(key, r, s, v):
        K.setdefault(key, np.zeros((rows, strips_x), np.int64))[r, s] += v
    for it in range(n_items):
        tag = int(tags[it])
        if tag not in (3, 4): continue
        npt = int(items[it, 3]); pix = int(items[it, 4])
        pts = scene[pix: pix + 8 * npt].view(np.float32).reshape(npt, 2).astype(np.float64)
        if tag == 3: a = pts; b = np.roll(pts, -1, axis=0)
        else: a = pts[:-1]; b = pts[1:]
        nseg = len(a)
        if nseg == 0: continue
        xmin = np.minimum(a[:, 0], b[:, 0]); xmax = np.maximum(a[:, 0], b[:, 0])
        ymin = np.minimum(a[:, 1], b[:, 1]); ymax = np.maximum(a[:, 1], b[:, 1])
        hw = 0.0
        if tag == 4: hw = 0.5 * float(items[it, 2:3].view(np.float32)[0]) + 0.5
        boxes = {}
        for cs in CS:
            nch = (nseg + cs - 1) // cs
            pad = nch * cs - nseg
            def padr(v, fill, n=nch, c=cs, p=pad): return np.concatenate([v, np.full(p, fill)]).reshape(n, c)
            cb = (padr(xmin, 1e30).min(1), padr(ymin, 1e30).min(1), padr(xmax, -1e30).max(1), padr(ymax, -1e30).max(1))
            nsu = (nch + SUP - 1) // SUP
            p2 = nsu * SUP - nch
            def pads(v, fill): return np.concatenate([v, np.full(p2, fill)]).reshape(nsu, SUP)
            sb = (pads(cb[0], 1e30).min(1), pads(cb[1], 1e30).min(1), pads(cb[2], -1e30).max(1), pads(cb[3], -1e30).max(1))
            boxes[cs] = (nch, cb, nsu, sb)
        bx, by, bz, bw = bbox[it]
        s_lo, s_hi = max(0, bx // 256), min(strips_x - 1, bz // 256)
        r_lo, r_hi = max(0, by // 16), min(rows - 1, bw // 16)
        for r in range(r_lo, r_hi + 1):
            y0 = 16.0 * r; y1 = y0 + 16.0
            sy0 = 32.0 * (r // 2)
            for s in range(s_lo, s_hi + 1):
                x0 = 256.0 * s; x1 = x0 + 256.0
                acc("cands", r, s, 1)
                for cs in CS:
                    nch, cb, nsu, sb = boxes[cs]
                    def test(bb, tight):
                        if tag == 3:
                            sv = (bb[3] >= y0) & (bb[1] < y1) & (bb[0] < x1)
                            if tight: sv &= (bb[2] > x0) | (bb[1] <= y0)
                            return sv
                        return (bb[3] > sy0 - hw) & (bb[1] < sy0 + 32 + hw) & (bb[2] > x0 - hw) & (bb[0] < x1 + hw)
                    acc(f"c{cs} tests", r, s, nch)
                    acc(f"c{cs} slots loose", r, s, test(cb, False).sum() * cs)
                    acc(f"c{cs} slots tight", r, s, test(cb, True).sum() * cs)
                    ssv = test(sb, True)
                    acc(f"c{cs} super tests", r, s, nsu)
                    acc(f"c{cs} chunk tests after super", r, s, ssv.sum() * SUP)
    act = K["cands"] > 0
    print(name, "active strip rows", act.sum())
    for k in sorted(K):
        v = K[k][act]
        print(f"{k:32s} total {v.sum():9d} mean {v.mean():8.1f} p50 {np.median(v):7.0f} p90 {np.percentile(v, 90):8.0f} p99 {np.percentile(v, 99):8.0f} max {v.max():6d}")

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
