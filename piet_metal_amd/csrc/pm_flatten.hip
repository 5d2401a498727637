// On-device flatten + encode: src/flatten.rs:10-47 and make_tiger's two passes
// (src/lib.rs:293-367) moved onto the GPU.  Input: SVG-style path elements in
// f64 (what kurbo::BezPath holds), output: the piet-metal scene buffer, resident
// in HBM, byte-identical to what the reference's CPU encoder would write.
//
// The reference flattens every path 3-4 times on the CPU (count pass + encode
// pass, fill + stroke, src/lib.rs:293-326).  Here:
//   K_count   one thread per element: subdivision count n of each cubic
//             (kurbo CubicBez::to_quads rule, see oracle/pmo_flatten.c header)
//   K_scan    one workgroup: exclusive scans over elements (points, sub-paths)
//             and over paths (item / point bases) -> exact output layout
//   K_points  one thread per element: evaluates the cubic at (k+1)/n in f64,
//             rounds to f32, writes the fill copy and the stroke copy of the
//             points, keeps the element's f64 bounding box
//   K_items   one thread per sub-path: unions the element boxes, writes the
//             ShortBbox + PietFill / PietStrokePolyLine records (thin-line rule
//             of src/lib.rs:353-362 included)
// f64 arithmetic is kept (gfx950 has full-rate f64 FMA pipes; 2k cubics is
// nothing) so that the bytes match the CPU path exactly.  -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/piet_metal_amd.h"
#include "pm_flatten.h"
#include "pm_layout.h"

namespace pm {

namespace {

constexpr double kTolerance = 0.1;  // src/lib.rs:330
constexpr float kThinLine = 0.7f;   // src/lib.rs:351

struct Affine {
    double m[6];
};

__device__ __forceinline__ void Xform(const Affine &a, double x, double y, double *ox, double *oy) {
    // kurbo Affine * Point
    *ox = a.m[0] * x + a.m[2] * y + a.m[4];
    *oy = a.m[1] * x + a.m[3] * y + a.m[5];
}

// End point of the nearest earlier Move/Line/Curve element of the same path
// (flatten.rs keeps last_pt only across those; QuadTo / ClosePath fall to `_ => ()`).
__device__ bool LastPoint(const pm_path_el *els, uint32_t el_begin, uint32_t i, const Affine &a, double *lx, double *ly) {
    for (uint32_t j = i; j > el_begin;) {
        --j;
        const uint32_t t = els[j].tag;
        if (t == PM_EL_MOVE || t == PM_EL_LINE) {
            Xform(a, els[j].p[0], els[j].p[1], lx, ly);
            return true;
        }
        if (t == PM_EL_CURVE) {
            Xform(a, els[j].p[4], els[j].p[5], lx, ly);
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ uint32_t PathOf(const pm_path *paths, uint32_t n_paths, uint32_t el) {
    uint32_t lo = 0, hi = n_paths;  // paths are consecutive element ranges
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (paths[mid].el_begin <= el) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ double CubicEval(double p0, double p1, double p2, double p3, double t) {
    const double mt = 1.0 - t;  // kurbo CubicBez::eval
    return p0 * (mt * mt * mt) + (p1 * (mt * mt * 3.0) + (p2 * (mt * 3.0) + p3 * t) * t) * t;
}

// kurbo to_quads: n = ceil(x^(1/6)), at least 1 -- defined without libm so that host, device and
// reference cannot disagree in the last ulp of pow(): the smallest n >= 1 with n^6 >= x,
// n^6 = ((n*n)*(n*n))*(n*n) in binary64.  pow() only supplies the starting guess.
__device__ __forceinline__ uint32_t SubdivisionCount(double x) {
    if (!(x > 1.0)) return 1u;
    if (x > 1e54) return 1u << 30;  // (no viewport-sized path gets here; keeps the arithmetic below in range)
    const double g = ceil(pow(x, 1.0 / 6.0));
    unsigned long long n = g >= 1.0 ? static_cast<unsigned long long>(g) : 1ull;
    auto p6 = [](unsigned long long v) {
        const double d = static_cast<double>(v);
        return ((d * d) * (d * d)) * (d * d);
    };
    while (n > 1 && p6(n - 1) >= x) --n;
    while (p6(n) < x) ++n;
    return static_cast<uint32_t>(n);
}

__global__ void KCount(const pm_path *paths, uint32_t n_paths, const pm_path_el *els, uint32_t n_els, Affine aff,
                       uint32_t *el_npts, uint32_t *el_move, uint32_t *err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_els) return;
    const uint32_t p = PathOf(paths, n_paths, i);
    const uint32_t tag = els[i].tag;
    uint32_t n = 0, mv = 0;
    if (i >= paths[p].el_begin && i < paths[p].el_end) {
        if (tag == PM_EL_MOVE) {
            n = 1;
            mv = 1;
        } else if (tag == PM_EL_LINE || tag == PM_EL_CURVE) {
            double lx, ly;
            // a sub-path must have been opened (cur_path.as_mut().unwrap(), flatten.rs:24,:36)
            // (a path that starts with a MoveTo -- every well-formed one -- answers with one load; walking back
            //  to the sub-path's MoveTo is a chain of dependent loads as long as the sub-path: 32 us for the
            //  Tiger's longest, the whole kernel's duration)
            bool opened = i > paths[p].el_begin && els[paths[p].el_begin].tag == PM_EL_MOVE;
            if (!opened) {
                for (uint32_t j = i; j > paths[p].el_begin;) {
                    --j;
                    if (els[j].tag == PM_EL_MOVE) { opened = true; break; }
                }
            }
            if (!opened) {
                atomicExch(err, 1u);
            } else if (tag == PM_EL_LINE) {
                n = 1;
            } else {
                LastPoint(els, paths[p].el_begin, i, aff, &lx, &ly);
                double p1x, p1y, p2x, p2y, p3x, p3y;
                Xform(aff, els[i].p[0], els[i].p[1], &p1x, &p1y);
                Xform(aff, els[i].p[2], els[i].p[3], &p2x, &p2y);
                Xform(aff, els[i].p[4], els[i].p[5], &p3x, &p3y);
                const double accuracy = kTolerance * 1e-2;  // flatten.rs:35
                const double max_hypot2 = 432.0 * accuracy * accuracy;
                const double ax = p1x * 3.0 - lx, ay = p1y * 3.0 - ly;
                const double bx = p2x * 3.0 - p3x, by = p2y * 3.0 - p3y;
                const double dx = bx - ax, dy = by - ay;
                const double e = dx * dx + dy * dy;
                n = SubdivisionCount(e / max_hypot2);
            }
        }
    }
    el_npts[i] = n;
    el_move[i] = mv;
}

// One workgroup.  Exclusive scans with totals at index n.
constexpr int kScanThreads = 1024;
constexpr size_t kMaxFlattenEls = (1u << 26) - 1u;  // elements / paths of one pm_flatten_and_encode call

__device__ uint32_t BlockScan1024(uint32_t v, uint32_t *s_w, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= static_cast<uint32_t>(d)) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
        const uint32_t x = s_w[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void KScan(const pm_path *paths, uint32_t n_paths, uint32_t n_els,
                                                      const uint32_t *el_npts, const uint32_t *el_move,
                                                      uint32_t *el_ptoff, uint32_t *el_mvoff,
                                                      uint32_t *path_item_base, uint32_t *path_pt_base,
                                                      uint32_t *totals) {
    __shared__ uint32_t s_w[kScanThreads / 64];
    uint32_t carry_p = 0, carry_m = 0;
    for (uint32_t base = 0; base < n_els; base += kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t vp = (i < n_els) ? el_npts[i] : 0u;
        const uint32_t vm = (i < n_els) ? el_move[i] : 0u;
        uint32_t tp, tm;
        const uint32_t op = BlockScan1024(vp, s_w, &tp);
        const uint32_t om = BlockScan1024(vm, s_w, &tm);
        if (i < n_els) {
            el_ptoff[i] = carry_p + op;
            el_mvoff[i] = carry_m + om;
        }
        carry_p += tp;
        carry_m += tm;
    }
    if (threadIdx.x == 0) {
        el_ptoff[n_els] = carry_p;
        el_mvoff[n_els] = carry_m;
    }
    __syncthreads();
    __threadfence_block();
    uint32_t carry_i = 0, carry_q = 0;
    for (uint32_t base = 0; base < n_paths; base += kScanThreads) {
        const uint32_t p = base + threadIdx.x;
        uint32_t vi = 0, vq = 0;
        if (p < n_paths) {
            const uint32_t n_sub = el_mvoff[paths[p].el_end] - el_mvoff[paths[p].el_begin];
            const uint32_t n_pts = el_ptoff[paths[p].el_end] - el_ptoff[paths[p].el_begin];
            if (paths[p].flags & PM_PATH_FILL) {
                // a compound fill is ONE item; its point array also holds a separator per sub-path
                const bool compound = (paths[p].flags & PM_PATH_COMPOUND) != 0;
                vi += compound ? (n_sub ? 1u : 0u) : n_sub;
                vq += n_pts + (compound ? n_sub : 0u);
            }
            if (paths[p].flags & PM_PATH_STROKE) {
                vi += n_sub;
                vq += n_pts;
            }
        }
        uint32_t ti, tq;
        const uint32_t oi = BlockScan1024(vi, s_w, &ti);
        const uint32_t oq = BlockScan1024(vq, s_w, &tq);
        if (p < n_paths) {
            path_item_base[p] = carry_i + oi;
            path_pt_base[p] = carry_q + oq;
        }
        carry_i += ti;
        carry_q += tq;
    }
    if (threadIdx.x == 0) {
        totals[0] = carry_i;  // n_items
        totals[1] = carry_q;  // encoded points (fill + stroke copies)
        totals[2] = carry_m;  // sub-paths
    }
}

// ---- the same two prefix sums for LARGE inputs: blocks of 1 024 in parallel (round 5) ------------------------------------
// One workgroup walking 60 000 elements is 60 dependent steps of two block scans each: 125 us of config 4's 270 us view change,
// 250 us for 110 000 elements.  Above kScanSplit elements the work is cut into blocks: every block scans its 1 024 values (KScanA,
// KScanD), ONE workgroup scans the blocks' totals (KScanTops), and a last pass adds each block's base (KScanApply).  Five small
// launches instead of one long one; below the threshold the single workgroup is quicker (the Tiger's 2 500 elements: 3 us).
constexpr uint32_t kScanSplit = 16384;
// (PM_SCAN_SPLIT=<elements> moves the threshold: the tests put small scenes through the block-parallel sums with it)
static uint32_t ScanSplit() {  // (read per scene, not cached: a flatten is not a per-frame call)
    const char *e = getenv("PM_SCAN_SPLIT");
    return e && *e ? static_cast<uint32_t>(strtoul(e, nullptr, 10)) : kScanSplit;
}

__global__ __launch_bounds__(kScanThreads) void KScanA(uint32_t n_els, const uint32_t *el_npts, const uint32_t *el_move, uint32_t *el_ptoff,
                                                       uint32_t *el_mvoff, uint32_t *tops) {
    __shared__ uint32_t s_w[kScanThreads / 64];
    const uint32_t i = blockIdx.x * kScanThreads + threadIdx.x;
    const uint32_t vp = (i < n_els) ? el_npts[i] : 0u;
    const uint32_t vm = (i < n_els) ? el_move[i] : 0u;
    uint32_t tp, tm;
    const uint32_t op = BlockScan1024(vp, s_w, &tp);
    const uint32_t om = BlockScan1024(vm, s_w, &tm);
    if (i < n_els) {  // (offsets inside the block: KScanApply adds the block's base)
        el_ptoff[i] = op;
        el_mvoff[i] = om;
    }
    if (threadIdx.x == 0) {
        tops[2u * blockIdx.x] = tp;
        tops[2u * blockIdx.x + 1u] = tm;
    }
}

// tops[2 b], tops[2 b + 1] (two interleaved arrays of n_blocks totals) -> exclusive prefix sums in place; the grand totals to out0 / out1
__global__ __launch_bounds__(kScanThreads) void KScanTops(uint32_t n_blocks, uint32_t *tops, uint32_t *out0, uint32_t *out1, uint32_t *out2_copy_of_1) {
    __shared__ uint32_t s_w[kScanThreads / 64];
    uint32_t carry_a = 0, carry_b = 0;
    for (uint32_t base = 0; base < n_blocks; base += kScanThreads) {
        const uint32_t b = base + threadIdx.x;
        const uint32_t va = (b < n_blocks) ? tops[2u * b] : 0u;
        const uint32_t vb = (b < n_blocks) ? tops[2u * b + 1u] : 0u;
        uint32_t ta, tb;
        const uint32_t oa = BlockScan1024(va, s_w, &ta);
        const uint32_t ob = BlockScan1024(vb, s_w, &tb);
        if (b < n_blocks) {
            tops[2u * b] = carry_a + oa;
            tops[2u * b + 1u] = carry_b + ob;
        }
        carry_a += ta;
        carry_b += tb;
    }
    if (threadIdx.x == 0) {
        *out0 = carry_a;
        *out1 = carry_b;
        if (out2_copy_of_1) *out2_copy_of_1 = carry_b;
    }
}

// per path: items and encoded points (KScan's second loop), scanned inside blocks of 1 024 paths; the element offsets are read as
// "offset inside the block + the block's base" (KScanApply has not run yet: it must not, other blocks still read the unapplied values)
__global__ __launch_bounds__(kScanThreads) void KScanD(const pm_path *paths, uint32_t n_paths, uint32_t n_els, const uint32_t *el_ptoff,
                                                       const uint32_t *el_mvoff, const uint32_t *tops_e, uint32_t *path_item_base,
                                                       uint32_t *path_pt_base, uint32_t *tops_p) {
    __shared__ uint32_t s_w[kScanThreads / 64];
    const uint32_t p = blockIdx.x * kScanThreads + threadIdx.x;
    auto ptoff = [&](uint32_t i) { return i == n_els ? el_ptoff[n_els] : el_ptoff[i] + tops_e[2u * (i / kScanThreads)]; };
    auto mvoff = [&](uint32_t i) { return i == n_els ? el_mvoff[n_els] : el_mvoff[i] + tops_e[2u * (i / kScanThreads) + 1u]; };
    uint32_t vi = 0, vq = 0;
    if (p < n_paths) {
        const uint32_t n_sub = mvoff(paths[p].el_end) - mvoff(paths[p].el_begin);
        const uint32_t n_pts = ptoff(paths[p].el_end) - ptoff(paths[p].el_begin);
        if (paths[p].flags & PM_PATH_FILL) {
            const bool compound = (paths[p].flags & PM_PATH_COMPOUND) != 0;
            vi += compound ? (n_sub ? 1u : 0u) : n_sub;
            vq += n_pts + (compound ? n_sub : 0u);
        }
        if (paths[p].flags & PM_PATH_STROKE) {
            vi += n_sub;
            vq += n_pts;
        }
    }
    uint32_t ti, tq;
    const uint32_t oi = BlockScan1024(vi, s_w, &ti);
    const uint32_t oq = BlockScan1024(vq, s_w, &tq);
    if (p < n_paths) {
        path_item_base[p] = oi;
        path_pt_base[p] = oq;
    }
    if (threadIdx.x == 0) {
        tops_p[2u * blockIdx.x] = ti;
        tops_p[2u * blockIdx.x + 1u] = tq;
    }
}

__global__ __launch_bounds__(kScanThreads) void KScanApply(uint32_t n_els, uint32_t *el_ptoff, uint32_t *el_mvoff, const uint32_t *tops_e, uint32_t n_paths,
                                                           uint32_t *path_item_base, uint32_t *path_pt_base, const uint32_t *tops_p) {
    const uint32_t i = blockIdx.x * kScanThreads + threadIdx.x;
    if (i < n_els) {
        el_ptoff[i] += tops_e[2u * blockIdx.x];
        el_mvoff[i] += tops_e[2u * blockIdx.x + 1u];
    }
    if (i < n_paths) {
        path_item_base[i] += tops_p[2u * blockIdx.x];
        path_pt_base[i] += tops_p[2u * blockIdx.x + 1u];
    }
}

// (n_items and the other totals are read from device memory: the host does not wait for KScan before
//  it launches the kernels that depend on them)
__global__ void KPoints(const pm_path *paths, uint32_t n_paths, const pm_path_el *els, uint32_t n_els, Affine aff,
                        const uint32_t *el_npts, const uint32_t *el_ptoff, const uint32_t *el_mvoff,
                        const uint32_t *path_pt_base, const uint32_t *totals, uint8_t *scene, uint32_t scene_cap,
                        double *el_bbox, uint32_t *sub_first_el) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_els) return;
    const uint32_t n_items = totals[0];
    const uint32_t n = el_npts[i];
    if (n == 0) return;
    const uint32_t p = PathOf(paths, n_paths, i);
    const pm_path path = paths[p];
    const uint32_t tag = els[i].tag;
    if (tag == PM_EL_MOVE) sub_first_el[el_mvoff[i]] = i;
    const uint32_t path_pts = el_ptoff[path.el_end] - el_ptoff[path.el_begin];
    const uint32_t local = el_ptoff[i] - el_ptoff[path.el_begin];
    const size_t points_start = sizeof(SimpleGroup) + static_cast<size_t>(n_items) * (sizeof(ShortBbox) + kItemSize);
    const bool has_fill = (path.flags & PM_PATH_FILL) != 0;
    const bool has_stroke = (path.flags & PM_PATH_STROKE) != 0;
    // compound fill: the fill copy has a separator after every sub-path, so this element's points
    // sit `sub` entries further (sub = index of its sub-path), and the stroke copy starts n_sub later
    const bool compound = has_fill && (path.flags & PM_PATH_COMPOUND) != 0;
    const uint32_t n_sub_path = el_mvoff[path.el_end] - el_mvoff[path.el_begin];
    const uint32_t sub = el_mvoff[i] + (tag == PM_EL_MOVE ? 1u : 0u) - el_mvoff[path.el_begin] - 1u;  // (an element before the first MoveTo was rejected by KCount)
    const size_t base = points_start + 8 * static_cast<size_t>(path_pt_base[p]);
    const size_t dst0 = base + 8 * (static_cast<size_t>(local) + (compound ? sub : 0u));
    const size_t dst1 = base + 8 * (static_cast<size_t>(path_pts) + (compound ? n_sub_path : 0u) + local);  // stroke copy when a fill copy exists
    double bx0, by0, bx1, by1;
    auto emit = [&](uint32_t k, double x, double y) {
        if (k == 0) {
            bx0 = bx1 = x;  // Rect::from_points(pt, pt)
            by0 = by1 = y;
        } else {
            bx0 = fmin(bx0, x); by0 = fmin(by0, y);  // Rect::union_pt
            bx1 = fmax(bx1, x); by1 = fmax(by1, y);
        }
        const float2 f = make_float2(static_cast<float>(x), static_cast<float>(y));  // point_to_f32s
        const size_t o = static_cast<size_t>(k) * 8;
        if (has_fill) {
            if (dst0 + o + 8 <= scene_cap) *reinterpret_cast<float2 *>(scene + dst0 + o) = f;
            if (has_stroke && dst1 + o + 8 <= scene_cap) *reinterpret_cast<float2 *>(scene + dst1 + o) = f;
        } else if (has_stroke) {
            if (dst0 + o + 8 <= scene_cap) *reinterpret_cast<float2 *>(scene + dst0 + o) = f;
        }
    };
    if (tag == PM_EL_MOVE || tag == PM_EL_LINE) {
        double x, y;
        Xform(aff, els[i].p[0], els[i].p[1], &x, &y);
        emit(0, x, y);
    } else {  // curve
        double lx = 0.0, ly = 0.0;
        LastPoint(els, path.el_begin, i, aff, &lx, &ly);
        double p1x, p1y, p2x, p2y, p3x, p3y;
        Xform(aff, els[i].p[0], els[i].p[1], &p1x, &p1y);
        Xform(aff, els[i].p[2], els[i].p[3], &p2x, &p2y);
        Xform(aff, els[i].p[4], els[i].p[5], &p3x, &p3y);
        for (uint32_t k = 0; k < n; ++k) {
            const double t1 = static_cast<double>(k + 1) / static_cast<double>(n);
            emit(k, CubicEval(lx, p1x, p2x, p3x, t1), CubicEval(ly, p1y, p2y, p3y, t1));
        }
    }
    el_bbox[4 * static_cast<size_t>(i) + 0] = bx0;
    el_bbox[4 * static_cast<size_t>(i) + 1] = by0;
    el_bbox[4 * static_cast<size_t>(i) + 2] = bx1;
    el_bbox[4 * static_cast<size_t>(i) + 3] = by1;
}

__device__ __forceinline__ uint16_t SatU16(double v) { return static_cast<uint16_t>(fmin(fmax(v, 0.0), 65535.0)); }

// min / max of four doubles over the 64 lanes of a wave (all lanes get the result)
__device__ __forceinline__ void WaveBox(double &x0, double &y0, double &x1, double &y1) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x0 = fmin(x0, __shfl_xor(x0, d, 64));
        y0 = fmin(y0, __shfl_xor(y0, d, 64));
        x1 = fmax(x1, __shfl_xor(x1, d, 64));
        y1 = fmax(y1, __shfl_xor(y1, d, 64));
    }
}

// One WAVE per sub-path: the lanes stride over its elements and the element boxes are united with
// shuffles (a thread per sub-path walked a 2 488-point outline alone: 57 us of a 0.15 ms re-encode).
// fmin / fmax are exact and associative: the union is the one Rect::union_pt builds point by point.
__global__ void KItems(const pm_path *paths, uint32_t n_paths, const pm_path_el *els, float width_scale,
                       const uint32_t *el_npts, const uint32_t *el_ptoff, const uint32_t *el_mvoff,
                       const uint32_t *path_item_base, const uint32_t *path_pt_base, const uint32_t *sub_first_el,
                       const double *el_bbox, const uint32_t *totals, uint8_t *scene, uint32_t scene_cap) {
    const uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // sub-path of this wave
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_items = totals[0], n_subs = totals[2];
    if (s >= n_subs) return;  // (whole waves)
    (void)els;
    const uint32_t first = sub_first_el[s];
    const uint32_t p = PathOf(paths, n_paths, first);
    const pm_path path = paths[p];
    const uint32_t sub0 = el_mvoff[path.el_begin];         // first sub-path of this path
    const uint32_t n_sub_path = el_mvoff[path.el_end] - sub0;
    const uint32_t j = s - sub0;
    const uint32_t last = (j + 1 < n_sub_path) ? sub_first_el[s + 1] : path.el_end;
    // union of the element boxes (f64), elements without output are skipped
    // (start from NaN: fmin / fmax return the other operand, so NaN boxes drop out exactly as they do when
    //  the boxes are united one after the other, and a sub-path of nothing but NaN stays NaN)
    auto unite = [&](uint32_t e0, uint32_t e1, double &bx0, double &by0, double &bx1, double &by1) {
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        bx0 = by0 = bx1 = by1 = nan;
        bool any = false;
        for (uint32_t i = e0 + lane; i < e1; i += 64u) {
            if (el_npts[i] == 0) continue;
            const double *b = el_bbox + 4 * static_cast<size_t>(i);
            bx0 = fmin(bx0, b[0]); by0 = fmin(by0, b[1]);
            bx1 = fmax(bx1, b[2]); by1 = fmax(by1, b[3]);
            any = true;
        }
        WaveBox(bx0, by0, bx1, by1);
        if (__ballot(any) == 0ull) bx0 = by0 = bx1 = by1 = 0.0;  // (no element with output: the zero box)
    };
    double bx0, by0, bx1, by1;
    unite(first, last, bx0, by0, bx1, by1);
    const uint32_t n_points = el_ptoff[last] - el_ptoff[first];
    const uint32_t path_pts = el_ptoff[path.el_end] - el_ptoff[path.el_begin];
    const uint32_t local = el_ptoff[first] - el_ptoff[path.el_begin];
    const size_t bbox_start = sizeof(SimpleGroup);
    const size_t items_start = bbox_start + static_cast<size_t>(n_items) * sizeof(ShortBbox);
    const size_t points_start = items_start + static_cast<size_t>(n_items) * kItemSize;
    const bool has_fill = (path.flags & PM_PATH_FILL) != 0;
    const bool has_stroke = (path.flags & PM_PATH_STROKE) != 0;
    uint32_t item = path_item_base[p] + j;
    size_t pts_ix = points_start + 8 * (static_cast<size_t>(path_pt_base[p]) + local);
    if (has_fill && (path.flags & PM_PATH_COMPOUND)) {
        // Extension D11 (pm_layout.h): the path's sub-paths are ONE Fill item.  Every sub-path's
        // wave writes the separator that follows its points; the first one also writes the item.
        const size_t fill_base = points_start + 8 * static_cast<size_t>(path_pt_base[p]);
        const size_t sep_at = fill_base + 8 * (static_cast<size_t>(el_ptoff[last] - el_ptoff[path.el_begin]) + j);
        if (lane == 0 && sep_at + 8 <= scene_cap) {
            uint32_t *sep = reinterpret_cast<uint32_t *>(scene + sep_at);
            sep[0] = kSubpathSeparatorBits;
            sep[1] = local + j;  // index of this sub-path's first point in the item's array
        }
        item = path_item_base[p];
        if (j == 0) {  // (uniform)
            double ux0, uy0, ux1, uy1;  // (Rect::union_pt over every point of the path)
            unite(path.el_begin, path.el_end, ux0, uy0, ux1, uy1);
            if (lane == 0 && items_start + (static_cast<size_t>(item) + 1) * kItemSize <= scene_cap) {
                ShortBbox sb{SatU16(floor(ux0)), SatU16(floor(uy0)), SatU16(ceil(ux1)), SatU16(ceil(uy1))};
                *reinterpret_cast<ShortBbox *>(scene + bbox_start + static_cast<size_t>(item) * sizeof(ShortBbox)) = sb;
                uint32_t *it = reinterpret_cast<uint32_t *>(scene + items_start + static_cast<size_t>(item) * kItemSize);
                it[0] = kItemFill;
                it[1] = kFillCompound | ((path.flags & PM_PATH_EVEN_ODD) ? kFillEvenOdd : 0u);
                it[2] = __builtin_bswap32(path.fill_rgba);
                it[3] = path_pts + n_sub_path;
                it[4] = static_cast<uint32_t>(fill_base);
                it[5] = it[6] = it[7] = 0;
            }
        }
        item = path_item_base[p] + 1u + j;
        pts_ix = fill_base + 8 * (static_cast<size_t>(path_pts) + n_sub_path + local);
    } else if (has_fill) {
        // Encoder::fill, src/lib.rs:195-207
        if (lane == 0 && items_start + (static_cast<size_t>(item) + 1) * kItemSize <= scene_cap) {
            ShortBbox sb{SatU16(floor(bx0)), SatU16(floor(by0)), SatU16(ceil(bx1)), SatU16(ceil(by1))};
            *reinterpret_cast<ShortBbox *>(scene + bbox_start + static_cast<size_t>(item) * sizeof(ShortBbox)) = sb;
            uint32_t *it = reinterpret_cast<uint32_t *>(scene + items_start + static_cast<size_t>(item) * kItemSize);
            it[0] = kItemFill;
            it[1] = (path.flags & PM_PATH_EVEN_ODD) ? kFillEvenOdd : 0u;  // PietFill.flags (0 in the reference)
            it[2] = __builtin_bswap32(path.fill_rgba);
            it[3] = n_points;
            it[4] = static_cast<uint32_t>(pts_ix);
            it[5] = it[6] = it[7] = 0;  // write_struct copies 20 bytes; the Metal buffer starts zeroed
        }
        item += n_sub_path;
        pts_ix += 8 * static_cast<size_t>(path_pts);
    }
    if (has_stroke && lane == 0) {
        // encode_path_stroke + Encoder::polyline, src/lib.rs:353-367, :209-222
        float width = path.stroke_width * width_scale;  // src/lib.rs:320
        uint32_t rgba = path.stroke_rgba;
        if (width < kThinLine) {
            float alpha = static_cast<float>(rgba & 0xffu);
            alpha = alpha * sqrtf(width / kThinLine);
            rgba = (rgba & ~0xffu) | static_cast<uint32_t>(alpha);
            width = kThinLine;
        }
        if (items_start + (static_cast<size_t>(item) + 1) * kItemSize <= scene_cap) {
            const double hw = static_cast<double>(width * 0.5f);
            ShortBbox sb{SatU16(floor(bx0 - hw)), SatU16(floor(by0 - hw)), SatU16(ceil(bx1 + hw)), SatU16(ceil(by1 + hw))};
            *reinterpret_cast<ShortBbox *>(scene + bbox_start + static_cast<size_t>(item) * sizeof(ShortBbox)) = sb;
            uint32_t *it = reinterpret_cast<uint32_t *>(scene + items_start + static_cast<size_t>(item) * kItemSize);
            it[0] = kItemPoly;
            it[1] = __builtin_bswap32(rgba);
            it[2] = __float_as_uint(width);
            it[3] = n_points;
            it[4] = static_cast<uint32_t>(pts_ix);
            it[5] = it[6] = it[7] = 0;
        }
    }
}

__global__ void KHeader(uint8_t *scene, const uint32_t *totals, uint32_t fixed_n_items, uint32_t scene_cap) {
    // Encoder::begin_group, src/lib.rs:132-144
    const uint32_t n_items = totals ? totals[0] : fixed_n_items;
    if (sizeof(SimpleGroup) > scene_cap) return;
    SimpleGroup g;
    g.n_items = n_items;
    g.items_ix = static_cast<uint32_t>(sizeof(SimpleGroup) + static_cast<size_t>(n_items) * sizeof(ShortBbox));
    *reinterpret_cast<SimpleGroup *>(scene) = g;
}

}  // namespace

#define PM_HIP_TRY(expr)                        \
    do {                                        \
        hipError_t e_ = (expr);                 \
        if (e_ != hipSuccess) { hip_err = e_; goto fail; } \
    } while (0)

void FlattenCache::Free() {
    if (d_paths) (void)hipFree(d_paths);
    if (d_els) (void)hipFree(d_els);
    if (d_u32) (void)hipFree(d_u32);
    if (d_bbox) (void)hipFree(d_bbox);
    if (h_meta) (void)hipHostFree(h_meta);
    *this = FlattenCache();
}

namespace {
template <typename T>
hipError_t Grow(T **p, size_t *cap, size_t need) {
    if (need <= *cap && *p) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + (need >> 2) + 16;
    const hipError_t e = hipMalloc(p, want * sizeof(T));
    if (e == hipSuccess) *cap = want;
    return e;
}
}  // namespace

hipError_t FlattenCache::Reserve(size_t n_paths, size_t n_els) {
    const size_t n_u32 = n_els * 2 + (n_els + 1) * 2 + n_paths * 2 + n_els + 16 + 2 * ((n_els + 1023) / 1024 + (n_paths + 1023) / 1024);
    hipError_t e = Grow(&d_paths, &cap_paths, n_paths);
    if (e == hipSuccess) e = Grow(&d_els, &cap_els, n_els);
    if (e == hipSuccess) e = Grow(&d_u32, &cap_u32, n_u32);
    if (e == hipSuccess) e = Grow(&d_bbox, &cap_bbox, n_els * 4);
    if (e == hipSuccess && !h_meta) {
        e = hipHostMalloc(&h_meta, 1u << 20, hipHostMallocDefault);
        if (e == hipSuccess) cap_meta = 1u << 20;
    }
    return e;
}


int FlattenEncodeOnDevice(hipStream_t stream, FlattenCache *cache, bool use_resident, const pm_path *h_paths, size_t n_paths, const pm_path_el *h_els,
                          size_t n_els, const double affine[6], float width_scale, uint8_t *d_scene, size_t scene_cap,
                          size_t *scene_bytes, uint32_t *n_items_out, hipError_t *hip_error) {
    hipError_t hip_err = hipSuccess;
    int status = PM_OK;
    if (use_resident) {
        if (!cache->resident) return PM_ERR_INVALID;
        n_paths = cache->n_paths;
        n_els = cache->n_els;
    }
    // (KItems runs a wave per sub-path on a 32-bit grid, the scratch offsets are 32-bit too: documented limit,
    //  include/piet_metal_amd.h)
    if (n_els > kMaxFlattenEls || n_paths > kMaxFlattenEls) return PM_ERR_CAPACITY;
    pm_path *d_paths = nullptr;
    pm_path_el *d_els = nullptr;
    uint32_t *d_u32 = nullptr;  // el_npts, el_move, el_ptoff(+1), el_mvoff(+1), path_item_base, path_pt_base, sub_first, totals(4), err
    double *d_bbox = nullptr;
    Affine aff;
    for (int k = 0; k < 6; ++k) aff.m[k] = affine[k];
    const uint32_t ne = static_cast<uint32_t>(n_els), np = static_cast<uint32_t>(n_paths);
    // (+ the parallel scan's block totals: two words per block of 1 024 elements / paths)
    const size_t n_u32 = static_cast<size_t>(ne) * 2 + (static_cast<size_t>(ne) + 1) * 2 + static_cast<size_t>(np) * 2 + ne + 16 +
                         2 * ((static_cast<size_t>(ne) + 1023) / 1024 + (static_cast<size_t>(np) + 1023) / 1024);

    // host-side structural check: paths must tile the element array in order
    if (!use_resident) {
        uint32_t expect = 0;
        for (size_t p = 0; p < n_paths; ++p) {
            if (h_paths[p].el_begin != expect || h_paths[p].el_end < h_paths[p].el_begin || h_paths[p].el_end > ne) return PM_ERR_INVALID;
            expect = h_paths[p].el_end;
        }
        if (expect != ne) return PM_ERR_INVALID;
        cache->resident = false;
    }
    if (n_paths == 0 || n_els == 0) {
        // empty group
        if (scene_cap < sizeof(SimpleGroup)) return PM_ERR_CAPACITY;
        hipLaunchKernelGGL(KHeader, dim3(1), dim3(1), 0, stream, d_scene, static_cast<const uint32_t *>(nullptr), 0u, static_cast<uint32_t>(scene_cap));
        PM_HIP_TRY(hipStreamSynchronize(stream));
        *scene_bytes = sizeof(SimpleGroup);
        *n_items_out = 0;
        cache->meta_bytes = 0;
        return PM_OK;
    }

    PM_HIP_TRY(Grow(&cache->d_paths, &cache->cap_paths, n_paths));
    PM_HIP_TRY(Grow(&cache->d_els, &cache->cap_els, n_els));
    PM_HIP_TRY(Grow(&cache->d_u32, &cache->cap_u32, n_u32));
    PM_HIP_TRY(Grow(&cache->d_bbox, &cache->cap_bbox, n_els * 4));
    d_paths = cache->d_paths;
    d_els = cache->d_els;
    d_u32 = cache->d_u32;
    d_bbox = cache->d_bbox;
    if (!use_resident) {
        PM_HIP_TRY(hipMemcpyAsync(d_paths, h_paths, n_paths * sizeof(pm_path), hipMemcpyHostToDevice, stream));
        PM_HIP_TRY(hipMemcpyAsync(d_els, h_els, n_els * sizeof(pm_path_el), hipMemcpyHostToDevice, stream));
        // sub-paths open with a MoveTo: an upper bound of the items the encode can produce (a fill and a
        // stroke per sub-path, plus one compound fill per path)
        size_t moves = 0;
        for (size_t i = 0; i < n_els; ++i) moves += h_els[i].tag == PM_EL_MOVE ? 1u : 0u;
        cache->max_items = 2 * moves + n_paths;
    }
    {
        // The four kernels go out back to back: what KPoints / KItems need from KScan (item count, sub-path
        // count) they read from device memory, every store is checked against scene_cap, and the host looks
        // at the totals, the error flag and the head of the scene (header, boxes, items: what validation and
        // arena sizing read) after ONE wait at the end.
        uint32_t *el_npts = d_u32;
        uint32_t *el_move = el_npts + ne;
        uint32_t *el_ptoff = el_move + ne;
        uint32_t *el_mvoff = el_ptoff + ne + 1;
        uint32_t *path_item_base = el_mvoff + ne + 1;
        uint32_t *path_pt_base = path_item_base + np;
        uint32_t *sub_first = path_pt_base + np;
        uint32_t *d_totals = sub_first + ne;
        uint32_t *d_err = d_totals + 4;
        const size_t meta_want = std::min<size_t>(scene_cap, sizeof(SimpleGroup) + cache->max_items * (sizeof(ShortBbox) + kItemSize));
        if (meta_want + 32 > cache->cap_meta || !cache->h_meta) {
            if (cache->h_meta) (void)hipHostFree(cache->h_meta);
            cache->h_meta = nullptr;
            cache->cap_meta = 0;
            const size_t want = meta_want + (meta_want >> 2) + 4096;
            PM_HIP_TRY(hipHostMalloc(&cache->h_meta, want, hipHostMallocDefault));
            cache->cap_meta = want;
        }
        uint32_t *h_totals = reinterpret_cast<uint32_t *>(cache->h_meta);  // [0..3] totals, [4] error flag; the scene head follows at +32
        PM_HIP_TRY(hipMemsetAsync(d_totals, 0, 8 * sizeof(uint32_t), stream));
        const uint32_t tb = 256;
        const uint32_t cap32 = static_cast<uint32_t>(std::min<size_t>(scene_cap, 0xffffffffull));
        hipLaunchKernelGGL(KCount, dim3((ne + tb - 1) / tb), dim3(tb), 0, stream, d_paths, np, d_els, ne, aff, el_npts, el_move, d_err);
        if (ne <= ScanSplit()) {
            hipLaunchKernelGGL(KScan, dim3(1), dim3(kScanThreads), 0, stream, d_paths, np, ne, el_npts, el_move, el_ptoff, el_mvoff,
                               path_item_base, path_pt_base, d_totals);
        } else {
            const uint32_t nb_e = (ne + kScanThreads - 1u) / kScanThreads, nb_p = (np + kScanThreads - 1u) / kScanThreads;
            uint32_t *tops_e = d_err + 4;        // [2 nb_e] (behind the totals and the error word)
            uint32_t *tops_p = tops_e + 2u * nb_e;  // [2 nb_p]
            hipLaunchKernelGGL(KScanA, dim3(nb_e), dim3(kScanThreads), 0, stream, ne, el_npts, el_move, el_ptoff, el_mvoff, tops_e);
            // (totals[2] = sub-paths: the grand total of the moves, which also closes el_mvoff)
            hipLaunchKernelGGL(KScanTops, dim3(1), dim3(kScanThreads), 0, stream, nb_e, tops_e, el_ptoff + ne, el_mvoff + ne, d_totals + 2);
            hipLaunchKernelGGL(KScanD, dim3(nb_p), dim3(kScanThreads), 0, stream, d_paths, np, ne, el_ptoff, el_mvoff, tops_e, path_item_base,
                               path_pt_base, tops_p);
            hipLaunchKernelGGL(KScanTops, dim3(1), dim3(kScanThreads), 0, stream, nb_p, tops_p, d_totals, d_totals + 1, static_cast<uint32_t *>(nullptr));
            hipLaunchKernelGGL(KScanApply, dim3(std::max(nb_e, nb_p)), dim3(kScanThreads), 0, stream, ne, el_ptoff, el_mvoff, tops_e, np, path_item_base,
                               path_pt_base, tops_p);
        }
        hipLaunchKernelGGL(KHeader, dim3(1), dim3(1), 0, stream, d_scene, static_cast<const uint32_t *>(d_totals), 0u, cap32);
        hipLaunchKernelGGL(KPoints, dim3((ne + tb - 1) / tb), dim3(tb), 0, stream, d_paths, np, d_els, ne, aff, el_npts, el_ptoff,
                           el_mvoff, path_pt_base, static_cast<const uint32_t *>(d_totals), d_scene, cap32, d_bbox, sub_first);
        // (a wave per sub-path; sub-paths <= elements)
        hipLaunchKernelGGL(KItems, dim3((ne * 64u + tb - 1) / tb), dim3(tb), 0, stream, d_paths, np, d_els, width_scale, el_npts, el_ptoff,
                           el_mvoff, path_item_base, path_pt_base, sub_first, d_bbox, static_cast<const uint32_t *>(d_totals), d_scene, cap32);
        PM_HIP_TRY(hipGetLastError());
        PM_HIP_TRY(hipMemcpyAsync(h_totals, d_totals, sizeof(uint32_t) * 5, hipMemcpyDeviceToHost, stream));
        PM_HIP_TRY(hipMemcpyAsync(cache->h_meta + 32, d_scene, meta_want, hipMemcpyDeviceToHost, stream));
        PM_HIP_TRY(hipStreamSynchronize(stream));
        cache->meta_bytes = 0;
        if (h_totals[4]) {
            status = PM_ERR_INVALID;  // LineTo/CurveTo before MoveTo: the reference panics
            goto fail;
        }
        const uint32_t n_items = h_totals[0];
        const size_t need = sizeof(SimpleGroup) + static_cast<size_t>(n_items) * (sizeof(ShortBbox) + kItemSize) + static_cast<size_t>(h_totals[1]) * 8;
        if (need > scene_cap || need > 0xffffffffull) {
            status = PM_ERR_CAPACITY;
            *scene_bytes = need;
            goto fail;
        }
        *scene_bytes = need;
        *n_items_out = n_items;
        if (sizeof(SimpleGroup) + static_cast<size_t>(n_items) * (sizeof(ShortBbox) + kItemSize) <= meta_want) cache->meta_bytes = meta_want;
        cache->resident = true;
        cache->n_paths = n_paths;
        cache->n_els = n_els;
    }
fail:
    if (hip_err != hipSuccess) {
        if (hip_error) *hip_error = hip_err;
        return PM_ERR_HIP;
    }
    return status;
}

}  // namespace pm
