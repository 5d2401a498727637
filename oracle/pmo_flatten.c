/*
 * ORACLE (test infrastructure, NOT product code) -- see pmo.h.
 *
 * Restatement of src/flatten.rs:10-47 (flatten_path) and of the third-party
 * arithmetic it calls, which is NOT under /root/reference:
 *
 *   kurbo 0.5.6 (Cargo.toml:12, Cargo.lock:8-14), published algorithm restated:
 *     Affine * Point      x' = a*x + c*y + e ; y' = b*x + d*y + f
 *     CubicBez::to_quads(accuracy):
 *         max_hypot2 = 432.0 * accuracy * accuracy      (= (36/sqrt(3))^2 acc^2)
 *         p1x2 = 3*p1 - p0 ; p2x2 = 3*p2 - p3 ; err = |p2x2 - p1x2|^2
 *         n = max(1, ceil((err / max_hypot2)^(1/6)))
 *         piece i = subsegment(i/n .. (i+1)/n); flatten.rs uses only q.p2,
 *         which is the piece's end point = eval((i+1)/n)
 *     CubicBez::eval(t):  mt = 1 - t
 *         p0*(mt*mt*mt) + (p1*(mt*mt*3) + (p2*(mt*3) + p3*t)*t)*t
 *
 * PARITY UNPINNED for this third-party step (no reference test or vector
 * exercises it; the crate source is absent offline).
 */
#include "pmo.h"

#include <math.h>

static void xform(const double m[6], double x, double y, double *ox, double *oy) {
    *ox = m[0] * x + m[2] * y + m[4];
    *oy = m[1] * x + m[3] * y + m[5];
}

static double cubic_eval(double p0, double p1, double p2, double p3, double t) {
    double mt = 1.0 - t;
    return p0 * (mt * mt * mt) + (p1 * (mt * mt * 3.0) + (p2 * (mt * 3.0) + p3 * t) * t) * t;
}

/* kurbo to_quads: n = ceil((err / max_hypot2)^(1/6)), at least 1.  The 6th root through pow() makes
 * the count depend on the last ulp of somebody's libm (glibc here, the device's OCML there, Rust's
 * powf in the reference); the count is an integer property of x, so it is DEFINED without libm:
 * the smallest n >= 1 with n^6 >= x, n^6 = ((n*n)*(n*n))*(n*n) in binary64 (exact below 2^53,
 * monotone beyond).  pow() only supplies the starting guess.  Non-finite x: as the cast does. */
size_t pmo_subdivision_count(double x) {
    if (!(x > 1.0)) return 1;           /* also NaN */
    if (x > 1e96) return (size_t)1e16;  /* (inf as usize saturates in Rust; never reached by real paths) */
    double g = ceil(pow(x, 1.0 / 6.0));
    uint64_t n = g >= 1.0 ? (uint64_t)g : 1;
#define P6(v) ((((double)(v)) * ((double)(v))) * (((double)(v)) * ((double)(v))) * (((double)(v)) * ((double)(v))))
    while (n > 1 && P6(n - 1) >= x) n--;
    while (P6(n) < x) n++;
#undef P6
    return (size_t)n;
}

int64_t pmo_flatten_path(const pmo_path_el *els, uint32_t el_begin, uint32_t el_end,
                         const double affine[6], double tolerance, uint32_t *sub_counts,
                         size_t sub_cap, double *pts, size_t pts_cap, size_t *n_points_out) {
    size_t n_sub = 0;  /* completed + current subpaths */
    size_t n_pts = 0;
    int have_cur = 0;  /* cur_path.is_some() */
    uint32_t cur_n = 0;
    double lx = 0.0, ly = 0.0; /* last_pt = Point::default() */
    int overflow = 0;

#define PUSH(px, py)                      \
    do {                                  \
        if (n_pts < pts_cap) {            \
            pts[2 * n_pts] = (px);        \
            pts[2 * n_pts + 1] = (py);    \
        } else {                          \
            overflow = 1;                 \
        }                                 \
        n_pts++;                          \
        cur_n++;                          \
    } while (0)

    for (uint32_t i = el_begin; i < el_end; i++) {
        const pmo_path_el *el = &els[i];
        switch (el->tag) {
            case PMO_EL_MOVE: { /* flatten.rs:16-22 */
                if (have_cur) {
                    if (n_sub < sub_cap) sub_counts[n_sub] = cur_n; else overflow = 1;
                    n_sub++;
                }
                double x, y;
                xform(affine, el->p[0], el->p[1], &x, &y);
                have_cur = 1;
                cur_n = 0;
                PUSH(x, y);
                lx = x;
                ly = y;
                break;
            }
            case PMO_EL_LINE: { /* flatten.rs:23-26 */
                if (!have_cur) return -2; /* cur_path.as_mut().unwrap() panics */
                double x, y;
                xform(affine, el->p[0], el->p[1], &x, &y);
                PUSH(x, y);
                lx = x;
                ly = y;
                break;
            }
            case PMO_EL_CURVE: { /* flatten.rs:27-39 */
                if (!have_cur) return -2;
                double p1x, p1y, p2x, p2y, p3x, p3y;
                xform(affine, el->p[0], el->p[1], &p1x, &p1y);
                xform(affine, el->p[2], el->p[3], &p2x, &p2y);
                xform(affine, el->p[4], el->p[5], &p3x, &p3y);
                double accuracy = tolerance * 1e-2; /* flatten.rs:35 */
                double max_hypot2 = 432.0 * accuracy * accuracy;
                double ax = p1x * 3.0 - lx, ay = p1y * 3.0 - ly;       /* p1x2 */
                double bx = p2x * 3.0 - p3x, by = p2y * 3.0 - p3y;     /* p2x2 */
                double dx = bx - ax, dy = by - ay;
                double err = dx * dx + dy * dy;
                size_t n = pmo_subdivision_count(err / max_hypot2); /* (ceil(x^(1/6)) as usize).max(1) */
                for (size_t k = 0; k < n; k++) {
                    double t1 = (double)(k + 1) / (double)n;
                    double x = cubic_eval(lx, p1x, p2x, p3x, t1);
                    double y = cubic_eval(ly, p1y, p2y, p3y, t1);
                    PUSH(x, y);
                }
                lx = p3x;
                ly = p3y;
                break;
            }
            default: /* QuadTo, ClosePath ignored: flatten.rs:40 */
                break;
        }
    }
    if (have_cur) { /* flatten.rs:43-45 */
        if (n_sub < sub_cap) sub_counts[n_sub] = cur_n; else overflow = 1;
        n_sub++;
    }
#undef PUSH
    if (n_points_out) *n_points_out = n_pts;
    return overflow ? -1 : (int64_t)n_sub;
}
