/*
 * ORACLE (test infrastructure, NOT product code) -- see pmo.h.
 *
 * Restatement of the scene encoder and test scenes of the reference:
 *   src/lib.rs:15-77    #[repr(C)] scene structs
 *   src/lib.rs:79-254   Encoder
 *   src/lib.rs:257-284  make_cardioid / make_path_test
 *   src/lib.rs:286-367  make_tiger's two passes, encode_path, encode_path_stroke
 */
#include "pmo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- little helpers ------------------------------------------------------ */

static void put_u32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static void put_f32(uint8_t *p, float v) { memcpy(p, &v, 4); }

/* u32::to_be on a little-endian host (src/lib.rs:181, :200, :213) */
static uint32_t to_be(uint32_t v) {
    return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
}

typedef struct {
    double x0, y0, x1, y1;
} rect;

/* ShortBbox::from_rect, src/lib.rs:88-97.  Rust f64::max/min ignore NaN. */
static uint16_t clamp_u16(double v) {
    v = fmax(v, 0.0);
    v = fmin(v, 65535.0);
    return (uint16_t)v;
}

static void short_bbox(rect r, uint16_t out[4]) {
    out[0] = clamp_u16(floor(r.x0));
    out[1] = clamp_u16(floor(r.y0));
    out[2] = clamp_u16(ceil(r.x1));
    out[3] = clamp_u16(ceil(r.y1));
}

/* ---- Encoder -------------------------------------------------------------- */

void pmo_encoder_init(pmo_encoder *e, uint8_t *buf, size_t cap) {
    /* Encoder::new, src/lib.rs:104-112 */
    e->buf = buf;
    e->cap = cap;
    e->free_space = 0;
    e->group_count = 0;
    e->group_start = 0;
    e->group_ix = 0;
    e->error = 0;
    e->open = 0;
    e->depth = 0;
}

size_t pmo_encoder_alloc(pmo_encoder *e, size_t size) {
    /* Encoder::alloc, src/lib.rs:114-118 (no bounds check there; the slice
     * index in write_struct panics instead -- here: error flag) */
    size_t result = e->free_space;
    e->free_space += size;
    if (e->free_space > e->cap) e->error = 1;
    return result;
}

/* Encoder::write_struct, src/lib.rs:122-130: copies exactly sizeof(T) bytes */
static void write_bytes(pmo_encoder *e, size_t ix, const void *s, size_t len) {
    if (ix + len > e->cap) {
        e->error = 1;
        return;
    }
    memcpy(e->buf + ix, s, len);
}

void pmo_encoder_begin_group(pmo_encoder *e, size_t n_items) {
    /* src/lib.rs:132-144 */
    if (e->open) { /* extension: a nested group takes the next item slot of the open one */
        if (!(e->group_ix < e->group_count) || e->depth >= 32) {
            e->error = 1;
            return;
        }
        e->stack[e->depth][0] = e->group_count;
        e->stack[e->depth][1] = e->group_ix;
        e->stack[e->depth][2] = e->group_start;
        e->depth++;
        e->group_ix = 0;
    }
    e->open = 1;
    size_t item_start = PMO_GROUP_HDR + n_items * PMO_BBOX_SIZE;
    size_t total_size = item_start + n_items * PMO_ITEM_SIZE;
    e->group_start = pmo_encoder_alloc(e, total_size);
    e->group_count = n_items;
    uint8_t g[8];
    put_u32(g, (uint32_t)n_items);
    put_u32(g + 4, (uint32_t)(e->group_start + item_start));
    write_bytes(e, e->group_start, g, 8);
}

static void add_item(pmo_encoder *e, const void *item, size_t item_len, const uint16_t bbox[4]);

void pmo_encoder_end_group(pmo_encoder *e) {
    /* src/lib.rs:146-149: assert_eq!(group_ix, group_count) */
    if (e->group_ix != e->group_count) e->error = 1;
    if (e->depth == 0) {
        e->open = 0;
        return;
    }
    /* extension ("when we have nested groups", :148): the child becomes a group item of its
     * parent, boxed by the union of its children's boxes */
    uint16_t u[4] = {0xffff, 0xffff, 0, 0};
    int any = 0;
    for (size_t i = 0; i < e->group_count && !e->error; i++) {
        size_t at = e->group_start + PMO_GROUP_HDR + i * PMO_BBOX_SIZE;
        uint16_t b[4];
        if (at + 8 > e->cap) break;
        memcpy(b, e->buf + at, 8);
        if (b[0] < u[0]) u[0] = b[0];
        if (b[1] < u[1]) u[1] = b[1];
        if (b[2] > u[2]) u[2] = b[2];
        if (b[3] > u[3]) u[3] = b[3];
        any = 1;
    }
    if (!any) u[0] = u[1] = u[2] = u[3] = 0;
    uint8_t item[12];
    put_u32(item + 0, PMO_ITEM_GROUP);
    put_u32(item + 4, 0);
    put_u32(item + 8, (uint32_t)e->group_start);
    e->depth--;
    e->group_count = e->stack[e->depth][0];
    e->group_ix = e->stack[e->depth][1];
    e->group_start = e->stack[e->depth][2];
    add_item(e, item, 12, u);
}

/* Encoder::add_item, src/lib.rs:151-163 */
static void add_item(pmo_encoder *e, const void *item, size_t item_len, const uint16_t bbox[4]) {
    if (!(e->group_ix < e->group_count)) {
        e->error = 1;
        return;
    }
    size_t bbox_ix = e->group_start + PMO_GROUP_HDR + e->group_ix * PMO_BBOX_SIZE;
    write_bytes(e, bbox_ix, bbox, PMO_BBOX_SIZE);
    size_t item_ix = e->group_start + PMO_GROUP_HDR + e->group_count * PMO_BBOX_SIZE +
                     e->group_ix * PMO_ITEM_SIZE;
    write_bytes(e, item_ix, item, item_len);
    e->group_ix += 1;
}

void pmo_encoder_circle(pmo_encoder *e, double cx, double cy, double r) {
    /* src/lib.rs:167-174; PietCircle is just the 4-byte tag (src/lib.rs:33-37).
     * kurbo Circle::bounding_box = (cx-r, cy-r, cx+r, cy+r). */
    uint8_t item[4];
    put_u32(item, PMO_ITEM_CIRCLE);
    rect bb = {cx - r, cy - r, cx + r, cy + r};
    uint16_t sb[4];
    short_bbox(bb, sb);
    add_item(e, item, 4, sb);
}

void pmo_encoder_ellipse(pmo_encoder *e, double cx, double cy, double rx, double ry) {
    /* Extension D10: a Circle item with the ellipse bit; the shape is the ellipse inscribed in the
     * item's (integer) bbox, as the circle is the one inscribed in its bbox (PietRender.metal:484-490). */
    uint8_t item[4];
    put_u32(item, PMO_ITEM_CIRCLE | PMO_CIRCLE_ELLIPSE);
    rect bb = {cx - rx, cy - ry, cx + rx, cy + ry};
    uint16_t sb[4];
    short_bbox(bb, sb);
    add_item(e, item, 4, sb);
}

void pmo_encoder_stroke_line(pmo_encoder *e, double x0, double y0, double x1, double y1,
                             float width, uint32_t rgba) {
    /* src/lib.rs:177-192; PietStrokeLine layout src/lib.rs:39-48 (32 bytes) */
    uint8_t item[32];
    put_u32(item + 0, PMO_ITEM_LINE);
    put_u32(item + 4, 0);
    put_u32(item + 8, to_be(rgba));
    put_f32(item + 12, width);
    put_f32(item + 16, (float)x0); /* point_to_f32s, src/lib.rs:99-101 */
    put_f32(item + 20, (float)y0);
    put_f32(item + 24, (float)x1);
    put_f32(item + 28, (float)y1);
    double hw = (double)(width * 0.5f);
    /* Line::bounding_box = Rect::from_points(p0, p1) (normalised), inflate(hw, hw) */
    rect bb = {fmin(x0, x1) - hw, fmin(y0, y1) - hw, fmax(x0, x1) + hw, fmax(y0, y1) + hw};
    uint16_t sb[4];
    short_bbox(bb, sb);
    add_item(e, item, 32, sb);
}

/* Encoder::encode_points, src/lib.rs:224-240.  Returns points_ix, fills bbox. */
static size_t encode_points(pmo_encoder *e, const double *pts, size_t n, rect *bbox_out) {
    size_t points_ix = pmo_encoder_alloc(e, n * 8);
    size_t dst = points_ix;
    rect bb = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        double x = pts[2 * i], y = pts[2 * i + 1];
        if (i == 0) {
            bb.x0 = bb.x1 = x; /* Rect::from_points(pt, pt) */
            bb.y0 = bb.y1 = y;
        } else {
            bb.x0 = fmin(bb.x0, x); /* Rect::union_pt */
            bb.y0 = fmin(bb.y0, y);
            bb.x1 = fmax(bb.x1, x);
            bb.y1 = fmax(bb.y1, y);
        }
        float f[2] = {(float)x, (float)y};
        write_bytes(e, dst, f, 8);
        dst += 8;
    }
    if (n == 0) e->error = 1; /* .expect("encoded empty points vector") */
    *bbox_out = bb;
    return points_ix;
}

void pmo_encoder_fill(pmo_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba) {
    pmo_encoder_fill_rule(e, pts_xy, n_points, rgba, 0);
}

void pmo_encoder_fill_rule(pmo_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba, uint32_t fill_flags) {
    /* src/lib.rs:195-207; PietFill layout src/lib.rs:50-58 (20 bytes written); flags = 0 there */
    rect bb;
    size_t points_ix = encode_points(e, pts_xy, n_points, &bb);
    uint8_t item[20];
    put_u32(item + 0, PMO_ITEM_FILL);
    put_u32(item + 4, fill_flags);
    put_u32(item + 8, to_be(rgba));
    put_u32(item + 12, (uint32_t)n_points);
    put_u32(item + 16, (uint32_t)points_ix);
    uint16_t sb[4];
    short_bbox(bb, sb);
    add_item(e, item, 20, sb);
}

void pmo_encoder_fill_compound(pmo_encoder *e, const double *pts_xy, const uint32_t *sub_counts, size_t n_sub, uint32_t rgba,
                               uint32_t fill_flags) {
    /* Extension D11 (pmo.h): the sub-paths' points back to back, a separator after each. */
    size_t total = 0;
    for (size_t k = 0; k < n_sub; k++) total += (size_t)sub_counts[k] + 1;
    size_t points_ix = pmo_encoder_alloc(e, total * 8);
    size_t dst = points_ix;
    rect bb = {0, 0, 0, 0};
    int first = 1;
    uint32_t at = 0;
    for (size_t k = 0; k < n_sub; k++) {
        uint32_t start = at;
        if (sub_counts[k] == 0) e->error = 1; /* .expect("encoded empty points vector") */
        for (uint32_t i = 0; i < sub_counts[k]; i++, pts_xy += 2, at++, dst += 8) {
            double x = pts_xy[0], y = pts_xy[1];
            if (first) {
                bb.x0 = bb.x1 = x;
                bb.y0 = bb.y1 = y;
                first = 0;
            } else {
                bb.x0 = fmin(bb.x0, x);
                bb.y0 = fmin(bb.y0, y);
                bb.x1 = fmax(bb.x1, x);
                bb.y1 = fmax(bb.y1, y);
            }
            float f[2] = {(float)x, (float)y};
            write_bytes(e, dst, f, 8);
        }
        uint32_t sep[2] = {0x7fc00000u, start};
        write_bytes(e, dst, sep, 8);
        dst += 8;
        at++;
    }
    if (n_sub == 0) e->error = 1;
    uint8_t item[20];
    put_u32(item + 0, PMO_ITEM_FILL);
    put_u32(item + 4, fill_flags | PMO_FILL_COMPOUND);
    put_u32(item + 8, to_be(rgba));
    put_u32(item + 12, (uint32_t)total);
    put_u32(item + 16, (uint32_t)points_ix);
    uint16_t sb[4];
    short_bbox(bb, sb);
    add_item(e, item, 20, sb);
}

void pmo_encoder_polyline(pmo_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba,
                          float width) {
    /* src/lib.rs:209-222; PietStrokePolyLine layout src/lib.rs:60-68 (20 bytes) */
    rect bb;
    size_t points_ix = encode_points(e, pts_xy, n_points, &bb);
    uint8_t item[20];
    put_u32(item + 0, PMO_ITEM_POLY);
    put_u32(item + 4, to_be(rgba));
    put_f32(item + 8, width);
    put_u32(item + 12, (uint32_t)n_points);
    put_u32(item + 16, (uint32_t)points_ix);
    double hw = (double)(width * 0.5f);
    rect ib = {bb.x0 - hw, bb.y0 - hw, bb.x1 + hw, bb.y1 + hw};
    uint16_t sb[4];
    short_bbox(ib, sb);
    add_item(e, item, 20, sb);
}

/* ---- scenes ---------------------------------------------------------------- */

int64_t pmo_scene_cardioid(uint8_t *buf, size_t cap) {
    /* make_cardioid, src/lib.rs:257-270 */
    pmo_encoder e;
    pmo_encoder_init(&e, buf, cap);
    const int n = 97;
    const double dth = M_PI * 2.0 / (double)n;
    const double cx = 1024.0, cy = 768.0, r = 750.0;
    pmo_encoder_begin_group(&e, (size_t)(n - 1) * 2);
    for (int i = 1; i < n; i++) {
        double th0 = (double)i * dth;
        double th1 = (double)((i * 2) % n) * dth;
        /* center + Vec2::from_angle(th) * r */
        double p0x = cx + cos(th0) * r, p0y = cy + sin(th0) * r;
        double p1x = cx + cos(th1) * r, p1y = cy + sin(th1) * r;
        pmo_encoder_circle(&e, p0x, p0y, 8.0);
        pmo_encoder_stroke_line(&e, p0x, p0y, p1x, p1y, 2.0f, 0x000080e0u);
    }
    pmo_encoder_end_group(&e);
    return e.error ? -1 : (int64_t)e.free_space;
}

int64_t pmo_scene_path_test(uint8_t *buf, size_t cap) {
    /* make_path_test, src/lib.rs:273-284 */
    pmo_encoder e;
    pmo_encoder_init(&e, buf, cap);
    pmo_encoder_begin_group(&e, 1);
    const double pts[6] = {10.0, 10.0, 15.0, 800.0, 300.0, 500.0};
    pmo_encoder_fill(&e, pts, 3, 0x80e0u);
    pmo_encoder_end_group(&e);
    return e.error ? -1 : (int64_t)e.free_space;
}

/* ---- make_tiger on parsed paths ------------------------------------------- */

#define TOLERANCE 0.1  /* src/lib.rs:330 */
#define THIN_LINE 0.7f /* src/lib.rs:351 */

typedef struct {
    uint32_t *sub_counts;
    size_t sub_cap;
    double *pts;
    size_t pts_cap;
} scratch;

static int flatten_grow(const pmo_path_el *els, const pmo_path *p, const double affine[6],
                        scratch *s, int64_t *n_sub, size_t *n_pts) {
    for (;;) {
        int64_t r = pmo_flatten_path(els, p->el_begin, p->el_end, affine, TOLERANCE,
                                     s->sub_counts, s->sub_cap, s->pts, s->pts_cap, n_pts);
        if (r >= 0) {
            *n_sub = r;
            return 0;
        }
        if (r == -2) return -1; /* malformed path (LineTo before MoveTo) */
        /* capacity: grow and retry */
        size_t need_sub = (size_t)(p->el_end - p->el_begin) + 1;
        if (s->sub_cap < need_sub) {
            s->sub_cap = need_sub;
            s->sub_counts = (uint32_t *)realloc(s->sub_counts, s->sub_cap * sizeof(uint32_t));
        }
        if (s->pts_cap < *n_pts) {
            s->pts_cap = *n_pts;
            s->pts = (double *)realloc(s->pts, s->pts_cap * 2 * sizeof(double));
        }
        if (!s->sub_counts || !s->pts) return -1;
    }
}

int64_t pmo_scene_from_paths(uint8_t *buf, size_t cap, const pmo_path *paths, size_t n_paths,
                             const pmo_path_el *els, size_t n_els, const double affine[6],
                             uint32_t *n_items_out) {
    (void)n_els;
    scratch s = {NULL, 0, NULL, 0};
    s.sub_cap = 64;
    s.pts_cap = 4096;
    s.sub_counts = (uint32_t *)malloc(s.sub_cap * sizeof(uint32_t));
    s.pts = (double *)malloc(s.pts_cap * 2 * sizeof(double));
    pmo_encoder e;
    pmo_encoder_init(&e, buf, cap);
    int64_t result = -1;

    /* pass 1 (src/lib.rs:293-306): count items */
    size_t n_items = 0;
    for (size_t i = 0; i < n_paths; i++) {
        int64_t n_sub;
        size_t n_pts;
        if (flatten_grow(els, &paths[i], affine, &s, &n_sub, &n_pts)) goto done;
        if (paths[i].flags & PMO_PATH_FILL) /* count_fill_items; a compound fill (extension D11) is one item */
            n_items += (paths[i].flags & PMO_PATH_COMPOUND) ? (size_t)(n_sub > 0) : (size_t)n_sub;
        if (paths[i].flags & PMO_PATH_STROKE) n_items += (size_t)n_sub; /* count_stroke_items */
    }
    if (n_items_out) *n_items_out = (uint32_t)n_items;
    pmo_encoder_begin_group(&e, n_items); /* src/lib.rs:308 */

    /* pass 2 (src/lib.rs:309-326) */
    for (size_t i = 0; i < n_paths; i++) {
        int64_t n_sub;
        size_t n_pts;
        if (flatten_grow(els, &paths[i], affine, &s, &n_sub, &n_pts)) goto done;
        if (paths[i].flags & PMO_PATH_FILL) {
            /* encode_path, src/lib.rs:342-347 */
            const double *pp = s.pts;
            const uint32_t rule = (paths[i].flags & PMO_PATH_EVEN_ODD) ? PMO_FILL_EVEN_ODD : 0u;
            if ((paths[i].flags & PMO_PATH_COMPOUND) && n_sub > 0) {
                pmo_encoder_fill_compound(&e, pp, s.sub_counts, (size_t)n_sub, paths[i].fill_rgba, rule);
            } else {
                for (int64_t k = 0; k < n_sub; k++) {
                    pmo_encoder_fill_rule(&e, pp, s.sub_counts[k], paths[i].fill_rgba, rule);
                    pp += 2 * (size_t)s.sub_counts[k];
                }
            }
        }
        if (paths[i].flags & PMO_PATH_STROKE) {
            /* encode_path_stroke, src/lib.rs:353-367 */
            float width = paths[i].stroke_width;
            uint32_t rgba = paths[i].stroke_rgba;
            if (width < THIN_LINE) {
                float alpha = (float)(rgba & 0xffu);
                alpha = alpha * sqrtf(width / THIN_LINE);
                rgba = (rgba & ~0xffu) | (uint32_t)alpha;
                width = THIN_LINE;
            }
            const double *pp = s.pts;
            for (int64_t k = 0; k < n_sub; k++) {
                pmo_encoder_polyline(&e, pp, s.sub_counts[k], rgba, width);
                pp += 2 * (size_t)s.sub_counts[k];
            }
        }
    }
    pmo_encoder_end_group(&e); /* src/lib.rs:327 */
    result = e.error ? -1 : (int64_t)e.free_space;
done:
    free(s.sub_counts);
    free(s.pts);
    return result;
}

/* ---- nested groups -> flat paint order (extension, see pmo.h) ----------------- */

typedef struct {
    const uint8_t *sc;
    size_t len;
    uint8_t *bb, *it; /* growing arrays of 8-byte boxes and 32-byte items */
    size_t n, cap;
    int bad;
} flat_t;

static uint32_t fl_u32(flat_t *f, size_t off) {
    uint32_t v = 0;
    if (off + 4 > f->len) { f->bad = 1; return 0; }
    memcpy(&v, f->sc + off, 4);
    return v;
}

static void flatten_group(flat_t *f, size_t group, int depth) {
    if (depth > 32) { f->bad = 1; return; }
    uint32_t n = fl_u32(f, group), items = fl_u32(f, group + 4);
    if (f->bad) return;
    if (group + 8 + 8ull * n > f->len || (size_t)items + 32ull * n > f->len) { f->bad = 1; return; }
    for (uint32_t i = 0; i < n && !f->bad; i++) {
        size_t it = (size_t)items + 32ull * i;
        if ((fl_u32(f, it) & 0xffffu) == PMO_ITEM_GROUP) {
            flatten_group(f, fl_u32(f, it + 8), depth + 1);
            continue;
        }
        if (f->n == f->cap) {
            f->cap = f->cap ? 2 * f->cap : 256;
            f->bb = (uint8_t *)realloc(f->bb, f->cap * 8);
            f->it = (uint8_t *)realloc(f->it, f->cap * 32);
        }
        if (f->n >= (1u << 24)) { f->bad = 1; return; } /* (also stops cyclic scenes) */
        memcpy(f->bb + 8 * f->n, f->sc + group + 8 + 8ull * i, 8);
        memcpy(f->it + 32 * f->n, f->sc + it, 32);
        f->n++;
    }
}

int64_t pmo_scene_flatten_groups(const uint8_t *scene, size_t scene_len, uint8_t *out, size_t out_cap, size_t *root_out) {
    if (scene_len < 8 || out_cap < scene_len) return -1;
    flat_t f = {scene, scene_len, NULL, NULL, 0, 0, 0};
    uint32_t n = fl_u32(&f, 0), items = fl_u32(&f, 4);
    int nested = 0;
    if ((size_t)items + 32ull * n > scene_len) return -1;
    for (uint32_t i = 0; i < n; i++)
        if ((fl_u32(&f, (size_t)items + 32ull * i) & 0xffffu) == PMO_ITEM_GROUP) nested = 1;
    memcpy(out, scene, scene_len);
    if (root_out) *root_out = 0;
    if (!nested) return (int64_t)scene_len;
    flatten_group(&f, 0, 0);
    int64_t result = -1;
    size_t root = (scene_len + 7u) & ~(size_t)7u;
    size_t total = root + 8 + 40 * f.n;
    if (!f.bad && total <= out_cap) {
        memset(out + scene_len, 0, root - scene_len);
        put_u32(out + root, (uint32_t)f.n);
        put_u32(out + root + 4, (uint32_t)(root + 8 + 8 * f.n));
        memcpy(out + root + 8, f.bb, 8 * f.n);
        memcpy(out + root + 8 + 8 * f.n, f.it, 32 * f.n);
        if (root_out) *root_out = root;
        result = (int64_t)total;
    }
    free(f.bb);
    free(f.it);
    return result;
}
