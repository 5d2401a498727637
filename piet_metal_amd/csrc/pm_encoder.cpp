// Host-side scene encoder: the piet-metal `Encoder` API surface and byte layout
// (src/lib.rs:79-254), re-implemented in C++ behind the C ABI of
// include/piet_metal_amd.h.  Pure host code; writes into a caller-owned buffer
// (normally the pinned staging buffer returned by pm_scene_buffer).
#include "pm_encoder.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <new>

namespace pm {

namespace {

// u32::to_be() on a little-endian host: colours are stored R,G,B,A in memory
// (src/lib.rs:181, :200, :213).
inline uint32_t ByteSwap(uint32_t v) { return __builtin_bswap32(v); }

struct Rect {
    double x0, y0, x1, y1;
    static Rect FromPoint(double x, double y) { return {x, y, x, y}; }
    void UnionPt(double x, double y) {
        x0 = std::fmin(x0, x);
        y0 = std::fmin(y0, y);
        x1 = std::fmax(x1, x);
        y1 = std::fmax(y1, y);
    }
    Rect Inflate(double w, double h) const { return {x0 - w, y0 - h, x1 + w, y1 + h}; }
};

inline uint16_t SatU16(double v) { return static_cast<uint16_t>(std::fmin(std::fmax(v, 0.0), 65535.0)); }

// ShortBbox::from_rect (src/lib.rs:88-97): floor the min corner, ceil the max.
inline ShortBbox ToShortBbox(const Rect &r) {
    return {SatU16(std::floor(r.x0)), SatU16(std::floor(r.y0)), SatU16(std::ceil(r.x1)),
            SatU16(std::ceil(r.y1))};
}

}  // namespace

Encoder::Encoder(uint8_t *buf, size_t cap) : buf_(buf), cap_(cap) {}

size_t Encoder::Alloc(size_t size) {
    const size_t at = free_space_;
    free_space_ += size;
    if (free_space_ > cap_) status_ = kCapacity;
    return at;
}

void Encoder::Put(size_t at, const void *src, size_t len) {
    if (at + len > cap_) {
        status_ = kCapacity;
        return;
    }
    std::memcpy(buf_ + at, src, len);
}

void Encoder::BeginGroup(size_t n_items) {
    if (group_open_) {
        // nested: the new group takes the next item slot of the open one
        if (group_ix_ >= group_count_ || depth_ >= kMaxDepth) {
            if (status_ == kOk) status_ = kMisuse;
            return;
        }
        stack_[depth_++] = Frame{group_count_, group_ix_, group_start_};
        group_ix_ = 0;
    }
    group_open_ = true;
    const size_t item_start = sizeof(SimpleGroup) + n_items * sizeof(ShortBbox);
    group_start_ = Alloc(item_start + n_items * kItemSize);
    group_count_ = n_items;
    SimpleGroup g{static_cast<uint32_t>(n_items), static_cast<uint32_t>(group_start_ + item_start)};
    Put(group_start_, &g, sizeof(g));
}

void Encoder::EndGroup() {
    if (group_ix_ != group_count_ && status_ == kOk) status_ = kMisuse;  // assert_eq!, :147
    if (depth_ == 0) {
        group_open_ = false;  // (the reference keeps no such flag: one group per scene)
        return;
    }
    // "This will get more interesting when we have nested groups" (:148): the closed group becomes
    // a PietGroup item of its parent, boxed by the union of its children's boxes
    ShortBbox u{0xffff, 0xffff, 0, 0};
    bool any = false;
    for (size_t i = 0; i < group_count_ && status_ == kOk; ++i) {
        ShortBbox b;
        const size_t at = group_start_ + sizeof(SimpleGroup) + i * sizeof(ShortBbox);
        if (at + sizeof(b) > cap_) break;
        std::memcpy(&b, buf_ + at, sizeof(b));
        u.x0 = std::min(u.x0, b.x0);
        u.y0 = std::min(u.y0, b.y0);
        u.x1 = std::max(u.x1, b.x1);
        u.y1 = std::max(u.y1, b.y1);
        any = true;
    }
    if (!any) u = ShortBbox{0, 0, 0, 0};
    const PietGroup item{kItemGroup, 0, static_cast<uint32_t>(group_start_)};
    const Frame f = stack_[--depth_];
    group_count_ = f.count;
    group_ix_ = f.ix;
    group_start_ = f.start;
    AddItem(item, u);
}

template <typename Item>
void Encoder::AddItem(const Item &item, const ShortBbox &bbox) {
    if (group_ix_ >= group_count_) {  // assert!, :152
        if (status_ == kOk) status_ = kMisuse;
        return;
    }
    Put(group_start_ + sizeof(SimpleGroup) + group_ix_ * sizeof(ShortBbox), &bbox, sizeof(bbox));
    Put(group_start_ + sizeof(SimpleGroup) + group_count_ * sizeof(ShortBbox) + group_ix_ * kItemSize,
        &item, sizeof(Item));
    ++group_ix_;
}

void Encoder::Circle(double cx, double cy, double r) {
    PietCircle item{kItemCircle};
    AddItem(item, ToShortBbox(Rect{cx - r, cy - r, cx + r, cy + r}));
}

void Encoder::Ellipse(double cx, double cy, double rx, double ry) {
    PietCircle item{kItemCircle | kCircleEllipse};  // (extension D10: the ellipse inscribed in the bbox)
    AddItem(item, ToShortBbox(Rect{cx - rx, cy - ry, cx + rx, cy + ry}));
}

void Encoder::StrokeLine(double x0, double y0, double x1, double y1, float width, uint32_t rgba) {
    PietStrokeLine item{};
    item.item_type = kItemLine;
    item.flags = 0;
    item.rgba = ByteSwap(rgba);
    item.width = width;
    item.start[0] = static_cast<float>(x0);
    item.start[1] = static_cast<float>(y0);
    item.end[0] = static_cast<float>(x1);
    item.end[1] = static_cast<float>(y1);
    const double hw = static_cast<double>(width * 0.5f);
    Rect bb = Rect::FromPoint(x0, y0);
    bb.UnionPt(x1, y1);
    AddItem(item, ToShortBbox(bb.Inflate(hw, hw)));
}

size_t Encoder::EncodePoints(const double *pts_xy, size_t n, double bbox_out[4]) {
    const size_t points_ix = Alloc(n * 2 * sizeof(float));
    if (n == 0) {  // .expect("encoded empty points vector"), :238
        if (status_ == kOk) status_ = kMisuse;
        return points_ix;
    }
    Rect bb = Rect::FromPoint(pts_xy[0], pts_xy[1]);
    size_t at = points_ix;
    for (size_t i = 0; i < n; ++i, at += 2 * sizeof(float)) {
        const double x = pts_xy[2 * i], y = pts_xy[2 * i + 1];
        if (i) bb.UnionPt(x, y);
        const float xy[2] = {static_cast<float>(x), static_cast<float>(y)};
        Put(at, xy, sizeof(xy));
    }
    bbox_out[0] = bb.x0;
    bbox_out[1] = bb.y0;
    bbox_out[2] = bb.x1;
    bbox_out[3] = bb.y1;
    return points_ix;
}

void Encoder::Fill(const double *pts_xy, size_t n, uint32_t rgba, uint32_t flags) {
    double bb[4] = {0, 0, 0, 0};
    const size_t points_ix = EncodePoints(pts_xy, n, bb);
    PietFill item{kItemFill, flags, ByteSwap(rgba), static_cast<uint32_t>(n),
                  static_cast<uint32_t>(points_ix)};
    AddItem(item, ToShortBbox(Rect{bb[0], bb[1], bb[2], bb[3]}));
}

void Encoder::FillCompound(const double *pts_xy, const uint32_t *sub_counts, size_t n_sub, uint32_t rgba, uint32_t flags) {
    // Extension D11 (pm_layout.h): the sub-paths' points back to back, a separator after each
    size_t total = 0;
    for (size_t k = 0; k < n_sub; ++k) total += static_cast<size_t>(sub_counts[k]) + 1;
    const size_t points_ix = Alloc(total * 2 * sizeof(float));
    if (n_sub == 0 && status_ == kOk) status_ = kMisuse;
    Rect bb{0, 0, 0, 0};
    bool first = true;
    size_t at = points_ix;
    uint32_t index = 0;
    for (size_t k = 0; k < n_sub; ++k) {
        const uint32_t start = index;
        if (sub_counts[k] == 0 && status_ == kOk) status_ = kMisuse;  // .expect("encoded empty points vector"), :238
        for (uint32_t i = 0; i < sub_counts[k]; ++i, pts_xy += 2, ++index, at += 2 * sizeof(float)) {
            if (first) bb = Rect::FromPoint(pts_xy[0], pts_xy[1]);
            else bb.UnionPt(pts_xy[0], pts_xy[1]);
            first = false;
            const float xy[2] = {static_cast<float>(pts_xy[0]), static_cast<float>(pts_xy[1])};
            Put(at, xy, sizeof(xy));
        }
        const uint32_t sep[2] = {kSubpathSeparatorBits, start};
        Put(at, sep, sizeof(sep));
        at += sizeof(sep);
        ++index;
    }
    PietFill item{kItemFill, flags | kFillCompound, ByteSwap(rgba), static_cast<uint32_t>(total), static_cast<uint32_t>(points_ix)};
    AddItem(item, ToShortBbox(bb));
}

void Encoder::Polyline(const double *pts_xy, size_t n, uint32_t rgba, float width) {
    double bb[4] = {0, 0, 0, 0};
    const size_t points_ix = EncodePoints(pts_xy, n, bb);
    PietStrokePolyLine item{kItemPoly, ByteSwap(rgba), width, static_cast<uint32_t>(n),
                            static_cast<uint32_t>(points_ix)};
    const double hw = static_cast<double>(width * 0.5f);
    AddItem(item, ToShortBbox(Rect{bb[0], bb[1], bb[2], bb[3]}.Inflate(hw, hw)));
}

// ---- the reference's flatten-free test scenes ---------------------------------

int64_t SceneCardioid(uint8_t *buf, size_t cap) {  // make_cardioid, src/lib.rs:257-270
    Encoder enc(buf, cap);
    constexpr int n = 97;
    const double dth = M_PI * 2.0 / static_cast<double>(n);
    constexpr double cx = 1024.0, cy = 768.0, r = 750.0;
    enc.BeginGroup((n - 1) * 2);
    for (int i = 1; i < n; ++i) {
        const double a0 = static_cast<double>(i) * dth;
        const double a1 = static_cast<double>((i * 2) % n) * dth;
        const double p0x = cx + std::cos(a0) * r, p0y = cy + std::sin(a0) * r;
        const double p1x = cx + std::cos(a1) * r, p1y = cy + std::sin(a1) * r;
        enc.Circle(p0x, p0y, 8.0);
        enc.StrokeLine(p0x, p0y, p1x, p1y, 2.0f, 0x000080e0u);
    }
    enc.EndGroup();
    return enc.ok() ? static_cast<int64_t>(enc.bytes_used()) : enc.c_status();
}

int64_t ScenePathTest(uint8_t *buf, size_t cap) {  // make_path_test, src/lib.rs:273-284
    Encoder enc(buf, cap);
    enc.BeginGroup(1);
    const double tri[] = {10.0, 10.0, 15.0, 800.0, 300.0, 500.0};
    enc.Fill(tri, 3, 0x80e0u);
    enc.EndGroup();
    return enc.ok() ? static_cast<int64_t>(enc.bytes_used()) : enc.c_status();
}

}  // namespace pm

// ---- C ABI -------------------------------------------------------------------------

struct pm_encoder {
    pm::Encoder enc;
    pm_encoder(uint8_t *b, size_t c) : enc(b, c) {}
};

extern "C" {

pm_encoder *pm_encoder_new(uint8_t *buf, size_t cap) {
    if (!buf) return nullptr;
    return new (std::nothrow) pm_encoder(buf, cap);
}
void pm_encoder_free(pm_encoder *e) { delete e; }
size_t pm_encoder_alloc(pm_encoder *e, size_t size) { return e ? e->enc.Alloc(size) : 0; }
int pm_encoder_begin_group(pm_encoder *e, size_t n_items) {
    if (!e) return PM_ERR_INVALID;
    e->enc.BeginGroup(n_items);
    return e->enc.c_status();
}
int pm_encoder_end_group(pm_encoder *e) {
    if (!e) return PM_ERR_INVALID;
    e->enc.EndGroup();
    return e->enc.c_status();
}
int pm_encoder_circle(pm_encoder *e, double cx, double cy, double r) {
    if (!e) return PM_ERR_INVALID;
    e->enc.Circle(cx, cy, r);
    return e->enc.c_status();
}
// ---- paths with curves (src/lib.rs:194: "Signature will change, need to deal with subpaths and
//      also want curves") -------------------------------------------------------------------
// encode_path / encode_path_stroke of make_tiger (src/lib.rs:342-367) for ONE kurbo BezPath under
// the identity transform: flatten.rs:10-47 on the host (tolerance 0.1, src/lib.rs:330), then the
// encoder calls.  Same arithmetic, in binary64, as the device flatten kernels (pm_flatten.hip) and
// the oracle: the three produce the same bytes.  This is the convenience API for callers that build
// scenes item by item; whole documents go through pm_flatten_and_encode on the device.
namespace {

uint64_t HostSubdivisionCount(double x) {  // kurbo to_quads: the smallest n >= 1 with n^6 >= x (no libm dependency)
    if (!(x > 1.0)) return 1;
    if (x > 1e54) return 1ull << 30;
    const double g = std::ceil(std::pow(x, 1.0 / 6.0));
    uint64_t n = g >= 1.0 ? static_cast<uint64_t>(g) : 1;
    auto p6 = [](uint64_t v) {
        const double d = static_cast<double>(v);
        return ((d * d) * (d * d)) * (d * d);
    };
    while (n > 1 && p6(n - 1) >= x) --n;
    while (p6(n) < x) ++n;
    return n;
}

double HostCubicEval(double p0, double p1, double p2, double p3, double t) {  // kurbo CubicBez::eval
    const double mt = 1.0 - t;
    return p0 * (mt * mt * mt) + (p1 * (mt * mt * 3.0) + (p2 * (mt * 3.0) + p3 * t) * t) * t;
}

// PM_ERR_INVALID: a LineTo / CurveTo before any MoveTo (the reference panics: cur_path.as_mut().unwrap()).
// PM_ERR_CAPACITY: more points than `max_points` -- what the encoder's buffer could still hold; an
// extreme (or NaN-adjacent) curve asks for up to 2^30 subdivisions, which must not be generated first.
int HostFlatten(const pm_path_el *els, size_t n_els, size_t max_points, std::vector<double> *pts, std::vector<uint32_t> *sub_counts) {
    constexpr double kTolerance = 0.1;  // src/lib.rs:330
    bool open = false;
    double lx = 0.0, ly = 0.0;
    for (size_t i = 0; i < n_els; ++i) {
        const pm_path_el &el = els[i];
        if (el.tag == PM_EL_MOVE) {
            sub_counts->push_back(1);
            open = true;
            lx = el.p[0];
            ly = el.p[1];
            pts->push_back(lx);
            pts->push_back(ly);
        } else if (el.tag == PM_EL_LINE) {
            if (!open) return PM_ERR_INVALID;
            lx = el.p[0];
            ly = el.p[1];
            pts->push_back(lx);
            pts->push_back(ly);
            sub_counts->back() += 1;
        } else if (el.tag == PM_EL_CURVE) {
            if (!open) return PM_ERR_INVALID;
            const double accuracy = kTolerance * 1e-2;  // flatten.rs:35
            const double max_hypot2 = 432.0 * accuracy * accuracy;
            const double ax = el.p[0] * 3.0 - lx, ay = el.p[1] * 3.0 - ly;
            const double bx = el.p[2] * 3.0 - el.p[4], by = el.p[3] * 3.0 - el.p[5];
            const double dx = bx - ax, dy = by - ay;
            const uint64_t n = HostSubdivisionCount((dx * dx + dy * dy) / max_hypot2);
            if (pts->size() / 2 + n > max_points) return PM_ERR_CAPACITY;
            for (uint64_t k = 0; k < n; ++k) {
                const double t1 = static_cast<double>(k + 1) / static_cast<double>(n);
                pts->push_back(HostCubicEval(lx, el.p[0], el.p[2], el.p[4], t1));
                pts->push_back(HostCubicEval(ly, el.p[1], el.p[3], el.p[5], t1));
            }
            sub_counts->back() += static_cast<uint32_t>(n);
            lx = el.p[4];
            ly = el.p[5];
        }  // QuadTo, ClosePath: `_ => ()`, flatten.rs:40
    }
    return PM_OK;
}

}  // namespace

int pm_encoder_fill_path(pm_encoder *e, const pm_path_el *els, size_t n_els, uint32_t rgba, uint32_t fill_flags) {
    if (!e || (n_els && !els)) return PM_ERR_INVALID;
    std::vector<double> pts;
    std::vector<uint32_t> subs;
    if (const int fr = HostFlatten(els, n_els, e->enc.bytes_free() / 8, &pts, &subs)) return fr;
    const uint32_t rule = fill_flags & PM_FILL_EVEN_ODD;
    if ((fill_flags & PM_FILL_COMPOUND) && !subs.empty()) {
        e->enc.FillCompound(pts.data(), subs.data(), subs.size(), rgba, rule);
    } else {
        const double *pp = pts.data();
        for (uint32_t n : subs) {  // encode_path: one Fill per sub-path, src/lib.rs:343-347
            e->enc.Fill(pp, n, rgba, rule);
            pp += 2 * static_cast<size_t>(n);
        }
    }
    return e->enc.c_status();
}

int pm_encoder_stroke_path(pm_encoder *e, const pm_path_el *els, size_t n_els, uint32_t rgba, float width) {
    if (!e || (n_els && !els)) return PM_ERR_INVALID;
    std::vector<double> pts;
    std::vector<uint32_t> subs;
    if (const int fr = HostFlatten(els, n_els, e->enc.bytes_free() / 8, &pts, &subs)) return fr;
    constexpr float kThinLine = 0.7f;  // src/lib.rs:351
    if (width < kThinLine) {           // encode_path_stroke, src/lib.rs:353-362
        float alpha = static_cast<float>(rgba & 0xffu);
        alpha = alpha * std::sqrt(width / kThinLine);
        rgba = (rgba & ~0xffu) | static_cast<uint32_t>(alpha);
        width = kThinLine;
    }
    const double *pp = pts.data();
    for (uint32_t n : subs) {
        e->enc.Polyline(pp, n, rgba, width);
        pp += 2 * static_cast<size_t>(n);
    }
    return e->enc.c_status();
}

int pm_encoder_fill_compound(pm_encoder *e, const double *pts_xy, const uint32_t *sub_counts, size_t n_subpaths, uint32_t rgba,
                             uint32_t fill_flags) {
    if (!e || (n_subpaths && (!pts_xy || !sub_counts))) return PM_ERR_INVALID;
    e->enc.FillCompound(pts_xy, sub_counts, n_subpaths, rgba, fill_flags & PM_FILL_EVEN_ODD);
    return e->enc.c_status();
}
int pm_encoder_ellipse(pm_encoder *e, double cx, double cy, double rx, double ry) {
    if (!e) return PM_ERR_INVALID;
    e->enc.Ellipse(cx, cy, rx, ry);
    return e->enc.c_status();
}
int pm_encoder_stroke_line(pm_encoder *e, double x0, double y0, double x1, double y1, float width,
                           uint32_t rgba) {
    if (!e) return PM_ERR_INVALID;
    e->enc.StrokeLine(x0, y0, x1, y1, width, rgba);
    return e->enc.c_status();
}
int pm_encoder_fill(pm_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba) {
    if (!e || (!pts_xy && n_points)) return PM_ERR_INVALID;
    e->enc.Fill(pts_xy, n_points, rgba);
    return e->enc.c_status();
}
int pm_encoder_fill_rule(pm_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba, uint32_t fill_flags) {
    if (!e || (!pts_xy && n_points) || (fill_flags & ~PM_FILL_EVEN_ODD)) return PM_ERR_INVALID;
    e->enc.Fill(pts_xy, n_points, rgba, fill_flags);
    return e->enc.c_status();
}
int pm_encoder_polyline(pm_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba,
                        float width) {
    if (!e || (!pts_xy && n_points)) return PM_ERR_INVALID;
    e->enc.Polyline(pts_xy, n_points, rgba, width);
    return e->enc.c_status();
}
int pm_encoder_write_struct(pm_encoder *e, size_t ix, const void *src, size_t len) {
    if (!e || (!src && len)) return PM_ERR_INVALID;
    e->enc.WriteStruct(ix, src, len);
    return e->enc.c_status();
}

int pm_encoder_encode_points(pm_encoder *e, const double *pts_xy, size_t n_points, size_t *points_ix, double bbox[4]) {
    if (!e || (!pts_xy && n_points)) return PM_ERR_INVALID;
    double bb[4] = {0.0, 0.0, 0.0, 0.0};
    const size_t ix = e->enc.EncodePoints(pts_xy, n_points, bb);
    if (points_ix) *points_ix = ix;
    if (bbox) std::memcpy(bbox, bb, sizeof(bb));
    return e->enc.c_status();
}

size_t pm_encoder_bytes_used(const pm_encoder *e) { return e ? e->enc.bytes_used() : 0; }

// Developer / test hook for the generated layout code (pm_layout_gen.h): reads `scene` through
// the generated accessors and through the hand-written structs, takes every command apart with
// the generated loaders and rebuilds it with the generated packers.  0 = everything agrees,
// else the (negative) number of the first check that failed.
int pm_layout_selfcheck(const uint8_t *scene, size_t scene_len, const pm_cmd *cmds, size_t n_cmds) {
    namespace gs = pm::gen::scene;
    namespace gp = pm::gen::ptcl;
    if (scene && scene_len >= 8) {
        pm::SimpleGroup g;
        std::memcpy(&g, scene, sizeof(g));
        if (gs::SimpleGroup_n_items(scene, 0) != g.n_items || gs::SimpleGroup_items_ix(scene, 0) != g.items_ix) return -1;
        if (static_cast<uint64_t>(g.items_ix) + 32ull * g.n_items > scene_len) return -2;
        for (uint32_t i = 0; i < g.n_items; ++i) {
            const uint32_t ref = g.items_ix + i * gs::PIET_ITEM_SIZE;
            const gs::PietItem rec = gs::PietItem_read(scene, ref);
            uint32_t tag;
            std::memcpy(&tag, scene + ref, 4);
            if (gs::PietItem_tag(scene, ref) != tag || rec.tag != tag) return -3;
            pm::ShortBbox bb;
            std::memcpy(&bb, scene + 8 + 8ull * i, 8);
            const pm::gen::pm_u16x4 gb = gs::SimpleGroup_bbox(scene, 8 * i);  // element i of the box array that starts at `bbox`
            if (gb.v[0] != bb.x0 || gb.v[1] != bb.y0 || gb.v[2] != bb.x1 || gb.v[3] != bb.y1) return -4;
            gs::PietItem again;
            switch (tag & 0xffffu) {
                case gs::PietItem_Circle:
                    again = gs::PietItem_Circle_pack();
                    again.tag |= tag & pm::kCircleEllipse;  // (the ellipse bit lives in the tag word's upper half)
                    break;
                case gs::PietItem_Line: {
                    pm::PietStrokeLine h;
                    std::memcpy(&h, scene + ref, sizeof(h));
                    const gs::PietStrokeLinePacked q = gs::PietStrokeLine_load(rec);
                    if (q.flags != h.flags || q.rgba_color != h.rgba || q.width != h.width || q.start.v[0] != h.start[0] || q.start.v[1] != h.start[1] ||
                        q.end.v[0] != h.end[0] || q.end.v[1] != h.end[1] || gs::PietStrokeLine_width(scene, ref) != h.width)
                        return -5;
                    again = gs::PietItem_Line_pack(q.flags, q.rgba_color, q.width, q.start, q.end);
                    break;
                }
                case gs::PietItem_Fill: {
                    pm::PietFill h;
                    std::memcpy(&h, scene + ref, sizeof(h));
                    const gs::PietFillPacked q = gs::PietFill_load(rec);
                    if (q.flags != h.flags || q.rgba_color != h.rgba || q.n_points != h.n_points || q.points_ix != h.points_ix ||
                        gs::PietFill_points_ix(scene, ref) != h.points_ix)
                        return -6;
                    again = gs::PietItem_Fill_pack(q.flags, q.rgba_color, q.n_points, q.points_ix);
                    break;
                }
                case gs::PietItem_Poly: {
                    pm::PietStrokePolyLine h;
                    std::memcpy(&h, scene + ref, sizeof(h));
                    const gs::PietStrokePolyLinePacked q = gs::PietStrokePolyLine_load(rec);
                    if (q.rgba_color != h.rgba || q.width != h.width || q.n_points != h.n_points || q.points_ix != h.points_ix) return -7;
                    again = gs::PietItem_Poly_pack(q.rgba_color, q.width, q.n_points, q.points_ix);
                    break;
                }
                case gs::PietItem_Group: {
                    pm::PietGroup h;
                    std::memcpy(&h, scene + ref, sizeof(h));
                    const gs::PietGroupPacked q = gs::PietGroup_load(rec);
                    if (q.flags != h.flags || q.group_ix != h.group_ix) return -8;
                    again = gs::PietItem_Group_pack(q.flags, q.group_ix);
                    break;
                }
                default:
                    return -9;
            }
            // the encoder writes sizeof(T) bytes of a variant into a zeroed 32-byte slot (src/lib.rs:122-130)
            if (std::memcmp(&again, scene + ref, sizeof(again)) != 0) return -10;
        }
    }
    for (size_t k = 0; k < n_cmds; ++k) {
        gp::Cmd c;
        std::memcpy(&c, &cmds[k], sizeof(c));
        gp::Cmd again;
        switch (c.tag) {
            case gp::Cmd_End:
            case gp::Cmd_Bail:
                std::memset(&again, 0, sizeof(again));
                again.tag = c.tag;
                break;
            case gp::Cmd_Circle: { const gp::CmdCirclePacked q = gp::CmdCircle_load(c); again = gp::Cmd_Circle_pack(q.flags, q.bbox); break; }
            case gp::Cmd_Line: { const gp::CmdLinePacked q = gp::CmdLine_load(c); again = gp::Cmd_Line_pack(q.start, q.end); break; }
            case gp::Cmd_Fill: { const gp::CmdFillPacked q = gp::CmdFill_load(c); again = gp::Cmd_Fill_pack(q.start, q.end); break; }
            case gp::Cmd_Stroke: { const gp::CmdStrokePacked q = gp::CmdStroke_load(c); again = gp::Cmd_Stroke_pack(q.halfWidth, q.rgba_color, q.rg, q.ba); break; }
            case gp::Cmd_FillEdge: { const gp::CmdFillEdgePacked q = gp::CmdFillEdge_load(c); again = gp::Cmd_FillEdge_pack(q.sign, q.y); break; }
            case gp::Cmd_DrawFill: { const gp::CmdDrawFillPacked q = gp::CmdDrawFill_load(c); again = gp::Cmd_DrawFill_pack(q.backdrop, q.rgba_color, q.rg, q.ba, q.flags); break; }
            case gp::Cmd_Solid: { const gp::CmdSolidPacked q = gp::CmdSolid_load(c); again = gp::Cmd_Solid_pack(q.rgba_color, q.rg, q.ba); break; }
            default:
                return -20;
        }
        if (std::memcmp(&again, &c, sizeof(c)) != 0) return -21 - static_cast<int>(c.tag);
    }
    return 0;
}

int64_t pm_scene_cardioid(uint8_t *buf, size_t cap) {
    return buf ? pm::SceneCardioid(buf, cap) : PM_ERR_INVALID;
}
int64_t pm_scene_path_test(uint8_t *buf, size_t cap) {
    return buf ? pm::ScenePathTest(buf, cap) : PM_ERR_INVALID;
}

}  // extern "C"
