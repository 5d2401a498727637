// SVG front-end: just enough XML scanning and SVG path-data parsing to turn a
// document such as Ghostscript_Tiger.svg into the (paths, path elements) arrays
// the on-device flatten stage consumes.
//
// Replaces, for the hot path's input, what make_tiger (src/lib.rs:286-328) gets
// from roxmltree 0.6.0 and kurbo 0.5.6 `BezPath::from_svg` -- both third-party
// crates absent from the reference tree, so their behaviour is re-derived from
// the SVG 1.1 path grammar rather than copied:
//   * every <path> element is taken in document order (the Tiger keeps them all
//     in one <g>, src/lib.rs:291-295);
//   * `fill` / `stroke` attribute PRESENT => a fill / stroke item (src/lib.rs:299-304);
//   * colours through parse_color (src/lib.rs:375-385);
//   * path data: M m L l H h V v C c S s Q q T t A a Z z, implicit command
//     repetition, relative forms, smooth-curve reflection, current point reset to
//     the sub-path start after Z;
//   * elliptical arcs are converted to cubic Beziers (one per <=90 degree slice);
//     PM_SVG_REJECT_ARC_PATHS drops such paths instead (whether
//     kurbo 0.5.6 accepted arcs is not determinable offline).
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/piet_metal_amd.h"

#ifndef PM_TIGER_SVG_PATH
#error "build must define PM_TIGER_SVG_PATH (absolute path of assets/Ghostscript_Tiger.svg)"
#endif

// The reference embeds the asset with include_bytes! (src/lib.rs:288); same here.
__asm__(
    ".section .rodata\n"
    ".global pm_tiger_svg_begin\n"
    ".global pm_tiger_svg_end\n"
    "pm_tiger_svg_begin:\n"
    ".incbin \"" PM_TIGER_SVG_PATH "\"\n"
    "pm_tiger_svg_end:\n"
    ".byte 0\n"
    ".previous\n");
extern "C" const char pm_tiger_svg_begin[];
extern "C" const char pm_tiger_svg_end[];

struct pm_svg {
    std::vector<pm_path> paths;
    std::vector<pm_path_el> els;
};

namespace {

// ---- path data ------------------------------------------------------------------

class PathLexer {
public:
    PathLexer(const char *p, const char *end) : p_(p), end_(end) {}

    void SkipWs() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r' || *p_ == '\f')) ++p_;
    }
    void SkipWsComma() {
        SkipWs();
        if (p_ < end_ && *p_ == ',') {
            ++p_;
            SkipWs();
        }
    }
    bool AtEnd() {
        SkipWs();
        return p_ >= end_;
    }
    // Next command letter, or 0 if the next token is a number (implicit repeat).
    char PeekCommand() {
        SkipWs();
        if (p_ < end_ && std::isalpha(static_cast<unsigned char>(*p_)) && *p_ != 'e' && *p_ != 'E') return *p_;
        return 0;
    }
    void Advance() { ++p_; }

    bool Number(double *out) {
        SkipWsComma();
        const char *s = p_;
        const char *q = s;
        if (q < end_ && (*q == '+' || *q == '-')) ++q;
        bool digits = false;
        while (q < end_ && std::isdigit(static_cast<unsigned char>(*q))) { ++q; digits = true; }
        if (q < end_ && *q == '.') {
            ++q;
            while (q < end_ && std::isdigit(static_cast<unsigned char>(*q))) { ++q; digits = true; }
        }
        if (!digits) return false;
        if (q < end_ && (*q == 'e' || *q == 'E')) {
            const char *r = q + 1;
            if (r < end_ && (*r == '+' || *r == '-')) ++r;
            if (r < end_ && std::isdigit(static_cast<unsigned char>(*r))) {
                while (r < end_ && std::isdigit(static_cast<unsigned char>(*r))) ++r;
                q = r;
            }
        }
        std::string tok(s, q);  // strtod must not run past the token (".5.5" is two numbers)
        *out = std::strtod(tok.c_str(), nullptr);
        p_ = q;
        return true;
    }
    bool Flag(bool *out) {  // arc flags may be packed: "a1 1 0 01-2 3"
        SkipWsComma();
        if (p_ < end_ && (*p_ == '0' || *p_ == '1')) {
            *out = (*p_ == '1');
            ++p_;
            return true;
        }
        return false;
    }

private:
    const char *p_;
    const char *end_;
};

struct PathBuilder {
    std::vector<pm_path_el> *els;
    void Push(uint32_t tag, double a = 0, double b = 0, double c = 0, double d = 0, double e = 0, double f = 0) {
        pm_path_el el{};
        el.tag = tag;
        el.p[0] = a; el.p[1] = b; el.p[2] = c; el.p[3] = d; el.p[4] = e; el.p[5] = f;
        els->push_back(el);
    }
};

// SVG 1.1 implementation notes F.6: endpoint -> centre parameterisation, then one
// cubic per slice of at most 90 degrees.
void ArcToCubics(PathBuilder &out, double x0, double y0, double rx, double ry, double phi_deg,
                 bool large, bool sweep, double x1, double y1) {
    if (x0 == x1 && y0 == y1) return;  // F.6.2: omit the segment
    rx = std::fabs(rx);
    ry = std::fabs(ry);
    if (rx == 0.0 || ry == 0.0) {
        out.Push(PM_EL_LINE, x1, y1);
        return;
    }
    const double phi = phi_deg * M_PI / 180.0;
    const double cphi = std::cos(phi), sphi = std::sin(phi);
    const double dx2 = (x0 - x1) / 2.0, dy2 = (y0 - y1) / 2.0;
    const double x1p = cphi * dx2 + sphi * dy2;
    const double y1p = -sphi * dx2 + cphi * dy2;
    const double lam = (x1p * x1p) / (rx * rx) + (y1p * y1p) / (ry * ry);
    if (lam > 1.0) {
        const double s = std::sqrt(lam);
        rx *= s;
        ry *= s;
    }
    const double num = rx * rx * ry * ry - rx * rx * y1p * y1p - ry * ry * x1p * x1p;
    const double den = rx * rx * y1p * y1p + ry * ry * x1p * x1p;
    double coef = (den == 0.0) ? 0.0 : std::sqrt(std::fmax(0.0, num / den));
    if (large == sweep) coef = -coef;
    const double cxp = coef * (rx * y1p / ry);
    const double cyp = coef * -(ry * x1p / rx);
    const double cx = cphi * cxp - sphi * cyp + (x0 + x1) / 2.0;
    const double cy = sphi * cxp + cphi * cyp + (y0 + y1) / 2.0;
    auto angle = [](double ux, double uy, double vx, double vy) {
        const double dot = ux * vx + uy * vy;
        const double len = std::sqrt(ux * ux + uy * uy) * std::sqrt(vx * vx + vy * vy);
        double a = std::acos(std::fmin(1.0, std::fmax(-1.0, dot / len)));
        if (ux * vy - uy * vx < 0.0) a = -a;
        return a;
    };
    const double ux = (x1p - cxp) / rx, uy = (y1p - cyp) / ry;
    const double vx = (-x1p - cxp) / rx, vy = (-y1p - cyp) / ry;
    const double theta1 = angle(1.0, 0.0, ux, uy);
    double dtheta = angle(ux, uy, vx, vy);
    if (!sweep && dtheta > 0.0) dtheta -= 2.0 * M_PI;
    if (sweep && dtheta < 0.0) dtheta += 2.0 * M_PI;
    const int n = std::max(1, static_cast<int>(std::ceil(std::fabs(dtheta) / (M_PI / 2.0) - 1e-9)));
    const double step = dtheta / n;
    const double k = 4.0 / 3.0 * std::tan(step / 4.0);
    double th = theta1;
    for (int i = 0; i < n; ++i) {
        const double th2 = th + step;
        const double c1 = std::cos(th), s1 = std::sin(th), c2 = std::cos(th2), s2 = std::sin(th2);
        // points on the unit circle + tangents, mapped through (rx,ry), phi, centre
        const double e1x = c1 - k * s1, e1y = s1 + k * c1;
        const double e2x = c2 + k * s2, e2y = s2 - k * c2;
        auto map = [&](double ex, double ey, double *ox, double *oy) {
            *ox = cx + cphi * rx * ex - sphi * ry * ey;
            *oy = cy + sphi * rx * ex + cphi * ry * ey;
        };
        double p1x, p1y, p2x, p2y, p3x, p3y;
        map(e1x, e1y, &p1x, &p1y);
        map(e2x, e2y, &p2x, &p2y);
        if (i == n - 1) {
            p3x = x1;  // land exactly on the stated end point
            p3y = y1;
        } else {
            map(c2, s2, &p3x, &p3y);
        }
        out.Push(PM_EL_CURVE, p1x, p1y, p2x, p2y, p3x, p3y);
        th = th2;
    }
}

// Returns false on a syntax error.  *has_arc reports whether A/a occurred.
bool ParsePathData(const char *d, size_t len, std::vector<pm_path_el> *els, bool *has_arc) {
    PathLexer lx(d, d + len);
    PathBuilder out{els};
    double cur_x = 0, cur_y = 0;      // current point
    double start_x = 0, start_y = 0;  // current sub-path start
    double ctrl_x = 0, ctrl_y = 0;    // last control point (for S/s, T/t)
    bool have_ctrl = false;
    char cmd = 0;
    *has_arc = false;
    while (!lx.AtEnd()) {
        char c = lx.PeekCommand();
        if (c) {
            lx.Advance();
            cmd = c;
        } else {
            if (cmd == 0) return false;               // data must start with a command
            if (cmd == 'M') cmd = 'L';                // implicit lineto after moveto
            else if (cmd == 'm') cmd = 'l';
            else if (cmd == 'Z' || cmd == 'z') return false;
        }
        const bool rel = std::islower(static_cast<unsigned char>(cmd)) != 0;
        auto pair = [&](double *x, double *y) {
            if (!lx.Number(x) || !lx.Number(y)) return false;
            if (rel) {
                *x += cur_x;
                *y += cur_y;
            }
            return true;
        };
        switch (cmd) {
            case 'M': case 'm': {
                double x, y;
                if (!pair(&x, &y)) return false;
                out.Push(PM_EL_MOVE, x, y);
                cur_x = start_x = x;
                cur_y = start_y = y;
                ctrl_x = x; ctrl_y = y; have_ctrl = true;
                break;
            }
            case 'L': case 'l': {
                double x, y;
                if (!pair(&x, &y)) return false;
                out.Push(PM_EL_LINE, x, y);
                cur_x = x; cur_y = y;
                ctrl_x = x; ctrl_y = y; have_ctrl = true;
                break;
            }
            case 'H': case 'h': {
                double x;
                if (!lx.Number(&x)) return false;
                if (rel) x += cur_x;
                out.Push(PM_EL_LINE, x, cur_y);
                cur_x = x;
                ctrl_x = cur_x; ctrl_y = cur_y; have_ctrl = true;
                break;
            }
            case 'V': case 'v': {
                double y;
                if (!lx.Number(&y)) return false;
                if (rel) y += cur_y;
                out.Push(PM_EL_LINE, cur_x, y);
                cur_y = y;
                ctrl_x = cur_x; ctrl_y = cur_y; have_ctrl = true;
                break;
            }
            case 'C': case 'c': {
                double x1, y1, x2, y2, x3, y3;
                if (!pair(&x1, &y1) || !pair(&x2, &y2) || !pair(&x3, &y3)) return false;
                out.Push(PM_EL_CURVE, x1, y1, x2, y2, x3, y3);
                ctrl_x = x2; ctrl_y = y2; have_ctrl = true;
                cur_x = x3; cur_y = y3;
                break;
            }
            case 'S': case 's': {
                double x1 = cur_x, y1 = cur_y;
                if (have_ctrl) {
                    x1 = 2.0 * cur_x - ctrl_x;
                    y1 = 2.0 * cur_y - ctrl_y;
                }
                double x2, y2, x3, y3;
                if (!pair(&x2, &y2) || !pair(&x3, &y3)) return false;
                out.Push(PM_EL_CURVE, x1, y1, x2, y2, x3, y3);
                ctrl_x = x2; ctrl_y = y2; have_ctrl = true;
                cur_x = x3; cur_y = y3;
                break;
            }
            case 'Q': case 'q': {
                double x1, y1, x2, y2;
                if (!pair(&x1, &y1) || !pair(&x2, &y2)) return false;
                out.Push(PM_EL_QUAD, x1, y1, x2, y2);
                ctrl_x = x1; ctrl_y = y1; have_ctrl = true;
                cur_x = x2; cur_y = y2;
                break;
            }
            case 'T': case 't': {
                double x1 = cur_x, y1 = cur_y;
                if (have_ctrl) {
                    x1 = 2.0 * cur_x - ctrl_x;
                    y1 = 2.0 * cur_y - ctrl_y;
                }
                double x2, y2;
                if (!pair(&x2, &y2)) return false;
                out.Push(PM_EL_QUAD, x1, y1, x2, y2);
                ctrl_x = x1; ctrl_y = y1; have_ctrl = true;
                cur_x = x2; cur_y = y2;
                break;
            }
            case 'A': case 'a': {
                double rx, ry, rot, x, y;
                bool large, sweep;
                if (!lx.Number(&rx) || !lx.Number(&ry) || !lx.Number(&rot) || !lx.Flag(&large) ||
                    !lx.Flag(&sweep) || !pair(&x, &y))
                    return false;
                *has_arc = true;
                ArcToCubics(out, cur_x, cur_y, rx, ry, rot, large, sweep, x, y);
                cur_x = x; cur_y = y;
                ctrl_x = x; ctrl_y = y; have_ctrl = true;
                break;
            }
            case 'Z': case 'z':
                out.Push(PM_EL_CLOSE);
                cur_x = start_x; cur_y = start_y;
                ctrl_x = cur_x; ctrl_y = cur_y; have_ctrl = true;
                break;
            default:
                return false;
        }
    }
    return true;
}

// ---- XML scan ------------------------------------------------------------------------

struct Attr {
    const char *name;
    size_t name_len;
    const char *val;
    size_t val_len;
};

// Parses the attributes of a start tag whose body is [p, end).  Returns false on junk.
bool ScanAttrs(const char *p, const char *end, std::vector<Attr> *out) {
    out->clear();
    while (p < end) {
        while (p < end && std::isspace(static_cast<unsigned char>(*p))) ++p;
        if (p >= end || *p == '/') break;
        const char *n0 = p;
        while (p < end && !std::isspace(static_cast<unsigned char>(*p)) && *p != '=' && *p != '/') ++p;
        const char *n1 = p;
        while (p < end && std::isspace(static_cast<unsigned char>(*p))) ++p;
        if (p >= end || *p != '=') return n1 == n0;  // bare token: tolerate nothing else
        ++p;
        while (p < end && std::isspace(static_cast<unsigned char>(*p))) ++p;
        if (p >= end || (*p != '"' && *p != '\'')) return false;
        const char q = *p++;
        const char *v0 = p;
        while (p < end && *p != q) ++p;
        if (p >= end) return false;
        out->push_back({n0, static_cast<size_t>(n1 - n0), v0, static_cast<size_t>(p - v0)});
        ++p;
    }
    return true;
}

const Attr *Find(const std::vector<Attr> &attrs, const char *name) {
    const size_t n = std::strlen(name);
    for (const Attr &a : attrs)
        if (a.name_len == n && std::memcmp(a.name, name, n) == 0) return &a;
    return nullptr;
}

uint32_t ParseColor(const char *s, size_t len) {  // parse_color, src/lib.rs:375-385
    if (len >= 1 && s[0] == '#') {
        uint32_t hex = 0;
        for (size_t i = 1; i < len; ++i) {
            const char ch = s[i];
            uint32_t d;
            if (ch >= '0' && ch <= '9') d = ch - '0';
            else if (ch >= 'a' && ch <= 'f') d = ch - 'a' + 10;
            else if (ch >= 'A' && ch <= 'F') d = ch - 'A' + 10;
            else return 0xff00ff80u;  // reference would panic on from_str_radix
            hex = (hex << 4) | d;
        }
        if (len == 4) hex = (hex >> 8) * 0x110000u + ((hex >> 4) & 0xfu) * 0x1100u + (hex & 0xfu) * 0x11u;
        return (hex << 8) + 0xffu;
    }
    return 0xff00ff80u;
}

int ParseDocument(const char *text, size_t len, int flags, pm_svg *out) {
    const char *p = text;
    const char *end = text + len;
    std::vector<Attr> attrs;
    while (p < end) {
        const char *lt = static_cast<const char *>(std::memchr(p, '<', end - p));
        if (!lt) break;
        p = lt + 1;
        if (p >= end) break;
        if (end - p >= 3 && std::memcmp(p, "!--", 3) == 0) {  // comment
            const char *q = p + 3;
            while (q + 2 < end && std::memcmp(q, "-->", 3) != 0) ++q;
            p = (q + 3 <= end) ? q + 3 : end;
            continue;
        }
        if (*p == '?' || *p == '!' || *p == '/') {  // PI, doctype, end tag
            const char *gt = static_cast<const char *>(std::memchr(p, '>', end - p));
            p = gt ? gt + 1 : end;
            continue;
        }
        const char *n0 = p;
        while (p < end && !std::isspace(static_cast<unsigned char>(*p)) && *p != '>' && *p != '/') ++p;
        const size_t name_len = p - n0;
        // find the closing '>' outside quotes
        const char *q = p;
        char quote = 0;
        while (q < end && (quote || *q != '>')) {
            if (quote) {
                if (*q == quote) quote = 0;
            } else if (*q == '"' || *q == '\'') {
                quote = *q;
            }
            ++q;
        }
        if (q >= end) return PM_ERR_PARSE;
        if (name_len == 4 && std::memcmp(n0, "path", 4) == 0) {
            if (!ScanAttrs(p, q, &attrs)) return PM_ERR_PARSE;
            const Attr *d = Find(attrs, "d");
            if (!d) return PM_ERR_PARSE;  // .attribute("d").unwrap(), src/lib.rs:295
            const size_t el0 = out->els.size();
            bool has_arc = false;
            const bool ok = ParsePathData(d->val, d->val_len, &out->els, &has_arc);
            if (!ok || (has_arc && (flags & PM_SVG_REJECT_ARC_PATHS))) {
                out->els.resize(el0);  // `if let Ok(ref bp) = ...` skips the path, src/lib.rs:296
            } else {
                pm_path path{};
                path.el_begin = static_cast<uint32_t>(el0);
                path.el_end = static_cast<uint32_t>(out->els.size());
                if (const Attr *f = Find(attrs, "fill")) {
                    path.flags |= PM_PATH_FILL;
                    path.fill_rgba = ParseColor(f->val, f->val_len);
                }
                if (const Attr *s = Find(attrs, "stroke")) {
                    path.flags |= PM_PATH_STROKE;
                    path.stroke_rgba = ParseColor(s->val, s->val_len);
                    path.stroke_width = 1.0f;
                    if (const Attr *w = Find(attrs, "stroke-width")) {
                        std::string tok(w->val, w->val_len);
                        path.stroke_width = std::strtof(tok.c_str(), nullptr);  // f32::from_str
                    }
                }
                out->paths.push_back(path);
            }
        }
        p = q + 1;
    }
    return PM_OK;
}

}  // namespace

extern "C" {

pm_svg *pm_svg_parse(const char *text, size_t len, int flags, int *err) {
    int dummy;
    if (!err) err = &dummy;
    if (!text) {
        *err = PM_ERR_INVALID;
        return nullptr;
    }
    pm_svg *s = new (std::nothrow) pm_svg();
    if (!s) {
        *err = PM_ERR_CAPACITY;
        return nullptr;
    }
    *err = ParseDocument(text, len, flags, s);
    if (*err != PM_OK) {
        delete s;
        return nullptr;
    }
    return s;
}

pm_svg *pm_svg_tiger(int flags, int *err) {
    return pm_svg_parse(pm_tiger_svg_begin, static_cast<size_t>(pm_tiger_svg_end - pm_tiger_svg_begin), flags, err);
}

void pm_svg_free(pm_svg *s) { delete s; }
size_t pm_svg_n_paths(const pm_svg *s) { return s ? s->paths.size() : 0; }
size_t pm_svg_n_els(const pm_svg *s) { return s ? s->els.size() : 0; }
const pm_path *pm_svg_paths(const pm_svg *s) { return s ? s->paths.data() : nullptr; }
const pm_path_el *pm_svg_els(const pm_svg *s) { return s ? s->els.data() : nullptr; }
uint32_t pm_parse_color(const char *s) { return s ? ParseColor(s, std::strlen(s)) : 0xff00ff80u; }

}  // extern "C"
