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
    // the outermost <svg>: viewBox (if any), width / height in user units (0: absent or a percentage)
    bool has_viewbox = false, seen_root = false;
    double viewbox[4] = {0, 0, 0, 0};
    double width = 0, height = 0;
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

// ---- the document: element tree, inherited presentation state -------------------------------
// make_tiger reads d / fill / stroke / stroke-width of every <path> and nothing else
// (src/lib.rs:291-326).  That is this front-end on the Tiger, byte for byte; on other documents
// it goes on where the reference stops: <g> nesting with inherited properties, `transform`,
// `style="..."`, opacity / fill-opacity / stroke-opacity, fill-rule, rgb() and the basic colour
// names, and the basic shapes (rect, circle, ellipse, line, polyline, polygon) as paths.
// <use> draws the element (or <symbol>) its href names, wherever that is defined.
// Not understood (ignored): patterns and -- unless PM_SVG_FLAT_GRADIENTS turns them into their mean
// colour -- gradients (painted as if `none`), clipping, masks,
// text, CSS selectors beyond element / .class / #id, units other than user units / px, stroke
// joins / caps / dashes.

struct Affine {  // x' = a x + c y + e, y' = b x + d y + f (the SVG matrix(a b c d e f))
    double a = 1, b = 0, c = 0, d = 1, e = 0, f = 0;
    Affine Then(const Affine &m) const {  // this * m: m is applied first
        return {a * m.a + c * m.b, b * m.a + d * m.b, a * m.c + c * m.d, b * m.c + d * m.d, a * m.e + c * m.f + e, b * m.e + d * m.f + f};
    }
    void Apply(double *x, double *y) const {
        const double nx = a * *x + c * *y + e, ny = b * *x + d * *y + f;
        *x = nx;
        *y = ny;
    }
    bool IsIdentity() const { return a == 1 && b == 0 && c == 0 && d == 1 && e == 0 && f == 0; }
};

struct Paint {
    bool none = true;
    uint32_t rgb = 0;  // 0xRRGGBB
};

struct Style {
    Paint fill, stroke;
    float stroke_width = 1.0f;
    double fill_server_alpha = 1.0, stroke_server_alpha = 1.0;  // mean stop-opacity of a flattened gradient paint
    double opacity = 1.0, fill_opacity = 1.0, stroke_opacity = 1.0;  // opacity: product of the ancestors'
    bool even_odd = false;
    bool display_none = false;  // `display: none` on the element or an ancestor: nothing of the subtree is drawn
    bool hidden = false;        // `visibility: hidden | collapse` (inherited; a descendant may turn it back on)
    Affine ctm;
};

std::string Trim(const char *s, size_t n) {
    while (n && std::isspace(static_cast<unsigned char>(*s))) ++s, --n;
    while (n && std::isspace(static_cast<unsigned char>(s[n - 1]))) --n;
    return std::string(s, n);
}

// url(#id) paints: with PM_SVG_FLAT_GRADIENTS the document layer installs a resolver that turns a
// gradient into ONE colour (the mean of its stops) -- the renderer has no gradients, and a flat
// stand-in is closer to the picture than nothing.  Without it a url() paint is `none`.
struct PaintServerResolver {
    virtual bool Resolve(const std::string &id, uint32_t *rgb, double *alpha) const = 0;
    virtual ~PaintServerResolver() = default;
};
thread_local const PaintServerResolver *t_paint_servers = nullptr;

bool ParsePaint(const std::string &v, Paint *out, double *server_alpha = nullptr) {
    // the 147 colour keywords of SVG 1.1 (section 4.4)
    static const struct { const char *name; uint32_t rgb; } kNames[] = {
        {"aliceblue", 0xf0f8ff}, {"antiquewhite", 0xfaebd7}, {"aqua", 0x00ffff}, {"aquamarine", 0x7fffd4}, {"azure", 0xf0ffff}, {"beige", 0xf5f5dc},
        {"bisque", 0xffe4c4}, {"black", 0x000000}, {"blanchedalmond", 0xffebcd}, {"blue", 0x0000ff}, {"blueviolet", 0x8a2be2}, {"brown", 0xa52a2a},
        {"burlywood", 0xdeb887}, {"cadetblue", 0x5f9ea0}, {"chartreuse", 0x7fff00}, {"chocolate", 0xd2691e}, {"coral", 0xff7f50},
        {"cornflowerblue", 0x6495ed}, {"cornsilk", 0xfff8dc}, {"crimson", 0xdc143c}, {"cyan", 0x00ffff}, {"darkblue", 0x00008b},
        {"darkcyan", 0x008b8b}, {"darkgoldenrod", 0xb8860b}, {"darkgray", 0xa9a9a9}, {"darkgreen", 0x006400}, {"darkgrey", 0xa9a9a9},
        {"darkkhaki", 0xbdb76b}, {"darkmagenta", 0x8b008b}, {"darkolivegreen", 0x556b2f}, {"darkorange", 0xff8c00}, {"darkorchid", 0x9932cc},
        {"darkred", 0x8b0000}, {"darksalmon", 0xe9967a}, {"darkseagreen", 0x8fbc8f}, {"darkslateblue", 0x483d8b}, {"darkslategray", 0x2f4f4f},
        {"darkslategrey", 0x2f4f4f}, {"darkturquoise", 0x00ced1}, {"darkviolet", 0x9400d3}, {"deeppink", 0xff1493}, {"deepskyblue", 0x00bfff},
        {"dimgray", 0x696969}, {"dimgrey", 0x696969}, {"dodgerblue", 0x1e90ff}, {"firebrick", 0xb22222}, {"floralwhite", 0xfffaf0},
        {"forestgreen", 0x228b22}, {"fuchsia", 0xff00ff}, {"gainsboro", 0xdcdcdc}, {"ghostwhite", 0xf8f8ff}, {"gold", 0xffd700},
        {"goldenrod", 0xdaa520}, {"gray", 0x808080}, {"grey", 0x808080}, {"green", 0x008000}, {"greenyellow", 0xadff2f}, {"honeydew", 0xf0fff0},
        {"hotpink", 0xff69b4}, {"indianred", 0xcd5c5c}, {"indigo", 0x4b0082}, {"ivory", 0xfffff0}, {"khaki", 0xf0e68c}, {"lavender", 0xe6e6fa},
        {"lavenderblush", 0xfff0f5}, {"lawngreen", 0x7cfc00}, {"lemonchiffon", 0xfffacd}, {"lightblue", 0xadd8e6}, {"lightcoral", 0xf08080},
        {"lightcyan", 0xe0ffff}, {"lightgoldenrodyellow", 0xfafad2}, {"lightgray", 0xd3d3d3}, {"lightgreen", 0x90ee90}, {"lightgrey", 0xd3d3d3},
        {"lightpink", 0xffb6c1}, {"lightsalmon", 0xffa07a}, {"lightseagreen", 0x20b2aa}, {"lightskyblue", 0x87cefa}, {"lightslategray", 0x778899},
        {"lightslategrey", 0x778899}, {"lightsteelblue", 0xb0c4de}, {"lightyellow", 0xffffe0}, {"lime", 0x00ff00}, {"limegreen", 0x32cd32},
        {"linen", 0xfaf0e6}, {"magenta", 0xff00ff}, {"maroon", 0x800000}, {"mediumaquamarine", 0x66cdaa}, {"mediumblue", 0x0000cd},
        {"mediumorchid", 0xba55d3}, {"mediumpurple", 0x9370db}, {"mediumseagreen", 0x3cb371}, {"mediumslateblue", 0x7b68ee},
        {"mediumspringgreen", 0x00fa9a}, {"mediumturquoise", 0x48d1cc}, {"mediumvioletred", 0xc71585}, {"midnightblue", 0x191970},
        {"mintcream", 0xf5fffa}, {"mistyrose", 0xffe4e1}, {"moccasin", 0xffe4b5}, {"navajowhite", 0xffdead}, {"navy", 0x000080},
        {"oldlace", 0xfdf5e6}, {"olive", 0x808000}, {"olivedrab", 0x6b8e23}, {"orange", 0xffa500}, {"orangered", 0xff4500}, {"orchid", 0xda70d6},
        {"palegoldenrod", 0xeee8aa}, {"palegreen", 0x98fb98}, {"paleturquoise", 0xafeeee}, {"palevioletred", 0xdb7093}, {"papayawhip", 0xffefd5},
        {"peachpuff", 0xffdab9}, {"peru", 0xcd853f}, {"pink", 0xffc0cb}, {"plum", 0xdda0dd}, {"powderblue", 0xb0e0e6}, {"purple", 0x800080},
        {"red", 0xff0000}, {"rosybrown", 0xbc8f8f}, {"royalblue", 0x4169e1}, {"saddlebrown", 0x8b4513}, {"salmon", 0xfa8072},
        {"sandybrown", 0xf4a460}, {"seagreen", 0x2e8b57}, {"seashell", 0xfff5ee}, {"sienna", 0xa0522d}, {"silver", 0xc0c0c0}, {"skyblue", 0x87ceeb},
        {"slateblue", 0x6a5acd}, {"slategray", 0x708090}, {"slategrey", 0x708090}, {"snow", 0xfffafa}, {"springgreen", 0x00ff7f},
        {"steelblue", 0x4682b4}, {"tan", 0xd2b48c}, {"teal", 0x008080}, {"thistle", 0xd8bfd8}, {"tomato", 0xff6347}, {"turquoise", 0x40e0d0},
        {"violet", 0xee82ee}, {"wheat", 0xf5deb3}, {"white", 0xffffff}, {"whitesmoke", 0xf5f5f5}, {"yellow", 0xffff00}, {"yellowgreen", 0x9acd32},
    };
    if (v.empty() || v == "inherit" || v == "currentColor") return false;  // leave the inherited value
    if (v.compare(0, 4, "url(") == 0 && t_paint_servers) {
        size_t a = v.find('#'), b = v.find(')');
        if (a != std::string::npos && b != std::string::npos && b > a + 1) {
            std::string id = Trim(v.c_str() + a + 1, b - a - 1);
            if (!id.empty() && (id.back() == '"' || id.back() == '\'')) id.pop_back();
            uint32_t rgb = 0;
            double alpha = 1.0;
            if (t_paint_servers->Resolve(id, &rgb, &alpha)) {
                out->none = false;
                out->rgb = rgb;
                if (server_alpha) *server_alpha = alpha;
                return true;
            }
        }
    }
    if (v == "none" || v == "transparent" || v.compare(0, 4, "url(") == 0) {
        out->none = true;
        return true;
    }
    if (v[0] == '#') {
        if (v.size() != 4 && v.size() != 7) return false;
        const uint32_t c = ParseColor(v.c_str(), v.size());
        if (c == 0xff00ff80u) return false;
        out->none = false;
        out->rgb = c >> 8;
        return true;
    }
    const bool is_rgb = v.compare(0, 4, "rgb(") == 0, is_rgba = v.compare(0, 5, "rgba(") == 0;
    if (is_rgb || is_rgba) {  // CSS colour functions; an alpha channel folds into the item's alpha
        double ch[4] = {0, 0, 0, 1};
        const char *p = v.c_str() + (is_rgba ? 5 : 4);
        for (int i = 0; i < (is_rgba ? 4 : 3); ++i) {
            char *q = nullptr;
            ch[i] = std::strtod(p, &q);
            if (q == p) return false;
            p = q;
            while (std::isspace(static_cast<unsigned char>(*p))) ++p;
            if (*p == '%') {
                ch[i] = i < 3 ? ch[i] * 255.0 / 100.0 : ch[i] / 100.0;
                ++p;
            }
            while (*p == ',' || *p == '/' || std::isspace(static_cast<unsigned char>(*p))) ++p;
        }
        uint32_t rgb = 0;
        for (int i = 0; i < 3; ++i) rgb = (rgb << 8) | static_cast<uint32_t>(std::lround(std::fmin(255.0, std::fmax(0.0, ch[i]))));
        out->none = false;
        out->rgb = rgb;
        if (server_alpha) *server_alpha = std::fmin(1.0, std::fmax(0.0, ch[3]));
        return true;
    }
    const bool is_hsl = v.compare(0, 4, "hsl(") == 0, is_hsla = v.compare(0, 5, "hsla(") == 0;
    if (is_hsl || is_hsla) {  // CSS Color 3, section 4.2.4
        double ch[4] = {0, 0, 0, 1};
        const char *p = v.c_str() + (is_hsla ? 5 : 4);
        for (int i = 0; i < (is_hsla ? 4 : 3); ++i) {
            char *q = nullptr;
            ch[i] = std::strtod(p, &q);
            if (q == p) return false;
            p = q;
            while (*p && (std::isalpha(static_cast<unsigned char>(*p)) || std::isspace(static_cast<unsigned char>(*p)))) ++p;  // "deg"
            if (*p == '%') {
                if (i == 3) ch[i] /= 100.0;
                ++p;
            }
            while (*p == ',' || *p == '/' || std::isspace(static_cast<unsigned char>(*p))) ++p;
        }
        const double h = std::fmod(std::fmod(ch[0], 360.0) + 360.0, 360.0) / 360.0;
        const double sat = std::fmin(1.0, std::fmax(0.0, ch[1] / 100.0)), l = std::fmin(1.0, std::fmax(0.0, ch[2] / 100.0));
        const double m2 = l <= 0.5 ? l * (sat + 1.0) : l + sat - l * sat, m1 = l * 2.0 - m2;
        auto hue = [&](double t) {
            if (t < 0) t += 1;
            if (t > 1) t -= 1;
            if (t * 6 < 1) return m1 + (m2 - m1) * t * 6;
            if (t * 2 < 1) return m2;
            if (t * 3 < 2) return m1 + (m2 - m1) * (2.0 / 3.0 - t) * 6;
            return m1;
        };
        const double rgbf[3] = {hue(h + 1.0 / 3.0), hue(h), hue(h - 1.0 / 3.0)};
        uint32_t rgb = 0;
        for (int i = 0; i < 3; ++i) rgb = (rgb << 8) | static_cast<uint32_t>(std::lround(255.0 * rgbf[i]));
        out->none = false;
        out->rgb = rgb;
        if (server_alpha) *server_alpha = std::fmin(1.0, std::fmax(0.0, ch[3]));
        return true;
    }
    for (const auto &n : kNames)
        if (v == n.name) {
            out->none = false;
            out->rgb = n.rgb;
            return true;
        }
    return false;
}

// transform="matrix(..) translate(..) scale(..) rotate(..) skewX(..) skewY(..)", left to right
bool ParseTransform(const std::string &v, Affine *out) {
    Affine m;
    const char *p = v.c_str();
    while (*p) {
        while (std::isspace(static_cast<unsigned char>(*p)) || *p == ',') ++p;
        if (!*p) break;
        const char *n0 = p;
        while (std::isalpha(static_cast<unsigned char>(*p))) ++p;
        const std::string name(n0, p);
        while (std::isspace(static_cast<unsigned char>(*p))) ++p;
        if (*p != '(') return false;
        ++p;
        double arg[6];
        int n = 0;
        while (*p && *p != ')') {
            char *q = nullptr;
            const double x = std::strtod(p, &q);
            if (q == p || n >= 6) return false;
            arg[n++] = x;
            p = q;
            while (std::isspace(static_cast<unsigned char>(*p)) || *p == ',') ++p;
        }
        if (*p != ')') return false;
        ++p;
        Affine t;
        if (name == "matrix" && n == 6) {
            t = {arg[0], arg[1], arg[2], arg[3], arg[4], arg[5]};
        } else if (name == "translate" && (n == 1 || n == 2)) {
            t.e = arg[0];
            t.f = n == 2 ? arg[1] : 0.0;
        } else if (name == "scale" && (n == 1 || n == 2)) {
            t.a = arg[0];
            t.d = n == 2 ? arg[1] : arg[0];
        } else if (name == "rotate" && (n == 1 || n == 3)) {
            const double th = arg[0] * M_PI / 180.0, cs = std::cos(th), sn = std::sin(th);
            Affine r{cs, sn, -sn, cs, 0, 0};
            if (n == 3) {
                Affine to{1, 0, 0, 1, arg[1], arg[2]}, back{1, 0, 0, 1, -arg[1], -arg[2]};
                r = to.Then(r).Then(back);
            }
            t = r;
        } else if (name == "skewX" && n == 1) {
            t.c = std::tan(arg[0] * M_PI / 180.0);
        } else if (name == "skewY" && n == 1) {
            t.b = std::tan(arg[0] * M_PI / 180.0);
        } else {
            return false;
        }
        m = m.Then(t);
    }
    *out = m;
    return true;
}

double ParseOpacity(const std::string &v, double dflt) {
    char *q = nullptr;
    double x = std::strtod(v.c_str(), &q);
    if (q == v.c_str()) return dflt;
    if (*q == '%') x /= 100.0;
    return std::fmin(1.0, std::fmax(0.0, x));
}

// A length (SVG 1.1 section 7.10) in user units: px and bare numbers as they are, the absolute units
// at 96 per inch, em / ex as 16 / 8 (there is no font), a percentage of `ref` (the viewBox width,
// height or normalised diagonal the caller passes; 0 when the document gives none).
double LengthFromString(const std::string &v, double dflt, double ref) {
    char *q = nullptr;
    const double x = std::strtod(v.c_str(), &q);
    if (q == v.c_str()) return dflt;
    while (*q && std::isspace(static_cast<unsigned char>(*q))) ++q;
    const std::string unit = Trim(q, std::strlen(q));
    if (unit.empty() || unit == "px") return x;
    if (unit == "%") return x * ref / 100.0;
    if (unit == "pt") return x * (96.0 / 72.0);
    if (unit == "pc") return x * 16.0;
    if (unit == "mm") return x * (96.0 / 25.4);
    if (unit == "cm") return x * (96.0 / 2.54);
    if (unit == "in") return x * 96.0;
    if (unit == "em") return x * 16.0;
    if (unit == "ex") return x * 8.0;
    return x;
}

double ParseLength(const Attr *a, double dflt = 0.0, double ref = 0.0) {
    if (!a) return dflt;
    return LengthFromString(std::string(a->val, a->val_len), dflt, ref);
}

// One presentation property, from an attribute or a `style` declaration (which wins, CSS cascade).
void ApplyProperty(const std::string &name, const std::string &value, Style *st) {
    if (name == "fill") {
        double a = 1.0;
        if (ParsePaint(value, &st->fill, &a)) st->fill_server_alpha = a;
    } else if (name == "stroke") {
        double a = 1.0;
        if (ParsePaint(value, &st->stroke, &a)) st->stroke_server_alpha = a;
    } else if (name == "stroke-width") {
        // f32::from_str, src/lib.rs:320: a bare number is parsed as binary32 (no double rounding); only
        // a value that carries a unit goes through the conversion
        char *rest = nullptr;
        const float plain = std::strtof(value.c_str(), &rest);
        while (rest && *rest && std::isspace(static_cast<unsigned char>(*rest))) ++rest;
        st->stroke_width = (rest && *rest) ? static_cast<float>(LengthFromString(value, 0.0, 0.0)) : plain;
    } else if (name == "fill-rule") {
        if (value == "evenodd") st->even_odd = true;
        else if (value == "nonzero") st->even_odd = false;
    } else if (name == "display") {
        if (value == "none") st->display_none = true;  // (sticky: descendants cannot undo it, SVG 1.1 section 11.5)
    } else if (name == "visibility") {
        if (value == "hidden" || value == "collapse") st->hidden = true;
        else if (value == "visible") st->hidden = false;
    } else if (name == "fill-opacity") {
        st->fill_opacity = ParseOpacity(value, st->fill_opacity);
    } else if (name == "stroke-opacity") {
        st->stroke_opacity = ParseOpacity(value, st->stroke_opacity);
    }
}

// A <style> sheet, as far as this front-end reads one: rules whose selector is a bare element name,
// .class or #id (comma lists allowed; anything else -- combinators, attributes, pseudo classes, @rules
// -- is skipped).  Cascade: presentation attributes < element rules < class rules < id rules < the
// style attribute (CSS 2.1 section 6.4.3 for these selectors; later rules win within a kind).
struct CssRule {
    int kind;  // 0 element, 1 class, 2 id
    std::string name, decls;
};

void ApplyDeclarations(const char *p, const char *end, Style *st, double *opacity_factor) {
    while (p < end) {
        const char *semi = static_cast<const char *>(std::memchr(p, ';', end - p));
        const char *de = semi ? semi : end;
        const char *colon = static_cast<const char *>(std::memchr(p, ':', de - p));
        if (colon) {
            const std::string name = Trim(p, colon - p), value = Trim(colon + 1, de - colon - 1);
            if (name == "opacity") *opacity_factor = ParseOpacity(value, 1.0);  // (the last declaration wins, then it multiplies once)
            else ApplyProperty(name, value, st);
        }
        p = de + 1;
    }
}

void ParseStyleSheet(const char *p, const char *end, std::vector<CssRule> *rules) {
    std::string css;
    while (p < end) {  // drop comments and CDATA markers
        if (end - p >= 2 && p[0] == '/' && p[1] == '*') {
            const char *q = p + 2;
            while (q + 1 < end && !(q[0] == '*' && q[1] == '/')) ++q;
            p = q + 2 <= end ? q + 2 : end;
        } else if (end - p >= 9 && std::memcmp(p, "<![CDATA[", 9) == 0) {
            p += 9;
        } else if (end - p >= 3 && std::memcmp(p, "]]>", 3) == 0) {
            p += 3;
        } else {
            css.push_back(*p++);
        }
    }
    size_t at = 0;
    while (at < css.size()) {
        const size_t open = css.find('{', at);
        if (open == std::string::npos) break;
        const size_t close = css.find('}', open);
        if (close == std::string::npos) break;
        const std::string sel = css.substr(at, open - at), decls = css.substr(open + 1, close - open - 1);
        at = close + 1;
        size_t s0 = 0;
        while (s0 <= sel.size()) {
            size_t comma = sel.find(',', s0);
            if (comma == std::string::npos) comma = sel.size();
            const std::string one = Trim(sel.c_str() + s0, comma - s0);
            s0 = comma + 1;
            if (one.empty()) continue;
            const int kind = one[0] == '.' ? 1 : (one[0] == '#' ? 2 : 0);
            const std::string name = kind ? one.substr(1) : one;
            bool simple = !name.empty();
            for (char ch : name)
                if (!(std::isalnum(static_cast<unsigned char>(ch)) || ch == '-' || ch == '_')) simple = false;
            if (simple) rules->push_back({kind, name, decls});
        }
    }
}

bool ApplyElementStyle(const std::vector<Attr> &attrs, Style *st, const std::vector<CssRule> *css = nullptr, const char *el_name = nullptr,
                       size_t el_name_len = 0) {
    static const char *kProps[] = {"fill", "stroke", "stroke-width", "fill-rule", "fill-opacity", "stroke-opacity", "display", "visibility"};
    for (const char *pn : kProps)
        if (const Attr *a = Find(attrs, pn)) ApplyProperty(pn, Trim(a->val, a->val_len), st);
    double opacity_factor = 1.0;  // not inherited: the element's own value multiplies the ancestors'
    if (const Attr *a = Find(attrs, "opacity")) opacity_factor = ParseOpacity(Trim(a->val, a->val_len), 1.0);
    if (css && !css->empty()) {
        const Attr *cls = Find(attrs, "class"), *id = Find(attrs, "id");
        for (int kind = 0; kind < 3; ++kind)
            for (const CssRule &r : *css) {
                if (r.kind != kind) continue;
                bool match = false;
                if (kind == 0) {
                    match = el_name && r.name.size() == el_name_len && std::memcmp(r.name.data(), el_name, el_name_len) == 0;
                } else if (kind == 2) {
                    match = id && r.name.size() == id->val_len && std::memcmp(r.name.data(), id->val, id->val_len) == 0;
                } else if (cls) {
                    const char *c = cls->val, *ce = cls->val + cls->val_len;
                    while (c < ce && !match) {
                        while (c < ce && std::isspace(static_cast<unsigned char>(*c))) ++c;
                        const char *w = c;
                        while (c < ce && !std::isspace(static_cast<unsigned char>(*c))) ++c;
                        match = static_cast<size_t>(c - w) == r.name.size() && std::memcmp(w, r.name.data(), r.name.size()) == 0;
                    }
                }
                if (match) ApplyDeclarations(r.decls.data(), r.decls.data() + r.decls.size(), st, &opacity_factor);
            }
    }
    if (const Attr *a = Find(attrs, "style")) ApplyDeclarations(a->val, a->val + a->val_len, st, &opacity_factor);
    st->opacity *= opacity_factor;
    if (const Attr *a = Find(attrs, "transform")) {
        Affine t;
        if (!ParseTransform(std::string(a->val, a->val_len), &t)) return false;
        st->ctm = st->ctm.Then(t);
    }
    return true;
}

uint32_t PaintRgba(const Paint &p, double opacity) {
    const uint32_t a = static_cast<uint32_t>(std::lround(255.0 * std::fmin(1.0, std::fmax(0.0, opacity))));
    return (p.rgb << 8) | a;
}

// elements [el0, end) through the current transformation matrix (an affine map of a Bezier's
// control points is the map of the curve)
void TransformEls(std::vector<pm_path_el> *els, size_t el0, const Affine &m) {
    if (m.IsIdentity()) return;
    for (size_t i = el0; i < els->size(); ++i) {
        pm_path_el &e = (*els)[i];
        const int npt = e.tag == PM_EL_CURVE ? 3 : (e.tag == PM_EL_QUAD ? 2 : (e.tag == PM_EL_CLOSE ? 0 : 1));
        for (int k = 0; k < npt; ++k) m.Apply(&e.p[2 * k], &e.p[2 * k + 1]);
    }
}

// basic shapes as path elements (SVG 1.1 section 9); false = the shape renders nothing
// (vw, vh: what percentages refer to -- the outermost viewBox, or the document's width / height)
bool ShapeToEls(const char *name, size_t name_len, const std::vector<Attr> &attrs, double vw, double vh, std::vector<pm_path_el> *els,
                bool *closed) {
    const double vd = std::sqrt((vw * vw + vh * vh) / 2.0);  // 7.10: percentages of "other" lengths
    auto X = [&](const char *k, double dflt = 0.0) { return ParseLength(Find(attrs, k), dflt, vw); };
    auto Y = [&](const char *k, double dflt = 0.0) { return ParseLength(Find(attrs, k), dflt, vh); };
    PathBuilder out{els};
    const std::string n(name, name_len);
    *closed = true;
    auto ellipse = [&](double cx, double cy, double rx, double ry) {
        if (!(rx > 0.0) || !(ry > 0.0)) return false;
        out.Push(PM_EL_MOVE, cx + rx, cy);  // four quarter arcs, as 9.3 / 9.4 prescribe
        ArcToCubics(out, cx + rx, cy, rx, ry, 0.0, false, true, cx, cy + ry);
        ArcToCubics(out, cx, cy + ry, rx, ry, 0.0, false, true, cx - rx, cy);
        ArcToCubics(out, cx - rx, cy, rx, ry, 0.0, false, true, cx, cy - ry);
        ArcToCubics(out, cx, cy - ry, rx, ry, 0.0, false, true, cx + rx, cy);
        out.Push(PM_EL_CLOSE);
        return true;
    };
    if (n == "rect") {
        const double x = X("x"), y = Y("y");
        const double w = X("width"), h = Y("height");
        if (!(w > 0.0) || !(h > 0.0)) return false;
        const Attr *arx = Find(attrs, "rx"), *ary = Find(attrs, "ry");
        double rx = ParseLength(arx, -1.0, vw), ry = ParseLength(ary, -1.0, vh);
        if (rx < 0.0 && ry < 0.0) rx = ry = 0.0;
        else if (rx < 0.0) rx = ry;
        else if (ry < 0.0) ry = rx;
        rx = std::fmin(rx, w / 2.0);
        ry = std::fmin(ry, h / 2.0);
        if (rx == 0.0 || ry == 0.0) {
            out.Push(PM_EL_MOVE, x, y);
            out.Push(PM_EL_LINE, x + w, y);
            out.Push(PM_EL_LINE, x + w, y + h);
            out.Push(PM_EL_LINE, x, y + h);
            out.Push(PM_EL_CLOSE);
        } else {  // 9.2: the rounded outline
            out.Push(PM_EL_MOVE, x + rx, y);
            out.Push(PM_EL_LINE, x + w - rx, y);
            ArcToCubics(out, x + w - rx, y, rx, ry, 0.0, false, true, x + w, y + ry);
            out.Push(PM_EL_LINE, x + w, y + h - ry);
            ArcToCubics(out, x + w, y + h - ry, rx, ry, 0.0, false, true, x + w - rx, y + h);
            out.Push(PM_EL_LINE, x + rx, y + h);
            ArcToCubics(out, x + rx, y + h, rx, ry, 0.0, false, true, x, y + h - ry);
            out.Push(PM_EL_LINE, x, y + ry);
            ArcToCubics(out, x, y + ry, rx, ry, 0.0, false, true, x + rx, y);
            out.Push(PM_EL_CLOSE);
        }
        return true;
    }
    if (n == "circle") {
        const double r = ParseLength(Find(attrs, "r"), 0.0, vd);
        return ellipse(X("cx"), Y("cy"), r, r);
    }
    if (n == "ellipse")
        return ellipse(X("cx"), Y("cy"), X("rx"), Y("ry"));
    if (n == "line") {
        *closed = false;
        out.Push(PM_EL_MOVE, X("x1"), Y("y1"));
        out.Push(PM_EL_LINE, X("x2"), Y("y2"));
        return true;
    }
    if (n == "polyline" || n == "polygon") {
        const Attr *pts = Find(attrs, "points");
        if (!pts) return false;
        PathLexer lx(pts->val, pts->val + pts->val_len);
        double x, y;
        size_t count = 0;
        while (lx.Number(&x)) {
            if (!lx.Number(&y)) break;  // an odd coordinate count ends the list (error handling of 9.7)
            out.Push(count ? PM_EL_LINE : PM_EL_MOVE, x, y);
            ++count;
        }
        if (count < 2) {
            els->resize(els->size() - count);
            return false;
        }
        *closed = n == "polygon";
        if (*closed) out.Push(PM_EL_CLOSE);
        return true;
    }
    return false;
}

bool IsShape(const char *n, size_t len) {
    static const char *k[] = {"rect", "circle", "ellipse", "line", "polyline", "polygon"};
    for (const char *s : k)
        if (std::strlen(s) == len && std::memcmp(s, n, len) == 0) return true;
    return false;
}

// Elements that carry an id: the source range of the whole element (start tag to end tag), for <use>
struct IdRange {
    std::string id;
    const char *begin;
    const char *end;
};

// '>' that closes the tag starting at p (quotes respected), or nullptr
const char *TagEnd(const char *p, const char *end) {
    char quote = 0;
    while (p < end && (quote || *p != '>')) {
        if (quote) {
            if (*p == quote) quote = 0;
        } else if (*p == '"' || *p == '\'') {
            quote = *p;
        }
        ++p;
    }
    return p < end ? p : nullptr;
}

void IndexIds(const char *text, const char *end, std::vector<IdRange> *ids) {
    std::vector<int> open;  // per open element: its entry in ids, or -1
    std::vector<Attr> attrs;
    const char *p = text;
    while (p < end) {
        const char *lt = static_cast<const char *>(std::memchr(p, '<', end - p));
        if (!lt || lt + 1 >= end) break;
        p = lt + 1;
        if (end - p >= 3 && std::memcmp(p, "!--", 3) == 0) {
            const char *q = p + 3;
            while (q + 2 < end && std::memcmp(q, "-->", 3) != 0) ++q;
            p = (q + 3 <= end) ? q + 3 : end;
            continue;
        }
        const char *gt = (*p == '?' || *p == '!' || *p == '/') ? static_cast<const char *>(std::memchr(p, '>', end - p)) : TagEnd(p, end);
        if (!gt) break;
        if (*p == '/') {
            if (!open.empty()) {
                if (open.back() >= 0) (*ids)[open.back()].end = gt + 1;
                open.pop_back();
            }
        } else if (*p != '?' && *p != '!') {
            const char *n1 = p;
            while (n1 < gt && !std::isspace(static_cast<unsigned char>(*n1)) && *n1 != '/') ++n1;
            const bool self_closing = gt > p && gt[-1] == '/';
            int entry = -1;
            if (ScanAttrs(n1, gt, &attrs))
                if (const Attr *a = Find(attrs, "id")) {
                    entry = static_cast<int>(ids->size());
                    ids->push_back({std::string(a->val, a->val_len), lt, gt + 1});
                }
            if (!self_closing) open.push_back(entry);
        }
        p = gt + 1;
    }
}

// Work a hostile document may ask for is bounded: <use> nests 8 deep, so k references per level
// would expand to k^8 elements; every expansion counts against this budget.
constexpr uint64_t kMaxUseExpansions = 1u << 20;

struct Doc {
    int flags;
    pm_svg *out;
    uint64_t use_expansions = 0;
    std::vector<IdRange> ids;
    std::vector<CssRule> css;
};

// PM_SVG_FLAT_GRADIENTS: a <linearGradient> / <radialGradient> as one colour -- the arithmetic mean
// of its stops' colours (sRGB components) and of their stop-opacity; a gradient without stops takes
// those of the gradient its href names.
struct GradientResolver : PaintServerResolver {
    const Doc *doc = nullptr;
    bool Stops(const std::string &id, int depth, double sum[4], int *n) const {
        for (const IdRange &r : doc->ids) {
            if (r.id != id) continue;
            const char *p = r.begin + 1;
            const char *n1 = p;
            while (n1 < r.end && !std::isspace(static_cast<unsigned char>(*n1)) && *n1 != '>' && *n1 != '/') ++n1;
            const std::string el(p, n1 - p);
            if (el != "linearGradient" && el != "radialGradient") return false;
            std::vector<Attr> attrs;
            for (const char *q = r.begin; q < r.end;) {
                const char *lt = static_cast<const char *>(std::memchr(q, '<', r.end - q));
                if (!lt || r.end - lt < 6) break;
                q = lt + 1;
                if (std::memcmp(lt, "<stop", 5) != 0 || !(std::isspace(static_cast<unsigned char>(lt[5])) || lt[5] == '/' || lt[5] == '>')) continue;
                const char *gt = TagEnd(lt + 1, r.end);
                if (!gt) break;
                if (!ScanAttrs(lt + 5, gt, &attrs)) continue;
                Style st;  // stop-color / stop-opacity: presentation attributes or style declarations
                Paint c;
                c.none = false;
                c.rgb = 0;  // initial stop-color: black
                double op = 1.0;
                auto take = [&](const std::string &name, const std::string &value) {
                    if (name == "stop-color") (void)ParsePaint(value, &c);
                    else if (name == "stop-opacity") op = ParseOpacity(value, op);
                };
                if (const Attr *a = Find(attrs, "stop-color")) take("stop-color", Trim(a->val, a->val_len));
                if (const Attr *a = Find(attrs, "stop-opacity")) take("stop-opacity", Trim(a->val, a->val_len));
                if (const Attr *a = Find(attrs, "style")) {
                    const char *d = a->val, *de = a->val + a->val_len;
                    while (d < de) {
                        const char *semi = static_cast<const char *>(std::memchr(d, ';', de - d));
                        const char *e2 = semi ? semi : de;
                        const char *colon = static_cast<const char *>(std::memchr(d, ':', e2 - d));
                        if (colon) take(Trim(d, colon - d), Trim(colon + 1, e2 - colon - 1));
                        d = e2 + 1;
                    }
                }
                if (c.none) continue;
                sum[0] += (c.rgb >> 16) & 0xffu;
                sum[1] += (c.rgb >> 8) & 0xffu;
                sum[2] += c.rgb & 0xffu;
                sum[3] += op;
                *n += 1;
                q = gt + 1;
            }
            if (*n == 0 && depth < 4) {  // inherit the stops of the referenced gradient
                const char *gt = TagEnd(r.begin + 1, r.end);
                if (gt && ScanAttrs(n1, gt, &attrs)) {
                    const Attr *h = Find(attrs, "href");
                    if (!h) h = Find(attrs, "xlink:href");
                    if (h && h->val_len > 1 && h->val[0] == '#') return Stops(std::string(h->val + 1, h->val_len - 1), depth + 1, sum, n);
                }
            }
            return *n > 0;
        }
        return false;
    }
    bool Resolve(const std::string &id, uint32_t *rgb, double *alpha) const override {
        double sum[4] = {0, 0, 0, 0};
        int n = 0;
        const PaintServerResolver *saved = t_paint_servers;
        t_paint_servers = nullptr;  // (stop colours are plain colours)
        const bool ok = Stops(id, 0, sum, &n);
        t_paint_servers = saved;
        if (!ok) return false;
        const auto ch = [&](double v) { return static_cast<uint32_t>(std::lround(v / n)); };
        *rgb = (ch(sum[0]) << 16) | (ch(sum[1]) << 8) | ch(sum[2]);
        *alpha = sum[3] / n;
        return true;
    }
};

// Walks the elements of [text, end) under the inherited style `initial`.  use_depth > 0: the range
// is the target of a <use> (a <symbol> at its start is entered like a group).
int ParseRange(Doc *doc, const char *text, const char *end, const Style &initial, int use_depth) {
    const int flags = doc->flags;
    pm_svg *out = doc->out;
    const char *p = text;
    std::vector<Attr> attrs;
    std::vector<Style> stack(1, initial);
    std::vector<bool> container;  // open elements: does this one own a stack entry
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
        if (*p == '?' || *p == '!') {  // PI, doctype
            const char *gt = static_cast<const char *>(std::memchr(p, '>', end - p));
            p = gt ? gt + 1 : end;
            continue;
        }
        if (*p == '/') {  // end tag: leave the element
            const char *gt = static_cast<const char *>(std::memchr(p, '>', end - p));
            p = gt ? gt + 1 : end;
            if (!container.empty()) {
                if (container.back() && stack.size() > 1) stack.pop_back();
                container.pop_back();
            }
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
        const bool self_closing = q > p && q[-1] == '/';
        const bool is_path = name_len == 4 && std::memcmp(n0, "path", 4) == 0;
        const bool is_shape = !is_path && IsShape(n0, name_len);
        const bool is_symbol = name_len == 6 && std::memcmp(n0, "symbol", 6) == 0;
        const bool is_use = name_len == 3 && std::memcmp(n0, "use", 3) == 0;
        const bool is_group = (name_len == 1 && n0[0] == 'g') || (name_len == 3 && std::memcmp(n0, "svg", 3) == 0) ||
                              (name_len == 1 && n0[0] == 'a') || (name_len == 6 && std::memcmp(n0, "switch", 6) == 0) ||
                              (is_symbol && use_depth > 0 && lt == text);
        const bool skipped = (name_len == 4 && std::memcmp(n0, "defs", 4) == 0) || (name_len == 8 && std::memcmp(n0, "clipPath", 8) == 0) ||
                             (name_len == 4 && std::memcmp(n0, "mask", 4) == 0) || (is_symbol && !(use_depth > 0 && lt == text)) ||
                             (name_len == 7 && std::memcmp(n0, "pattern", 7) == 0) || (name_len == 6 && std::memcmp(n0, "marker", 6) == 0);
        if (skipped && !self_closing) {
            // definitions are not rendered directly: skip to the matching end tag
            const std::string close = "</" + std::string(n0, name_len);
            int depth = 1;
            const char *r = q + 1;
            const std::string open = "<" + std::string(n0, name_len);
            while (r < end && depth > 0) {
                const char *nx = static_cast<const char *>(std::memchr(r, '<', end - r));
                if (!nx) { r = end; break; }
                if (static_cast<size_t>(end - nx) >= close.size() && std::memcmp(nx, close.c_str(), close.size()) == 0) --depth;
                else if (static_cast<size_t>(end - nx) > open.size() && std::memcmp(nx, open.c_str(), open.size()) == 0 &&
                         (nx[open.size()] == '>' || std::isspace(static_cast<unsigned char>(nx[open.size()])))) ++depth;
                r = nx + 1;
            }
            const char *gt = r < end ? static_cast<const char *>(std::memchr(r, '>', end - r)) : nullptr;
            p = gt ? gt + 1 : end;
            continue;
        }
        if (is_use) {
            // <use href="#id" x= y=>: the referenced element drawn here, under this element's
            // inherited properties and transform * translate(x, y) (SVG 1.1 section 5.6; a <symbol>
            // is entered like a group, its viewBox is not applied).  References nest at most 8 deep.
            if (!ScanAttrs(p, q, &attrs)) return PM_ERR_PARSE;
            Style st = stack.back();
            if (!ApplyElementStyle(attrs, &st, &doc->css, n0, name_len)) return PM_ERR_PARSE;
            Affine shift;
            if (const Attr *a = Find(attrs, "x")) shift.e = std::strtod(std::string(a->val, a->val_len).c_str(), nullptr);
            if (const Attr *a = Find(attrs, "y")) shift.f = std::strtod(std::string(a->val, a->val_len).c_str(), nullptr);
            st.ctm = st.ctm.Then(shift);
            const Attr *href = Find(attrs, "href");
            if (!href) href = Find(attrs, "xlink:href");
            if (href && href->val_len > 1 && href->val[0] == '#' && use_depth < 8) {
                const std::string id(href->val + 1, href->val_len - 1);
                for (const IdRange &r : doc->ids)
                    if (r.id == id) {
                        if (!(lt >= r.begin && lt < r.end)) {  // (an element cannot use its own ancestor)
                            if (++doc->use_expansions > kMaxUseExpansions) return PM_ERR_PARSE;
                            const int rc = ParseRange(doc, r.begin, r.end, st, use_depth + 1);
                            if (rc != PM_OK) return rc;
                        }
                        break;
                    }
            }
            if (!self_closing) container.push_back(false);
            p = q + 1;
            continue;
        }
        if (is_path || is_shape || is_group) {
            if (!ScanAttrs(p, q, &attrs)) return PM_ERR_PARSE;
            if (name_len == 3 && n0[0] == 's' && !out->seen_root) {  // the outermost <svg>: where the picture lives
                out->seen_root = true;
                auto length = [](const Attr *a) -> double {  // user units / px; a percentage has no meaning here
                    if (!a) return 0.0;
                    const std::string v(a->val, a->val_len);
                    char *rest = nullptr;
                    const double d = std::strtod(v.c_str(), &rest);
                    while (rest && *rest && std::isspace(static_cast<unsigned char>(*rest))) ++rest;
                    const std::string unit = rest ? rest : "";
                    if (unit.empty() || unit == "px") return d > 0 ? d : 0.0;
                    if (unit == "pt") return d * (96.0 / 72.0);
                    if (unit == "in") return d * 96.0;
                    if (unit == "mm") return d * (96.0 / 25.4);
                    if (unit == "cm") return d * (96.0 / 2.54);
                    return 0.0;
                };
                out->width = length(Find(attrs, "width"));
                out->height = length(Find(attrs, "height"));
                if (const Attr *vb = Find(attrs, "viewBox")) {
                    std::string v(vb->val, vb->val_len);
                    for (char &ch : v)
                        if (ch == ',') ch = ' ';
                    double n[4];
                    const char *c = v.c_str();
                    int got = 0;
                    for (; got < 4; ++got) {
                        char *e2 = nullptr;
                        n[got] = std::strtod(c, &e2);
                        if (e2 == c) break;
                        c = e2;
                    }
                    if (got == 4 && n[2] > 0 && n[3] > 0) {
                        out->has_viewbox = true;
                        std::memcpy(out->viewbox, n, sizeof(n));
                    }
                }
            }
            Style st = stack.back();
            if (!ApplyElementStyle(attrs, &st, &doc->css, n0, name_len)) return PM_ERR_PARSE;
            if (is_group) {
                if (!self_closing) {
                    stack.push_back(st);
                    container.push_back(true);
                }
            } else {
                const size_t el0 = out->els.size();
                bool ok = true, has_arc = false, closed = true;
                if (is_path) {
                    const Attr *d = Find(attrs, "d");
                    if (!d) return PM_ERR_PARSE;  // .attribute("d").unwrap(), src/lib.rs:295
                    ok = ParsePathData(d->val, d->val_len, &out->els, &has_arc);
                    if (has_arc && (flags & PM_SVG_REJECT_ARC_PATHS)) ok = false;
                } else {
                    ok = ShapeToEls(n0, name_len, attrs, out->has_viewbox ? out->viewbox[2] : out->width,
                                    out->has_viewbox ? out->viewbox[3] : out->height, &out->els, &closed);
                }
                if (!ok) {
                    out->els.resize(el0);  // `if let Ok(ref bp) = ...` skips the path, src/lib.rs:296
                } else {
                    TransformEls(&out->els, el0, st.ctm);
                    pm_path path{};
                    path.el_begin = static_cast<uint32_t>(el0);
                    path.el_end = static_cast<uint32_t>(out->els.size());
                    const bool shown = !st.display_none && !st.hidden;
                    if (shown && !st.fill.none && (closed || is_path || name_len == 8 /* polyline fills its implicit closure */)) {
                        path.flags |= PM_PATH_FILL;
                        if (st.even_odd) path.flags |= PM_PATH_EVEN_ODD;
                        if (flags & PM_SVG_SPEC_DEFAULTS) {
                            // SVG fills a path's sub-paths TOGETHER (holes, overlaps by the fill rule);
                            // make_tiger fills them one by one (src/lib.rs:343-347), which stays the default
                            size_t moves = 0;
                            for (size_t k = el0; k < out->els.size(); ++k) moves += out->els[k].tag == PM_EL_MOVE;
                            if (moves > 1) path.flags |= PM_PATH_COMPOUND;
                        }
                        path.fill_rgba = PaintRgba(st.fill, st.opacity * st.fill_opacity * st.fill_server_alpha);
                    }
                    if (shown && !st.stroke.none) {
                        path.flags |= PM_PATH_STROKE;
                        path.stroke_rgba = PaintRgba(st.stroke, st.opacity * st.stroke_opacity * st.stroke_server_alpha);
                        // widths scale with the geometric mean of the matrix's stretch (exact for similarities)
                        const double det = std::fabs(st.ctm.a * st.ctm.d - st.ctm.b * st.ctm.c);
                        path.stroke_width = st.ctm.IsIdentity() ? st.stroke_width : static_cast<float>(st.stroke_width * std::sqrt(det));
                    }
                    if (path.flags & (PM_PATH_FILL | PM_PATH_STROKE)) out->paths.push_back(path);
                    else out->els.resize(el0);
                }
                if (!self_closing) container.push_back(false);
            }
        } else if (!self_closing) {
            container.push_back(false);  // an element this front-end does not draw (its children may still be drawn)
        }
        p = q + 1;
    }
    return PM_OK;
}

int ParseDocument(const char *text, size_t len, int flags, pm_svg *out) {
    Doc doc;
    doc.flags = flags;
    doc.out = out;
    IndexIds(text, text + len, &doc.ids);
    for (const char *p = text, *end = text + len; p < end;) {  // every <style> element of the document
        const char *st0 = static_cast<const char *>(std::memchr(p, '<', end - p));
        if (!st0) break;
        if (end - st0 >= 7 && std::memcmp(st0, "<style", 6) == 0 && (st0[6] == '>' || std::isspace(static_cast<unsigned char>(st0[6])))) {
            const char *gt = TagEnd(st0 + 1, end);
            if (!gt) break;
            const char *body = gt + 1;
            const char *close = body;
            while (close + 8 <= end && std::memcmp(close, "</style>", 8) != 0) ++close;
            if (gt[-1] != '/' && close + 8 <= end) ParseStyleSheet(body, close, &doc.css);
            p = close;
        } else {
            p = st0 + 1;
        }
    }
    Style initial;
    // SVG's initial fill is black; make_tiger only fills a path that HAS a fill attribute
    // (src/lib.rs:299).  The Tiger wraps everything in <g fill="none">, where both readings
    // agree; elsewhere the reference's reading is the default and PM_SVG_SPEC_DEFAULTS the SVG one.
    initial.fill.none = (flags & PM_SVG_SPEC_DEFAULTS) == 0;
    initial.fill.rgb = 0;
    GradientResolver gradients;
    gradients.doc = &doc;
    t_paint_servers = (flags & PM_SVG_FLAT_GRADIENTS) ? &gradients : nullptr;
    const int rc = ParseRange(&doc, text, text + len, initial, 0);
    t_paint_servers = nullptr;
    return rc;
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
int pm_svg_viewbox(const pm_svg *s, double viewbox[4], double *width, double *height) {
    if (!s) return 0;
    if (viewbox) std::memcpy(viewbox, s->viewbox, sizeof(s->viewbox));
    if (width) *width = s->width;
    if (height) *height = s->height;
    return s->has_viewbox ? 1 : 0;
}
uint32_t pm_parse_color(const char *s) { return s ? ParseColor(s, std::strlen(s)) : 0xff00ff80u; }

}  // extern "C"
