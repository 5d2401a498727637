/*
 * ORACLE (test infrastructure, NOT product code) -- see pmo.h.
 *
 * Restatement of the per-pixel interpreter and the composite:
 *   TestApp/PietRender.metal:49-60    stroke(), renderDf()
 *   TestApp/PietRender.metal:457-566  renderKernel
 *   TestApp/PietRender.metal:16-44    vertexShader/fragmentShader composite
 *                                     (solid tiles bypass the per-pixel texture)
 *
 * Numeric pins (decisions D1-D8), all taken here:
 *   D1 mix(x,y,a) = x + (y - x)*a, each op rounded in the operand type.
 *   D2 final encode on half3: literals take the vector's element type, so
 *      thr = half(0.0031308), k = half(12.92), s = half(1.055), o = half(0.055),
 *      exponent e = half(float(1)/2.4f); pow is the correctly rounded binary16
 *      result of x^e (computed in f64, rounded once); the affine runs in half.
 *   D3 unpack_unorm4x8_srgb_to_half: exact sRGB EOTF in f64 rounded once to
 *      binary16; alpha = a/255 in f64 rounded once to binary16.
 *   D4 half -> unorm8: clamp to [0,1], times 255.0f in f32, round-half-even.
 *   D5 0/0 in stroke(): saturate(NaN) = 0 (fmax/fmin drop the NaN).
 *   D6 length(v) = sqrtf(x*x + y*y); dot = x0*x1 + y0*y1; no FMA.
 *   D7 PMO_MODE_HALF (source types) or PMO_MODE_F32 (all accumulators f32).
 *   D8 RGBA8 output by default, BGRA8 with PMO_FMT_BGRA8.
 * min/max are IEEE minNum/maxNum (fminf/fmaxf); saturate(x) = fminf(fmaxf(x,0),1).
 */
#include "pmo.h"
#include "pmo_half.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- lookup tables ----------------------------------------------------------- */

/* f64 -> binary16, one rounding (RNE), for values with |d| < 65520. */
static pmo_half d2h(double d) {
    if (d != d) return 0x7e00;
    uint16_t sign = 0;
    if (d < 0 || (d == 0 && 1.0 / d < 0)) { sign = 0x8000; d = -d; }
    if (d >= 65520.0) return sign | 0x7c00;
    int e;
    (void)frexp(d, &e);       /* d = m * 2^e, m in [0.5,1) */
    int qe = e - 11;          /* quantum exponent for 11 significant bits */
    if (qe < -24) qe = -24;   /* subnormal quantum */
    double q = nearbyint(ldexp(d, -qe)); /* default rounding mode = RNE */
    float f = (float)ldexp(q, qe);       /* exactly representable */
    return sign | pmo_f2h(f);
}

void pmo_lut_srgb_to_linear_half(uint16_t out[256]) {
    for (int i = 0; i < 256; i++) {
        double c = (double)i / 255.0;
        double l = (c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
        out[i] = d2h(l);
    }
}

void pmo_lut_unorm_to_half(uint16_t out[256]) {
    for (int i = 0; i < 256; i++) out[i] = d2h((double)i / 255.0);
}

static uint8_t unorm8(float v) { /* D4 */
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    return (uint8_t)rintf(v * 255.0f);
}

void pmo_lut_linear_half_to_srgb8(uint8_t out[65536]) {
    const pmo_half thr = pmo_f2h(0.0031308f);
    const pmo_half k = pmo_f2h(12.92f);
    const pmo_half s = pmo_f2h(1.055f);
    const pmo_half o = pmo_f2h(0.055f);
    const double e = (double)pmo_h2f(pmo_f2h(1.0f / 2.4f));
    for (uint32_t h = 0; h < 65536; h++) {
        float x = pmo_h2f((pmo_half)h);
        pmo_half y;
        if (x != x) {
            out[h] = 0;
            continue;
        }
        if (x < pmo_h2f(thr)) {
            y = pmo_hmul(k, (pmo_half)h);
        } else {
            pmo_half p = d2h(pow((double)x, e));
            y = pmo_hsub(pmo_hmul(s, p), o);
        }
        out[h] = unorm8(pmo_h2f(y));
    }
}

static uint16_t g_srgb2lin[256];
static uint16_t g_unorm2h[256];
static uint8_t g_lin2srgb[65536];
static int g_luts_ready = 0;

static void ensure_luts(void) {
    if (g_luts_ready) return;
    pmo_lut_srgb_to_linear_half(g_srgb2lin);
    pmo_lut_unorm_to_half(g_unorm2h);
    pmo_lut_linear_half_to_srgb8(g_lin2srgb);
    g_luts_ready = 1;
}

/* ---- shared f32 pieces ------------------------------------------------------- */

static float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Extension D10: coverage of the ellipse inscribed in the bbox, the "shade an ellipse properly"
 * of PietRender.metal:488-489 with the first-order distance of WebRender's ellipse.glsl that the
 * comment points to: for F(p) = px^2/rx^2 + py^2/ry^2 - 1, distance ~ F / |grad F|.  binary32,
 * one rounding per operation, in exactly this order (the HIP kernels do the same):
 *   ux = px / (rx * rx), uy = py / (ry * ry); g = (px * ux + py * uy) - 1;
 *   len = 2 * sqrt(ux * ux + uy * uy); alpha = saturate(-(g / len)).
 * At the centre len = 0 and g = -1: -(g / len) = +inf, alpha = 1.  A bbox without area draws nothing. */
static float ellipse_alpha(float ddx, float ddy, float rx, float ry) {
    if (!(rx > 0.0f) || !(ry > 0.0f)) return 0.0f;
    float ux = ddx / (rx * rx), uy = ddy / (ry * ry);
    float g = (ddx * ux + ddy * uy) - 1.0f;
    float len = 2.0f * sqrtf(ux * ux + uy * uy);
    return saturatef(-(g / len));
}

/* stroke(), PietRender.metal:49-55 */
static void stroke_df(float *df, float px, float py, float sx, float sy, float ex, float ey) {
    float lx = ex - sx, ly = ey - sy;
    float dx = px - sx, dy = py - sy;
    float t = saturatef((lx * dx + ly * dy) / (lx * lx + ly * ly));
    float fx = lx * t - dx, fy = ly * t - dy;
    float field = sqrtf(fx * fx + fy * fy);
    *df = fminf(*df, field);
}

/* Cmd_Fill body, PietRender.metal:508-529: returns 1 and *contrib = area*(w.x-w.y) */
static int fill_area(float px, float py, float sx, float sy, float ex, float ey, float *contrib) {
    float stx = sx - px, sty = sy - py;
    float enx = ex - px, eny = ey - py;
    float wx = saturatef(sty), wy = saturatef(eny);
    if (wx != wy) {
        float tx = (wx - sty) / (eny - sty);
        float ty = (wy - sty) / (eny - sty);
        float xsx = stx + (enx - stx) * tx;
        float xsy = stx + (enx - stx) * ty;
        float xmin = fminf(fminf(xsx, xsy), 1.0f) - 1e-6f;
        float xmax = fmaxf(xsx, xsy);
        float b = fminf(xmax, 1.0f);
        float c = fmaxf(b, 0.0f);
        float d = fmaxf(xmin, 0.0f);
        float area = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
        *contrib = area * (wx - wy);
        return 1;
    }
    return 0;
}

/* ---- renderKernel, half accumulators ------------------------------------------ */

/* Returns 0 if the pixel was written (rgb out), 1 for Bail, 2 for bad tag. */
static int render_pixel_half(const pmo_cmd *cmds, uint32_t x, uint32_t y, uint8_t out_rgb[3]) {
    float px = (float)x, py = (float)y; /* xy = float2(gid) :466 */
    pmo_half rgb[3] = {PMO_H_ONE, PMO_H_ONE, PMO_H_ONE}; /* :470 */
    float df = 1e9f;                                      /* :471 */
    pmo_half signed_area = PMO_H_ZERO;                    /* :472 */
    for (const pmo_cmd *cmd = cmds;; cmd++) {             /* :474 */
        uint32_t tag = cmd->tag;
        if (tag == PMO_CMD_END) break;
        switch (tag) {
            case PMO_CMD_CIRCLE: { /* :481-494 */
                float x0 = (float)(cmd->body[1] & 0xffffu), y0 = (float)(cmd->body[1] >> 16);
                float x1 = (float)(cmd->body[2] & 0xffffu), y1 = (float)(cmd->body[2] >> 16);
                float cx = x0 + (x1 - x0) * 0.5f, cy = y0 + (y1 - y0) * 0.5f;
                float ddx = px - cx, ddy = py - cy;
                float r = sqrtf(ddx * ddx + ddy * ddy);
                float circle_r = fminf(cx - x0, cy - y0);
                float alpha = saturatef(circle_r - r);
                if (cmd->body[0] & 1u) alpha = ellipse_alpha(ddx, ddy, cx - x0, cy - y0);
                pmo_half ha = pmo_f2h(alpha);
                for (int k = 0; k < 3; k++) rgb[k] = pmo_hmix(rgb[k], PMO_H_ZERO, ha);
                break;
            }
            case PMO_CMD_LINE: /* :495-499 */
                stroke_df(&df, px, py, u2f(cmd->body[1]), u2f(cmd->body[2]), u2f(cmd->body[3]), u2f(cmd->body[4]));
                break;
            case PMO_CMD_STROKE: { /* :500-507 */
                float half_width = u2f(cmd->body[0]);
                uint32_t rgba = cmd->body[1];
                pmo_half alpha = pmo_f2h(saturatef(half_width + 0.5f - df)); /* renderDf :58-60 */
                pmo_half fa = pmo_hmul(g_unorm2h[rgba >> 24], alpha);
                for (int k = 0; k < 3; k++)
                    rgb[k] = pmo_hmix(rgb[k], g_srgb2lin[(rgba >> (8 * k)) & 0xffu], fa);
                df = 1e9f;
                break;
            }
            case PMO_CMD_FILL: { /* :508-529 */
                float contrib;
                if (fill_area(px, py, u2f(cmd->body[1]), u2f(cmd->body[2]), u2f(cmd->body[3]), u2f(cmd->body[4]), &contrib))
                    signed_area = pmo_hadd(signed_area, pmo_f2h(contrib));
                break;
            }
            case PMO_CMD_FILL_EDGE: { /* :530-534: half + float => f32 add, one rounding */
                float sgn = (float)(int32_t)cmd->body[0];
                float fy = u2f(cmd->body[1]);
                float v = sgn * saturatef((float)y - fy + 1.0f);
                signed_area = pmo_f2h(pmo_h2f(signed_area) + v);
                break;
            }
            case PMO_CMD_DRAW_FILL: { /* :535-545 */
                int32_t backdrop = (int32_t)cmd->body[0];
                uint32_t rgba = cmd->body[1];
                pmo_half alpha = pmo_hadd(signed_area, pmo_f2h((float)backdrop));
                if (cmd->body[4] & PMO_FILL_EVEN_ODD) {
                    /* the reference's comment (:539): alpha = abs(alpha - 2.0 * round(0.5 * alpha)), in
                     * half.  Decision D9: round = nearest integer, ties to even -- a tie means alpha is an
                     * odd integer, where either neighbour gives |+-1| = 1, so the tie rule cannot show. */
                    pmo_half t = pmo_hmul(pmo_f2h(0.5f), alpha);
                    pmo_half r = pmo_f2h(rintf(pmo_h2f(t)));
                    pmo_half v = pmo_hsub(alpha, pmo_hmul(pmo_f2h(2.0f), r));
                    alpha = pmo_f2h(fabsf(pmo_h2f(v)));
                } else {
                    float fa_abs = fabsf(pmo_h2f(alpha));
                    alpha = pmo_f2h(fminf(fa_abs, 1.0f)); /* min(abs(alpha), 1.0h) */
                }
                pmo_half fa = pmo_hmul(g_unorm2h[rgba >> 24], alpha);
                for (int k = 0; k < 3; k++)
                    rgb[k] = pmo_hmix(rgb[k], g_srgb2lin[(rgba >> (8 * k)) & 0xffu], fa);
                signed_area = PMO_H_ZERO;
                break;
            }
            case PMO_CMD_SOLID: { /* :546-551 */
                uint32_t rgba = cmd->body[0];
                pmo_half fa = g_unorm2h[rgba >> 24];
                for (int k = 0; k < 3; k++)
                    rgb[k] = pmo_hmix(rgb[k], g_srgb2lin[(rgba >> (8 * k)) & 0xffu], fa);
                break;
            }
            case PMO_CMD_BAIL: /* :552-553 */
                return 1;
            default: /* :555-557 magenta */
                out_rgb[0] = 255; out_rgb[1] = 0; out_rgb[2] = 255;
                return 2;
        }
    }
    for (int k = 0; k < 3; k++) out_rgb[k] = g_lin2srgb[rgb[k]]; /* :563-565 */
    return 0;
}

/* ---- renderKernel, f32 accumulators (D7 second mode) --------------------------- */

static float mixf(float x, float y, float a) { return x + (y - x) * a; }

static float srgb_to_linear_f32(uint32_t c8) {
    double c = (double)c8 / 255.0;
    return (float)((c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
}

static int render_pixel_f32(const pmo_cmd *cmds, uint32_t x, uint32_t y, uint8_t out_rgb[3]) {
    float px = (float)x, py = (float)y;
    float rgb[3] = {1.0f, 1.0f, 1.0f};
    float df = 1e9f;
    float signed_area = 0.0f;
    for (const pmo_cmd *cmd = cmds;; cmd++) {
        uint32_t tag = cmd->tag;
        if (tag == PMO_CMD_END) break;
        switch (tag) {
            case PMO_CMD_CIRCLE: {
                float x0 = (float)(cmd->body[1] & 0xffffu), y0 = (float)(cmd->body[1] >> 16);
                float x1 = (float)(cmd->body[2] & 0xffffu), y1 = (float)(cmd->body[2] >> 16);
                float cx = x0 + (x1 - x0) * 0.5f, cy = y0 + (y1 - y0) * 0.5f;
                float ddx = px - cx, ddy = py - cy;
                float r = sqrtf(ddx * ddx + ddy * ddy);
                float alpha = saturatef(fminf(cx - x0, cy - y0) - r);
                if (cmd->body[0] & 1u) alpha = ellipse_alpha(ddx, ddy, cx - x0, cy - y0);
                for (int k = 0; k < 3; k++) rgb[k] = mixf(rgb[k], 0.0f, alpha);
                break;
            }
            case PMO_CMD_LINE:
                stroke_df(&df, px, py, u2f(cmd->body[1]), u2f(cmd->body[2]), u2f(cmd->body[3]), u2f(cmd->body[4]));
                break;
            case PMO_CMD_STROKE: {
                uint32_t rgba = cmd->body[1];
                float alpha = saturatef(u2f(cmd->body[0]) + 0.5f - df);
                float fa = ((float)(rgba >> 24) / 255.0f) * alpha;
                for (int k = 0; k < 3; k++)
                    rgb[k] = mixf(rgb[k], srgb_to_linear_f32((rgba >> (8 * k)) & 0xffu), fa);
                df = 1e9f;
                break;
            }
            case PMO_CMD_FILL: {
                float contrib;
                if (fill_area(px, py, u2f(cmd->body[1]), u2f(cmd->body[2]), u2f(cmd->body[3]), u2f(cmd->body[4]), &contrib))
                    signed_area += contrib;
                break;
            }
            case PMO_CMD_FILL_EDGE:
                signed_area += (float)(int32_t)cmd->body[0] * saturatef((float)y - u2f(cmd->body[1]) + 1.0f);
                break;
            case PMO_CMD_DRAW_FILL: {
                uint32_t rgba = cmd->body[1];
                float alpha = signed_area + (float)(int32_t)cmd->body[0];
                if (cmd->body[4] & PMO_FILL_EVEN_ODD) alpha = fabsf(alpha - 2.0f * rintf(0.5f * alpha));
                else alpha = fminf(fabsf(alpha), 1.0f);
                float fa = ((float)(rgba >> 24) / 255.0f) * alpha;
                for (int k = 0; k < 3; k++)
                    rgb[k] = mixf(rgb[k], srgb_to_linear_f32((rgba >> (8 * k)) & 0xffu), fa);
                signed_area = 0.0f;
                break;
            }
            case PMO_CMD_SOLID: {
                uint32_t rgba = cmd->body[0];
                float fa = (float)(rgba >> 24) / 255.0f;
                for (int k = 0; k < 3; k++)
                    rgb[k] = mixf(rgb[k], srgb_to_linear_f32((rgba >> (8 * k)) & 0xffu), fa);
                break;
            }
            case PMO_CMD_BAIL:
                return 1;
            default:
                out_rgb[0] = 255; out_rgb[1] = 0; out_rgb[2] = 255;
                return 2;
        }
    }
    for (int k = 0; k < 3; k++) {
        float v = rgb[k];
        float e = (v < 0.0031308f) ? 12.92f * v : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
        out_rgb[k] = unorm8(e);
    }
    return 0;
}

/* ---- whole tiles + composite ---------------------------------------------------- */

int pmo_render_rows(const pmo_ptcl *p, uint32_t width, uint32_t height, uint32_t ty0,
                    uint32_t ty1, uint32_t flags, uint8_t *out) {
    ensure_luts();
    const uint32_t tiles_x = pmo_ptcl_tiles_x(p);
    const uint32_t tiles_y = pmo_ptcl_tiles_y(p);
    if (ty1 > tiles_y) ty1 = tiles_y;
    const int f32_mode = (flags & PMO_MODE_F32) != 0;
    const int bgra = (flags & PMO_FMT_BGRA8) != 0;
    const size_t stride = (size_t)width * 4;
    for (uint32_t ty = ty0; ty < ty1; ty++) {
        for (uint32_t tx = 0; tx < tiles_x; tx++) {
            const pmo_cmd *cmds = pmo_ptcl_cmds(p, tx, ty);
            const uint32_t solid = pmo_ptcl_solid(p, tx, ty);
            for (uint32_t yy = 0; yy < PMO_TILE_H; yy++) {
                uint32_t y = ty * PMO_TILE_H + yy;
                if (y >= height) break;
                uint8_t *row = out + (size_t)(y - ty0 * PMO_TILE_H) * stride;
                for (uint32_t xx = 0; xx < PMO_TILE_W; xx++) {
                    uint32_t x = tx * PMO_TILE_W + xx;
                    if (x >= width) break;
                    uint8_t c[4];
                    if (solid != 0) {
                        /* fragmentShader :34-44: loSample.a != 0 => the tile colour,
                         * bytes as stored (R,G,B,A in memory order) */
                        c[0] = (uint8_t)(solid & 0xff);
                        c[1] = (uint8_t)((solid >> 8) & 0xff);
                        c[2] = (uint8_t)((solid >> 16) & 0xff);
                        c[3] = (uint8_t)(solid >> 24);
                    } else {
                        int r = f32_mode ? render_pixel_f32(cmds, x, y, c) : render_pixel_half(cmds, x, y, c);
                        if (r == 1) return -1; /* Bail with solid==0 cannot happen */
                        c[3] = 255;            /* half4(rgb, 1.0) :564 */
                    }
                    uint8_t *px = row + (size_t)x * 4;
                    if (bgra) { px[0] = c[2]; px[1] = c[1]; px[2] = c[0]; px[3] = c[3]; }
                    else { px[0] = c[0]; px[1] = c[1]; px[2] = c[2]; px[3] = c[3]; }
                }
            }
        }
    }
    return 0;
}

int pmo_render(const uint8_t *scene, size_t scene_len, uint32_t width, uint32_t height,
               uint32_t flags, uint8_t *out) {
    pmo_ptcl *p = pmo_ptcl_build(scene, scene_len, width, height);
    if (!p) return -1;
    int r = pmo_render_rows(p, width, height, 0, pmo_ptcl_tiles_y(p), flags, out);
    pmo_ptcl_free(p);
    return r;
}

/* ---- coverage of one Fill item, f32 accumulation -------------------------------- */

int pmo_fill_coverage(const uint8_t *scene, size_t scene_len, uint32_t item_ix, uint32_t width,
                      uint32_t height, float *out) {
    /* Isolate the item into a one-item scene (same point bytes), run tileKernel,
     * and accumulate its Fill/FillEdge/DrawFill commands in f32. */
    if (scene_len < 8) return -1;
    uint32_t n, items_ix;
    memcpy(&n, scene, 4);
    memcpy(&items_ix, scene + 4, 4);
    if (item_ix >= n) return -1;
    size_t need = 8 + 8 + 32;
    uint8_t *mini = (uint8_t *)calloc(1, scene_len + need);
    /* layout: [hdr 8][bbox 8][item 32][copy of the whole original scene] so that
     * points_ix only needs a constant shift */
    uint32_t one = 1, ix = 16;
    memcpy(mini, &one, 4);
    memcpy(mini + 4, &ix, 4);
    memcpy(mini + 8, scene + 8 + (size_t)item_ix * 8, 8);
    memcpy(mini + 16, scene + items_ix + (size_t)item_ix * 32, 32);
    memcpy(mini + need, scene, scene_len);
    uint32_t tag, pix;
    memcpy(&tag, mini + 16, 4);
    if (tag != PMO_ITEM_FILL) { free(mini); return -2; }
    memcpy(&pix, mini + 16 + 16, 4);
    pix += (uint32_t)need;
    memcpy(mini + 16 + 16, &pix, 4);
    pmo_ptcl *p = pmo_ptcl_build(mini, scene_len + need, width, height);
    free(mini);
    if (!p) return -1;
    for (uint32_t y = 0; y < height; y++) {
        for (uint32_t x = 0; x < width; x++) {
            uint32_t tx = x / PMO_TILE_W, ty = y / PMO_TILE_H;
            const pmo_cmd *cmd = pmo_ptcl_cmds(p, tx, ty);
            float sa = 0.0f, cov = 0.0f;
            if (pmo_ptcl_solid(p, tx, ty) != 0 && pmo_ptcl_solid(p, tx, ty) != 0xffffffffu) cov = 1.0f;
            for (; cmd->tag != PMO_CMD_END && cmd->tag != PMO_CMD_BAIL; cmd++) {
                if (cmd->tag == PMO_CMD_FILL) {
                    float contrib;
                    if (fill_area((float)x, (float)y, u2f(cmd->body[1]), u2f(cmd->body[2]), u2f(cmd->body[3]), u2f(cmd->body[4]), &contrib))
                        sa += contrib;
                } else if (cmd->tag == PMO_CMD_FILL_EDGE) {
                    sa += (float)(int32_t)cmd->body[0] * saturatef((float)y - u2f(cmd->body[1]) + 1.0f);
                } else if (cmd->tag == PMO_CMD_DRAW_FILL) {
                    cov = sa + (float)(int32_t)cmd->body[0];
                    if (cmd->body[4] & PMO_FILL_EVEN_ODD) cov = fabsf(cov - 2.0f * rintf(0.5f * cov));
                    else cov = fminf(fabsf(cov), 1.0f);
                    sa = 0.0f;
                } else if (cmd->tag == PMO_CMD_SOLID) {
                    cov = 1.0f;
                }
            }
            out[(size_t)y * width + x] = cov;
        }
    }
    pmo_ptcl_free(p);
    return 0;
}
