/*
 * ORACLE (test infrastructure, NOT product code) -- see pmo.h.
 *
 * Restatement of TileEncoder (TestApp/PietRender.metal:69-157) and tileKernel
 * (TestApp/PietRender.metal:160-454) as a lane-by-lane simulation of the
 * reference's 16x2-tile threadgroups (32 lanes, TestApp/PietShaderTypes.h:21-22,
 * dispatch geometry TestApp/PietRenderer.m:63-77).  The threadgroup bitmap /
 * barrier dance (:175-208, :296-302, :400-406) is simulated by computing every
 * lane's vote first and then replaying every lane's second phase.
 *
 * All geometry is IEEE binary32, one rounding per source-level operation, no
 * contraction (compile with -ffp-contract=off).  sign(0) = 0.
 *
 * Quirks Q1-Q5 of SURVEY.md section 3.3 are reproduced on purpose; Q5 (the
 * fixed 4096-byte tile buffer) is replaced by an unbounded list.
 */
#include "pmo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

struct pmo_ptcl {
    uint32_t tiles_x, tiles_y;
    uint32_t *count;  /* per tile */
    pmo_cmd **cmds;   /* per tile */
    uint32_t *solid;  /* per tile, TileEncoder::end() */
};

/* ---- scene accessors (TestApp/GenTypes.h:21-328) -------------------------- */

typedef struct {
    const uint8_t *p;
    size_t len;
    int oob;
    size_t root; /* offset of the SimpleGroup the tile pass walks (0, or the flat form of nested groups) */
} scene_t;

static uint32_t rd_u32(scene_t *s, size_t off) {
    uint32_t v = 0;
    if (off + 4 > s->len) { s->oob = 1; return 0; }
    memcpy(&v, s->p + off, 4);
    return v;
}
static float rd_f32(scene_t *s, size_t off) {
    float v = 0;
    if (off + 4 > s->len) { s->oob = 1; return 0; }
    memcpy(&v, s->p + off, 4);
    return v;
}
static uint16_t rd_u16(scene_t *s, size_t off) {
    uint16_t v = 0;
    if (off + 2 > s->len) { s->oob = 1; return 0; }
    memcpy(&v, s->p + off, 2);
    return v;
}

/* Segments of a Fill item.  Plain (the reference, :262-263): point k to point k + 1, the last one
 * back to point 0.  Compound (extension D11, pmo.h): entries with x = NaN separate sub-paths and
 * start no segment; a point followed by a separator closes to the index the separator's y holds
 * (clamped into the array: a malformed scene must not read outside it). */
static int fill_seg_exists(scene_t *s, size_t pts, int compound, uint32_t k) {
    if (!compound) return 1;
    float x = rd_f32(s, pts + (size_t)k * 8);
    return !(x != x);
}
static uint32_t fill_seg_end(scene_t *s, size_t pts, uint32_t n_points, int compound, uint32_t k) {
    uint32_t nxt = (k + 1 == n_points) ? 0 : k + 1;
    if (compound) {
        float x = rd_f32(s, pts + (size_t)nxt * 8);
        if (x != x) {
            uint32_t start = rd_u32(s, pts + (size_t)nxt * 8 + 4);
            nxt = start < n_points ? start : n_points - 1;
        }
    }
    return nxt;
}

/* ---- TileEncoder (PietRender.metal:69-157) --------------------------------- */

typedef struct {
    pmo_cmd *cmds;
    uint32_t n, cap;
    uint32_t solid_color;
} tile_enc;

static pmo_cmd *enc_push(tile_enc *e) {
    if (e->n == e->cap) {
        e->cap = e->cap ? e->cap * 2 : 16;
        e->cmds = (pmo_cmd *)realloc(e->cmds, e->cap * sizeof(pmo_cmd));
    }
    pmo_cmd *c = &e->cmds[e->n++];
    memset(c, 0, sizeof(*c));
    return c;
}

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void enc_circle(tile_enc *e, const uint16_t bbox[4], uint32_t ellipse) { /* :76-83 */
    pmo_cmd *c = enc_push(e);
    c->tag = PMO_CMD_CIRCLE;
    c->body[0] = ellipse; /* extension D10, in the padding word of CmdCirclePacked */
    /* CmdCirclePacked {uint tag; ushort4 bbox}: bbox at byte 8 (ushort4 is 8-aligned) */
    c->body[1] = (uint32_t)bbox[0] | ((uint32_t)bbox[1] << 16);
    c->body[2] = (uint32_t)bbox[2] | ((uint32_t)bbox[3] << 16);
    e->solid_color = 0;
}
static void enc_line(tile_enc *e, float sx, float sy, float ex, float ey) { /* :84-92 */
    pmo_cmd *c = enc_push(e);
    c->tag = PMO_CMD_LINE;
    c->body[1] = f2u(sx); c->body[2] = f2u(sy); c->body[3] = f2u(ex); c->body[4] = f2u(ey);
    e->solid_color = 0;
}
static void enc_stroke(tile_enc *e, uint32_t rgba, float width) { /* :93-101 */
    pmo_cmd *c = enc_push(e);
    c->tag = PMO_CMD_STROKE;
    c->body[0] = f2u(0.5f * width);
    c->body[1] = rgba;
    e->solid_color = 0;
}
static void enc_fill(tile_enc *e, float sx, float sy, float ex, float ey) { /* :102-109 */
    pmo_cmd *c = enc_push(e);
    c->tag = PMO_CMD_FILL;
    c->body[1] = f2u(sx); c->body[2] = f2u(sy); c->body[3] = f2u(ex); c->body[4] = f2u(ey);
}
static void enc_fill_edge(tile_enc *e, float sign, float y) { /* :110-117 */
    pmo_cmd *c = enc_push(e);
    c->tag = PMO_CMD_FILL_EDGE;
    c->body[0] = (uint32_t)(int32_t)sign; /* cmd.sign is int */
    c->body[1] = f2u(y);
}
static void enc_draw_fill(tile_enc *e, uint32_t rgba, int backdrop, uint32_t even_odd) { /* :118-126 */
    pmo_cmd *c = enc_push(e);
    c->tag = PMO_CMD_DRAW_FILL;
    c->body[0] = (uint32_t)backdrop;
    c->body[1] = rgba;
    c->body[4] = even_odd; /* extension: the rule travels in the command's unused last word */
    e->solid_color = 0;
}
static void enc_solid(tile_enc *e, uint32_t rgba) { /* :127-142 */
    if ((rgba & 0xff000000u) == 0xff000000u) {
        e->solid_color = rgba;
        e->n = 0; /* dst = tileBegin */
    }
    pmo_cmd *c = enc_push(e);
    c->tag = PMO_CMD_SOLID;
    c->body[0] = rgba;
}
static uint32_t enc_end(tile_enc *e) { /* :144-151 */
    if (e->solid_color) {
        e->n = 0;
        pmo_cmd *c = enc_push(e);
        c->tag = PMO_CMD_BAIL;
    } else {
        pmo_cmd *c = enc_push(e);
        c->tag = PMO_CMD_END;
    }
    return e->solid_color;
}

static float signf(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

/* ---- tileKernel -------------------------------------------------------------- */

#define LANES 32
#define STW (PMO_TILER_GROUP_W * PMO_TILE_W) /* 256 */
#define STH (PMO_TILER_GROUP_H * PMO_TILE_H) /* 32 */

static void run_group(scene_t *sc, uint32_t gx, uint32_t gy, tile_enc enc[LANES]) {
    const int sx0 = (int)(gx * STW);
    const int sy0 = (int)(gy * STH);
    int x0[LANES], y0[LANES];
    for (int t = 0; t < LANES; t++) {
        x0[t] = sx0 + (t & 15) * PMO_TILE_W;
        y0[t] = sy0 + (t >> 4) * PMO_TILE_H;
        enc[t].n = 0;
        enc[t].solid_color = 0xffffffffu; /* :74 */
    }
    const uint32_t n = rd_u32(sc, sc->root);         /* SimpleGroup_n_items(scene, 0) :187 */
    const uint32_t items_ref = rd_u32(sc, sc->root + 4); /* :188 */
    const size_t bboxes = sc->root + 8;               /* &group->bbox :186 */

    for (uint32_t i = 0; i < n; i += LANES) { /* :189, tgs = 32 */
        uint32_t rd = 0;
        for (uint32_t tix = 0; tix < LANES; tix++) { /* :194-202 */
            if (i + tix < n) {
                size_t bo = bboxes + (size_t)(i + tix) * 8;
                int bx = rd_u16(sc, bo), by = rd_u16(sc, bo + 2);
                int bz = rd_u16(sc, bo + 4), bw = rd_u16(sc, bo + 6);
                if (bz >= sx0 && bx < sx0 + STW && bw >= sy0 && by < sy0 + STH) rd |= 1u << (tix & 31);
            }
        }
        uint32_t v = rd;
        while (v) { /* :211 */
            uint32_t ix = i + (uint32_t)__builtin_ctz(v);
            size_t bo = bboxes + (size_t)ix * 8;
            uint16_t bbox[4] = {rd_u16(sc, bo), rd_u16(sc, bo + 2), rd_u16(sc, bo + 4), rd_u16(sc, bo + 6)};
            int hit[LANES];
            for (int t = 0; t < LANES; t++) /* :214 */
                hit[t] = bbox[2] >= x0[t] && bbox[0] < x0[t] + PMO_TILE_W && bbox[3] >= y0[t] &&
                         bbox[1] < y0[t] + PMO_TILE_H;
            size_t item_ref = (size_t)items_ref + (size_t)ix * PMO_ITEM_SIZE;
            uint32_t item_word = rd_u32(sc, item_ref);
            uint32_t item_type = item_word & 0xffffu; /* ushort itemType :216 */
            switch (item_type) {
                case PMO_ITEM_CIRCLE: /* :218-222 */
                    for (int t = 0; t < LANES; t++)
                        if (hit[t]) enc_circle(&enc[t], bbox, (item_word & PMO_CIRCLE_ELLIPSE) ? 1u : 0u);
                    break;
                case PMO_ITEM_LINE: { /* :223-247 */
                    uint32_t rgba = rd_u32(sc, item_ref + 8);
                    float width = rd_f32(sc, item_ref + 12);
                    float sx = rd_f32(sc, item_ref + 16), sy = rd_f32(sc, item_ref + 20);
                    float ex = rd_f32(sc, item_ref + 24), ey = rd_f32(sc, item_ref + 28);
                    for (int t = 0; t < LANES; t++) {
                        if (!hit[t]) continue;
                        float a = ey - sy;
                        float b = sx - ex;
                        float c = -(a * sx + b * sy);
                        float hw = 0.5f * width + 0.5f;
                        float left = a * ((float)x0[t] - hw);
                        float right = a * ((float)(x0[t] + PMO_TILE_W) + hw);
                        float top = b * ((float)y0[t] - hw);
                        float bot = b * ((float)(y0[t] + PMO_TILE_H) + hw);
                        float s00 = signf(top + left + c);
                        float s01 = signf(top + right + c);
                        float s10 = signf(bot + left + c);
                        float s11 = signf(bot + right + c);
                        if (s00 * s01 + s00 * s10 + s00 * s11 < 3.0f) {
                            enc_line(&enc[t], sx, sy, ex, ey);
                            enc_stroke(&enc[t], rgba, width);
                        }
                    }
                    break;
                }
                case PMO_ITEM_FILL: { /* :248-365 */
                    uint32_t rgba = rd_u32(sc, item_ref + 8);
                    uint32_t even_odd = rd_u32(sc, item_ref + 4) & PMO_FILL_EVEN_ODD; /* PietFill.flags (extension) */
                    int compound = (rd_u32(sc, item_ref + 4) & PMO_FILL_COMPOUND) != 0; /* sub-paths (extension D11) */
                    uint32_t n_points = rd_u32(sc, item_ref + 12);
                    size_t pts = rd_u32(sc, item_ref + 16);
                    float backdrop[LANES];
                    int any_fill[LANES];
                    for (int t = 0; t < LANES; t++) { backdrop[t] = 0.0f; any_fill[t] = 0; }
                    for (uint32_t j = 0; j < n_points; j += 16) { /* :257 */
                        uint32_t votes = 0;
                        for (uint32_t tix = 0; tix < LANES; tix++) { /* phase 1 :258-295 */
                            int fill_hit = 0;
                            uint32_t fill_ix = j + (tix & 15);
                            if (fill_ix < n_points && fill_seg_exists(sc, pts, compound, fill_ix)) {
                                uint32_t nxt = fill_seg_end(sc, pts, n_points, compound, fill_ix);
                                float stx = rd_f32(sc, pts + (size_t)fill_ix * 8), sty = rd_f32(sc, pts + (size_t)fill_ix * 8 + 4);
                                float enx = rd_f32(sc, pts + (size_t)nxt * 8), eny = rd_f32(sc, pts + (size_t)nxt * 8 + 4);
                                float xmin = fminf(stx, enx), ymin = fminf(sty, eny);
                                float xmax = fmaxf(stx, enx), ymax = fmaxf(sty, eny);
                                int ly0 = y0[tix];
                                if (ymax >= (float)ly0 && ymin < (float)(ly0 + PMO_TILE_H) && xmin < (float)(sx0 + STW)) {
                                    float a = eny - sty;
                                    float b = stx - enx;
                                    float c = -(a * stx + b * sty);
                                    float left = a * (float)sx0;
                                    float right = a * (float)(sx0 + STW);
                                    float ytop = fmaxf((float)ly0, ymin);
                                    float ybot = fminf((float)(ly0 + PMO_TILE_H), ymax);
                                    float top = b * ytop;
                                    float bot = b * ybot;
                                    float s_top_left = signf(right - a * (float)PMO_TILE_W + (float)ly0 * b + c);
                                    float s00 = signf(top + left + c);
                                    float s01 = signf(top + right + c);
                                    float s10 = signf(bot + left + c);
                                    float s11 = signf(bot + right + c);
                                    if (s_top_left == signf(a) && ymin <= (float)ly0) fill_hit = 1;
                                    if (s00 * s01 + s00 * s10 + s00 * s11 < 3.0f && xmax > (float)sx0) fill_hit = 1;
                                }
                            }
                            if (fill_hit) votes |= 1u << tix;
                        }
                        for (uint32_t tix = 0; tix < LANES; tix++) { /* phase 2 :302-357 */
                            uint32_t fill_vote = (votes >> (tix & 16)) & 0xffffu;
                            if (!hit[tix]) continue; /* body is entirely under if (hit) */
                            const int lx0 = x0[tix], ly0 = y0[tix];
                            tile_enc *E = &enc[tix];
                            while (fill_vote) {
                                uint32_t sub = (uint32_t)__builtin_ctz(fill_vote);
                                uint32_t fill_ix = j + sub;
                                uint32_t nxt = fill_seg_end(sc, pts, n_points, compound, fill_ix);
                                float stx = rd_f32(sc, pts + (size_t)fill_ix * 8), sty = rd_f32(sc, pts + (size_t)fill_ix * 8 + 4);
                                float enx = rd_f32(sc, pts + (size_t)nxt * 8), eny = rd_f32(sc, pts + (size_t)nxt * 8 + 4);
                                float xmin = fminf(stx, enx), ymin = fminf(sty, eny);
                                float xmax = fmaxf(stx, enx), ymax = fmaxf(sty, eny);
                                float a = eny - sty;
                                float b = stx - enx;
                                float c = -(a * stx + b * sty);
                                float left = a * (float)lx0;
                                float right = a * (float)(lx0 + PMO_TILE_W);
                                float ytop = fmaxf((float)ly0, ymin);
                                float ybot = fminf((float)(ly0 + PMO_TILE_H), ymax);
                                float top = b * ytop;
                                float bot = b * ybot;
                                float s_top_left = signf(left + (float)ly0 * b + c);
                                float s00 = signf(top + left + c);
                                float s01 = signf(top + right + c);
                                float s10 = signf(bot + left + c);
                                float s11 = signf(bot + right + c);
                                if (s_top_left == signf(a) && ymin <= (float)ly0) backdrop[tix] -= s00;
                                if (xmin < (float)lx0 && xmax > (float)lx0) {
                                    /* mix(start.y, end.y, (start.x - x0) / b) */
                                    float tt = (stx - (float)lx0) / b;
                                    float y_edge = sty + (eny - sty) * tt;
                                    if (y_edge >= (float)ly0 && y_edge < (float)(ly0 + PMO_TILE_H)) {
                                        enc_fill_edge(E, s00, y_edge);
                                        if (b > 0.0f) enc_fill(E, stx, sty, (float)lx0, y_edge);
                                        else enc_fill(E, (float)lx0, y_edge, enx, eny);
                                        any_fill[tix] = 1;
                                    } else if (s00 * s01 + s00 * s10 + s00 * s11 < 3.0f) {
                                        enc_fill(E, stx, sty, enx, eny);
                                        any_fill[tix] = 1;
                                    }
                                } else if (s00 * s01 + s00 * s10 + s00 * s11 < 3.0f &&
                                           xmin < (float)(lx0 + PMO_TILE_W) && xmax > (float)lx0) {
                                    enc_fill(E, stx, sty, enx, eny);
                                    any_fill[tix] = 1;
                                }
                                fill_vote &= ~(1u << sub);
                            }
                        }
                    }
                    for (int t = 0; t < LANES; t++) { /* :359-363 (state only changes under hit) */
                        if (any_fill[t]) enc_draw_fill(&enc[t], rgba, (int)backdrop[t], even_odd);
                        /* a tile wholly inside: covered if the winding is non-zero / odd */
                        else if (even_odd ? (((int)backdrop[t]) & 1) != 0 : backdrop[t] != 0.0f) enc_solid(&enc[t], rgba);
                    }
                    break;
                }
                case PMO_ITEM_POLY: { /* :366-445 */
                    uint32_t rgba = rd_u32(sc, item_ref + 4);
                    float width = rd_f32(sc, item_ref + 8);
                    uint32_t n_points = rd_u32(sc, item_ref + 12) - 1u; /* :369 */
                    size_t pts = rd_u32(sc, item_ref + 16);
                    int any_stroke[LANES];
                    for (int t = 0; t < LANES; t++) any_stroke[t] = 0;
                    float hw = 0.5f * width + 0.5f;
                    for (uint32_t j = 0; j < n_points; j += 32) { /* :373 */
                        uint32_t votes = 0;
                        for (uint32_t tix = 0; tix < LANES; tix++) { /* phase 1 :374-399 */
                            uint32_t poly_ix = j + tix;
                            if (poly_ix < n_points) {
                                float stx = rd_f32(sc, pts + (size_t)poly_ix * 8), sty = rd_f32(sc, pts + (size_t)poly_ix * 8 + 4);
                                float enx = rd_f32(sc, pts + (size_t)(poly_ix + 1) * 8), eny = rd_f32(sc, pts + (size_t)(poly_ix + 1) * 8 + 4);
                                float xmin = fminf(stx, enx), ymin = fminf(sty, eny);
                                float xmax = fmaxf(stx, enx), ymax = fmaxf(sty, eny);
                                int ly0 = y0[tix];
                                if (ymax > (float)sy0 - hw && ymin < (float)(sy0 + STH) + hw &&
                                    xmax > (float)sx0 - hw && xmin < (float)(sx0 + STW) + hw) {
                                    float a = eny - sty;
                                    float b = stx - enx;
                                    float c = -(a * stx + b * sty);
                                    float left = a * ((float)sx0 - hw);
                                    float right = a * ((float)(sx0 + STW) + hw);
                                    float top = b * ((float)ly0 - hw);
                                    float bot = b * ((float)(ly0 + PMO_TILE_H) + hw);
                                    float s00 = signf(top + left + c);
                                    float s01 = signf(top + right + c);
                                    float s10 = signf(bot + left + c);
                                    float s11 = signf(bot + right + c);
                                    if (s00 * s01 + s00 * s10 + s00 * s11 < 3.0f) votes |= 1u << tix;
                                }
                            }
                        }
                        for (uint32_t tix = 0; tix < LANES; tix++) { /* phase 2 :406-440 */
                            if (!hit[tix]) continue;
                            const int lx0 = x0[tix], ly0 = y0[tix];
                            uint32_t poly_vote = votes;
                            while (poly_vote) {
                                uint32_t sub = (uint32_t)__builtin_ctz(poly_vote);
                                uint32_t poly_ix = j + sub;
                                float stx = rd_f32(sc, pts + (size_t)poly_ix * 8), sty = rd_f32(sc, pts + (size_t)poly_ix * 8 + 4);
                                float enx = rd_f32(sc, pts + (size_t)(poly_ix + 1) * 8), eny = rd_f32(sc, pts + (size_t)(poly_ix + 1) * 8 + 4);
                                float xmin = fminf(stx, enx), ymin = fminf(sty, eny);
                                float xmax = fmaxf(stx, enx), ymax = fmaxf(sty, eny);
                                if (ymax > (float)ly0 - hw && ymin < (float)(ly0 + PMO_TILE_H) + hw &&
                                    xmax > (float)lx0 - hw && xmin < (float)(lx0 + PMO_TILE_W) + hw) {
                                    float a = eny - sty;
                                    float b = stx - enx;
                                    float c = -(a * stx + b * sty);
                                    float hw2 = 0.5f * width + 0.5f; /* :420 */
                                    float left = a * ((float)lx0 - hw2);
                                    float right = a * ((float)(lx0 + PMO_TILE_W) + hw2);
                                    float top = b * ((float)ly0 - hw2);
                                    float bot = b * ((float)(ly0 + PMO_TILE_H) + hw2);
                                    float s00 = signf(top + left + c);
                                    float s01 = signf(top + right + c);
                                    float s10 = signf(bot + left + c);
                                    float s11 = signf(bot + right + c);
                                    if (s00 * s01 + s00 * s10 + s00 * s11 < 3.0f) {
                                        enc_line(&enc[tix], stx, sty, enx, eny);
                                        any_stroke[tix] = 1;
                                    }
                                }
                                poly_vote &= ~(1u << sub);
                            }
                        }
                    }
                    for (int t = 0; t < LANES; t++)
                        if (any_stroke[t]) enc_stroke(&enc[t], rgba, width); /* :441-443 */
                    break;
                }
                default:
                    break;
            }
            v &= v - 1; /* :447 */
        }
    }
}

/* Tile-group rows [gy0, gy1) only (a group row = PMO_TILER_GROUP_H tile rows; the other tiles stay
 * empty): threadgroups are independent, so a host with many cores can run the tile pass of one
 * frame in slices (bench.py's all-cores CPU baseline). */
pmo_ptcl *pmo_ptcl_build_rows(const uint8_t *scene, size_t scene_len, uint32_t width, uint32_t height, uint32_t gy0, uint32_t gy1) {
    if (scene_len < 8) return NULL;
    pmo_ptcl *p = (pmo_ptcl *)calloc(1, sizeof(*p));
    /* PietRenderer.m:63-67 */
    p->tiles_x = (width + PMO_TILE_W - 1) / PMO_TILE_W;
    p->tiles_y = (height + PMO_TILE_H - 1) / PMO_TILE_H;
    size_t nt = (size_t)p->tiles_x * p->tiles_y;
    p->count = (uint32_t *)calloc(nt ? nt : 1, sizeof(uint32_t));
    p->cmds = (pmo_cmd **)calloc(nt ? nt : 1, sizeof(pmo_cmd *));
    p->solid = (uint32_t *)calloc(nt ? nt : 1, sizeof(uint32_t));
    uint32_t groups_x = (p->tiles_x + PMO_TILER_GROUP_W - 1) / PMO_TILER_GROUP_W;
    uint32_t groups_y = (p->tiles_y + PMO_TILER_GROUP_H - 1) / PMO_TILER_GROUP_H;
    /* nested groups (extension): the tile pass walks the flat form */
    uint8_t *flat = NULL;
    size_t root = 0;
    {
        uint32_t n0 = 0, it0 = 0;
        memcpy(&n0, scene, 4);
        memcpy(&it0, scene + 4, 4);
        int nested = 0;
        if ((size_t)it0 + 32ull * n0 <= scene_len)
            for (uint32_t i = 0; i < n0; i++) {
                uint32_t tg;
                memcpy(&tg, scene + it0 + 32ull * i, 4);
                if ((tg & 0xffffu) == PMO_ITEM_GROUP) nested = 1;
            }
        if (nested) {
            size_t cap = scene_len * 8 + 4096; /* a flat group is at most 40 B per item reached */
            int64_t fl = -1;
            for (int tries = 0; tries < 6 && fl < 0; tries++, cap *= 4) {
                free(flat);
                flat = (uint8_t *)malloc(cap);
                fl = pmo_scene_flatten_groups(scene, scene_len, flat, cap, &root);
            }
            if (fl < 0) {
                free(flat);
                pmo_ptcl_free(p);
                return NULL;
            }
            scene = flat;
            scene_len = (size_t)fl;
        }
    }
    scene_t sc = {scene, scene_len, 0, root};
    tile_enc enc[LANES];
    memset(enc, 0, sizeof(enc));
    if (gy1 > groups_y) gy1 = groups_y;
    for (uint32_t gy = gy0; gy < gy1; gy++) {
        for (uint32_t gx = 0; gx < groups_x; gx++) {
            run_group(&sc, gx, gy, enc);
            for (uint32_t t = 0; t < LANES; t++) {
                uint32_t tx = gx * PMO_TILER_GROUP_W + (t & 15);
                uint32_t ty = gy * PMO_TILER_GROUP_H + (t >> 4);
                uint32_t solid = enc_end(&enc[t]); /* :452 */
                if (tx >= p->tiles_x || ty >= p->tiles_y) continue;
                size_t ti = (size_t)ty * p->tiles_x + tx;
                p->count[ti] = enc[t].n;
                p->cmds[ti] = (pmo_cmd *)malloc(enc[t].n * sizeof(pmo_cmd));
                memcpy(p->cmds[ti], enc[t].cmds, enc[t].n * sizeof(pmo_cmd));
                p->solid[ti] = solid;
            }
        }
    }
    for (int t = 0; t < LANES; t++) free(enc[t].cmds);
    free(flat);
    if (sc.oob) {
        pmo_ptcl_free(p);
        return NULL;
    }
    return p;
}

pmo_ptcl *pmo_ptcl_build(const uint8_t *scene, size_t scene_len, uint32_t width, uint32_t height) {
    return pmo_ptcl_build_rows(scene, scene_len, width, height, 0, 0xffffffffu);
}

void pmo_ptcl_free(pmo_ptcl *p) {
    if (!p) return;
    size_t nt = (size_t)p->tiles_x * p->tiles_y;
    for (size_t i = 0; i < nt; i++) free(p->cmds[i]);
    free(p->cmds);
    free(p->count);
    free(p->solid);
    free(p);
}

uint32_t pmo_ptcl_tiles_x(const pmo_ptcl *p) { return p->tiles_x; }
uint32_t pmo_ptcl_tiles_y(const pmo_ptcl *p) { return p->tiles_y; }
uint32_t pmo_ptcl_count(const pmo_ptcl *p, uint32_t tx, uint32_t ty) {
    return p->count[(size_t)ty * p->tiles_x + tx];
}
const pmo_cmd *pmo_ptcl_cmds(const pmo_ptcl *p, uint32_t tx, uint32_t ty) {
    return p->cmds[(size_t)ty * p->tiles_x + tx];
}
uint32_t pmo_ptcl_solid(const pmo_ptcl *p, uint32_t tx, uint32_t ty) {
    return p->solid[(size_t)ty * p->tiles_x + tx];
}
uint64_t pmo_ptcl_total_cmds(const pmo_ptcl *p, uint32_t *max_per_tile) {
    uint64_t tot = 0;
    uint32_t mx = 0;
    size_t nt = (size_t)p->tiles_x * p->tiles_y;
    for (size_t i = 0; i < nt; i++) {
        tot += p->count[i];
        if (p->count[i] > mx) mx = p->count[i];
    }
    if (max_per_tile) *max_per_tile = mx;
    return tot;
}
