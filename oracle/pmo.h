/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement, in plain C, of the piet-metal hot path
 * (reference = /root/reference, linebender/piet-metal):
 *
 *   src/lib.rs:15-254          scene structs + Encoder          -> pmo_encoder.c
 *   src/lib.rs:257-385         test scenes, thin-line rule      -> pmo_encoder.c
 *   src/flatten.rs:10-47       cubic -> polyline                -> pmo_flatten.c
 *   TestApp/PietRender.metal:69-157, :160-454  TileEncoder + tileKernel -> pmo_tile.c
 *   TestApp/PietRender.metal:49-60, :457-566   stroke/renderDf/renderKernel -> pmo_render.c
 *   TestApp/PietRender.metal:16-44             composite (solid tile vs per-pixel) -> pmo_render.c
 *
 * PARITY UNPINNED: the reference holds no tests, golden images or known-answer
 * vectors and cannot be built here (Rust + Metal, crates
 * un-vendored).  Third-party arithmetic restated from the published algorithm:
 * kurbo 0.5.6 `CubicBez::to_quads` / `eval` (Cargo.lock:8-14).  The Metal
 * shaders are compiled with MTL_FAST_MATH=YES, so "the reference" here means
 * the source-level semantics with IEEE-754 arithmetic, no contraction, and
 * the decisions D1-D8 of SURVEY.md section 3.3, each marked where it is taken.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * import, call, link or execute anything under oracle/.
 */
#ifndef PMO_H
#define PMO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants: TestApp/PietShaderTypes.h:17-32 ------------------------ */
#define PMO_TILE_W 16
#define PMO_TILE_H 16
#define PMO_TILER_GROUP_W 16
#define PMO_TILER_GROUP_H 2

/* ---- scene item tags: src/lib.rs:70-77, TestApp/GenTypes.h:325-328 ----- */
#define PMO_ITEM_CIRCLE 1
/* Extension (decision D10): bit 16 of a Circle's item_type word (the reference reads the tag as a
 * ushort, PietRender.metal:216) asks for the ellipse inscribed in the item's bbox -- the shading
 * PietRender.metal:488-489 leaves as a TODO.  Carried to the tile's list in CmdCircle's padding word. */
#define PMO_CIRCLE_ELLIPSE 0x10000u
#define PMO_ITEM_LINE 2
#define PMO_ITEM_FILL 3
#define PMO_ITEM_POLY 4
/* Extensions beyond the reference (product and oracle define them together):
 * a nested group as an item of its parent (src/lib.rs:148 "when we have nested groups"), and the
 * winding-rule bit of PietFill.flags (src/lib.rs:54, TestApp/SceneEncoder.h:44) selecting the
 * even-odd formula the reference leaves in a comment (TestApp/PietRender.metal:539-540). */
#define PMO_ITEM_GROUP 5        /* {item_type, flags, group_ix}: group_ix = offset of a SimpleGroup */
#define PMO_FILL_EVEN_ODD 1u
/* Extension (decision D11, "need to deal with subpaths", src/lib.rs:194): PietFill.flags bit 1 = the
 * point array holds SEVERAL closed sub-paths that share one winding sum and one DrawFill.  Every
 * sub-path is followed by a separator entry {x = NaN, y = bits of the index of the sub-path's first
 * point}; n_points counts points and separators.  Segment k runs from point k to point k + 1, or --
 * when entry k + 1 is a separator -- back to the sub-path's first point; separators start no segment. */
#define PMO_FILL_COMPOUND 2u
#define PMO_ITEM_SIZE 32  /* sizeof(union PietItem), src/lib.rs:27-31 */
#define PMO_BBOX_SIZE 8   /* ShortBbox, src/lib.rs:22-24 */
#define PMO_GROUP_HDR 8   /* SimpleGroup, src/lib.rs:15-20 */

/* ---- per-tile command tags: TestApp/GenTypes.h:440-495 ----------------- */
#define PMO_CMD_END 1
#define PMO_CMD_CIRCLE 2
#define PMO_CMD_LINE 3
#define PMO_CMD_FILL 4
#define PMO_CMD_STROKE 5
#define PMO_CMD_FILL_EDGE 6
#define PMO_CMD_DRAW_FILL 7
#define PMO_CMD_SOLID 8
#define PMO_CMD_BAIL 9

/* 24-byte command, TestApp/GenTypes.h:430-433 */
typedef struct {
    uint32_t tag;
    uint32_t body[5];
} pmo_cmd;

/* ---- path elements (kurbo::PathEl), input of the flatten stage --------- */
#define PMO_EL_MOVE 0
#define PMO_EL_LINE 1
#define PMO_EL_QUAD 2
#define PMO_EL_CURVE 3
#define PMO_EL_CLOSE 4

typedef struct {
    uint32_t tag;
    uint32_t pad;
    double p[6]; /* up to three points x,y */
} pmo_path_el;  /* 56 bytes */

#define PMO_PATH_FILL 1u
#define PMO_PATH_STROKE 2u
#define PMO_PATH_EVEN_ODD 4u
#define PMO_PATH_COMPOUND 8u /* the sub-paths of a filled path become ONE compound Fill item (holes) */

typedef struct {
    uint32_t el_begin;     /* first element index */
    uint32_t el_end;       /* one past last */
    uint32_t flags;        /* PMO_PATH_FILL | PMO_PATH_STROKE */
    uint32_t fill_rgba;    /* 0xRRGGBBAA as parse_color returns, src/lib.rs:375-385 */
    uint32_t stroke_rgba;  /* 0xRRGGBBAA */
    float stroke_width;    /* already multiplied by scale, src/lib.rs:319-320 */
} pmo_path;     /* 24 bytes */

/* ---- encoder (src/lib.rs:79-254) ---------------------------------------- */
typedef struct {
    uint8_t *buf;
    size_t cap;
    size_t free_space;
    size_t group_count;
    size_t group_ix;
    size_t group_start;
    int error; /* set instead of panicking */
    int open;  /* a group is being filled (extension: begin_group then nests) */
    int depth;
    size_t stack[32][3]; /* enclosing groups: {count, ix, start} */
} pmo_encoder;

void pmo_encoder_init(pmo_encoder *e, uint8_t *buf, size_t cap);
size_t pmo_encoder_alloc(pmo_encoder *e, size_t size);
void pmo_encoder_begin_group(pmo_encoder *e, size_t n_items);
void pmo_encoder_end_group(pmo_encoder *e);
void pmo_encoder_circle(pmo_encoder *e, double cx, double cy, double r);
/* sub_counts[n_sub] points per sub-path, pts_xy holds them back to back (extension D11) */
void pmo_encoder_fill_compound(pmo_encoder *e, const double *pts_xy, const uint32_t *sub_counts, size_t n_sub, uint32_t rgba,
                               uint32_t fill_flags);
void pmo_encoder_ellipse(pmo_encoder *e, double cx, double cy, double rx, double ry); /* extension D10 */
void pmo_encoder_stroke_line(pmo_encoder *e, double x0, double y0, double x1, double y1,
                             float width, uint32_t rgba);
void pmo_encoder_fill(pmo_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba);
void pmo_encoder_fill_rule(pmo_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba,
                           uint32_t fill_flags);
void pmo_encoder_polyline(pmo_encoder *e, const double *pts_xy, size_t n_points,
                          uint32_t rgba, float width);

/* A scene with PMO_ITEM_GROUP items means: the same items inlined, depth first, in paint order.
 * Writes that flat form (the original bytes followed by one flat SimpleGroup) into out (cap
 * out_cap) and returns its length, *root_out = offset of the flat group; returns the input
 * unchanged (root 0) if it has no nested group, -1 on malformed input or capacity. */
int64_t pmo_scene_flatten_groups(const uint8_t *scene, size_t scene_len, uint8_t *out, size_t out_cap,
                                 size_t *root_out);

/* Scenes of src/lib.rs:257-284.  Return bytes used, or -1 on error. */
int64_t pmo_scene_cardioid(uint8_t *buf, size_t cap);
int64_t pmo_scene_path_test(uint8_t *buf, size_t cap);

/* make_tiger's two passes (src/lib.rs:293-327) on already-parsed paths:
 * `affine` = [a b c d e f] applied as kurbo Affine * Point
 * (x' = a*x + c*y + e, y' = b*x + d*y + f), tolerance as src/lib.rs:330.
 * Returns bytes used or -1; *n_items_out receives the item count. */
int64_t pmo_scene_from_paths(uint8_t *buf, size_t cap, const pmo_path *paths, size_t n_paths,
                             const pmo_path_el *els, size_t n_els, const double affine[6],
                             uint32_t *n_items_out);

/* Segments a cubic is cut into (kurbo to_quads' n, libm-free: smallest n >= 1 with n^6 >= x). */
size_t pmo_subdivision_count(double x);

/* src/flatten.rs:10-47 on elements [el_begin, el_end) after `affine`.
 * Writes subpath point counts into sub_counts (cap sub_cap) and points (x,y
 * doubles) into pts (cap pts_cap points).  Returns number of subpaths or -1 if
 * a capacity was exceeded (counts are still returned in n_points_out). */
int64_t pmo_flatten_path(const pmo_path_el *els, uint32_t el_begin, uint32_t el_end,
                         const double affine[6], double tolerance, uint32_t *sub_counts,
                         size_t sub_cap, double *pts, size_t pts_cap, size_t *n_points_out);

/* ---- tileKernel (PietRender.metal:160-454) ------------------------------ */
typedef struct pmo_ptcl pmo_ptcl; /* per-tile command lists for one viewport */

/* Build per-tile command lists for a width x height viewport.  Lists are
 * unbounded (reference quirk Q5: its 4096-byte tile buffer silently overflows
 * past 170 commands; the oracle defines overflow as "list keeps growing"). */
pmo_ptcl *pmo_ptcl_build(const uint8_t *scene, size_t scene_len, uint32_t width, uint32_t height);
void pmo_ptcl_free(pmo_ptcl *p);
uint32_t pmo_ptcl_tiles_x(const pmo_ptcl *p);
/* tile-group rows [gy0, gy1) only (2 tile rows per group row); the other tiles stay empty */
pmo_ptcl *pmo_ptcl_build_rows(const uint8_t *scene, size_t scene_len, uint32_t width, uint32_t height, uint32_t gy0, uint32_t gy1);
uint32_t pmo_ptcl_tiles_y(const pmo_ptcl *p);
/* Number of commands of tile (tx,ty) including the terminating End, or 1 for a
 * Bail tile (list is then just {Bail}). */
uint32_t pmo_ptcl_count(const pmo_ptcl *p, uint32_t tx, uint32_t ty);
const pmo_cmd *pmo_ptcl_cmds(const pmo_ptcl *p, uint32_t tx, uint32_t ty);
/* TileEncoder::end() return value = loTexture texel (0 => per-pixel tile). */
uint32_t pmo_ptcl_solid(const pmo_ptcl *p, uint32_t tx, uint32_t ty);
/* total command count over all tiles (End/Bail included) and max per tile */
uint64_t pmo_ptcl_total_cmds(const pmo_ptcl *p, uint32_t *max_per_tile);

/* ---- renderKernel + composite (PietRender.metal:457-566, :16-44) -------- */
#define PMO_MODE_HALF 0u  /* accumulators binary16, as the source declares */
#define PMO_MODE_F32 1u   /* accumulators f32 (SURVEY D7 second oracle mode) */
#define PMO_FMT_RGBA8 0u
#define PMO_FMT_BGRA8 4u  /* reference drawable is BGRA8Unorm (PietRenderer.m:29) */

/* Render tile rows [ty0, ty1) of the viewport into `out` (tightly packed,
 * stride = width*4, row 0 = pixel row ty0*16).  flags = mode | fmt. */
int pmo_render_rows(const pmo_ptcl *p, uint32_t width, uint32_t height, uint32_t ty0,
                    uint32_t ty1, uint32_t flags, uint8_t *out);
/* Whole pipeline: tileKernel + renderKernel + composite. */
int pmo_render(const uint8_t *scene, size_t scene_len, uint32_t width, uint32_t height,
               uint32_t flags, uint8_t *out);
/* Per-pixel winding coverage (alpha before colour) of ONE Fill item, f32
 * accumulation, for the "coverage within 1 ULP" check.  out = width*height f32. */
int pmo_fill_coverage(const uint8_t *scene, size_t scene_len, uint32_t item_ix, uint32_t width,
                      uint32_t height, float *out);

/* Lookup tables that pin decisions D2/D3/D4 (see pmo_render.c). */
void pmo_lut_srgb_to_linear_half(uint16_t out[256]);      /* D3 */
void pmo_lut_unorm_to_half(uint16_t out[256]);            /* alpha = a/255 */
void pmo_lut_linear_half_to_srgb8(uint8_t out[65536]);    /* D2 + D4 */

#ifdef __cplusplus
}
#endif
#endif /* PMO_H */
