/*
 * piet_metal_amd -- C ABI of the MI355X-native (gfx950) replacement for the
 * compute path of linebender/piet-metal.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to the piet-metal repository).  No C++/torch types cross this boundary:
 * plain pointers, sizes and status codes only.  Nothing here ever throws or
 * unwinds; the reference's Rust panics / NSLog+nil become negative status codes.
 *
 * One pm_ctx = one GPU + one HIP stream.  A ctx is not thread-safe (the
 * reference renderer is only ever driven from the main thread,
 * TestApp/PietRenderer.m:59).  Multi-GPU = one process and one ctx per GPU,
 * each rendering a band of tile rows (pm_set_band); the bands are gathered by
 * the host layer over RCCL.
 */
#ifndef PIET_METAL_AMD_H
#define PIET_METAL_AMD_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h> /* ssize_t */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------- */
#define PM_OK 0
#define PM_ERR_INVALID (-1)   /* bad argument / encoder misuse (Rust assert!, src/lib.rs:147,152) */
#define PM_ERR_NO_DEVICE (-2) /* no usable gfx950 device: the product has no CPU fallback */
#define PM_ERR_HIP (-3)       /* HIP runtime error (see pm_last_error) */
#define PM_ERR_CAPACITY (-4)  /* scene buffer / arena too small */
#define PM_ERR_SCENE (-5)     /* malformed scene buffer */
#define PM_ERR_PARSE (-6)     /* SVG / path-data syntax error */

/* ---- constants kept from TestApp/PietShaderTypes.h:17-18 -------------------- */
#define PM_TILE_W 16
#define PM_TILE_H 16
/* maxTilesWidth/Height (256) and tileBufSize (4096) of PietShaderTypes.h:27-32 are
 * gone: the tile grid and the per-tile command storage are dynamic. */

/* ==== 1. scene producer ====================================================== */

/* Drop-in for include/piet_metal.h:3 (impl src/lib.rs:387-393): fills `buf`
 * with the Ghostscript Tiger scene at scale 8, flattened ON DEVICE 0 and copied
 * back.  Same signature, no status channel (errors leave buf untouched and are
 * readable through pm_last_error()).
 * PARITY UNPINNED: the bytes equal the in-repo oracle's (oracle/, a restatement of make_tiger +
 * flatten.rs + the kurbo 0.5.6 / roxmltree steps they call), not a buffer produced by the Rust
 * encoder -- none can be produced in this image; the Tiger's 6 arc paths and the current point
 * after `Z` are product-defined. */
void init_test_scene(uint8_t *buf, ssize_t buf_size);

/* Encoder, mirrors `impl Encoder` (src/lib.rs:103-254) method for method.
 * Pure host code: it only writes bytes into the caller's buffer. */
typedef struct pm_encoder pm_encoder;
pm_encoder *pm_encoder_new(uint8_t *buf, size_t cap);                 /* Encoder::new      :104 */
void pm_encoder_free(pm_encoder *e);
size_t pm_encoder_alloc(pm_encoder *e, size_t size);                  /* Encoder::alloc    :114 */
/* Encoder::write_struct :122 (`pub unsafe fn write_struct<T>(&mut self, ix: usize, s: &T)`): copies the
 * `len` bytes of a #[repr(C)] value to offset `ix` of the scene buffer.  The Rust slice index panics
 * past the end; here nothing is written and PM_ERR_CAPACITY is returned (and stays the encoder's status). */
int pm_encoder_write_struct(pm_encoder *e, size_t ix, const void *src, size_t len);
/* Encoder::encode_points :224 (`pub fn encode_points(&mut self, points: &[Point]) -> (usize, Rect)`):
 * allocates n_points (f32, f32) pairs, writes the points rounded to f32 and returns their offset in
 * *points_ix and the f64 bounding box {x0, y0, x1, y1} in bbox.  An empty slice is the reference's
 * .expect("encoded empty points vector") panic: PM_ERR_INVALID. */
int pm_encoder_encode_points(pm_encoder *e, const double *pts_xy, size_t n_points, size_t *points_ix, double bbox[4]);
int pm_encoder_begin_group(pm_encoder *e, size_t n_items);            /* begin_group       :132 */
int pm_encoder_end_group(pm_encoder *e);                              /* end_group         :146 */
int pm_encoder_circle(pm_encoder *e, double cx, double cy, double r); /* circle            :167 */
int pm_encoder_stroke_line(pm_encoder *e, double x0, double y0, double x1, double y1,
                           float width, uint32_t rgba);               /* stroke_line       :177 */
int pm_encoder_fill(pm_encoder *e, const double *pts_xy, size_t n_points,
                    uint32_t rgba);                                   /* fill              :195 */
int pm_encoder_polyline(pm_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba,
                        float width);                                 /* polyline          :209 */
/* Growth beyond the reference's Encoder, in its own terms:
 *  - PietFill.flags (src/lib.rs:54 "flags", TestApp/SceneEncoder.h:44 "will be used for winding
 *    number rule"): bit 0 selects the even-odd rule, the formula the reference leaves in a
 *    comment (TestApp/PietRender.metal:539-540);
 *  - nested groups (src/lib.rs:148 "when we have nested groups"): pm_encoder_begin_group inside
 *    an open group starts a child group that takes one item slot of its parent (item type 5,
 *    PietGroup {item_type, flags, group_ix}); pm_encoder_end_group closes the innermost group.
 *    A scene with groups renders exactly like the same items inlined in paint order. */
#define PM_FILL_EVEN_ODD 1u
/* Beyond the reference (src/lib.rs:194, "need to deal with subpaths"): ONE Fill item made of
 * n_subpaths closed sub-paths (sub_counts[k] points each, back to back in pts_xy) that share a
 * winding sum -- holes and overlapping contours resolve by fill_flags' rule.  Encoded as PietFill
 * with flags bit 1 (PM_FILL_COMPOUND); every sub-path is followed by a separator entry
 * {x = NaN, y = bits of the index of its first point}, n_points counts both. */
#define PM_FILL_COMPOUND 2u
int pm_encoder_fill_compound(pm_encoder *e, const double *pts_xy, const uint32_t *sub_counts, size_t n_subpaths, uint32_t rgba,
                             uint32_t fill_flags);
/* Beyond the reference (PietRender.metal:488-489, "I should make this shade an ellipse properly"):
 * the ellipse inscribed in the bbox of (cx +- rx, cy +- ry), black like the circle.  Encoded as a
 * Circle item with bit 16 of its item_type word set (the reference reads the tag as a ushort). */
int pm_encoder_ellipse(pm_encoder *e, double cx, double cy, double rx, double ry);
int pm_encoder_fill_rule(pm_encoder *e, const double *pts_xy, size_t n_points, uint32_t rgba,
                         uint32_t fill_flags);
size_t pm_encoder_bytes_used(const pm_encoder *e);                    /* free_space */

/* The reference's two other test scenes (host only, no flattening involved).
 * Return bytes written or a negative status. */
int64_t pm_scene_cardioid(uint8_t *buf, size_t cap);  /* make_cardioid  src/lib.rs:257-270 */
int64_t pm_scene_path_test(uint8_t *buf, size_t cap); /* make_path_test src/lib.rs:273-284 */

/* ==== 2. SVG front-end (replaces roxmltree + kurbo::BezPath::from_svg as used
 *         by make_tiger, src/lib.rs:286-328) ==================================== */

#define PM_EL_MOVE 0
#define PM_EL_LINE 1
#define PM_EL_QUAD 2
#define PM_EL_CURVE 3
#define PM_EL_CLOSE 4

typedef struct {
    uint32_t tag; /* PM_EL_* (kurbo::PathEl) */
    uint32_t pad;
    double p[6];  /* up to three points, x then y */
} pm_path_el;     /* 56 bytes */

#define PM_PATH_FILL 1u
#define PM_PATH_STROKE 2u
#define PM_PATH_EVEN_ODD 4u /* fill-rule="evenodd": the fill item gets PM_FILL_EVEN_ODD */
#define PM_PATH_COMPOUND 8u /* the path's sub-paths become ONE compound Fill item (PM_FILL_COMPOUND): holes work */

typedef struct {
    uint32_t el_begin, el_end; /* element range of this <path> */
    uint32_t flags;            /* PM_PATH_FILL | PM_PATH_STROKE (attribute present) */
    uint32_t fill_rgba;        /* 0xRRGGBBAA, parse_color src/lib.rs:375-385 */
    uint32_t stroke_rgba;
    float stroke_width;        /* user units (NOT yet scaled) */
} pm_path;                     /* 24 bytes */

#define PM_SVG_REJECT_ARC_PATHS 1 /* drop any path whose data holds A/a */
#define PM_SVG_FLAT_GRADIENTS 4   /* url(#gradient) paints become ONE colour, the mean of the gradient's stops (the renderer has
                                     no gradients; default: such a paint is `none`, the element is not drawn with it) */
#define PM_SVG_SPEC_DEFAULTS 2    /* SVG's initial `fill: black`, and the sub-paths of a path are filled TOGETHER
                                     (PM_PATH_COMPOUND: holes); default: make_tiger's rules -- only a fill property
                                     (own or inherited) fills (src/lib.rs:299), every sub-path is its own item (:343) */

/* Beyond what make_tiger reads (d / fill / stroke / stroke-width of every <path>), pm_svg_parse
 * understands: <g>/<svg> nesting with inherited presentation properties, `transform`
 * (matrix translate scale rotate skewX skewY), `style="..."`, <style> sheets with element / .class /
 * #id selectors (presentation attributes < element < class < id rules < style attribute), opacity / fill-opacity /
 * stroke-opacity (folded into the items' alpha: no group compositing), display: none / visibility, fill-rule (evenodd ->
 * PM_PATH_EVEN_ODD), #rgb / #rrggbb / rgb() rgba() hsl() hsla() / the 147 colour keywords / none, and rect (rounded too),
 * circle, ellipse, line, polyline, polygon as paths; <use> (href / xlink:href, x, y: the referenced
 * element or <symbol>, from anywhere in the document, nested at most 8 deep); <defs> and friends
 * are not drawn where they stand; lengths in px / pt / pc / mm / cm / in and percentages of the outermost
 * viewBox.  Coordinates
 * come out in the root user space; stroke widths are scaled by sqrt|det| of the matrix. */

/* Beyond the reference (src/lib.rs:194, "need to deal with subpaths and also want curves"): the
 * encoder takes a whole kurbo-style path -- what make_tiger's encode_path / encode_path_stroke do
 * (src/lib.rs:342-367) for one BezPath under the identity transform: flatten.rs on the host
 * (tolerance 0.1), then one Fill per sub-path (or ONE compound Fill with PM_FILL_COMPOUND) resp. one
 * poly-line per sub-path with the thin-line rule.  Same bytes as pm_flatten_and_encode on that path. */
int pm_encoder_fill_path(pm_encoder *e, const pm_path_el *els, size_t n_els, uint32_t rgba, uint32_t fill_flags);
int pm_encoder_stroke_path(pm_encoder *e, const pm_path_el *els, size_t n_els, uint32_t rgba, float width);

typedef struct pm_svg pm_svg;
pm_svg *pm_svg_parse(const char *text, size_t len, int flags, int *err);
pm_svg *pm_svg_tiger(int flags, int *err); /* the embedded Ghostscript_Tiger.svg (src/lib.rs:288) */
void pm_svg_free(pm_svg *s);
size_t pm_svg_n_paths(const pm_svg *s);
size_t pm_svg_n_els(const pm_svg *s);
/* The outermost <svg>: returns 1 and fills viewbox = {min-x, min-y, width, height} if it has a valid
 * viewBox (else 0, viewbox zeroed); width / height = its size attributes in px (0: absent, or a
 * percentage).  Path coordinates stay in user space: mapping the viewBox to pixels is the caller's affine. */
int pm_svg_viewbox(const pm_svg *s, double viewbox[4], double *width, double *height);
const pm_path *pm_svg_paths(const pm_svg *s);
const pm_path_el *pm_svg_els(const pm_svg *s);
uint32_t pm_parse_color(const char *s); /* parse_color src/lib.rs:375-385 */

/* ==== 3. renderer (replaces PietRenderer, TestApp/PietRenderer.{h,m}) ========= */

typedef struct pm_ctx pm_ctx;

/* -initWithMetalKitView: (PietRenderer.m:23-57): device, pipelines, 16 MiB scene
 * buffer.  Returns NULL and sets *err on failure (reference: NSLog + nil). */
pm_ctx *pm_create(int device, int *err);
void pm_destroy(pm_ctx *c);
const char *pm_last_error(void);

/* The ABI this library was built with, as 100 x round + revision: a caller compiled against this header checks
 * pm_abi_version() == PM_ABI_VERSION before it hands structs over (round-5 advisor: pm_scene_timings was 16 bytes in round 4 and is 12
 * again since round 5, and nothing at run time told the two layouts apart).  Libraries before round 6 do not export the symbol. */
#define PM_ABI_VERSION 600u
uint32_t pm_abi_version(void);

/* -mtkView:drawableSizeWillChange: (PietRenderer.m:105-146) minus the scene
 * init: (re)allocates the framebuffer and the dynamic tile grid. */
int pm_resize(pm_ctx *c, uint32_t width, uint32_t height);
/* Render only tile rows [row0,row1) of the viewport (multi-GPU sharding). The
 * framebuffer then holds just that band, row 0 = pixel row row0*16. */
int pm_set_band(pm_ctx *c, uint32_t tile_row0, uint32_t tile_row1);

/* _sceneBuf.contents (PietRenderer.m:52-53, :204): pinned host staging buffer the
 * caller encodes into; pm_upload_scene makes `bytes` of it resident in HBM. */
/* The pointer stays valid until pm_scene_reserve asks for more (or pm_destroy): uploads, device
 * flattening and nested-group scenes grow only the device copy. */
uint8_t *pm_scene_buffer(pm_ctx *c, size_t *cap);
int pm_scene_reserve(pm_ctx *c, size_t cap);
int pm_upload_scene(pm_ctx *c, size_t bytes);
/* flatten.rs moved on-device: flatten + encode make_tiger-style paths directly
 * into the device scene buffer (src/lib.rs:293-327, src/flatten.rs:10-47).
 * affine = kurbo Affine coefficients [a b c d e f]; stroke widths are multiplied
 * by `width_scale` in f32 (src/lib.rs:320).  On return the scene is resident.
 * Limits: n_els and n_paths <= 2^26 - 1 (PM_ERR_CAPACITY beyond). */
int pm_flatten_and_encode(pm_ctx *c, const pm_path *paths, size_t n_paths,
                          const pm_path_el *els, size_t n_els, const double affine[6],
                          float width_scale, size_t *scene_bytes, uint32_t *n_items);
/* Animation: flatten + encode the paths of the last pm_flatten_and_encode AGAIN under another
 * affine (they stay resident on the device; nothing is uploaded or allocated).  The reference
 * re-encodes on the CPU when the view changes (PietRenderer.m:90-101, :145); here a view change
 * is four small kernels plus the scene index. */
int pm_reflatten(pm_ctx *c, const double affine[6], float width_scale, size_t *scene_bytes, uint32_t *n_items);
/* Copy the resident scene back (parity checks, init_test_scene). */
int pm_download_scene(pm_ctx *c, uint8_t *dst, size_t cap, size_t *bytes);

/* -drawInMTKView: (PietRenderer.m:59-103): enqueue one frame (binning + per-pixel
 * kernels) on the ctx stream; asynchronous like [commandBuffer commit]. */
int pm_render(pm_ctx *c);
/* Same, into a caller-owned device buffer (e.g. a torch tensor) on a caller
 * stream (hipStream_t; NULL = ctx stream).  stride in bytes, multiple of 4. */
int pm_render_to(pm_ctx *c, void *dev_framebuffer, size_t stride_bytes, void *hip_stream);
int pm_sync(pm_ctx *c);

#define PM_FMT_RGBA8 0
#define PM_FMT_BGRA8 1 /* the reference drawable's byte order, PietRenderer.m:29 */
/* Byte order in which pm_render / pm_render_to STORE pixels from now on: PM_FMT_RGBA8 (default) or
 * PM_FMT_BGRA8 = MTLPixelFormatBGRA8Unorm, what the reference's renderKernel writes into
 * (view.colorPixelFormat, PietRenderer.m:29) -- a caller-owned BGRA surface is then filled directly,
 * no swizzle pass.  Same pixels, R and B exchanged. */
int pm_set_target_format(pm_ctx *c, int fmt);
/* Read the (band of the) framebuffer back: height rows of width*4 bytes, in byte order `fmt`
 * (swizzled on the host if the frame was stored in the other order). */
int pm_read_pixels(pm_ctx *c, uint8_t *dst, size_t dst_stride, int fmt);
void *pm_framebuffer_device_ptr(pm_ctx *c, size_t *stride_bytes, uint32_t *rows);
/* The resident scene on the device.  Valid until the scene is next replaced (pm_upload_scene,
 * pm_flatten_and_encode, pm_reflatten): the device flatten writes the NEXT scene into a second buffer while frames
 * in flight still read this one, and the two change places -- a pointer kept across a replacement names the
 * buffer the replacement after that overwrites. */
void *pm_scene_device_ptr(pm_ctx *c, size_t *bytes);

/* Time `iters` frames with HIP events (any pointer may be NULL).
 * total_ms = the whole batch submitted back to back through the frame pipeline, as pm_render
 * does; bin/coarse/fine/clear_ms = average per-launch duration of the frame kernels
 * (pm_bin_kernel, pm_coarse_kernel, pm_fine_kernel; pm_clear_kernel only with PM_FOLD_CLEAR=0,
 * else its work is part of pm_fine_kernel's launch and clear_ms is 0), each ALONE on the GPU: a second pass
 * of `iters` frames serialized on one stream, every launch bracketed by events. */
int pm_time_frames(pm_ctx *c, int iters, float *total_ms, float *bin_ms, float *coarse_ms, float *fine_ms, float *clear_ms);
/* Same, but the per-kernel durations are taken INSIDE the pipelined batch: every launch is
 * bracketed by events on the stream it runs on while frames overlap (iters <= 4096).  These
 * are the durations a kernel trace of pm_render traffic shows. */
int pm_time_frames_pipelined(pm_ctx *c, int iters, float *total_ms, float *bin_ms, float *coarse_ms, float *fine_ms,
                             float *clear_ms);

/* Latency of ONE frame with nothing else in flight: begin of its first kernel to end of its
 * last one, two HIP events on the frame's stream around the plain launches pm_render makes (two
 * by default: pm_bin_kernel, pm_fine_kernel);
 * median and minimum over `iters` frames (SURVEY.md 8d's t_frame; PietRenderer.m has no timing
 * at all). */
int pm_frame_latency(pm_ctx *c, int iters, float *median_ms, float *min_ms);

/* Frames whose tile kernel was the one-wave-per-tile instantiation (six workgroups per CU): what frames get after a frame
 * of the same scene and viewport whose tile kernel found it dense -- long lists enough to occupy every wave (PM_DENSE_KERNEL=0: never). */
int pm_tile_kernel_info(pm_ctx *c, uint32_t *dense_frames);

/* Frames binned with ONE wave per strip row (pm_bin_kernel<.., 1>) since pm_create: out[0] all of them, out[1] those that got it only
 * because they were submitted behind running frames, out[2] of those the ones with a wave for EVERY strip row where the plan chains
 * rows for its own, smaller grid of workgroups (1 280 < strip rows <= 4 096 light ones). */
int pm_binning_info(pm_ctx *c, uint32_t out[3]);

/* The binning plan in force (the host side of tileKernel's dispatch geometry, PietRenderer.m:63-77): out[0] entries of the work list
 * (strip rows some item reaches; a strip row cut in two counts twice), out[1] strip rows cut in two -- the heaviest rows, binned by two
 * workgroups side by side (tiles 0-7 and 8-15) while the resident grid has workgroups to spare --, out[2] plans remade since pm_create
 * from what the frames' binning kernels reported (segment slots per strip row), out[3] 1 if the plan in force was made from such a report. */
int pm_binning_plan_info(pm_ctx *c, uint32_t out[4]);

/* One launch per frame (pm_frame_kernel: the two dispatches of PietRenderer.m:69-88 as roles of one resident
 * grid).  *frames = frames submitted that way since pm_create; *applies = 1 if a frame of the resident scene
 * and viewport, alone on the device, would be (0: two launches -- PM_ONE_LAUNCH=0, more strip rows than resident
 * workgroups, per-tile-row item lists, or a one-launch frame once gave up waiting). */
int pm_one_launch_info(pm_ctx *c, uint32_t *frames, int *applies);
/* Duration of that launch, frame alone, from events carried by the dispatch itself (what a kernel trace shows);
 * average over `iters` frames.  PM_ERR_INVALID when *applies is 0. */
int pm_time_one_launch(pm_ctx *c, int iters, float *kernel_ms);
/* Developer hook: one one-launch frame by its profiling instantiation -- per workgroup 32 x u64 (10 ns clocks and
 * counts, layout at pm_frame_kernel in pm_frame.hip); *n_wgs = the launch's workgroups. */
int pm_debug_time_frame(pm_ctx *c, uint64_t *out, size_t max_wgs, size_t *n_wgs);

/* Developer hook: the lone frame taken apart -- medians over `iters` frames of {pm_bin_kernel,
 * gap, pm_coarse_kernel, gap, pm_fine_kernel, first begin -> last end}, ms, from events attached to
 * the dispatches (which stretch the gaps: a breakdown, not t_frame -- that is pm_frame_latency;
 * needs the default PM_FOLD_CLEAR=1; the coarse entries are 0 in the default fused path). */
int pm_debug_frame_timeline(pm_ctx *c, int iters, float *out6);

typedef struct {
    uint32_t tiles_x, tiles_y;    /* tile grid of the viewport */
    uint32_t band_row0, band_row1;
    uint32_t n_items;             /* scene items */
    uint32_t queued_tiles;        /* tiles handed to the per-tile kernel last frame */
    uint32_t arena_used_dwords;   /* binning arena: dwords of records written last frame */
    uint32_t arena_cap_dwords;
    uint32_t overflow;            /* 1 if the last frame ran out of arena */
    uint32_t scene_bytes;
    uint32_t heavy_tiles;         /* of queued_tiles: scheduled first (long segment streams) */
    uint32_t ptcl_used_cmds;      /* tile arena: 16-byte quads reserved last frame (per-tile pieces + command
                                     lists, a 24-byte command = 1.5 quads), summed over the arena's parts */
} pm_stats;
int pm_get_stats(pm_ctx *c, pm_stats *out); /* synchronises */

/* ==== 4. multi-GPU presentation (new: the reference is single-device, PietRenderer.m:27) =====
 * One process and one pm_ctx per GPU, each rendering a band of tile rows (pm_set_band); the
 * bands meet once, here: a grouped ncclSend/ncclRecv over RCCL (xGMI inside a node) puts every
 * rank's band straight into its rows of the root's final image.  RCCL is bound
 * with dlopen at the first call: the copy the process has mapped already if there is one (a host
 * that also runs torch.distributed keeps ONE RCCL), else librccl.so.1; PM_RCCL_LIB overrides. */
#define PM_COMM_ID_BYTES 128 /* = sizeof(ncclUniqueId) */
typedef struct pm_comm pm_comm;
/* Rank 0 makes the id; the host carries it to the other ranks (file, env, MPI, a torch store). */
int pm_comm_unique_id(uint8_t id[PM_COMM_ID_BYTES]);
/* Collective over all `world` ranks (ncclCommInitRank on the context's device). */
pm_comm *pm_comm_create(pm_ctx *c, const uint8_t id[PM_COMM_ID_BYTES], int rank, int world, int *err);
void pm_comm_destroy(pm_comm *m);
/* What got bound: the path of the shared object ncclCommInitRank came from (lib_path, NUL-terminated,
 * truncated to lib_path_cap) and ncclCommCount of the communicator (*ranks; 0 when m is NULL).  Either
 * output may be NULL.  Loads RCCL like the other calls; not a collective. */
int pm_comm_info(pm_comm *m, char *lib_path, size_t lib_path_cap, int *ranks);
/* Collective.  band_tile_rows = {row0, row1} per rank (2*world entries, what each rank passed to
 * pm_set_band; a rank whose own entry is empty, row0 == row1, sends nothing and may pass any context
 * of the device -- a band gathered in sub-bands, one context each, pipelined under the render).  src_band = this rank's band (NULL: the context's last frame), tightly packed
 * rows of width*4 bytes; dst_image (root only) = the full width*4 x height image.  Asynchronous
 * on hip_stream (NULL: the context's stream); ordered behind the last frame when src_band is NULL.
 * Errors: an argument error (PM_ERR_INVALID for bad rows, strides or a root without dst_image) is
 * returned before anything is posted -- if only SOME ranks fail that way the others wait for them, so
 * tear the communicator down; a root whose own band cannot be placed (overlap, copy failure) still
 * posts its receives, so that the peers' sends complete, and reports the error afterwards. */
int pm_gather(pm_ctx *c, pm_comm *m, const void *src_band, size_t src_stride, const uint32_t *band_tile_rows, int root,
              void *dst_image, size_t dst_stride, void *hip_stream);

/* Host wall-clock cost of the last scene replacement ("flatten+encode timed
 * separately"): the reference does this work once per resize on the CPU (src/lib.rs:286-328,
 * PietRenderer.m:145) and never times it.
 *   flatten_encode_ms  pm_flatten_and_encode: the four flatten kernels incl. their scan
 *                      read-backs and buffer management (0 after pm_upload_scene)
 *   scene_index_ms     header/item read-back, validation, chunk index (pm_index_kernel)
 *   arena_setup_ms     per-(scene, viewport, band) binning-arena sizing on the host, paid by the
 *                      first pm_render after a scene or viewport change */
typedef struct {
    float flatten_encode_ms, scene_index_ms, arena_setup_ms;
} pm_scene_timings; /* 12 bytes, as in round 3: the struct does not grow (a caller built against an older header owns 12 bytes) */
int pm_get_scene_timings(pm_ctx *c, pm_scene_timings *out);
/* Binning plans made since pm_create (strip-row regions sized, lists uploaded): a scene after pm_reflatten keeps the
 * plan in force while every item stays inside the box it was planned with -- one tile wider on each side -- and
 * keeps its segment count.  (Round 4 had this counter as a fourth member of pm_scene_timings; a getter of its own
 * since round 5: ABI note in INTEGRATION.md.) */
int pm_get_binning_plans(pm_ctx *c, uint32_t *plans);

/* Debug/parity hook: re-run the last frame's per-tile kernel with command
 * capture and return every tile's command list in the reference's 24-byte
 * format (TestApp/GenTypes.h:430-495), including the trailing End / the lone
 * Bail.  counts[tiles] and solid[tiles] are per tile (band-relative, row-major);
 * cmds receives max_cmds_per_tile * tiles entries.  Lists longer than
 * max_cmds_per_tile are truncated (count still reports the full length). */
typedef struct {
    uint32_t tag;
    uint32_t body[5];
} pm_cmd;
int pm_debug_capture_ptcl(pm_ctx *c, uint32_t max_cmds_per_tile, uint32_t *counts,
                          uint32_t *solid, pm_cmd *cmds);

/* Winding coverage (alpha before colour) of ONE Fill item of the resident scene over the viewport
 * band, accumulated in f32 instead of the frame path's binary16 (the reference declares
 * signedArea `half`, TestApp/PietRender.metal:472; north star: "coverage within 1 ULP of the f32
 * reference").  The item is binned and encoded alone, exactly as a frame would, then its commands
 * are evaluated per pixel in binary32.  dst = rows of `width` floats, row stride in floats.
 * A validation call: it synchronises and rebuilds the scene index twice. */
int pm_fill_coverage(pm_ctx *c, uint32_t item_ix, float *dst, size_t dst_stride_floats);

/* Developer / test hook for the generated layout code (piet_metal_amd/csrc/pm_layout_gen.h, printed by
 * pm_layoutgen from piet_metal_amd/layout/piet_layout.pgpu -- the HIP / C++ target of the reference's
 * piet-gpu-derive generator, piet-gpu-derive/src/lib.rs): every item of `scene`'s root group and
 * every command of `cmds` goes through the generated readers, loaders and writers and must come
 * back byte for byte.  0 = agreement. */
int pm_layout_selfcheck(const uint8_t *scene, size_t scene_len, const pm_cmd *cmds, size_t n_cmds);

/* Developer profiling hook: re-run the last frame's per-tile kernel recording, per queue
 * slot, {start clock, end clock (100 MHz wall clock), tile | quarter << 31,
 * wave << 32 | commands interpreted, ticks in phase A, ticks in phase B (tiles rendered by a
 * whole workgroup), clock when the tile's list was complete (fused kernel; else 0), ticks on
 * the wave's own items, then four words of list-building stages (fused kernel, packed pairs:
 * header | candidates, owners | scan, segments | emission, rounds | records)}.
 * out receives 12 u64 per slot. */
int pm_debug_time_tiles(pm_ctx *c, uint64_t *out, size_t max_slots, size_t *n_slots); /* pm_fine_kernel */
/* Same for pm_bin_kernel: renders one frame recording 12 u64 per strip row {start, item scan done,
 * headers done, segment stream done, record finalised, queues done, chunks streamed, end}. */
int pm_debug_time_bins(pm_ctx *c, uint64_t *out, size_t max_rows, size_t *n_rows);

#ifdef __cplusplus
}
#endif
#endif /* PIET_METAL_AMD_H */
