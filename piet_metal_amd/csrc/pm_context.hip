// Host runtime behind the C ABI: the MI355X replacement for PietRenderer
// (TestApp/PietRenderer.{h,m}).  One context = one GPU + one HIP stream.
//
//   -initWithMetalKitView:               -> pm_create      (PietRenderer.m:23-57)
//   -mtkView:drawableSizeWillChange:     -> pm_resize      (PietRenderer.m:105-146)
//   init_test_scene(_sceneBuf.contents)  -> pm_scene_buffer / pm_upload_scene /
//                                           pm_flatten_and_encode (PietRenderer.m:203-205)
//   -drawInMTKView:                      -> pm_render      (PietRenderer.m:59-103)
//
// Frames overlap like the reference's command buffers ([commandBuffer commit] never waits,
// PietRenderer.m:102): frame N runs its three kernels back to back on stream N % 4 and owns
// frame slot N % 4 (arena, queues, command lists, framebuffer), so frames in flight share no
// mutable state and need no cross-stream events; the in-order queues do the ordering.  At 4K
// every kernel alone leaves most of the chip idle (its span is set by its longest dependent
// chain), and four frames side by side fill it.  Measured alternatives: one stream per kernel
// stage chained by events (each cross-stream dependency costs 10-15 us to release the next
// dispatch) was 5 % slower; more than four streams (ROCm's default number of hardware queues)
// was slower still.  A frame rendered into a caller-owned buffer (pm_render_to) runs on the
// caller's stream.
//
// There is deliberately no CPU rendering path in this library: without a gfx950
// device pm_create fails with PM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/piet_metal_amd.h"
#include "pm_device.h"
#include "pm_flatten.h"
#include "pm_layout.h"

namespace {

thread_local std::string g_last_error;

void SetError(const std::string &s) { g_last_error = s; }

int HipFail(hipError_t e, const char *what) {
    SetError(std::string(what) + ": " + hipGetErrorString(e));
    return PM_ERR_HIP;
}

#define PM_TRY(expr)                                         \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) return HipFail(e_, #expr);     \
    } while (0)

// ---- binary16 helpers for the pinned lookup tables ----------------------------------------
// (host-side only; the kernels use native _Float16)

uint16_t HalfBitsFromDouble(double d) {  // one rounding, nearest-even
    if (d != d) return 0x7e00;
    uint16_t sign = 0;
    if (std::signbit(d)) {
        sign = 0x8000;
        d = -d;
    }
    if (d == 0.0) return sign;
    if (d >= 65520.0) return sign | 0x7c00;
    int e;
    const double m = std::frexp(d, &e);  // d = m * 2^e, m in [0.5, 1)
    int E = e - 1;
    if (E < -14) return sign | static_cast<uint16_t>(std::nearbyint(std::ldexp(d, 24)));
    double q = std::nearbyint(std::ldexp(m, 11));  // [1024, 2048]
    if (q == 2048.0) {
        q = 1024.0;
        E += 1;
    }
    return sign | static_cast<uint16_t>(((E + 15) << 10) | (static_cast<int>(q) - 1024));
}

float HalfBitsToFloat(uint16_t h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = std::ldexp(static_cast<float>(m), -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp(static_cast<float>(m + 1024), e - 25);
    return s ? -v : v;
}

inline uint16_t RoundHalf(float f) { return HalfBitsFromDouble(static_cast<double>(f)); }

struct Luts {
    uint32_t srgb2lin[256];
    uint32_t unorm2h[256];
    uint8_t lin2srgb[65536];
};

// The three tables pin what Metal leaves to its half library:
//   unpack_unorm4x8_srgb_to_half  : exact EOTF, rounded once to binary16
//   final select(1.055*pow(rgb,1/2.4)-0.055, 12.92*rgb, rgb<0.0031308) on half3
//   (PietRender.metal:563), constants and exponent taken in binary16, pow
//   correctly rounded to binary16, then clamp*255 round-half-even to unorm8.
void BuildLuts(Luts *l) {
    for (int i = 0; i < 256; ++i) {
        const double c = i / 255.0;
        const double lin = (c <= 0.04045) ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
        l->srgb2lin[i] = HalfBitsFromDouble(lin);
        l->unorm2h[i] = HalfBitsFromDouble(c);
    }
    const float thr = HalfBitsToFloat(RoundHalf(0.0031308f));
    const float k = HalfBitsToFloat(RoundHalf(12.92f));
    const float s = HalfBitsToFloat(RoundHalf(1.055f));
    const float o = HalfBitsToFloat(RoundHalf(0.055f));
    const double ex = HalfBitsToFloat(RoundHalf(1.0f / 2.4f));
    for (uint32_t h = 0; h < 65536; ++h) {
        const float x = HalfBitsToFloat(static_cast<uint16_t>(h));
        if (x != x) {
            l->lin2srgb[h] = 0;
            continue;
        }
        float y;
        if (x < thr) {
            y = HalfBitsToFloat(RoundHalf(k * x));
        } else {
            const float p = HalfBitsToFloat(HalfBitsFromDouble(std::pow(static_cast<double>(x), ex)));
            const float sp = HalfBitsToFloat(RoundHalf(s * p));
            y = HalfBitsToFloat(RoundHalf(sp - o));
        }
        const float cl = std::fmin(std::fmax(y, 0.0f), 1.0f);
        l->lin2srgb[h] = static_cast<uint8_t>(std::rintf(cl * 255.0f));
    }
}

}  // namespace


namespace {
constexpr int kMaxSlots = 16;
constexpr int kDefaultFrameStreams = 4;  // = ROCm's default hardware queues; measured best on Tiger 4K

int EnvInt(const char *name, int dflt, int lo, int hi) {
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    return std::max(lo, std::min(hi, std::atoi(v)));
}
}

struct FrameSlot {
    uint8_t *d_fb = nullptr;  // this slot's framebuffer (pm_render); pm_render_to uses the caller's
    size_t fb_cap = 0;        // bytes allocated for it, and entries the queue / per-tile tables were allocated for: a viewport that
    size_t tables_cap = 0;    // fits keeps the buffers (grow only: a resize to anything smaller than what came before allocates nothing)
    uint32_t *d_arena = nullptr;
    uint32_t arena_cap = 0;  // dwords allocated for this slot (allocated when the slot is first used)
    uint4 *d_queue = nullptr;
    uint32_t *d_tile_state = nullptr;
    uint32_t *d_tile_ptcl = nullptr;
    uint32_t *d_tile_ncmd = nullptr;
    uint4 *d_ptcl = nullptr;  // tile arena: per-tile pieces + command lists
    uint32_t ptcl_cap = 0;    // quads (16 B)
    uint2 *d_row_bbox = nullptr;  // per-tile-row item lists of this slot's frame (large scenes)
    uint32_t *d_row_item = nullptr;
    uint64_t row_cap = 0;         // entries allocated for them
    uint32_t state_epoch = 0;     // arena_epoch its tile_state was last reset for (0: never)
    uint32_t vp_epoch = 0;        // viewport its buffers were allocated for (pm_ctx::vp_epoch; stale ones go when the slot is next used)
    pm::Counters *d_ctr = nullptr;  // two: a frame's binning kernel zeroes the one the slot's next frame uses
    uint32_t *h_overflow = nullptr;  // pinned: raised by a frame of this slot whose tile arena ran out (what pm_sync looks at first)
    uint32_t *d_overflow = nullptr;  // ... as the device sees it ([1]: a one-launch frame of this slot gave up waiting)
    uint4 *d_fifo = nullptr;         // one-launch frames: kFifos x fifo_cap queue entries, all zero between frames (made when first needed)
    uint32_t fifo_cap = 0;
    uint32_t parity = 0;
    hipEvent_t ev_done = nullptr;  // end of the slot's last frame
    bool in_flight = false;        // a frame using this slot was submitted; ev_done marks its end
    bool needs_check = false;      // ... and pm_sync has not yet looked at its overflow flag (SyncAll only waits)
    pm::FrameParams params{};
    hipStream_t frame_stream = nullptr;  // stream the slot's last frame ran on (compared, never dereferenced, if user_stream)
    bool user_stream = false;            // the frame ran on a caller-owned stream: ev_done was recorded behind it at submit
};

struct pm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // == streams[0]: scene upload, flatten, index, debug replays
    std::vector<hipStream_t> streams;  // frame N runs on streams[N % n]; stream == streams[0]
    bool fold_clear = true;  // a frame alone: pm_fine_kernel's launch also writes the resolved tiles (no pm_clear_kernel launch)
    int fold_clear_mode = 2;  // PM_FOLD_CLEAR: 0 never, 1 always, 2 (default) per frame -- folded when the frame is alone
    bool fused = true;       // pm_fine_kernel<true>: each tile's list is built and interpreted by the same wave(s)
    int handout = 0;         // tile hand-out: 0 = drawn for a lone frame, static when frames overlap; 1 = static; 2 = drawn
    int target_fmt = PM_FMT_RGBA8;  // byte order the kernels store pixels in (pm_set_target_format)
    uint32_t split_mode = 1;  // fine kernel: long lists get 4 waves per tile (16 measured no faster)
    // A frame whose tile kernel found it dense (above) is followed by frames whose tile kernel is the one-wave-per-tile instantiation:
    // 80 VGPRs and 19 KB of LDS instead of 96 and 31, six workgroups per CU instead of five (PM_DENSE_KERNEL=0: never).  The verdict is
    // the kernel's own, left in a pinned word by every frame (FrameSlot::h_overflow[2]); a new scene or viewport starts undecided.
    int dense_kernel_mode = 1;
    uint32_t fine_wg_dense = 6, fine_wg_dense_inflight = 4;  // PM_FINE_WG_PER_CU_DENSE, PM_FINE_WG_PER_CU_DENSE_INFLIGHT
    uint32_t frames_dense_kernel = 0;  // frames whose tile kernel was that instantiation (pm_tile_kernel_info)
    uint32_t dense_factor = 4;  // PM_DENSE_FACTOR: a frame whose long lists x this would fill every wave renders each tile with one wave
    uint32_t heavy_stream = 72, heavy_stream_lone = 40, vheavy_stream = 112;  // list-length classes (PM_HEAVY_STREAM / PM_VHEAVY_STREAM)
    uint32_t bin_waves_env = 0;     // PM_BIN_WAVES: 4 / 1 waves per strip row in pm_bin_kernel (0: by the number of strip rows, EnsureArena)
    uint32_t bin_waves = 4;
    uint32_t bin_waves_inflight = 1;  // PM_BIN_WAVES_INFLIGHT: waves per strip row for frames submitted behind frames still running (1 / 4)
    uint64_t row_want = 0;          // entries of a slot's per-tile-row item lists (use_row_lists)
    uint64_t plan_cands = 0;        // (item, strip row) pairs of the plan in force: candidates the strip rows will look at
    uint32_t bin_wg_per_cu = 0xff;  // pm_bin_kernel's workgroups per CU (PM_BIN_WG_PER_CU; 0 = one per strip row, default: by the number of strip rows)
    uint32_t coarse_wg_per_cu = 5, fine_wg_per_cu = 5;  // persistent grids (PM_COARSE_WG_PER_CU, PM_FINE_WG_PER_CU)
    uint32_t fine_wg_per_cu_inflight = 3;               // ... of a frame behind other frames (PM_FINE_WG_PER_CU_INFLIGHT)
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n_cus = 0;
    // One launch per frame (pm_frame_kernel, pm_frame.hip): PM_ONE_LAUNCH=1 for every frame that has the device to itself,
    // whose strip rows fit the resident grid two to a workgroup, and that needs nothing the one kernel does not do (per-tile-row
    // item lists, a wave per strip row, separate list building).  Off by default: measured slower than two launches on every
    // scene ("one launch per frame") -- the chains that set a frame's length are strings of memory round trips, and
    // they stretch when the rest of the chip renders tiles beside them instead of idling.
    int one_launch_mode = 0;
    bool one_launch_split = false;    // PM_ONE_LAUNCH_SPLIT=1 (developer experiment): the kernel's two roles as two launches of it
    bool one_launch_broken = false;   // a frame gave up waiting inside the launch (pm_sync rendered it again): two launches from then on
    uint32_t frame_wg_per_cu = 0;     // resident workgroups of pm_frame_kernel per CU (0: cannot run five, one-launch frames are off)
    uint32_t frame_spin_ticks = 200000;  // PM_ONE_LAUNCH_SPIN_US x 100: a wait inside the launch gives up after this long
    uint32_t *d_sr_next_one = nullptr;  // [n_sr_active] chains of the one-launch grid (0: the workgroup's last strip row)
    size_t sr_next_one_cap = 0;
    uint32_t one_grid_rows = 0;       // binning workgroups of a one-launch frame (0: its strip rows do not fit two per workgroup)
    std::vector<uint32_t> stage_next_one;
    uint32_t *d_idle_sr = nullptr;    // strip rows of the band no item reaches (the launch writes their pixels too)
    size_t idle_sr_cap = 0;
    uint32_t n_idle_sr = 0;
    std::vector<uint32_t> stage_idle;
    uint32_t frames_one_launch = 0;   // frames submitted as one launch (pm_one_launch_info)
    // pm_debug_capture_ptcl of a one-launch frame: the frame is rendered once more with the capture instantiation of its kernel
    uint32_t *cap_counts = nullptr, *cap_solid = nullptr;
    pm::Cmd *cap_cmds = nullptr;
    uint32_t cap_max = 0;

    // scene
    uint8_t *h_scene = nullptr;  // pinned staging (pm_scene_buffer): only pm_scene_reserve moves it
    size_t scene_cap = 0;        // its capacity
    uint8_t *d_scene = nullptr;
    size_t dev_scene_cap = 0;    // >= scene_cap: the device copy also grows on its own (flat groups, device flatten)
    // what the flatten kernels write while frames in flight still read d_scene; the two change places when the new scene is in
    // (a view change then no longer begins by waiting for the previous frame: 0.197 -> 0.15 ms per re-encoded 4K Tiger frame)
    uint8_t *d_scene_alt = nullptr;
    size_t dev_scene_alt_cap = 0;
    size_t scene_bytes = 0;       // resident bytes (with the flat form of nested groups appended)
    size_t user_scene_bytes = 0;  // what the caller uploaded / the flatten kernels wrote
    uint32_t dev_bbox_ix = 8, dev_items_ix = 0;  // the drawn group's ShortBbox / item arrays in d_scene
    uint32_t n_items = 0;
    std::vector<uint8_t> item_meta;  // copy of header + bboxes + items (arena sizing)
    std::vector<uint32_t> chunk_base_host;  // source of the asynchronous upload of the chunk table
    std::vector<uint64_t> stage_need_diff;  // StripRowBounds' scratch
    std::vector<uint4> stage_desc;          // ... of the strip-row work list,
    std::vector<uint2> stage_list;          // ... its rows' item lists,
    std::vector<uint2> stage_bbs;           // ... the band's item boxes
    std::vector<uint32_t> stage_ids, stage_rb;  // ... and indices, the per-row list offsets
    size_t row_base_cap = 0;
    uint32_t *d_chunk_base = nullptr;  // scene index: first chunk of every item (+ total)
    float4 *d_chunk_bbox = nullptr;    // scene index: bounding box of every chunk of segments
    float4 *d_sup_bbox = nullptr;      // ... and of every super-chunk (kSuperChunks consecutive chunk-table entries)
    size_t chunk_base_cap = 0, chunk_bbox_cap = 0, sup_bbox_cap = 0;
    uint32_t n_chunks = 0;

    // viewport
    uint32_t width = 0, height = 0, tiles_x = 0, tiles_y = 0, strips_x = 0;
    uint32_t row0 = 0, row1 = 0;
    size_t fb_stride = 0;
    size_t fb_bytes = 0;
    uint32_t vp_epoch = 1;  // bumped by every resize / band change

    // binning state shared by the slots
    size_t sr_desc_cap = 0, band_cap = 0;
    uint4 *d_sr_desc = nullptr;     // strip rows some item reaches: {strip row, arena region begin, end, next of the chain}
    uint2 *d_sr_list = nullptr;     // ... and, for large scenes, where their tile row's item list is
    uint32_t n_sr_active = 0;
    uint32_t bin_grid = 1;          // workgroups of pm_bin_kernel (each walks a chain of strip rows)
    uint32_t bin_prio_slots = 1024; // PM_BIN_PRIO_SLOTS
    // Heavy strip rows cut in two (round 6): binning ends with its heaviest strip rows (a 4K Tiger frame: 1 109 rows, the mean one
    // through at 15 us, the heaviest at 24), and a row's length is dependent steps of its four waves, so the heaviest rows get EIGHT:
    // two entries of the work list, tiles 0-7 and 8-15, a workgroup each, while the plan's rows leave workgroups of the resident
    // grid free.  Which rows: those the previous frames of this scene and viewport found heaviest (segment slots per strip row, left
    // in a pinned array by pm_bin_kernel; before any frame has said anything: by candidates, PM_BIN_SPLIT_CANDS).
    int bin_wt_mode = 2;             // PM_BIN_WT: binning's output stored write-through: 0 never, 1 always, 2 frames whose strip rows all fit the resident grid
    int bin_split_mode = 1;          // PM_BIN_SPLIT: 0 never, 1 while the resident grid has room, 2 every strip row (tests)
    uint32_t bin_split_cands = 1u << 30;  // PM_BIN_SPLIT_CANDS: a row with at least this many candidates is cut before any frame has reported (default: none --
                                          // candidates predict a row's segment slots poorly: of the 4K Tiger's 60 heaviest rows they name 32)
    uint32_t bin_split_slots = 224;  // PM_BIN_SPLIT_SLOTS: ... with at least this many segment slots, once frames have reported
    uint32_t bin_split_fill = 15;    // PM_BIN_SPLIT_FILL: cut rows while the work list stays within this many sixteenths of the resident grid
    uint32_t n_sr_split = 0;         // strip rows of the plan in force that are cut in two
    uint4 *d_sr_desc_whole = nullptr;  // [sr_desc_cap] the same work list with every cut strip row whole again (same arena regions): what frames
    uint32_t n_sr_whole = 0;           // behind running frames bin from -- there instructions count, not the frame's own latency (sustained -3 % with cuts)
    std::vector<uint4> stage_desc_whole;
    uint32_t *d_sr_slots = nullptr;  // [sr_desc_cap] segment slots every entry of the work list found in the latest frame
    std::vector<uint32_t> fb_slots;  // per strip row of the band: the slots the frames of this scene and viewport reported (empty: nothing yet)
    std::vector<uint32_t> stage_slots;
    uint32_t frames_on_plan = 0;     // frames submitted since the plan in force was made
    bool fb_applied = false;         // the plan in force was made with fb_slots
    uint32_t plans_fed_back = 0;     // plans remade from the frames' report (pm_binning_info)
    uint32_t sr_empty_dwords = 0;   // size of a region no item reaches
    uint2 *d_band_bbox = nullptr;   // items that reach the band (bbox, scene index), paint order
    uint32_t *d_band_item = nullptr;
    uint32_t n_band_items = 0;
    uint32_t *d_row_base = nullptr;  // per-tile-row item lists (large scenes): offsets
    uint32_t frames_bin_wave = 0, frames_bin_wave_inflight = 0, frames_bin_no_chains = 0;  // pm_binning_info
    uint32_t row_total = 0;
    uint32_t row_parts = 1, row_part_items = pm::kRowCullStep;  // pm_rowcull_kernel's workgroups per tile row and their share of the items
    bool use_row_lists = false;
    uint32_t arena_cap = 0;         // dwords per slot
    uint64_t ptcl_want = 0;         // commands a slot's list arena starts with / was grown to
    bool arena_dirty = true;
    // The plan in force (strip-row work list, arena regions, band list) and the item boxes it was made for.  A view change
    // (pm_reflatten) plans with every box widened by a tile, so that the scenes that follow can keep the plan while their boxes
    // stay inside the widened ones and their items keep their segment counts: no host sizing, no uploads, no waits
    // (0.04 of an animated frame's 0.18 ms).
    std::vector<uint16_t> plan_box;   // [4 n] widened boxes
    std::vector<uint32_t> plan_per;   // [n] arena dwords per item
    bool plan_reusable = false;       // the plan was made with widened boxes, for the whole item list in scene order, without per-row lists
    bool replan_wide = false;         // the next plan widens the boxes (set by pm_reflatten)
    bool band_identity = false;       // the band's item list IS the scene's item list: the kernels read the scene's own boxes
    uint32_t arena_epoch = 0;       // bumped whenever the set of strip rows without a workgroup changes (EnsureArena)

    std::vector<FrameSlot> slot;
    uint32_t frame = 0;
    std::chrono::steady_clock::time_point last_submit{};  // when the previous frame was submitted, and what was found then:
    bool prev_running = false;                            // its predecessor was still running (hipStreamQuery) -- an answer kept for
    uint32_t query_keep = 0;                              // this many more frames submitted back to back (Enqueue)
    bool idle_seen = false;                               // the last answer was "idle", asked within a run of back-to-back frames
    uint32_t query_every = 8;                             // PM_QUERY_EVERY: ... asked again every this many frames
    int last_slot = -1;  // slot of the most recently submitted frame

    pm::FlattenCache flatten_cache;  // resident paths + scratch of the flatten stage

    // wall-clock cost of the last scene replacement, host view (pm_get_scene_timings)
    float t_flatten_ms = 0, t_index_ms = 0, t_arena_ms = 0;
    uint32_t plans_made = 0;

    // tables
    uint32_t *d_lut_srgb2lin = nullptr;
    uint32_t *d_lut_unorm2h = nullptr;
    uint8_t *d_lut_lin2srgb = nullptr;
};

namespace {

struct WallTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    float ms() const { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

uint32_t BandRows(const pm_ctx *c) { return c->row1 - c->row0; }
size_t BandTiles(const pm_ctx *c) { return std::max<size_t>(static_cast<size_t>(BandRows(c)) * c->tiles_x, 1); }

// Waits for everything this context submitted.  A caller-owned stream (pm_render_to) is never
// touched after submission -- the caller may have destroyed it: the slot's event, recorded behind
// the frame at submit time, is waited on instead.
int SyncAll(pm_ctx *c) {
    for (hipStream_t q : c->streams) PM_TRY(hipStreamSynchronize(q));
    for (auto &s : c->slot) {
        if (s.in_flight && s.user_stream) PM_TRY(hipEventSynchronize(s.ev_done));
        s.in_flight = false;
    }
    return PM_OK;
}

void FreeSlotViewport(FrameSlot *s) {
    if (s->d_fb) (void)hipFree(s->d_fb);
    if (s->d_queue) (void)hipFree(s->d_queue);  // (the three per-tile tables live behind the queues: one allocation)
    if (s->d_fifo) (void)hipFree(s->d_fifo);
    s->d_fifo = nullptr;
    s->fifo_cap = 0;
    s->d_fb = nullptr;
    s->d_queue = nullptr;
    s->fb_cap = s->tables_cap = 0;
    s->d_tile_state = s->d_tile_ptcl = s->d_tile_ncmd = nullptr;
    s->state_epoch = 0;
}

void FreeViewport(pm_ctx *c) {
    for (auto &s : c->slot) {
        FreeSlotViewport(&s);
        s.in_flight = false;
        s.needs_check = false;
    }
    c->last_slot = -1;
}

int AllocSlotViewport(pm_ctx *c, FrameSlot *s) {
    if (s->d_fb && s->vp_epoch == c->vp_epoch) return PM_OK;  // (all five buffers exist, or none: a partial set is released below)
    const size_t tiles = BandTiles(c);
    const size_t tables = std::max<size_t>(tiles, 4);
    const size_t fb_want = std::max<size_t>(c->fb_bytes, 16);
    hipError_t e = hipSuccess;
    if (s->d_fb && s->d_queue && s->fb_cap >= fb_want && s->tables_cap >= tables) {
        // the buffers of an earlier (larger or equal) viewport serve this one: nothing is released, nothing allocated -- a one-launch
        // frame's FIFOs excepted, whose capacity is the viewport's (they come back when they are next needed, EnsureFifo)
        if (s->d_fifo) (void)hipFree(s->d_fifo);
        s->d_fifo = nullptr;
        s->fifo_cap = 0;
    } else {
        FreeSlotViewport(s);  // (buffers of an earlier viewport: released now, when the slot is used again, not by the resize)
        // two allocations (each costs 50-300 us): the pixels, and -- behind one another -- the class queues (one per cost class)
        // and the three per-tile tables
        e = hipMalloc(&s->d_fb, fb_want);
        if (e == hipSuccess) e = hipMalloc(&s->d_queue, pm::kClasses * tables * sizeof(uint4) + 3 * tables * sizeof(uint32_t));
        if (e == hipSuccess) {
            s->fb_cap = fb_want;
            s->tables_cap = tables;
        }
    }
    s->vp_epoch = c->vp_epoch;
    if (e == hipSuccess) {
        s->d_tile_state = reinterpret_cast<uint32_t *>(s->d_queue + pm::kClasses * tables);
        s->d_tile_ptcl = s->d_tile_state + tables;
        s->d_tile_ncmd = s->d_tile_ptcl + tables;
    }
    s->state_epoch = 0;  // (never initialised)
    if (e != hipSuccess) {
        if (s->d_fb) (void)hipFree(s->d_fb);
        if (s->d_queue) (void)hipFree(s->d_queue);
        s->d_fb = nullptr;
        s->d_queue = nullptr;
        s->fb_cap = s->tables_cap = 0;
        s->d_tile_state = s->d_tile_ptcl = s->d_tile_ncmd = nullptr;
        return HipFail(e, "hipMalloc(frame slot viewport buffers)");
    }
    return PM_OK;
}

int AllocViewport(pm_ctx *c) {
    // (a resize releases and re-allocates frame slot 0's buffers only -- twenty hipFree of a large viewport are 1.5 ms --:
    //  the other slots' go when a frame next uses them, AllocSlotViewport)
    c->vp_epoch += 1;
    for (auto &s : c->slot) {
        s.in_flight = false;
        s.needs_check = false;
        if (s.h_overflow) s.h_overflow[2] = 0;  // (the tile kernel's dense / not dense verdict belongs to the old viewport)
    }
    c->last_slot = -1;
    const uint32_t rows = BandRows(c);
    c->fb_stride = static_cast<size_t>(c->width) * 4;
    c->fb_bytes = c->fb_stride * static_cast<size_t>(rows) * pm::kTileH;
    // (slot 0 now -- pm_framebuffer_device_ptr names its framebuffer before the first frame --, the
    //  others when they are first used: a resize followed by one frame pays for one set of buffers)
    const int r0 = AllocSlotViewport(c, &c->slot[0]);
    if (r0 != PM_OK) return r0;
    c->arena_dirty = true;
    c->plan_reusable = false;  // (another viewport or band: the plan goes with it)
    c->fb_slots.clear();       // (... and what its frames said about its strip rows)
    c->fb_applied = false;
    return PM_OK;
}

// Walks header + bboxes + items (host copy) and checks every offset the kernels
// will dereference.  The kernels trust the scene after this.
int ValidateScene(const uint8_t *meta, size_t meta_len, size_t scene_bytes, uint32_t *n_items_out) {
    if (scene_bytes < 8 || meta_len < 8) return PM_ERR_SCENE;
    uint32_t n, items_ix;
    std::memcpy(&n, meta, 4);
    std::memcpy(&items_ix, meta + 4, 4);
    const uint64_t bbox_end = 8ull + 8ull * n;
    const uint64_t items_end = static_cast<uint64_t>(items_ix) + 32ull * n;
    if (bbox_end > scene_bytes || items_end > scene_bytes || items_ix < bbox_end) return PM_ERR_SCENE;
    if (items_ix & 7u) return PM_ERR_SCENE;  // the encoder keeps everything 8-byte aligned (src/lib.rs:113-130)
    if (items_end > meta_len) return PM_ERR_SCENE;
    for (uint32_t i = 0; i < n; ++i) {
        const uint8_t *it = meta + items_ix + 32ull * i;
        uint32_t tag, npt, pix;
        std::memcpy(&tag, it, 4);
        tag &= 0xffffu;
        if (tag == pm::kItemFill || tag == pm::kItemPoly) {
            std::memcpy(&npt, it + 12, 4);
            std::memcpy(&pix, it + 16, 4);
            if ((pix & 7u) != 0 || static_cast<uint64_t>(pix) + 8ull * npt > scene_bytes) return PM_ERR_SCENE;
        }
    }
    *n_items_out = n;
    return PM_OK;
}

// Class thresholds in stream elements, descending: three classes of long lists (a workgroup per
// tile: > vheavy, > midway, > heavy), five of short ones.
void SetClassThresholds(const pm_ctx *c, pm::FrameParams *p, uint32_t heavy) {
    const uint32_t h = heavy, v = std::max(c->vheavy_stream, h);
    const uint32_t thr[pm::kClasses - 1] = {v, (v + h) / 2, h, h * 3 / 4, h / 2, h * 5 / 16, h * 5 / 32};
    for (uint32_t k = 0; k < pm::kClasses - 1; ++k) p->class_thr[k] = thr[k];
}

// A slot's binning arena and command-list arena, allocated (or grown to what EnsureArena asked for)
// when the slot is about to be used.
int EnsureSlotBuffers(pm_ctx *c, FrameSlot *s) {
    const int rv = AllocSlotViewport(c, s);
    if (rv != PM_OK) return rv;
    if (!s->d_arena || s->arena_cap < c->arena_cap) {
        if (s->d_arena) (void)hipFree(s->d_arena);  // (hipFree waits for the device: nothing is using it any more)
        s->d_arena = nullptr;
        s->arena_cap = 0;
        PM_TRY(hipMalloc(&s->d_arena, static_cast<size_t>(std::max<uint32_t>(c->arena_cap, pm::kArenaBase)) * sizeof(uint32_t)));
        s->arena_cap = c->arena_cap;
    }
    if (c->use_row_lists && (!s->d_row_bbox || s->row_cap < c->row_want)) {
        // per-tile-row item lists of this slot's frames (pm_rowcull_kernel writes them): boxes, then indices, ONE allocation
        if (s->d_row_bbox) (void)hipFree(s->d_row_bbox);
        s->d_row_bbox = nullptr;
        s->d_row_item = nullptr;
        s->row_cap = 0;
        const uint64_t cap = c->row_want + c->row_want / 4;
        PM_TRY(hipMalloc(&s->d_row_bbox, cap * (sizeof(uint2) + sizeof(uint32_t))));
        s->d_row_item = reinterpret_cast<uint32_t *>(s->d_row_bbox + cap);
        s->row_cap = cap;
    }
    if (!s->d_ptcl || s->ptcl_cap < c->ptcl_want) {
        if (s->d_ptcl) (void)hipFree(s->d_ptcl);
        s->d_ptcl = nullptr;
        s->ptcl_cap = 0;
        const uint64_t quads = std::max<uint64_t>(c->ptcl_want, 64);
        PM_TRY(hipMalloc(&s->d_ptcl, quads * sizeof(uint4)));
        s->ptcl_cap = static_cast<uint32_t>(quads);
    }
    return PM_OK;
}

// Worst-case binning-arena demand of every strip row of the band (dwords): every candidate item
// costs kChunkSegs segment slots (16 B) + meta words per chunk (every chunk surviving), plus a token
// amount that tells a strip row some item reaches from one nothing reaches.  The binning kernel
// bump-allocates inside these private regions, so the bound must be exact or larger.
// (need_half, cnt: optional -- the same bound for the two halves of every strip row [2 i], [2 i + 1] (tiles 0-7, 8-15), and the
//  candidates of every strip row: what EnsureArena cuts heavy strip rows in two with)
void StripRowBounds(pm_ctx *c, std::vector<uint64_t> *need, int margin, std::vector<uint64_t> *need_half = nullptr, std::vector<uint32_t> *cnt = nullptr) {
    const uint8_t *meta = c->item_meta.data();
    uint32_t n, items_ix;
    std::memcpy(&n, meta, 4);
    std::memcpy(&items_ix, meta + 4, 4);
    const uint32_t rows = BandRows(c);
    need->assign(static_cast<size_t>(rows) * c->strips_x, 0);
    const size_t w = static_cast<size_t>(c->strips_x) + 1;
    std::vector<uint64_t> &diff = c->stage_need_diff;
    diff.assign((static_cast<size_t>(rows) + 1) * w, 0);
    const size_t wh = 2 * static_cast<size_t>(c->strips_x) + 1;
    std::vector<uint64_t> diff_half;
    std::vector<int64_t> diff_cnt;
    if (need_half) diff_half.assign((static_cast<size_t>(rows) + 1) * wh, 0);
    if (cnt) diff_cnt.assign((static_cast<size_t>(rows) + 1) * w, 0);
    c->plan_box.resize(4ull * n);
    c->plan_per.resize(n);
    c->plan_cands = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint16_t bb[4];
        std::memcpy(bb, meta + 8 + 8ull * i, 8);
        // (the box the plan is made for: the item's own, or widened by `margin` pixels on every side)
        bb[0] = static_cast<uint16_t>(std::max(0, static_cast<int>(bb[0]) - margin));
        bb[1] = static_cast<uint16_t>(std::max(0, static_cast<int>(bb[1]) - margin));
        bb[2] = static_cast<uint16_t>(std::min(65535, static_cast<int>(bb[2]) + margin));
        bb[3] = static_cast<uint16_t>(std::min(65535, static_cast<int>(bb[3]) + margin));
        std::memcpy(&c->plan_box[4ull * i], bb, 8);
        const uint8_t *it = meta + items_ix + 32ull * i;
        uint32_t tag, npt = 0;
        std::memcpy(&tag, it, 4);
        tag &= 0xffffu;
        std::memcpy(&npt, it + 12, 4);
        uint64_t nseg = 0;
        if (tag == pm::kItemFill) nseg = npt;
        if (tag == pm::kItemPoly && npt >= 2) nseg = npt - 1;
        if (tag == pm::kItemLine) nseg = 1;
        if (margin > 0 && tag != pm::kItemLine) nseg += nseg / 4 + 2 * pm::kChunkSegs;  // (a zooming view: curves flatten into more segments as they grow)
        const uint64_t nch = (nseg + pm::kChunkSegs - 1) / pm::kChunkSegs;
        const uint64_t per = 4u + static_cast<uint64_t>(pm::kSlotDwords) * pm::kChunkSegs * nch;
        c->plan_per[i] = static_cast<uint32_t>(std::min<uint64_t>(per, 0xffffffffull));
        // strips: bz >= sx0 && bx < sx0 + 256 ; rows: bw >= y0 && by < y0 + 16
        const int64_t s_lo = bb[0] / 256, s_hi = std::min<int64_t>(bb[2] / 256, static_cast<int64_t>(c->strips_x) - 1);
        const int64_t r_lo = std::max<int64_t>(bb[1] / 16, c->row0), r_hi = std::min<int64_t>(bb[3] / 16, static_cast<int64_t>(c->row1) - 1);
        if (r_lo > r_hi || s_lo > s_hi) continue;
        c->plan_cands += static_cast<uint64_t>(r_hi - r_lo + 1) * static_cast<uint64_t>(s_hi - s_lo + 1);
        // (a 2-D difference array: four updates per item instead of one per strip row it covers -- config 5's
        //  7 600 items cover millions of them: arena sizing 0.26 -> 0.10 ms)
        const size_t a = static_cast<size_t>(r_lo - c->row0), b = static_cast<size_t>(r_hi - c->row0) + 1;
        diff[a * w + static_cast<size_t>(s_lo)] += per;
        diff[a * w + static_cast<size_t>(s_hi) + 1] -= per;
        diff[b * w + static_cast<size_t>(s_lo)] -= per;
        diff[b * w + static_cast<size_t>(s_hi) + 1] += per;
        if (cnt) {
            diff_cnt[a * w + static_cast<size_t>(s_lo)] += 1;
            diff_cnt[a * w + static_cast<size_t>(s_hi) + 1] -= 1;
            diff_cnt[b * w + static_cast<size_t>(s_lo)] -= 1;
            diff_cnt[b * w + static_cast<size_t>(s_hi) + 1] += 1;
        }
        if (need_half) {  // half strips: bz >= hx0 && bx < hx0 + 128
            const size_t h_lo = bb[0] / 128, h_hi = std::min<size_t>(bb[2] / 128, 2 * static_cast<size_t>(c->strips_x) - 1);
            diff_half[a * wh + h_lo] += per;
            diff_half[a * wh + h_hi + 1] -= per;
            diff_half[b * wh + h_lo] -= per;
            diff_half[b * wh + h_hi + 1] += per;
        }
    }
    if (cnt) {
        cnt->assign(static_cast<size_t>(rows) * c->strips_x, 0);
        for (size_t r = 0; r < rows; ++r) {
            int64_t run = 0;
            for (size_t sx = 0; sx < c->strips_x; ++sx) {
                run += diff_cnt[r * w + sx];
                (*cnt)[r * c->strips_x + sx] = static_cast<uint32_t>(run + (r ? static_cast<int64_t>((*cnt)[(r - 1) * c->strips_x + sx]) : 0));
            }
        }
    }
    if (need_half) {
        const size_t hs = 2 * static_cast<size_t>(c->strips_x);
        need_half->assign(static_cast<size_t>(rows) * hs, 0);
        for (size_t r = 0; r < rows; ++r) {
            uint64_t run = 0;
            for (size_t hx = 0; hx < hs; ++hx) {
                run += diff_half[r * wh + hx];
                (*need_half)[r * hs + hx] = run + (r ? (*need_half)[(r - 1) * hs + hx] : 0);
            }
        }
    }
    for (size_t r = 0; r < rows; ++r) {  // prefix sums along the strips, then down the rows (mod 2^64: the totals are exact)
        uint64_t run = 0;
        for (size_t sx = 0; sx < c->strips_x; ++sx) {
            run += diff[r * w + sx];
            (*need)[r * c->strips_x + sx] = run + (r ? (*need)[(r - 1) * c->strips_x + sx] : 0);
        }
    }
}

int EnsureArena(pm_ctx *c) {
    if (!c->arena_dirty && c->arena_cap != 0) return PM_OK;
    if (c->item_meta.empty() || c->tiles_x == 0) return PM_OK;  // nothing to size against yet
    const WallTimer timer;
    {
        // Does the plan in force still hold?  Same items in the same order, every item with the arena need it was planned with
        // and its box inside the (widened) box it was planned with.
        const uint8_t *meta = c->item_meta.data();
        uint32_t n, items_ix;
        std::memcpy(&n, meta, 4);
        std::memcpy(&items_ix, meta + 4, 4);
        bool ok = c->plan_reusable && c->arena_cap != 0 && c->band_identity && n == c->plan_per.size() && n == c->n_band_items;
        for (uint32_t i = 0; i < n && ok; ++i) {
            uint16_t bb[4];
            std::memcpy(bb, meta + 8 + 8ull * i, 8);
            const uint16_t *pb = &c->plan_box[4ull * i];
            const uint8_t *it = meta + items_ix + 32ull * i;
            uint32_t tag, npt = 0;
            std::memcpy(&tag, it, 4);
            tag &= 0xffffu;
            std::memcpy(&npt, it + 12, 4);
            uint64_t nseg = 0;
            if (tag == pm::kItemFill) nseg = npt;
            if (tag == pm::kItemPoly && npt >= 2) nseg = npt - 1;
            if (tag == pm::kItemLine) nseg = 1;
            const uint64_t per = 4u + static_cast<uint64_t>(pm::kSlotDwords) * pm::kChunkSegs * ((nseg + pm::kChunkSegs - 1) / pm::kChunkSegs);
            ok = per <= c->plan_per[i] && bb[0] >= pb[0] && bb[1] >= pb[1] && bb[2] <= pb[2] && bb[3] <= pb[3] && bb[2] >= bb[0] && bb[3] >= bb[1];
        }
        if (ok) {
            // (the scene index of the new scene is still being built on the context's stream; frames run on the others)
            PM_TRY(hipStreamSynchronize(c->stream));
            c->arena_dirty = false;
            c->t_arena_ms = timer.ms();
            return PM_OK;
        }
    }
    // (host work first: the scene-index kernel of a scene replacement is still running on the device)
    const int margin = c->replan_wide ? static_cast<int>(pm::kTileW) : 0;
    std::vector<uint64_t> need, need_half;
    std::vector<uint32_t> row_cands;
    bool may_split = c->bin_split_mode != 0 && c->one_launch_mode == 0;  // (scenes with per-tile-row item lists: decided below, once the band's list is made)
    StripRowBounds(c, &need, margin);
    if (may_split && c->bin_split_mode == 1) {  // (cuts only while the plan's rows leave room in the resident grid: large frames pay for no second sizing pass)
        size_t n_rows = 0;
        for (size_t i = 0; i < need.size(); ++i) n_rows += need[i] != 0 ? 1u : 0u;
        may_split = n_rows < static_cast<size_t>(c->n_cus) * 5u;
    }
    if (may_split) StripRowBounds(c, &need, margin, &need_half, &row_cands);
    // (host work before any upload: the band's item list, the arena regions)
    c->sr_empty_dwords = 0;  // (a strip row no item's bbox reaches has nothing reserved)
    // the items whose bbox reaches the band (rows: bw >= y0 && by < y1, PietRender.metal:198/:214), paint order
    std::vector<uint2> &bbs = c->stage_bbs;
    std::vector<uint32_t> &ids = c->stage_ids;
    {
        const uint8_t *meta = c->item_meta.data();
        bbs.clear();
        ids.clear();
        const uint32_t y0 = c->row0 * pm::kTileH, y1 = c->row1 * pm::kTileH;
        for (uint32_t i = 0; i < c->n_items; ++i) {
            uint32_t w[2];
            std::memcpy(w, meta + 8 + 8ull * i, 8);
            const uint16_t *pb = &c->plan_box[4ull * i];  // (the box the plan is made for: widened for a view change)
            const uint32_t bx = pb[0], by = pb[1], bw = pb[3];
            if (bw >= y0 && by < y1 && bx < c->strips_x * pm::kGroupW) {
                bbs.push_back(make_uint2(w[0], w[1]));
                ids.push_back(i);
            }
        }
        c->n_band_items = static_cast<uint32_t>(ids.size());
        // every item of the scene, in scene order: the kernels can read the scene's own boxes (nothing to upload, and the
        // list stays right for the next scene with the same items)
        const int min_items = EnvInt("PM_ROW_LIST_MIN_ITEMS", 2048, 0, 1 << 30);
        c->use_row_lists = static_cast<int>(ids.size()) >= min_items && !ids.empty();
        c->band_identity = ids.size() == c->n_items && !c->use_row_lists;
    }
    // The strip rows some item's bbox reaches get a workgroup of pm_bin_kernel each; the others are
    // background for as long as this scene and viewport last: their tile_state is set to white
    // once, here.  (An empty list still launches one workgroup: it resets the frame counters.)
    std::vector<uint4> &desc = c->stage_desc;  // (sources of asynchronous uploads live in the context)
    desc.clear();
    uint32_t per_cu = c->bin_wg_per_cu;
    if (per_cu == 0xffu) per_cu = 5u;
    // Which strip rows are cut in two (bin_split_mode): the heaviest ones, as many as the resident grid has workgroups to spare.
    std::vector<uint8_t> cut(need.size(), 0);
    c->n_sr_split = 0;
    if (may_split && !c->use_row_lists) {  // (per-tile-row item lists are addressed by entry of the one work list)
        size_t n_rows = 0;
        for (size_t i = 0; i < need.size(); ++i) n_rows += ((need[i] + 3u) & ~3ull) != c->sr_empty_dwords ? 1u : 0u;
        // (not to the last workgroup the chip holds: with all 1 280 places taken -- 171 rows cut at the 4K Tiger -- binning was 1.7 us
        //  SLOWER than with 1 214; the dispatcher's placement is not perfectly even, and a workgroup that has to wait for a place
        //  starts when a first one ends)
        const size_t resident = static_cast<size_t>(c->n_cus) * (per_cu ? per_cu : 5u) * static_cast<size_t>(c->bin_split_fill) / 16u;
        const size_t room = c->bin_split_mode == 2 ? need.size() : (resident > n_rows ? resident - n_rows : 0u);
        std::vector<std::pair<uint32_t, uint32_t>> heavy;  // {weight, strip row}
        const bool fed = c->fb_slots.size() == need.size();  // (frames of this scene and viewport have reported)
        for (size_t i = 0; i < need.size() && room != 0; ++i) {
            if (((need[i] + 3u) & ~3ull) == c->sr_empty_dwords) continue;
            if (c->bin_split_mode == 2) heavy.emplace_back(row_cands[i], static_cast<uint32_t>(i));
            else if (fed ? c->fb_slots[i] >= c->bin_split_slots : row_cands[i] >= c->bin_split_cands) heavy.emplace_back(fed ? c->fb_slots[i] : row_cands[i], static_cast<uint32_t>(i));
        }
        c->fb_applied = fed;
        if (heavy.size() > room) {
            std::partial_sort(heavy.begin(), heavy.begin() + static_cast<ptrdiff_t>(room), heavy.end(),
                              [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
            heavy.resize(room);
        }
        for (const auto &h : heavy) cut[h.second] = 1;
    }
    uint64_t total = pm::kArenaBase;  // offset 0 means "no record"
    constexpr uint32_t kWholeStrip = 15u << 12, kLeftHalf = 7u << 12, kRightHalf = (7u << 12) | (8u << 8);  // tiles - 1 << 12 | first tile << 8
    auto push = [&](size_t i, uint32_t run, uint64_t need_dwords) -> bool {
        const uint64_t sz = (need_dwords + 3u) & ~3ull;
        if (sz == c->sr_empty_dwords) return true;
        const uint64_t begin = total;
        total += sz;
        if (total > 0xfffffff0ull) return false;
        desc.push_back(make_uint4(static_cast<uint32_t>(i % c->strips_x) | run | (static_cast<uint32_t>(i / c->strips_x) << 16), static_cast<uint32_t>(begin),
                                  static_cast<uint32_t>(total), 0u));
        return true;
    };
    for (size_t i = 0; i < need.size(); ++i) {
        bool ok = true;
        const size_t h = (i / c->strips_x) * 2 * c->strips_x + 2 * (i % c->strips_x);
        // (a half no item reaches gets no entry -- and nobody would write its pixels: such a row stays whole)
        if (cut[i] && ((need_half[h] + 3u) & ~3ull) != c->sr_empty_dwords && ((need_half[h + 1] + 3u) & ~3ull) != c->sr_empty_dwords) {
            ok = push(i, kLeftHalf, need_half[h]) && push(i, kRightHalf, need_half[h + 1]);
            c->n_sr_split += 1;
        } else {
            ok = push(i, kWholeStrip, need[i]);
        }
        if (!ok) {
            SetError("scene x viewport needs a binning arena beyond 16 GiB");
            return PM_ERR_CAPACITY;
        }
    }
    if (desc.empty()) desc.push_back(make_uint4(kWholeStrip, pm::kArenaBase, pm::kArenaBase, 0u));
    // (exact for a context's first scene; a quarter of headroom when it has to GROW: an animation's
    //  demand creeps from frame to frame, and re-allocating four 100 MB arenas costs milliseconds)
    const uint64_t alloc_dwords = total <= c->arena_cap ? c->arena_cap : (c->arena_cap == 0 ? total : std::min<uint64_t>(0xfffffff0ull, total + total / 4));
    // (the slots' arenas themselves are allocated when a slot is first used, EnsureSlotBuffers: the
    //  first frame of a scene pays for one arena, not for four -- hundreds of MB each at 8K)
    {
        // Tile arena (per-tile pieces + command lists, 16-byte quads): sized from what binning
        // actually finds, so there is no static bound; start generously (HBM is 288 GB), in proportion
        // to EVERY scene that comes (a max: what pm_sync grew it to is kept), and let pm_sync grow it on
        // overflow.
        uint64_t cmds = std::max<uint64_t>(1u << 22, 64ull * c->n_chunks * pm::kChunkSegs);
        if (const char *v = std::getenv("PM_PTCL_INITIAL_CMDS"))  // tests: force the overflow -> grow -> re-render path
            cmds = std::max<uint64_t>(64, std::strtoull(v, nullptr, 10));
        // (a frame behind running frames may bin with a wave per strip row, Enqueue: records of 64 candidates instead of 256,
        //  up to four pieces -- four headers, four times the slack behind the candidates -- where a lone frame leaves one.  The
        //  arena a plan starts with holds that variant too: a frame that fits alone must not overflow in flight.  Round-4 advisor.)
        if (c->bin_waves_inflight == 1 && !std::getenv("PM_PTCL_INITIAL_CMDS"))
            cmds = std::max<uint64_t>(cmds, 80ull * c->n_chunks * pm::kChunkSegs + 4ull * BandTiles(c));
        c->ptcl_want = std::max<uint64_t>(c->ptcl_want, std::min<uint64_t>(cmds * pm::kCmdQuadsNum / pm::kCmdQuadsDen, 0x7fffffffull));
    }
    c->arena_cap = std::max<uint32_t>(c->arena_cap, static_cast<uint32_t>(alloc_dwords));
    // (strip rows stay in their natural order: heaviest-first was measured 2.5 us slower -- the heavy
    //  ones then share CUs -- and so was a snake order over CU periods; neighbouring strip rows share
    //  data and belong together)
    c->n_sr_active = static_cast<uint32_t>(desc.size());
    {
        // pm_bin_kernel's grid is no larger than what the chip holds at once: five workgroups per CU (its LDS
        // is sized for that).  A group walks a chain of strip rows (desc.w = index of the next one, 0 = none):
        // row b, b + grid, b + 2 grid ... in natural order -- a grid larger than the residency would start its
        // last groups when the first END their chains.  Measured and not kept: any permutation of ALL rows, +25 %
        // (neighbouring strip rows share data); pairing the lightest rows by their arena need, +9 % (the need is
        // the worst case of the chunk test, not the work); rows DRAWN from counters instead of dealt (a returning
        // atomic per row, sent off a row ahead): config 5 0.229 -> 0.211 ms with a workgroup per row but config 4
        // 0.140 -> 0.150, and 0.195 -> 0.207 with a wave per row.
        const size_t n = desc.size();
        // A wave per strip row instead of a workgroup when there are several times more strip rows than the chip holds
        // workgroups AND the rows are light (a record of 64 candidates holds nearly every row's): sixteen one-wave
        // groups share a CU, no workgroup barriers, a quarter fewer vector and a third fewer scalar instructions per
        // strip row -- a row takes twice as long, three times as many are in flight.  Config 5 (16 160 rows of 11
        // candidates): binning 0.229 -> 0.195 ms, sustained +7 %; config 4 (4 096 rows of 108): 0.140 -> 0.162, stays
        // with workgroups; so does any frame whose rows all fit the chip at once (the 4K Tiger: 25 -> 52 us).
        const bool many_light_rows = n >= static_cast<size_t>(c->n_cus) * 5u * 3u && c->plan_cands <= 32ull * n;
        c->bin_waves = c->bin_waves_env == 1 || c->bin_waves_env == 4 ? c->bin_waves_env : (many_light_rows ? 1u : 4u);
        if (c->bin_waves == 1 && c->bin_wg_per_cu == 0xffu) per_cu = 16u;
        const size_t grid = per_cu == 0 ? n : std::min<size_t>(n, static_cast<size_t>(c->n_cus) * per_cu);
        if (n > grid) {
            for (size_t i = 0; i + grid < n; ++i) desc[i].w = static_cast<uint32_t>(i + grid);
        }
        c->bin_grid = static_cast<uint32_t>(grid);
    }
    PM_TRY(SyncAll(c) == PM_OK ? hipSuccess : hipErrorUnknown);  // frames in flight still read the lists replaced below
    if (desc.size() > c->sr_desc_cap || !c->d_sr_list || !c->d_sr_slots || !c->d_sr_desc_whole) {  // (grow only: an animation re-sizes every frame)
        if (c->d_sr_desc) (void)hipFree(c->d_sr_desc);
        if (c->d_sr_list) (void)hipFree(c->d_sr_list);
        c->d_sr_desc = nullptr;
        c->d_sr_list = nullptr;
        c->sr_desc_cap = 0;
        const size_t want = std::max<size_t>(desc.size() + desc.size() / 4 + 16, c->sr_desc_cap);
        PM_TRY(hipMalloc(&c->d_sr_desc, want * sizeof(uint4)));
        PM_TRY(hipMalloc(&c->d_sr_list, want * sizeof(uint2)));
        if (c->d_sr_slots) (void)hipFree(c->d_sr_slots);
        c->d_sr_slots = nullptr;
        PM_TRY(hipMalloc(&c->d_sr_slots, want * sizeof(uint32_t)));
        if (c->d_sr_desc_whole) (void)hipFree(c->d_sr_desc_whole);
        c->d_sr_desc_whole = nullptr;
        PM_TRY(hipMalloc(&c->d_sr_desc_whole, want * sizeof(uint4)));
        c->sr_desc_cap = want;
    }
    {
        // the work list without the cuts: the two halves of a cut strip row lie next to each other, in the list and in the arena
        std::vector<uint4> &whole = c->stage_desc_whole;
        whole.clear();
        for (size_t k = 0; k < desc.size() && c->n_sr_split != 0; ++k) {
            const uint32_t key = desc[k].x & 0xffff00ffu;
            if (!whole.empty() && (whole.back().x & 0xffff00ffu) == key) whole.back().z = desc[k].z;
            else whole.push_back(make_uint4(key | (15u << 12), desc[k].y, desc[k].z, 0u));
        }
        c->n_sr_whole = static_cast<uint32_t>(whole.size());
        if (!whole.empty()) PM_TRY(hipMemcpyAsync(c->d_sr_desc_whole, whole.data(), whole.size() * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
    }
    PM_TRY(hipMemsetAsync(c->d_sr_slots, 0, desc.size() * sizeof(uint32_t), c->stream));
    c->frames_on_plan = 0;
    PM_TRY(hipMemcpyAsync(c->d_sr_desc, desc.data(), desc.size() * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
    {
        // Chains of the one-launch grid (four workgroups per CU, all resident): workgroup b bins strip row b; with more strip rows
        // than workgroups the rows beyond go, heaviest first, to the workgroups whose own row is lightest (by the arena bound,
        // the only estimate there is) -- a second light row ends before the frame's heaviest row does.
        std::vector<uint32_t> &nx = c->stage_next_one;
        const size_t n = desc.size();
        nx.assign(n, 0u);
        const size_t resident = static_cast<size_t>(c->n_cus) * c->frame_wg_per_cu;
        c->one_grid_rows = 0;
        if (c->one_launch_mode != 0 && resident != 0 && n <= 2 * resident) {  // (only a context that renders one-launch frames pays for their lists)
            const size_t rows = std::min(n, resident);
            c->one_grid_rows = static_cast<uint32_t>(rows);
            if (n > rows) {
                std::vector<uint32_t> first(rows), extra(n - rows);
                for (size_t i = 0; i < rows; ++i) first[i] = static_cast<uint32_t>(i);
                for (size_t i = rows; i < n; ++i) extra[i - rows] = static_cast<uint32_t>(i);
                auto weight = [&](uint32_t i) { return desc[i].z - desc[i].y; };
                std::stable_sort(first.begin(), first.end(), [&](uint32_t a, uint32_t b) { return weight(a) < weight(b); });
                std::stable_sort(extra.begin(), extra.end(), [&](uint32_t a, uint32_t b) { return weight(a) > weight(b); });
                for (size_t k = 0; k < extra.size(); ++k) nx[first[k]] = extra[k];
            }
        }
        if (c->one_launch_mode != 0 && (n > c->sr_next_one_cap || !c->d_sr_next_one)) {  // (only a context that renders one-launch frames pays for the table)
            if (c->d_sr_next_one) (void)hipFree(c->d_sr_next_one);
            c->d_sr_next_one = nullptr;
            c->sr_next_one_cap = 0;
            const size_t want = n + n / 4 + 16;
            PM_TRY(hipMalloc(&c->d_sr_next_one, want * sizeof(uint32_t)));
            c->sr_next_one_cap = want;
        }
        if (c->one_launch_mode != 0) PM_TRY(hipMemcpyAsync(c->d_sr_next_one, nx.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    }
    {
        // the strip rows without a workgroup: the binning launch (clear_in_bin) or a one-launch frame writes their (background) pixels from this list
        std::vector<uint32_t> &idle = c->stage_idle;
        idle.clear();
        std::vector<uint8_t> listed(need.size(), 0);  // (the empty list's one workgroup "has" strip row 0; a strip row cut in two is listed twice)
        for (const uint4 &d : desc) {
            const size_t i = static_cast<size_t>(d.x >> 16) * c->strips_x + (d.x & 0xffu);
            if (i < listed.size()) listed[i] = 1;
        }
        for (size_t i = 0; i < need.size() && (c->one_grid_rows != 0 || c->fold_clear_mode >= 3); ++i)
            if (!listed[i]) idle.push_back(static_cast<uint32_t>(i));
        c->n_idle_sr = static_cast<uint32_t>(idle.size());
        if (idle.size() > c->idle_sr_cap || !c->d_idle_sr) {
            if (c->d_idle_sr) (void)hipFree(c->d_idle_sr);
            c->d_idle_sr = nullptr;
            c->idle_sr_cap = 0;
            const size_t want = idle.size() + idle.size() / 4 + 16;
            PM_TRY(hipMalloc(&c->d_idle_sr, want * sizeof(uint32_t)));
            c->idle_sr_cap = want;
        }
        if (!idle.empty()) PM_TRY(hipMemcpyAsync(c->d_idle_sr, idle.data(), idle.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    }
    c->arena_epoch += 1;  // (a slot's tile_state is reset to "background" on the frame's own stream when the slot is next used)
    {
        if (ids.size() > c->band_cap || !c->d_band_bbox) {
            if (c->d_band_bbox) (void)hipFree(c->d_band_bbox);
            if (c->d_band_item) (void)hipFree(c->d_band_item);
            c->d_band_bbox = nullptr;
            c->d_band_item = nullptr;
            c->band_cap = 0;
            const size_t want = ids.size() + ids.size() / 4 + 16;
            PM_TRY(hipMalloc(&c->d_band_bbox, want * sizeof(uint2)));
            PM_TRY(hipMalloc(&c->d_band_item, want * sizeof(uint32_t)));
            c->band_cap = want;
        }
        if (!ids.empty() && !c->band_identity) {
            PM_TRY(hipMemcpyAsync(c->d_band_bbox, bbs.data(), ids.size() * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
            PM_TRY(hipMemcpyAsync(c->d_band_item, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        }
        // Large scenes: every tile row gets its own item list each frame (pm_rowcull_kernel); the
        // host only sizes the lists, with the kernel's predicate.
        if (c->use_row_lists) {
            const uint32_t rows = BandRows(c);
            // A tile row's scan is cut into parts (workgroups of pm_rowcull_kernel), about four workgroups per CU in all, a part no
            // shorter than one step of the kernel: rb[r * parts + p] = where part p of row r writes.
            const uint32_t n_it = static_cast<uint32_t>(bbs.size());
            const uint32_t parts_wanted = std::max<uint32_t>(1u, (4u * static_cast<uint32_t>(c->n_cus)) / std::max<uint32_t>(rows, 1u));
            const uint64_t per_part = static_cast<uint64_t>(pm::kRowCullStep) * parts_wanted;
            // (PM_ROW_LIST_PART_ITEMS: the tests cut small scenes into many parts)
            const uint32_t part_items = static_cast<uint32_t>(EnvInt("PM_ROW_LIST_PART_ITEMS", static_cast<int>(pm::kRowCullStep * static_cast<uint32_t>(std::max<uint64_t>(1u, (n_it + per_part - 1u) / per_part))), 1, 1 << 30));
            const uint32_t parts = std::max<uint32_t>(1u, (n_it + part_items - 1u) / part_items);
            c->row_parts = parts;
            c->row_part_items = part_items;
            std::vector<uint32_t> &rb = c->stage_rb;
            rb.assign(static_cast<size_t>(rows) * parts + 1, 0u);
            for (uint32_t k = 0; k < n_it; ++k) {
                const uint2 &b = bbs[k];
                const uint32_t by = b.x >> 16, bw = b.y >> 16;
                const uint32_t r_lo = std::max(by / pm::kTileH, c->row0), r_hi = std::min(bw / pm::kTileH, c->row1 - 1);
                const uint32_t pk = k / part_items;
                for (uint32_t r = r_lo; r <= r_hi && r_hi >= r_lo; ++r) rb[static_cast<size_t>(r - c->row0) * parts + pk + 1] += 1;
            }
            uint64_t run = 0;
            for (size_t i = 0; i < static_cast<size_t>(rows) * parts; ++i) {
                const uint32_t n_i = rb[i + 1];
                rb[i] = static_cast<uint32_t>(run);
                run += n_i;
            }
            if (run > 0xfffffff0ull) {
                SetError("per-row item lists beyond 2^32 entries");
                return PM_ERR_CAPACITY;
            }
            rb[static_cast<size_t>(rows) * parts] = static_cast<uint32_t>(run);
            c->row_total = static_cast<uint32_t>(run);
            if (rb.size() > c->row_base_cap || !c->d_row_base) {  // (grow only)
                if (c->d_row_base) (void)hipFree(c->d_row_base);
                c->d_row_base = nullptr;
                c->row_base_cap = 0;
                PM_TRY(hipMalloc(&c->d_row_base, (rb.size() + 64) * sizeof(uint32_t)));
                c->row_base_cap = rb.size() + 64;
            }
            PM_TRY(hipMemcpyAsync(c->d_row_base, rb.data(), rb.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            // every strip row's descriptor gets its tile row's {first entry, entries} next to it
            std::vector<uint2> &sl = c->stage_list;
            sl.resize(c->stage_desc.size());
            for (size_t i = 0; i < sl.size(); ++i) {
                const uint32_t r = c->stage_desc[i].x >> 16;
                sl[i] = r < rows ? make_uint2(rb[static_cast<size_t>(r) * parts], rb[static_cast<size_t>(r + 1) * parts] - rb[static_cast<size_t>(r) * parts]) : make_uint2(0u, 0u);
            }
            PM_TRY(hipMemcpyAsync(c->d_sr_list, sl.data(), sl.size() * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
            // (the slots' own list buffers come into being when a slot is next used, EnsureSlotBuffers: the first frame of a
            //  scene pays for one slot's, not for four -- eight allocations, 0.4 ms of config 5's first frame)
            c->row_want = std::max<uint64_t>(run, 1);
        }
    }
    PM_TRY(hipStreamSynchronize(c->stream));  // ONE wait: the lists are in place before a frame on any stream reads them
    c->plan_reusable = margin > 0 && c->band_identity;
    c->plans_made += 1;
    c->arena_dirty = false;
    c->t_arena_ms = timer.ms();
    return PM_OK;
}

uint32_t FineGrid(const pm_ctx *c);

int BuildParams(pm_ctx *c, FrameSlot *s, uint8_t *fb, size_t stride, pm::FrameParams *p) {
    if (!c->d_scene || c->scene_bytes < 8) {
        SetError("no scene resident (pm_upload_scene / pm_flatten_and_encode first)");
        return PM_ERR_INVALID;
    }
    if (c->tiles_x == 0 || BandRows(c) == 0) {
        SetError("no viewport (pm_resize first)");
        return PM_ERR_INVALID;
    }
    const WallTimer tm;
    int r = EnsureArena(c);
    if (r != PM_OK) return r;
    const float t_plan = tm.ms();
    r = EnsureSlotBuffers(c, s);  // (the slot's buffers come into being when it is first used)
    if (r != PM_OK) return r;
    if (tm.ms() > 0.05f && std::getenv("PM_HOST_TIMING")) std::fprintf(stderr, "BuildParams: plan %.3f slot buffers %.3f ms (arena %u dwords, tile arena %llu quads)\n", t_plan, tm.ms() - t_plan, c->arena_cap, static_cast<unsigned long long>(c->ptcl_want));
    if (!fb) fb = s->d_fb;
    std::memset(p, 0, sizeof(*p));
    p->scene = c->d_scene;
    p->scene_bytes = static_cast<uint32_t>(c->scene_bytes);
    p->n_items = c->n_items;
    p->items_ix = c->dev_items_ix;
    p->bbox_ix = c->dev_bbox_ix;
    p->width = c->width;
    p->height = c->height;
    p->tiles_x = c->tiles_x;
    p->tiles_y = c->tiles_y;
    p->row0 = c->row0;
    p->row1 = c->row1;
    p->strips_x = c->strips_x;
    p->fb = fb;
    p->fb_stride = static_cast<uint32_t>(stride);
    p->fb_vec16 = ((reinterpret_cast<uintptr_t>(fb) & 15u) == 0 && (stride & 15u) == 0) ? 1u : 0u;
    p->fb_bgra = c->target_fmt == PM_FMT_BGRA8 ? 1u : 0u;
    p->arena = s->d_arena;
    p->arena_cap = s->arena_cap;
    p->sr_desc = c->d_sr_desc;
    p->sr_list = c->d_sr_list;
    p->n_sr_active = c->n_sr_active;
    p->sr_slots = c->d_sr_slots;
    p->bin_prio_slots = c->bin_prio_slots;
    p->bin_grid = c->bin_grid;
    p->sr_empty_dwords = c->sr_empty_dwords;
    p->queue = s->d_queue;
    p->queue_cap = static_cast<uint32_t>(BandTiles(c));
    p->tile_state = s->d_tile_state;
    p->tarena = s->d_ptcl;
    p->tarena_cap = s->ptcl_cap;
    p->tile_ptcl = s->d_tile_ptcl;
    p->tile_ncmd = s->d_tile_ncmd;
    p->ctr_cur = s->d_ctr + s->parity;
    p->ctr_next = s->d_ctr + (s->parity ^ 1u);
    p->host_overflow = s->d_overflow;
    p->host_fail = s->d_overflow + 1;
    p->host_dense = s->d_overflow + 2;
    p->fine_dense = 0;
    p->fifo = s->d_fifo;
    p->fifo_cap = s->fifo_cap;
    p->one_launch = 0;
    p->sr_next_one = c->d_sr_next_one;
    p->one_grid_rows = c->one_grid_rows;
    p->idle_sr = c->d_idle_sr;
    p->n_idle_sr = c->n_idle_sr;
    p->spin_ticks = c->frame_spin_ticks;
    // (the whole item list in scene order: the scene's own boxes, no list of indices)
    p->band_bbox = c->band_identity ? reinterpret_cast<const uint2 *>(c->d_scene + c->dev_bbox_ix) : c->d_band_bbox;
    p->band_item = c->band_identity ? nullptr : c->d_band_item;
    p->n_band_items = c->n_band_items;
    p->bin_waves = c->bin_waves;
    // (PM_FOLD_CLEAR=3: a frame whose strip rows are chained -- config 5: lone frame 0.482 -> 0.456 ms -- and, Enqueue, any frame behind
    //  running frames: 4K Tiger sustained +1.5 %.  A small frame alone keeps the fold into the tile kernel's launch: with the
    //  clearing its binning kernel ends a microsecond later, and that is on the frame's critical path.  PM_FOLD_CLEAR=4: always.)
    p->clear_in_bin = c->fold_clear_mode == 4 || (c->fold_clear_mode == 3 && c->n_sr_active > c->bin_grid) ? 1u : 0u;
    p->bin_wt = c->bin_wt_mode == 1 || (c->bin_wt_mode == 2 && c->n_sr_active <= c->bin_grid) ? 1u : 0u;
    p->split_mode = c->split_mode;
    p->dense_factor = c->dense_factor;
    p->verdict_waves = static_cast<uint32_t>(c->n_cus) * c->fine_wg_per_cu * 4u;
    {
        SetClassThresholds(c, p, c->heavy_stream_lone);
        p->n_heavy_classes = 3;
        p->handout_static = c->handout == 1 ? 1u : 0u;  // (Enqueue decides per frame when PM_HANDOUT is 0)
    }
    p->fine_grid = FineGrid(c);
    p->use_row_lists = c->use_row_lists ? 1u : 0u;
    p->row_base = c->d_row_base;
    p->row_parts = c->row_parts;
    p->row_part_items = c->row_part_items;
    p->row_bbox = s->d_row_bbox;
    p->row_item = s->d_row_item;
    p->chunk_base = c->d_chunk_base;
    p->chunk_bbox = c->d_chunk_bbox;
    p->sup_bbox = c->d_sup_bbox;
    p->lut_srgb2lin = c->d_lut_srgb2lin;
    p->lut_unorm2h = c->d_lut_unorm2h;
    p->lut_lin2srgb = c->d_lut_lin2srgb;
    return PM_OK;
}

// Persistent grids: workgroups of 4 waves, one wave per tile (or per part of a tile).
// (per CU: 6 workgroups for pm_coarse_kernel, 4 for pm_fine_kernel by default -- with four frames in
//  flight anything from 2 to 8 measures within 2 %)

uint32_t CoarseGrid(const pm_ctx *c) {
    const uint32_t tiles = static_cast<uint32_t>(BandTiles(c));
    return std::max(1u, std::min((tiles + 3u) / 4u, static_cast<uint32_t>(c->n_cus) * c->coarse_wg_per_cu));
}

uint32_t FineGrid(const pm_ctx *c) {
    const uint32_t tiles = static_cast<uint32_t>(BandTiles(c));
    return std::max(1u, std::min(tiles, static_cast<uint32_t>(c->n_cus) * c->fine_wg_per_cu));
}

// Strip rows no item reaches get no workgroup: their tiles are background for as long as this scene
// and viewport last -- set once per slot and arena epoch, in stream order before the frame.
hipError_t ResetTileState(pm_ctx *c, FrameSlot *s, hipStream_t q) {
    if (s->state_epoch == c->arena_epoch) return hipSuccess;
    const hipError_t e = hipMemsetAsync(s->d_tile_state, 0xff, BandTiles(c) * sizeof(uint32_t), q);
    if (e == hipSuccess) s->state_epoch = c->arena_epoch;
    return e;
}

void Submitted(pm_ctx *c, int si, const pm::FrameParams &p, hipStream_t frame_stream) {
    FrameSlot *s = &c->slot[si];
    s->in_flight = true;
    s->needs_check = true;
    s->user_stream = false;
    s->params = p;
    s->frame_stream = frame_stream;
    s->parity ^= 1u;
    c->last_slot = si;
    c->frame += 1;
}

// The tile kernel's instantiation for this frame: the one-wave-per-tile one when the latest frame of this scene and viewport that said
// anything called itself dense (its pinned word; read without waiting -- a verdict a frame or two old is as good).
void ChooseTileKernel(pm_ctx *c, pm::FrameParams *p) {
    if (!c->dense_kernel_mode || !c->fused || c->last_slot < 0) return;
    uint32_t verdict = 0;
    for (size_t k = 0; k < c->slot.size() && verdict == 0; ++k) {  // newest first
        const FrameSlot &t = c->slot[(static_cast<size_t>(c->last_slot) + c->slot.size() - k) % c->slot.size()];
        if (t.h_overflow) verdict = static_cast<volatile uint32_t *>(t.h_overflow)[2];
    }
    if (verdict != 2u) return;
    p->fine_dense = 1u;
    c->frames_dense_kernel += 1;
    const uint32_t per_cu = p->handout_static ? c->fine_wg_dense_inflight : c->fine_wg_dense;
    p->fine_grid = std::max(1u, std::min(static_cast<uint32_t>(BandTiles(c)), static_cast<uint32_t>(c->n_cus) * per_cu));
}

// Workgroups of a one-launch frame (0: this frame takes two launches).  Every strip row some item reaches needs its own
// workgroup, all resident at once; beyond those, as many as there could be tiles to take from the FIFOs.
uint32_t OneLaunchGrid(const pm_ctx *c) {
    if (c->one_launch_mode == 0 || c->frame_wg_per_cu == 0 || !c->fused || c->use_row_lists || c->bin_waves != 4) return 0u;
    const uint32_t resident = static_cast<uint32_t>(c->n_cus) * c->frame_wg_per_cu;
    if (c->one_grid_rows == 0u) return 0u;  // (more than two strip rows per resident workgroup: EnsureArena)
    const uint64_t tiles = BandTiles(c);
    if (tiles > 0xffffffu) return 0u;
    return std::max(c->one_grid_rows, std::min(resident, static_cast<uint32_t>((tiles + 3u) / 4u)));
}

// The slot's FIFO entries (zeroed once: whoever takes an entry leaves a zero behind).
// (zeroed on the stream the frame is about to run on: the context's streams do not wait for the null stream)
int EnsureFifo(pm_ctx *c, FrameSlot *s, hipStream_t q) {
    const uint32_t cap = static_cast<uint32_t>(BandTiles(c));
    if (s->d_fifo && s->fifo_cap >= cap) return PM_OK;
    if (s->d_fifo) (void)hipFree(s->d_fifo);
    s->d_fifo = nullptr;
    s->fifo_cap = 0;
    const size_t bytes = static_cast<size_t>(pm::kFifos) * cap * sizeof(uint4);
    PM_TRY(hipMalloc(&s->d_fifo, bytes));
    PM_TRY(hipMemsetAsync(s->d_fifo, 0, bytes, q));
    s->fifo_cap = cap;
    return PM_OK;
}

// The third frame of a plan: what the first frames' binning kernels found per strip row (segment slots) comes back, once, and the plan
// is made again with the heaviest strip rows cut in two (EnsureArena, bin_split_mode).  A wait for the frames in flight and a copy
// of a few kilobytes: about 0.1 ms, once per scene and viewport.
int FeedBackStripRows(pm_ctx *c) {
    if (c->arena_dirty || c->arena_cap == 0) return PM_OK;  // (a new plan is about to be made anyway)
    if (c->frames_on_plan == 0xffffffffu) return PM_OK;
    c->frames_on_plan += 1;
    if (c->frames_on_plan != 3u || c->bin_split_mode != 1 || c->one_launch_mode != 0 || c->fb_applied || !c->d_sr_slots) return PM_OK;
    // (only where cutting can happen at all: the plan's rows leave workgroups of the resident grid free)
    if (c->n_sr_active >= static_cast<uint32_t>(c->n_cus) * (c->bin_wg_per_cu == 0xffu ? 5u : std::max(1u, c->bin_wg_per_cu))) return PM_OK;
    int r = SyncAll(c);
    if (r != PM_OK) return r;
    std::vector<uint32_t> &sl = c->stage_slots;
    sl.resize(c->n_sr_active);
    PM_TRY(hipMemcpy(sl.data(), c->d_sr_slots, sl.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const size_t n_rows = static_cast<size_t>(BandRows(c)) * c->strips_x;
    c->fb_slots.assign(n_rows, 0u);
    uint32_t heaviest = 0;
    for (size_t k = 0; k < sl.size() && k < c->stage_desc.size(); ++k) {
        const uint32_t x = c->stage_desc[k].x;
        const size_t i = static_cast<size_t>(x >> 16) * c->strips_x + (x & 0xffu);
        if (i < n_rows) c->fb_slots[i] += sl[k];  // (a row already cut: its halves' slots)
        if (i < n_rows) heaviest = std::max(heaviest, c->fb_slots[i]);
    }
    if (heaviest < c->bin_split_slots && c->n_sr_split == 0) {  // nothing to cut, nothing cut: the plan stands
        c->fb_applied = true;
        return PM_OK;
    }
    c->arena_dirty = true;
    c->plan_reusable = false;
    c->plans_fed_back += 1;
    return PM_OK;
}

// One frame: its kernels back to back on one in-order stream -- the context's stream
// frame % n, or the caller's.  tev (timing passes): eight events {begin, end} x {bin, clear,
// coarse, fine} carried by the dispatches themselves, so that a timed frame puts exactly the
// same packets on the queue as an untimed one.
// (developer: PM_HOST_TIMING=2 prints, when the context goes, where pm_render's host time went -- mean microseconds per section)
struct SubmitProfile {
    bool on = std::getenv("PM_HOST_TIMING") && std::atoi(std::getenv("PM_HOST_TIMING")) >= 2;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    uint64_t n = 0;
    std::chrono::steady_clock::time_point t;
    void start() { if (on) t = std::chrono::steady_clock::now(); }
    void mark(int k) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        acc[k] += std::chrono::duration<double, std::micro>(now - t).count();
        t = now;
    }
    ~SubmitProfile() {
        if (on && n) std::fprintf(stderr, "pm_render host time per frame over %llu frames (us): params %.2f | slot/stream order %.2f | in-flight query %.2f | policy %.2f | binning launch %.2f | tile launch + bookkeeping %.2f\n",
                                  static_cast<unsigned long long>(n), acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n);
    }
};
SubmitProfile g_submit_profile;

int Enqueue(pm_ctx *c, uint8_t *fb, size_t stride, hipStream_t user_stream, hipEvent_t *tev = nullptr) {
    g_submit_profile.start();
    const int si = static_cast<int>(c->frame % c->slot.size());
    FrameSlot *s = &c->slot[si];
    if (c->tiles_x != 0) {  // (a slot's viewport buffers come into being when the slot is first used)
        const int ra = AllocSlotViewport(c, s);
        if (ra != PM_OK) return ra;
    }
    if (!fb) fb = s->d_fb;
    pm::FrameParams p;
    hipStream_t q = user_stream ? user_stream : c->streams[c->frame % c->streams.size()];
    int r = FeedBackStripRows(c);
    if (r != PM_OK) return r;
    r = BuildParams(c, s, fb, stride, &p);
    if (r != PM_OK) return r;
    g_submit_profile.mark(0);
    hipEvent_t none[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t *t = tev ? tev : none;
    // the slot's previous frame (same stream unless the caller's streams are involved or the
    // slot count differs from the stream count: then the event orders the reuse)
    // (no event is recorded per frame: one recorded on the other stream when the need arises
    //  marks the end of everything submitted there so far, the slot's frame included)
    if (s->in_flight && s->frame_stream != q) {
        if (!s->user_stream) PM_TRY(hipEventRecord(s->ev_done, s->frame_stream));
        PM_TRY(hipStreamWaitEvent(q, s->ev_done, 0));
    }
    // frames that target the same caller-owned buffer must not overlap each other
    if (c->last_slot >= 0 && c->last_slot != si) {
        FrameSlot &l = c->slot[c->last_slot];
        if (l.in_flight && l.params.fb == fb && l.frame_stream != q) {
            if (!l.user_stream) PM_TRY(hipEventRecord(l.ev_done, l.frame_stream));
            PM_TRY(hipStreamWaitEvent(q, l.ev_done, 0));
        }
    }
    // Tile hand-out: drawn when this frame will have the device to itself, static when the previous
    // frame is still running (the frames then overlap, and neighbours fill what a static hand-out
    // leaves idle): lone frame -4.6 us, sustained throughput as before.  PM_HANDOUT=1 / 2 pins it.
    g_submit_profile.mark(1);
    p.handout_static = c->handout == 1 ? 1u : 0u;
    if (c->handout == 0 && c->last_slot >= 0) {
        const FrameSlot &l = c->slot[c->last_slot];
        if (l.in_flight) {
            // The question "is the previous frame still running?" costs 3.3 us of host time (hipStreamQuery, measured: a quarter of a
            // pm_render call, and small frames in flight are bound by exactly that).  Frames submitted back to back get the same
            // answer most of the time: it is asked once in eight such frames and kept in between (a gap of 0.2 ms asks again).
            const auto now = std::chrono::steady_clock::now();
            const bool back_to_back = std::chrono::duration<double, std::micro>(now - c->last_submit).count() < 200.0;
            c->last_submit = now;
            if (back_to_back && c->query_keep != 0u) {
                c->query_keep -= 1u;
            } else {
                const hipError_t st = l.user_stream ? hipEventQuery(l.ev_done) : hipStreamQuery(l.frame_stream);
                const bool was_running = c->prev_running;
                c->prev_running = st == hipErrorNotReady;
                // (kept for the rest of the eight -- except the FIRST "idle" behind "running" or behind a pause, kept for one frame: the first
                //  frames of a burst find the device idle, and a burst planned as lone frames for eight frames on the strength of that was
                //  measured 14 % slower; a loop of lone frames -- idle, idle, idle -- asks once in eight as before: asking every other
                //  frame cost the 4K Tiger's lone frame 0.5 us)
                c->query_keep = (c->prev_running || (!was_running && back_to_back && c->idle_seen)) ? c->query_every - 1u : min(1u, c->query_every - 1u);
                c->idle_seen = !c->prev_running && back_to_back;
                (void)hipGetLastError();  // (hipErrorNotReady is an answer, not a failure)
            }
            p.handout_static = c->prev_running && l.frame_stream != q ? 1u : 0u;  // (behind it on the same stream: alone all the same)
        } else {
            c->query_keep = 0u;
        }
    }
    g_submit_profile.mark(2);
    // overlapping frames also keep fewer tiles for whole workgroups (a workgroup tile idles three
    // waves while its list is built: cheap when the frame is alone and its longest lists set the
    // span, wasteful when neighbours could use the SIMDs): lone frame -1.4 us, sustained +2.6 %
    if (p.handout_static) SetClassThresholds(c, &p, c->heavy_stream);
    // ... and bin with a wave per strip row: behind other frames what counts is instructions issued, not this frame's
    // latency (a wave per row: a quarter fewer of them; 4K Tiger sustained 255 -> 266 k Mpix/s).  Only frames with enough
    // strip rows to fill the chip that way (at 1080p, 544 rows, a row's 50 us then set the pace: 142 -> 96 k), and only where
    // no workgroup walks a chain of rows -- the chains are linked for one grid (EnsureArena).
    // ... and only LIGHT rows: a wave's record holds 64 candidates, and a strip row with more of them pays a second pass over
    // everything (held-out workload 2, 2 k blobs at 2048^2, 79 candidates per row: sustained 164 -> 139 us per frame with
    // workgroups; the rule is the one that picks a wave per row for a frame alone, EnsureArena).
    // (round 5, found by the held-out policy check: held-out 3 -- 2 025 light strip rows, more than the 1 280 workgroups of the plan's
    //  grid, so its rows are chained -- ran 14 % faster in flight with a wave per row.  Sixteen one-wave groups per CU hold 4 096 rows at
    //  once: such a frame gets a wave for EVERY row and does not walk the chains.)
    // ... from the work list WITHOUT the cuts (a cut strip row is two rows' worth of fixed work: worth it on a lone frame's critical
    // path, a loss where frames overlap): the same arena regions, every cut row whole again
    uint32_t n_rows_frame = c->n_sr_active;
    if (p.handout_static && c->n_sr_split != 0 && c->n_sr_whole != 0) {
        p.sr_desc = c->d_sr_desc_whole;
        p.n_sr_active = c->n_sr_whole;
        p.bin_grid = std::min(p.bin_grid, c->n_sr_whole);
        p.sr_slots = nullptr;  // (the report is by entry of the list with the cuts)
        n_rows_frame = c->n_sr_whole;
    }
    if (p.handout_static && c->bin_waves_inflight == 1 && c->bin_waves == 4 && n_rows_frame >= 4u * static_cast<uint32_t>(c->n_cus) &&
        c->plan_cands <= 32ull * n_rows_frame) {
        if (n_rows_frame <= p.bin_grid) {
            p.bin_waves = 1;
            c->frames_bin_wave_inflight += 1;
        } else if (n_rows_frame <= 16u * static_cast<uint32_t>(c->n_cus) && c->bin_wg_per_cu == 0xffu) {
            p.bin_waves = 1;
            p.bin_grid = n_rows_frame;
            p.bin_no_chains = 1u;
            c->frames_bin_wave_inflight += 1;
            c->frames_bin_no_chains += 1;
        }
    }
    if (p.bin_waves == 1) c->frames_bin_wave += 1;
    // ... and a smaller persistent grid: three tile workgroups per CU leave two slots (LDS, VGPRs) to
    // the neighbours' binning workgroups (Tiger 4K sustained 221 -> 227 k Mpix/s, the other configurations
    // unchanged; two cost config 4 2 %; alone, five end the frame 0.8 us earlier)
    if (p.handout_static && c->fine_wg_per_cu_inflight < c->fine_wg_per_cu)
        p.fine_grid = std::max(1u, std::min(p.fine_grid, static_cast<uint32_t>(c->n_cus) * c->fine_wg_per_cu_inflight));
    // (the waves a frame's dense / not dense verdict is judged against: the GENERAL kernel's grid for this kind of frame, whichever
    //  instantiation runs it -- round-5 advisor: judged against its own, larger grid the one-wave kernel called a scene "not dense"
    //  that the general kernel called dense, and such a scene changed kernels every other frame)
    p.verdict_waves = p.fine_grid * 4u;
    ChooseTileKernel(c, &p);
    const uint32_t n_striprows = BandRows(c) * c->strips_x;
    PM_TRY(ResetTileState(c, s, q));
    // One launch for the whole frame (pm_frame.hip) when the frame has the device to itself -- two such launches at once could
    // each hold the slots the other's binning workgroups wait for -- and every strip row gets its own resident workgroup.
    const uint32_t frame_grid = OneLaunchGrid(c);
    if (frame_grid != 0u && !tev && !p.handout_static && !c->one_launch_broken) {
        r = EnsureFifo(c, s, q);
        if (r != PM_OK) return r;
        p.fifo = s->d_fifo;
        p.fifo_cap = s->fifo_cap;
        if (c->cap_counts) {
            p.dbg_counts = c->cap_counts;
            p.dbg_solid = c->cap_solid;
            p.dbg_cmds = c->cap_cmds;
            p.dbg_max = c->cap_max;
        }
#ifdef PM_EMU
        const bool split = true;  // (the CPU emulation runs workgroups one after the other: the binning roles first, then everybody takes from the FIFOs)
#else
        const bool split = c->one_launch_split;
#endif
        if (split) {
            p.one_launch = 1u;
            pm::LaunchFrame(p, frame_grid, q);
            p.one_launch = 2u;
            pm::LaunchFrame(p, frame_grid, q);
        } else {
            p.one_launch = 3u;
            pm::LaunchFrame(p, frame_grid, q);
        }
        PM_TRY(hipGetLastError());
        c->frames_one_launch += 1;
        Submitted(c, si, p, q);
        s->user_stream = user_stream != nullptr && std::find(c->streams.begin(), c->streams.end(), q) == c->streams.end();
        if (s->user_stream) PM_TRY(hipEventRecord(s->ev_done, q));
        return PM_OK;
    }
    // The resolved tiles' pixels (26 MB of stores at Tiger 4K): extra workgroups of the tile kernel's
    // launch for a frame alone (one launch less: -7 us), a launch of its own between the two kernels
    // when other frames are in flight -- there it fills the SIMDs binning's tail leaves idle instead of
    // queueing behind the persistent tile workgroups (sustained +3 %).
    // (Small frames keep the fold: at 1080p a pipelined frame is 17 us, about what submitting two launches
    //  costs the host, and a third one took the sustained rate from 122 k to 100 k Mpix/s.)
    // (round 6, PM_FOLD_CLEAR=3, the default: neither -- the binning launch writes them, every strip row's workgroup its own at the
    //  row's end and extra workgroups those of the strip rows no item reaches: stores under a kernel of dependent chains, and the
    //  tile kernel's first tiles no longer share the chip with 2 025 clearing workgroups)
    if (c->fold_clear_mode == 3 && p.handout_static) p.clear_in_bin = 1u;
    const bool fold = p.clear_in_bin == 0u && (c->fold_clear_mode == 1 || (c->fold_clear_mode >= 2 && (!p.handout_static || BandTiles(c) < 16384u)));
    g_submit_profile.mark(3);
    pm::LaunchBin(p, q, t[0], t[1]);
    g_submit_profile.mark(4);
    if (!fold && p.clear_in_bin == 0u) pm::LaunchClear(p, n_striprows, q, t[2], t[3]);  // (needs tile_state)
    if (!c->fused) pm::LaunchCoarse(p, CoarseGrid(c), false, q, t[4], t[5]);
    pm::LaunchFine(p, fold ? n_striprows : 0u, c->fused, q, t[6], t[7]);
    PM_TRY(hipGetLastError());
    Submitted(c, si, p, q);
    s->user_stream = user_stream != nullptr && std::find(c->streams.begin(), c->streams.end(), q) == c->streams.end();
    if (s->user_stream) PM_TRY(hipEventRecord(s->ev_done, q));  // the only handle kept on a caller's stream
    g_submit_profile.mark(5);
    g_submit_profile.n += 1;
    return PM_OK;
}

// Scene index: chunk table offsets on the host (a prefix sum over the item headers we
// already copied), chunk boxes on the device.  Built once per scene, like the
// reference builds its per-item ShortBbox array at encode time (src/lib.rs:88-97).
int BuildSceneIndex(pm_ctx *c) {
    const uint8_t *meta = c->item_meta.data();
    uint32_t n, items_ix;
    std::memcpy(&n, meta, 4);
    std::memcpy(&items_ix, meta + 4, 4);
    std::vector<uint32_t> &base = c->chunk_base_host;  // (the source of an asynchronous copy: it lives in the context)
    base.resize(static_cast<size_t>(n) + 1);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        base[i] = static_cast<uint32_t>(total);
        const uint8_t *it = meta + items_ix + 32ull * i;
        uint32_t tag, npt;
        std::memcpy(&tag, it, 4);
        std::memcpy(&npt, it + 12, 4);
        tag &= 0xffffu;
        uint64_t nseg = 0;
        if (tag == pm::kItemFill) nseg = npt;
        else if (tag == pm::kItemPoly && npt >= 2) nseg = npt - 1;
        total += (nseg + pm::kChunkSegs - 1) / pm::kChunkSegs;
    }
    base[n] = static_cast<uint32_t>(total);
    if (total > 0xffffffull) {  // (a chunk's index inside its item travels in 24 bits, pm_bin_kernel's survivor lists)
        SetError("scene has too many segments for the chunk index (2^24 chunks of 4 segments)");
        return PM_ERR_CAPACITY;
    }
    if (base.size() > c->chunk_base_cap) {
        if (c->d_chunk_base) (void)hipFree(c->d_chunk_base);
        c->d_chunk_base = nullptr;
        PM_TRY(hipMalloc(&c->d_chunk_base, base.size() * sizeof(uint32_t)));
        c->chunk_base_cap = base.size();
    }
    if (total > c->chunk_bbox_cap || !c->d_chunk_bbox) {
        if (c->d_chunk_bbox) (void)hipFree(c->d_chunk_bbox);
        c->d_chunk_bbox = nullptr;
        PM_TRY(hipMalloc(&c->d_chunk_bbox, std::max<uint64_t>(total, 1) * sizeof(float4)));
        c->chunk_bbox_cap = std::max<uint64_t>(total, 1);
    }
    const uint64_t n_sup = (total + pm::kSuperChunks - 1) / pm::kSuperChunks;
    if (n_sup > c->sup_bbox_cap || !c->d_sup_bbox) {
        if (c->d_sup_bbox) (void)hipFree(c->d_sup_bbox);
        c->d_sup_bbox = nullptr;
        PM_TRY(hipMalloc(&c->d_sup_bbox, std::max<uint64_t>(n_sup, 1) * sizeof(float4)));
        c->sup_bbox_cap = std::max<uint64_t>(n_sup, 1);
    }
    c->n_chunks = static_cast<uint32_t>(total);
    PM_TRY(hipMemcpyAsync(c->d_chunk_base, base.data(), base.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    pm::LaunchIndex(c->d_scene, n, c->dev_items_ix, c->d_chunk_base, c->n_chunks, c->d_chunk_bbox, c->d_sup_bbox, c->stream);
    PM_TRY(hipGetLastError());
    // (no wait: frames run on streams that are ordered behind c->stream where it matters -- Enqueue below --
    //  and the next scene replacement starts with SyncAll before `base` is touched again)
    return PM_OK;
}

// A scene replacement starts by forgetting the old scene: if the upload, the flatten kernels or
// the validation fail, no later pm_render can pair the old item count / index / arena bounds
// with new, unvalidated device bytes (BuildParams refuses to render without a scene).
void InvalidateScene(pm_ctx *c) {
    c->scene_bytes = 0;
    c->user_scene_bytes = 0;
    c->n_items = 0;
    c->n_chunks = 0;
    c->item_meta.clear();
    c->last_slot = -1;
    c->arena_dirty = true;
    for (auto &t : c->slot)  // (another scene: whether its frames are dense is for its own tile kernels to say)
        if (t.h_overflow) t.h_overflow[2] = 0;
}

int ReserveDevice(pm_ctx *c, size_t cap, size_t keep_bytes);

// Nested groups (extension, src/lib.rs:148): a scene with PietGroup items renders like the same
// items inlined depth first, in paint order.  The kernels only ever see a flat group, so the
// host appends that flat form -- {n, items_ix}, boxes, 32-byte items, the items' point arrays
// stay where they are -- behind the scene bytes.  Every offset is checked here.
struct FlatGroup {
    std::vector<uint8_t> boxes, items;
    uint32_t n = 0;
    uint64_t visits = 0;  // groups entered: a DAG of (empty) groups shared by many parents is cut off too
};

bool FlattenGroup(const uint8_t *sc, size_t len, uint64_t group, int depth, FlatGroup *out) {
    if (depth > 32 || group + 8 > len || (group & 3u) || ++out->visits > (1u << 24)) return false;
    uint32_t n, items_ix;
    std::memcpy(&n, sc + group, 4);
    std::memcpy(&items_ix, sc + group + 4, 4);
    if (group + 8ull + 8ull * n > len || static_cast<uint64_t>(items_ix) + 32ull * n > len || (items_ix & 3u)) return false;
    for (uint32_t i = 0; i < n; ++i) {
        const uint8_t *it = sc + items_ix + 32ull * i;
        uint32_t tag, child;
        std::memcpy(&tag, it, 4);
        if ((tag & 0xffffu) == pm::kItemGroup) {
            std::memcpy(&child, it + 8, 4);
            if (!FlattenGroup(sc, len, child, depth + 1, out)) return false;
            continue;
        }
        if (out->n >= (1u << 24)) return false;  // (also ends cyclic scenes)
        out->boxes.insert(out->boxes.end(), sc + group + 8 + 8ull * i, sc + group + 16 + 8ull * i);
        out->items.insert(out->items.end(), it, it + 32);
        out->n += 1;
    }
    return true;
}

// The scene bytes are resident in d_scene (`host` = the same bytes in host memory, or nullptr for
// a scene the flatten kernels wrote).  Builds the host copy of the drawn group in normal form
// {n, 8 + 8n}{boxes}{items} (validation, arena sizing) and the scene index.
int SetScene(pm_ctx *c, size_t bytes, const uint8_t *host) {
    if (bytes < 8) return PM_ERR_SCENE;
    const WallTimer timer;
    auto fail = [&](const char *what) {
        SetError(what);
        InvalidateScene(c);
        return PM_ERR_SCENE;
    };
    // (a scene the flatten kernels wrote: its head -- header, boxes, items -- came back with the totals)
    const uint8_t *head = c->flatten_cache.meta_bytes >= 8 && !host ? c->flatten_cache.h_meta + 32 : nullptr;
    const size_t head_bytes = head ? c->flatten_cache.meta_bytes : 0;
    uint32_t hdr[2];
    if (host) {
        std::memcpy(hdr, host, 8);
    } else if (head) {
        std::memcpy(hdr, head, 8);
    } else {
        PM_TRY(hipMemcpyAsync(hdr, c->d_scene, 8, hipMemcpyDeviceToHost, c->stream));
        PM_TRY(hipStreamSynchronize(c->stream));
    }
    const uint64_t n0 = hdr[0], items0 = hdr[1];
    if (items0 + 32ull * n0 > bytes || items0 < 8ull + 8ull * n0 || (items0 & 7u)) return fail("scene header out of range");
    std::vector<uint8_t> meta(8 + 40 * n0);
    uint32_t dev_bbox_ix = 8, dev_items_ix = hdr[1];
    size_t total = bytes;
    bool nested = false;
    if (host)
        for (uint64_t i = 0; i < n0 && !nested; ++i) {
            uint32_t tag;
            std::memcpy(&tag, host + items0 + 32 * i, 4);
            nested = (tag & 0xffffu) == pm::kItemGroup;
        }
    uint32_t n = hdr[0];
    if (nested) {
        FlatGroup flat;
        if (!FlattenGroup(host, bytes, 0, 0, &flat)) return fail("nested groups: offset out of range, too deep, or cyclic");
        n = flat.n;
        const size_t root = (bytes + 7u) & ~static_cast<size_t>(7u);
        total = root + 8 + 40ull * n;
        std::vector<uint8_t> blk(total - bytes, 0);
        const uint32_t ghdr[2] = {n, static_cast<uint32_t>(root + 8 + 8ull * n)};
        std::memcpy(blk.data() + (root - bytes), ghdr, 8);
        std::memcpy(blk.data() + (root - bytes) + 8, flat.boxes.data(), flat.boxes.size());
        std::memcpy(blk.data() + (root - bytes) + 8 + 8ull * n, flat.items.data(), flat.items.size());
        if (total > c->dev_scene_cap) {
            PM_TRY(hipStreamSynchronize(c->stream));  // the upload of the scene bytes is still in flight
            // (device copy only: the pinned buffer the caller encodes into stays where it is)
            const int rr = ReserveDevice(c, total + (total >> 3), bytes);
            if (rr != PM_OK) {
                InvalidateScene(c);
                return rr;
            }
        }
        PM_TRY(hipMemcpyAsync(c->d_scene + bytes, blk.data(), blk.size(), hipMemcpyHostToDevice, c->stream));
        PM_TRY(hipStreamSynchronize(c->stream));  // blk is a stack-owned source
        dev_bbox_ix = static_cast<uint32_t>(root + 8);
        dev_items_ix = ghdr[1];
        meta.resize(8 + 40ull * n);
        std::memcpy(meta.data() + 8, flat.boxes.data(), flat.boxes.size());
        std::memcpy(meta.data() + 8 + 8ull * n, flat.items.data(), flat.items.size());
    } else if (host) {
        std::memcpy(meta.data() + 8, host + 8, 8 * n0);
        std::memcpy(meta.data() + 8 + 8 * n0, host + items0, 32 * n0);
    } else if (head && items0 + 32 * n0 <= head_bytes) {
        std::memcpy(meta.data() + 8, head + 8, 8 * n0);
        std::memcpy(meta.data() + 8 + 8 * n0, head + items0, 32 * n0);
    } else if (n0) {
        PM_TRY(hipMemcpyAsync(meta.data() + 8, c->d_scene + 8, 8 * n0, hipMemcpyDeviceToHost, c->stream));
        PM_TRY(hipMemcpyAsync(meta.data() + 8 + 8 * n0, c->d_scene + items0, 32 * n0, hipMemcpyDeviceToHost, c->stream));
        PM_TRY(hipStreamSynchronize(c->stream));
    }
    const uint32_t mhdr[2] = {n, 8u + 8u * n};
    std::memcpy(meta.data(), mhdr, 8);
    uint32_t n_checked = 0;
    const int r = ValidateScene(meta.data(), meta.size(), total, &n_checked);
    if (r != PM_OK) {
        (void)fail("scene buffer failed validation");
        return r;
    }
    c->item_meta.swap(meta);
    c->n_items = n;
    c->dev_bbox_ix = dev_bbox_ix;
    c->dev_items_ix = dev_items_ix;
    c->arena_dirty = true;
    c->last_slot = -1;
    const int ri = BuildSceneIndex(c);
    if (ri != PM_OK) {
        InvalidateScene(c);
        return ri;
    }
    c->user_scene_bytes = bytes;
    c->scene_bytes = total;  // only now is there a scene to render
    c->t_index_ms = timer.ms();
    return PM_OK;
}

int CheckSceneCap(size_t cap) {
    if (cap > 0xffffffffull) {  // offsets in the scene format (and in the kernels' bounds checks) are u32
        SetError("scene buffers are limited to 4 GiB - 1 (u32 offsets in the scene format)");
        return PM_ERR_CAPACITY;
    }
    return PM_OK;
}

// Grows the device copy of the scene (keeping its first keep_bytes).  The pinned host buffer is not
// touched: a pointer pm_scene_buffer returned stays valid (round-2 advisor finding).
int ReserveDevice(pm_ctx *c, size_t cap, size_t keep_bytes = 0) {
    if (cap <= c->dev_scene_cap) return PM_OK;
    keep_bytes = std::max(keep_bytes, c->scene_bytes);
    const int rc = CheckSceneCap(cap);
    if (rc != PM_OK) return rc;
    uint8_t *d = nullptr;
    PM_TRY(hipMalloc(&d, cap));
    if (c->d_scene) {
        if (keep_bytes) (void)hipMemcpy(d, c->d_scene, std::min(keep_bytes, c->dev_scene_cap), hipMemcpyDeviceToDevice);
        (void)hipFree(c->d_scene);
    }
    c->d_scene = d;
    c->dev_scene_cap = cap;
    return PM_OK;
}

// pm_create / pm_scene_reserve: the pinned host buffer AND the device copy.
int ReserveScene(pm_ctx *c, size_t cap) {
    if (cap > c->scene_cap) {
        const int rc = CheckSceneCap(cap);
        if (rc != PM_OK) return rc;
        uint8_t *h = nullptr;
        PM_TRY(hipHostMalloc(&h, cap, hipHostMallocDefault));
        std::memset(h, 0, cap);
        if (c->h_scene) {
            std::memcpy(h, c->h_scene, c->scene_cap);
            (void)hipHostFree(c->h_scene);
        }
        c->h_scene = h;
        c->scene_cap = cap;
    }
    return ReserveDevice(c, cap);
}

}  // namespace

// ---- what pm_gather.hip needs to know about a context -------------------------------------
namespace pm {
void SetLastError(const std::string &s) { SetError(s); }
int ContextDevice(const pm_ctx *c) { return c->device; }
hipStream_t ContextStream(pm_ctx *c) { return c->stream; }
int ContextViewport(const pm_ctx *c, uint32_t *width, uint32_t *height, uint32_t *row0, uint32_t *row1) {
    if (!c || c->tiles_x == 0) return PM_ERR_INVALID;
    *width = c->width;
    *height = c->height;
    *row0 = c->row0;
    *row1 = c->row1;
    return PM_OK;
}
// The last frame's framebuffer; work submitted to gather_stream after this call runs behind the frame.
int ContextLastFrame(pm_ctx *c, const void **fb, size_t *stride, hipStream_t gather_stream) {
    if (!c || c->last_slot < 0) {
        SetError("pm_gather: no frame rendered yet");
        return PM_ERR_INVALID;
    }
    FrameSlot &s = c->slot[c->last_slot];
    if (s.in_flight && s.frame_stream != gather_stream) {
        if (!s.user_stream) PM_TRY(hipEventRecord(s.ev_done, s.frame_stream));
        PM_TRY(hipStreamWaitEvent(gather_stream, s.ev_done, 0));
    }
    *fb = s.params.fb;
    *stride = s.params.fb_stride;
    return PM_OK;
}
}  // namespace pm

extern "C" {

const char *pm_last_error(void) { return g_last_error.c_str(); }

uint32_t pm_abi_version(void) { return PM_ABI_VERSION; }

pm_ctx *pm_create(int device, int *err) {
    int dummy;
    if (!err) err = &dummy;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        SetError("no HIP device visible: piet_metal_amd has no CPU fallback");
        *err = PM_ERR_NO_DEVICE;
        return nullptr;
    }
    if (device < 0 || device >= count) {
        SetError("device index out of range");
        *err = PM_ERR_INVALID;
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        SetError(std::string("device is not gfx950 (MI355X): ") + prop.gcnArchName);
        *err = PM_ERR_NO_DEVICE;
        return nullptr;
    }
    pm_ctx *c = new (std::nothrow) pm_ctx();
    if (!c) {
        *err = PM_ERR_CAPACITY;
        return nullptr;
    }
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    auto fail = [&](hipError_t e, const char *what) {
        HipFail(e, what);
        *err = PM_ERR_HIP;
        pm_destroy(c);
        return static_cast<pm_ctx *>(nullptr);
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
    // Frames in flight (see the top of this file); tunable for experiments.
    const int n_streams = EnvInt("PM_FRAME_STREAMS", kDefaultFrameStreams, 1, kMaxSlots);
    c->slot.resize(static_cast<size_t>(EnvInt("PM_SLOTS", n_streams, 1, kMaxSlots)));
    for (int i = 0; i < n_streams; ++i) {
        hipStream_t q = nullptr;
        if ((e = hipStreamCreateWithFlags(&q, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        c->streams.push_back(q);
    }
    c->stream = c->streams[0];
    c->coarse_wg_per_cu = static_cast<uint32_t>(EnvInt("PM_COARSE_WG_PER_CU", 5, 1, 16));
    c->split_mode = static_cast<uint32_t>(EnvInt("PM_FINE_SPLIT", 1, 0, 1));
    c->dense_factor = static_cast<uint32_t>(EnvInt("PM_DENSE_FACTOR", 4, 1, 1 << 20));
    c->dense_kernel_mode = EnvInt("PM_DENSE_KERNEL", 1, 0, 1);
    c->fine_wg_dense = static_cast<uint32_t>(EnvInt("PM_FINE_WG_PER_CU_DENSE", 6, 1, 16));
    c->fine_wg_dense_inflight = static_cast<uint32_t>(EnvInt("PM_FINE_WG_PER_CU_DENSE_INFLIGHT", 4, 1, 16));
    c->heavy_stream = static_cast<uint32_t>(EnvInt("PM_HEAVY_STREAM", 72, 1, 1 << 20));
    c->heavy_stream_lone = static_cast<uint32_t>(EnvInt("PM_HEAVY_STREAM_LONE", std::min<int>(40, static_cast<int>(c->heavy_stream)), 1, 1 << 20));
    c->vheavy_stream = static_cast<uint32_t>(EnvInt("PM_VHEAVY_STREAM", 112, 1, 1 << 20));
    c->fold_clear_mode = EnvInt("PM_FOLD_CLEAR", 3, 0, 4);
    c->fold_clear = c->fold_clear_mode != 0;
    c->fused = EnvInt("PM_FUSED", 1, 0, 1) != 0;
    c->handout = EnvInt("PM_HANDOUT", 0, 0, 2);
    c->fine_wg_per_cu = static_cast<uint32_t>(EnvInt("PM_FINE_WG_PER_CU", 5, 1, 16));
    c->fine_wg_per_cu_inflight = static_cast<uint32_t>(EnvInt("PM_FINE_WG_PER_CU_INFLIGHT", 3, 1, 16));
    c->bin_wg_per_cu = static_cast<uint32_t>(EnvInt("PM_BIN_WG_PER_CU", 0xff, 0, 0xff));
    c->bin_waves_env = static_cast<uint32_t>(EnvInt("PM_BIN_WAVES", 0, 0, 4));
    c->bin_waves_inflight = c->bin_waves_env ? c->bin_waves_env : static_cast<uint32_t>(EnvInt("PM_BIN_WAVES_INFLIGHT", 1, 1, 4));
    c->bin_prio_slots = static_cast<uint32_t>(EnvInt("PM_BIN_PRIO_SLOTS", 320, 0, 1 << 30));
    c->query_every = static_cast<uint32_t>(EnvInt("PM_QUERY_EVERY", 8, 1, 1 << 20));
    c->bin_split_mode = EnvInt("PM_BIN_SPLIT", 1, 0, 2);
    c->bin_wt_mode = EnvInt("PM_BIN_WT", 2, 0, 2);
    c->bin_split_cands = static_cast<uint32_t>(EnvInt("PM_BIN_SPLIT_CANDS", 1 << 30, 1, 1 << 30));
    c->bin_split_slots = static_cast<uint32_t>(EnvInt("PM_BIN_SPLIT_SLOTS", 224, 1, 1 << 30));
    c->bin_split_fill = static_cast<uint32_t>(EnvInt("PM_BIN_SPLIT_FILL", 15, 1, 16));
    c->one_launch_mode = EnvInt("PM_ONE_LAUNCH", 0, 0, 1);
    c->one_launch_split = EnvInt("PM_ONE_LAUNCH_SPLIT", 0, 0, 1) != 0;
    c->frame_spin_ticks = static_cast<uint32_t>(EnvInt("PM_ONE_LAUNCH_SPIN_US", 2000, 1, 1000000)) * 100u;
    {
        // the one-launch grid must be resident as a whole: what the occupancy API says a CU holds, four at most (the
        // kernel's registers are sized for four)
        const int res = pm::FrameKernelResidency();
        c->frame_wg_per_cu = res >= 4 ? 4u : 0u;
        if (const char *v = std::getenv("PM_FRAME_WG_PER_CU")) c->frame_wg_per_cu = static_cast<uint32_t>(std::max(0, std::min(res, std::atoi(v))));
    }

    for (auto &ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(e, "hipEventCreate");
    for (auto &s : c->slot) {
        if ((e = hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipMalloc(&s.d_ctr, 2 * sizeof(pm::Counters))) != hipSuccess) return fail(e, "hipMalloc(counters)");
        if ((e = hipMemset(s.d_ctr, 0, 2 * sizeof(pm::Counters))) != hipSuccess) return fail(e, "hipMemset(counters)");
        if ((e = hipHostMalloc(&s.h_overflow, 64, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc(overflow word)");
        s.h_overflow[0] = s.h_overflow[1] = s.h_overflow[2] = 0;
        void *dp = nullptr;
        if ((e = hipHostGetDevicePointer(&dp, s.h_overflow, 0)) != hipSuccess) return fail(e, "hipHostGetDevicePointer");
        s.d_overflow = static_cast<uint32_t *>(dp);
    }
    Luts *l = new (std::nothrow) Luts();
    if (!l) {
        *err = PM_ERR_CAPACITY;
        pm_destroy(c);
        return nullptr;
    }
    BuildLuts(l);
    e = hipMalloc(&c->d_lut_srgb2lin, sizeof(l->srgb2lin));
    if (e == hipSuccess) e = hipMalloc(&c->d_lut_unorm2h, sizeof(l->unorm2h));
    if (e == hipSuccess) e = hipMalloc(&c->d_lut_lin2srgb, sizeof(l->lin2srgb));
    if (e == hipSuccess) e = hipMemcpy(c->d_lut_srgb2lin, l->srgb2lin, sizeof(l->srgb2lin), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_lut_unorm2h, l->unorm2h, sizeof(l->unorm2h), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_lut_lin2srgb, l->lin2srgb, sizeof(l->lin2srgb), hipMemcpyHostToDevice);
    delete l;
    if (e != hipSuccess) return fail(e, "lookup tables");
    // The working set of an ordinary scene is reserved now (HBM is 288 GB; a context is created once):
    // flatten scratch, scene index, binning lists and frame slot 0's arenas.  The first scene and the
    // first frame then allocate nothing; larger scenes grow the buffers as before.
    if (EnvInt("PM_PREALLOC", 1, 0, 1)) {
        e = c->flatten_cache.Reserve(4096, 65536);
        if (e == hipSuccess) e = hipMalloc(&c->d_chunk_base, (65536 + 1) * sizeof(uint32_t));
        if (e == hipSuccess) c->chunk_base_cap = 65536 + 1;
        if (e == hipSuccess) e = hipMalloc(&c->d_chunk_bbox, (1u << 19) * sizeof(float4));
        if (e == hipSuccess) c->chunk_bbox_cap = 1u << 19;
        if (e == hipSuccess) e = hipMalloc(&c->d_sup_bbox, (1u << 16) * sizeof(float4));
        if (e == hipSuccess) c->sup_bbox_cap = 1u << 16;
        if (e == hipSuccess) e = hipMalloc(&c->d_sr_desc, 65536 * sizeof(uint4));
        if (e == hipSuccess) e = hipMalloc(&c->d_sr_list, 65536 * sizeof(uint2));
        if (e == hipSuccess) e = hipMalloc(&c->d_sr_slots, 65536 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&c->d_sr_desc_whole, 65536 * sizeof(uint4));
        if (e == hipSuccess) c->sr_desc_cap = 65536;
        if (e == hipSuccess) e = hipMalloc(&c->d_band_bbox, 65536 * sizeof(uint2));
        if (e == hipSuccess) e = hipMalloc(&c->d_band_item, 65536 * sizeof(uint32_t));
        if (e == hipSuccess) c->band_cap = 65536;
        // ... and every frame slot's viewport buffers for anything up to 8192 x 8192 (BASELINE config 5: 256 MB of pixels, 42 MB of queues
        // and per-tile tables) and its arenas, sized for config 5 (2.2 GB of worst-case binning regions, 1.7 GB of tile arena): four slots are
        // 16 GB, 6 % of the HBM, and ~0.4 s of pm_create.  A resize within that allocates nothing, whichever slot the next frame lands on
        // (round 6 first reserved slot 0 only: config 5's first frame was 1.2 ms when it landed there and 85-115 ms -- one hipMalloc of
        // 3.8 GB -- when the frame counter stood elsewhere; PM_PREALLOC_SLOTS / _MPIXELS / _ARENA_MB / _TILE_MB).
#ifdef PM_EMU
        const int arena_mb = 192, tile_mb = 96, slots_dflt = 1;  // (the CPU emulation's "HBM" is the host's memory)
#else
        const int arena_mb = 2304, tile_mb = 1792, slots_dflt = static_cast<int>(kMaxSlots);
#endif
        const size_t n_pre = std::min<size_t>(c->slot.size(), static_cast<size_t>(EnvInt("PM_PREALLOC_SLOTS", slots_dflt, 1, static_cast<int>(kMaxSlots))));
        const size_t px = static_cast<size_t>(EnvInt("PM_PREALLOC_MPIXELS", 64, 0, 4096)) << 20;
        const size_t tables = std::max<size_t>(px / (pm::kTileW * pm::kTileH), 4);
        const uint32_t arena0 = static_cast<uint32_t>(EnvInt("PM_PREALLOC_ARENA_MB", arena_mb, 1, 12288)) << 18;  // dwords
        const uint32_t tile0 = std::max<uint32_t>(static_cast<uint32_t>(EnvInt("PM_PREALLOC_TILE_MB", tile_mb, 1, 16384)) << 16,  // quads
                                                  ((1u << 22) * pm::kCmdQuadsNum) / pm::kCmdQuadsDen);  // (what EnsureArena asks for at least)
        for (size_t si = 0; si < n_pre && e == hipSuccess; ++si) {
            FrameSlot &fs = c->slot[si];
            if (px != 0) {
                e = hipMalloc(&fs.d_fb, px * 4);
                if (e == hipSuccess) e = hipMalloc(&fs.d_queue, pm::kClasses * tables * sizeof(uint4) + 3 * tables * sizeof(uint32_t));
                if (e == hipSuccess) {
                    fs.fb_cap = px * 4;
                    fs.tables_cap = tables;
                }
            }
            if (e == hipSuccess) e = hipMalloc(&fs.d_arena, static_cast<size_t>(arena0) * sizeof(uint32_t));
            if (e == hipSuccess) fs.arena_cap = arena0;
            if (e == hipSuccess && !std::getenv("PM_PTCL_INITIAL_CMDS")) {
                e = hipMalloc(&fs.d_ptcl, static_cast<size_t>(tile0) * sizeof(uint4));
                if (e == hipSuccess) fs.ptcl_cap = tile0;
            }
        }
        if (e != hipSuccess) return fail(e, "working-set reservation");
    }
    if (ReserveScene(c, 16u << 20) != PM_OK) {  // 16 MiB like PietRenderer.m:53
        *err = PM_ERR_HIP;
        pm_destroy(c);
        return nullptr;
    }
    // -initWithMetalKitView: builds its pipeline states up front (PietRenderer.m:33-47); here every
    // kernel of the path runs once on a three-point scene, so that the first real scene and frame do
    // not pay for code-object loading (about 1.5 ms spread over a dozen first launches).
    if (EnvInt("PM_WARMUP", 1, 0, 1)) {
        const pm_path wp = {0u, 3u, PM_PATH_FILL | PM_PATH_STROKE, 0x000000ffu, 0x000000ffu, 1.0f};
        pm_path_el we[3] = {};
        we[0].tag = PM_EL_MOVE; we[0].p[0] = 2.0; we[0].p[1] = 2.0;
        we[1].tag = PM_EL_LINE; we[1].p[0] = 12.0; we[1].p[1] = 3.0;
        we[2].tag = PM_EL_CURVE; we[2].p[0] = 12.0; we[2].p[1] = 8.0; we[2].p[2] = 8.0; we[2].p[3] = 12.0; we[2].p[4] = 3.0; we[2].p[5] = 12.0;
        const double ident[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 0.0};
        int rw = pm_resize(c, 40, 24);
        if (rw == PM_OK) rw = pm_flatten_and_encode(c, &wp, 1, we, 3, ident, 1.0f, nullptr, nullptr);
        if (rw == PM_OK) rw = pm_render(c);
        if (rw == PM_OK) rw = pm_sync(c);
        if (rw != PM_OK) {
            *err = rw;
            pm_destroy(c);
            return nullptr;
        }
        // back to a context without scene and viewport (the slots KEEP their buffers: the first real viewport that fits them -- slot
        // 0's are reserved for anything up to 8192 x 8192 -- allocates nothing, AllocSlotViewport)
        InvalidateScene(c);
        c->vp_epoch += 1;
        for (auto &s : c->slot) {
            s.in_flight = false;
            s.needs_check = false;
        }
        c->last_slot = -1;
        c->flatten_cache.resident = false;
        c->flatten_cache.meta_bytes = 0;
        c->width = c->height = c->tiles_x = c->tiles_y = c->strips_x = c->row0 = c->row1 = 0;
        c->frame = 0;
        c->t_flatten_ms = c->t_index_ms = c->t_arena_ms = 0;
        // ... and without the warm-up scene's arena sizes: the first real scene sizes its own (round-3 advisor
        // finding: the floor the three-point scene asked for used to stay in force for every scene after it)
        c->ptcl_want = 0;
        c->arena_cap = 0;
        if (std::getenv("PM_PTCL_INITIAL_CMDS"))  // tests of the overflow path: what pm_sync grew during the warm-up goes too
            for (auto &s : c->slot) {
                if (s.d_ptcl) (void)hipFree(s.d_ptcl);
                s.d_ptcl = nullptr;
                s.ptcl_cap = 0;
            }
    }
    *err = PM_OK;
    return c;
}

void pm_destroy(pm_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)SyncAll(c);
    FreeViewport(c);
    for (auto &s : c->slot) {
        if (s.d_arena) (void)hipFree(s.d_arena);
        if (s.d_ptcl) (void)hipFree(s.d_ptcl);
        if (s.d_row_bbox) (void)hipFree(s.d_row_bbox);  // (the indices live behind the boxes)
        if (s.d_ctr) (void)hipFree(s.d_ctr);
        if (s.h_overflow) (void)hipHostFree(s.h_overflow);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    }
    if (c->d_sr_desc) (void)hipFree(c->d_sr_desc);
    if (c->d_sr_list) (void)hipFree(c->d_sr_list);
    if (c->d_sr_slots) (void)hipFree(c->d_sr_slots);
    if (c->d_sr_desc_whole) (void)hipFree(c->d_sr_desc_whole);
    if (c->d_band_bbox) (void)hipFree(c->d_band_bbox);
    if (c->d_band_item) (void)hipFree(c->d_band_item);
    if (c->d_row_base) (void)hipFree(c->d_row_base);
    if (c->d_idle_sr) (void)hipFree(c->d_idle_sr);
    if (c->d_sr_next_one) (void)hipFree(c->d_sr_next_one);
    c->flatten_cache.Free();
    if (c->d_scene) (void)hipFree(c->d_scene);
    if (c->d_scene_alt) (void)hipFree(c->d_scene_alt);
    if (c->h_scene) (void)hipHostFree(c->h_scene);
    if (c->d_chunk_base) (void)hipFree(c->d_chunk_base);
    if (c->d_chunk_bbox) (void)hipFree(c->d_chunk_bbox);
    if (c->d_sup_bbox) (void)hipFree(c->d_sup_bbox);
    if (c->d_lut_srgb2lin) (void)hipFree(c->d_lut_srgb2lin);
    if (c->d_lut_unorm2h) (void)hipFree(c->d_lut_unorm2h);
    if (c->d_lut_lin2srgb) (void)hipFree(c->d_lut_lin2srgb);
    for (auto &ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t q : c->streams) (void)hipStreamDestroy(q);
    delete c;
}

int pm_resize(pm_ctx *c, uint32_t width, uint32_t height) {
    if (!c || width == 0 || height == 0 || width > 65535 || height > 65535) return PM_ERR_INVALID;
    const WallTimer tm;
    PM_TRY(hipSetDevice(c->device));
    const float t_a = tm.ms();
    int r = SyncAll(c);
    if (r != PM_OK) return r;
    if (std::getenv("PM_HOST_TIMING")) std::fprintf(stderr, "pm_resize: setdevice %.3f syncall %.3f ms\n", t_a, tm.ms() - t_a);
    c->width = width;
    c->height = height;
    c->tiles_x = (width + pm::kTileW - 1) / pm::kTileW;   // PietRenderer.m:63-64
    c->tiles_y = (height + pm::kTileH - 1) / pm::kTileH;
    c->strips_x = (c->tiles_x + pm::kStripTiles - 1) / pm::kStripTiles;
    c->row0 = 0;
    c->row1 = c->tiles_y;
    const float t_b = tm.ms();
    r = AllocViewport(c);
    if (std::getenv("PM_HOST_TIMING")) std::fprintf(stderr, "pm_resize: viewport buffers %.3f ms\n", tm.ms() - t_b);
    return r;
}

int pm_set_band(pm_ctx *c, uint32_t tile_row0, uint32_t tile_row1) {
    if (!c || c->tiles_y == 0 || tile_row0 >= tile_row1 || tile_row1 > c->tiles_y) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r = SyncAll(c);
    if (r != PM_OK) return r;
    c->row0 = tile_row0;
    c->row1 = tile_row1;
    return AllocViewport(c);
}

uint8_t *pm_scene_buffer(pm_ctx *c, size_t *cap) {
    if (!c) return nullptr;
    if (cap) *cap = c->scene_cap;
    return c->h_scene;
}

int pm_scene_reserve(pm_ctx *c, size_t cap) {
    if (!c) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r = SyncAll(c);
    if (r != PM_OK) return r;
    return ReserveScene(c, cap);
}

int pm_upload_scene(pm_ctx *c, size_t bytes) {
    if (!c || bytes > c->scene_cap || bytes < 8) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r = SyncAll(c);  // frames in flight still read the old scene
    if (r != PM_OK) return r;
    InvalidateScene(c);
    c->fb_slots.clear();  // (another scene: its own frames will say which strip rows are heavy -- a view change, pm_reflatten, keeps the report)
    c->fb_applied = false;
    c->replan_wide = false;
    c->plan_reusable = false;
    c->t_flatten_ms = 0;
    // (a context that goes back to host-encoded scenes gives the device flatten's second scene buffer back: round-4 advisor finding)
    if (c->d_scene_alt) {
        (void)hipFree(c->d_scene_alt);
        c->d_scene_alt = nullptr;
        c->dev_scene_alt_cap = 0;
    }
    PM_TRY(hipMemcpyAsync(c->d_scene, c->h_scene, bytes, hipMemcpyHostToDevice, c->stream));
    return SetScene(c, bytes, c->h_scene);  // (stream order: the appended flat group follows the upload)
}

namespace {
int FlattenAndEncode(pm_ctx *c, bool resident, const pm_path *paths, size_t n_paths, const pm_path_el *els, size_t n_els,
                     const double affine[6], float width_scale, size_t *scene_bytes, uint32_t *n_items);
}

int pm_flatten_and_encode(pm_ctx *c, const pm_path *paths, size_t n_paths, const pm_path_el *els, size_t n_els,
                          const double affine[6], float width_scale, size_t *scene_bytes, uint32_t *n_items) {
    if (!c || !affine || (n_paths && !paths) || (n_els && !els)) return PM_ERR_INVALID;
    c->fb_slots.clear();  // (new paths: their own frames will say which strip rows are heavy)
    c->fb_applied = false;
    c->replan_wide = false;
    c->plan_reusable = false;  // (new paths: planned afresh, for their own boxes)
    return FlattenAndEncode(c, false, paths, n_paths, els, n_els, affine, width_scale, scene_bytes, n_items);
}

namespace {
int FlattenAndEncode(pm_ctx *c, bool resident, const pm_path *paths, size_t n_paths, const pm_path_el *els, size_t n_els,
                     const double affine[6], float width_scale, size_t *scene_bytes, uint32_t *n_items) {
    PM_TRY(hipSetDevice(c->device));
    size_t bytes = 0;
    uint32_t items = 0;
    hipError_t he = hipSuccess;
    const WallTimer timer;
    // The kernels write the OTHER scene buffer: frames in flight keep reading the current one meanwhile (nothing reads
    // the other one: every scene replacement ends with all frames waited for, below).
    auto grow_alt = [&](size_t cap) -> int {
        if (cap <= c->dev_scene_alt_cap && c->d_scene_alt) return PM_OK;
        const int rc = CheckSceneCap(cap);
        if (rc != PM_OK) return rc;
        if (c->d_scene_alt) (void)hipFree(c->d_scene_alt);
        c->d_scene_alt = nullptr;
        c->dev_scene_alt_cap = 0;
        PM_TRY(hipMalloc(&c->d_scene_alt, cap));
        c->dev_scene_alt_cap = cap;
        return PM_OK;
    };
    int r = grow_alt(c->dev_scene_cap);
    if (r == PM_OK)
        r = pm::FlattenEncodeOnDevice(c->stream, &c->flatten_cache, resident, paths, n_paths, els, n_els, affine, width_scale, c->d_scene_alt,
                                      c->dev_scene_alt_cap, &bytes, &items, &he);
    if (r == PM_ERR_CAPACITY && bytes > c->dev_scene_alt_cap) {
        // grow it (the kernels write it; nothing is staged on the host) and retry once
        r = grow_alt(bytes + (bytes >> 3));
        if (r == PM_OK)
            r = pm::FlattenEncodeOnDevice(c->stream, &c->flatten_cache, resident, paths, n_paths, els, n_els, affine, width_scale, c->d_scene_alt,
                                          c->dev_scene_alt_cap, &bytes, &items, &he);
    }
    const float flatten_ms = timer.ms();
    {
        // ... and only now the frames in flight are waited for: they read the current scene buffer, and the scene index and
        // binning lists SetScene / the next frame replace.  Whatever happened above, the old scene is gone after this call.
        const int rs = SyncAll(c);
        InvalidateScene(c);
        if (rs != PM_OK) return rs;
    }
    if (r == PM_ERR_HIP) return HipFail(he, "flatten kernels");
    if (r != PM_OK) {
        if (r != PM_ERR_CAPACITY || g_last_error.empty()) SetError("flatten/encode rejected the paths");
        return r;
    }
    std::swap(c->d_scene, c->d_scene_alt);
    std::swap(c->dev_scene_cap, c->dev_scene_alt_cap);
    c->t_flatten_ms = flatten_ms;
    r = SetScene(c, bytes, nullptr);
    if (r != PM_OK) return r;
    if (scene_bytes) *scene_bytes = bytes;
    if (n_items) *n_items = items;
    return PM_OK;
}
}  // namespace

int pm_reflatten(pm_ctx *c, const double affine[6], float width_scale, size_t *scene_bytes, uint32_t *n_items) {
    if (!c || !affine) return PM_ERR_INVALID;
    if (!c->flatten_cache.resident) {
        SetError("pm_reflatten: no paths resident (pm_flatten_and_encode first)");
        return PM_ERR_INVALID;
    }
    c->replan_wide = true;  // (a view change: the next plan is made to last, EnsureArena)
    return FlattenAndEncode(c, true, nullptr, 0, nullptr, 0, affine, width_scale, scene_bytes, n_items);
}

int pm_download_scene(pm_ctx *c, uint8_t *dst, size_t cap, size_t *bytes) {
    if (!c || !dst) return PM_ERR_INVALID;
    if (bytes) *bytes = c->user_scene_bytes;
    if (cap < c->user_scene_bytes) return PM_ERR_CAPACITY;
    PM_TRY(hipSetDevice(c->device));
    PM_TRY(hipMemcpyAsync(dst, c->d_scene, c->user_scene_bytes, hipMemcpyDeviceToHost, c->stream));
    PM_TRY(hipStreamSynchronize(c->stream));
    return PM_OK;
}

int pm_render(pm_ctx *c) {
    if (!c) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    return Enqueue(c, nullptr, c->fb_stride, nullptr);
}

int pm_render_to(pm_ctx *c, void *dev_framebuffer, size_t stride_bytes, void *hip_stream) {
    if (!c || !dev_framebuffer || (stride_bytes & 3u) || stride_bytes < static_cast<size_t>(c->width) * 4) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    return Enqueue(c, static_cast<uint8_t *>(dev_framebuffer), stride_bytes, hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream);
}

// pm_sync also repairs frames whose command-list arena ran out: pm_bin_kernel marks their tiles
// "no list", the tile kernels skip them, and the frame is left with holes.  EVERY slot that
// carried a frame since the last sync is inspected (up to four frames are in flight, and
// pm_render_to frames each own a caller buffer): the arena grows and each distinct target whose
// most recent frame overflowed is rendered again, oldest first, so that after PM_OK every
// framebuffer handed to this context holds a complete frame.  (A caller that synchronises only
// its own stream never learns of an overflow: pm_sync is the status channel.)
int pm_set_target_format(pm_ctx *c, int fmt) {
    if (!c || (fmt != PM_FMT_RGBA8 && fmt != PM_FMT_BGRA8)) return PM_ERR_INVALID;
    c->target_fmt = fmt;  // (frames already submitted keep the order they were submitted with)
    return PM_OK;
}

int pm_sync(pm_ctx *c) {
    if (!c) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    auto own_fb = [&](const uint8_t *fb) {
        for (auto &t : c->slot)
            if (fb == t.d_fb) return true;
        return false;
    };
    for (int attempt = 0; attempt < 6; ++attempt) {
        // The frame that defines each target's content: per caller-owned buffer the newest frame
        // submitted since the last sync; of the context's own slot buffers only the last frame's
        // (the one pm_read_pixels / pm_framebuffer_device_ptr name).  Oldest first.
        std::vector<int> latest;
        for (size_t k = 0; k < c->slot.size(); ++k) {
            const int si = static_cast<int>((c->frame + k) % c->slot.size());  // slot c->frame % n holds the oldest frame
            const FrameSlot &s = c->slot[si];
            // (not `in_flight`: every call that waits for the device clears that -- pm_get_stats,
            //  pm_frame_latency, ... -- and a frame with holes must not slip through behind them)
            if (!s.needs_check) continue;
            if (own_fb(s.params.fb)) {
                if (si != c->last_slot) continue;
            } else {
                latest.erase(std::remove_if(latest.begin(), latest.end(), [&](int o) { return c->slot[o].params.fb == s.params.fb; }),
                             latest.end());
            }
            latest.push_back(si);
        }
        int r = SyncAll(c);
        if (r != PM_OK) return r;
        for (auto &t : c->slot) t.needs_check = false;  // (the frames left out of `latest` were superseded on their target)
        std::vector<pm::FrameParams> redo;
        uint64_t want = 0;
        bool gave_up = false;
        // A one-launch frame gave up waiting inside its launch (its workgroups were not all resident, or somebody else's work
        // held the device): the frame has holes, and its slot's hand-over state -- FIFO entries that were pushed and never
        // popped, counters -- is no longer all zero.  EVERY slot whose frame gave up is put back to all zero, also a slot whose
        // frame has since been superseded on its target and needs no second rendering (round-5 advisor: left alone, the next
        // one-launch frame of that slot could pop a stale entry); every frame from now on takes two launches.
        {
            bool repaired = false;
            for (auto &t : c->slot) {
                if (!t.h_overflow || static_cast<volatile uint32_t *>(t.h_overflow)[1] == 0u) continue;
                c->one_launch_broken = true;
                if (t.d_fifo) PM_TRY(hipMemset(t.d_fifo, 0, static_cast<size_t>(pm::kFifos) * t.fifo_cap * sizeof(uint4)));
                if (t.d_ctr) PM_TRY(hipMemset(t.d_ctr, 0, 2 * sizeof(pm::Counters)));
                repaired = true;
            }
            if (repaired) PM_TRY(hipDeviceSynchronize());  // (the frame streams do not wait for the null stream's memsets)
        }
        for (int si : latest) {
            FrameSlot &s = c->slot[si];
            if (static_cast<volatile uint32_t *>(s.h_overflow)[1] != 0u) {  // ... and the frame is rendered again with two launches
                gave_up = true;
                redo.push_back(s.params);
                continue;
            }
            // (the slot's pinned word first: no frame of the slot has run out of arena since it was last cleared -- the usual
            //  case -- and pm_sync costs no copy from the device, 15 us each)
            if (*static_cast<volatile uint32_t *>(s.h_overflow) == 0u) continue;
            uint32_t overflow = 0;
            PM_TRY(hipMemcpy(&overflow, &s.params.ctr_cur->overflow, sizeof(overflow), hipMemcpyDeviceToHost));
            if (!overflow) continue;
            // (every part of the arena has to hold what ITS strip rows asked for: size all by the fullest)
            pm::Counters k;
            PM_TRY(hipMemcpy(&k, s.params.ctr_cur, offsetof(pm::Counters, cls), hipMemcpyDeviceToHost));
            uint64_t top = 0;
            for (uint32_t i = 0; i < pm::kArenaShards; ++i) top = std::max<uint64_t>(top, k.ptcl[i].top);
            top = (top + 2) * pm::kArenaShards;
            want = std::max<uint64_t>(want, std::max<uint64_t>(4ull * s.ptcl_cap, 2ull * top));
            redo.push_back(s.params);
        }
        for (auto &t : c->slot) t.h_overflow[0] = t.h_overflow[1] = 0;  // (nothing is in flight: every frame that raised one has been looked at or superseded)
        if (redo.empty()) return PM_OK;
        want = std::min<uint64_t>(0x7fffffffull, want);
        if (want <= c->ptcl_want && !gave_up) break;
        c->ptcl_want = std::max<uint64_t>(c->ptcl_want, want);  // (every slot grows when it is used next, EnsureSlotBuffers)
        // render the damaged targets again on the context's own streams (a caller's stream may be
        // gone by now; the next pass of this loop waits for them and checks them again)
        for (const pm::FrameParams &p : redo) {
            r = Enqueue(c, own_fb(p.fb) ? nullptr : p.fb, p.fb_stride, nullptr);
            if (r != PM_OK) return r;
        }
    }
    SetError("tile arena overflow (frame needs more than 32 GiB of pieces and command lists)");
    return PM_ERR_CAPACITY;
}

int pm_read_pixels(pm_ctx *c, uint8_t *dst, size_t dst_stride, int fmt) {
    if (!c || !dst || c->last_slot < 0 || dst_stride < static_cast<size_t>(c->width) * 4) return PM_ERR_INVALID;
    const int r = pm_sync(c);
    if (r != PM_OK) return r;
    const FrameSlot *s = &c->slot[c->last_slot];
    const uint32_t rows = std::min(BandRows(c) * pm::kTileH, c->height - c->row0 * pm::kTileH);
    PM_TRY(hipMemcpy2D(dst, dst_stride, s->params.fb, s->params.fb_stride, static_cast<size_t>(c->width) * 4, rows, hipMemcpyDeviceToHost));
    if ((fmt == PM_FMT_BGRA8) != (s->params.fb_bgra != 0)) {  // the frame was stored in the other byte order
        for (uint32_t y = 0; y < rows; ++y) {
            uint8_t *row = dst + static_cast<size_t>(y) * dst_stride;
            for (uint32_t x = 0; x < c->width; ++x) std::swap(row[4 * x], row[4 * x + 2]);
        }
    }
    return PM_OK;
}

void *pm_framebuffer_device_ptr(pm_ctx *c, size_t *stride_bytes, uint32_t *rows) {
    if (!c) return nullptr;
    if (stride_bytes) *stride_bytes = c->fb_stride;
    if (rows) *rows = BandRows(c) * pm::kTileH;
    return c->slot[c->last_slot >= 0 ? c->last_slot : 0].d_fb;
}

void *pm_scene_device_ptr(pm_ctx *c, size_t *bytes) {
    if (!c) return nullptr;
    if (bytes) *bytes = c->user_scene_bytes;
    return c->d_scene;
}

int pm_time_frames(pm_ctx *c, int iters, float *total_ms, float *bin_ms, float *coarse_ms, float *fine_ms, float *clear_ms) {
    if (!c || iters <= 0) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r;
    if ((r = SyncAll(c)) != PM_OK) return r;
    if (total_ms) {
        // pipelined, as pm_render submits them: start on the binning stream, end on the tile stream
        PM_TRY(hipEventRecord(c->ev[0], c->stream));  // everything is idle: fires at once
        for (int i = 0; i < iters; ++i)
            if ((r = Enqueue(c, nullptr, c->fb_stride, nullptr)) != PM_OK) return r;
        for (auto &s : c->slot)  // join: the end event follows the last frame of every stream
            if (s.in_flight && s.frame_stream != c->stream) {
                if (!s.user_stream) PM_TRY(hipEventRecord(s.ev_done, s.frame_stream));
                PM_TRY(hipStreamWaitEvent(c->stream, s.ev_done, 0));
            }
        PM_TRY(hipEventRecord(c->ev[1], c->stream));
        PM_TRY(hipEventSynchronize(c->ev[1]));
        PM_TRY(hipEventElapsedTime(total_ms, c->ev[0], c->ev[1]));
        if ((r = SyncAll(c)) != PM_OK) return r;
    }
    if (bin_ms || coarse_ms || fine_ms || clear_ms) {
        // one kernel at a time on one stream, each launch bracketed by events
        double a1 = 0, a2 = 0, a3 = 0, a4 = 0;
        for (int i = 0; i < iters; ++i) {
            const int si = static_cast<int>(c->frame % c->slot.size());
            FrameSlot *s = &c->slot[si];
            pm::FrameParams p;
            if ((r = BuildParams(c, s, nullptr, c->fb_stride, &p)) != PM_OK) return r;
            // each dispatch carries its own begin / end events: pure kernel durations
            PM_TRY(ResetTileState(c, s, c->stream));
            pm::LaunchBin(p, c->stream, c->ev[0], c->ev[1]);
            if (!c->fused) pm::LaunchCoarse(p, CoarseGrid(c), false, c->stream, c->ev[2], c->ev[3]);
            pm::LaunchFine(p, c->fold_clear && !p.clear_in_bin ? BandRows(c) * c->strips_x : 0u, c->fused, c->stream, c->ev[4], c->ev[5]);
            if (!c->fold_clear && !p.clear_in_bin) pm::LaunchClear(p, BandRows(c) * c->strips_x, c->stream, c->ev[6], c->ev[7]);
            PM_TRY(hipStreamSynchronize(c->stream));
            Submitted(c, si, p, c->stream);
            s->in_flight = false;  // (waited for just above: the next frame is alone too, and is launched as such)
            float t1 = 0, t2 = 0, t3 = 0, t4 = 0;
            PM_TRY(hipEventElapsedTime(&t1, c->ev[0], c->ev[1]));
            if (!c->fused) PM_TRY(hipEventElapsedTime(&t2, c->ev[2], c->ev[3]));
            PM_TRY(hipEventElapsedTime(&t3, c->ev[4], c->ev[5]));
            if (!c->fold_clear && !p.clear_in_bin) PM_TRY(hipEventElapsedTime(&t4, c->ev[6], c->ev[7]));
            a1 += t1;
            a2 += t2;
            a3 += t3;
            a4 += t4;
        }
        if (bin_ms) *bin_ms = static_cast<float>(a1 / iters);
        if (coarse_ms) *coarse_ms = static_cast<float>(a2 / iters);
        if (fine_ms) *fine_ms = static_cast<float>(a3 / iters);
        if (clear_ms) *clear_ms = static_cast<float>(a4 / iters);
    }
    return pm_sync(c);
}

int pm_time_frames_pipelined(pm_ctx *c, int iters, float *total_ms, float *bin_ms, float *coarse_ms, float *fine_ms,
                             float *clear_ms) {
    if (!c || iters <= 0 || iters > 4096) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r;
    if ((r = SyncAll(c)) != PM_OK) return r;
    std::vector<hipEvent_t> tev(static_cast<size_t>(iters) * 8, nullptr);
    hipError_t e = hipSuccess;
    for (auto &v : tev)
        if (e == hipSuccess) e = hipEventCreate(&v);
    r = PM_OK;
    if (e != hipSuccess) r = HipFail(e, "hipEventCreate");
    if (r == PM_OK) e = hipEventRecord(c->ev[0], c->stream);
    // (every frame of the batch as a frame among others is launched: the per-frame decision would fold the
    //  first one's clearing into its tile kernel and leave that frame's clear events unrecorded)
    // ... as Enqueue launches frames behind other frames: clearing in a launch of its own for large viewports, folded into
    // the tile kernel's for small ones (round-3 advisor finding: the batch used to be timed unfolded at every size)
    const int fold_mode = c->fold_clear_mode;
    if (fold_mode == 2) c->fold_clear_mode = BandTiles(c) < 16384u ? 1 : 0;
    const bool folded = c->fold_clear_mode == 1 || c->fold_clear_mode >= 3;  // (no clearing launch of its own: folded into the tile kernel's, or into binning's)
    for (int i = 0; i < iters && r == PM_OK; ++i) r = Enqueue(c, nullptr, c->fb_stride, nullptr, &tev[static_cast<size_t>(i) * 8]);
    c->fold_clear_mode = fold_mode;
    if (r == PM_OK) {
        for (auto &s : c->slot) {
            if (!s.in_flight || s.frame_stream == c->stream) continue;
            if (e == hipSuccess && !s.user_stream) e = hipEventRecord(s.ev_done, s.frame_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, s.ev_done, 0);
        }
        if (e == hipSuccess) e = hipEventRecord(c->ev[1], c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev[1]);
        if (e == hipSuccess) r = SyncAll(c);
        double acc[4] = {0, 0, 0, 0};  // bin, clear, coarse, fine
        for (int i = 0; i < iters && e == hipSuccess && r == PM_OK; ++i)
            for (int k = 0; k < 4 && e == hipSuccess; ++k) {
                if (k == 1 && folded) continue;  // no separate clear launch
                if (k == 2 && c->fused) continue;       // no separate coarse launch
                float t = 0;
                e = hipEventElapsedTime(&t, tev[static_cast<size_t>(i) * 8 + 2 * k], tev[static_cast<size_t>(i) * 8 + 2 * k + 1]);
                acc[k] += t;
            }
        float tt = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&tt, c->ev[0], c->ev[1]);
        if (e != hipSuccess) r = HipFail(e, "pipelined timing");
        if (total_ms) *total_ms = tt;
        if (bin_ms) *bin_ms = static_cast<float>(acc[0] / iters);
        if (clear_ms) *clear_ms = static_cast<float>(acc[1] / iters);
        if (coarse_ms) *coarse_ms = static_cast<float>(acc[2] / iters);
        if (fine_ms) *fine_ms = static_cast<float>(acc[3] / iters);
    }
    (void)SyncAll(c);
    for (auto &v : tev)
        if (v) (void)hipEventDestroy(v);
    return r == PM_OK ? pm_sync(c) : r;
}

int pm_frame_latency(pm_ctx *c, int iters, float *median_ms, float *min_ms) {
    if (!c || iters <= 0 || iters > 4096) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r;
    if ((r = SyncAll(c)) != PM_OK) return r;
    std::vector<float> lat;
    for (int i = 0; i < iters; ++i) {
        // One frame exactly as pm_render submits it (three plain launches), nothing else in
        // flight, bracketed by two events on its stream: first kernel begin to last kernel end
        // (SURVEY 8d's t_frame).  Dispatches that carry their own timestamps
        // (pm_debug_frame_timeline) show the same kernels but stretch every gap between them
        // by ~3.5 us, so they are kept for the breakdown only.
        hipStream_t q = c->streams[c->frame % c->streams.size()];
        PM_TRY(hipEventRecord(c->ev[0], q));
        if ((r = Enqueue(c, nullptr, c->fb_stride, nullptr)) != PM_OK) return r;
        PM_TRY(hipEventRecord(c->ev[1], q));
        if ((r = SyncAll(c)) != PM_OK) return r;
        float t = 0;
        PM_TRY(hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
        lat.push_back(t);
    }
    std::sort(lat.begin(), lat.end());
    if (median_ms) *median_ms = lat[lat.size() / 2];
    if (min_ms) *min_ms = lat.front();
    return pm_sync(c);
}

int pm_binning_info(pm_ctx *c, uint32_t out[3]) {
    if (!c || !out) return PM_ERR_INVALID;
    out[0] = c->frames_bin_wave;
    out[1] = c->frames_bin_wave_inflight;
    out[2] = c->frames_bin_no_chains;
    return PM_OK;
}

int pm_binning_plan_info(pm_ctx *c, uint32_t out[4]) {
    if (!c || !out) return PM_ERR_INVALID;
    out[0] = c->n_sr_active;
    out[1] = c->n_sr_split;
    out[2] = c->plans_fed_back;
    out[3] = c->fb_applied ? 1u : 0u;
    return PM_OK;
}

int pm_tile_kernel_info(pm_ctx *c, uint32_t *dense_frames) {
    if (!c || !dense_frames) return PM_ERR_INVALID;
    *dense_frames = c->frames_dense_kernel;
    return PM_OK;
}

int pm_one_launch_info(pm_ctx *c, uint32_t *frames, int *applies) {
    if (!c) return PM_ERR_INVALID;
    if (frames) *frames = c->frames_one_launch;
    if (applies) {
        *applies = 0;
        if (c->d_scene && c->tiles_x != 0) {
            const int r = EnsureArena(c);
            if (r != PM_OK) return r;
            *applies = OneLaunchGrid(c) != 0u && !c->one_launch_broken && c->handout != 1 ? 1 : 0;
        }
    }
    return PM_OK;
}

int pm_time_one_launch(pm_ctx *c, int iters, float *kernel_ms) {
    if (!c || iters <= 0 || !kernel_ms) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r;
    if ((r = SyncAll(c)) != PM_OK) return r;
    double acc = 0;
    for (int i = 0; i < iters; ++i) {
        const int si = static_cast<int>(c->frame % c->slot.size());
        FrameSlot *s = &c->slot[si];
        pm::FrameParams p;
        if ((r = BuildParams(c, s, nullptr, c->fb_stride, &p)) != PM_OK) return r;
        const uint32_t grid = OneLaunchGrid(c);
        if (grid == 0u || c->one_launch_broken) return PM_ERR_INVALID;
        if ((r = EnsureFifo(c, s, c->stream)) != PM_OK) return r;
        p.fifo = s->d_fifo;
        p.fifo_cap = s->fifo_cap;
        p.one_launch = 3u;
        PM_TRY(ResetTileState(c, s, c->stream));
        pm::LaunchFrame(p, grid, c->stream, c->ev[0], c->ev[1]);
        PM_TRY(hipStreamSynchronize(c->stream));
        Submitted(c, si, p, c->stream);
        s->in_flight = false;
        float t = 0;
        PM_TRY(hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
        acc += t;
    }
    *kernel_ms = static_cast<float>(acc / iters);
    return pm_sync(c);
}

int pm_debug_time_frame(pm_ctx *c, uint64_t *out, size_t max_wgs, size_t *n_wgs) {
    if (!c || !out) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r;
    if ((r = SyncAll(c)) != PM_OK) return r;
    const int si = static_cast<int>(c->frame % c->slot.size());
    FrameSlot *s = &c->slot[si];
    pm::FrameParams p;
    if ((r = BuildParams(c, s, nullptr, c->fb_stride, &p)) != PM_OK) return r;
    const uint32_t grid = OneLaunchGrid(c);
    if (n_wgs) *n_wgs = grid;
    if (grid == 0u || c->one_launch_broken) return PM_ERR_INVALID;
    if (grid > max_wgs) return PM_ERR_CAPACITY;
    if ((r = EnsureFifo(c, s, c->stream)) != PM_OK) return r;
    unsigned long long *d = nullptr;
    PM_TRY(hipMalloc(&d, static_cast<size_t>(grid) * 32 * sizeof(unsigned long long)));
    p.fifo = s->d_fifo;
    p.fifo_cap = s->fifo_cap;
    p.one_launch = 3u;
    p.dbg_time = d;
    hipError_t e = ResetTileState(c, s, c->stream);
    if (e == hipSuccess) {
        if (c->one_launch_split) {
            p.one_launch = 1u;
            pm::LaunchFrame(p, grid, c->stream);
            p.one_launch = 2u;
        }
        pm::LaunchFrame(p, grid, c->stream);
        e = hipStreamSynchronize(c->stream);
    }
    p.dbg_time = nullptr;
    Submitted(c, si, p, c->stream);
    s->in_flight = false;
    if (e == hipSuccess) e = hipMemcpy(out, d, static_cast<size_t>(grid) * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return HipFail(e, "frame timeline");
    return pm_sync(c);
}

int pm_debug_frame_timeline(pm_ctx *c, int iters, float *out6) {
    if (!c || !out6 || iters <= 0 || iters > 4096) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r;
    if ((r = SyncAll(c)) != PM_OK) return r;
    hipEvent_t tev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (auto &v : tev)
        if (e == hipSuccess) e = hipEventCreate(&v);
    std::vector<float> col[6];
    r = e == hipSuccess ? PM_OK : HipFail(e, "hipEventCreate");
    for (int i = 0; i < iters && r == PM_OK; ++i) {
        r = Enqueue(c, nullptr, c->fb_stride, nullptr, tev);
        if (r == PM_OK) r = SyncAll(c);
        float v[6] = {0, 0, 0, 0, 0, 0};
        bool ok = r == PM_OK && hipEventElapsedTime(&v[0], tev[0], tev[1]) == hipSuccess &&
                  hipEventElapsedTime(&v[4], tev[6], tev[7]) == hipSuccess && hipEventElapsedTime(&v[5], tev[0], tev[7]) == hipSuccess;
        if (ok && c->fused)  // bin, gap, (no coarse launch), fine
            ok = hipEventElapsedTime(&v[1], tev[1], tev[6]) == hipSuccess;
        else if (ok)
            ok = hipEventElapsedTime(&v[1], tev[1], tev[4]) == hipSuccess && hipEventElapsedTime(&v[2], tev[4], tev[5]) == hipSuccess &&
                 hipEventElapsedTime(&v[3], tev[5], tev[6]) == hipSuccess;
        if (ok)
            for (int k = 0; k < 6; ++k) col[k].push_back(v[k]);
    }
    for (auto &v : tev)
        if (v) (void)hipEventDestroy(v);
    if (r != PM_OK) return r;
    if (col[0].empty()) return PM_ERR_HIP;
    for (int k = 0; k < 6; ++k) {
        std::sort(col[k].begin(), col[k].end());
        out6[k] = col[k][col[k].size() / 2];
    }
    return pm_sync(c);
}

int pm_get_stats(pm_ctx *c, pm_stats *out) {
    if (!c || !out) return PM_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    PM_TRY(hipSetDevice(c->device));
    int r = SyncAll(c);
    if (r != PM_OK) return r;
    out->tiles_x = c->tiles_x;
    out->tiles_y = c->tiles_y;
    out->band_row0 = c->row0;
    out->band_row1 = c->row1;
    out->n_items = c->n_items;
    out->arena_cap_dwords = c->arena_cap;
    out->scene_bytes = static_cast<uint32_t>(c->user_scene_bytes);
    if (c->last_slot >= 0) {
        pm::Counters k;
        PM_TRY(hipMemcpy(&k, c->slot[c->last_slot].params.ctr_cur, sizeof(k), hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < pm::kClasses; ++q) out->queued_tiles += k.cls[q].count;
        for (uint32_t q = 0; q < 3; ++q) out->heavy_tiles += k.cls[q].count;  // (n_heavy_classes)
        out->arena_used_dwords = 0;
        for (uint32_t i = 0; i < pm::kArenaShards; ++i) out->arena_used_dwords += k.ptcl[i].bin_dwords;
        out->ptcl_used_cmds = 0;
        for (uint32_t i = 0; i < pm::kArenaShards; ++i) out->ptcl_used_cmds += k.ptcl[i].top;
        out->overflow = k.overflow;
    }
    return PM_OK;
}

int pm_get_scene_timings(pm_ctx *c, pm_scene_timings *out) {
    if (!c || !out) return PM_ERR_INVALID;
    out->flatten_encode_ms = c->t_flatten_ms;
    out->scene_index_ms = c->t_index_ms;
    out->arena_setup_ms = c->t_arena_ms;
    return PM_OK;
}

int pm_get_binning_plans(pm_ctx *c, uint32_t *plans) {
    if (!c || !plans) return PM_ERR_INVALID;
    *plans = c->plans_made;
    return PM_OK;
}

int pm_fill_coverage(pm_ctx *c, uint32_t item_ix, float *dst, size_t dst_stride_floats) {
    if (!c || !dst || dst_stride_floats < c->width) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r = pm_sync(c);
    if (r != PM_OK) return r;
    if (c->scene_bytes < 8 || item_ix >= c->n_items || c->tiles_x == 0) {
        SetError("pm_fill_coverage: no such item (or no scene / viewport)");
        return PM_ERR_INVALID;
    }
    const uint8_t *meta = c->item_meta.data();  // normal form: {n, 8 + 8n}{boxes}{items}
    const uint8_t *item = meta + 8 + 8ull * c->n_items + 32ull * item_ix;
    uint32_t tag;
    std::memcpy(&tag, item, 4);
    if ((tag & 0xffffu) != pm::kItemFill) {
        SetError("pm_fill_coverage: the item is not a Fill");
        return PM_ERR_INVALID;
    }
    // the item alone as a one-item group behind the resident bytes (its points stay where they are)
    const size_t root = (c->scene_bytes + 7u) & ~static_cast<size_t>(7u);
    uint8_t mini[8 + 8 + 32];
    const uint32_t hdr[2] = {1u, static_cast<uint32_t>(root + 16)};
    std::memcpy(mini, hdr, 8);
    std::memcpy(mini + 8, meta + 8 + 8ull * item_ix, 8);
    std::memcpy(mini + 16, item, 32);
    if (root + sizeof(mini) > c->dev_scene_cap) {
        r = ReserveDevice(c, root + sizeof(mini) + 4096, c->scene_bytes);
        if (r != PM_OK) return r;
    }
    PM_TRY(hipMemcpy(c->d_scene + root, mini, sizeof(mini), hipMemcpyHostToDevice));
    // swap the drawn group for the mini group, render bin + coarse (+ recorded solid colours),
    // run the coverage kernel, put everything back
    std::vector<uint8_t> saved_meta;
    saved_meta.swap(c->item_meta);
    const uint32_t saved_n = c->n_items, saved_bbox = c->dev_bbox_ix, saved_items = c->dev_items_ix;
    c->item_meta.assign(8 + 40, 0);
    const uint32_t mhdr[2] = {1u, 16u};
    std::memcpy(c->item_meta.data(), mhdr, 8);
    std::memcpy(c->item_meta.data() + 8, mini + 8, 40);
    c->n_items = 1;
    c->dev_bbox_ix = static_cast<uint32_t>(root + 8);
    c->dev_items_ix = static_cast<uint32_t>(root + 16);
    c->arena_dirty = true;
    const size_t tiles = static_cast<size_t>(BandRows(c)) * c->tiles_x;
    const size_t rows_px = std::min<size_t>(static_cast<size_t>(BandRows(c)) * pm::kTileH, c->height - c->row0 * pm::kTileH);
    uint32_t *d_counts = nullptr, *d_solid = nullptr;
    float *d_out = nullptr;
    int status = BuildSceneIndex(c);
    hipError_t e = hipSuccess;
    if (status == PM_OK) {
        e = hipMalloc(&d_counts, std::max<size_t>(tiles, 1) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&d_solid, std::max<size_t>(tiles, 1) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&d_out, static_cast<size_t>(BandRows(c)) * pm::kTileH * c->width * sizeof(float));
        if (e == hipSuccess) e = hipMemset(d_solid, 0, std::max<size_t>(tiles, 1) * sizeof(uint32_t));
    }
    if (status == PM_OK && e == hipSuccess) {
        const int si = static_cast<int>(c->frame % c->slot.size());
        FrameSlot *s = &c->slot[si];
        pm::FrameParams p;
        status = BuildParams(c, s, nullptr, c->fb_stride, &p);
        if (status == PM_OK) {
            p.dbg_counts = d_counts;
            p.dbg_solid = d_solid;
            p.dbg_cmds = nullptr;
            p.dbg_max = 0;
            (void)ResetTileState(c, s, c->stream);
            pm::LaunchBin(p, c->stream);
            pm::LaunchCoarse(p, CoarseGrid(c), true, c->stream);
            pm::LaunchCoverage(p, static_cast<uint32_t>(tiles), d_solid, d_out, c->width, c->stream);
            p.dbg_counts = p.dbg_solid = nullptr;
            Submitted(c, si, p, c->stream);
            e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess)
                e = hipMemcpy2D(dst, dst_stride_floats * sizeof(float), d_out, static_cast<size_t>(c->width) * sizeof(float),
                                static_cast<size_t>(c->width) * sizeof(float), rows_px, hipMemcpyDeviceToHost);
            pm::Counters k;
            if (e == hipSuccess) e = hipMemcpy(&k, p.ctr_cur, sizeof(k), hipMemcpyDeviceToHost);
            if (e == hipSuccess && k.overflow) {
                SetError("pm_fill_coverage: command-list arena overflow");
                status = PM_ERR_CAPACITY;
            }
        }
    }
    if (d_counts) (void)hipFree(d_counts);
    if (d_solid) (void)hipFree(d_solid);
    if (d_out) (void)hipFree(d_out);
    // restore the scene
    c->item_meta.swap(saved_meta);
    c->n_items = saved_n;
    c->dev_bbox_ix = saved_bbox;
    c->dev_items_ix = saved_items;
    c->arena_dirty = true;
    c->last_slot = -1;
    const int rb = BuildSceneIndex(c);
    if (e != hipSuccess) return HipFail(e, "pm_fill_coverage");
    return status != PM_OK ? status : rb;
}

int pm_debug_capture_ptcl(pm_ctx *c, uint32_t max_cmds_per_tile, uint32_t *counts, uint32_t *solid, pm_cmd *cmds) {
    if (!c || !counts || !solid || (max_cmds_per_tile && !cmds)) return PM_ERR_INVALID;
    if (c->last_slot < 0) {
        SetError("pm_debug_capture_ptcl needs a rendered frame");
        return PM_ERR_INVALID;
    }
    PM_TRY(hipSetDevice(c->device));
    int r = pm_sync(c);
    if (r != PM_OK) return r;
    FrameSlot *s = &c->slot[c->last_slot];
    const size_t tiles = static_cast<size_t>(BandRows(c)) * c->tiles_x;
    uint32_t *d_counts = nullptr, *d_solid = nullptr;
    pm::Cmd *d_cmds = nullptr;
    PM_TRY(hipMalloc(&d_counts, std::max<size_t>(tiles, 1) * sizeof(uint32_t)));
    PM_TRY(hipMalloc(&d_solid, std::max<size_t>(tiles, 1) * sizeof(uint32_t)));
    PM_TRY(hipMalloc(&d_cmds, std::max<size_t>(tiles * max_cmds_per_tile, 1) * sizeof(pm::Cmd)));
    // tiles the binning kernel resolved itself never reach the tile kernels: {Bail} + its colour
    std::vector<uint32_t> h_counts(tiles, 1u), h_solid(tiles, 0xffffffffu);
    std::vector<pm::Cmd> h_cmds(tiles * max_cmds_per_tile);
    if (tiles) PM_TRY(hipMemcpy(h_solid.data(), s->d_tile_state, tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t t = 0; t < tiles && max_cmds_per_tile; ++t) {
        h_cmds[t * max_cmds_per_tile].tag = pm::kCmdBail;
        std::memset(h_cmds[t * max_cmds_per_tile].body, 0, sizeof(h_cmds[0].body));
    }
    int status = PM_OK;
    hipError_t e = hipMemcpy(d_counts, h_counts.data(), tiles * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_solid, h_solid.data(), tiles * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && max_cmds_per_tile) e = hipMemcpy(d_cmds, h_cmds.data(), h_cmds.size() * sizeof(pm::Cmd), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        pm::FrameParams p = s->params;  // same arena / queues / counters as the last frame
        if (p.one_launch != 0u) {
            // the last frame was ONE launch: no queues to replay -- the frame is rendered again (same target, same bytes) by the
            // capture instantiation of that kernel
            c->cap_counts = d_counts;
            c->cap_solid = d_solid;
            c->cap_cmds = d_cmds;
            c->cap_max = max_cmds_per_tile;
            bool own = false;
            for (auto &t : c->slot) own = own || p.fb == t.d_fb;
            status = Enqueue(c, own ? nullptr : p.fb, p.fb_stride, nullptr);
            c->cap_counts = c->cap_solid = nullptr;
            c->cap_cmds = nullptr;
            c->cap_max = 0;
            if (status == PM_OK) status = pm_sync(c);
            if (status == PM_OK && c->slot[c->last_slot].params.dbg_counts == nullptr) {
                SetError("pm_debug_capture_ptcl: the frame could not be rendered again as one launch");
                status = PM_ERR_INVALID;
            }
            if (status != PM_OK) e = hipErrorUnknown;
        } else {
        p.dbg_counts = d_counts;
        p.dbg_solid = d_solid;
        p.dbg_cmds = d_cmds;
        p.dbg_max = max_cmds_per_tile;
        if (c->fused) {
            // the frame path's own kernel, capture switched on (it rebuilds the same lists and renders
            // the same pixels: idempotent); the frame's hand-out counters are spent: deal again
            e = hipMemsetAsync(&p.ctr_cur->ticket, 0, sizeof(p.ctr_cur->ticket), c->stream);
            pm::LaunchFine(p, 0u, true, c->stream);
        } else {
            pm::LaunchCoarse(p, CoarseGrid(c), true, c->stream);  // replays the last frame's queues
        }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(counts, d_counts, tiles * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(solid, d_solid, tiles * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && max_cmds_per_tile) e = hipMemcpy(cmds, d_cmds, tiles * max_cmds_per_tile * sizeof(pm::Cmd), hipMemcpyDeviceToHost);
    if (e != hipSuccess && status == PM_OK) status = HipFail(e, "ptcl capture");
    (void)hipFree(d_counts);
    (void)hipFree(d_solid);
    (void)hipFree(d_cmds);
    return status;
}

int pm_debug_time_bins(pm_ctx *c, uint64_t *out, size_t max_rows, size_t *n_rows) {
    if (!c || !out) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r = SyncAll(c);
    if (r != PM_OK) return r;
    r = EnsureArena(c);  // (the work list: one row of the timeline per entry, in the list's order)
    if (r != PM_OK) return r;
    const size_t rows = std::max<size_t>(static_cast<size_t>(BandRows(c)) * c->strips_x, c->n_sr_active);
    if (n_rows) *n_rows = rows;
    if (rows > max_rows) return PM_ERR_CAPACITY;
    unsigned long long *d = nullptr;
    PM_TRY(hipMalloc(&d, std::max<size_t>(rows, 1) * 16 * sizeof(unsigned long long)));
    PM_TRY(hipMemset(d, 0, std::max<size_t>(rows, 1) * 16 * sizeof(unsigned long long)));
    const int si = static_cast<int>(c->frame % c->slot.size());
    FrameSlot *s = &c->slot[si];
    pm::FrameParams p;
    r = BuildParams(c, s, nullptr, c->fb_stride, &p);
    if (r == PM_OK) {
        p.dbg_bin = d;
        hipError_t e = ResetTileState(c, s, c->stream);
        pm::LaunchBin(p, c->stream);
        if (!c->fold_clear && !p.clear_in_bin) pm::LaunchClear(p, BandRows(c) * c->strips_x, c->stream);
        if (!c->fused) pm::LaunchCoarse(p, CoarseGrid(c), false, c->stream);
        pm::LaunchFine(p, c->fold_clear && !p.clear_in_bin ? BandRows(c) * c->strips_x : 0u, c->fused, c->stream);
        p.dbg_bin = nullptr;
        Submitted(c, si, p, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(out, d, rows * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (e != hipSuccess) r = HipFail(e, "bin timeline");
    }
    (void)hipFree(d);
    return r;
}

int pm_debug_time_tiles(pm_ctx *c, uint64_t *out, size_t max_slots, size_t *n_slots) {
    if (!c || !out || c->last_slot < 0) return PM_ERR_INVALID;
    PM_TRY(hipSetDevice(c->device));
    int r = pm_sync(c);
    if (r != PM_OK) return r;
    FrameSlot *s = &c->slot[c->last_slot];
    pm::Counters k;
    PM_TRY(hipMemcpy(&k, s->params.ctr_cur, sizeof(k), hipMemcpyDeviceToHost));
    // as pm_fine_kernel decides
    size_t heavy = 0, total = 0;
    for (uint32_t q = 0; q < pm::kClasses; ++q) {
        total += k.cls[q].count;
        if (q < 3) heavy += k.cls[q].count;
    }
    const bool dense = heavy * c->dense_factor >= static_cast<size_t>(s->params.fine_grid) * 4u || c->split_mode == 0;
    const size_t slots = dense ? total : 4 * heavy + (total - heavy);
    if (n_slots) *n_slots = slots;
    if (slots > max_slots) return PM_ERR_CAPACITY;
    unsigned long long *d = nullptr;
    PM_TRY(hipMalloc(&d, std::max<size_t>(slots, 1) * 12 * sizeof(unsigned long long)));
    pm::FrameParams p = s->params;
    p.dbg_time = d;
    // the frame's hand-out counters are spent: deal again
    PM_TRY(hipMemsetAsync(&p.ctr_cur->ticket, 0, sizeof(p.ctr_cur->ticket), c->stream));
    pm::LaunchFine(p, 0u, c->fused, c->stream);  // (the fused kernel rebuilds the same lists: idempotent)
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d, slots * 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return HipFail(e, "tile timeline");
    return PM_OK;
}

// Drop-in for include/piet_metal.h:3 / src/lib.rs:387-393 (make_test_scene -> make_tiger).
void init_test_scene(uint8_t *buf, ssize_t buf_size) {
    if (!buf || buf_size <= 0) {
        SetError("init_test_scene: bad buffer");
        return;
    }
    int err = PM_OK;
    pm_ctx *c = pm_create(0, &err);
    if (!c) return;
    pm_svg *svg = pm_svg_tiger(0, &err);
    if (svg) {
        const double scale = 8.0;  // src/lib.rs:287
        const double affine[6] = {scale, 0.0, 0.0, scale, 0.0, 0.0};
        size_t bytes = 0;
        uint32_t items = 0;
        err = pm_flatten_and_encode(c, pm_svg_paths(svg), pm_svg_n_paths(svg), pm_svg_els(svg), pm_svg_n_els(svg), affine,
                                    static_cast<float>(scale), &bytes, &items);
        if (err == PM_OK) {
            if (bytes > static_cast<size_t>(buf_size)) SetError("init_test_scene: buffer too small for the Tiger scene");
            else (void)pm_download_scene(c, buf, static_cast<size_t>(buf_size), &bytes);
        }
        pm_svg_free(svg);
    }
    pm_destroy(c);
}

}  // extern "C"
