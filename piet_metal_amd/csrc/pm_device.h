// Device-side frame description shared by the host context and the kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "pm_layout.h"

namespace pm {

// Tiles with something to draw are queued in kClasses cost classes by the number of stream
// elements binning counted for them (class 0 = longest lists).  The tile kernels hand the
// queue slots out statically, longest class first, in snake order over their persistent waves:
// with fine classes that is a longest-processing-time-first schedule, every wave ends up with
// one long and one short tile instead of whatever the strip rows' atomics happened to interleave.
constexpr uint32_t kClasses = 8;

// Decks of the drawn part of the tile hand-out (pm_fine_kernel).  A deck is a counter on its own cache line; a wave draws
// from deck wave % n only, so a deck's cards run out when ITS waves have drawn them: with 128 decks of 40 waves the decks
// emptied 1.5 us apart and the last cards were drawn late (same-box A/B, `tools/gpu.sh ab`: 32 decks end the 4K Tiger's tile
// kernel 0.4 us earlier than 128, 16 are no different from 32, 8 are slower -- the line then serves 60 draws per us).
constexpr uint32_t kTicketParts = 32;  // decks of the drawn part of the tile hand-out (pm_fine_kernel)

// Per-frame counters; two copies alternate between frames so that frame N's binning
// kernel can reset frame N+1's copy (no memset launch on the critical path).
constexpr uint32_t kArenaShards = 16;  // parts of the tile arena, each with its own allocation counter
constexpr uint32_t kFifoShards = 8;    // one-launch frames: FIFOs of single-wave tiles, one per XCD
constexpr uint32_t kFifos = 1 + kFifoShards;

// A FIFO's counters (its entries live in FrameParams::fifo): both on one 128-byte line -- a waiting wave reads them with one load.
struct Fifo {
    uint32_t tail;  // entries pushed (reserved: an entry is in place when its words are non-zero)
    uint32_t head;  // tickets handed out
    uint32_t pad[30];
};

struct Counters {
    // Every strip row of a frame adds to these with returning atomics.  The L2 executes
    // same-cache-line atomics one after the other (~90 per us measured), so each hot counter
    // lives on its own 128-byte line.
    // The tile arena is cut into kArenaShards equal parts, each with its own bump pointer (quads used of
    // the part): strip row r allocates from part r % kArenaShards -- pieces per record, lists per row, 2 200
    // returning atomics per 4K Tiger frame that, on ONE line, took 25 us of L2 time to serve (round 3 found
    // binning bound by exactly that: 35 -> 30 us).
    struct {
        uint32_t top;
        uint32_t bin_dwords;  // dwords of binning records written by the part's strip rows (statistics)
        uint32_t pad[30];
    } ptcl[kArenaShards];
    struct {
        uint32_t count;  // tiles queued in this class
        uint32_t pad[31];
    } cls[kClasses];
    uint32_t overflow;   // set if the command-list arena ran out
    uint32_t pad4[31];
    struct {
        uint32_t count;  // cards drawn from this deck of tiles (pm_fine_kernel's hand-out)
        uint32_t pad[31];
    } ticket[kTicketParts];
    // One launch per frame (pm_frame_kernel): tiles a strip row's own workgroup does not render itself wait in FIFOs for whoever
    // is free -- fifo[0] the tiles a whole workgroup renders, fifo[1 + s] the single-wave tiles pushed by the workgroups of XCD s
    // (block b runs on XCD b % 8).  Pushes reserve places with ONE returning atomic per strip row and FIFO, pops take a ticket each.
    Fifo fifo[kFifos];
    struct {
        uint32_t count;  // workgroups of this part (blockIdx.x % kFifoShards) that have handed their strip row's tiles over
        uint32_t pad[31];
    } done_part[kFifoShards];
    struct {
        uint32_t parts;  // parts whose workgroups are all through
        uint32_t done;   // 1: every strip row has been binned and handed over -- the FIFOs' tails are final
        uint32_t pad[30];
    } done_top;
};

// Binning works in two address spaces of HBM (per frame slot):
//
// * the BINNING ARENA: every strip row owns a private region [sr_desc.y, sr_desc.z) sized by the
//   host for the worst case, so pm_bin_kernel allocates with plain arithmetic (no atomics, no
//   counting pass).  A record of one (strip row, batch of <= 256 candidate items) is the
//   intermediate of the vote pass, read back by the same workgroup and by nobody else:
//     segs: 16 B slots (start.xy, end.xy); surviving chunk s owns slots [s*kChunkSegs, (s+1)*kChunkSegs),
//         one per segment of the chunk, paint order; slots of segments that lost the vote stay unused
//     meta: one word per slot { tiles of the strip where the segment can emit | candidate << 16 |
//         voted << 31 }, 0 for an unused slot
//   The meta words of a strip row's records grow from the front of its region, the segs from its back
//   (slot f of a record at back - 1 - f, in 16-byte units): neither needs the record's size before its
//   chunks are tested.
//
// * the TILE ARENA (16-byte "quads", bump-allocated with one atomic per record and one per strip
//   row; grown by pm_sync when it runs out): what the tile kernel reads and writes.
//     piece -- everything ONE tile needs from ONE record, contiguous, paint order:
//         quad 0            { next piece of the tile (quad index, 0 = none), its candidates | segments << 9, 0, 0 }
//         1 quad per relevant segment (start.xy, end.xy), the candidates' segments back to back
//         2 quads per candidate that can emit a command in this tile:
//                           { tag, rgba, aux0, aux1 } { backdrop << 20 | relevant segments, item index, rg, ba }
//             aux0/aux1 = bbox words (circle), width bits (line, polyline) or PietFill.flags (fill);
//             rg/ba = the colour already through unpack_unorm4x8_srgb_to_half (4 x binary16);
//             backdrop = the reference's per-tile left-ray winding sum (PietRender.metal:326-333)
//             over all voted segments of the item, done once in pm_bin_kernel
//         (the candidates come last: the piece is reserved from the bbox count of candidates before
//          binning knows how many of them really emit here; the slack stays unused behind them)
//       One load of the queue entry, then one batch of loads (candidates + segments) starts a tile:
//       list building is a compute pass over a contiguous run, not a scan of the strip row's record.
//     command list -- the tile's 24-byte Cmd records (TestApp/GenTypes.h:430-495), space claimed from
//       the estimate (3 x stream elements + 1).
//
// Tile queues: kClasses class queues of 16-byte entries {tile (column | row of the band << 16), first quad of the command list,
// first piece, candidates | segments << kPieceHitBits of that piece}; the tile kernels walk them
// statically, longest first, so the expensive tiles start first and the cheap ones fill the tail.
//
// Scene index (built once per scene upload by pm_index_kernel, like the ShortBbox
// array the encoder builds at encode time): segments are grouped in chunks of kChunkSegs
// consecutive segments of one Fill / StrokePolyLine item; chunk_bbox holds each
// chunk's float bounding box {xmin, ymin, xmax, ymax}; chunk_base[i] is the first
// chunk of item i (chunk_base[n_items] = total).  The binning kernel streams only
// the chunks whose box can reach its strip row.
// Two levels (round 4): chunks of kChunkSegs = 4 segments, and SUPER-CHUNKS of kSuperChunks = 8 consecutive
// entries of the (global) chunk table -- sup_bbox[g] is the union of chunk_bbox[8 g .. 8 g + 7], whatever
// items those chunks belong to.  A strip row first tests the supers its candidates' chunk ranges touch, then
// only the chunks of the surviving supers: at the 4K Tiger the heaviest strip rows went from 1 770 chunk tests
// and 1 072 segment slots (chunks of 8, one level) to 460 + 700 tests and 530 slots.
constexpr uint32_t kChunkSegs = 4;
constexpr uint32_t kSuperChunks = 8;
constexpr uint32_t kArenaBase = 4;     // offset 0 means "none"
constexpr int kBinWaves = 4;           // waves of one binning workgroup
constexpr uint32_t kRowCullPer = 8;    // pm_rowcull_kernel: items tested per lane and step
constexpr uint32_t kRowCullStep = 64u * kBinWaves * kRowCullPer;  // ... per workgroup and step
constexpr uint32_t kCtShift = 20;            // per (candidate, tile): backdrop << 20 | relevant-segment count
constexpr uint32_t kCtCountMask = (1u << kCtShift) - 1u;
constexpr uint32_t kSlotDwords = 5;          // binning arena: 16 B segment + 4 B meta word per slot
constexpr uint32_t kPieceHitBits = 9;        // queue entry / piece header: candidates (<= 256 per record) | segments << 9
constexpr uint32_t kPieceHitMask = (1u << kPieceHitBits) - 1u;
constexpr uint32_t kCmdQuadsNum = 3, kCmdQuadsDen = 2;  // a 24-byte Cmd = 1.5 quads

struct FrameParams {
    const uint8_t *scene;
    uint32_t scene_bytes;
    uint32_t n_items, items_ix;  // the group the frame draws (the scene header, or the flat form of nested groups), validated on the host
    uint32_t bbox_ix;            // its ShortBbox array (8 for the scene header)
    uint32_t width, height;
    uint32_t tiles_x, tiles_y;
    uint32_t row0, row1;  // band of tile rows rendered by this context
    uint32_t strips_x;
    uint8_t *fb;          // band framebuffer, RGBA8, row 0 = pixel row row0*16
    uint32_t fb_stride;
    uint32_t fb_vec16;    // 1 if fb and stride are 16-byte aligned
    uint32_t fb_bgra;     // 1: pixels are stored B,G,R,A (MTLPixelFormatBGRA8Unorm, PietRenderer.m:29) instead of R,G,B,A
    uint32_t *arena;
    uint32_t arena_cap;   // dwords
    const uint4 *sr_desc;     // [n_sr_active] {strip | tile row of the band << 16, its private arena region begin, end, next entry of the workgroup's chain}
    const uint2 *sr_list;     // [n_sr_active] large scenes: {first entry, entries} of the strip row's tile row in row_bbox / row_item (what row_base says, next to the descriptor: one round trip less)
    uint32_t n_sr_active;     // strip rows some item reaches: pm_bin_kernel's work list
    uint32_t *sr_slots;       // [n_sr_active] what every entry of the work list found: its segment slots (pm_bin_kernel leaves them; the host cuts the heaviest strip rows in two by them)
    uint32_t bin_grid;        // its grid: what the chip holds at once (five workgroups per CU), or a workgroup per strip row
    uint32_t bin_prio_slots;  // strip rows with at least this many segment slots raise their waves' issue priority
    uint32_t sr_empty_dwords; // size of a region no item's bbox reaches
    uint4 *queue;             // kClasses class queues of {tile, command-list quad, first piece, its candidates | segments << 9}, queue_cap entries each
    uint32_t queue_cap;
    uint32_t *tile_state;     // [tiles of the band] 0 = queued for the tile kernels, else resolved colour
    uint4 *tarena;            // tile arena: pieces + per-tile command lists (24-byte records, TestApp/GenTypes.h:430-495)
    uint32_t tarena_cap;      // in quads (16 B)
    uint32_t *tile_ptcl;      // [tiles] first quad of the tile's command list
    uint32_t *tile_ncmd;      // [tiles] commands in the list (0: resolved to one colour)
    Counters *ctr_cur;
    Counters *ctr_next;
    uint32_t *host_overflow;  // pinned host word the kernels ALSO raise when the tile arena runs out: pm_sync looks there, no copy from the device
    const uint2 *band_bbox;        // [n_band_items] bboxes of the items that reach this band, paint order
    const uint32_t *band_item;     // [n_band_items] their scene indices (nullptr: every item in scene order -- band_bbox is the scene's own ShortBbox array)
    uint32_t n_band_items;
    uint32_t bin_waves;            // pm_bin_kernel: waves that share one strip row (4: a workgroup per strip row; 1: a wave per strip row)
    uint32_t fine_grid;            // persistent workgroups of pm_fine_kernel (blocks beyond it clear strip rows)
    uint32_t split_mode;           // fine kernel: 0 = one wave per tile always, 1 = a workgroup per tile with a long list
    uint32_t fine_dense;           // 1: this frame's tile kernel is the one-wave-per-tile instantiation (six workgroups per CU)
    uint32_t *host_dense;          // pinned host word: the tile kernel says whether ITS frame was dense (1 no, 2 yes): the next frame's choice
    uint32_t verdict_waves;        // ... judged against THIS many waves whichever instantiation runs (the general kernel's lone grid: a verdict that
                                   // depended on the judging kernel's own grid flipped kernels every other frame for scenes between the two thresholds)
    uint32_t dense_factor;         // ... unless the long lists, at this many waves each, would occupy every wave of the grid (4: what a workgroup is)
    uint32_t class_thr[kClasses - 1];  // descending: a tile with more stream elements than class_thr[c] is in class <= c
    uint32_t n_heavy_classes;          // classes 0 .. n-1 are rendered by a whole workgroup per tile (long lists)
    uint32_t handout_static;           // 1: every pass of the tile kernel is handed out statically (other frames in flight)
    // large scenes: per-tile-row item lists written each frame by pm_rowcull_kernel
    uint32_t use_row_lists;
    const uint32_t *row_base;      // [band rows x row_parts + 1] list offsets (host-computed sizes): where part p of tile row r writes
    uint32_t bin_no_chains;        // this launch has a workgroup (wave) for EVERY strip row: the chains linked for the plan's own grid are not walked
    uint32_t clear_in_bin;         // the binning launch also writes the pixels of the tiles it resolves: every strip row's workgroup its own, extra workgroups (blocks >= bin_grid) those of the strip rows no item reaches (idle_sr)
    uint32_t bin_wt;               // binning stores what it leaves for the tile kernel write-through (a small frame: nothing dirty when the kernel ends)
    uint32_t row_parts;            // workgroups of pm_rowcull_kernel per tile row (each scans row_part_items of the band's items)
    uint32_t row_part_items;       // (multiples of kRowCullStep, unless a test says otherwise)
    uint2 *row_bbox;               // [row_base[rows]]
    uint32_t *row_item;
    const uint32_t *chunk_base;    // [n_items + 1]
    const float4 *chunk_bbox;      // [chunk_base[n_items]]
    const float4 *sup_bbox;        // [ceil(chunk_base[n_items] / kSuperChunks)]
    const uint32_t *lut_srgb2lin;  // [256] binary16 bits of the sRGB EOTF
    const uint32_t *lut_unorm2h;   // [256] binary16 bits of a/255
    const uint8_t *lut_lin2srgb;   // [65536] binary16 bits -> sRGB unorm8
    // command capture (debug / parity tests only)
    uint32_t *dbg_counts;
    uint32_t *dbg_solid;
    Cmd *dbg_cmds;
    uint32_t dbg_max;
    // per-slot timeline of the tile kernel (developer profiling only): 4 x u64 per slot
    // {start clock, end clock, tile | quarter << 31, wave << 32 | commands interpreted}
    unsigned long long *dbg_time;
    unsigned long long *dbg_bin;  // per strip row of pm_bin_kernel: 8 x u64 phase clocks (developer profiling)
    // one launch per frame (pm_frame_kernel)
    uint4 *fifo;               // kFifos x fifo_cap queue entries, all zero between frames (whoever pops an entry zeroes it)
    uint32_t fifo_cap;
    uint32_t one_launch;       // roles of the launch's workgroups: bit 0 bin their strip row and render its first tiles, bit 1 take tiles from the FIFOs until the frame is done (the GPU runs both in one launch)
    const uint32_t *sr_next_one;  // [n_sr_active] the one-launch grid's chains: next strip row of the workgroup (0: none)
    uint32_t one_grid_rows;    // workgroups of the launch that bin strip rows (the first ones)
    const uint32_t *idle_sr;   // strip rows no item reaches: their pixels are written by the launch too
    uint32_t n_idle_sr;
    uint32_t spin_ticks;       // a wait inside the launch gives up after this many 10 ns ticks (and raises *host_fail)
    uint32_t *host_fail;       // pinned host word: a one-launch frame gave up waiting -- pm_sync renders it again with two launches
};

void LaunchIndex(const uint8_t *scene, uint32_t n_items, uint32_t items_ix, const uint32_t *chunk_base, uint32_t n_chunks,
                 float4 *chunk_bbox, float4 *sup_bbox, hipStream_t stream);
// (t0, t1): optional timing events carried by the dispatch itself
void LaunchBin(const FrameParams &p, hipStream_t stream, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void LaunchClear(const FrameParams &p, uint32_t n_striprows, hipStream_t stream, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void LaunchCoarse(const FrameParams &p, uint32_t grid, bool capture, hipStream_t stream, hipEvent_t t0 = nullptr,
                  hipEvent_t t1 = nullptr);
void LaunchCoverage(const FrameParams &p, uint32_t n_tiles, const uint32_t *tile_solid, float *out, uint32_t out_stride, hipStream_t stream);
// clear_blocks: strip rows whose resolved tiles the launch also writes (0: pm_clear_kernel did)
void LaunchFine(const FrameParams &p, uint32_t clear_blocks, bool fused, hipStream_t stream, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
// the whole frame in one launch of `grid` resident workgroups (p.one_launch says which roles they play)
void LaunchFrame(const FrameParams &p, uint32_t grid, hipStream_t stream, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
// workgroups of pm_frame_kernel one CU holds at once (the occupancy API's answer)
int FrameKernelResidency();

}  // namespace pm
