// BinStripRows: the strip-level half of tileKernel for the strip rows of one workgroup (or wave) -- the body of
// pm_bin_kernel (pm_bin.hip) and of the binning role of pm_frame_kernel (pm_frame.hip, one launch per frame).
// (see pm_kernels_common.h for the decomposition and the rules shared by the kernel files)
#pragma once
#include "pm_kernels_common.h"
#include <pm_params.h>  // gfx950/pm_params.h: kernel arguments held in a VGPR, read with v_readlane
#include "pm_frame_row.h"

namespace pm {

// =====================================================================================
// K1: binning, one workgroup per strip row
// =====================================================================================

namespace {

#define PM_PU(field) ParamU32<offsetof(FrameParams, field)>(PR)
#define PM_PP(field) ParamPtr<decltype(FrameParams::field), offsetof(FrameParams, field)>(PR)

}  // namespace

// The binning workgroup's LDS, ONE object: every array is then a constant offset from the same base
// and an access costs `tid * 4` plus an immediate (as separate __shared__ arrays each one gets its own
// hoisted base + index register, and a dozen of them end up in scratch).
constexpr uint32_t kCtStride = kStripTiles + 1;  // row stride 17: a thread per candidate walking its row, and 16 lanes adding to one row, are both free of bank conflicts
// (overridable at build time so that the test builds of tests/emu reach the spill path with small scenes)
#ifndef PM_BIN_SURV_LDS
#define PM_BIN_SURV_LDS 512
#endif
constexpr uint32_t kSurvLds = PM_BIN_SURV_LDS;
constexpr uint32_t kSupCPL = 2;                         // super-chunks tested per lane and round (a round leaves at most threads x this many survivors)
constexpr uint32_t kChunkCPL = 4;                       // chunks tested per lane and round: half a super-chunk
static_assert(kSuperChunks == 2 * kChunkCPL && kSupCPL <= kCtStride, "two lanes per surviving super; the list fits in s_ct");
// 0.5 * width + 0.5 of a polyline / line candidate, from the width bits its aux0 word carries (the
// expression the header phase used to store per candidate: one LDS array less)
__device__ __forceinline__ float HalfWidthOf(uint32_t aux0) { return 0.5f * __uint_as_float(aux0) + 0.5f; }

// kW = waves that share one strip row: 4 (a workgroup per strip row), or 1 -- a wave per strip row, no workgroup
// barrier anywhere, four times as many strip rows resident (frames with many more strip rows than the chip holds
// workgroups, config 5: every phase of a light row is a chain of dependent round trips, not work).
// (kStamps: the developer timeline's clocks -- the one-launch kernel, which has its hand-over words beside this struct, goes without)
template <bool kStamps>
struct BinStamps {
    uint32_t s_stamp[14];  // developer timeline (kProfile builds): the clock's low 32 bits (43 s at 100 MHz) -- with 64-bit stamps the
                           // profiled kernel's LDS crossed the five-workgroups-per-CU line and its last workgroups queued behind the first
};
template <>
struct BinStamps<false> {};
template <int kW, bool kStamps = true>
struct BinLds : BinStamps<kStamps> {
    static constexpr int kThreads = 64 * kW;   // (hides pm::kThreads: candidates of a record = threads of the group)
    static constexpr int kBinWaves = kW;
    static constexpr uint32_t kSurvLds = kW == 1 ? (pm::kSurvLds < 256u ? pm::kSurvLds : 256u) : pm::kSurvLds;
    uint32_t s_part[kBinWaves];
    uint32_t s_cidx[kThreads];   // candidate item index
    uint32_t s_cmask[kThreads];  // candidate per-tile hit mask (16 bits) | item tag << 16
    uint32_t s_cpts[kThreads];   // points_ix (or byte offset of start/end for lines)
    uint32_t s_cnpt[kThreads];
    uint32_t s_cchunk[kThreads]; // first chunk-table entry of the item
    uint32_t s_soff[kThreads + 1];   // super-chunk stream offsets (a candidate's supers: those its chunk range touches)
    // per (candidate, tile): backdrop steps << 20 | relevant segments.  Row stride 17: a thread per
    // candidate walking its row, and 16 lanes adding to one row, are both free of bank conflicts.
    // (while the chunks are tested its first kSupLds words hold the surviving super-chunks of a test round, c << 24 | index
    //  in the candidate's range of supers: the counters are zeroed when the last round is through)
    uint32_t s_ct[kThreads * kCtStride];
    uint32_t s_surv[kSurvLds];  // surviving chunks of the record (c << 24 | j), while they fit
    uint32_t s_est[kStripTiles];  // per tile: stream elements the tile kernel will see
    // per tile, in paint order across batches: the last candidate that can emit anything, and the
    // last one that is nothing but an opaque Solid (backdrop-only fill, alpha 0xff).  If they
    // coincide the tile's list is {Solid(opaque)} -> Bail: the tile is that colour, written here.
    uint32_t s_last_kept[kStripTiles];
    uint32_t s_last_solid[kStripTiles];
    uint32_t s_solid_rgba[kStripTiles];
    uint32_t s_crgba[kThreads], s_caux0[kThreads], s_caux1[kThreads];  // candidate colour / payload
    uint32_t s_lut[256];  // sRGB->linear half bits | a/255 half bits << 16
    // per tile, across the records of the strip row: the tile's first piece {quad, candidates | segments << 9}
    // and its latest piece (whose header is patched when a later record adds one)
    uint32_t s_head_q[kStripTiles], s_head_n[kStripTiles], s_prev_q[kStripTiles];
    uint32_t s_piece_q[kStripTiles];  // this record's piece of the tile (quad index, 0 = none)
    uint32_t s_piece_n[kStripTiles];  // its candidates | segments << 9
    uint32_t s_wcnt[kBinWaves][kStripTiles];  // relevant segments per tile in each wave's share of the slots
    // finalisation, per wave of candidates and tile: candidates that can emit, their relevant segments,
    // pseudo elements (candidates without segments), last candidate that can emit / last opaque Solid (index + 1)
    uint32_t s_wh[kStripTiles];
    uint32_t s_wlk[kStripTiles], s_wls[kStripTiles];
    uint32_t s_whub[kBinWaves][kStripTiles];  // per wave of candidates and tile: candidates whose bbox reaches the tile
    uint32_t s_alloc[2];  // {first quad of this record's pieces (0xffffffff: the tile arena ran out), overflow seen}
};

// BinLds<4> is kept at 30.5 KB (tag and bbox mask share a word, segment counts and stroke half-widths are
// derived, 512 survivors in LDS): FIVE workgroups then share a CU -- measured: 31 184 B does, 32 208 B does
// not -- its own, or the tile kernel's (30.6 KB each) of the neighbouring frames.  Config 5 alone: binning
// 0.313 -> 0.270 ms; sustained throughput +4 % in every configuration.
static_assert((sizeof(BinLds<4>) <= 31184 && sizeof(BinLds<4, false>) <= 31184) || kSurvLds != 512, "five workgroups per CU (the profiled kernel too)");
static_assert(sizeof(BinLds<1>) <= 10240, "sixteen one-wave groups per CU");

// kOne (pm_frame_kernel, one launch per frame): the workgroup bins ONE strip row -- blockIdx.x's, no chain -- and then renders
// tiles itself.  What the tile stage reads (pieces, FIFO entries) is stored write-through, because a tile may be rendered by
// a workgroup on another XCD, whose L2 knows nothing of this one's; the row's tiles are handed over in `R` (the first ones,
// which this workgroup renders itself) and through the frame's FIFOs (pm_frame_row.h) once every wave's stores have drained.
// What a strip row's chain of dependent loads STARTS from.  pm_bin_kernel takes these as separate scalar kernel arguments in front of
// the FrameParams block (the same values as the block's fields of these names): the build preloads the first kernel arguments
// into SGPRs (-amdgpu-kernarg-preload-count, csrc/Makefile), so the strip row's descriptor, the first boxes of its item list and
// the colour tables are requested in the kernel's first instructions, next to the load of the parameter block itself, instead
// of behind it -- one round trip (cold: the first access of a launch) off every strip row's chain.
struct BinEntry {
    const uint4 *sr_desc;
    const uint2 *band_bbox;
    const uint32_t *band_item;
    const uint32_t *lut_srgb2lin, *lut_unorm2h;
    uint32_t n_band_items, use_row_lists;
};
template <bool kProfile, int kW, bool kOne = false>
__device__ __forceinline__ void BinStripRows(const FrameParams &P, BinLds<kW, kProfile || !kOne> &L, FrameRowLds *const R = nullptr, const BinEntry *const E = nullptr) {
    constexpr bool kEntry = !kOne;  // (pm_bin_kernel hands E over; the one-launch kernel reads everything from P)
    // (the names the body was written with, for kW waves per strip row: they hide the namespace's)
    constexpr int kBinWaves = kW;
    constexpr int kBinThreads = 64 * kW;
    constexpr uint32_t kBatch = 64u * kW;               // candidates per record
    constexpr uint32_t kSupLds = kBinThreads * kSupCPL;  // surviving supers a round can leave
    constexpr uint32_t kSurvLds = BinLds<kW>::kSurvLds;
    constexpr uint32_t kTPW = kStripTiles / kW;          // tiles per wave in the candidates pass
    static_assert(kW == 1 || kW == 4, "one wave or one workgroup per strip row");
    // a barrier among the waves that share the strip row: with one wave, program order (and a compiler fence)
    auto LdsBarrier = [] {
        if constexpr (kW == 1) WaveSync();
        else pm::LdsBarrier();
    };
    const ParamRegs PR = LoadParams(P);
    const uint32_t tid = threadIdx.x;
    // The first strip row's descriptor and boxes and the colour tables: requested before anything waits for the parameter block.
    // The boxes wait for the item scan in this thread's own words of the (still unused) counter array: a vector value that lives
    // from the kernel's entry into the strip-row loop is spilled (the loop's preheader is where register pressure peaks).
    uint4 srd_e = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (kEntry) {
        srd_e = E->sr_desc[blockIdx.x];  // (the grid never exceeds the work list: pm_context.hip, EnsureArena)
        uint2 bb_e = make_uint2(0u, 0u);
        uint32_t it_e = 0;
        if (!E->use_row_lists && tid < E->n_band_items) {
            bb_e = E->band_bbox[tid];
            it_e = E->band_item != nullptr ? E->band_item[tid] : tid;
        }
        // (the tables do not change from row to row: in LDS once, visible behind the first row's first barrier)
#pragma unroll
        for (uint32_t u = 0; u < 256u / kBinThreads; ++u)
            L.s_lut[tid + u * kBinThreads] = E->lut_srgb2lin[tid + u * kBinThreads] | (E->lut_unorm2h[tid + u * kBinThreads] << 16);
        L.s_ct[tid] = bb_e.x;
        L.s_ct[kBinThreads + tid] = bb_e.y;
        L.s_ct[2 * kBinThreads + tid] = it_e;
        srd_e = Scalar4(srd_e);  // (uniform: in SGPRs -- four VGPRs that live into the loop are spilled)
    }
    // a 16-byte record of the tile arena that the tile stage reads
    // (write-through in the two-launch kernel too where the host asks for it, round 6 -- FrameParams::bin_wt, a frame whose strip rows
    //  all fit the resident grid: what binning leaves for the tile kernel is read by other XCDs' workgroups anyway, and a line written
    //  through is not dirty when the kernel ends -- the release at the end of a kernel writes every dirty line of the L2s back before
    //  the next dispatch starts: 4K Tiger frame -0.5 us.  Large frames keep plain stores: config 4's 70 MB of pieces as 16-byte
    //  write-through stores took its binning from 0.124 to 0.162 ms.)
    const bool bin_wt = kOne || PM_PU(bin_wt) != 0u;  // (uniform)
    auto put_quad = [&](uint4 *q, const uint4 v) {
        if (bin_wt) StoreWT16(q, v);
        else *q = v;
    };
    const uint32_t lane = LaneId();
    const uint32_t wave = kW == 1 ? 0u : tid >> 6;
    if (blockIdx.x == 0) {
        for (uint32_t k = tid; k < kTicketParts; k += kBinThreads) PM_PP(ctr_next)->ticket[k].count = 0;
        // (the hand-over state of a one-launch frame: either kind of frame leaves the other kind's counters ready too)
        for (uint32_t k = tid; k < kFifos; k += kBinThreads) {
            PM_PP(ctr_next)->fifo[k].tail = 0;
            PM_PP(ctr_next)->fifo[k].head = 0;
        }
        for (uint32_t k = tid; k < kFifoShards; k += kBinThreads) PM_PP(ctr_next)->done_part[k].count = 0;
    }
    if (blockIdx.x == 0 && tid == 0) {
        // The counters of the NEXT frame (the other parity) are idle now: reset them
        // here so that no separate memset launch is needed.
#pragma unroll
        for (uint32_t k = 0; k < kArenaShards; ++k) {
            PM_PP(ctr_next)->ptcl[k].top = 0;
            PM_PP(ctr_next)->ptcl[k].bin_dwords = 0;
        }
#pragma unroll
        for (uint32_t k = 0; k < kClasses; ++k) PM_PP(ctr_next)->cls[k].count = 0;
        PM_PP(ctr_next)->overflow = 0;
        PM_PP(ctr_next)->done_top.parts = 0;
        PM_PP(ctr_next)->done_top.done = 0;
    }
    // Workgroup -> strip rows: the host lists the strip rows some item's bbox reaches (it sized
    // their arena regions from the same predicate); the others are background for the whole
    // life of the scene and never get a workgroup.  The grid is no larger than what the chip holds
    // at once (a workgroup that has to wait for a slot starts when the first ones END, 20 us into
    // the launch, and then sets its span): with more strip rows than that, a workgroup walks a chain
    // of rows the host linked (the lightest rows share workgroups: pm_context.hip, EnsureArena).
    for (uint32_t rix = blockIdx.x, rix_next = 0; rix < PM_PU(n_sr_active); rix = rix_next) {
    if (rix != blockIdx.x) LdsBarrier();  // the previous strip row's LDS is done with
    // One 16-byte load: {strip | tile row << 16, region, end, next strip row of this group (0: none)}.
    const bool first_row = kEntry && rix == blockIdx.x;  // (uniform) what this row starts from was requested at the kernel's entry
    uint4 srd = srd_e;
    if (!first_row) srd = PM_PP(sr_desc)[rix];
    // In flight together with it: where the strip row's item list is (large scenes: its tile row's list), or -- the
    // band's list does not depend on the strip row -- the first 256 boxes of the list themselves.  Every dependent
    // access the item scan does not make is half a microsecond of every strip row.
    uint32_t n_band = PM_PU(n_band_items);
    const uint2 *band_bbox = PM_PP(band_bbox);
    const uint32_t *band_item = PM_PP(band_item);
    uint2 bb_next = make_uint2(0u, 0u);
    uint32_t it_next = 0;
    if (PM_PU(use_row_lists)) {  // large scene: this tile row's list from pm_rowcull_kernel
        const uint2 srl = PM_PP(sr_list)[rix];
        const uint32_t lo = __builtin_amdgcn_readfirstlane(srl.x);
        n_band = __builtin_amdgcn_readfirstlane(srl.y);
        band_bbox = PM_PP(row_bbox) + lo;
        band_item = PM_PP(row_item) + lo;
    }
    // (band_item == nullptr: the list is the scene's item list itself, band_bbox its ShortBbox array)
    if (first_row && !PM_PU(use_row_lists)) {
        bb_next = make_uint2(L.s_ct[tid], L.s_ct[kBinThreads + tid]);  // (this thread's own words)
        it_next = L.s_ct[2 * kBinThreads + tid];
    } else if (tid < n_band) {
        bb_next = band_bbox[tid];
        it_next = band_item != nullptr ? band_item[tid] : tid;
    }
    if constexpr (kOne) rix_next = __builtin_amdgcn_readfirstlane(PM_PP(sr_next_one)[rix]);  // (the one-launch grid's own chains)
    else rix_next = PM_PU(bin_no_chains) ? 0u : __builtin_amdgcn_readfirstlane(srd.w);
    if (rix_next == 0) rix_next = 0xffffffffu;
    // kOne: the workgroup renders tiles of its LAST strip row itself; an earlier row of its chain hands every tile over
    const bool last_row = rix_next == 0xffffffffu;
    auto one_keep = [&](uint32_t n_heavy, uint32_t n_queued) -> uint32_t { return last_row ? OneLaunchKeep(n_heavy, n_queued) : 0u; };
    // the two colour tables ride along with the first bbox load (finalisation reads them from LDS)
    uint32_t lut_word[256 / kBinThreads] = {};
    if constexpr (!kEntry) {
#pragma unroll
        for (uint32_t u = 0; u < 256u / kBinThreads; ++u) lut_word[u] = PM_PP(lut_srgb2lin)[tid + u * kBinThreads] | (PM_PP(lut_unorm2h)[tid + u * kBinThreads] << 16);
    }
    // (the strip row by strip | tile row of the band << 16: no division by the number of strips)
    // Bits 8-15: the run of the strip's tiles this entry stands for (first tile | tiles - 1 << 4) -- the whole strip, or, where the
    // host cut a heavy strip row in two (pm_context.hip, EnsureArena), one HALF of it: two workgroups then bin the row side by
    // side, each with the candidates, chunks and segments that can matter to its own tiles.
    const uint32_t srx = __builtin_amdgcn_readfirstlane(srd.x);
    const uint32_t strip = srx & 0xffu, row_rel = srx >> 16;
    const uint32_t t_beg = (srx >> 8) & 15u, t_cnt = ((srx >> 12) & 15u) + 1u;
    const uint32_t run_mask = ((1u << t_cnt) - 1u) << t_beg;  // this entry's tiles of the strip
    const uint32_t sr = row_rel * PM_PU(strips_x) + strip;
    // this strip row's part of the tile arena (pm_device.h, Counters)
    const uint32_t shard = rix % kArenaShards;
    const uint32_t shard_quads = PM_PU(tarena_cap) / kArenaShards;
    const uint32_t shard_base = shard * shard_quads;
    const uint32_t ty = PM_PU(row0) + row_rel;
    const int sx0 = static_cast<int>(strip * kGroupW);
    const int y0 = static_cast<int>(ty * kTileH);
    const int sy0 = y0 & ~static_cast<int>(kGroupH - 1);
    // (wave-uniform floats: converted on the vector unit, then kept in SGPRs -- as VGPRs they are live
    //  through the whole kernel and end up spilled)
    auto uniform_f = [](int v) { return __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(static_cast<float>(v)))))); };
    // (the x extent of this entry's run of tiles: what boxes are culled against; the reference's own predicates keep the strip's sx0)
    const float fsx0 = uniform_f(sx0 + static_cast<int>(t_beg * kTileW)), fsx1 = uniform_f(sx0 + static_cast<int>((t_beg + t_cnt) * kTileW));
    const float fstrip0 = uniform_f(sx0), fstrip1 = uniform_f(sx0 + static_cast<int>(kGroupW));  // (the whole strip's: the reference's votes)
    const float fy0 = uniform_f(y0), fy1 = uniform_f(y0 + static_cast<int>(kTileH));
    const float fsy0 = uniform_f(sy0), fsy1 = uniform_f(sy0 + static_cast<int>(kGroupH));

    // Developer timeline (kProfile builds only): thread 0 stores the clock straight to memory, so
    // that the profiled kernel keeps the register allocation of the production one.
    // slots: 0 entry, 1 item scan done, 2 first record's headers done, 3 last segment stream done,
    //        4 last record finalised, 5 queues done, 6 chunks tested (count), 7 exit
    // (the clocks go to LDS and leave for memory in one burst when the wave ends: a global store per
    //  stamp would sit in front of the next loads -- vector memory operations complete in order -- and
    //  charge every phase a store acknowledgement)
    auto stamp = [&](uint32_t k) {
        if (kProfile) {
            if constexpr (kProfile) {
                if (tid == 0) L.s_stamp[k] = static_cast<uint32_t>(wall_clock64());
            }
        }
    };
    bool prof_first = true;
    uint32_t prof_chunks = 0;
    if constexpr (kProfile) {
        if (tid < 14) L.s_stamp[tid] = 0;
    }
    stamp(0);
    if (tid < kStripTiles) {
        L.s_est[tid] = 0;
        L.s_last_kept[tid] = 0;
        L.s_last_solid[tid] = 0;
        L.s_solid_rgba[tid] = 0;
        L.s_head_q[tid] = 0;
        L.s_head_n[tid] = 0;
        L.s_prev_q[tid] = 0;
    }
    if (tid == 0) L.s_alloc[1] = 0;
    LdsBarrier();

    const uint8_t *scene = PM_PP(scene);
    // wave-uniform values are pinned to SGPRs (readfirstlane): the record pointers and loop
    // bounds derived from them then live on the scalar unit instead of in 64-bit VGPR pairs
    const uint32_t items_ix = PM_PU(items_ix);  // kernel argument: no load on the critical path
    // This strip row owns arena[sr_base[b] .. sr_base[b+1]): the host sized it for the worst
    // case (every chunk of every candidate survives), so records are bump-allocated without
    // atomics and without a counting pass.
    uint32_t cursor = __builtin_amdgcn_readfirstlane(srd.y);
    const uint32_t region_begin = cursor;
    const uint32_t region_end = __builtin_amdgcn_readfirstlane(srd.z);
    uint32_t cursor_back = region_end;  // (the records' meta words grow down from here)

    // ---- the strip row's tail (one wave): queue the tiles with something to draw, mark the others ----
    // Lane t owns tile t of the strip row; the class masks are ballots, the command-list offsets a
    // wave scan, and the atomics' results travel by v_readlane.  In two halves: RowTailIssue sends the
    // atomics off (tile-arena space for the lists, class queue positions), RowTailFinish looks at what
    // they returned and writes the queue entries -- the wave places its share of candidates and
    // segments in between, under the atomics' round trip.
    bool tail_done = false;
    struct TailState {
        uint32_t qres;      // what this lane's atomic returned (lane c < kClasses: class c's queue; lane kClasses: tile arena)
        uint32_t list_off;  // the tile's command list inside the strip row's allocation, in quads
        uint32_t packed;    // is_queued | class << 1 | rank of the tile among the row's tiles of its class << 4
        uint32_t qtotal;    // (uniform) quads of the strip row's lists; 0: nothing to queue
        uint32_t pos;       // kOne: the tile's position among the row's queued tiles, longest list first
        uint32_t n_heavy;   // kOne (uniform): queued tiles of the row that a whole workgroup renders
        uint32_t n_queued;  // kOne (uniform)
    };
    auto RowTailIssue = [&]() -> TailState {
        const uint32_t tiles_here = min(kStripTiles, PM_PU(tiles_x) - strip * kStripTiles);
        const bool tile_lane = lane < tiles_here && ((run_mask >> (Opaque(lane) & 15u)) & 1u) != 0u;
        // (the tile's index made here, from a lane number the compiler cannot see through: hoisted to the kernel's entry it
        //  is spilled, and the reload's wait also waits for every store the wave has in flight)
        const uint32_t tl = Opaque(lane) & (kStripTiles - 1u);
        const uint32_t est = tile_lane ? L.s_est[tl] : 0u;
        // {Solid(opaque)} -> Bail: the tile is one opaque colour (TileEncoder::end, :144-151)
        const bool is_solid = est != 0 && L.s_last_kept[tl] == L.s_last_solid[tl];
        const bool is_queued = est != 0 && !is_solid;
        // cost class of the tile's list (0 = longest): the number of thresholds the estimate does not exceed
        static_assert(kClasses == 8, "seven thresholds spelled out below");
#define PM_THR(k) ((est <= ParamU32<offsetof(FrameParams, class_thr) + 4 * (k)>(PR)) ? 1u : 0u)
        const uint32_t cls = PM_THR(0) + PM_THR(1) + PM_THR(2) + PM_THR(3) + PM_THR(4) + PM_THR(5) + PM_THR(6);
#undef PM_THR
        uint32_t my_mask = 0;    // queued tiles of this lane's class
        uint32_t lane_cnt = 0;   // lane c < kClasses: tiles of class c in this strip row
        uint32_t before = 0;     // kOne: queued tiles of the row in the classes of longer lists
        uint32_t n_heavy = 0, n_queued = 0;  // (uniform)
        const uint32_t heavy_classes = PM_PU(split_mode) ? PM_PU(n_heavy_classes) : 0u;
        ForClasses([&](auto kc) {
            constexpr uint32_t k = decltype(kc)::value;
            const uint32_t mk = static_cast<uint32_t>(__ballot(is_queued && cls == k));
            if (cls == k) my_mask = mk;
            if constexpr (kOne) {
                const uint32_t nk = static_cast<uint32_t>(__popc(mk));
                if (cls > k) before += nk;
                if (k < heavy_classes) n_heavy += nk;
                n_queued += nk;
            } else {
                WriteLane<k>(lane_cnt, static_cast<uint32_t>(__popc(mk)));
            }
        });
        // command-list space of a queued tile, in quads: an element emits at most 2 commands + its
        // item's closing command, plus End; 24 bytes each
        const uint32_t slots = is_queued ? ((3u * est + 1u) * kCmdQuadsNum + kCmdQuadsDen - 1u) / kCmdQuadsDen : 0u;
        const uint32_t slots_incl = WaveInclusiveScan(slots);
        TailState ts;
        ts.qtotal = WaveLast(slots_incl);
        ts.list_off = slots_incl - slots;
        ts.packed = (is_queued ? 1u : 0u) | (cls << 1) | (static_cast<uint32_t>(__popc(my_mask & ((1u << Opaque(lane)) - 1u))) << 4);  // (Opaque: made here, not hoisted to the kernel's entry and spilled)
        ts.pos = before + (ts.packed >> 4);
        ts.n_heavy = n_heavy;
        ts.n_queued = n_queued;
        // tile-arena space and the class queue positions: ONE atomic instruction, a lane per counter
        ts.qres = 0;
        if (ts.qtotal) {  // uniform
            if constexpr (kOne) {
                // The row's own workgroup renders its longest list if a workgroup is what that takes, else its first four tiles
                // (a wave each); places for the others in the frame's FIFOs (lane 0: the workgroup tiles', lane 1: the
                // single-wave tiles' of this workgroup's shard)
                const uint32_t n_keep = one_keep(n_heavy, n_queued);
                const uint32_t give_heavy = n_heavy - (n_heavy ? n_keep : 0u);
                const uint32_t give_light = n_queued - n_heavy - (n_heavy ? 0u : n_keep);
                const uint32_t give = lane == 0u ? give_heavy : give_light;
                Fifo *const ff = &PM_PP(ctr_cur)->fifo[lane == 0u ? 0u : 1u + (blockIdx.x & (kFifoShards - 1u))];
                if (lane < 2u && give) ts.qres = atomicAdd(&ff->tail, give);
            } else {
                if (lane < kClasses && lane_cnt) ts.qres = atomicAdd(&PM_PP(ctr_cur)->cls[lane].count, lane_cnt);
            }
            if (lane == kClasses) ts.qres = AtomicAddOneLane(&PM_PP(ctr_cur)->ptcl[shard].top, ts.qtotal);  // (RowTailFinish looks at it)
        }
        // tiles with nothing to draw are background: no item touches them, or every touching
        // item lost all its segments in phase 1 (the reference writes Bail/white for them).  Their
        // pixels are written by the clearing workgroups of the tile kernel's launch from tile_state:
        // 25 MB of stores per 4K frame that would otherwise stall these latency-bound workgroups in bursts.
        const uint32_t tile = row_rel * PM_PU(tiles_x) + strip * kStripTiles + Opaque(lane);
        const uint32_t state = is_queued ? 0u : (is_solid ? L.s_solid_rgba[tl] : 0xffffffffu);
        // (clear_in_bin: the workgroup writes the resolved tiles' pixels itself when the row is through -- what it decided per tile
        //  of ITS run of the strip waits in the row's solid-colour words, which nothing reads any more)
        if constexpr (!kOne) {
            if (lane < kStripTiles) L.s_solid_rgba[tl] = tile_lane ? state : 0u;
        }
        if (tile_lane) {  // what this kernel decided per tile: 0 = queued, else the tile's colour
            if (bin_wt) StoreWT4(PM_PP(tile_state) + tile, state);
            else PM_PP(tile_state)[tile] = state;
        }
        if constexpr (kOne) {  // (the row's workgroup writes the resolved tiles' pixels itself)
            if (last_row) {  // ... in its first idle moment, from here
                if (lane < kStripTiles) R->state[tl] = tile_lane ? state : 0u;
                if (lane == 0u) R->striprow = sr;
            } else {
                // ... of an earlier row of its chain now: lane -> 4 pixels of tile lane / 4, sixteen rows
                const uint32_t st = static_cast<uint32_t>(__shfl(static_cast<int>(tile_lane ? state : 0u), static_cast<int>(lane >> 2)));
                const uint32_t px = strip * kGroupW + lane * 4u;
                if (st != 0u && px < PM_PU(width)) {
                    const uint32_t col = StoreOrder(st, PM_PU(fb_bgra));
                    uint8_t *const base = PM_PP(fb) + static_cast<size_t>(row_rel * kTileH) * PM_PU(fb_stride) + static_cast<size_t>(px) * 4;
                    const uint32_t rows_here = min(kTileH, PM_PU(height) - min(PM_PU(height), ty * kTileH));
                    const bool vec = px + 4u <= PM_PU(width) && PM_PU(fb_vec16) != 0u;
                    for (uint32_t rr = 0; rr < rows_here; ++rr) {
                        uint8_t *dst = base + static_cast<size_t>(rr) * PM_PU(fb_stride);
                        if (vec) *reinterpret_cast<uint4 *>(dst) = make_uint4(col, col, col, col);
                        else
                            for (uint32_t k = 0; k < 4u && px + k < PM_PU(width); ++k) reinterpret_cast<uint32_t *>(dst)[k] = col;
                    }
                }
            }
        }
        return ts;
    };
    auto RowTailFinish = [&](const TailState &ts) {
        if constexpr (kOne) {
            if (lane == 0u) {
                R->n_keep = one_keep(ts.n_heavy, ts.n_queued);
                R->keep_heavy = ts.n_heavy ? 1u : 0u;
            }
        }
        if (!ts.qtotal) return;
        const uint32_t cls = (ts.packed >> 1) & 7u;
        const uint32_t used = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ts.qres), kClasses)) + 1u;  // (a part's quad 0 stays unused: 0 = "no piece")
        const uint32_t base = shard_base + used;
        const uint32_t q_base = static_cast<uint32_t>(__shfl(static_cast<int>(ts.qres), static_cast<int>(cls)));  // my class's queue position
        // (L.s_alloc[1]: a record of this strip row found the tile arena full -- its pieces do not exist)
        const bool fits = used + ts.qtotal <= shard_quads && used + ts.qtotal >= used && L.s_alloc[1] == 0u;
        // (on overflow the tiles are still queued but marked "no list": the tile kernels skip
        //  them, the frame has holes, and pm_sync re-renders it with a larger arena)
        if (!fits && lane == 0) {
            PM_PP(ctr_cur)->overflow = 1;
            *PM_PP(host_overflow) = 1;
        }
        if (ts.packed & 1u) {
            const uint32_t ol = Opaque(lane);
            const uint32_t tile = row_rel * PM_PU(tiles_x) + strip * kStripTiles + ol;
            const uint32_t list_slot = fits ? base + ts.list_off : 0xffffffffu;
            PM_PP(tile_ptcl)[tile] = list_slot;
            // A queue entry is everything the tile kernels need to start: {tile (column | row of the band << 16), first
            // quad of its command list, its first piece, that piece's candidates | segments << 9}
            uint4 entry = make_uint4((strip * kStripTiles + ol) | (row_rel << 16), list_slot, L.s_head_q[ol], L.s_head_n[ol]);
            if constexpr (kOne) {
                if (!fits) entry.w |= 1u;  // (an entry is "in place" when its words y and w are non-zero; CoarseTile stops at y)
                const uint32_t n_keep = one_keep(ts.n_heavy, ts.n_queued);
                const uint32_t h_base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ts.qres), 0));
                const uint32_t l_base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ts.qres), 1));
                const uint32_t cap = PM_PU(fifo_cap);
                if (ts.pos < n_keep) {
                    R->entry[ts.pos] = entry;
                } else if (ts.pos < ts.n_heavy) {
                    const uint32_t ix = h_base + ts.pos - n_keep;
                    if (ix < cap) StoreWT16(PM_PP(fifo) + ix, entry);
                } else {
                    const uint32_t ix = l_base + ts.pos - (ts.n_heavy ? ts.n_heavy : n_keep);
                    if (ix < cap) StoreWT16(PM_PP(fifo) + static_cast<size_t>(1u + (blockIdx.x & (kFifoShards - 1u))) * cap + ix, entry);
                }
            } else {
                put_quad(PM_PP(queue) + (cls * PM_PU(queue_cap) + q_base + (ts.packed >> 4)), entry);
            }
        }
    };

    // Records hold up to kBatch CANDIDATES (not items): item bboxes are scanned kBatch at a time
    // and the survivors accumulate; a record is cut only when the next scan step would not fit.
    // Most strip rows therefore produce a single record.
    // Every dependent global access costs 1-2 us here, so the scan keeps the NEXT step's bboxes
    // in flight while it ranks the current ones.
    uint32_t ncand = 0;
    // The scan runs over the items whose bbox reaches this context's band of tile rows (a
    // paint-ordered subset the host lists once per scene / viewport; with one GPU it is every
    // item in view), not over the whole scene: with the rows sharded over N GPUs each rank
    // looks at its own share only.
    // The host sized this strip row's arena region from the same bbox predicate: a region that
    // only holds the fixed header allowance means no item can land here -- nothing to scan.
    if (region_end - cursor == PM_PU(sr_empty_dwords)) n_band = 0;
    if constexpr (!kEntry) {
#pragma unroll
        for (uint32_t u = 0; u < 256u / kBinThreads; ++u) L.s_lut[tid + u * kBinThreads] = lut_word[u];
    }
    for (uint32_t ib = 0;; ib += kBatch) {
        const bool more = ib < n_band;  // uniform
        const uint32_t j = ib + tid;
        bool cand = false;
        uint32_t mask = 0;
        const uint2 bb = bb_next;
        const uint32_t i = it_next;  // scene index of band item j
        if (j + kBatch < n_band) {
            bb_next = band_bbox[j + kBatch];
            it_next = band_item != nullptr ? band_item[j + kBatch] : j + kBatch;
        }
        if (more && tid < kBatch && j < n_band) {
            const int bx = static_cast<int>(bb.x & 0xffffu), by = static_cast<int>(bb.x >> 16);
            const int bz = static_cast<int>(bb.y & 0xffffu), bw = static_cast<int>(bb.y >> 16);
            // the tile `hit` test of PietRender.metal:214, y part + strip-wide x part
            cand = bz >= sx0 && bx < sx0 + static_cast<int>(kGroupW) && bw >= y0 && by < y0 + static_cast<int>(kTileH);
            if (cand) {
                const int t_lo = (bx > sx0) ? ((bx - sx0) >> 4) : 0;
                int t_hi = (bz - sx0) >> 4;
                if (t_hi > 15) t_hi = 15;
                mask = ((2u << t_hi) - 1u) & ~((1u << t_lo) - 1u) & run_mask;
                cand = mask != 0u;
            }
        }
        uint32_t nb = 0;
        uint32_t cpos = 0;
        if (more) cpos = BlockRank<kBinWaves>(cand, L.s_part, &nb);
        nb = __builtin_amdgcn_readfirstlane(nb);
        if (more && ncand + nb <= kBatch) {
            // append and keep scanning
            if (cand) {
                L.s_cidx[ncand + cpos] = i;
                L.s_cmask[ncand + cpos] = mask;
            }
            ncand += nb;
            continue;
        }
        if (ncand == 0) {
            if (!more) break;
            continue;  // (nb > kBatch cannot happen: a scan step tests kBatch items)
        }
        LdsBarrier();  // the appended candidates are visible
        if (kProfile && prof_first) stamp(1);  // first record starts (item scan done)

        // ---- candidate headers + chunk-stream offsets ---------------------------------
        uint32_t nch = 0;
        if (tid < ncand) {
            uint32_t tag = 0, rgba = 0, aux0 = 0, aux1 = 0;
            const uint32_t idx = L.s_cidx[tid];
            const uint8_t *item = scene + items_ix + static_cast<size_t>(idx) * kItemSize;
            // the first 20 bytes of the item, its bbox and its chunk-table entry: all loads are
            // issued before any of them is looked at (one round trip instead of a tag-dependent two)
            uint2 w01v, w23v, ibbv;
            uint32_t w4v;
            const uint2 w01 = *reinterpret_cast<const uint2 *>(item);
            const uint2 w23 = *reinterpret_cast<const uint2 *>(item + 8);
            const uint32_t w4 = LoadU32(item + 16);
            const uint2 ibb = *reinterpret_cast<const uint2 *>(scene + PM_PU(bbox_ix) + static_cast<size_t>(idx) * 8);
            uint32_t cbase = PM_PP(chunk_base)[idx];
            {   // keep the compiler from sinking any of these loads into the tag branches below
                uint32_t a0 = w01.x, a1 = w01.y, a2 = w23.x, a3 = w23.y, a4 = w4, a5 = ibb.x, a6 = ibb.y;
                PinLoaded8(a0, a1, a2, a3, a4, a5, a6, cbase);
                w01v = make_uint2(a0, a1);
                w23v = make_uint2(a2, a3);
                w4v = a4;
                ibbv = make_uint2(a5, a6);
            }
            tag = w01v.x & 0xffffu;
            uint32_t pts = 0, npt = 0, nseg = 0;
            if (tag == kItemCircle) {
                rgba = (w01v.x & kCircleEllipse) ? kCmdCircleEllipse : 0u;  // (a circle has no colour: the slot carries CmdCircle.flags)
                aux0 = ibbv.x;
                aux1 = ibbv.y;
            } else if (tag == kItemLine) {
                rgba = w23v.x;
                aux0 = w23v.y;  // width bits
                pts = items_ix + idx * static_cast<uint32_t>(kItemSize) + 16;  // start,end live in the item
                nseg = 1;
                nch = 1;  // never culled at strip level (PietRender.metal:223-247)
            } else if (tag == kItemFill) {
                rgba = w23v.x;
                aux0 = w01v.y & (kFillEvenOdd | kFillCompound);  // PietFill.flags: the winding rule, sub-path separators
                npt = w23v.y;
                pts = w4v;
                nseg = FillSegs(npt);
                nch = (nseg + kChunkSegs - 1) / kChunkSegs;
            } else if (tag == kItemPoly) {
                rgba = w01v.y;
                aux0 = w23v.x;  // width bits
                npt = w23v.y;
                pts = w4v;
                nseg = PolySegs(npt);
                nch = (nseg + kChunkSegs - 1) / kChunkSegs;
            } else {
                tag = 0;
            }
            L.s_cmask[tid] = (L.s_cmask[tid] & 0xffffu) | (tag << 16);  // (this thread's own candidate)
            L.s_crgba[tid] = rgba;
            L.s_caux0[tid] = aux0;
            L.s_caux1[tid] = aux1;
            L.s_cpts[tid] = pts;
            L.s_cnpt[tid] = npt;
            L.s_cchunk[tid] = cbase;
            // the super-chunks the item's chunks [cbase, cbase + nch) lie in (a line: one, never culled)
            if (tag == kItemLine) nch = 1;
            else if (nch) nch = (cbase + nch - 1u) / kSuperChunks - cbase / kSuperChunks + 1u;
        }
        {   // per tile: how many of the wave's candidates reach it with their bbox (an upper bound of the
            // candidates of the tile's piece); the per-share segment counters start at zero
            const uint32_t cm = tid < ncand ? (L.s_cmask[tid] & 0xffffu) : 0u;
            uint32_t hub = 0;
            ForStripTiles([&](auto tc) {
                constexpr uint32_t t = decltype(tc)::value;
                WriteLane<t>(hub, static_cast<uint32_t>(__popcll(__ballot((cm >> t) & 1u))));
            });
            if (lane < kStripTiles) {
                const uint32_t ol = Opaque(lane);  // (an address made here: hoisted out of the strip-row loop it is spilled)
                L.s_whub[wave][ol] = hub;
                L.s_wcnt[wave][ol] = 0;
            }
        }
        uint32_t total_sup;
        const uint32_t soff = BlockExclusiveScan<kBinWaves>(nch, L.s_part, &total_sup);
        total_sup = __builtin_amdgcn_readfirstlane(total_sup);
        if (tid < ncand) L.s_soff[tid] = soff;
        if (tid == 0) L.s_soff[ncand] = total_sup;

        // ---- the record: meta words from the front of the strip row's region (ascending with the slot: dword
        //      accesses of neighbouring lanes coalesce), segment slots from its back (uniform arithmetic, no
        //      allocation traffic; nothing depends on the record's size) ----
        uint32_t *const meta = PM_PP(arena) + cursor;
        float4 *const segs_top = reinterpret_cast<float4 *>(PM_PP(arena) + cursor_back) - 1;  // slot f's segment: segs_top[-f]
#define PM_META(f) meta[f]
#define PM_SEG(f) segs_top[-static_cast<ptrdiff_t>(f)]
        LdsBarrier();  // L.s_soff, s_c* visible to every wave
        if (kProfile && prof_first) stamp(2);  // headers + scan done

        // ---- super-chunk stream -> surviving chunks --------------------------------------------------
        // Two levels.  Rounds of kSupLds super-chunks (the supers the candidates' chunk ranges touch form one
        // flat stream): those whose box cannot reach the strip row are dropped, the survivors are listed in
        // stream order; then their chunks are tested the same way, half a super per lane, and the surviving
        // chunks get consecutive indices (paint order).  Every surviving chunk OWNS kChunkSegs segment slots
        // (slot = chunk_index * kChunkSegs + segment_in_chunk), so the expansion needs no compaction at all.
        // The list of surviving chunks (c << 24 | j) lives in LDS; beyond kSurvLds of them it continues in the
        // first meta word of the chunk's own slots (read before the vote overwrites it).
        uint32_t sbase = 0;  // surviving chunks so far
        // what survives: the necessary part of the segment pre-conditions for ANY segment inside the box
        auto box_survives = [&](uint32_t ctag, float4 bb, uint32_t aux0) -> bool {
            if (ctag == kItemLine) return true;  // never culled at strip level (PietRender.metal:223-247)
            if (ctag == kItemFill)  // :264-265; a box wholly LEFT of the strip can only add to backdrops (:283-286),
                                    // and only with a segment that reaches the row's top edge: ymin <= y0
                return bb.w >= fy0 && bb.y < fy1 && bb.x < fsx1 && (bb.z > fsx0 || bb.y <= fy0);
            const float hw = HalfWidthOf(aux0);  // :378-379
            return bb.w > fsy0 - hw && bb.y < fsy1 + hw && bb.z > fsx0 - hw && bb.x < fsx1 + hw;
        };
        uint32_t *const s_sup = L.s_ct;  // (kSupLds words; the counters are zeroed below)
        for (uint32_t r0 = 0; r0 < total_sup; r0 += kSupLds) {
            if (kProfile && prof_first && r0 == 0) stamp(8);
            uint32_t n_sup_surv;
            {
                const uint32_t eb = r0 + kSupCPL * tid;  // this lane's consecutive supers (stream order)
                uint32_t svb = 0;
                uint32_t pk[kSupCPL];
                if (eb < total_sup) {
                    uint32_t c = FindOwner(L.s_soff, ncand, eb);
                    uint32_t cc[kSupCPL];
                    float4 bb[kSupCPL];
#pragma unroll
                    for (uint32_t u = 0; u < kSupCPL; ++u) {
                        const uint32_t e = eb + u;
                        while (c + 1 < ncand && L.s_soff[c + 1] <= e) ++c;  // owners only move forward
                        cc[u] = c;
                        const uint32_t j = e - L.s_soff[c];
                        pk[u] = (c << 24) | j;
                        bb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (e < total_sup && (L.s_cmask[c] >> 16) != kItemLine) bb[u] = PM_PP(sup_bbox)[L.s_cchunk[c] / kSuperChunks + j];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kSupCPL; ++u)
                        if (eb + u < total_sup && box_survives(L.s_cmask[cc[u]] >> 16, bb[u], L.s_caux0[cc[u]])) svb |= 1u << u;
                }
                uint32_t srank = BlockExclusiveScan<kBinWaves>(static_cast<uint32_t>(__popc(svb)), L.s_part, &n_sup_surv);
                n_sup_surv = __builtin_amdgcn_readfirstlane(n_sup_surv);
#pragma unroll
                for (uint32_t u = 0; u < kSupCPL; ++u)
                    if ((svb >> u) & 1u) s_sup[srank++] = pk[u];
            }
            LdsBarrier();  // the round's surviving supers are listed
            if (kProfile && prof_first && r0 == 0) stamp(9);
            // their chunks: lane -> (surviving super, half of it)
            for (uint32_t q0 = 0; q0 < n_sup_surv * 2u; q0 += kBinThreads) {
                const uint32_t h = q0 + tid;
                uint32_t svb = 0;
                uint32_t pk[kChunkCPL];
                if (h < n_sup_surv * 2u) {
                    const uint32_t spk = s_sup[h >> 1];
                    const uint32_t c = spk >> 24;
                    const uint32_t ctag = L.s_cmask[c] >> 16;
                    const uint32_t cbase = L.s_cchunk[c];
                    const uint32_t nch = (SegsOf(ctag, L.s_cnpt[c]) + kChunkSegs - 1u) / kChunkSegs;  // (a line: its one)
                    // global chunk-table entries of this half super; a line's one chunk is entry 0 of its one super
                    const uint32_t g0 = ctag == kItemLine ? cbase : (cbase / kSuperChunks + (spk & 0xffffffu)) * kSuperChunks + (h & 1u) * kChunkCPL;
                    float4 bb[kChunkCPL];
                    bool in[kChunkCPL];
#pragma unroll
                    for (uint32_t u = 0; u < kChunkCPL; ++u) {
                        const uint32_t jg = g0 + u;
                        in[u] = jg >= cbase && jg - cbase < nch && (ctag != kItemLine || (h & 1u) == 0u);  // the item's own chunks
                        pk[u] = (c << 24) | (jg - cbase);
                        bb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (in[u] && ctag != kItemLine) bb[u] = PM_PP(chunk_bbox)[jg];
                    }
                    const uint32_t aux0 = L.s_caux0[c];
#pragma unroll
                    for (uint32_t u = 0; u < kChunkCPL; ++u)
                        if (in[u] && box_survives(ctag, bb[u], aux0)) svb |= 1u << u;
                }
                uint32_t ns;
                uint32_t srank = BlockExclusiveScan<kBinWaves>(static_cast<uint32_t>(__popc(svb)), L.s_part, &ns);
                ns = __builtin_amdgcn_readfirstlane(ns);
#pragma unroll
                for (uint32_t u = 0; u < kChunkCPL; ++u)
                    if ((svb >> u) & 1u) {
                        const uint32_t ix = sbase + srank++;
                        if (ix < kSurvLds) L.s_surv[ix] = pk[u];
                        else if (cursor + kSlotDwords * kChunkSegs * (ix + 1u) <= cursor_back) PM_META(ix * kChunkSegs) = pk[u];  // (else: the overflow check below)
                    }
                sbase += ns;
            }
        }
        // the per (candidate, tile) counters start at zero (their first words held the supers until now: every
        // lane is past its last look at them -- the scans above end in a barrier)
        if (tid < ncand) {
#pragma unroll
            for (uint32_t t = 0; t < kStripTiles; ++t) L.s_ct[tid * kCtStride + t] = 0;
        }
        const uint32_t n_slots = sbase * kChunkSegs;  // slots of the record in use
        if (cursor + n_slots > cursor_back - 4u * n_slots || cursor_back - 4u * n_slots > cursor_back) {  // cannot happen unless the host bound is wrong
            if (tid == 0) {
                PM_PP(ctr_cur)->overflow = 1;
                *PM_PP(host_overflow) = 1;
            }
            break;
        }
        // the heaviest strip rows set the span of the launch: their waves win the issue arbitration
        if constexpr (!kOne) {  // (a one-launch frame: every strip row outranks the tiles rendered beside it, pm_frame.hip)
            if (n_slots >= PM_PU(bin_prio_slots)) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        if (sbase > kSurvLds) __syncthreads();  // (survivors beyond the LDS list sit in global memory, written by any wave)
        else LdsBarrier();        // the survivor list is complete
        if (kProfile && prof_first) {
            stamp(10);
            if constexpr (kProfile) {
                if (tid == 0) L.s_stamp[6] = n_slots;  // (slot 6: segment slots of the first record)
            }
        }

        // ---- segment votes: a wave owns a CONTIGUOUS share of the slots (so that what it counts and
        //      later places follows slot = paint order); each lane votes one segment (phase 1), writes
        //      its slot's meta word (0 = no vote) and, if voted, the segment -------------------------------
        const uint32_t q_share = ((n_slots + kBinWaves * 64u - 1u) / (kBinWaves * 64u)) * 64u;  // slots per wave, a multiple of 64 (equal shares of whole chunks: measured no better)
        // (as scalars: `wave` is tid >> 6 to the compiler, not provably uniform -- the vote loop's control was vector compares and exec
        //  masks, and the wait for the next round's points sat in front of the loop's exit test: also behind the LAST round, where it
        //  only waits for that round's stores)
        const uint32_t wave_s = kW == 1 ? 0u : WaveId();
        const uint32_t w_lo = __builtin_amdgcn_readfirstlane(min(n_slots, wave_s * q_share)), w_hi = __builtin_amdgcn_readfirstlane(min(n_slots, w_lo + q_share));
        // A wave's FIRST 64 slots never leave its registers (round 6): nine strip rows in ten of a 4K Tiger frame have at most 64
        // slots per wave, and their segments used to go to the binning arena and straight back -- two stores per lane in the vote
        // round, and in front of the scatter a read-back that waited for those stores' acknowledgements (vector memory completes
        // in order), a round trip through memory on every row's critical path.  Only the rounds behind the first use the arena.
        float4 r0_seg = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t r0_mw = 0;
        {
            // The loop is software-pipelined by hand: the NEXT round's segment end points are requested
            // before this round's votes are computed and stored.  Vector memory operations complete in
            // order, so a wait for loads issued BEFORE the stores does not wait for the stores'
            // acknowledgements (microseconds), and the loads' own latency runs under the arithmetic.
            // fetch: slot f -> its candidate, segment index, item type (0: no segment there) and the two end
            // points as the item stores them (a compound fill's separators are sorted out by the consumer)
            auto fetch = [&](uint32_t f, uint32_t &vc, uint32_t &k, uint32_t &ctag, float2 &a, float2 &b) {
                vc = 0;
                k = 0;
                ctag = 0;
                a = b = make_float2(0.f, 0.f);
                if (f < w_hi) {
                    const uint32_t six = f / kChunkSegs;
                    // (two plain accesses, not a select of two pointers: that becomes a FLAT load, which waits on both counters)
                    uint32_t spk = Opaque(L.s_surv[min(six, kSurvLds - 1u)]);
                    if (sbase > kSurvLds) {  // (uniform: only a record whose list went beyond LDS pays for the load's wait -- a vmcnt(0) that
                                             //  would also wait for the previous round's stores)
                        if (six >= kSurvLds) spk = PM_META(six * kChunkSegs);
                    }
                    vc = spk >> 24;
                    k = (spk & 0xffffffu) * kChunkSegs + (f % kChunkSegs);
                    const uint32_t vtag = L.s_cmask[vc] >> 16;
                    if (k < SegsOf(vtag, L.s_cnpt[vc])) {
                        ctag = vtag;
                        const uint8_t *pts = scene + L.s_cpts[vc];
                        // Fill: point k to point k + 1, the last one back to point 0 (:262-263); polyline:
                        // k to k + 1 (:376-377); line: its start and end sit in the item itself
                        uint32_t ka = k, kb = k + 1u;
                        if (ctag == kItemFill && kb == L.s_cnpt[vc]) kb = 0u;
                        if (ctag == kItemLine) ka = 0u, kb = 1u;
                        a = LoadF2(pts + static_cast<size_t>(ka) * 8);
                        b = LoadF2(pts + static_cast<size_t>(kb) * 8);
                    }
                }
            };
            uint32_t vc_n, k_n, ctag_n;
            float2 a_n, b_n;
            fetch(w_lo + lane, vc_n, k_n, ctag_n, a_n, b_n);
            WaveSync();
            for (uint32_t f0 = w_lo; f0 < w_hi; f0 += 64u) {
                const bool round0 = f0 == w_lo;  // (uniform) the wave's first 64 slots stay in registers: r0_seg, r0_mw
                const uint32_t f = f0 + lane;
                const uint32_t vc = vc_n, k = k_n, ctag_f = ctag_n;
                float2 a = a_n, b = b_n;
                if (f0 + 64u < w_hi) fetch(f + 64u, vc_n, k_n, ctag_n, a_n, b_n);  // (uniform: the last round requests nothing and waits for nothing)
                // (survivors beyond the LDS list are read from the first meta word of their chunk: every lane
                //  has done so before the chunk's first lane overwrites it below -- in lockstep on the GPU
                //  anyway; the statement keeps a lane-by-lane execution of this source honest)
                WaveSync();
                bool vote = false;
                float4 seg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ctag_f == kItemFill) {
                    bool exists = true;
                    if (L.s_caux0[vc] & kFillCompound) {
                        // compound (extension D11, pm_layout.h): NaN entries separate sub-paths and start no
                        // segment; a point followed by a separator closes to the index the separator carries
                        if (a.x != a.x) exists = false;
                        else if (b.x != b.x)
                            b = LoadF2(scene + L.s_cpts[vc] + static_cast<size_t>(min(__float_as_uint(b.y), L.s_cnpt[vc] - 1u)) * 8);
                    }
                    if (exists) {
                        seg = make_float4(a.x, a.y, b.x, b.y);
                        vote = VoteFillF(seg, fy0, fy1, fstrip0, fstrip1);
                    }
                } else if (ctag_f == kItemPoly) {
                    seg = make_float4(a.x, a.y, b.x, b.y);
                    // (the row of the lane that votes for this segment in the reference: lane = segment index & 31, row = lane >> 4, quirk Q4)
                    const bool second_row = ((k & 31u) >> 4) != 0u;
                    const float fyt0 = second_row ? fsy0 + static_cast<float>(kTileH) : fsy0;
                    vote = VotePolyF(seg, HalfWidthOf(L.s_caux0[vc]), fyt0, fyt0 + static_cast<float>(kTileH), fstrip0, fstrip1, fsy0, fsy1);
                } else if (ctag_f == kItemLine) {
                    seg = make_float4(a.x, a.y, b.x, b.y);
                    vote = true;
                }
                uint32_t mword = 0;
                if (f < w_hi) {
                    if (vote) {
                        // Per tile of the strip: (a) can this segment emit a command there -- the
                        // x/box pre-conditions of phase 2 (:334, :349-350, :416-417); (b) for fills,
                        // the backdrop term of :326-333, which the reference accumulates per tile over
                        // EVERY voted segment of the row, is summed once per (item, tile) here.
                        const uint32_t tm = L.s_cmask[vc];
                        const uint32_t ctag = tm >> 16, hm = tm & 0xffffu;
                        uint32_t M = 0;
                        const float xmin = fminf(seg.x, seg.z), ymin = fminf(seg.y, seg.w);
                        const float xmax = fmaxf(seg.x, seg.z), ymax = fmaxf(seg.y, seg.w);
                        if (ctag == kItemFill) {
                            // xmin < fx1 and xmax > fx0 against integer tile edges: exact in integers
                            const int fl = static_cast<int>(floorf(fmaxf(fminf(xmin, 1048576.0f), -1048576.0f)));
                            const int ce = static_cast<int>(ceilf(fmaxf(fminf(xmax, 1048576.0f), -1048576.0f)));
                            const int t_lo = max(0, (fl - sx0) >> 4);                 // first t with x0+16 > xmin
                            const int t_hi = min(15, ((ce - sx0 + 15) >> 4) - 1);      // last t with x0 < xmax
                            if (t_hi >= t_lo) M = ((2u << t_hi) - 1u) & ~((1u << t_lo) - 1u);
                            if (ymin <= fy0) {
                                // backdrop: sign(line(x0, y0)) == sign(a) holds on a suffix of the tiles
                                // (every rounding in a*x0 + y0*b + c is monotone in x0), so one bisection
                                // finds the first tile; there s00 is the same expression, i.e. sign(a).
                                const float a = seg.w - seg.y;
                                const float b = seg.x - seg.z;
                                const float cc = -(a * seg.x + b * seg.y);
                                const float sa = Sgn(a);
                                const float yb = fy0 * b;
                                if (sa != 0.0f) {
                                    int lo = 0, hi = 16;  // first t in [0,16] where the predicate holds
                                    while (lo < hi) {
                                        const int mid = (lo + hi) >> 1;
                                        const float fxm = static_cast<float>(sx0 + mid * static_cast<int>(kTileW));
                                        if (Sgn(a * fxm + yb + cc) == sa) hi = mid; else lo = mid + 1;
                                    }
                                    if (lo < 16) atomicAdd(&L.s_ct[vc * kCtStride + lo], static_cast<uint32_t>(-static_cast<int>(sa)) << kCtShift);
                                }
                            }
                        } else if (ctag == kItemPoly) {
                            const float hw = HalfWidthOf(L.s_caux0[vc]);
                            if (ymax > fy0 - hw && ymin < fy1 + hw) {
                                // tiles t with xmax > fx0(t) - hw && xmin < fx1(t) + hw (:416-417).  Both
                                // bounds are monotone in t (tile edges are integers, every rounding is
                                // monotone): the first holds on a prefix of the tiles, the second on a
                                // suffix -- two bisections with the very expressions, no table of 17 edges.
                                int lo = 0, hi = 16;  // tiles where the first condition holds: [0, lo)
                                while (lo < hi) {
                                    const int mid = (lo + hi) >> 1;
                                    if (xmax > static_cast<float>(sx0 + mid * static_cast<int>(kTileW)) - hw) lo = mid + 1; else hi = mid;
                                }
                                const int t_end = lo;
                                lo = 0, hi = 16;      // ... the second: [lo, 16)
                                while (lo < hi) {
                                    const int mid = (lo + hi) >> 1;
                                    if (xmin < static_cast<float>(sx0 + (mid + 1) * static_cast<int>(kTileW)) + hw) hi = mid; else lo = mid + 1;
                                }
                                if (lo < t_end) M = ((1u << t_end) - 1u) & ~((1u << lo) - 1u);
                            }
                        } else {
                            M = 0xffffu;  // a line is tested by every tile its bbox hits (:223-247)
                        }
                        M &= hm;
                        if (!round0) PM_SEG(f) = seg;
                        mword = M | (vc << 16) | 0x80000000u;  // bit 31: a voted segment lives here
                    }
                    if (!round0) PM_META(f) = mword;
                }
                if (round0) {
                    r0_seg = seg;
                    r0_mw = mword;
                }
                // relevant-segment counts per (candidate, tile).  The 4 lanes of a chunk (a quad) share one
                // candidate: spread the 16 tile bits to 16 nibbles (64 bits), add the 4 lanes with two DPP
                // steps (4 <= 15 fits a nibble), and let lane j of the quad add the counts of tiles 4j .. 4j+3
                // -- instead of 16 ballots per distinct candidate.
                const uint32_t mm = mword & 0xffffu;
                {
                    static_assert(kChunkSegs == 4, "one chunk = one quad of lanes");
                    uint32_t lo8 = SpreadNibbles(mm & 0xffu), hi8 = SpreadNibbles(mm >> 8);
                    lo8 += DppQuadXor1(lo8); hi8 += DppQuadXor1(hi8);
                    lo8 += DppQuadXor2(lo8); hi8 += DppQuadXor2(hi8);
                    const uint32_t j = lane & 3u;
                    const uint32_t four = (((j < 2u) ? lo8 : hi8) >> (16u * (j & 1u))) & 0xffffu;  // tiles 4j .. 4j+3, a nibble each
                    if (f < w_hi && four) {
                        // (and per wave share and tile: what the scatter below starts from)
                        uint32_t *row = &L.s_ct[vc * kCtStride + 4u * j];
                        uint32_t *wrow = &L.s_wcnt[wave][4u * j];
#pragma unroll
                        for (uint32_t t = 0; t < 4u; ++t) {
                            const uint32_t n = (four >> (4u * t)) & 15u;
                            if (n) {
                                atomicAdd(row + t, n);
                                atomicAdd(wrow + t, n);
                            }
                        }
                    }
                }
            }
        }
        LdsBarrier();  // every wave's L.s_ct contributions and L.s_wcnt are in
        stamp(3);  // segment stream done
        if (kProfile) prof_chunks += total_sup;

        // ---- the tiles' pieces of this record: reserved NOW (tail wave), so that the atomic's round trip
        //      runs under the candidates pass.  Segments per tile are known (L.s_wcnt); candidates only by
        //      their upper bound (bbox masks, L.s_whub): a piece is {header, segments, candidates} and the
        //      slack sits unused behind the candidates it really gets ----------------------------------------
        constexpr uint32_t kTailWave = kBinWaves - 1;  // (its share of the slots is the one that may be short)
        uint32_t nrel_t = 0, pq_rel = 0, alloc_q = 0, alloc_total = 0;  // tail wave, lane t < 16
        if (wave == kTailWave) {
            uint32_t nhub = 0;
            if (lane < kStripTiles) {
#pragma unroll
                for (uint32_t w = 0; w < static_cast<uint32_t>(kBinWaves); ++w) {
                    nrel_t += L.s_wcnt[w][lane];
                    nhub += L.s_whub[w][lane];
                }
            }
            const uint32_t quads = nhub ? 1u + nrel_t + 2u * nhub : 0u;
            const uint32_t incl = WaveInclusiveScan(quads);
            alloc_total = WaveLast(incl);
            pq_rel = incl - quads;
            if (alloc_total && lane == 0) alloc_q = AtomicAddOneLane(&PM_PP(ctr_cur)->ptcl[shard].top, alloc_total);  // (looked at after the pass below)
        }

        // ---- candidates pass.  Lane = candidate (64 at a time), and every wave takes a QUARTER of the
        //      strip's tiles for ALL candidates: per tile the backdrop (prefix of the recorded steps),
        //      whether the candidate can emit anything there (its hit bit), whether it is nothing but an
        //      opaque Solid; ballots over the candidates give, per tile, the candidates of its piece, the
        //      pseudo elements (candidates without segments), the last candidate that can emit and the
        //      last opaque Solid.  (Most strip rows have well under 64 candidates: split by candidates,
        //      one wave would walk all 16 tiles while three wait.)
        constexpr uint32_t kGroups = kBatch / 64u;
        static_assert(kGroups * kTPW <= 32u, "hit bits of a lane's candidates: kTPW per group in one register");
        const uint32_t wq = wave;       // this wave's tiles: kTPW * wq .. kTPW * wq + kTPW - 1
        const uint32_t t0 = kTPW * wq;
        const uint32_t n_groups = (ncand + 63u) / 64u;  // (uniform)
        uint32_t hq = 0;                // this lane's candidates (one per group): hit bits in the wave's tiles, 4 bits per group
        // per candidate: what both passes need
        auto cand_flags = [&](uint32_t c, uint32_t &cm, uint32_t &fill_bit, uint32_t &circle_bit, uint32_t &opaque_bit, uint32_t &rule) {
            // (threads beyond the candidates read stale rows: with an empty bbox mask nothing of it counts)
            const uint32_t tm = L.s_cmask[c];
            cm = c < ncand ? (tm & 0xffffu) : 0u;
            const uint32_t tag = tm >> 16, rgba = L.s_crgba[c];
            fill_bit = tag == kItemFill ? 1u : 0u;
            circle_bit = tag == kItemCircle ? 1u : 0u;
            opaque_bit = (rgba & 0xff000000u) == 0xff000000u ? fill_bit : 0u;
            // a tile wholly inside a fill is covered if its winding is non-zero (all bits) / odd (bit 0)
            rule = (L.s_caux0[c] & kFillEvenOdd) ? 1u : 0xffffffffu;
        };
        {
            uint32_t nh_q[kTPW] = {}, lk_q[kTPW] = {}, ls_q[kTPW] = {};  // (uniform)
#pragma unroll 1
            for (uint32_t g = 0; g < n_groups; ++g) {
                const uint32_t c = g * 64u + Opaque(lane);
                uint32_t cm, fill_bit, circle_bit, opaque_bit, rule;
                cand_flags(c, cm, fill_bit, circle_bit, opaque_bit, rule);
                const uint32_t *const ct_row = &L.s_ct[c * kCtStride];
                int run = 0;  // backdrop steps were recorded at the first tile they apply to: the tiles before this wave's
                // (all twelve words requested at once, whatever the wave's quarter: one LDS round trip instead of
                //  up to twelve dependent ones on the wave the others then wait for)
#pragma unroll
                for (uint32_t t = 0; t < kStripTiles - kTPW; ++t) {
                    const int v = static_cast<int>(ct_row[t]) >> kCtShift;
                    run += t < t0 ? v : 0;
                }
                uint32_t hb = 0;
#pragma unroll
                for (uint32_t j = 0; j < kTPW; ++j) {
                    const uint32_t t = t0 + j;
                    const uint32_t raw = ct_row[t];
                    run += static_cast<int>(raw) >> kCtShift;
                    const uint32_t cnt = raw & kCtCountMask;
                    const uint32_t inside = (static_cast<uint32_t>(run) & rule) != 0u ? fill_bit : 0u;
                    // a hit bit only where the candidate can emit something: a relevant segment, a
                    // non-zero backdrop (Solid / DrawFill), or a circle
                    const uint32_t some = (cnt != 0u ? 1u : 0u) | inside | circle_bit;
                    const uint32_t hit = some & (cm >> t) & 1u;
                    const uint32_t nos = cnt == 0u ? hit : 0u;
                    hb |= hit << j;
                    const uint64_t bh = __ballot(hit), bs = __ballot(nos & opaque_bit);
                    nh_q[j] += static_cast<uint32_t>(__popcll(bh));
                    if (bh) lk_q[j] = g * 64u + 64u - static_cast<uint32_t>(__builtin_clzll(bh));  // candidate index + 1
                    if (bs) ls_q[j] = g * 64u + 64u - static_cast<uint32_t>(__builtin_clzll(bs));
                }
                hq |= hb << (kTPW * g);
                if (wq == g % kBinWaves && c < ncand) {  // (one wave per group) the colour already through unpack_unorm4x8_srgb_to_half
                    const uint32_t rgba = L.s_crgba[c];
                    L.s_cpts[c] = (L.s_lut[rgba & 0xffu] & 0xffffu) | (L.s_lut[(rgba >> 8) & 0xffu] << 16);         // rg
                    L.s_cnpt[c] = (L.s_lut[(rgba >> 16) & 0xffu] & 0xffffu) | (L.s_lut[rgba >> 24] & 0xffff0000u);  // ba
                }
            }
            if (lane < kTPW) {
                uint32_t v_h = 0, v_lk = 0, v_ls = 0;  // (lane j takes the j-th of the uniform values)
#pragma unroll
                for (uint32_t j = 0; j < kTPW; ++j) {
                    v_h = lane == j ? nh_q[j] : v_h;
                    v_lk = lane == j ? lk_q[j] : v_lk;
                    v_ls = lane == j ? ls_q[j] : v_ls;
                }
                const uint32_t ol = t0 + Opaque(lane);
                L.s_wh[ol] = v_h;
                L.s_wlk[ol] = v_lk;
                L.s_wls[ol] = v_ls;
            }
        }
        // the tail wave: where the pieces went
        if (wave == kTailWave) {
            uint32_t base_q = 1u;
            if (alloc_total) {  // uniform
                const uint32_t used = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(alloc_q))) + 1u;  // (a part's quad 0 stays unused: 0 = "no piece")
                base_q = used + alloc_total <= shard_quads && used + alloc_total >= used ? shard_base + used : 0xffffffffu;
            }
            if (lane < kStripTiles) {
                L.s_piece_q[lane] = base_q != 0xffffffffu ? base_q + pq_rel : 0u;
                L.s_piece_n[lane] = nrel_t;  // (segments; the candidates of the piece are known after the barrier)
            }
            if (lane == 0) {
                L.s_alloc[0] = base_q;
                if (base_q == 0xffffffffu) L.s_alloc[1] = 1u;
            }
        }
        // The scatter below stores this wave's slots (meta word + segment) into the tiles' pieces: the first round's from the
        // registers they were voted in, later rounds' read back from the arena.  Vector memory operations complete in order: a
        // load issued behind stores waits for their acknowledgements, and a wait for loads in front of a run of stores whose
        // number the compiler cannot count becomes vmcnt(0) -- later chunks are loaded and waited for in front of their own stores.
        constexpr uint32_t kAhead = 1;
        uint32_t mw_a[kAhead];
        float4 seg_a[kAhead];
        mw_a[0] = r0_mw;  // (the wave's first round: still in its registers)
        seg_a[0] = r0_seg;
        LdsBarrier();  // hit bits, per-wave totals, pieces
        if (kProfile) stamp(12);
        const uint32_t base_q = L.s_alloc[0];
        const bool last_record = !more;  // uniform
        TailState tail_state{0u, 0u, 0u, 0u};
        // ---- the tail wave: piece headers, the strip row's running estimates; after the strip row's
        //      LAST record also the row's tail -- classes, command-list space, queue entries -- while
        //      the other waves already place candidates and segments --------------------------------------
        if (wave == kTailWave) {
            uint32_t nh = 0, lkm = 0, lsm = 0;
            uint32_t hdr_q = 0, hdr_prev = 0, hdr_n = 0;
            if (lane < kStripTiles) {
                nh = L.s_wh[lane];
                lkm = L.s_wlk[lane];
                lsm = L.s_wls[lane];
                L.s_est[lane] += nrel_t + nh;  // (segments + closing commands: what the list will be about as long as -- and an upper bound basis for its space, 3 x this + 1)
                if (lkm) L.s_last_kept[lane] = L.s_cidx[lkm - 1u] + 1u;  // records come in paint order
                if (lsm) {
                    L.s_last_solid[lane] = L.s_cidx[lsm - 1u] + 1u;
                    L.s_solid_rgba[lane] = L.s_crgba[lsm - 1u];
                }
                if (nh && base_q != 0xffffffffu) {
                    // this piece's header (no successor yet); the tile's previous piece learns about it
                    // (the stores follow the tail's atomic: an atomic issued after them would wait for them)
                    const uint32_t pq = base_q + pq_rel;
                    const uint32_t pn = nh | (nrel_t << kPieceHitBits);
                    hdr_q = pq;
                    hdr_prev = L.s_prev_q[lane];
                    hdr_n = pn;
                    if (!hdr_prev) {
                        L.s_head_q[lane] = pq;
                        L.s_head_n[lane] = pn;
                    }
                    L.s_prev_q[lane] = pq;
                }
            }
            if (last_record) tail_state = RowTailIssue();
            if (hdr_q) {
                const uint32_t z = OpaqueZero();  // (a zero made here: as a literal it is hoisted to the kernel's entry and spilled)
                put_quad(PM_PP(tarena) + hdr_q, make_uint4(z, z, z, z));
                if (hdr_prev) {
                    if (bin_wt) StoreWT8(reinterpret_cast<uint2 *>(PM_PP(tarena) + hdr_prev), make_uint2(hdr_q, hdr_n));
                    else *reinterpret_cast<uint2 *>(PM_PP(tarena) + hdr_prev) = make_uint2(hdr_q, hdr_n);
                }
            }
        }
        if (base_q != 0xffffffffu) {  // uniform (else: the tile arena ran out; the strip row's tiles are marked "no list")
            // ---- candidate entries, the same way: lane = candidate, the wave's quarter of the tiles for every
            //      group of 64 candidates; a candidate's rank in a tile's piece is the hits of the earlier
            //      groups plus a ballot; two quads per (candidate, tile) behind the piece's segments --------
            {
                // lane t < 16: quad of the tile's first candidate entry
                uint32_t cq = 0;
                if (lane < kStripTiles) cq = L.s_piece_q[lane] + 1u + L.s_piece_n[lane];
                uint32_t rank_q[kTPW] = {};  // (uniform) entries written so far in the wave's tiles
#pragma unroll 1
                for (uint32_t g = 0; g < n_groups; ++g) {
                    const uint32_t hb = (hq >> (kTPW * g)) & ((1u << kTPW) - 1u);
                    if (__ballot(hb != 0u) == 0ull) continue;  // uniform: nothing of this group in the wave's tiles
                    const uint32_t c = min(g * 64u + Opaque(lane), ncand - 1u);
                    const uint4 e0 = make_uint4(L.s_cmask[c] >> 16, L.s_crgba[c], L.s_caux0[c], L.s_caux1[c]);
                    const uint32_t e1y = L.s_cidx[c], e1z = L.s_cpts[c], e1w = L.s_cnpt[c];
                    const uint32_t *const ct_row = &L.s_ct[c * kCtStride];
                    int run = 0;
#pragma unroll
                    for (uint32_t t = 0; t < kStripTiles - kTPW; ++t) {
                        const int v = static_cast<int>(ct_row[t]) >> kCtShift;
                        run += t < t0 ? v : 0;
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kTPW; ++j) {
                        const uint32_t t = t0 + j;
                        const uint32_t raw = ct_row[t];
                        run += static_cast<int>(raw) >> kCtShift;
                        const bool hit = (hb >> j) & 1u;
                        const uint64_t bh = __ballot(hit);
                        const uint32_t q0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cq), static_cast<int>(t)));
                        if (hit) {
                            uint4 *e = PM_PP(tarena) + q0 + 2u * (rank_q[j] + RankBelow(bh));
                            put_quad(e, e0);
                            put_quad(e + 1, make_uint4((static_cast<uint32_t>(run) << kCtShift) | (raw & kCtCountMask), e1y, e1z, e1w));
                        }
                        rank_q[j] += static_cast<uint32_t>(__popcll(bh));
                    }
                }
            }
            if (kProfile) stamp(13);  // candidate entries written
            // ---- scatter: every relevant (segment, tile) pair to its place in the tile's piece; lane
            //      t < 16 keeps the quad of the next segment of tile t written by this wave.  The next
            //      round's slots are fetched while this round's are placed ------------------------------------
            {
                uint32_t next_q = 0;
                if (lane < kStripTiles) {
                    uint32_t before = 0;
                    for (uint32_t w = 0; w < wave; ++w) before += L.s_wcnt[w][lane];
                    next_q = L.s_piece_q[lane] + 1u + before;
                }
                for (uint32_t f0 = w_lo; f0 < w_hi; f0 += 64u * kAhead) {
                    if (f0 != w_lo) {  // (uniform) a later chunk: loaded and waited for in front of its own stores
#pragma unroll
                        for (uint32_t u = 0; u < kAhead; ++u) {
                            mw_a[u] = 0;
                            const uint32_t fa = f0 + 64u * u + lane;
                            if (fa < w_hi) {
                                mw_a[u] = PM_META(fa);
                                seg_a[u] = PM_SEG(fa);
                            }
                        }
#pragma unroll
                        for (uint32_t u = 0; u < kAhead; ++u) PinSlot(mw_a[u], seg_a[u]);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kAhead; ++u) {
                    if (f0 + 64u * u >= w_hi) break;  // uniform
                    const uint32_t mm = mw_a[u] & 0xffffu;
                    const float4 seg = seg_a[u];
                    if (__ballot(mm != 0u) == 0ull) continue;  // uniform: no relevant segment among the 64 slots
                    // A segment's place in tile t's piece = the wave's running position for t + the number of LOWER lanes
                    // with a segment for t.  All sixteen tiles at once: a byte per tile, four tiles per word, four prefix
                    // sums over the wave (counts <= 64 fit a byte) -- then every lane walks the tiles of ITS segment (one or
                    // two, rarely more) instead of the wave walking every tile present among the 64 slots, 28 instructions
                    // each, 6-12 of them per round.
                    uint32_t exc[4], tot[4];  // per word of four tiles: lower lanes' counts (a byte per tile); the wave's totals (uniform)
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) {
                        const uint32_t cnt = (((mm >> (4u * k)) & 15u) * 0x00204081u) & 0x01010101u;  // bit j of the nibble -> byte j
                        const uint32_t inc = WaveInclusiveScan(cnt);
                        tot[k] = WaveLast(inc);
                        exc[k] = inc - cnt;
                    }
                    uint32_t m = mm;
                    while (__ballot(m != 0u) != 0ull) {  // uniform: as many rounds as the most tiles one segment of the 64 reaches
                        const uint32_t t = m != 0u ? static_cast<uint32_t>(__builtin_ctz(m)) : 0u;
                        // (read from lane t by every lane, active or not: a lane outside the exec mask hands over nothing)
                        const uint32_t q0 = static_cast<uint32_t>(__shfl(static_cast<int>(next_q), static_cast<int>(t)));
                        if (m != 0u) {
                            const uint32_t w = t >> 2;
                            const uint32_t below = w == 0u ? exc[0] : (w == 1u ? exc[1] : (w == 2u ? exc[2] : exc[3]));
                            put_quad(PM_PP(tarena) + q0 + ((below >> (8u * (t & 3u))) & 0xffu),
                                     make_uint4(__float_as_uint(seg.x), __float_as_uint(seg.y), __float_as_uint(seg.z), __float_as_uint(seg.w)));
                            m &= m - 1u;
                        }
                    }
                    // lane t: the wave's position in tile t moves on by the round's total for it
                    {
                        const uint32_t w = (lane >> 2) & 3u;
                        const uint32_t tw = w == 0u ? tot[0] : (w == 1u ? tot[1] : (w == 2u ? tot[2] : tot[3]));
                        if (lane < kStripTiles) next_q += (tw >> (8u * (lane & 3u))) & 0xffu;
                    }
                    }  // rounds of the chunk
                }
            }
        }
        cursor += n_slots;   // (uniform: the next record's meta words and segs)
        cursor_back -= 4u * n_slots;
        if (!more) {  // the strip row's last record: nothing left to wait for (its stores drain on their own)
            if constexpr (kOne) {  // ... unless the tiles are handed over inside the launch: every wave's pieces first
                DrainStores();
                LdsBarrier();
            }
            if (wave == kTailWave) RowTailFinish(tail_state);
            tail_done = true;
            stamp(4);  // record finalised
            break;
        }
        LdsBarrier();  // s_c* arrays are rewritten by the next record
        stamp(4);
        prof_first = false;
        ncand = 0;
        if (cand) {  // the scan step that did not fit opens the next record
            L.s_cidx[cpos] = i;
            L.s_cmask[cpos] = mask;
        }
        ncand = nb;
    }
    if (tid == 0) {
        atomicAdd(&PM_PP(ctr_cur)->ptcl[shard].bin_dwords, (cursor - region_begin) + (region_end - cursor_back));  // dwords used (stats only)
        if (PM_PP(sr_slots) != nullptr) PM_PP(sr_slots)[rix] = cursor - region_begin;  // the row's segment slots: what the host weighs strip rows by
    }
#undef PM_META
#undef PM_SEG
    // the strip row's tail, unless its last record took care of it
    if (!tail_done) {  // uniform
        if constexpr (kOne) DrainStores();  // (pieces of the row's records, in place before its tiles are handed over)
        LdsBarrier();  // L.s_est, s_last_*, s_head_* of the last record are in
        if (wave == kBinWaves - 1) RowTailFinish(RowTailIssue());
    }
    // The pixels of the tiles this strip row resolved (background, or one opaque colour): written here, at the row's very end, when
    // the binning launch does the clearing (FrameParams::clear_in_bin) -- a kilobyte per tile of pure stores from a workgroup that
    // has nothing left to do, on a chip whose store path binning leaves idle; the tile kernel's launch then starts without 2 025
    // clearing workgroups beside its first tiles.  (lane -> 4 pixels of tile lane / 4; pixel row it * kW + wave)
    if constexpr (!kOne) {
        if (PM_PU(clear_in_bin)) {  // uniform
            LdsBarrier();  // the row's tail (RowTailIssue, the tail wave) left the states in L.s_solid_rgba
            const uint32_t ol = Opaque(lane);
            const uint32_t st = L.s_solid_rgba[ol >> 2];
            const uint32_t px = strip * kGroupW + ol * 4u;
            if (st != 0u && px < PM_PU(width)) {
                const uint32_t col = StoreOrder(st, PM_PU(fb_bgra));
                const bool vec = px + 4u <= PM_PU(width) && PM_PU(fb_vec16) != 0u;
                const uint32_t rows_here = min(kTileH, PM_PU(height) - min(PM_PU(height), ty * kTileH));
                uint8_t *const base = PM_PP(fb) + static_cast<size_t>(row_rel * kTileH) * PM_PU(fb_stride) + static_cast<size_t>(px) * 4;
#pragma unroll
                for (uint32_t it = 0; it < kTileH / static_cast<uint32_t>(kW); ++it) {
                    const uint32_t rr = it * static_cast<uint32_t>(kW) + wave;
                    if (rr < rows_here) {
                        uint8_t *dst = base + static_cast<size_t>(rr) * PM_PU(fb_stride);
                        if (vec) StorePixels4(dst, make_uint4(col, col, col, col));
                        else
                            for (uint32_t k = 0; k < 4u && px + k < PM_PU(width); ++k) reinterpret_cast<uint32_t *>(dst)[k] = col;
                    }
                }
            }
        }
    }
    (void)prof_chunks;
    if (kProfile) {
        // wave 0's stamps (0 entry, 1 item scan done, 2 first record's headers done, 8 / 9 chunk tests of
        // round 0, 10 survivors listed (6: slots), 3 votes done, 12 candidates pass done, 4 entries +
        // scatter done, 7 exit), then the tail wave's own end (14)
        stamp(7);
        if constexpr (kProfile) {
            if (tid < 14) PM_PP(dbg_bin)[16ull * rix + tid] = L.s_stamp[tid];
        }
        if (wave == kBinWaves - 1 && lane == 0) PM_PP(dbg_bin)[16ull * rix + 14] = static_cast<uint32_t>(wall_clock64());
    }
    }  // strip rows of this workgroup
}

}  // namespace pm
