// Hand-written HIP kernels for gfx950 (CDNA4, wave64): the compute path of
// piet-metal re-designed for MI355X.
//
// Reference semantics being reproduced (bit-exact against oracle/):
//   tileKernel   TestApp/PietRender.metal:160-454  (+ TileEncoder :69-157)
//   renderKernel TestApp/PietRender.metal:457-566  (+ stroke/renderDf :49-60)
//   composite    TestApp/PietRender.metal:16-44
//
// Decomposition (NOT the reference's thread-per-tile / 256 MiB tile buffer) -- one kernel per
// level of parallelism, chained by events over the frame pipeline of pm_context.hip:
//
//   pm_index_kernel   (once per scene) float bounding box of every chunk of 8 consecutive
//       segments -- the segment-level analogue of the ShortBbox array the encoder builds per item.
//   pm_rowcull_kernel (large scenes only) per tile row, the paint-ordered list of items whose
//       bbox reaches the row.
//   pm_bin_kernel     one 256-thread workgroup per strip row (16 tiles x 1 tile).
//       - item bboxes vs strip row: wave64 ballots + prefix ranks compact the candidate items in
//         paint order;
//       - the chunks of all candidates form one flat stream; chunks whose box cannot reach the
//         strip row are dropped; every surviving chunk owns 8 segment slots of an arena record
//         and each lane evaluates the reference's "phase 1" segment vote for one slot
//         (PietRender.metal:258-295 fills, :374-399 polylines), the 16-bit mask of tiles the
//         segment can matter to, and for fills the backdrop step;
//       - per (item, tile) counts and backdrops, the TileEncoder solid rule per tile, the
//         command-list space of every tile and its place in one of three class queues.
//   pm_clear_kernel   pixels of the tiles binning resolved (background / one opaque colour).
//   pm_coarse_kernel  persistent; ONE WAVE PER QUEUED TILE, no workgroup barriers.
//       - candidates are filtered by a per-tile hit bit; the record's slots carrying the tile's
//         bit are gathered in paint order; each lane runs the reference's "phase 2" test for
//         (tile, segment) (:302-357, :406-440) and emits 0..3 commands; ballots / mbcnt prefix
//         ranks give every command its slot in the tile's command list in HBM (the reference's
//         24-byte Cmd records);
//       - opaque-solid detection (TileEncoder::encodeSolid/end) restarts the list; Bail tiles are
//         written as one constant here.
//   pm_fine_kernel    persistent; interprets a tile's command list (renderKernel) for its 256
//       pixels with 1, 4 or 16 waves by list length.  Everything that only depends on y
//       (segment window, the two divides of the area integral, FillEdge) is computed once per
//       lane, colour blending runs as packed half2 math, and each lane finishes with one
//       16-byte store.  Accumulators are binary16 exactly where the source declares `half`;
//       commands are applied in list order (half accumulation is order dependent).
//
// Compile with -ffp-contract=off: every source-level f32/f16 operation is one
// IEEE rounding, as in the oracle.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "pm_device.h"

namespace pm {

namespace {

// waves per SIMD the register allocation of the tile kernels aims at (5 = 96 VGPRs: no spills;
// 6 = 80 VGPRs spills to scratch)
#ifndef PM_COARSE_WPS
#define PM_COARSE_WPS 5
#endif
#ifndef PM_FINE_WPS
#define PM_FINE_WPS 5
#endif
constexpr int kThreads = 256;   // coarse / fine kernel workgroup
constexpr int kWaves = kThreads / 64;
constexpr int kBinThreads = 64 * kBinWaves;  // binning workgroup: its waves share one strip row's segment stream
constexpr uint32_t kBatch = 256;   // candidate items per binning batch
constexpr uint32_t kHeavyStream = 32;       // stream elements above which a tile is split over 4 waves
constexpr uint32_t kVeryHeavyStream = 96;   // ... over 16 waves (one pixel row each)

// ---------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t LaneId() { return __lane_id(); }

__device__ __forceinline__ uint32_t RankBelow(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Inclusive prefix sum over the 64 lanes in six DPP adds (no LDS crossbar): Hillis-Steele inside
// each row of 16 lanes (row_shr 1, 2, 4, 8), then the row totals ripple with row_bcast15 /
// row_bcast31 (the sequence LLVM's atomic optimizer emits for gfx9).
__device__ __forceinline__ uint32_t WaveInclusiveScan(uint32_t v) {
    int x = static_cast<int>(v);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast31 -> rows 2, 3
    return static_cast<uint32_t>(x);
}

// Inclusive prefix maximum (unsigned), same DPP ladder; 0 is the identity.
__device__ __forceinline__ uint32_t WaveInclusiveMax(uint32_t v) {
    int x = static_cast<int>(v);
    auto mx = [](int a, int b) { return static_cast<int>(max(static_cast<uint32_t>(a), static_cast<uint32_t>(b))); };
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
    return static_cast<uint32_t>(x);
}

// Value of the highest lane set in a (non-empty, wave-uniform) ballot mask, as a scalar.
__device__ __forceinline__ uint32_t WaveAtHighest(uint32_t v, uint64_t mask) {
    const int l = __builtin_amdgcn_readfirstlane(63 - __builtin_clzll(mask));
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l));
}

// Value of lane 63 (the total after an inclusive scan) as a wave-uniform scalar.
__device__ __forceinline__ uint32_t WaveLast(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

// Compiler-level ordering of LDS traffic inside one wave (the LDS itself executes a
// wave's instructions in order); no instruction is emitted.
__device__ __forceinline__ void WaveSync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

__device__ __forceinline__ float Sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
__device__ __forceinline__ float Sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// "not all four corners strictly on one side" test used throughout tileKernel
// (PietRender.metal:241, :289, :340, :349, :394, :431).
__device__ __forceinline__ bool Straddles(float s00, float s01, float s10, float s11) {
    return s00 * s01 + s00 * s10 + s00 * s11 < 3.0f;
}

__device__ __forceinline__ uint32_t LoadU32(const uint8_t *p) { return *reinterpret_cast<const uint32_t *>(p); }
__device__ __forceinline__ float2 LoadF2(const uint8_t *p) { return *reinterpret_cast<const float2 *>(p); }

// Segments of an item as the kernels count them.
__device__ __forceinline__ uint32_t FillSegs(uint32_t npt) { return npt; }                      // implicitly closed (:262)
__device__ __forceinline__ uint32_t PolySegs(uint32_t npt) { return npt >= 2 ? npt - 1 : 0; }  // open (:369)

// ---------------------------------------------------------------------------------
// phase-1 votes (strip level)
// ---------------------------------------------------------------------------------

// PietRender.metal:258-295.  y0 = the voting lane's tile row, sx0 = group strip x.
__device__ __forceinline__ bool VoteFill(float4 s, int y0, int sx0) {
    const float xmin = fminf(s.x, s.z), ymin = fminf(s.y, s.w);
    const float xmax = fmaxf(s.x, s.z), ymax = fmaxf(s.y, s.w);
    const float fy0 = static_cast<float>(y0);
    const float fy1 = static_cast<float>(y0 + static_cast<int>(kTileH));
    if (!(ymax >= fy0 && ymin < fy1 && xmin < static_cast<float>(sx0 + static_cast<int>(kGroupW)))) return false;
    const float a = s.w - s.y;
    const float b = s.x - s.z;
    const float c = -(a * s.x + b * s.y);
    const float left = a * static_cast<float>(sx0);
    const float right = a * static_cast<float>(sx0 + static_cast<int>(kGroupW));
    const float ytop = fmaxf(fy0, ymin);
    const float ybot = fminf(fy1, ymax);
    const float top = b * ytop;
    const float bot = b * ybot;
    const float s_top_left = Sgn(right - a * static_cast<float>(kTileW) + fy0 * b + c);
    const float s00 = Sgn(top + left + c);
    const float s01 = Sgn(top + right + c);
    const float s10 = Sgn(bot + left + c);
    const float s11 = Sgn(bot + right + c);
    bool hit = (s_top_left == Sgn(a)) && (ymin <= fy0);
    if (Straddles(s00, s01, s10, s11) && xmax > static_cast<float>(sx0)) hit = true;
    return hit;
}

// PietRender.metal:374-399.  y_test = row of the lane that votes for this segment
// (lane = segment index & 31, row = lane >> 4: quirk Q4), sx0/sy0 = group origin.
__device__ __forceinline__ bool VotePoly(float4 s, float hw, int y_test, int sx0, int sy0) {
    const float xmin = fminf(s.x, s.z), ymin = fminf(s.y, s.w);
    const float xmax = fmaxf(s.x, s.z), ymax = fmaxf(s.y, s.w);
    if (!(ymax > static_cast<float>(sy0) - hw && ymin < static_cast<float>(sy0 + static_cast<int>(kGroupH)) + hw &&
          xmax > static_cast<float>(sx0) - hw && xmin < static_cast<float>(sx0 + static_cast<int>(kGroupW)) + hw))
        return false;
    const float a = s.w - s.y;
    const float b = s.x - s.z;
    const float c = -(a * s.x + b * s.y);
    const float left = a * (static_cast<float>(sx0) - hw);
    const float right = a * (static_cast<float>(sx0 + static_cast<int>(kGroupW)) + hw);
    const float top = b * (static_cast<float>(y_test) - hw);
    const float bot = b * (static_cast<float>(y_test + static_cast<int>(kTileH)) + hw);
    const float s00 = Sgn(top + left + c);
    const float s01 = Sgn(top + right + c);
    const float s10 = Sgn(bot + left + c);
    const float s11 = Sgn(bot + right + c);
    return Straddles(s00, s01, s10, s11);
}

// Block-wide ordered rank of a predicate (NW waves).  s_part must hold NW words.
// Contains two barriers.
template <int NW>
__device__ __forceinline__ uint32_t BlockRank(bool pred, uint32_t *s_part, uint32_t *total) {
    const uint64_t m = __ballot(pred);
    const uint32_t wave = threadIdx.x >> 6;
    if (LaneId() == 0) s_part[wave] = static_cast<uint32_t>(__popcll(m));
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t v = s_part[w];
        if (w < static_cast<int>(wave)) base += v;
        tot += v;
    }
    __syncthreads();
    *total = tot;
    return base + RankBelow(m);
}

// Block-wide exclusive scan of arbitrary u32 values.  Two barriers.
template <int NW>
__device__ __forceinline__ uint32_t BlockExclusiveScan(uint32_t v, uint32_t *s_part, uint32_t *total) {
    const uint32_t incl = WaveInclusiveScan(v);
    const uint32_t wave = threadIdx.x >> 6;
    if (LaneId() == 63) s_part[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t x = s_part[w];
        if (w < static_cast<int>(wave)) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// bit k of an 8-bit value -> bit 4k
__device__ __forceinline__ uint32_t SpreadNibbles(uint32_t x) {
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x;
}
// cross-lane moves inside groups of 4 / 8 lanes (DPP: no LDS traffic)
__device__ __forceinline__ uint32_t DppQuadXor1(uint32_t v) {  // quad_perm [1,0,3,2]
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ uint32_t DppQuadXor2(uint32_t v) {  // quad_perm [2,3,0,1]
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ uint32_t DppHalfMirror(uint32_t v) {  // row_half_mirror: lane i <-> 7 - i of each 8
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x141, 0xf, 0xf, true));
}

// Largest c in [0, n) with off[c] <= e (off ascending, off[0] == 0, n >= 1).
__device__ __forceinline__ uint32_t FindOwner(const uint32_t *off, uint32_t n, uint32_t e) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

}  // namespace

// =====================================================================================
// K0: scene index, once per scene
// =====================================================================================

__global__ void pm_index_kernel(const uint8_t *scene, uint32_t n_items, const uint32_t *chunk_base, uint32_t n_chunks,
                                float4 *chunk_bbox) {
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_chunks) return;
    const uint32_t item = FindOwner(chunk_base, n_items, ch);
    const uint32_t items_ix = LoadU32(scene + 4);
    const uint8_t *it = scene + items_ix + static_cast<size_t>(item) * kItemSize;
    const uint32_t tag = LoadU32(it) & 0xffffu;
    const uint32_t npt = LoadU32(it + 12);
    const uint8_t *pts = scene + LoadU32(it + 16);
    const uint32_t nseg = (tag == kItemFill) ? FillSegs(npt) : PolySegs(npt);
    const uint32_t k0 = (ch - chunk_base[item]) * kChunkSegs;
    const uint32_t k1 = min(k0 + kChunkSegs, nseg);
    float xmin = 0.f, ymin = 0.f, xmax = 0.f, ymax = 0.f;
    // points k0 .. k1 (the fill's closing segment wraps to point 0)
    for (uint32_t k = k0; k <= k1; ++k) {
        const uint32_t pi = (tag == kItemFill && k == npt) ? 0u : k;
        const float2 p = LoadF2(pts + static_cast<size_t>(pi) * 8);
        if (k == k0) {
            xmin = xmax = p.x;
            ymin = ymax = p.y;
        } else {
            xmin = fminf(xmin, p.x); ymin = fminf(ymin, p.y);
            xmax = fmaxf(xmax, p.x); ymax = fmaxf(ymax, p.y);
        }
    }
    chunk_bbox[ch] = make_float4(xmin, ymin, xmax, ymax);
}

// =====================================================================================
// K1a: per-tile-row item lists (large scenes only), one workgroup per tile row
// =====================================================================================
//
// With thousands of items every strip-row workgroup of pm_bin_kernel would scan every bbox of
// the band (PietRender.metal:191-208 does exactly that per threadgroup).  The row part of that
// test does not depend on the strip, so for large scenes it is done once per tile row here and
// the strip rows of the row scan the (much shorter) row list instead.  Lists keep paint order;
// their sizes are known to the host from the same predicate, so row r writes exactly
// row_base[r+1] - row_base[r] entries.
__global__ __launch_bounds__(kBinThreads) void pm_rowcull_kernel(FrameParams P) {
    __shared__ uint32_t s_part[kBinWaves];
    const uint32_t tid = threadIdx.x;
    const uint32_t row_rel = blockIdx.x;
    const int y0 = static_cast<int>((P.row0 + row_rel) * kTileH);
    uint32_t out = P.row_base[row_rel];
    for (uint32_t jb = 0; jb < P.n_band_items; jb += kBinThreads) {
        const uint32_t j = jb + tid;
        bool hit = false;
        uint2 bb = make_uint2(0u, 0u);
        uint32_t it = 0;
        if (j < P.n_band_items) {
            bb = P.band_bbox[j];
            it = P.band_item[j];
            const int by = static_cast<int>(bb.x >> 16), bw = static_cast<int>(bb.y >> 16);
            hit = bw >= y0 && by < y0 + static_cast<int>(kTileH);  // row part of :198 / :214
        }
        uint32_t total;
        const uint32_t pos = BlockRank<kBinWaves>(hit, s_part, &total);
        if (hit) {
            P.row_bbox[out + pos] = bb;
            P.row_item[out + pos] = it;
        }
        out += total;
    }
}

// =====================================================================================
// K1: binning, one workgroup per strip row
// =====================================================================================

namespace {

// The kernel arguments, one dword per lane.  pm_bin_kernel is short of SGPRs: left to the
// compiler, every late use of a FrameParams field becomes its own s_load + s_waitcnt (each a
// 0.2 us scalar round trip, a dozen of them before the first useful load).  Instead the
// whole struct is fetched with ONE vector load at entry and fields are picked out with
// v_readlane -- no memory traffic, no waits.
struct ParamRegs {
    uint32_t w[(sizeof(FrameParams) / 4 + 63) / 64];
};

__device__ __forceinline__ ParamRegs LoadParams(const FrameParams &P) {
    static_assert(sizeof(FrameParams) % 4 == 0, "FrameParams is read dword-wise");
    ParamRegs r;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&P);
    const uint32_t lane = LaneId();
#pragma unroll
    for (uint32_t k = 0; k < sizeof(r.w) / 4; ++k) {
        const uint32_t ix = k * 64u + lane;
        r.w[k] = ix < sizeof(FrameParams) / 4 ? src[ix] : 0u;
    }
    return r;
}

template <size_t kOff>
__device__ __forceinline__ uint32_t ParamU32(const ParamRegs &r) {
    static_assert(kOff % 4 == 0 && kOff < sizeof(FrameParams), "field offset");
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(r.w[kOff / 256]), static_cast<int>((kOff / 4) & 63)));
}

template <typename T, size_t kOff>
__device__ __forceinline__ T ParamPtr(const ParamRegs &r) {
    const uint64_t lo = ParamU32<kOff>(r), hi = ParamU32<kOff + 4>(r);
    return reinterpret_cast<T>(lo | (hi << 32));
}

#define PM_PU(field) ParamU32<offsetof(FrameParams, field)>(PR)
#define PM_PP(field) ParamPtr<decltype(FrameParams::field), offsetof(FrameParams, field)>(PR)

}  // namespace

template <bool kProfile>
__global__ __launch_bounds__(kBinThreads, 4) void pm_bin_kernel(FrameParams P) {
    const ParamRegs PR = LoadParams(P);
    __shared__ uint32_t s_part[kBinWaves];
    __shared__ uint32_t s_cidx[kThreads];   // candidate item index
    __shared__ uint32_t s_cmask[kThreads];  // candidate per-tile hit mask (16 bits)
    __shared__ uint32_t s_ctag[kThreads];
    __shared__ uint32_t s_cpts[kThreads];   // points_ix (or byte offset of start/end for lines)
    __shared__ uint32_t s_cnseg[kThreads];  // segments of the item
    __shared__ uint32_t s_cnpt[kThreads];
    __shared__ float s_chw[kThreads];       // 0.5*width + 0.5 for polylines
    __shared__ uint32_t s_cchunk[kThreads]; // first chunk-table entry of the item
    __shared__ uint32_t s_choff[kThreads + 1];  // chunk-stream offsets
    // per (candidate, tile): backdrop steps << 20 | relevant segments.  Row stride 17: a thread per
    // candidate walking its row, and 16 lanes adding to one row, are both free of bank conflicts.
    constexpr uint32_t kCtStride = kStripTiles + 1;
    __shared__ uint32_t s_ct[kThreads * kCtStride];
    __shared__ uint32_t s_surv[kBinWaves][256];  // [0][..]: surviving chunks of one round (c << 24 | j); later scratch  // surviving chunks of one wave round: c << 24 | j
    __shared__ uint32_t s_est[kStripTiles];  // per tile: stream elements the tile kernel will see
    // per tile, in paint order across batches: the last candidate that can emit anything, and the
    // last one that is nothing but an opaque Solid (backdrop-only fill, alpha 0xff).  If they
    // coincide the tile's list is {Solid(opaque)} -> Bail: the tile is that colour, written here.
    __shared__ uint32_t s_last_kept[kStripTiles];
    __shared__ uint32_t s_last_solid[kStripTiles];
    __shared__ uint32_t s_solid_rgba[kStripTiles];
    __shared__ uint32_t s_crgba[kThreads], s_caux0[kThreads], s_caux1[kThreads];  // candidate colour / payload
    __shared__ uint32_t s_lut[256];  // sRGB->linear half bits | a/255 half bits << 16

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = LaneId();
    const uint32_t wave = tid >> 6;
    // Workgroup -> strip row: the host lists the strip rows some item's bbox reaches (it sized
    // their arena regions from the same predicate); the others are background for the whole
    // life of the scene and never get a workgroup.  One 16-byte load: {strip row, region, end}.
    const uint4 srd = PM_PP(sr_desc)[blockIdx.x];
    const uint32_t sr = __builtin_amdgcn_readfirstlane(srd.x);
    const uint32_t strip = sr % PM_PU(strips_x);
    const uint32_t row_rel = sr / PM_PU(strips_x);
    const uint32_t ty = PM_PU(row0) + row_rel;
    const int sx0 = static_cast<int>(strip * kGroupW);
    const int y0 = static_cast<int>(ty * kTileH);
    const int sy0 = y0 & ~static_cast<int>(kGroupH - 1);
    const float fsx0 = static_cast<float>(sx0), fsx1 = static_cast<float>(sx0 + static_cast<int>(kGroupW));
    const float fy0 = static_cast<float>(y0), fy1 = static_cast<float>(y0 + static_cast<int>(kTileH));
    const float fsy0 = static_cast<float>(sy0), fsy1 = static_cast<float>(sy0 + static_cast<int>(kGroupH));

    if (blockIdx.x == 0 && tid == 0) {
        // The counters of the NEXT frame (the other parity) are idle now: reset them
        // here so that no separate memset launch is needed.
        PM_PP(ctr_next)->arena_top = 0;
        PM_PP(ctr_next)->ptcl_top = 0;
        PM_PP(ctr_next)->vheavy_count = 0;
        PM_PP(ctr_next)->heavy_count = 0;
        PM_PP(ctr_next)->light_count = 0;
        PM_PP(ctr_next)->overflow = 0;
    }
    // Developer timeline (kProfile builds only): thread 0 stores the clock straight to memory, so
    // that the profiled kernel keeps the register allocation of the production one.
    // slots: 0 entry, 1 item scan done, 2 first record's headers done, 3 last segment stream done,
    //        4 last record finalised, 5 queues done, 6 chunks tested (count), 7 exit
    auto stamp = [&](uint32_t k) {
        if (kProfile) {
            if (tid == 0) PM_PP(dbg_bin)[12ull * sr + k] = wall_clock64();
        }
    };
    bool prof_first = true;
    uint32_t prof_chunks = 0;
    stamp(0);
    if (tid < kStripTiles) {
        s_est[tid] = 0;
        s_last_kept[tid] = 0;
        s_last_solid[tid] = 0;
        s_solid_rgba[tid] = 0;
    }
    __syncthreads();

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
    uint32_t head = 0;       // first record of this strip row
    uint32_t prev_rec = 0;   // record whose `next` field is still open

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
    uint32_t n_band = PM_PU(n_band_items);
    const uint2 *band_bbox = PM_PP(band_bbox);
    const uint32_t *band_item = PM_PP(band_item);
    if (PM_PU(use_row_lists)) {  // large scene: this tile row's list from pm_rowcull_kernel
        const uint32_t *rb = PM_PP(row_base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane(rb[row_rel]);
        n_band = __builtin_amdgcn_readfirstlane(rb[row_rel + 1]) - lo;
        band_bbox = PM_PP(row_bbox) + lo;
        band_item = PM_PP(row_item) + lo;
    }
    // The host sized this strip row's arena region from the same bbox predicate: a region that
    // only holds the fixed header allowance means no item can land here -- nothing to scan.
    if (region_end - cursor == PM_PU(sr_empty_dwords)) n_band = 0;
    uint2 bb_next = make_uint2(0u, 0u);
    uint32_t it_next = 0;
    if (tid < n_band) {
        bb_next = band_bbox[tid];
        it_next = band_item[tid];
    }
    // the two colour tables ride along with the first bbox load (finalisation reads them from LDS)
    if (n_band) s_lut[tid] = PM_PP(lut_srgb2lin)[tid] | (PM_PP(lut_unorm2h)[tid] << 16);
    for (uint32_t ib = 0;; ib += kBatch) {
        const bool more = ib < n_band;  // uniform
        const uint32_t j = ib + tid;
        bool cand = false;
        uint32_t mask = 0;
        const uint2 bb = bb_next;
        const uint32_t i = it_next;  // scene index of band item j
        if (j + kBatch < n_band) {
            bb_next = band_bbox[j + kBatch];
            it_next = band_item[j + kBatch];
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
                mask = ((2u << t_hi) - 1u) & ~((1u << t_lo) - 1u);
            }
        }
        uint32_t nb = 0;
        uint32_t cpos = 0;
        if (more) cpos = BlockRank<kBinWaves>(cand, s_part, &nb);
        nb = __builtin_amdgcn_readfirstlane(nb);
        if (more && ncand + nb <= kBatch) {
            // append and keep scanning
            if (cand) {
                s_cidx[ncand + cpos] = i;
                s_cmask[ncand + cpos] = mask;
            }
            ncand += nb;
            continue;
        }
        if (ncand == 0) {
            if (!more) break;
            continue;  // (nb > kBatch cannot happen: a scan step tests kBatch items)
        }
        __syncthreads();  // the appended candidates are visible
        if (kProfile && prof_first) stamp(1);  // first record starts (item scan done)

        // ---- candidate headers + chunk-stream offsets ---------------------------------
        uint32_t nch = 0;
        if (tid < ncand) {
            uint32_t tag = 0, rgba = 0, aux0 = 0, aux1 = 0;
            const uint32_t idx = s_cidx[tid];
            const uint8_t *item = scene + items_ix + static_cast<size_t>(idx) * kItemSize;
            // the first 20 bytes of the item, its bbox and its chunk-table entry: all loads are
            // issued before any of them is looked at (one round trip instead of a tag-dependent two)
            uint2 w01v, w23v, ibbv;
            uint32_t w4v;
            const uint2 w01 = *reinterpret_cast<const uint2 *>(item);
            const uint2 w23 = *reinterpret_cast<const uint2 *>(item + 8);
            const uint32_t w4 = LoadU32(item + 16);
            const uint2 ibb = *reinterpret_cast<const uint2 *>(scene + 8 + static_cast<size_t>(idx) * 8);
            uint32_t cbase = PM_PP(chunk_base)[idx];
            {   // keep the compiler from sinking any of these loads into the tag branches below
                uint32_t a0 = w01.x, a1 = w01.y, a2 = w23.x, a3 = w23.y, a4 = w4, a5 = ibb.x, a6 = ibb.y;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(cbase));
                w01v = make_uint2(a0, a1);
                w23v = make_uint2(a2, a3);
                w4v = a4;
                ibbv = make_uint2(a5, a6);
            }
            tag = w01v.x & 0xffffu;
            uint32_t pts = 0, npt = 0, nseg = 0;
            float hw = 0.0f;
            if (tag == kItemCircle) {
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
                npt = w23v.y;
                pts = w4v;
                nseg = FillSegs(npt);
                nch = (nseg + kChunkSegs - 1) / kChunkSegs;
            } else if (tag == kItemPoly) {
                rgba = w01v.y;
                aux0 = w23v.x;  // width bits
                npt = w23v.y;
                pts = w4v;
                hw = 0.5f * __uint_as_float(aux0) + 0.5f;
                nseg = PolySegs(npt);
                nch = (nseg + kChunkSegs - 1) / kChunkSegs;
            } else {
                tag = 0;
            }
            s_ctag[tid] = tag;
            s_crgba[tid] = rgba;
            s_caux0[tid] = aux0;
            s_caux1[tid] = aux1;
            s_cpts[tid] = pts;
            s_cnpt[tid] = npt;
            s_cnseg[tid] = nseg;
            s_chw[tid] = hw;
            s_cchunk[tid] = cbase;
#pragma unroll
            for (uint32_t t = 0; t < kStripTiles; ++t) s_ct[tid * kCtStride + t] = 0;
        }
        uint32_t total_ch;
        const uint32_t choff = BlockExclusiveScan<kBinWaves>(nch, s_part, &total_ch);
        total_ch = __builtin_amdgcn_readfirstlane(total_ch);
        if (tid < ncand) s_choff[tid] = choff;
        if (tid == 0) s_choff[ncand] = total_ch;

        // ---- the record (uniform arithmetic, no allocation traffic) -----------------------
        const uint32_t mask_dwords = (ncand + 3u) & ~3u;
        const uint32_t rec = cursor;
        const uint32_t size = kRecHdrDwords + mask_dwords + (kCandDwords + kCtDwords) * ncand + 5u * kChunkSegs * total_ch;
        if (rec + size > region_end) {  // cannot happen unless the host bound is wrong
            if (tid == 0) PM_PP(ctr_cur)->overflow = 1;
            break;
        }
        cursor += size;
        uint32_t *hdr = PM_PP(arena) + rec;
        uint32_t *mask_tab = hdr + kRecHdrDwords;
        uint32_t *cand_rec = mask_tab + mask_dwords;
        uint32_t *ct_tab = cand_rec + kCandDwords * ncand;
        float4 *segs = reinterpret_cast<float4 *>(ct_tab + kCtDwords * ncand);
        uint32_t *meta = reinterpret_cast<uint32_t *>(segs + kChunkSegs * total_ch);
        if (tid == 0) {
            if (prev_rec) PM_PP(arena)[prev_rec] = rec;
            hdr[0] = 0;  // next
            hdr[1] = ncand;
            hdr[2] = total_ch;
        }
        if (head == 0) head = rec;
        prev_rec = rec;
        __syncthreads();  // s_choff, s_c* visible to every wave
        if (kProfile && prof_first) stamp(2);  // headers + scan done
        prof_first = false;

        // ---- chunk stream -> surviving chunks -> segment votes ---------------------------------
        // Block rounds of 256 chunks: chunks whose box cannot reach the strip row are dropped and
        // the survivors get consecutive indices (paint order).  Every surviving chunk OWNS
        // kChunkSegs segment slots (slot = chunk_index * kChunkSegs + segment_in_chunk), so the
        // expansion needs no compaction at all: each lane votes one segment (phase 1), writes
        // its slot's meta word (0 = no vote) and, if voted, the segment -- and the four waves
        // simply split the round's elements evenly.
        uint32_t sbase = 0;  // surviving chunks so far
        constexpr uint32_t kCPL = 4;  // chunks tested per lane per round: fewer rounds, fewer barriers
        for (uint32_t r0 = 0; r0 < total_ch; r0 += kBinThreads * kCPL) {
            const uint32_t eb = r0 + kCPL * tid;  // this lane's consecutive chunks (stream order)
            if (kProfile && r0 == 0) stamp(8);
            uint32_t svb = 0;
            uint32_t pk[kCPL];
            if (eb < total_ch) {
                uint32_t c = FindOwner(s_choff, ncand, eb);
                uint32_t cc[kCPL];
                float4 bb[kCPL];
#pragma unroll
                for (uint32_t u = 0; u < kCPL; ++u) {
                    const uint32_t e = eb + u;
                    while (c + 1 < ncand && s_choff[c + 1] <= e) ++c;  // owners only move forward
                    cc[u] = c;
                    const uint32_t j = e - s_choff[c];
                    pk[u] = (c << 24) | j;
                    bb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < total_ch && s_ctag[c] != kItemLine) bb[u] = PM_PP(chunk_bbox)[s_cchunk[c] + j];
                }
#pragma unroll
                for (uint32_t u = 0; u < kCPL; ++u) {
                    if (eb + u >= total_ch) continue;
                    const uint32_t ctag = s_ctag[cc[u]];
                    bool sv;
                    if (ctag == kItemLine) {
                        sv = true;
                    } else if (ctag == kItemFill) {  // necessary part of :264-265 for any segment of the chunk
                        sv = bb[u].w >= fy0 && bb[u].y < fy1 && bb[u].x < fsx1;
                    } else {  // necessary part of :378-379
                        const float hw = s_chw[cc[u]];
                        sv = bb[u].w > fsy0 - hw && bb[u].y < fsy1 + hw && bb[u].z > fsx0 - hw && bb[u].x < fsx1 + hw;
                    }
                    if (sv) svb |= 1u << u;
                }
            }
            uint32_t ns;
            uint32_t srank = BlockExclusiveScan<kBinWaves>(static_cast<uint32_t>(__popc(svb)), s_part, &ns);
            ns = __builtin_amdgcn_readfirstlane(ns);
            if (kProfile && r0 == 0) stamp(9);
            if (ns == 0) continue;  // uniform
#pragma unroll
            for (uint32_t u = 0; u < kCPL; ++u)
                if ((svb >> u) & 1u) (&s_surv[0][0])[srank++] = pk[u];
            __syncthreads();
            const uint32_t n_el = ns * kChunkSegs;
            if (kProfile && r0 == 0) stamp(10);
            for (uint32_t f0 = wave * 64u; f0 < n_el; f0 += kBinThreads) {
                const uint32_t f = f0 + lane;
                bool vote = false;
                uint32_t vc = 0;
                float4 seg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < n_el) {
                    const uint32_t spk = (&s_surv[0][0])[f / kChunkSegs];
                    vc = spk >> 24;
                    const uint32_t k = (spk & 0xffffffu) * kChunkSegs + (f % kChunkSegs);
                    if (k < s_cnseg[vc]) {
                        const uint32_t ctag = s_ctag[vc];
                        const uint8_t *pts = scene + s_cpts[vc];
                        if (ctag == kItemFill) {
                            const uint32_t k1 = (k + 1 == s_cnpt[vc]) ? 0u : k + 1;
                            const float2 a = LoadF2(pts + static_cast<size_t>(k) * 8);
                            const float2 b = LoadF2(pts + static_cast<size_t>(k1) * 8);
                            seg = make_float4(a.x, a.y, b.x, b.y);
                            vote = VoteFill(seg, y0, sx0);
                        } else if (ctag == kItemPoly) {
                            const float2 a = LoadF2(pts + static_cast<size_t>(k) * 8);
                            const float2 b = LoadF2(pts + static_cast<size_t>(k + 1) * 8);
                            seg = make_float4(a.x, a.y, b.x, b.y);
                            const int y_test = sy0 + static_cast<int>(((k & 31u) >> 4) * kTileH);
                            vote = VotePoly(seg, s_chw[vc], y_test, sx0, sy0);
                        } else {  // line
                            const float2 a = LoadF2(pts);
                            const float2 b = LoadF2(pts + 8);
                            seg = make_float4(a.x, a.y, b.x, b.y);
                            vote = true;
                        }
                    }
                }
                uint32_t mword = 0;
                if (f < n_el) {
                    const uint32_t slot = sbase * kChunkSegs + f;
                    if (vote) {
                // Per tile of the strip: (a) can this segment emit a command there -- the
                // x/box pre-conditions of phase 2 (:334, :349-350, :416-417); (b) for fills,
                // the backdrop term of :326-333, which the reference accumulates per tile over
                // EVERY voted segment of the row, is summed once per (item, tile) here.
                const uint32_t ctag = s_ctag[vc];
                const uint32_t hm = s_cmask[vc];
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
                            if (lo < 16) atomicAdd(&s_ct[vc * kCtStride + lo], static_cast<uint32_t>(-static_cast<int>(sa)) << kCtShift);
                        }
                    }
                } else if (ctag == kItemPoly) {
                    const float hw = s_chw[vc];
                    if (ymax > fy0 - hw && ymin < fy1 + hw) {
#pragma unroll 4
                        for (uint32_t t = 0; t < kStripTiles; ++t) {
                            const float fx0 = static_cast<float>(sx0 + static_cast<int>(t * kTileW));
                            const float fx1 = static_cast<float>(sx0 + static_cast<int>((t + 1) * kTileW));
                            if (xmax > fx0 - hw && xmin < fx1 + hw) M |= 1u << t;
                        }
                    }
                } else {
                    M = 0xffffu;  // a line is tested by every tile its bbox hits (:223-247)
                }
                M &= hm;
                        segs[slot] = seg;
                        mword = M | (vc << 16) | 0x80000000u;  // bit 31: a voted segment lives here
                    }
                    meta[slot] = mword;
                }
                // relevant-segment counts per (candidate, tile).  The 8 lanes of a chunk share one
                // candidate: spread the 16 tile bits to 16 nibbles (64 bits), add the 8 lanes with
                // three DPP steps (8 <= 15 fits a nibble), and let lane j of the chunk add the counts
                // of tiles 2j and 2j+1 -- ~40 instructions instead of 16 ballots per distinct candidate.
                {
                    static_assert(kChunkSegs == 8, "one chunk = 8 lanes");
                    const uint32_t mm = mword & 0xffffu;
                    uint32_t lo8 = SpreadNibbles(mm & 0xffu), hi8 = SpreadNibbles(mm >> 8);
                    lo8 += DppQuadXor1(lo8); hi8 += DppQuadXor1(hi8);
                    lo8 += DppQuadXor2(lo8); hi8 += DppQuadXor2(hi8);
                    lo8 += DppHalfMirror(lo8); hi8 += DppHalfMirror(hi8);
                    const uint32_t j = lane & 7u;
                    const uint32_t two = (((j < 4u) ? lo8 : hi8) >> (8u * (j & 3u))) & 0xffu;
                    if (f < n_el && two) {
                        uint32_t *row = &s_ct[vc * kCtStride + 2u * j];
                        if (two & 15u) atomicAdd(row, two & 15u);
                        if (two >> 4) atomicAdd(row + 1, two >> 4);
                    }
                }
            }
            if (kProfile && r0 == 0) {
                stamp(11);
                if (tid == 0) PM_PP(dbg_bin)[12ull * sr + 6] = n_el;  // (slot 6: elements of round 0)
            }
            sbase += ns;
            __syncthreads();  // s_surv is rewritten by the next round
        }
        if (tid == 0) hdr[3] = sbase * kChunkSegs;  // slots the tile kernel has to scan
        __syncthreads();  // every wave's s_ct contributions are in
        stamp(3);  // segment stream done
        if (kProfile) prof_chunks += total_ch;

        // ---- candidate records, per-(candidate, tile) table, mask table ------------------------
        if (tid < mask_dwords) {
            uint32_t w0 = 0;
            if (tid < ncand) {
                // keep a hit bit only where the candidate can emit something: a relevant
                // segment, a non-zero backdrop (Solid / DrawFill), or a circle
                uint32_t hm = 0;
                int run = 0;  // backdrop steps were recorded at the first tile they apply to
                const uint32_t cm = s_cmask[tid];
                const uint32_t tag = s_ctag[tid], rgba = s_crgba[tid];
                const bool opaque = (rgba & 0xff000000u) == 0xff000000u;
                uint4 *ctw = reinterpret_cast<uint4 *>(ct_tab + kCtDwords * tid);
#pragma unroll 1
                for (uint32_t q = 0; q < 4; ++q) {
                    uint32_t ct[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t t = 4 * q + k;
                        const uint32_t raw = s_ct[tid * kCtStride + t];
                        const uint32_t cnt = raw & kCtCountMask;
                        run += static_cast<int>(raw) >> kCtShift;
                        ct[k] = (static_cast<uint32_t>(run) << kCtShift) | cnt;
                        const bool pseudo = tag == kItemCircle || (tag == kItemFill && run != 0);
                        uint32_t n_el = cnt ? cnt : (pseudo ? 1u : 0u);
                        if (!((cm >> t) & 1u) || tag == 0) n_el = 0;
                        if (n_el) hm |= 1u << t;
                        // for the per-tile pass below: elements | "is nothing but an opaque Solid" << 31
                        s_ct[tid * kCtStride + t] = n_el | ((n_el && tag == kItemFill && cnt == 0 && opaque) ? 0x80000000u : 0u);
                    }
                    ctw[q] = make_uint4(ct[0], ct[1], ct[2], ct[3]);
                }
                w0 = tag | (hm << 16);
                const uint32_t rg = (s_lut[rgba & 0xffu] & 0xffffu) | (s_lut[(rgba >> 8) & 0xffu] << 16);
                const uint32_t ba = (s_lut[(rgba >> 16) & 0xffu] & 0xffffu) | (s_lut[rgba >> 24] & 0xffff0000u);
                uint4 *cr = reinterpret_cast<uint4 *>(cand_rec + kCandDwords * tid);
                cr[0] = make_uint4(w0, rgba, s_caux0[tid], s_caux1[tid]);
                cr[1] = make_uint4(s_cidx[tid], 0u, rg, ba);
                s_cpts[tid] = rgba;  // (points offsets are no longer needed) colour for the solid test
            }
            mask_tab[tid] = w0;
        }
        __syncthreads();
        // ---- per tile, in paint order: elements queued, last candidate that can emit, last one
        //      that is nothing but an opaque Solid.  thread = (tile, slice of the candidates)
        {
            const uint32_t t = tid & (kStripTiles - 1u), sl = tid >> 4;
            uint32_t est_p = 0, lk = 0, ls = 0;
            for (uint32_t c = sl; c < ncand; c += kThreads / kStripTiles) {
                const uint32_t v = s_ct[c * kCtStride + t];
                if (v) {
                    est_p += v & 0x7fffffffu;
                    lk = c + 1u;
                    if (v >> 31) ls = c + 1u;
                }
            }
            uint32_t *part = &s_surv[0][0];  // (free during finalisation) [3][16 slices][16 tiles]
            part[sl * kStripTiles + t] = est_p;
            part[256 + sl * kStripTiles + t] = lk;
            part[512 + sl * kStripTiles + t] = ls;
            __syncthreads();
            if (tid < kStripTiles) {
                uint32_t est = 0, lkm = 0, lsm = 0;
#pragma unroll
                for (uint32_t q = 0; q < kThreads / kStripTiles; ++q) {
                    est += part[q * kStripTiles + tid];
                    lkm = max(lkm, part[256 + q * kStripTiles + tid]);
                    lsm = max(lsm, part[512 + q * kStripTiles + tid]);
                }
                s_est[tid] += est;
                if (lkm) s_last_kept[tid] = s_cidx[lkm - 1u] + 1u;  // records come in paint order
                if (lsm) {
                    s_last_solid[tid] = s_cidx[lsm - 1u] + 1u;
                    s_solid_rgba[tid] = s_cpts[lsm - 1u];
                }
            }
        }
        __syncthreads();  // s_c* arrays are rewritten by the next record
        stamp(4);  // record finalised
        ncand = 0;
        if (!more) break;
        if (cand) {  // the scan step that did not fit opens the next record
            s_cidx[cpos] = i;
            s_cmask[cpos] = mask;
        }
        ncand = nb;
    }
    if (tid == 0) {
        PM_PP(striprow_head)[sr] = head;
        atomicAdd(&PM_PP(ctr_cur)->arena_top, cursor - region_begin);  // dwords used (stats only)
    }

    // ---- queue the tiles with something to draw, mark the others --------------------------
    // One wave is enough: lane t owns tile t of the strip row; the class masks are ballots, the
    // command-list offsets a wave scan, and the atomics' results travel by v_readlane.
    __syncthreads();  // s_est, s_last_* of the last record are in
    if (wave != 0) return;
    const uint32_t tiles_here = min(kStripTiles, PM_PU(tiles_x) - strip * kStripTiles);
    const bool tile_lane = lane < tiles_here;
    const uint32_t est = tile_lane ? s_est[lane] : 0u;
    // {Solid(opaque)} -> Bail: the tile is one opaque colour (TileEncoder::end, :144-151)
    const bool is_solid = est != 0 && s_last_kept[lane & (kStripTiles - 1u)] == s_last_solid[lane & (kStripTiles - 1u)];
    const bool is_queued = est != 0 && !is_solid;
    const uint32_t vheavy = static_cast<uint32_t>(__ballot(is_queued && est > kVeryHeavyStream));
    const uint32_t heavy = static_cast<uint32_t>(__ballot(is_queued && est > kHeavyStream && est <= kVeryHeavyStream));
    const uint32_t light = static_cast<uint32_t>(__ballot(is_queued && est <= kHeavyStream));
    const uint32_t queued = vheavy | heavy | light;
    // command-list slots of a queued tile: an element emits at most 2 commands + its item's
    // closing command, plus End
    const uint32_t slots = is_queued ? 3u * est + 1u : 0u;
    const uint32_t slots_incl = WaveInclusiveScan(slots);
    const uint32_t qtotal = WaveLast(slots_incl);
    // list space and the three queue positions: four atomics in flight at once
    uint32_t qres = 0;
    if (queued) {  // uniform
        if (lane == 3) qres = atomicAdd(&PM_PP(ctr_cur)->ptcl_top, qtotal);
        if (lane == 0 && vheavy) qres = atomicAdd(&PM_PP(ctr_cur)->vheavy_count, static_cast<uint32_t>(__popc(vheavy)));
        if (lane == 1 && heavy) qres = atomicAdd(&PM_PP(ctr_cur)->heavy_count, static_cast<uint32_t>(__popc(heavy)));
        if (lane == 2 && light) qres = atomicAdd(&PM_PP(ctr_cur)->light_count, static_cast<uint32_t>(__popc(light)));
    }
    // tiles with nothing to draw are background: no item touches them, or every touching
    // item lost all its segments in phase 1 (the reference writes Bail/white for them).  Their
    // pixels are written by pm_clear_kernel from tile_state: 25 MB of stores per 4K frame that
    // would otherwise stall these latency-bound workgroups in bursts.
    const uint32_t tile = row_rel * PM_PU(tiles_x) + strip * kStripTiles + lane;
    if (tile_lane)  // what this kernel decided per tile: 0 = queued, else the tile's colour
        PM_PP(tile_state)[tile] = is_queued ? 0u : (is_solid ? s_solid_rgba[lane] : 0xffffffffu);
    if (queued) {
        const uint32_t q_a = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(qres), 0));
        const uint32_t q_b = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(qres), 1));
        const uint32_t q_c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(qres), 2));
        const uint32_t base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(qres), 3));
        const bool fits = base + qtotal <= PM_PU(ptcl_cap) && base + qtotal >= base;
        // (on overflow the tiles are still queued but marked "no list": the tile kernels skip
        //  them, the frame has holes, and pm_sync re-renders it with a larger arena)
        if (!fits && lane == 0) PM_PP(ctr_cur)->overflow = 1;
        if (is_queued) {
            const uint32_t list_slot = fits ? base + (slots_incl - slots) : 0xffffffffu;
            PM_PP(tile_ptcl)[tile] = list_slot;
            // three queues, by expected list length: the fine kernel starts with the longest.  A
            // queue entry is everything the tile kernels need to start: {tile, first command
            // slot, first binning record of the strip row, commands written (pm_coarse_kernel)}
            const uint4 entry = make_uint4(tile, list_slot, head, 0u);
            const uint32_t below = (1u << lane) - 1u;
            if ((vheavy >> lane) & 1u) PM_PP(queue)[q_a + __popc(vheavy & below)] = entry;
            if ((heavy >> lane) & 1u) PM_PP(queue)[PM_PU(queue_cap) + q_b + __popc(heavy & below)] = entry;
            if ((light >> lane) & 1u) PM_PP(queue)[2u * PM_PU(queue_cap) + q_c + __popc(light & below)] = entry;
        }
    }
    stamp(5);  // queues + list slots done
    (void)prof_chunks;
    stamp(7);
}

// =====================================================================================
// K1b: pixels of the tiles binning resolved (background or one opaque colour) -- the composite
// of PietRender.metal:34-44 for tiles that never reach the tile kernels.  Pure store bandwidth;
// runs next to pm_coarse_kernel / pm_fine_kernel, which write the other tiles.
// =====================================================================================
__device__ __forceinline__ void ClearStripRow(const FrameParams &P, uint32_t striprow) {
    const uint32_t lane = LaneId(), wave = threadIdx.x >> 6;
    const uint32_t strip = striprow % P.strips_x;
    const uint32_t row_rel = striprow / P.strips_x;
    const uint32_t t = lane >> 2;  // tile of this lane's 4 pixels
    const uint32_t tx = strip * kStripTiles + t;
    if (tx >= P.tiles_x) return;
    const uint32_t col = P.tile_state[row_rel * P.tiles_x + tx];
    if (col == 0) return;  // queued: the tile kernels write it
    const uint32_t px = strip * kGroupW + lane * 4u;
    const uint32_t y0 = (P.row0 + row_rel) * kTileH;
    // 16 pixel rows x 1024 B per strip row: thread -> (row = it*4 + wave, 16 B = 4 px at lane*4)
#pragma unroll
    for (uint32_t it = 0; it < kTileH / kBinWaves; ++it) {
        const uint32_t r = it * kBinWaves + wave;
        const uint32_t py = y0 + r;
        if (py < P.height && px < P.width) {
            uint8_t *dst = P.fb + static_cast<size_t>(row_rel * kTileH + r) * P.fb_stride + static_cast<size_t>(px) * 4;
            if (px + 4 <= P.width && P.fb_vec16) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(col, col, col, col);
            } else {
                for (uint32_t k = 0; k < 4 && px + k < P.width; ++k) reinterpret_cast<uint32_t *>(dst)[k] = col;
            }
        }
    }
}

__global__ __launch_bounds__(kBinThreads) void pm_clear_kernel(FrameParams P) { ClearStripRow(P, blockIdx.x); }

// =====================================================================================
// K2: per-tile command lists (tile-level half of tileKernel), one wave per tile
// =====================================================================================

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

#ifndef PM_WAVE_CMDS
#define PM_WAVE_CMDS 256
#endif
constexpr uint32_t kFineChunk = PM_WAVE_CMDS;  // commands staged in LDS per wave by the interpreter
constexpr uint32_t kWaveCands = 64;  // candidates handled per pass
constexpr uint32_t kRing = 256;      // >= 64 (one round) + 127 (scan overshoot), power of two

// Pixels of one lane: 4 horizontally adjacent pixels (x0 .. x0+3, same y).
struct PixelState {
    half2_t r01, r23, g01, g23, b01, b23;  // half3 rgb (PietRender.metal:470), packed
    float df[4];                           // :471
    _Float16 sa[4];                        // half signedArea (:472)
};

// f32 -> binary16 of a value that is the result of f32 arithmetic.  The value is pinned in a
// register first: otherwise instruction selection folds `half(a * b)` (and friends) into
// v_fma_mixlo_f16, which rounds the exact result ONCE to binary16 -- not the f32 rounding followed
// by the conversion that the source (and the reference's `half(...)` casts, decision D1) specify.
// Measured on gfx950: 958 of 16.7 M random products differ (tools/probes/mix_probe.hip); a
// 2 000-scene fuzz run found two pixels off by one because of it.
__device__ __forceinline__ _Float16 ToHalf(float x) {
    asm volatile("" : "+v"(x));
    return static_cast<_Float16>(x);
}

__device__ __forceinline__ _Float16 HalfFromBits(uint32_t b) {
    const uint16_t u = static_cast<uint16_t>(b);
    return __builtin_bit_cast(_Float16, u);
}

__device__ __forceinline__ half2_t Splat(_Float16 v) { half2_t r; r.x = v; r.y = v; return r; }

// rgb = mix(rgb, fg.rgb, fg.a * alpha) per pixel (:505, :543, :549): x + (y - x) * a in half
__device__ __forceinline__ void Blend4(PixelState &st, uint32_t rg, uint32_t ba, const _Float16 alpha[4]) {
    const _Float16 fga = HalfFromBits(ba >> 16);
    half2_t a01, a23;
    a01.x = fga * alpha[0]; a01.y = fga * alpha[1];
    a23.x = fga * alpha[2]; a23.y = fga * alpha[3];
    const half2_t fr = Splat(HalfFromBits(rg)), fg = Splat(HalfFromBits(rg >> 16)), fb = Splat(HalfFromBits(ba));
    st.r01 = st.r01 + (fr - st.r01) * a01; st.r23 = st.r23 + (fr - st.r23) * a23;
    st.g01 = st.g01 + (fg - st.g01) * a01; st.g23 = st.g23 + (fg - st.g23) * a23;
    st.b01 = st.b01 + (fb - st.b01) * a01; st.b23 = st.b23 + (fb - st.b23) * a23;
}

// renderKernel's command loop (PietRender.metal:474-560) over an LDS-resident list.
// px0 = x of the lane's first pixel, py = its row.
__device__ __forceinline__ void Interpret(const Cmd *cmds, uint32_t n, float px0, float py, PixelState &st) {
    for (uint32_t i = 0; i < n; ++i) {
        const Cmd cmd = cmds[i];
        switch (cmd.tag) {
            case kCmdCircle: {  // :481-494
                const float x0 = static_cast<float>(cmd.body[1] & 0xffffu), y0 = static_cast<float>(cmd.body[1] >> 16);
                const float x1 = static_cast<float>(cmd.body[2] & 0xffffu), y1 = static_cast<float>(cmd.body[2] >> 16);
                const float cx = x0 + (x1 - x0) * 0.5f, cy = y0 + (y1 - y0) * 0.5f;
                const float circle_r = fminf(cx - x0, cy - y0);
                const float dy = py - cy;
                _Float16 alpha[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dx = (px0 + static_cast<float>(k)) - cx;
                    const float r = sqrtf(dx * dx + dy * dy);
                    alpha[k] = ToHalf(Sat(circle_r - r));
                }
                const half2_t zero = Splat(static_cast<_Float16>(0.0f));
                half2_t a01, a23;
                a01.x = alpha[0]; a01.y = alpha[1]; a23.x = alpha[2]; a23.y = alpha[3];
                st.r01 = st.r01 + (zero - st.r01) * a01; st.r23 = st.r23 + (zero - st.r23) * a23;
                st.g01 = st.g01 + (zero - st.g01) * a01; st.g23 = st.g23 + (zero - st.g23) * a23;
                st.b01 = st.b01 + (zero - st.b01) * a01; st.b23 = st.b23 + (zero - st.b23) * a23;
                break;
            }
            case kCmdLine: {  // stroke(), :49-55
                const float sx = __uint_as_float(cmd.body[1]), sy = __uint_as_float(cmd.body[2]);
                const float ex = __uint_as_float(cmd.body[3]), ey = __uint_as_float(cmd.body[4]);
                const float lx = ex - sx, ly = ey - sy;
                const float den = lx * lx + ly * ly;
                const float dy = py - sy;
                const float lydy = ly * dy;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dx = (px0 + static_cast<float>(k)) - sx;
                    const float t = Sat((lx * dx + lydy) / den);
                    const float fx = lx * t - dx, fy = ly * t - dy;
                    st.df[k] = fminf(st.df[k], sqrtf(fx * fx + fy * fy));
                }
                break;
            }
            case kCmdStroke: {  // :500-507, renderDf :58-60
                const float half_width = __uint_as_float(cmd.body[0]);
                _Float16 alpha[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    alpha[k] = ToHalf(Sat(half_width + 0.5f - st.df[k]));
                    st.df[k] = 1e9f;
                }
                Blend4(st, cmd.body[2], cmd.body[3], alpha);
                break;
            }
            case kCmdFill: {  // :508-529
                const float fsx = __uint_as_float(cmd.body[1]), fex = __uint_as_float(cmd.body[3]);
                const float sy = __uint_as_float(cmd.body[2]) - py;
                const float ey = __uint_as_float(cmd.body[4]) - py;
                const float wx = Sat(sy), wy = Sat(ey);
                if (wx != wy) {  // depends on y only: uniform over the lane's 4 pixels
                    const float tx = (wx - sy) / (ey - sy);
                    const float ty = (wy - sy) / (ey - sy);
                    const float wd = wx - wy;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float px = px0 + static_cast<float>(k);
                        const float sx = fsx - px, ex = fex - px;
                        const float xsx = sx + (ex - sx) * tx;
                        const float xsy = sx + (ex - sx) * ty;
                        const float xmin = fminf(fminf(xsx, xsy), 1.0f) - 1e-6f;
                        const float xmax = fmaxf(xsx, xsy);
                        const float b = fminf(xmax, 1.0f);
                        const float c = fmaxf(b, 0.0f);
                        const float d = fmaxf(xmin, 0.0f);
                        const float area = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
                        st.sa[k] = st.sa[k] + ToHalf(area * wd);
                    }
                }
                break;
            }
            case kCmdFillEdge: {  // :530-534 (half + float => f32 add, one rounding)
                const float sgn = static_cast<float>(static_cast<int>(cmd.body[0]));
                const float v = sgn * Sat(py - __uint_as_float(cmd.body[1]) + 1.0f);
#pragma unroll
                for (int k = 0; k < 4; ++k) st.sa[k] = ToHalf(static_cast<float>(st.sa[k]) + v);
                break;
            }
            case kCmdDrawFill: {  // :535-545
                const _Float16 bd = static_cast<_Float16>(static_cast<float>(static_cast<int>(cmd.body[0])));
                _Float16 alpha[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const _Float16 a = st.sa[k] + bd;
                    alpha[k] = ToHalf(fminf(fabsf(static_cast<float>(a)), 1.0f));
                    st.sa[k] = static_cast<_Float16>(0.0f);
                }
                Blend4(st, cmd.body[2], cmd.body[3], alpha);
                break;
            }
            case kCmdSolid: {  // :546-551
                const _Float16 one = static_cast<_Float16>(1.0f);
                const _Float16 alpha[4] = {one, one, one, one};
                Blend4(st, cmd.body[1], cmd.body[2], alpha);
                break;
            }
            default:
                break;
        }
    }
}


// ---- quarter-tile mode: one pixel per lane ---------------------------------------------
// Tiles with long command lists are rendered by four waves (4 pixel rows each): the list
// is walked in order by every wave, but a lone pixel per lane leaves the divide/area chains
// without instruction-level parallelism, so runs of consecutive Fill commands are evaluated
// four at a time (independent chains) and only ACCUMULATED in list order.
struct PixelState1 {
    _Float16 r, g, b;
    float df;
    _Float16 sa;
};

__device__ __forceinline__ void Blend1(PixelState1 &st, uint32_t rg, uint32_t ba, _Float16 alpha) {
    const _Float16 fa = HalfFromBits(ba >> 16) * alpha;
    const _Float16 fr = HalfFromBits(rg), fg = HalfFromBits(rg >> 16), fb = HalfFromBits(ba);
    st.r = st.r + (fr - st.r) * fa;
    st.g = st.g + (fg - st.g) * fa;
    st.b = st.b + (fb - st.b) * fa;
}

__device__ __forceinline__ void Interpret1(const Cmd *cmds, uint32_t n, float px, float py, PixelState1 &st) {
    for (uint32_t i = 0; i < n; ++i) {
        const Cmd cmd = cmds[i];
        switch (cmd.tag) {
            case kCmdCircle: {
                const float x0 = static_cast<float>(cmd.body[1] & 0xffffu), y0 = static_cast<float>(cmd.body[1] >> 16);
                const float x1 = static_cast<float>(cmd.body[2] & 0xffffu), y1 = static_cast<float>(cmd.body[2] >> 16);
                const float cx = x0 + (x1 - x0) * 0.5f, cy = y0 + (y1 - y0) * 0.5f;
                const float dx = px - cx, dy = py - cy;
                const float r = sqrtf(dx * dx + dy * dy);
                const _Float16 alpha = ToHalf(Sat(fminf(cx - x0, cy - y0) - r));
                const _Float16 zero = static_cast<_Float16>(0.0f);
                st.r = st.r + (zero - st.r) * alpha;
                st.g = st.g + (zero - st.g) * alpha;
                st.b = st.b + (zero - st.b) * alpha;
                break;
            }
            case kCmdLine: {
                const float sx = __uint_as_float(cmd.body[1]), sy = __uint_as_float(cmd.body[2]);
                const float ex = __uint_as_float(cmd.body[3]), ey = __uint_as_float(cmd.body[4]);
                const float lx = ex - sx, ly = ey - sy;
                const float dx = px - sx, dy = py - sy;
                const float t = Sat((lx * dx + ly * dy) / (lx * lx + ly * ly));
                const float fx = lx * t - dx, fy = ly * t - dy;
                st.df = fminf(st.df, sqrtf(fx * fx + fy * fy));
                break;
            }
            case kCmdStroke: {
                const _Float16 alpha = ToHalf(Sat(__uint_as_float(cmd.body[0]) + 0.5f - st.df));
                Blend1(st, cmd.body[2], cmd.body[3], alpha);
                st.df = 1e9f;
                break;
            }
            case kCmdFill: {
                uint32_t run = 1;
                while (run < 4u && i + run < n && cmds[i + run].tag == kCmdFill) ++run;
                float sy[4], ey[4], wx[4], wy[4], fsx[4], fex[4];
                bool live[4];
                bool any_live = false;
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) {
                    const Cmd cu = cmds[min(i + u, n - 1u)];
                    fsx[u] = __uint_as_float(cu.body[1]);
                    fex[u] = __uint_as_float(cu.body[3]);
                    sy[u] = __uint_as_float(cu.body[2]) - py;
                    ey[u] = __uint_as_float(cu.body[4]) - py;
                    wx[u] = Sat(sy[u]);
                    wy[u] = Sat(ey[u]);
                    live[u] = (u < run) && (wx[u] != wy[u]);
                    any_live = any_live || live[u];
                }
                if (any_live) {
                    float contrib[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {  // four independent chains
                        const float tx = (wx[u] - sy[u]) / (ey[u] - sy[u]);
                        const float ty = (wy[u] - sy[u]) / (ey[u] - sy[u]);
                        const float sx = fsx[u] - px, ex = fex[u] - px;
                        const float xsx = sx + (ex - sx) * tx;
                        const float xsy = sx + (ex - sx) * ty;
                        const float xmin = fminf(fminf(xsx, xsy), 1.0f) - 1e-6f;
                        const float xmax = fmaxf(xsx, xsy);
                        const float b = fminf(xmax, 1.0f);
                        const float c = fmaxf(b, 0.0f);
                        const float d = fmaxf(xmin, 0.0f);
                        const float area = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
                        contrib[u] = area * (wx[u] - wy[u]);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u)  // accumulate in list order (half adds do not commute)
                        if (live[u]) st.sa = st.sa + ToHalf(contrib[u]);
                }
                i += run - 1u;
                break;
            }
            case kCmdFillEdge: {
                const float sgn = static_cast<float>(static_cast<int>(cmd.body[0]));
                const float v = sgn * Sat(py - __uint_as_float(cmd.body[1]) + 1.0f);
                st.sa = ToHalf(static_cast<float>(st.sa) + v);
                break;
            }
            case kCmdDrawFill: {
                _Float16 alpha = st.sa + static_cast<_Float16>(static_cast<float>(static_cast<int>(cmd.body[0])));
                alpha = ToHalf(fminf(fabsf(static_cast<float>(alpha)), 1.0f));
                Blend1(st, cmd.body[2], cmd.body[3], alpha);
                st.sa = static_cast<_Float16>(0.0f);
                break;
            }
            case kCmdSolid:
                Blend1(st, cmd.body[1], cmd.body[2], static_cast<_Float16>(1.0f));
                break;
            default:
                break;
        }
    }
}

struct CoarseLds {
    uint32_t ring[kRing];     // indices of the record's segments relevant to this tile
    uint8_t hidx[kThreads];   // candidates of the record that hit this tile (indices)
    uint32_t htag[kWaveCands];
    uint32_t hrgba[kWaveCands];
    uint32_t haux0[kWaveCands];
    uint32_t haux1[kWaveCands];
    uint32_t hrel[kWaveCands];   // relevant segments of the candidate in this tile
    uint32_t hwoff[kWaveCands];  // index of its first relevant segment (ring position)
    uint32_t hcnt[kWaveCands];   // stream elements (relevant segments, or 1 pseudo element)
    uint32_t hrg[kWaveCands];
    uint32_t hba[kWaveCands];
    uint32_t hoff[kWaveCands + 1];
    uint32_t own[64];            // stream position of a round -> candidate that starts there
    int backdrop[kWaveCands];
    uint32_t any[kWaveCands];
};

}  // namespace

template <bool kCapture>
__global__ __launch_bounds__(kThreads, PM_COARSE_WPS) void pm_coarse_kernel(FrameParams P) {
    __shared__ CoarseLds s_lds[kWaves];
    CoarseLds &L = s_lds[threadIdx.x >> 6];

    const uint32_t lane = LaneId();
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    const uint32_t n_a = P.ctr_cur->vheavy_count, n_b = P.ctr_cur->heavy_count;
    const uint32_t n_total = n_a + n_b + P.ctr_cur->light_count;

    // Static snake hand-out over [longest lists..., shortest...]: pass k gives wave g the slot
    // k*G + g (k even) or k*G + (G-1-g) (k odd).  No atomics: one device-scope counter tops out
    // near 90 dequeues/us on this chip, far below the tile rate.
    const uint32_t wave_global = blockIdx.x * kWaves + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * kWaves;

    for (uint32_t pass = 0; pass * n_waves < n_total; ++pass) {
        const uint32_t slot = pass * n_waves + ((pass & 1u) ? (n_waves - 1u - wave_global) : wave_global);
        if (slot >= n_total) continue;
        uint4 *const qentry = P.queue + ((slot < n_a) ? slot
                                         : (slot < n_a + n_b) ? P.queue_cap + (slot - n_a)
                                                              : 2u * P.queue_cap + (slot - n_a - n_b));
        const uint4 qe = *qentry;
        const uint32_t tile = qe.x;
        if (qe.y == 0xffffffffu) {  // the command-list arena overflowed (pm_sync re-renders the frame)
            if (lane == 0) qentry->w = 0;
            continue;
        }
        Cmd *const out_cmds = P.ptcl + qe.y;  // this tile's private command slots
        const uint32_t tx = tile % P.tiles_x;
        const uint32_t ty_rel = tile / P.tiles_x;
        const uint32_t ty = P.row0 + ty_rel;
        const int x0 = static_cast<int>(tx * kTileW);
        const int y0 = static_cast<int>(ty * kTileH);
        const float fx0 = static_cast<float>(x0), fy0 = static_cast<float>(y0);
        const float fx1 = static_cast<float>(x0 + static_cast<int>(kTileW));
        const float fy1 = static_cast<float>(y0 + static_cast<int>(kTileH));
        const uint32_t tbit = tx & (kStripTiles - 1);

        uint32_t solid_color = 0xffffffffu;  // TileEncoder::solidColor (:74)
        uint32_t n_pending = 0;              // commands written to the tile's list so far
        uint32_t list_len = 0;               // logical list length since tileBegin (capture)

        uint32_t rec = qe.z;  // (= striprow_head[sr])
        while (rec != 0) {
            // header and mask table sit next to each other: all loads are in flight together
            const uint4 hdr = *reinterpret_cast<const uint4 *>(P.arena + rec);
            const uint4 mk = *reinterpret_cast<const uint4 *>(P.arena + rec + kRecHdrDwords + 4u * lane);
            const uint32_t next = hdr.x;
            const uint32_t ncand = hdr.y;
            const uint32_t mask_dwords = (ncand + 3u) & ~3u;
            const uint32_t *cand_rec = P.arena + rec + kRecHdrDwords + mask_dwords;
            const uint32_t *ct_tab = cand_rec + kCandDwords * ncand;
            const float4 *segs = reinterpret_cast<const float4 *>(ct_tab + kCtDwords * ncand);
            const uint32_t *meta = reinterpret_cast<const uint32_t *>(segs + kChunkSegs * hdr.z);
            // Worklist of the segments that matter to THIS tile: the record's segment slots are
            // scanned linearly (2 meta words per lane per step, independent loads) and the slots
            // carrying this tile's bit are kept, in paint order, in a small LDS ring.
            const uint32_t n_slots = hdr.w;
            uint32_t scan_pos = 0;  // next segment to scan
            uint32_t ring_cnt = 0;  // relevant segments found so far (ring write position)
            uint32_t rel_done = 0;  // relevant segments owned by earlier candidate passes

            // ---- candidates that hit this tile, in paint order (lane owns 4 consecutive) ------
            const uint32_t mw[4] = {mk.x, mk.y, mk.z, mk.w};
            uint32_t hbits = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (4u * lane + k < ncand && ((mw[k] >> (16 + tbit)) & 1u)) hbits |= 1u << k;
            const uint32_t hcount = __popc(hbits);
            const uint32_t hincl = WaveInclusiveScan(hcount);
            const uint32_t nhit = WaveLast(hincl);
            if (nhit == 0) {
                rec = next;
                continue;
            }
            WaveSync();
            {
                uint32_t hp = hincl - hcount;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if ((hbits >> k) & 1u) L.hidx[hp++] = static_cast<uint8_t>(4u * lane + k);
            }
            WaveSync();

            for (uint32_t cb = 0; cb < nhit; cb += kWaveCands) {
                const uint32_t nh = min(kWaveCands, nhit - cb);
                if (lane < nh) {
                    const uint4 *cr = reinterpret_cast<const uint4 *>(cand_rec + kCandDwords * L.hidx[cb + lane]);
                    const uint4 a = cr[0];
                    const uint4 b = cr[1];
                    L.htag[lane] = a.x & 0xffffu;
                    L.hrgba[lane] = a.y;
                    L.haux0[lane] = a.z;
                    L.haux1[lane] = a.w;
                    const uint32_t ct = ct_tab[kCtDwords * L.hidx[cb + lane] + tbit];
                    const uint32_t rel = ct & kCtCountMask;
                    L.hrel[lane] = rel;
                    L.hcnt[lane] = rel ? rel : 1u;  // circle / backdrop-only fill: one pseudo element
                    L.hrg[lane] = b.z;
                    L.hba[lane] = b.w;
                    L.backdrop[lane] = static_cast<int>(ct) >> kCtShift;
                    L.any[lane] = 0;
                }
                WaveSync();
                uint32_t stream_len, pass_rel;
                {
                    const uint32_t v = (lane < nh) ? L.hcnt[lane] : 0u;
                    const uint32_t r = (lane < nh) ? L.hrel[lane] : 0u;
                    const uint32_t incl = WaveInclusiveScan(v);
                    const uint32_t rincl = WaveInclusiveScan(r);
                    if (lane < nh) {
                        L.hoff[lane] = incl - v;
                        L.hwoff[lane] = rel_done + rincl - r;  // first relevant segment of the candidate
                    }
                    stream_len = WaveLast(incl);
                    pass_rel = WaveLast(rincl);
                    if (lane == 0) L.hoff[nh] = stream_len;
                }
                WaveSync();

                // ---- stream rounds: phase-2 tests -> ordered commands --------------------------
                uint32_t own_carry = 0;  // owner of the last element of the previous round
                for (uint32_t e0 = 0; e0 < stream_len; e0 += 64) {
                    const uint32_t e = e0 + lane;
                    // Owner of every element without a search: each candidate marks the stream
                    // position where it starts, a prefix maximum spreads the marks (owners only
                    // grow along the stream).
                    L.own[lane] = 0;
                    WaveSync();
                    if (lane < nh) {
                        const uint32_t p = L.hoff[lane] - e0;
                        if (p < 64u) L.own[p] = lane;
                    }
                    WaveSync();
                    const uint32_t owner = max(WaveInclusiveMax(L.own[lane]), own_carry);
                    own_carry = WaveLast(owner);
                    uint32_t n_em = 0;   // commands of this stream element (0..2)
                    Cmd c0, c1;
                    c0.tag = 0; c1.tag = 0;
                    bool is_last = false;
                    bool draws = false;  // any of this lane's commands clears solidColor
                    uint32_t c = 0, ctag = 0;
                    if (e < stream_len) {
                        c = owner;
                        const uint32_t k = e - L.hoff[c];
                        is_last = (k + 1 == L.hcnt[c]);
                        ctag = L.htag[c];
                        if (ctag == kItemCircle) {  // :218-222
                            n_em = 1;
                            c0.tag = kCmdCircle;
                            c0.body[0] = 0;
                            c0.body[1] = L.haux0[c];
                            c0.body[2] = L.haux1[c];
                            c0.body[3] = 0;
                            c0.body[4] = 0;
                            draws = true;
                        }
                    }
                    // make sure the ring holds every relevant segment this round needs
                    {
                        const bool wants = (e < stream_len) && ctag != kItemCircle && L.hrel[c] != 0;
                        const uint64_t wm = __ballot(wants);
                        if (wm) {
                            const uint32_t my_need = wants ? (L.hwoff[c] + (e - L.hoff[c]) + 1u) : 0u;
                            const uint32_t need = WaveAtHighest(my_need, wm);
                            while (ring_cnt < need && scan_pos < n_slots) {
                                const uint32_t cnt_x = n_slots;
                                const uint32_t st_x = 0;
                                const uint32_t i0 = scan_pos + 2u * lane;
                                uint2 mv = make_uint2(0u, 0u);
                                if (i0 < cnt_x) mv = *reinterpret_cast<const uint2 *>(meta + st_x + i0);
                                const uint32_t ma[2] = {mv.x, mv.y};
                                uint32_t rb = 0;
#pragma unroll
                                for (uint32_t q = 0; q < 2; ++q)
                                    if (i0 + q < cnt_x && ((ma[q] >> tbit) & 1u)) rb |= 1u << q;
                                const uint32_t rc = __popc(rb);
                                const uint32_t rincl = WaveInclusiveScan(rc);
                                uint32_t wp = ring_cnt + rincl - rc;
#pragma unroll
                                for (uint32_t q = 0; q < 2; ++q)
                                    if ((rb >> q) & 1u) L.ring[(wp++) & (kRing - 1u)] = st_x + i0 + q;
                                ring_cnt += WaveLast(rincl);
                                scan_pos += 128u;
                            }
                            WaveSync();
                        }
                    }
                    if (e < stream_len) {
                        if (ctag == kItemFill && L.hrel[c] == 0) {
                            // backdrop-only fill: nothing to test, the closing command decides
                        } else if (ctag != kItemCircle) {
                            const uint32_t k = e - L.hoff[c];
                            const float4 s = segs[L.ring[(L.hwoff[c] + k) & (kRing - 1u)]];
                            const float a = s.w - s.y;
                            const float b = s.x - s.z;
                            const float cc = -(a * s.x + b * s.y);
                            if (ctag == kItemFill) {  // :302-357
                                const float xmin = fminf(s.x, s.z), ymin = fminf(s.y, s.w);
                                const float xmax = fmaxf(s.x, s.z), ymax = fmaxf(s.y, s.w);
                                const float left = a * fx0;
                                const float right = a * fx1;
                                const float ytop = fmaxf(fy0, ymin);
                                const float ybot = fminf(fy1, ymax);
                                const float top = b * ytop;
                                const float bot = b * ybot;
                                const float s00 = Sgn(top + left + cc);
                                const float s01 = Sgn(top + right + cc);
                                const float s10 = Sgn(bot + left + cc);
                                const float s11 = Sgn(bot + right + cc);
                                // (the backdrop term of :326-333 was summed by the binning kernel)
                                const bool straddle = Straddles(s00, s01, s10, s11);
                                if (xmin < fx0 && xmax > fx0) {
                                    const float tt = (s.x - fx0) / b;
                                    const float y_edge = s.y + (s.w - s.y) * tt;  // mix(start.y, end.y, tt)
                                    if (y_edge >= fy0 && y_edge < fy1) {
                                        n_em = 2;
                                        c0.tag = kCmdFillEdge;
                                        c0.body[0] = static_cast<uint32_t>(static_cast<int>(s00));
                                        c0.body[1] = __float_as_uint(y_edge);
                                        c0.body[2] = 0; c0.body[3] = 0; c0.body[4] = 0;
                                        c1.tag = kCmdFill;
                                        c1.body[0] = 0;
                                        if (b > 0.0f) {
                                            c1.body[1] = __float_as_uint(s.x); c1.body[2] = __float_as_uint(s.y);
                                            c1.body[3] = __float_as_uint(fx0); c1.body[4] = __float_as_uint(y_edge);
                                        } else {
                                            c1.body[1] = __float_as_uint(fx0); c1.body[2] = __float_as_uint(y_edge);
                                            c1.body[3] = __float_as_uint(s.z); c1.body[4] = __float_as_uint(s.w);
                                        }
                                    } else if (straddle) {
                                        n_em = 1;
                                    }
                                } else if (straddle && xmin < fx1 && xmax > fx0) {
                                    n_em = 1;
                                }
                                if (n_em == 1) {
                                    c0.tag = kCmdFill;
                                    c0.body[0] = 0;
                                    c0.body[1] = __float_as_uint(s.x); c0.body[2] = __float_as_uint(s.y);
                                    c0.body[3] = __float_as_uint(s.z); c0.body[4] = __float_as_uint(s.w);
                                }
                                if (n_em) atomicOr(&L.any[c], 1u);
                            } else {
                                // Line (:223-247) and Poly phase 2 (:406-440) share the inflated-box test
                                const float width = __uint_as_float(L.haux0[c]);
                                const float hw = 0.5f * width + 0.5f;
                                bool pass = true;
                                if (ctag == kItemPoly) {
                                    const float xmin = fminf(s.x, s.z), ymin = fminf(s.y, s.w);
                                    const float xmax = fmaxf(s.x, s.z), ymax = fmaxf(s.y, s.w);
                                    pass = ymax > fy0 - hw && ymin < fy1 + hw && xmax > fx0 - hw && xmin < fx1 + hw;
                                }
                                if (pass) {
                                    const float left = a * (fx0 - hw);
                                    const float right = a * (fx1 + hw);
                                    const float top = b * (fy0 - hw);
                                    const float bot = b * (fy1 + hw);
                                    const float s00 = Sgn(top + left + cc);
                                    const float s01 = Sgn(top + right + cc);
                                    const float s10 = Sgn(bot + left + cc);
                                    const float s11 = Sgn(bot + right + cc);
                                    pass = Straddles(s00, s01, s10, s11);
                                }
                                if (pass) {
                                    n_em = 1;
                                    c0.tag = kCmdLine;
                                    c0.body[0] = 0;
                                    c0.body[1] = __float_as_uint(s.x); c0.body[2] = __float_as_uint(s.y);
                                    c0.body[3] = __float_as_uint(s.z); c0.body[4] = __float_as_uint(s.w);
                                    draws = true;
                                    atomicOr(&L.any[c], 1u);
                                }
                            }
                        }
                    }
                    WaveSync();  // per-candidate accumulators complete for elements <= this round

                    // ---- per-item closing command (DrawFill / Solid / Stroke) --------------------
                    bool has_fin = false;
                    bool opaque_solid = false;
                    Cmd fin;
                    fin.tag = 0;
                    fin.body[0] = fin.body[1] = fin.body[2] = fin.body[3] = fin.body[4] = 0;
                    if (is_last) {
                        const uint32_t frgba = L.hrgba[c];
                        const uint32_t rg = L.hrg[c], ba = L.hba[c];
                        if (ctag == kItemFill) {  // :359-363
                            const int backdrop = L.backdrop[c];
                            if (L.any[c]) {
                                has_fin = true;
                                fin.tag = kCmdDrawFill;
                                fin.body[0] = static_cast<uint32_t>(backdrop);
                                fin.body[1] = frgba; fin.body[2] = rg; fin.body[3] = ba; fin.body[4] = 0;
                                draws = true;
                            } else if (backdrop != 0) {
                                has_fin = true;
                                fin.tag = kCmdSolid;
                                fin.body[0] = frgba; fin.body[1] = rg; fin.body[2] = ba; fin.body[3] = 0; fin.body[4] = 0;
                                opaque_solid = (frgba & 0xff000000u) == 0xff000000u;  // :132
                            }
                        } else if (ctag == kItemPoly || ctag == kItemLine) {  // :441-443, :243
                            if (L.any[c]) {
                                has_fin = true;
                                fin.tag = kCmdStroke;
                                fin.body[0] = __float_as_uint(0.5f * __uint_as_float(L.haux0[c]));
                                fin.body[1] = frgba; fin.body[2] = rg; fin.body[3] = ba; fin.body[4] = 0;
                                draws = true;
                            }
                        }
                    }
                    const uint32_t lane_total = n_em + (has_fin ? 1u : 0u);  // 0..3

                    // ---- wave-wide slots (ballots + popcounts, no scan network) ---------------------
                    const uint64_t m0 = __ballot((lane_total & 1u) != 0);
                    const uint64_t m1 = __ballot((lane_total & 2u) != 0);
                    const uint32_t pos = static_cast<uint32_t>(__popcll(m0 & lanes_below)) + 2u * static_cast<uint32_t>(__popcll(m1 & lanes_below));
                    const uint32_t round_total = static_cast<uint32_t>(__popcll(m0)) + 2u * static_cast<uint32_t>(__popcll(m1));
                    if (round_total == 0) continue;  // uniform
                    const uint64_t ms = __ballot(opaque_solid);
                    const uint64_t md = __ballot(draws);
                    int last_solid = -1, last_draw = -1;
                    if (ms) last_solid = static_cast<int>(WaveAtHighest(pos + n_em, ms));
                    if (md) last_draw = static_cast<int>(WaveAtHighest(pos + lane_total, md)) - 1;

                    uint32_t base;        // list slot of round position 0 (may be "negative")
                    uint32_t first_kept;  // round positions below this are dropped
                    if (last_solid >= 0) {
                        // TileEncoder::encodeSolid with an opaque colour (:132-135): dst = tileBegin
                        first_kept = static_cast<uint32_t>(last_solid);
                        n_pending = 0;
                        list_len = 0;
                        base = 0u - first_kept;
                    } else {
                        first_kept = 0;
                        base = n_pending;
                    }
                    {
                        uint32_t p = pos;
                        if (n_em >= 1) {
                            if (p >= first_kept) {
                                out_cmds[base + p] = c0;
                                if (kCapture) {
                                    const uint32_t li = list_len + p - first_kept;
                                    if (li < P.dbg_max) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + li] = c0;
                                }
                            }
                            ++p;
                        }
                        if (n_em == 2) {
                            if (p >= first_kept) {
                                out_cmds[base + p] = c1;
                                if (kCapture) {
                                    const uint32_t li = list_len + p - first_kept;
                                    if (li < P.dbg_max) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + li] = c1;
                                }
                            }
                            ++p;
                        }
                        if (has_fin && p >= first_kept) {
                            out_cmds[base + p] = fin;
                            if (kCapture) {
                                const uint32_t li = list_len + p - first_kept;
                                if (li < P.dbg_max) {
                                    // capture in the reference's layout: no pre-converted colour words
                                    Cmd ref = fin;
                                    if (fin.tag == kCmdSolid) { ref.body[1] = 0; ref.body[2] = 0; }
                                    else { ref.body[2] = 0; ref.body[3] = 0; }
                                    P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + li] = ref;
                                }
                            }
                        }
                    }
                    n_pending = base + round_total;
                    list_len += round_total - first_kept;
                    if (last_solid >= 0) solid_color = WaveAtHighest(fin.body[0], ms);
                    if (last_draw > last_solid) solid_color = 0;  // encodeCircle/Line/Stroke/DrawFill (:81,:90,:99,:124)
                    WaveSync();
                }
                rel_done += pass_rel;
            }
            rec = next;
        }

        // ---- TileEncoder::end() (:144-151): Bail tiles are finished here (composite :34-44) ----
        if (lane == 0) {
            P.tile_ncmd[tile] = solid_color ? 0u : n_pending;
            qentry->w = solid_color ? 0u : n_pending;  // what pm_fine_kernel reads
        }
        if (solid_color != 0) {
            // the tile is one opaque colour, bytes as stored: 64 lanes x 16 B = the whole tile
            const uint32_t pxi = static_cast<uint32_t>(x0) + (lane & 3u) * 4u;
            const uint32_t prow = lane >> 2;
            const uint32_t pyi = static_cast<uint32_t>(y0) + prow;
            if (pyi < P.height && pxi < P.width) {
                uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + prow) * P.fb_stride + static_cast<size_t>(pxi) * 4;
                if (pxi + 4 <= P.width && P.fb_vec16) {
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(solid_color, solid_color, solid_color, solid_color);
                } else {
                    for (uint32_t k = 0; k < 4 && pxi + k < P.width; ++k) reinterpret_cast<uint32_t *>(dst)[k] = solid_color;
                }
            }
        }
        if (kCapture && lane == 0) {
            // list as the reference leaves it: {Bail} or cmds + End
            P.dbg_solid[tile] = solid_color;
            Cmd tail;
            tail.body[0] = tail.body[1] = tail.body[2] = tail.body[3] = tail.body[4] = 0;
            if (solid_color != 0) {
                P.dbg_counts[tile] = 1;
                tail.tag = kCmdBail;
                if (P.dbg_max > 0) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max] = tail;
            } else {
                P.dbg_counts[tile] = list_len + 1;
                tail.tag = kCmdEnd;
                if (list_len < P.dbg_max) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + list_len] = tail;
            }
        }
        WaveSync();  // L reuse by the next tile
    }
}

// ---- launch wrappers (called from pm_context.hip) -----------------------------------------

// =====================================================================================
// K3: per-pixel interpreter (renderKernel :457-566) over the per-tile command lists
// =====================================================================================
// Light tiles: one wave per tile, 4 adjacent pixels per lane.  Tiles with long lists: four
// waves per tile (4 pixel rows each, 1 pixel per lane, Fill runs evaluated 4 at a time).
// The list is staged through LDS in chunks with coalesced loads; interpreter state stays
// in registers across chunks.
__global__ __launch_bounds__(kThreads, PM_FINE_WPS) void pm_fine_kernel(FrameParams P) {
    __shared__ Cmd s_cmds[kWaves][kFineChunk];
    Cmd *const cmds = s_cmds[threadIdx.x >> 6];

    // Workgroups beyond the persistent grid write the pixels of the tiles binning resolved (see
    // pm_clear_kernel): pure stores that fill the SIMDs this kernel's long tail leaves idle, and
    // one launch less per frame.
    if (blockIdx.x >= P.fine_grid) {
        ClearStripRow(P, blockIdx.x - P.fine_grid);
        return;
    }
    const uint32_t lane = LaneId();
    const uint32_t n_a = P.ctr_cur->vheavy_count, n_b = P.ctr_cur->heavy_count, n_c = P.ctr_cur->light_count;
    const uint32_t wave_global = blockIdx.x * kWaves + (threadIdx.x >> 6);
    const uint32_t n_waves = P.fine_grid * kWaves;
    // slots: 4 per tile with a long list (16 for the very long ones in split mode 2), 1 per light tile.
    // Splitting a tile buys latency when few long lists set the span of the launch; with more
    // long lists than waves it only costs work (the y-only math is no longer shared by 4
    // pixels), so dense frames render every tile with one wave.
    const bool dense = n_a + n_b >= n_waves || P.split_mode == 0;
    const bool split4_only = P.split_mode == 1;  // (default) never 16 waves per tile: measured no faster than 4
    const uint32_t sh_a = dense ? 0u : (split4_only ? 2u : 4u), sh_b = dense ? 0u : 2u;
    const uint32_t s_a = n_a << sh_a, s_b = n_b << sh_b;
    const uint32_t n_slots = s_a + s_b + n_c;
    const uint8_t *lut = P.lut_lin2srgb;
    // linear -> sRGB + unorm8 (:563-565): the 65,536-entry table of decision D2.  (A compact
    // LDS-resident form of the table was measured slower: this kernel is bound by instruction
    // issue, and twelve byte loads per lane are fewer instructions than twelve decodes.)
    auto enc = [&](_Float16 r, _Float16 g, _Float16 b) -> uint32_t {
        return static_cast<uint32_t>(lut[__builtin_bit_cast(uint16_t, r)]) |
               (static_cast<uint32_t>(lut[__builtin_bit_cast(uint16_t, g)]) << 8) |
               (static_cast<uint32_t>(lut[__builtin_bit_cast(uint16_t, b)]) << 16) | 0xff000000u;
    };
    // slot -> queue entry index and the rows of the tile this wave renders: [row0, row0 + nrows)
    auto slot_entry = [&](uint32_t slot, uint32_t &row0, uint32_t &nrows) -> uint32_t {
        if (slot < s_a) {
            nrows = 16u >> sh_a;
            row0 = (slot & ((1u << sh_a) - 1u)) * nrows;
            return slot >> sh_a;
        }
        if (slot < s_a + s_b) {
            nrows = 16u >> sh_b;
            row0 = ((slot - s_a) & ((1u << sh_b) - 1u)) * nrows;
            return P.queue_cap + ((slot - s_a) >> sh_b);
        }
        row0 = 0;
        nrows = 16;
        return 2u * P.queue_cap + (slot - s_a - s_b);
    };
    auto pass_slot = [&](uint32_t pass) -> uint32_t {
        return pass * n_waves + ((pass & 1u) ? (n_waves - 1u - wave_global) : wave_global);
    };

    // The queue entry {tile, first command slot, -, commands} of the NEXT slot is fetched while
    // the current tile is interpreted: one exposed round trip per tile (the command list) instead
    // of three dependent ones.
    uint32_t slot = pass_slot(0);
    uint4 qe = make_uint4(0u, 0u, 0u, 0u);
    {
        uint32_t r0_, nr_;
        if (slot < n_slots) qe = P.queue[slot_entry(slot, r0_, nr_)];
    }
    for (uint32_t pass = 0; pass * n_waves < n_slots; ++pass) {
        const uint32_t cur_slot = slot;
        const uint4 cur = qe;
        slot = pass_slot(pass + 1u);
        {
            uint32_t r0_, nr_;
            if ((pass + 1u) * n_waves < n_slots && slot < n_slots) qe = P.queue[slot_entry(slot, r0_, nr_)];
        }
        if (cur_slot >= n_slots) continue;
        uint32_t row0, nrows;
        (void)slot_entry(cur_slot, row0, nrows);
        const uint32_t tile = cur.x;
        const bool quarter = nrows != 16u;  // one pixel per lane (lanes beyond nrows*16 idle)
        unsigned long long t_begin = 0;
        if (P.dbg_time) t_begin = wall_clock64();
        const uint32_t n_cmd = cur.w;
        if (n_cmd != 0) {  // 0: the coarse kernel found one opaque colour and wrote it
            const uint32_t *src = reinterpret_cast<const uint32_t *>(P.ptcl + cur.y);
            const uint32_t tx = tile % P.tiles_x;
            const uint32_t ty_rel = tile / P.tiles_x;
            const uint32_t x0 = tx * kTileW;
            const uint32_t y0 = (P.row0 + ty_rel) * kTileH;
            // whole-tile mode: lane -> 4 pixels, x = x0 + 4*(lane&3) + k, y = y0 + lane/4
            // split mode:      lane -> 1 pixel,  x = x0 + (lane&15),    y = y0 + row0 + lane/16
            const uint32_t pxi = x0 + (quarter ? (lane & 15u) : (lane & 3u) * 4u);
            const uint32_t prow = quarter ? (row0 + (lane >> 4)) : (lane >> 2);
            const uint32_t pyi = y0 + prow;
            const bool lane_on = !quarter || (lane >> 4) < nrows;
            const float px0 = static_cast<float>(pxi), py = static_cast<float>(pyi);
            PixelState1 s1;
            s1.r = s1.g = s1.b = static_cast<_Float16>(1.0f);
            s1.df = 1e9f;
            s1.sa = static_cast<_Float16>(0.0f);
            PixelState st;
            st.r01 = st.r23 = st.g01 = st.g23 = st.b01 = st.b23 = Splat(static_cast<_Float16>(1.0f));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                st.df[k] = 1e9f;
                st.sa[k] = static_cast<_Float16>(0.0f);
            }
            for (uint32_t c0 = 0; c0 < n_cmd; c0 += kFineChunk) {
                const uint32_t m = min(kFineChunk, n_cmd - c0);
                WaveSync();
                // 24-byte commands, 8-byte aligned: copy as 64-bit words, coalesced
                const uint2 *g = reinterpret_cast<const uint2 *>(src + 6u * c0);
                uint2 *l = reinterpret_cast<uint2 *>(cmds);
                for (uint32_t w = lane; w < 3u * m; w += 64u) l[w] = g[w];
                WaveSync();
                if (quarter) Interpret1(cmds, m, px0, py, s1);
                else Interpret(cmds, m, px0, py, st);
            }
            uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + prow) * P.fb_stride + static_cast<size_t>(pxi) * 4;
            if (quarter) {
                if (lane_on && pyi < P.height && pxi < P.width) *reinterpret_cast<uint32_t *>(dst) = enc(s1.r, s1.g, s1.b);
            } else if (pyi < P.height && pxi < P.width) {
                uint4 out;
                out.x = enc(st.r01.x, st.g01.x, st.b01.x);
                out.y = enc(st.r01.y, st.g01.y, st.b01.y);
                out.z = enc(st.r23.x, st.g23.x, st.b23.x);
                out.w = enc(st.r23.y, st.g23.y, st.b23.y);
                if (pxi + 4 <= P.width && P.fb_vec16) {
                    *reinterpret_cast<uint4 *>(dst) = out;
                } else {
                    const uint32_t o[4] = {out.x, out.y, out.z, out.w};
                    for (uint32_t k = 0; k < 4 && pxi + k < P.width; ++k) reinterpret_cast<uint32_t *>(dst)[k] = o[k];
                }
            }
        }
        if (P.dbg_time && lane == 0) {
            unsigned long long *d = P.dbg_time + 4ull * cur_slot;
            d[0] = t_begin;
            d[1] = wall_clock64();
            d[2] = tile | (quarter ? 0x80000000u : 0u);
            d[3] = (static_cast<unsigned long long>(wave_global) << 32) | n_cmd;
        }
    }
}

// ---- launch wrappers (called from pm_context.hip) -----------------------------------------

void LaunchIndex(const uint8_t *scene, uint32_t n_items, const uint32_t *chunk_base, uint32_t n_chunks, float4 *chunk_bbox,
                 hipStream_t stream) {
    if (n_chunks == 0) return;
    hipLaunchKernelGGL(pm_index_kernel, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, scene, n_items, chunk_base, n_chunks,
                       chunk_bbox);
}

// Launch wrappers.  With (t0, t1) the dispatch itself carries the two events
// (hipExtLaunchKernelGGL): their timestamps are the dispatch's own begin / end, what a
// kernel trace shows, and no extra packet goes on the queue.
#define PM_LAUNCH(kernel, grid, block, stream, t0, t1, ...)                                        \
    do {                                                                                           \
        if (t0)                                                                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, t0, t1, 0, __VA_ARGS__);         \
        else                                                                                       \
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                       \
    } while (0)

void LaunchBin(const FrameParams &p, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    const uint32_t n_striprows = p.n_sr_active;
    if (p.use_row_lists) hipLaunchKernelGGL(pm_rowcull_kernel, dim3(p.row1 - p.row0), dim3(kBinThreads), 0, stream, p);
    if (p.dbg_bin)
        PM_LAUNCH(pm_bin_kernel<true>, dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, p);
    else
        PM_LAUNCH(pm_bin_kernel<false>, dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, p);
}

void LaunchClear(const FrameParams &p, uint32_t n_striprows, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    PM_LAUNCH(pm_clear_kernel, dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, p);
}

void LaunchCoarse(const FrameParams &p, uint32_t grid, bool capture, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    if (capture)
        PM_LAUNCH(pm_coarse_kernel<true>, dim3(grid), dim3(kThreads), stream, t0, t1, p);
    else
        PM_LAUNCH(pm_coarse_kernel<false>, dim3(grid), dim3(kThreads), stream, t0, t1, p);
}

void LaunchFine(const FrameParams &p, uint32_t clear_blocks, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    PM_LAUNCH(pm_fine_kernel, dim3(p.fine_grid + clear_blocks), dim3(kThreads), stream, t0, t1, p);
}

#undef PM_LAUNCH

}  // namespace pm
