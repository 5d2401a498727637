// Shared by pm_bin.hip, pm_coarse.hip and pm_fine.hip -- hand-written HIP kernels for gfx950
// (CDNA4, wave64): the compute path of piet-metal re-designed for MI355X.
//
// Reference semantics being reproduced (bit-exact against oracle/):
//   tileKernel   TestApp/PietRender.metal:160-454  (+ TileEncoder :69-157)
//   renderKernel TestApp/PietRender.metal:457-566  (+ stroke/renderDf :49-60)
//   composite    TestApp/PietRender.metal:16-44
//
// Decomposition (NOT the reference's thread-per-tile / 256 MiB tile buffer) -- a frame is TWO
// launches, back to back on the frame's stream (pm_context.hip):
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
//       - per (item, tile) counts and backdrops; then, per tile, ONE contiguous PIECE in the tile
//         arena: the tile's relevant segments and the candidates that can emit there, both in
//         paint order (pm_device.h) -- all a tile needs from the record, nothing it does not;
//       - the TileEncoder solid rule per tile, the command-list space of every tile and its place
//         in one of eight cost-class queues.
//   pm_fine_kernel<fused>   persistent; per queued tile, by the wave(s) that will render it:
//       - CoarseTile (pm_coarse_tile.h; ONE WAVE, no workgroup barriers): walks the tile's pieces
//         -- one batch of loads each --; each lane runs the reference's "phase 2" test for (tile,
//         segment) (:302-357, :406-440) and emits 0..3 commands; ballots / mbcnt prefix ranks give
//         every command its slot in the tile's command list in HBM (the reference's 24-byte Cmd
//         records); opaque-solid detection (TileEncoder::encodeSolid/end) restarts the list; Bail
//         tiles are written as one constant there;
//       - renderKernel over the list for the tile's 256 pixels: one wave (4 px per lane, row-sparse
//         Fill evaluation), or the workgroup's four waves for a long list (items evaluated in
//         parallel into binary16 alpha images, blended in list order).  Accumulators are binary16
//         exactly where the source declares `half`; commands are applied in list order (half
//         accumulation is order dependent);
//       - extra workgroups of the same launch write the pixels of the tiles binning resolved
//         (background / one opaque colour): the composite for tiles that never get a list.
//   pm_coarse_kernel  CoarseTile as a launch of its own (PM_FUSED=0, pm_fill_coverage);
//       pm_clear_kernel likewise (PM_FOLD_CLEAR=0).  pm_debug_capture_ptcl captures the lists from
//       whichever kernel built them (the fused tile kernel's kCapture instantiation by default).
//
// Compile with -ffp-contract=off: every source-level f32/f16 operation is one
// IEEE rounding, as in the oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstddef>
#include <utility>

#include "pm_device.h"
#include <pm_pin.h>  // gfx950/pm_pin.h: register pins (GCN asm constraints)

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

// ---------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------

// wave index in the workgroup, as a scalar (the compiler cannot prove threadIdx.x >> 6 uniform)
__device__ __forceinline__ uint32_t WaveId() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// a 16-byte value every lane loaded from the same address, moved to scalar registers
__device__ __forceinline__ uint4 Scalar4(uint4 v) {
    return make_uint4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                      __builtin_amdgcn_readfirstlane(v.w));
}

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

// Workgroup barrier that orders LDS traffic ONLY: __syncthreads() also waits for every global store
// the wave has in flight (s_waitcnt vmcnt(0): their acknowledgements take microseconds), which a
// barrier between phases that hand each other LDS data does not need.  Global data written before it
// must not be read by ANOTHER wave after it.
// PM_STRICT_BARRIERS (libpiet_metal_amd_strict.so, tests only): every one of them is a full __syncthreads().
// If a phase ever comes to read global data another wave wrote before such a barrier, the two builds render
// different bytes (tests/test_gpu_parity.py::test_strict_barrier_build_renders_the_same_bytes) instead of one
// pixel in ten thousand frames.
__device__ __forceinline__ void LdsBarrier() {
#ifdef PM_STRICT_BARRIERS
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

__device__ __forceinline__ float Sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
__device__ __forceinline__ float Sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// "not all four corners strictly on one side" test used throughout tileKernel
// (PietRender.metal:241, :289, :340, :349, :394, :431).
__device__ __forceinline__ bool Straddles(float s00, float s01, float s10, float s11) {
    return s00 * s01 + s00 * s10 + s00 * s11 < 3.0f;
}

// A stored RGBA8 pixel (R in the low byte) in the target's byte order: R and B change places for a
// BGRA8 target (the reference's drawable, PietRenderer.m:29).
__device__ __forceinline__ uint32_t StoreOrder(uint32_t rgba, uint32_t bgra) {
    return bgra ? ((rgba & 0xff00ff00u) | ((rgba & 0xffu) << 16) | ((rgba >> 16) & 0xffu)) : rgba;
}

// A framebuffer store.  Pixels are written once and never read again by the frame: stored WRITE-THROUGH (sc1) they leave the XCD's
// L2 as they are written instead of sitting there dirty until the kernel ends -- the release at the end of a kernel writes every dirty
// line back before the next dispatch (or the frame's end) is signalled, and 33 MB of pixels are most of what a frame leaves dirty.
#ifndef PM_FB_WT
#define PM_FB_WT 1
#endif
__device__ __forceinline__ void StorePixels4(uint8_t *dst, uint4 v) {
#if PM_FB_WT
    StoreWT16(reinterpret_cast<uint4 *>(dst), v);
#else
    *reinterpret_cast<uint4 *>(dst) = v;
#endif
}
__device__ __forceinline__ void StorePixel(uint8_t *dst, uint32_t v) {
#if PM_FB_WT
    StoreWT4(reinterpret_cast<uint32_t *>(dst), v);
#else
    *reinterpret_cast<uint32_t *>(dst) = v;
#endif
}

__device__ __forceinline__ uint32_t LoadU32(const uint8_t *p) { return *reinterpret_cast<const uint32_t *>(p); }
__device__ __forceinline__ float2 LoadF2(const uint8_t *p) { return *reinterpret_cast<const float2 *>(p); }

// K1b: pixels of the tiles binning resolved (background or one opaque colour) -- the composite of PietRender.metal:34-44 for tiles
// that never reach the tile kernels.  Pure store bandwidth: a launch of its own (pm_clear_kernel), extra workgroups of the tile
// kernel's launch, or -- round 6, the default -- of the BINNING launch, for the strip rows no item reaches (the others' pixels are
// written by their own binning workgroups): binning is a kernel of dependent chains that leaves the store path idle.
// (kW: waves of the calling block -- four, or one for the one-wave blocks of pm_bin_kernel<.., 1>)
template <int kW = kBinWaves>
__device__ __forceinline__ void ClearStripRow(const FrameParams &P, uint32_t striprow) {
    const uint32_t lane = LaneId(), wave = kW == 1 ? 0u : WaveId();
    const uint32_t strip = striprow % P.strips_x;
    const uint32_t row_rel = striprow / P.strips_x;
    const uint32_t t = lane >> 2;  // tile of this lane's 4 pixels
    const uint32_t tx = strip * kStripTiles + t;
    if (tx >= P.tiles_x) return;
    const uint32_t state = P.tile_state[row_rel * P.tiles_x + tx];
    if (state == 0) return;  // queued: the tile kernels write it
    const uint32_t col = StoreOrder(state, P.fb_bgra);
    const uint32_t px = strip * kGroupW + lane * 4u;
    const uint32_t y0 = (P.row0 + row_rel) * kTileH;
    // 16 pixel rows x 1024 B per strip row: thread -> (row = it * kW + wave, 16 B = 4 px at lane*4)
#pragma unroll
    for (uint32_t it = 0; it < kTileH / kW; ++it) {
        const uint32_t r = it * kW + wave;
        const uint32_t py = y0 + r;
        if (py < P.height && px < P.width) {
            uint8_t *dst = P.fb + static_cast<size_t>(row_rel * kTileH + r) * P.fb_stride + static_cast<size_t>(px) * 4;
            if (px + 4 <= P.width && P.fb_vec16) {
                StorePixels4(dst, make_uint4(col, col, col, col));
            } else {
                for (uint32_t k = 0; k < 4 && px + k < P.width; ++k) reinterpret_cast<uint32_t *>(dst)[k] = col;
            }
        }
    }
}

// Segments of an item as the kernels count them.
__device__ __forceinline__ uint32_t FillSegs(uint32_t npt) { return npt; }                      // implicitly closed (:262)
__device__ __forceinline__ uint32_t PolySegs(uint32_t npt) { return npt >= 2 ? npt - 1 : 0; }  // open (:369)
__device__ __forceinline__ uint32_t SegsOf(uint32_t tag, uint32_t npt) {  // segments of an item (a line: its one)
    return tag == kItemFill ? FillSegs(npt) : (tag == kItemPoly ? PolySegs(npt) : (tag == kItemLine ? 1u : 0u));
}

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

// The same two votes with the strip row's edges handed in as FLOATS (wave-uniform, in SGPRs -- pm_bin_rows.h makes them once per strip
// row): the int -> float conversions of VoteFill / VotePoly are vector instructions, the compiler hoists their results out of the
// vote loop into VGPRs that live through the whole kernel, and under pressure it spills them -- a scratch reload (and its
// `s_waitcnt vmcnt(0)`, which also waits for the next round's points and this round's stores) in the middle of every vote.
// fsx0 / fsx1: the STRIP's x extent (sx0, sx0 + kGroupW), fy0 / fy1: the tile row's y extent.  Same expressions, same values.
__device__ __forceinline__ bool VoteFillF(float4 s, float fy0, float fy1, float fsx0, float fsx1) {
    const float xmin = fminf(s.x, s.z), ymin = fminf(s.y, s.w);
    const float xmax = fmaxf(s.x, s.z), ymax = fmaxf(s.y, s.w);
    if (!(ymax >= fy0 && ymin < fy1 && xmin < fsx1)) return false;
    const float a = s.w - s.y;
    const float b = s.x - s.z;
    const float c = -(a * s.x + b * s.y);
    const float left = a * fsx0;
    const float right = a * fsx1;
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
    if (Straddles(s00, s01, s10, s11) && xmax > fsx0) hit = true;
    return hit;
}
// fyt0 / fyt1: y extent of the row the voting lane tests with (y_test, y_test + kTileH: quirk Q4), fsy0 / fsy1: the group's (sy0, sy0 + kGroupH)
__device__ __forceinline__ bool VotePolyF(float4 s, float hw, float fyt0, float fyt1, float fsx0, float fsx1, float fsy0, float fsy1) {
    const float xmin = fminf(s.x, s.z), ymin = fminf(s.y, s.w);
    const float xmax = fmaxf(s.x, s.z), ymax = fmaxf(s.y, s.w);
    if (!(ymax > fsy0 - hw && ymin < fsy1 + hw && xmax > fsx0 - hw && xmin < fsx1 + hw)) return false;
    const float a = s.w - s.y;
    const float b = s.x - s.z;
    const float c = -(a * s.x + b * s.y);
    const float left = a * (fsx0 - hw);
    const float right = a * (fsx1 + hw);
    const float top = b * (fyt0 - hw);
    const float bot = b * (fyt1 + hw);
    const float s00 = Sgn(top + left + c);
    const float s01 = Sgn(top + right + c);
    const float s10 = Sgn(bot + left + c);
    const float s11 = Sgn(bot + right + c);
    return Straddles(s00, s01, s10, s11);
}

// End points of segment k of a Fill item (pts = its point array, npt entries).  Plain (the
// reference, PietRender.metal:262-263): point k to point k + 1, the last one back to point 0.
// Compound (extension D11, pm_layout.h): NaN entries separate sub-paths and start no segment
// (returns false); a point followed by a separator closes to the index the separator carries,
// clamped into the array.  The second load happens for closing segments only.
__device__ __forceinline__ bool FillSegmentEnds(const uint8_t *pts, uint32_t npt, bool compound, uint32_t k, float2 &a, float2 &b) {
    const uint32_t k1 = (k + 1 == npt) ? 0u : k + 1;
    a = LoadF2(pts + static_cast<size_t>(k) * 8);
    b = LoadF2(pts + static_cast<size_t>(k1) * 8);
    if (compound) {
        if (a.x != a.x) return false;
        if (b.x != b.x) b = LoadF2(pts + static_cast<size_t>(min(__float_as_uint(b.y), npt - 1u)) * 8);
    }
    return true;
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
// Contains two barriers (LDS-only).
template <int NW>
__device__ __forceinline__ uint32_t BlockRank(bool pred, uint32_t *s_part, uint32_t *total) {
    const uint64_t m = __ballot(pred);
    if constexpr (NW == 1) {  // (a single wave: the ballot is the whole answer)
        *total = static_cast<uint32_t>(__popcll(m));
        return RankBelow(m);
    }
    const uint32_t wave = threadIdx.x >> 6;
    if (LaneId() == 0) s_part[wave] = static_cast<uint32_t>(__popcll(m));
    LdsBarrier();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t v = s_part[w];
        if (w < static_cast<int>(wave)) base += v;
        tot += v;
    }
    LdsBarrier();
    *total = tot;
    return base + RankBelow(m);
}

// Block-wide exclusive scan of arbitrary u32 values.  Two barriers.
template <int NW>
__device__ __forceinline__ uint32_t BlockExclusiveScan(uint32_t v, uint32_t *s_part, uint32_t *total) {
    const uint32_t incl = WaveInclusiveScan(v);
    if constexpr (NW == 1) {
        *total = WaveLast(incl);
        return incl - v;
    }
    const uint32_t wave = threadIdx.x >> 6;
    if (LaneId() == 63) s_part[wave] = incl;
    LdsBarrier();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t x = s_part[w];
        if (w < static_cast<int>(wave)) base += x;
        tot += x;
    }
    LdsBarrier();
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

// f(std::integral_constant<uint32_t, t>) for t = 0 .. 15: a loop over the tiles of a strip whose index is
// a compile-time constant in every iteration (lane selects of v_writelane / v_readlane are immediates).
template <typename F, uint32_t... kT>
__device__ __forceinline__ void ForStripTilesImpl(F &&f, std::integer_sequence<uint32_t, kT...>) {
    (f(std::integral_constant<uint32_t, kT>{}), ...);
}
template <typename F>
__device__ __forceinline__ void ForStripTiles(F &&f) {
    ForStripTilesImpl(f, std::make_integer_sequence<uint32_t, kStripTiles>{});
}
template <typename F>
__device__ __forceinline__ void ForClasses(F &&f) {  // ... and k = 0 .. kClasses - 1
    ForStripTilesImpl(f, std::make_integer_sequence<uint32_t, kClasses>{});
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


// Launch helper.  With (t0, t1) the dispatch itself carries the two events
// (hipExtLaunchKernelGGL): their timestamps are the dispatch's own begin / end, what a
// kernel trace shows, and no extra packet goes on the queue.
#define PM_LAUNCH(kernel, grid, block, stream, t0, t1, ...)                                        \
    do {                                                                                           \
        if (t0)                                                                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, t0, t1, 0, __VA_ARGS__);         \
        else                                                                                       \
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                       \
    } while (0)

}  // namespace

}  // namespace pm
