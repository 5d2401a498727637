// The tile stage: renderKernel (TestApp/PietRender.metal:457-566) and the composite (:16-44) for ONE queued tile, by one wave or by
// the four waves of a workgroup -- shared by pm_fine_kernel (pm_fine.hip) and the tile role of pm_frame_kernel (pm_frame.hip).
#pragma once
#include <type_traits>
#include "pm_kernels_common.h"
#include "pm_coarse_tile.h"

namespace pm {

// (K1b, ClearStripRow -- the pixels of the tiles binning resolved: pm_kernels_common.h; the binning launch calls it too)

// =====================================================================================
namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// f32 -> binary16 of a value that is the result of f32 arithmetic.  The value is pinned in a
// register first: otherwise instruction selection folds `half(a * b)` (and friends) into
// v_fma_mixlo_f16, which rounds the exact result ONCE to binary16 -- not the f32 rounding followed
// by the conversion that the source (and the reference's `half(...)` casts, decision D1) specify.
// Measured on gfx950: 958 of 16.7 M random products differ (tools/probes/mix_probe.hip); a
// 2 000-scene fuzz run found two pixels off by one because of it.
__device__ __forceinline__ _Float16 ToHalf(float x) {
    PinF32(x);
    return static_cast<_Float16>(x);
}

__device__ __forceinline__ _Float16 HalfFromBits(uint32_t b) {
    const uint16_t u = static_cast<uint16_t>(b);
    return __builtin_bit_cast(_Float16, u);
}

// alpha of DrawFill from signedArea + backdrop (binary16): the non-zero rule (:538), or -- with
// the rule bit the command carries in its last word -- the even-odd formula the reference leaves
// in a comment (:539): abs(alpha - 2 * round(0.5 * alpha)), every step in binary16.  (round =
// nearest-even here and in the oracle: a tie means alpha is an odd integer, and either
// neighbour then gives |+-1| = 1.)
__device__ __forceinline__ _Float16 FillAlpha(_Float16 a, bool even_odd) {
    if (even_odd) {
        const _Float16 t = static_cast<_Float16>(0.5f) * a;
        const _Float16 r = __builtin_rintf16(t);
        const _Float16 v = a - static_cast<_Float16>(2.0f) * r;
        return __builtin_fabsf16(v);
    }
    float f = fminf(fabsf(static_cast<float>(a)), 1.0f);
    PinF32(f);
    return static_cast<_Float16>(f);
}

// ... the same for two pixels at once: min(abs(alpha), 1.0h) (:538) is exact in binary16 itself -- |x| flips a
// bit, and minNum of two binary16 values is one of them -- so the non-zero rule needs no trip through binary32.
__device__ __forceinline__ half2_t FillAlpha2(half2_t a, bool even_odd) {
    if (even_odd) {
        half2_t r;
        r.x = FillAlpha(a.x, true);
        r.y = FillAlpha(a.y, true);
        return r;
    }
    half2_t one;
    one.x = one.y = static_cast<_Float16>(1.0f);
    return __builtin_elementwise_min(__builtin_elementwise_abs(a), one);
}

// Coverage of Cmd_Circle for one pixel (d = pixel - centre; rx, ry = centre - bbox corner): the
// circle of PietRender.metal:486-490, or -- extension D10, CmdCircle.flags bit 0 -- the ellipse
// inscribed in the bbox with the first-order distance F / |grad F| of F = x^2/rx^2 + y^2/ry^2 - 1
// (the WebRender ellipse.glsl form the reference's TODO points to), every step binary32 in the
// order oracle/pmo_render.c ellipse_alpha() fixes.
__device__ __forceinline__ float CircleAlpha(float dx, float dy, float rx, float ry, bool ellipse) {
    if (ellipse) {
        if (!(rx > 0.0f) || !(ry > 0.0f)) return 0.0f;
        const float ux = dx / (rx * rx), uy = dy / (ry * ry);
        const float g = (dx * ux + dy * uy) - 1.0f;
        const float len = 2.0f * sqrtf(ux * ux + uy * uy);
        return Sat(-(g / len));
    }
    return Sat(fminf(rx, ry) - sqrtf(dx * dx + dy * dy));
}

// The distance field is kept SQUARED between its Line commands and the Stroke that consumes it:
// df = min(1e9, sqrt(d_1), sqrt(d_2), ...) (renderKernel :471, :495-499) equals
// min(1e9, sqrt(min(d_1, d_2, ...))) bit for bit, because a correctly rounded sqrt is monotone and
// min picks one of its arguments -- one sqrt per Stroke and pixel instead of one per Line and pixel
// (47 ns each for a lone wave, profiles/r02_issue_probe.txt).  "No line yet" is +infinity,
// materialized where it is used: as a plain literal the compiler hoists four copies of it out of
// the tile loop and then spills them.
__device__ __forceinline__ float FarAway() { return OpaqueInfinity(); }
__device__ __forceinline__ float StrokeDistance(float d2) { return fminf(1e9f, sqrtf(d2)); }

__device__ __forceinline__ half2_t Splat(_Float16 v) { half2_t r; r.x = v; r.y = v; return r; }

// ---- row-sparse Fill evaluation -----------------------------------------------------------
// A Fill command only changes the pixels of the rows its segment crosses (wx != wy,
// PietRender.metal:513-514): at Tiger 4K that is 3.4 of a tile's 16 rows on average, yet the
// straightforward interpreter above runs the whole area integral -- six IEEE divisions per
// lane -- for all 16.  Here the (command, row) pairs that are live become FRAGMENTS:
//   pass 1  lane = (Fill command, row), 4 commands x 16 rows per step: the y-only part (window,
//           both divides of :515-516), live pairs compacted with a ballot into fragment slots;
//   pass 2  lane = (fragment, 4 adjacent pixels), 16 fragments per step: the x part (:517-527)
//           exactly as written, the 16 half contributions of a fragment go to LDS;
//   pass 3  the command loop in list order; a Fill is one LDS read and one packed half add for
//           the rows named in the 16-bit row mask pass 1 left in the staged command.
// Every arithmetic expression is the one of Interpret(); only WHICH (command, row) pairs get
// evaluated changes, and those are exactly the pairs the reference adds a contribution for.
// Tiles with long lists are rendered by the 4 waves of a workgroup together: passes 1 and 2
// are split by command batch, pass 3 by pixel rows (1 pixel per lane).
constexpr uint32_t kSpChunk = 64;     // commands staged per chunk
constexpr uint32_t kMaxFrag = 64;     // fragment slots per wave (one step of pass 1 adds <= 64)
constexpr uint32_t kAlphaSlots = 16;  // workgroup mode: items evaluated ahead per round

// Per wave: the staged chunk of commands, and the fragment region of its Fills.  The fused kernel builds the
// tile's list first (CoarseTile): its scratch shares the fragment region's bytes, and the first chunk of the
// list is built straight into `cmds` -- what the renderer interprets never leaves the CU.
struct WaveFineLds {
    float4 fparam[kMaxFrag];        // {tx, ty, wx - wy, bits(command index | first hot pixel << 8 | hot pixels before this fragment << 16)}
    uint2 contrib[kMaxFrag][4];     // 16 binary16 contributions per fragment (x = 0..15)
    uint8_t fill_ix[kSpChunk];      // indices of the chunk's Fill commands, in order
    uint8_t hot_own[64];            // pass 2: position in a step of hot pixels -> fragment that starts there
};
struct WaveLds {
    Cmd cmds[kSpChunk];
    union {
        WaveFineLds f;
        CoarseLds c;
    };
};
static_assert(sizeof(CoarseLds) <= sizeof(WaveFineLds), "list building fits in the fragment region: five workgroups per CU");

struct SparseLds {
    WaveLds w[kWaves];
    // workgroup mode (tiles with long lists):
    uint2 alpha[kAlphaSlots][64];         // per item: 256 binary16 alphas, pixel-linear (row * 16 + x)
    uint2 rec[kSpChunk];                  // per item of the chunk: its colour {r | g << 16, b | a << 16} (binary16)
    uint2 carry_sa[2][64];                // signedArea / distance state of an item cut by the chunk boundary
    float4 carry_df[2][64];               //   (two copies, alternating per chunk)
    uint32_t wg_ncmd[2];                  // fused kernel: list length found by wave 0 (alternating per pass)
    uint32_t prep_over[2];                // per chunk (alternating): a wave's share of the chunk's Fill commands did not fit its fragment region
    CoarseShared coarse_shared;           // fused kernel: what the four waves exchange while they build a long list together
    uint32_t next_item;                   // next item of the round nobody has taken yet
    uint16_t item_se[kSpChunk + 1];       // per item: first command | blend command << 8 (last entry: the open tail)
};
// (five workgroups per CU need <= 31 184 B each -- measured, pm_bin.hip; this one is 30 608 B)
static_assert(sizeof(SparseLds) <= 40960, "four workgroups per CU");
// The working set of a kernel that renders EVERY tile with one wave (a dense frame: pm_fine_kernel<.., kDense>): no alpha images,
// no hand-over words -- 19 KB, so that six workgroups share a CU (and the code without the workgroup paths fits 80 VGPRs).
// (and room for a second chunk of every wave's list: a dense frame's lists are long -- config 4: 72 commands on average -- and what
//  does not fit in LDS goes to the tile's list in HBM and comes straight back, a store, a fence and a round trip per tile)
constexpr uint32_t kDenseLdsChunks = 2;
struct DenseLds {
    WaveLds w[kWaves];
    Cmd more[kWaves][(kDenseLdsChunks - 1u) * kSpChunk];  // commands 64 .. 127 of the wave's list
};
static_assert(sizeof(DenseLds) <= 26624, "six workgroups per CU");

__device__ __forceinline__ half2_t Half2FromBits(uint32_t b) { return __builtin_bit_cast(half2_t, b); }

// x part of Fill for one pixel (:517-527), then `half(area * (wx - wy))`
__device__ __forceinline__ _Float16 FillContribution(float fsx, float fex, float px, float tx, float ty, float wd) {
    const float sx = fsx - px, ex = fex - px;
    const float xsx = sx + (ex - sx) * tx;
    const float xsy = sx + (ex - sx) * ty;
    const float xmin = fminf(fminf(xsx, xsy), 1.0f) - 1e-6f;
    const float xmax = fmaxf(xsx, xsy);
    const float b = fminf(xmax, 1.0f);
    const float c = fmaxf(b, 0.0f);
    const float d = fmaxf(xmin, 0.0f);
    const float area = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
    return ToHalf(area * wd);
}

// One step of pass 1: the Fill commands [pos, pos + 4) of the chunk's Fill list (those below
// `limit`) x 16 rows.  Live pairs get the next fragment slots of this wave's region; returns
// false (and writes nothing) if the region cannot take them.
// slot_base: what the staged commands call this region's first fragment slot (workgroup tiles number the four waves' regions through:
// a command's fragments may have been made by another wave than the one that adds them up, see PrepareFillsShared).
__device__ __forceinline__ bool FillStep(WaveFineLds &W, Cmd *cmds, const uint8_t *fill_ix, uint32_t limit, uint32_t pos,
                                         uint32_t &nfrag, uint32_t y0, uint32_t slot_base = 0) {
    const uint32_t lane = LaneId();
    const uint32_t q = lane >> 4, row = lane & 15u;
    const uint32_t fi = pos + q;
    const bool valid = fi < limit;
    const uint32_t ci = fill_ix[valid ? fi : pos];
    const float py = static_cast<float>(y0 + row);
    const float sy = __uint_as_float(cmds[ci].body[2]) - py;
    const float ey = __uint_as_float(cmds[ci].body[4]) - py;
    const float wx = Sat(sy), wy = Sat(ey);
    const bool live = valid && wx != wy;
    const uint64_t mask = __ballot(live);
    if (mask == 0) return true;  // (the staged body[0] of a Fill is 0: no row, nothing to add)
    if (nfrag + static_cast<uint32_t>(__popcll(mask)) > kMaxFrag) return false;
    if (live) {
        const float tx = (wx - sy) / (ey - sy);
        const float ty = (wy - sy) / (ey - sy);
        W.fparam[nfrag + RankBelow(mask)] = make_float4(tx, ty, wx - wy, __uint_as_float(ci));
    }
    if (row == 0 && valid) {
        const uint32_t gm = static_cast<uint32_t>(mask >> (16u * q)) & 0xffffu;
        const uint32_t gb = nfrag + static_cast<uint32_t>(__popcll(mask & ((1ull << (16u * q)) - 1ull)));
        cmds[ci].body[0] = gm | ((slot_base + gb) << 16);
    }
    nfrag += static_cast<uint32_t>(__popcll(mask));
    return true;
}

// Pass 1, dense: the Fill commands [from, limit) of the chunk's Fill list (at most 64 of them: a chunk has 64 commands), a lane per
// (command, row) PAIR of the rows the command's segment can touch -- floor(min y) .. floor(max y), clipped to the tile: a superset of the
// rows FillStep finds live (live means Sat(sy) != Sat(ey): some of the segment lies strictly between py and py + 1) --, 64 pairs per step.
// FillStep spends a lane on each of the 16 rows of four commands, and 3.4 of them are live at the 4K Tiger (2.9 at config 4): a chunk's 25
// Fills are two steps here and seven there.  The same fragments in the same slots, the same words in the staged commands (a command
// without a live row keeps the 0 it was staged with).  Returns the ordinal of the first Fill NOT covered (the fragment region is full).
constexpr uint32_t kFillPairsMin = 9;  // fewer commands than this: FillStep (two steps cost what the set-up and one step cost here)
__device__ __forceinline__ uint32_t FillPairs(WaveFineLds &W, Cmd *cmds, const uint8_t *fill_ix, uint32_t limit, uint32_t from, uint32_t &nfrag,
                                              uint32_t y0, uint32_t slot_base) {
    const uint32_t lane = LaneId();
    const bool valid = from + lane < limit;
    const uint32_t ci = fill_ix[valid ? from + lane : from];
    const uint32_t wa = cmds[ci].body[2], wb = cmds[ci].body[4];
    uint32_t lo = 0, cnt = 0;
    if (valid) {
        const float ya = __uint_as_float(wa), yb = __uint_as_float(wb), fy0 = static_cast<float>(y0);
        // (floors of the absolute coordinates: integers, and so is y0 -- the differences are exact)
        const float flo = fminf(fmaxf(floorf(fminf(ya, yb)) - fy0, 0.0f), 16.0f), fhi = fminf(fmaxf(floorf(fmaxf(ya, yb)) - fy0 + 1.0f, 0.0f), 16.0f);
        const bool num = ya == ya && yb == yb;  // (a NaN takes part in FillStep's test as 0: every row is a candidate)
        lo = num ? static_cast<uint32_t>(flo) : 0u;
        cnt = num ? static_cast<uint32_t>(fmaxf(fhi - flo, 0.0f)) : 16u;
    }
    const uint32_t incl = WaveInclusiveScan(cnt);
    const uint32_t off = incl - cnt, total = WaveLast(incl);
    uint32_t covered = min(64u, limit - from);
    uint32_t own_carry = 0;
#pragma unroll 1
    for (uint32_t e0 = 0; e0 < total; e0 += 64u) {
        // the command every pair of the step belongs to: each command marks where its pairs begin, a prefix maximum spreads the marks
        W.hot_own[lane] = 0;
        WaveSync();
        if (cnt != 0u && off - e0 < 64u) W.hot_own[off - e0] = static_cast<uint8_t>(lane);
        WaveSync();
        const uint32_t owner = max(WaveInclusiveMax(W.hot_own[lane]), own_carry);
        own_carry = WaveLast(owner);
        const uint32_t e = e0 + lane;
        const bool pv = e < total;
        const uint32_t o_lo = static_cast<uint32_t>(__shfl(static_cast<int>(lo), static_cast<int>(owner)));
        const uint32_t o_off = static_cast<uint32_t>(__shfl(static_cast<int>(off), static_cast<int>(owner)));
        const uint32_t o_ci = static_cast<uint32_t>(__shfl(static_cast<int>(ci), static_cast<int>(owner)));
        const uint32_t o_wa = static_cast<uint32_t>(__shfl(static_cast<int>(wa), static_cast<int>(owner)));
        const uint32_t o_wb = static_cast<uint32_t>(__shfl(static_cast<int>(wb), static_cast<int>(owner)));
        const uint32_t row = (o_lo + (e - o_off)) & 15u;
        // ... and FillStep's body for the pair
        const float py = static_cast<float>(y0 + row);
        const float sy = __uint_as_float(o_wa) - py;
        const float ey = __uint_as_float(o_wb) - py;
        const float wx = Sat(sy), wy = Sat(ey);
        const bool live = pv && wx != wy;
        const uint64_t mask = __ballot(live);
        if (mask == 0) continue;
        const uint32_t nlive = static_cast<uint32_t>(__popcll(mask));
        if (nfrag + nlive > kMaxFrag) {
            // The region is full: this step's commands are not covered; one whose pairs began in an earlier step is taken back (the
            // slots it has are lost until the next call).  Never the call's first command: 64 pairs are at least four commands.
            const uint32_t first_owner = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(owner)));
            const uint32_t first_off = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(o_off)));
            if (lane == 0 && first_off < e0) cmds[o_ci].body[0] = 0;
            covered = first_owner;
            break;
        }
        const uint32_t rank = RankBelow(mask);
        const uint32_t before = cmds[o_ci].body[0];  // (read in front of this step's atomics: a wave's LDS operations complete in order)
        // live pairs in front of the command's first pair of this step = the rank the lane its pairs begin at computed (0: they began earlier)
        const int a = static_cast<int>(o_off) - static_cast<int>(e0);
        const uint32_t rank_a = static_cast<uint32_t>(__shfl(static_cast<int>(rank), a > 0 ? a : 0));
        if (live) {
            const float tx = (wx - sy) / (ey - sy);
            const float ty = (wy - sy) / (ey - sy);
            W.fparam[nfrag + rank] = make_float4(tx, ty, wx - wy, __uint_as_float(o_ci));
            // row mask | first fragment << 16: the step in which a command has its first live pair brings the slot (every live pair of
            // the command in that step ORs the same number in)
            uint32_t bits = 1u << row;
            if ((before & 0xffffu) == 0u) bits |= (slot_base + nfrag + (a > 0 ? rank_a : 0u)) << 16;
            atomicOr(&cmds[o_ci].body[0], bits);
        }
        nfrag += nlive;
    }
    return from + covered;
}

// Pass 2 over this wave's fragments [0, nfrag) (nfrag <= kMaxFrag = one lane each).
//
// Of a fragment's 16 pixels only those the segment's piece of the pixel row passes over need the area integral:
//   * a pixel wholly to the RIGHT of it has xmax <= 0, so b = xmax, c = d = 0 and area = (xmax - xmin) / (xmax - xmin),
//     exactly 1.0f (the very same subtraction twice; xmax - xmin >= 1e-6 as long as |x| < 32, where 1e-6 is not
//     absorbed) -- its contribution is half(wx - wy);
//   * a pixel wholly to the LEFT has min(xs) >= 1, so xmin = fl(1 - 1e-6) =: X, b = c = 1, d = X, and the numerator
//     1 + 0.5 (X X - 1) - X is exactly 0 in binary32 (X X rounds to 1 - 34 ulp, half of that is X - 1): area = +0,
//     the contribution +-0, which changes no binary16 sum (the sign of a zero signedArea is never looked at).
// Which pixels are wholly left / right is decided from the piece's x extent in tile coordinates with 1/8 pixel of
// slack on either side -- the roundings of `xs` (a few ulp of a coordinate < 65 536: < 0.03) cannot carry a
// pixel across -- and everything in between, or anything not finite, or a piece that begins 30 pixels left of the
// tile's last column, goes through FillContribution() as written.  (Checked against the full evaluation of all 16
// pixels: tests/test_oracle_cpu.py::test_fill_pixel_classes and every GPU parity test.)
// Step A, lane = fragment: the hot range, the constants of the other pixels into `contrib`.  Step B, lane = hot
// pixel (64 per step, owners by scatter + prefix maximum): the x part :517-527 exactly as written.
// Tiger 4K: 2.5 of a fragment's 16 pixels are hot; config 4: 2.2.
__device__ __forceinline__ void FillPass2(WaveFineLds &W, const Cmd *cmds, uint32_t nfrag, uint32_t x0) {
    const uint32_t lane = LaneId();
    // ---- A ----
    uint32_t n_hot = 0, hot0 = 0;
    if (lane < nfrag) {
        const float4 p = W.fparam[lane];
        const uint32_t ci = __float_as_uint(p.w);
        const float fsx = __uint_as_float(cmds[ci].body[1]), fex = __uint_as_float(cmds[ci].body[3]);
        const float xa = fsx + (fex - fsx) * p.x, xb = fsx + (fex - fsx) * p.y;  // the piece's ends (tile coordinates, approximate)
        const float lo = fminf(xa, xb), hi = fmaxf(xa, xb);
        const float fx0 = static_cast<float>(x0);
        uint32_t n_left = 0, first_right = 16;
        if (lo >= -1e30f && hi <= 1e30f) {  // (false for NaN)
            n_left = static_cast<uint32_t>(fminf(fmaxf(floorf((lo - fx0) - 1.125f) + 1.0f, 0.0f), 16.0f));
            if ((fx0 + 15.0f) - lo < 30.0f) first_right = static_cast<uint32_t>(fminf(fmaxf(ceilf((hi - fx0) + 0.125f), 0.0f), 16.0f));
        }
        first_right = max(first_right, n_left);
        hot0 = n_left;
        n_hot = first_right - n_left;
        // pixels [first_right, 16): half(1.0f * (wx - wy)); the others 0 (the hot ones are overwritten in step B)
        const uint32_t cw = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, ToHalf(p.z)));
        const uint32_t cw2 = cw | (cw << 16);
        uint32_t d[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) d[k] = first_right <= 2u * k ? cw2 : (first_right == 2u * k + 1u ? cw << 16 : 0u);
        W.contrib[lane][0] = make_uint2(d[0], d[1]);
        W.contrib[lane][1] = make_uint2(d[2], d[3]);
        W.contrib[lane][2] = make_uint2(d[4], d[5]);
        W.contrib[lane][3] = make_uint2(d[6], d[7]);
    }
    const uint32_t incl = WaveInclusiveScan(n_hot);
    const uint32_t total = WaveLast(incl);
    const uint32_t excl = incl - n_hot;
    if (lane < nfrag) reinterpret_cast<uint32_t *>(&W.fparam[lane])[3] = (__float_as_uint(W.fparam[lane].w) & 0xffu) | (hot0 << 8) | (excl << 16);
    // ---- B ----
    uint32_t own_carry = 0;
#pragma unroll 1
    for (uint32_t e0 = 0; e0 < total; e0 += 64u) {
        W.hot_own[lane] = 0;
        WaveSync();
        if (n_hot != 0u && excl - e0 < 64u) W.hot_own[excl - e0] = static_cast<uint8_t>(lane);
        WaveSync();
        const uint32_t f = max(WaveInclusiveMax(W.hot_own[lane]), own_carry);
        own_carry = WaveLast(f);
        const uint32_t e = e0 + lane;
        if (e < total) {
            const float4 p = W.fparam[f];
            const uint32_t w = __float_as_uint(p.w);
            const uint32_t ci = w & 0xffu;
            const uint32_t j = ((w >> 8) & 0xffu) + (e - (w >> 16));
            const float fsx = __uint_as_float(cmds[ci].body[1]), fex = __uint_as_float(cmds[ci].body[3]);
            const _Float16 h = FillContribution(fsx, fex, static_cast<float>(x0 + j), p.x, p.y, p.z);
            reinterpret_cast<_Float16 *>(&W.contrib[f][0])[j] = h;
        }
        WaveSync();
    }
}

// Single-wave mode: passes 1 and 2 for the Fill commands [from, ...) of the staged chunk, as many
// as the wave's fragment region takes.  Returns the ordinal of the first Fill NOT covered.
template <typename Lds>
__device__ __forceinline__ uint32_t PrepareFills(Lds &S, Cmd *cmds, const uint8_t *fill_ix, uint32_t nfill, uint32_t from,
                                                 uint32_t x0, uint32_t y0, uint32_t slot_base = 0) {
    WaveFineLds &W = S.w[WaveId()].f;
    uint32_t nfrag = 0;
    uint32_t pos = from;
    if (nfill - from >= kFillPairsMin) {  // uniform
        pos = FillPairs(W, cmds, fill_ix, nfill, from, nfrag, y0, slot_base);
    } else {
#pragma unroll 1
        while (pos < nfill) {  // (the first step always fits: it adds at most 64)
            if (!FillStep(W, cmds, fill_ix, nfill, pos, nfrag, y0, slot_base)) break;
            pos += 4u;
        }
    }
    WaveSync();
    FillPass2(W, cmds, nfrag, x0);
    WaveSync();
    return min(pos, nfill);
}

// A run of `run` consecutive Fill commands from command i on, all of them prepared (passes 1 and 2 done): pass 3 for the run, in
// list order (binary16 addition is not associative).  b0 = word 0 of the chunk's commands, lane k holding command k's AS PASS 1 LEFT IT
// (row mask | first fragment << 16: read back from the staged commands after the fragments were made) -- a command's word reaches the
// scalar unit with v_readlane, so a Fill is ONE dependent LDS access (its row's contribution), not two.
// One command per step: a step of four -- row masks fetched together, then the contributions -- cost the same for one command as for
// four, and config 4's runs are two or three Fills between a FillEdge and the DrawFill (its tile kernel 0.379 -> 0.349 ms, config 5 -3 %).
// No branch on the row's bit: every lane reads a slot -- its own row's if it has one, some slot of the region if not -- and the sums are
// SELECTED (a divergent branch anywhere in the command loop makes the compiler route the whole dispatch through flow blocks that copy the
// pixel state, see InterpretSparse).
// kShared (workgroup tiles): W is wave 0's region and a slot number names the region too -- slot >> 6, sizeof(WaveLds) bytes apart.
template <bool kShared = false>
__device__ __forceinline__ void AddFillRun(const WaveFineLds &W, const uint32_t b0, uint32_t i, uint32_t run, uint32_t row, uint32_t g,
                                           half2_t &sa01, half2_t &sa23) {
    const uint32_t below = (1u << row) - 1u;
#pragma unroll 1
    for (uint32_t r = 0; r < run; ++r) {
        const uint32_t hdr = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(b0), static_cast<int>(i + r)));
        const uint32_t slot = (hdr >> 16) + static_cast<uint32_t>(__popc(hdr & below));
        uint2 v;
        if constexpr (kShared) {
            const uint8_t *region = reinterpret_cast<const uint8_t *>(&W) + ((slot >> 6) & static_cast<uint32_t>(kWaves - 1)) * static_cast<uint32_t>(sizeof(WaveLds));
            v = reinterpret_cast<const WaveFineLds *>(region)->contrib[slot & (kMaxFrag - 1u)][g];
        } else {
            v = W.contrib[slot & (kMaxFrag - 1u)][g];
        }
        const bool on = ((hdr >> row) & 1u) != 0u;
        const half2_t n01 = sa01 + Half2FromBits(v.x), n23 = sa23 + Half2FromBits(v.y);
        sa01 = on ? n01 : sa01;
        sa23 = on ? n23 : sa23;
    }
}

// Commands of one kind in a row from command i on (bit i of fm, the chunk's mask of that kind, is set)
__device__ __forceinline__ uint32_t FillRunLength(uint64_t fm, uint32_t i) {
    const uint64_t rest = ~(fm >> i);
    return rest ? static_cast<uint32_t>(__builtin_ctzll(rest)) : 64u - i;
}

// Pixels of one lane, whole-tile layout, signedArea packed (the half adds are per element)
struct PixelStateS {
    half2_t r01, r23, g01, g23, b01, b23;
    float df[4];
    half2_t sa01, sa23;
};

__device__ __forceinline__ void Blend4S(PixelStateS &st, uint32_t rg, uint32_t ba, half2_t al01, half2_t al23) {
    const half2_t fga = Splat(HalfFromBits(ba >> 16));
    const half2_t a01 = fga * al01, a23 = fga * al23;
    const half2_t fr = Splat(HalfFromBits(rg)), fg = Splat(HalfFromBits(rg >> 16)), fb = Splat(HalfFromBits(ba));
    st.r01 = st.r01 + (fr - st.r01) * a01; st.r23 = st.r23 + (fr - st.r23) * a23;
    st.g01 = st.g01 + (fg - st.g01) * a01; st.g23 = st.g23 + (fg - st.g23) * a23;
    st.b01 = st.b01 + (fb - st.b01) * a01; st.b23 = st.b23 + (fb - st.b23) * a23;
}

// renderKernel's command loop (:474-560), whole tile per wave (lane -> row lane/4, 4 pixels)
template <typename Lds>
__device__ __forceinline__ void InterpretSparse(Lds &S, Cmd *cmds, const uint8_t *fill_ix, uint32_t n, uint32_t x0, uint32_t y0,
                                                PixelStateS &st) {
    const uint32_t lane = LaneId();
    const uint32_t ol = Opaque(lane);  // (row and column made from a lane number the compiler cannot see through: hoisted out of the tile loop they are spilled)
    const uint32_t row = ol >> 2, g = ol & 3u;
    const float px0 = static_cast<float>(x0 + 4u * g), py = static_cast<float>(y0 + row);
    // Lane i keeps command i of the chunk in registers; the loop below picks the command's words out with v_readlane
    // into SCALAR registers.  (Read from LDS command by command, every command began with a round trip to the LDS --
    // tag, then the words its case needs -- on the wave's critical path: 0.1 us each with twenty waves on the CU.)
    Cmd mine;
    mine.tag = 0;
    mine.body[0] = mine.body[1] = mine.body[2] = mine.body[3] = mine.body[4] = 0;
    if (lane < n) mine = cmds[lane];
    // the chunk's Fill commands, in order
    const bool isf = mine.tag == kCmdFill;
    const uint64_t fm = __ballot(isf);
    if (isf) const_cast<uint8_t *>(fill_ix)[RankBelow(fm)] = static_cast<uint8_t>(lane);
    const uint32_t nfill = static_cast<uint32_t>(__popcll(fm));
    WaveSync();
    // the chunk's Solid commands: runs of them (a tile inside several translucent shapes) are blended without the dispatch
    const uint64_t sm = __ballot(mine.tag == kCmdSolid);
    // The command loop proper has NO divergent branch in it: with one anywhere inside, the compiler structurizes the whole
    // dispatch -- every node of the switch becomes a flow block that copies the pixel state (a dozen v_mov per command
    // and a vmcnt(0) wait in the dense kernel's listing).  So the fragments of the chunk's Fills (divergent code) are
    // made OUTSIDE it, for as many Fills as the fragment region takes, and the loop runs up to the first Fill they do not cover.
    uint32_t fo = 0, i = 0;
    while (i < n) {
    uint32_t stop = n;
    if (fo < nfill) {
        const uint32_t prepared = PrepareFills(S, cmds, fill_ix, nfill, fo, x0, y0);  // > fo, uniform
        if (prepared < nfill) stop = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(fill_ix[prepared])));
        mine.body[0] = cmds[lane].body[0];  // (pass 1 left the Fills' row masks and fragment slots there; lanes >= n: never looked at)
    }
    for (; i < stop; ++i) {
        auto word = [&](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(i))); };
        switch (word(mine.tag)) {
            case kCmdCircle: {
                const uint32_t b1 = word(mine.body[1]), b2 = word(mine.body[2]);
                const float bx0 = static_cast<float>(b1 & 0xffffu), by0 = static_cast<float>(b1 >> 16);
                const float bx1 = static_cast<float>(b2 & 0xffffu), by1 = static_cast<float>(b2 >> 16);
                const float cx = bx0 + (bx1 - bx0) * 0.5f, cy = by0 + (by1 - by0) * 0.5f;
                const bool ellipse = (word(mine.body[0]) & kCmdCircleEllipse) != 0;
                const float dy = py - cy;
                _Float16 alpha[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dx = (px0 + static_cast<float>(k)) - cx;
                    alpha[k] = ToHalf(CircleAlpha(dx, dy, cx - bx0, cy - by0, ellipse));
                }
                const half2_t zero = Splat(static_cast<_Float16>(0.0f));
                half2_t a01, a23;
                a01.x = alpha[0]; a01.y = alpha[1]; a23.x = alpha[2]; a23.y = alpha[3];
                st.r01 = st.r01 + (zero - st.r01) * a01; st.r23 = st.r23 + (zero - st.r23) * a23;
                st.g01 = st.g01 + (zero - st.g01) * a01; st.g23 = st.g23 + (zero - st.g23) * a23;
                st.b01 = st.b01 + (zero - st.b01) * a01; st.b23 = st.b23 + (zero - st.b23) * a23;
                break;
            }
            case kCmdLine: {
                const float sx = __uint_as_float(word(mine.body[1])), sy = __uint_as_float(word(mine.body[2]));
                const float ex = __uint_as_float(word(mine.body[3])), ey = __uint_as_float(word(mine.body[4]));
                const float lx = ex - sx, ly = ey - sy;
                const float den = lx * lx + ly * ly;
                const float dy = py - sy;
                const float lydy = ly * dy;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dx = (px0 + static_cast<float>(k)) - sx;
                    const float t = Sat((lx * dx + lydy) / den);
                    const float fx = lx * t - dx, fy = ly * t - dy;
                    st.df[k] = fminf(st.df[k], fx * fx + fy * fy);  // (squared: see FarAway)
                }
                break;
            }
            case kCmdStroke: {
                const float half_width = __uint_as_float(word(mine.body[0]));
                _Float16 alpha[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    alpha[k] = ToHalf(Sat(half_width + 0.5f - StrokeDistance(st.df[k])));
                    st.df[k] = FarAway();
                }
                half2_t a01, a23;
                a01.x = alpha[0]; a01.y = alpha[1]; a23.x = alpha[2]; a23.y = alpha[3];
                Blend4S(st, word(mine.body[2]), word(mine.body[3]), a01, a23);
                break;
            }
            case kCmdFill: {
                const uint32_t run = min(FillRunLength(fm, i), stop - i);  // >= 1, all of them prepared
                AddFillRun(S.w[WaveId()].f, mine.body[0], i, run, row, g, st.sa01, st.sa23);
                fo += run;
                i += run - 1u;
                break;
            }
            case kCmdFillEdge: {
                const float sgn = static_cast<float>(static_cast<int>(word(mine.body[0])));
                const float v = sgn * Sat(py - __uint_as_float(word(mine.body[1])) + 1.0f);
                st.sa01.x = ToHalf(static_cast<float>(st.sa01.x) + v);
                st.sa01.y = ToHalf(static_cast<float>(st.sa01.y) + v);
                st.sa23.x = ToHalf(static_cast<float>(st.sa23.x) + v);
                st.sa23.y = ToHalf(static_cast<float>(st.sa23.y) + v);
                break;
            }
            case kCmdDrawFill: {
                const _Float16 bd = static_cast<_Float16>(static_cast<float>(static_cast<int>(word(mine.body[0]))));
                const half2_t s01 = st.sa01 + Splat(bd), s23 = st.sa23 + Splat(bd);
                const bool eo = (word(mine.body[4]) & kFillEvenOdd) != 0;
                const half2_t a01 = FillAlpha2(s01, eo), a23 = FillAlpha2(s23, eo);
                st.sa01 = st.sa23 = Splat(static_cast<_Float16>(0.0f));
                Blend4S(st, word(mine.body[2]), word(mine.body[3]), a01, a23);
                break;
            }
            case kCmdSolid: {
                // rgb = mix(rgb, fg.rgb, fg.a) (:546-549) for the whole run of Solids from here on
                const uint32_t run = FillRunLength(sm, i);  // >= 1
#pragma unroll 1
                for (uint32_t r = 0; r < run; ++r) {
                    const uint32_t rg = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine.body[1]), static_cast<int>(i + r)));
                    const uint32_t ba = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine.body[2]), static_cast<int>(i + r)));
                    const half2_t a = Splat(HalfFromBits(ba >> 16));  // (alpha 1: fg.a * 1 is fg.a)
                    const half2_t fr = Splat(HalfFromBits(rg)), fg = Splat(HalfFromBits(rg >> 16)), fb = Splat(HalfFromBits(ba));
                    st.r01 = st.r01 + (fr - st.r01) * a; st.r23 = st.r23 + (fr - st.r23) * a;
                    st.g01 = st.g01 + (fg - st.g01) * a; st.g23 = st.g23 + (fg - st.g23) * a;
                    st.b01 = st.b01 + (fb - st.b01) * a; st.b23 = st.b23 + (fb - st.b23) * a;
                }
                i += run - 1u;
                break;
            }
            default:
                break;
        }
    }
    }
}

// Tiles with long lists: the four waves of a workgroup render one tile together.
//
// A wave alone walks a list at 0.2-0.3 us per command whatever the number of pixels it owns:
// every command is a dependent chain (LDS fetch + readfirstlane + scalar dispatch 83 ns, an IEEE
// divide 29 ns, a square root 47 ns -- tools/probes/issue_probe.hip), so the span of the whole
// kernel used to be its longest list.  But signedArea and the distance field are reset by the
// command that consumes them (DrawFill :542, Stroke :506): the commands between two blending
// commands form an ITEM whose alpha does not depend on any other item.  So, per chunk:
//   phase A  (parallel over ITEMS, one item per wave at a time, 4 pixels per lane): the item's
//            Fill / FillEdge / Line commands in list order exactly as InterpretSparse() runs
//            them, then alpha of the closing DrawFill / Stroke / Circle / Solid for all 256
//            pixels -> a binary16 image in LDS;
//   phase B  (parallel over PIXELS, 1 pixel per lane): the blends (:505, :543, :549, :491) in
//            list order, one LDS read and three mixes per item, no dispatch at all.
// An item cut by the chunk boundary hands its accumulators on through LDS.  (The lists are what
// pm_coarse_kernel writes: Fill / FillEdge only before their DrawFill, Line only before its Stroke.)
struct PixelRGB {
    _Float16 r, g, b;
};

// Commands [s, e) of one item, whole tile per wave (lane -> row lane / 4, 4 pixels): Fill,
// FillEdge and Line exactly as in InterpretSparse(); fm = the chunk's Fill commands.
// mine = the chunk's commands, lane i holding command i (their words reach the loop through v_readlane).
// shared: the fragments of all the chunk's Fill commands are there already (PrepareFillsShared); otherwise this wave makes those of
// its item's, in its own region.
__device__ __forceinline__ void RunItemCommands(SparseLds &S, Cmd *cmds, const Cmd &mine, const uint8_t *fill_ix, uint64_t fm, uint32_t s, uint32_t e,
                                                uint32_t x0, uint32_t y0, half2_t &sa01, half2_t &sa23, float (&df)[4], const bool shared) {
    if (s >= e) return;
    const uint32_t lane = LaneId();
    const uint32_t ol = Opaque(lane);  // (row and column made from a lane number the compiler cannot see through: hoisted out of the tile loop they are spilled)
    const uint32_t row = ol >> 2, g = ol & 3u;
    const float px0 = static_cast<float>(x0 + 4u * g), py = static_cast<float>(y0 + row);
    uint32_t fo = static_cast<uint32_t>(__popcll(fm & ((1ull << s) - 1ull)));
    const uint32_t flimit = static_cast<uint32_t>(__popcll(fm & (e >= 64u ? ~0ull : ((1ull << e) - 1ull))));
    // (as in InterpretSparse: the fragments are made outside the command loop, which has no divergent branch in it)
    uint32_t b0 = cmds[lane].body[0];  // word 0 of the chunk's commands as pass 1 left it (shared: the pass is behind us; else read again below)
    uint32_t i = s;
    while (i < e) {
    uint32_t stop = e;
    if (!shared && fo < flimit) {
        const uint32_t prepared = PrepareFills(S, cmds, fill_ix, flimit, fo, x0, y0, WaveId() * kMaxFrag);  // > fo, uniform
        if (prepared < flimit) stop = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(fill_ix[prepared])));
        b0 = cmds[lane].body[0];
    }
#pragma unroll 1
    for (; i < stop; ++i) {
        auto word = [&](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(i))); };
        const uint32_t tag = word(mine.tag);
        if (tag == kCmdFill) {
            const uint32_t run = min(FillRunLength(fm, i), stop - i);  // >= 1, all of them prepared
            AddFillRun<true>(S.w[0].f, b0, i, run, row, g, sa01, sa23);
            fo += run;
            i += run - 1u;
        } else if (tag == kCmdFillEdge) {
            const float sgn = static_cast<float>(static_cast<int>(word(mine.body[0])));
            const float v = sgn * Sat(py - __uint_as_float(word(mine.body[1])) + 1.0f);
            sa01.x = ToHalf(static_cast<float>(sa01.x) + v);
            sa01.y = ToHalf(static_cast<float>(sa01.y) + v);
            sa23.x = ToHalf(static_cast<float>(sa23.x) + v);
            sa23.y = ToHalf(static_cast<float>(sa23.y) + v);
        } else if (tag == kCmdLine) {
            const float sx = __uint_as_float(word(mine.body[1])), sy = __uint_as_float(word(mine.body[2]));
            const float ex = __uint_as_float(word(mine.body[3])), ey = __uint_as_float(word(mine.body[4]));
            const float lx = ex - sx, ly = ey - sy;
            const float den = lx * lx + ly * ly;
            const float dy = py - sy;
            const float lydy = ly * dy;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dx = (px0 + static_cast<float>(k)) - sx;
                const float t = Sat((lx * dx + lydy) / den);
                const float fx = lx * t - dx, fy = ly * t - dy;
                df[k] = fminf(df[k], fx * fx + fy * fy);  // (squared: see FarAway)
            }
        }
    }
    }
}

// Workgroup tiles: passes 1 and 2 for ALL the chunk's Fill commands, a quarter of them per wave (in steps of four commands), each wave
// into its own fragment region, BEFORE the items are handed out.  An item is then pass 3 and its closing command only: the expensive
// half of a Fill no longer depends on which wave happens to take which item (the 4K Tiger's longest list is 13 items of 10-18 commands:
// four waves took 4, 3, 3 and 3 of them, and the launch waited for the four).  Returns false if this wave's share does not fit its
// region (a chunk of near-vertical segments: 16 fragments per command) -- the workgroup then falls back to fragments per item.
__device__ __forceinline__ bool PrepareFillsShared(SparseLds &S, Cmd *cmds, const uint8_t *fill_ix, uint32_t nfill, uint32_t x0, uint32_t y0) {
    const uint32_t wave = WaveId();
    WaveFineLds &W = S.w[wave].f;
    const uint32_t per = ((nfill + 15u) >> 4) << 2;
    const uint32_t from = min(wave * per, nfill), to = min(from + per, nfill);
    uint32_t nfrag = 0;
    bool fits = true;
    if (to - from >= kFillPairsMin) {  // uniform over the wave
        fits = FillPairs(W, cmds, fill_ix, to, from, nfrag, y0, wave * kMaxFrag) == to;
    } else {
#pragma unroll 1
        for (uint32_t pos = from; pos < to; pos += 4u) {
            if (!FillStep(W, cmds, fill_ix, to, pos, nfrag, y0, wave * kMaxFrag)) {
                fits = false;
                break;
            }
        }
    }
    WaveSync();
    if (nfrag != 0u) FillPass2(W, cmds, nfrag, x0);  // uniform
    return fits;
}

struct PhaseTicks {
    unsigned long long a = 0, b = 0, c = 0, busy = 0;
};

// One staged chunk (n commands, parity = chunk index & 1) of a tile rendered by the workgroup;
// pix = this lane's pixel in phase B (row * 16 + x).
template <bool kProf>
__device__ __forceinline__ void RenderChunkWG(SparseLds &S, Cmd *cmds, uint8_t *fill_ix, uint32_t n, uint32_t parity, uint32_t x0,
                                              uint32_t y0, uint32_t pix, PixelRGB &st, PhaseTicks &prof, const bool share_fills) {
    const uint32_t lane = LaneId(), wave = WaveId();
    Cmd mine;  // lane i: command i of the chunk
    mine.tag = 0;
    mine.body[0] = mine.body[1] = mine.body[2] = mine.body[3] = mine.body[4] = 0;
    if (lane < n) mine = cmds[lane];
    const uint32_t tag = mine.tag;
    const uint64_t fm = __ballot(tag == kCmdFill);
    const uint64_t bm = __ballot(tag == kCmdDrawFill || tag == kCmdStroke || tag == kCmdSolid || tag == kCmdCircle);
    if (tag == kCmdFill) fill_ix[RankBelow(fm)] = static_cast<uint8_t>(lane);  // (every wave keeps its own copy)
    const uint32_t nitems = static_cast<uint32_t>(__popcll(bm));
    if (wave == 0) {
        const uint64_t below = bm & ((1ull << lane) - 1ull);
        const uint32_t begin = below ? 64u - static_cast<uint32_t>(__builtin_clzll(below)) : 0u;  // after the previous blend
        if ((bm >> lane) & 1ull) {
            // the item's colour, as the blend takes it: Circle is black with alpha exactly `alpha` (:491)
            uint2 c = make_uint2(0u, 0x3c000000u);
            if (tag == kCmdSolid) c = make_uint2(mine.body[1], mine.body[2]);
            if (tag == kCmdDrawFill || tag == kCmdStroke) c = make_uint2(mine.body[2], mine.body[3]);
            S.rec[RankBelow(bm)] = c;
            S.item_se[RankBelow(bm)] = static_cast<uint16_t>(begin | (lane << 8));
        }
        if (lane == 0) {
            const uint32_t tail = bm ? 64u - static_cast<uint32_t>(__builtin_clzll(bm)) : 0u;
            S.item_se[nitems] = static_cast<uint16_t>(tail | (n << 8));
            S.next_item = 0;
        }
    }
    WaveSync();
    const uint32_t nfill = static_cast<uint32_t>(__popcll(fm));  // (the same in every wave)
    // (only for a chunk without Line commands: a stroke's item is the longest thing in a chunk, the other waves made their items' fragments
    //  behind it anyway -- a separate pass in front of the items then only adds to the wave that takes the stroke: 4K Tiger +0.6 us with it,
    //  the fills-only Tiger -1.6 us)
    // share_fills: the launch has a wave for every queued tile (pm_fine.hip).  On a full machine the pass's barrier waits for the slowest of
    // four waves that share their SIMDs with sixteen others: measured +0.5 us on the 4K Tiger's frame, where the fills-only 1080p Tiger
    // (2 655 tiles on 5 120 waves) is 1.3 us faster with it.
    bool try_shared = share_fills && nfill >= 16u && __ballot(tag == kCmdLine) == 0ull;
    if (try_shared) {
        // ... and only if every wave's share will fit its region: the rows a Fill's segment can touch bound its fragments (a share that
        // overflows costs the pass AND the fall-back: the Tiger's longest list, whiskers crossing all 16 rows, was slower with than without)
        uint32_t rows = 0;
        if (lane < nfill) {
            const uint32_t ci = fill_ix[lane];
            const float fy0 = static_cast<float>(y0);
            const float ya = __uint_as_float(cmds[ci].body[2]) - fy0, yb = __uint_as_float(cmds[ci].body[4]) - fy0;
            const float lo = fminf(fmaxf(floorf(fminf(ya, yb)), 0.0f), 16.0f), hi = fminf(fmaxf(floorf(fmaxf(ya, yb)) + 1.0f, 0.0f), 16.0f);
            rows = (ya == ya && yb == yb) ? static_cast<uint32_t>(fmaxf(hi - lo, 0.0f)) : 16u;
        }
        const uint32_t incl = WaveInclusiveScan(rows);
        const uint32_t per = ((nfill + 15u) >> 4) << 2;  // (PrepareFillsShared's shares)
        uint32_t before = 0;
#pragma unroll
        for (uint32_t q = 0; q < static_cast<uint32_t>(kWaves); ++q) {
            const uint32_t end = min((q + 1u) * per, nfill);
            const uint32_t upto = end ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), static_cast<int>(end - 1u))) : 0u;
            if (upto - before > kMaxFrag) try_shared = false;
            before = upto;
        }
    }
    if (try_shared) {
        const bool fits = PrepareFillsShared(S, cmds, fill_ix, nfill, x0, y0);
        if (!fits && lane == 0) S.prep_over[parity] = 1u;  // (zeroed during the previous chunk, or before the tile's first barrier)
    }
    const uint32_t r4 = Opaque(lane) >> 2, g = Opaque(lane) & 3u;
    uint32_t k0 = 0;
    do {  // rounds of kAlphaSlots items
        const uint32_t kend = min(k0 + kAlphaSlots, nitems);
        const bool last_round = kend == nitems;
        const uint32_t limit = last_round ? nitems + 1u : kend;  // (the open tail goes with the last round)
        unsigned long long t_a = 0;
        if (kProf) t_a = wall_clock64();
        __syncthreads();  // phase B of the previous round (or chunk) is done with the alpha images; every wave's fragments are in place
        const bool shared = try_shared && S.prep_over[parity] == 0u;
        if (k0 == 0 && try_shared && !shared) {  // (cannot happen while the bound above holds: the words the pass left are taken back, FillPairs adds to them)
            if (wave == 0 && lane < n && tag == kCmdFill) cmds[lane].body[0] = 0;
            __syncthreads();
        }
        if (k0 == 0 && wave == 0 && lane == 0) S.prep_over[parity ^ 1u] = 0u;  // the next chunk's (nobody touches it before this chunk's last barrier)
        // ---- phase A: every wave takes the next item nobody has taken --------------------------
#pragma unroll 1
        for (;;) {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&S.next_item, 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= limit) break;
            const bool is_tail = k == nitems;
            const uint32_t se = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(S.item_se[k]));
            const uint32_t s0 = se & 0xffu, e0 = se >> 8;
            unsigned long long t_i = 0;
            if (kProf) t_i = wall_clock64();
            half2_t sa01 = Splat(static_cast<_Float16>(0.0f)), sa23 = sa01;
            float df[4] = {FarAway(), FarAway(), FarAway(), FarAway()};
            if (s0 == 0) {  // the chunk opens inside an item: its accumulators so far
                const uint2 cs = S.carry_sa[parity][lane];
                const float4 cd = S.carry_df[parity][lane];
                sa01 = Half2FromBits(cs.x); sa23 = Half2FromBits(cs.y);
                df[0] = cd.x; df[1] = cd.y; df[2] = cd.z; df[3] = cd.w;
            }
            RunItemCommands(S, cmds, mine, fill_ix, fm, s0, e0, x0, y0, sa01, sa23, df, shared);
            if (is_tail) {  // (also when the tail is empty: the next chunk starts from a clean state)
                uint2 cs;
                cs.x = __builtin_bit_cast(uint32_t, sa01); cs.y = __builtin_bit_cast(uint32_t, sa23);
                S.carry_sa[parity ^ 1u][lane] = cs;
                S.carry_df[parity ^ 1u][lane] = make_float4(df[0], df[1], df[2], df[3]);
                continue;
            }
            Cmd cmd;  // the item's closing command, from lane e0
            cmd.tag = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine.tag), static_cast<int>(e0)));
#pragma unroll
            for (int w = 0; w < 5; ++w) cmd.body[w] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine.body[w]), static_cast<int>(e0)));
            _Float16 al[4];
            if (cmd.tag == kCmdDrawFill) {  // :535-542
                const _Float16 bd = static_cast<_Float16>(static_cast<float>(static_cast<int>(cmd.body[0])));
                const half2_t s01 = sa01 + Splat(bd), s23 = sa23 + Splat(bd);
                const bool eo = (cmd.body[4] & kFillEvenOdd) != 0;
                const half2_t q01 = FillAlpha2(s01, eo), q23 = FillAlpha2(s23, eo);
                al[0] = q01.x;
                al[1] = q01.y;
                al[2] = q23.x;
                al[3] = q23.y;
            } else if (cmd.tag == kCmdStroke) {  // :500-504
                const float half_width = __uint_as_float(cmd.body[0]);
#pragma unroll
                for (int u = 0; u < 4; ++u) al[u] = ToHalf(Sat(half_width + 0.5f - StrokeDistance(df[u])));
            } else if (cmd.tag == kCmdCircle) {  // :481-490
                const float bx0 = static_cast<float>(cmd.body[1] & 0xffffu), by0 = static_cast<float>(cmd.body[1] >> 16);
                const float bx1 = static_cast<float>(cmd.body[2] & 0xffffu), by1 = static_cast<float>(cmd.body[2] >> 16);
                const float cx = bx0 + (bx1 - bx0) * 0.5f, cy = by0 + (by1 - by0) * 0.5f;
                const bool ellipse = (cmd.body[0] & kCmdCircleEllipse) != 0;
                const float dy = static_cast<float>(y0 + r4) - cy;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = static_cast<float>(x0 + 4u * g + static_cast<uint32_t>(u)) - cx;
                    al[u] = ToHalf(CircleAlpha(dx, dy, cx - bx0, cy - by0, ellipse));
                }
            } else {  // Solid (:546-549): alpha 1
#pragma unroll
                for (int u = 0; u < 4; ++u) al[u] = static_cast<_Float16>(1.0f);
            }
            uint2 v;
            v.x = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, al[0])) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, al[1])) << 16);
            v.y = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, al[2])) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, al[3])) << 16);
            S.alpha[k & (kAlphaSlots - 1u)][lane] = v;
            if (kProf) prof.busy += wall_clock64() - t_i;
        }
        __syncthreads();
        if (wave == 0 && lane == 0) S.next_item = kend;  // (nobody takes an item before the next round's barrier)
        unsigned long long t_b = 0;
        if (kProf) {
            t_b = wall_clock64();
            prof.a += t_b - t_a;
        }
        // ---- phase B: rgb = mix(rgb, fg.rgb, fg.a * alpha), items in list order ---------------------
#pragma unroll 2
        for (uint32_t k = k0; k < kend; ++k) {
            const uint2 c = S.rec[k];
            const _Float16 a = reinterpret_cast<const _Float16 *>(&S.alpha[k & (kAlphaSlots - 1u)][0])[pix];
            const _Float16 fa = HalfFromBits(c.y >> 16) * a;
            st.r = st.r + (HalfFromBits(c.x) - st.r) * fa;
            st.g = st.g + (HalfFromBits(c.x >> 16) - st.g) * fa;
            st.b = st.b + (HalfFromBits(c.y) - st.b) * fa;
        }
        if (kProf) prof.b += wall_clock64() - t_b;
        k0 = kend;
    } while (k0 < nitems);
}


// One queued tile `cur` {tile column | row << 16, first quad of its command list, first piece, that piece's candidates |
// segments << 9}: its list is built (kFused) and interpreted by the calling wave -- or, wg_mode, by the four waves of the
// workgroup together (all of them call with the same arguments; `quarter` = which 64 of the 256 pixels the wave blends in phase
// B, `parity` alternates between consecutive workgroup tiles of a workgroup).  next_card() is called once, when only the
// encoding of the tile's pixels is left: the caller's moment to ask for its next tile.  Returns the commands interpreted.
// Lds = DenseLds: every tile is a single wave's (wg_mode_in is false) and the workgroup paths are not even compiled.
template <bool kFused, bool kProf, bool kCapture, bool kCoh, typename Next, typename Lds>
__device__ __forceinline__ uint32_t RenderQueuedTile(const FrameParams &P, Lds &S, const uint4 cur, const bool wg_mode_in, const uint32_t quarter,
                                                     const uint32_t parity, const uint32_t lane, const uint32_t wave, const uint64_t lanes_below,
                                                     Next &&next_card, PhaseTicks &prof, CoarseTicks &ct, const bool share_fills = false) {
    constexpr bool kWg = std::is_same<Lds, SparseLds>::value;
    const bool wg_mode = kWg && wg_mode_in;
    // linear -> sRGB + unorm8 (:563-565): the 65,536-entry table of decision D2.  (A compact
    // LDS-resident form of the table was measured slower: twelve byte loads per lane are fewer
    // instructions than twelve decodes.)
    // A tile's pixels are encoded at its end in two steps: the table reads of all channels, then -- behind the draw of the
    // next tile -- the packing (StoreOrder's swap for a BGRA8 target: uniform, two selects per pixel).
    const uint8_t *lut = P.lut_lin2srgb;
    const bool bgra = P.fb_bgra != 0;
    const uint32_t tx = cur.x & 0xffffu, ty_rel = cur.x >> 16;  // (queue entries name a tile by column | row << 16)
    const uint32_t tile = ty_rel * P.tiles_x + tx;

    uint32_t n_cmd = 0;
    if (!kFused) n_cmd = __builtin_amdgcn_readfirstlane(P.tile_ncmd[tile]);  // pm_coarse_kernel's launch left the list's length there
    if (kFused) {
        // (wave 0 of a workgroup-mode tile has wave == 0: its region is S.w[0] either way)
        // (a workgroup tile: chunks 0..2 of the list also go to the staged-command areas of waves 1..3)
        // (a workgroup tile: all four waves -- the longest lists are built a round of 64 stream elements per wave)
        CoarseShared *shared = nullptr;
        if constexpr (kWg) shared = wg_mode ? &S.coarse_shared : nullptr;
        // (chunks of the list that stay in LDS, and the bytes from one to the next: a single wave's first chunk is its own staged-command
        //  area; the one-wave kernel has a second one per wave, S.more)
        uint32_t lds_stride = static_cast<uint32_t>(sizeof(WaveLds)), lds_n = wg_mode ? kLdsChunks : 1u;
        if constexpr (!kWg) {
            lds_stride = static_cast<uint32_t>(reinterpret_cast<uint8_t *>(S.more[wave]) - reinterpret_cast<uint8_t *>(S.w[wave].cmds));
            lds_n = kDenseLdsChunks;
        }
        n_cmd = CoarseTile<kCapture, kProf, kWg, kCoh>(P, S.w[wave].c, cur, lane, lanes_below, &ct, reinterpret_cast<uint8_t *>(wg_mode ? S.w[1].cmds : S.w[wave].cmds),
                                                  lds_stride, lds_n, shared);
        if constexpr (kWg) {
            if (wg_mode) {
                if (wave == 0 && lane == 0) {
                    S.wg_ncmd[parity & 1u] = n_cmd;
                    S.prep_over[0] = 0u;  // (the first chunk's "a share of the Fills did not fit" flag)
                }
                __syncthreads();  // (workgroup-scope release/acquire: the list wave 0 wrote is visible)
                n_cmd = S.wg_ncmd[parity & 1u];
            }
        }
        if (!wg_mode && n_cmd > (kWg ? 1u : kDenseLdsChunks) * kSpChunk) {
            // (only a list longer than the chunk in LDS is read back from HBM: this wave's stores before its loads.  The
            //  release waits for EVERY store the wave has in flight -- the previous tile's pixels among them, microseconds
            //  under load -- so the tiles that need no read-back skip it)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (kProf) prof.c = wall_clock64();
    }
    if (n_cmd == 0) next_card();
    if (n_cmd != 0) {  // 0: the coarse kernel found one opaque colour and wrote it
        const uint32_t *src = reinterpret_cast<const uint32_t *>(P.tarena + cur.y);
        const uint32_t x0 = tx * kTileW;
        const uint32_t y0 = (P.row0 + ty_rel) * kTileH;
        bool as_workgroup = false;
        if constexpr (kWg) {
        if (wg_mode) {
            as_workgroup = true;
            // the longest lists set the span of the launch: their waves win the issue arbitration
            __builtin_amdgcn_s_setprio(2);
            // phase B: lane -> 1 pixel, pix = 64 * (slot & 3) + lane = row * 16 + x
            const uint32_t pix = 64u * (quarter & 3u) + lane;
            const uint32_t pxi = x0 + (pix & 15u), pyi = y0 + (pix >> 4);
            PixelRGB s1;
            s1.r = s1.g = s1.b = static_cast<_Float16>(1.0f);
            if (wave == 0) {  // no item is open when a list starts
                S.carry_sa[0][lane] = make_uint2(OpaqueZero(), OpaqueZero());
                S.carry_df[0][lane] = make_float4(FarAway(), FarAway(), FarAway(), FarAway());
                if (!kFused && lane == 0) S.prep_over[0] = 0u;  // (fused: zeroed before the barrier behind the list building)
            }
            uint32_t parity = 0;
            for (uint32_t c0 = 0; c0 < n_cmd; c0 += kSpChunk, parity ^= 1u) {
                const uint32_t m = min(kSpChunk, n_cmd - c0);
                Cmd *chunk = S.w[0].cmds;
                if (!kFused || c0 != 0) __syncthreads();  // the previous chunk (or tile) is done with the shared tables
                if (kFused && c0 < kLdsChunks * kSpChunk) {
                    chunk = S.w[1u + c0 / kSpChunk].cmds;  // CoarseTile left it there (visible since the barrier after it)
                } else {
                    const uint2 *g = reinterpret_cast<const uint2 *>(src + 6u * c0);
                    uint2 *l = reinterpret_cast<uint2 *>(S.w[0].cmds);
                    for (uint32_t w = Opaque(threadIdx.x); w < 3u * m; w += kThreads) l[w] = g[w];  // (Opaque: no hoisted address to spill)
                    __syncthreads();
                }
                RenderChunkWG<kProf>(S, chunk, S.w[wave].f.fill_ix, m, parity, x0, y0, pix, s1, prof, share_fills);
            }
            __syncthreads();  // the other waves may still read this wave's alpha images
            __builtin_amdgcn_s_setprio(0);
            // (the table reads first, the draw while they are in flight: see the single-wave path below)
            const uint32_t r8 = lut[static_cast<uint32_t>(__builtin_bit_cast(uint16_t, s1.r))];
            const uint32_t g8 = lut[static_cast<uint32_t>(__builtin_bit_cast(uint16_t, s1.g))];
            const uint32_t b8 = lut[static_cast<uint32_t>(__builtin_bit_cast(uint16_t, s1.b))];
            next_card();
            if (pyi < P.height && pxi < P.width) {
                uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + (pix >> 4)) * P.fb_stride + static_cast<size_t>(pxi) * 4;
                StorePixel(dst, (bgra ? b8 : r8) | (g8 << 8) | ((bgra ? r8 : b8) << 16) | 0xff000000u);
            }
        }
        }
        if (!as_workgroup) {
            // lane -> 4 pixels: x = x0 + 4 * (lane & 3) + k, row = lane / 4
            // (from a lane number the compiler cannot see through: `(lane & 3) * 4` hoisted out of the tile loop was SPILLED, and its
            //  reload -- scratch_load + s_waitcnt vmcnt(0) -- made every tile begin by waiting for the previous tile's pixel stores)
            const uint32_t ol = Opaque(lane);
            const uint32_t pxi = x0 + (ol & 3u) * 4u;
            const uint32_t prow = ol >> 2;
            const uint32_t pyi = y0 + prow;
            Cmd *const cmds = S.w[wave].cmds;
            PixelStateS st;
            st.r01 = st.r23 = st.g01 = st.g23 = st.b01 = st.b23 = Splat(static_cast<_Float16>(1.0f));
            st.sa01 = st.sa23 = Splat(static_cast<_Float16>(0.0f));
#pragma unroll
            for (int k = 0; k < 4; ++k) st.df[k] = FarAway();
            for (uint32_t c0 = 0; c0 < n_cmd; c0 += kSpChunk) {
                const uint32_t m = min(kSpChunk, n_cmd - c0);
                WaveSync();
                Cmd *chunk = cmds;
                bool in_lds = kFused && c0 == 0;  // (the fused kernel's CoarseTile left the first chunk right here)
                if constexpr (!kWg) {
                    if (kFused && c0 == kSpChunk) {  // ... and the one-wave kernel's the second one next to it
                        chunk = S.more[wave];
                        in_lds = true;
                    }
                }
                if (!in_lds) {
                    const uint2 *g = reinterpret_cast<const uint2 *>(src + 6u * c0);
                    uint2 *l = reinterpret_cast<uint2 *>(cmds);
                    for (uint32_t w = Opaque(lane); w < 3u * m; w += 64u) l[w] = g[w];  // (Opaque: no hoisted address to spill)
                }
                WaveSync();
                InterpretSparse(S, chunk, S.w[wave].f.fill_ix, m, x0, y0, st);
            }
            // The twelve table reads of the pixels' encoding are requested FIRST, the draw of the next tile goes out
            // while they are in flight (its wait is theirs too), and the next tile's queue entry is on its way while the
            // bytes are packed and stored: two round trips at the end of a tile, not three.
            uint32_t lv[12];
            {
                const _Float16 pr[4] = {st.r01.x, st.r01.y, st.r23.x, st.r23.y}, pg[4] = {st.g01.x, st.g01.y, st.g23.x, st.g23.y},
                               pb[4] = {st.b01.x, st.b01.y, st.b23.x, st.b23.y};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lv[3 * k] = lut[static_cast<uint32_t>(__builtin_bit_cast(uint16_t, pr[k]))];
                    lv[3 * k + 1] = lut[static_cast<uint32_t>(__builtin_bit_cast(uint16_t, pg[k]))];
                    lv[3 * k + 2] = lut[static_cast<uint32_t>(__builtin_bit_cast(uint16_t, pb[k]))];
                }
            }
            next_card();
            if (pyi < P.height && pxi < P.width) {
                uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + prow) * P.fb_stride + static_cast<size_t>(pxi) * 4;
                uint4 out;
                // (R and B change places for a BGRA8 target: uniform, two selects per pixel)
                auto pack = [&](uint32_t r8, uint32_t g8, uint32_t b8) { return (bgra ? b8 : r8) | (g8 << 8) | ((bgra ? r8 : b8) << 16) | 0xff000000u; };
                out.x = pack(lv[0], lv[1], lv[2]);
                out.y = pack(lv[3], lv[4], lv[5]);
                out.z = pack(lv[6], lv[7], lv[8]);
                out.w = pack(lv[9], lv[10], lv[11]);
                if (pxi + 4 <= P.width && P.fb_vec16) {
                    StorePixels4(dst, out);
                } else {
                    const uint32_t o[4] = {out.x, out.y, out.z, out.w};
                    for (uint32_t k = 0; k < 4 && pxi + k < P.width; ++k) reinterpret_cast<uint32_t *>(dst)[k] = o[k];
                }
            }
        }
    }
    return n_cmd;
}

}  // namespace

}  // namespace pm
