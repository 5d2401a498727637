// pm_fine_kernel (+ pm_clear_kernel): renderKernel and the composite
// (see pm_kernels_common.h for the decomposition and the rules shared by the three files)
#include "pm_kernels_common.h"

namespace pm {

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
namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

#ifndef PM_WAVE_CMDS
#define PM_WAVE_CMDS 256
#endif
constexpr uint32_t kFineChunk = PM_WAVE_CMDS;  // commands staged in LDS per wave by the interpreter

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
constexpr uint32_t kSpChunk = 64;   // commands staged per chunk
constexpr uint32_t kMaxFrag = 128;  // fragment slots per wave (one step of pass 1 adds <= 64)

struct SparseLds {
    Cmd cmds[kWaves][kSpChunk];
    float4 fparam[kWaves * kMaxFrag];     // {tx, ty, wx - wy, bits(command index)}
    uint2 contrib[kWaves * kMaxFrag][4];  // 16 binary16 contributions per fragment (x = 0..15)
    uint8_t fill_ix[kWaves][kSpChunk];    // indices of the chunk's Fill commands, in order
};

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

// Passes 1 and 2 for the Fill commands [from, ...) of the staged chunk.  Returns the ordinal
// of the first Fill NOT covered.  kWG: the four waves of the workgroup share the work (two
// steps of pass 1 each per call) and the call contains two workgroup barriers.
template <bool kWG>
__device__ __forceinline__ uint32_t PrepareFills(SparseLds &S, Cmd *cmds, const uint8_t *fill_ix, uint32_t nfill, uint32_t from,
                                                  uint32_t x0, uint32_t y0) {
    const uint32_t lane = LaneId(), wave = threadIdx.x >> 6;
    const uint32_t rb = wave * kMaxFrag;
    uint32_t nfrag = 0;
    auto step = [&](uint32_t pos) {
        const uint32_t q = lane >> 4, row = lane & 15u;
        const uint32_t fi = pos + q;
        const bool valid = fi < nfill;
        const uint32_t ci = fill_ix[valid ? fi : pos];
        const float py = static_cast<float>(y0 + row);
        const float sy = __uint_as_float(cmds[ci].body[2]) - py;
        const float ey = __uint_as_float(cmds[ci].body[4]) - py;
        const float wx = Sat(sy), wy = Sat(ey);
        const bool live = valid && wx != wy;
        const uint64_t mask = __ballot(live);
        if (mask == 0) return;  // (the staged body[0] of a Fill is 0: no row, nothing to add)
        if (live) {
            const float tx = (wx - sy) / (ey - sy);
            const float ty = (wy - sy) / (ey - sy);
            S.fparam[rb + nfrag + RankBelow(mask)] = make_float4(tx, ty, wx - wy, __uint_as_float(ci));
        }
        if (row == 0 && valid) {
            const uint32_t gm = static_cast<uint32_t>(mask >> (16u * q)) & 0xffffu;
            const uint32_t gb = rb + nfrag + static_cast<uint32_t>(__popcll(mask & ((1ull << (16u * q)) - 1ull)));
            cmds[ci].body[0] = gm | (gb << 16);
        }
        nfrag += static_cast<uint32_t>(__popcll(mask));
    };
    uint32_t done;
    if (kWG) {
        __syncthreads();  // every wave is through with the contributions of the previous call
#pragma unroll 1
        for (uint32_t b = 0; b < 2; ++b) {
            const uint32_t pos = from + 4u * (wave + kWaves * b);
            if (pos < nfill) step(pos);
        }
        done = min(from + 8u * kWaves, nfill);
    } else {
        uint32_t pos = from;
#pragma unroll 1
        while (pos < nfill && nfrag + 64u <= kMaxFrag) {
            step(pos);
            pos += 4u;
        }
        done = min(pos, nfill);
    }
    WaveSync();
#pragma unroll 1
    for (uint32_t f0 = 0; f0 < nfrag; f0 += 16u) {
        const uint32_t f = f0 + (lane >> 2), g = lane & 3u;
        if (f < nfrag) {
            const float4 p = S.fparam[rb + f];
            const uint32_t ci = __float_as_uint(p.w);
            const float fsx = __uint_as_float(cmds[ci].body[1]), fex = __uint_as_float(cmds[ci].body[3]);
            const float px0 = static_cast<float>(x0 + 4u * g);
            _Float16 h[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = FillContribution(fsx, fex, px0 + static_cast<float>(k), p.x, p.y, p.z);
            uint2 v;
            v.x = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h[0])) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h[1])) << 16);
            v.y = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h[2])) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h[3])) << 16);
            S.contrib[rb + f][g] = v;
        }
    }
    if (kWG) __syncthreads(); else WaveSync();
    return done;
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
__device__ __forceinline__ void InterpretSparse(SparseLds &S, Cmd *cmds, const uint8_t *fill_ix, uint32_t n, uint32_t x0, uint32_t y0,
                                                PixelStateS &st) {
    const uint32_t lane = LaneId();
    const uint32_t row = lane >> 2, g = lane & 3u;
    const float px0 = static_cast<float>(x0 + 4u * g), py = static_cast<float>(y0 + row);
    // the chunk's Fill commands, in order
    const bool isf = lane < n && cmds[lane].tag == kCmdFill;
    const uint64_t fm = __ballot(isf);
    if (isf) const_cast<uint8_t *>(fill_ix)[RankBelow(fm)] = static_cast<uint8_t>(lane);
    const uint32_t nfill = static_cast<uint32_t>(__popcll(fm));
    WaveSync();
    uint32_t fo = 0, prepared = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const Cmd cmd = cmds[i];
        switch (cmd.tag) {
            case kCmdCircle: {
                const float bx0 = static_cast<float>(cmd.body[1] & 0xffffu), by0 = static_cast<float>(cmd.body[1] >> 16);
                const float bx1 = static_cast<float>(cmd.body[2] & 0xffffu), by1 = static_cast<float>(cmd.body[2] >> 16);
                const float cx = bx0 + (bx1 - bx0) * 0.5f, cy = by0 + (by1 - by0) * 0.5f;
                const float circle_r = fminf(cx - bx0, cy - by0);
                const float dy = py - cy;
                _Float16 alpha[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dx = (px0 + static_cast<float>(k)) - cx;
                    alpha[k] = ToHalf(Sat(circle_r - sqrtf(dx * dx + dy * dy)));
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
            case kCmdStroke: {
                const float half_width = __uint_as_float(cmd.body[0]);
                _Float16 alpha[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    alpha[k] = ToHalf(Sat(half_width + 0.5f - st.df[k]));
                    st.df[k] = 1e9f;
                }
                half2_t a01, a23;
                a01.x = alpha[0]; a01.y = alpha[1]; a23.x = alpha[2]; a23.y = alpha[3];
                Blend4S(st, cmd.body[2], cmd.body[3], a01, a23);
                break;
            }
            case kCmdFill: {
                if (fo >= prepared) prepared = PrepareFills<false>(S, cmds, fill_ix, nfill, fo, x0, y0);  // uniform
                ++fo;
                const uint32_t hdr = cmds[i].body[0];  // row mask | first fragment << 16 (pass 1)
                if ((hdr >> row) & 1u) {
                    const uint32_t f = (hdr >> 16) + static_cast<uint32_t>(__popc(hdr & ((1u << row) - 1u)));
                    const uint2 v = S.contrib[f][g];
                    st.sa01 = st.sa01 + Half2FromBits(v.x);
                    st.sa23 = st.sa23 + Half2FromBits(v.y);
                }
                break;
            }
            case kCmdFillEdge: {
                const float sgn = static_cast<float>(static_cast<int>(cmd.body[0]));
                const float v = sgn * Sat(py - __uint_as_float(cmd.body[1]) + 1.0f);
                st.sa01.x = ToHalf(static_cast<float>(st.sa01.x) + v);
                st.sa01.y = ToHalf(static_cast<float>(st.sa01.y) + v);
                st.sa23.x = ToHalf(static_cast<float>(st.sa23.x) + v);
                st.sa23.y = ToHalf(static_cast<float>(st.sa23.y) + v);
                break;
            }
            case kCmdDrawFill: {
                const _Float16 bd = static_cast<_Float16>(static_cast<float>(static_cast<int>(cmd.body[0])));
                const half2_t s01 = st.sa01 + Splat(bd), s23 = st.sa23 + Splat(bd);
                half2_t a01, a23;
                a01.x = ToHalf(fminf(fabsf(static_cast<float>(s01.x)), 1.0f));
                a01.y = ToHalf(fminf(fabsf(static_cast<float>(s01.y)), 1.0f));
                a23.x = ToHalf(fminf(fabsf(static_cast<float>(s23.x)), 1.0f));
                a23.y = ToHalf(fminf(fabsf(static_cast<float>(s23.y)), 1.0f));
                st.sa01 = st.sa23 = Splat(static_cast<_Float16>(0.0f));
                Blend4S(st, cmd.body[2], cmd.body[3], a01, a23);
                break;
            }
            case kCmdSolid: {
                const half2_t one = Splat(static_cast<_Float16>(1.0f));
                Blend4S(st, cmd.body[1], cmd.body[2], one, one);
                break;
            }
            default:
                break;
        }
    }
}

// The same loop for a quarter of a tile (4 pixel rows, 1 pixel per lane): tiles with long lists,
// all four waves of the workgroup walk the list together (PrepareFills<true> has barriers).
__device__ __forceinline__ void InterpretSparseWG(SparseLds &S, Cmd *cmds, uint8_t *fill_ix, uint32_t n, uint32_t x0, uint32_t y0,
                                                  uint32_t row, uint32_t xi, PixelState1 &st) {
    const uint32_t lane = LaneId();
    const float px = static_cast<float>(x0 + xi), py = static_cast<float>(y0 + row);
    const bool isf = lane < n && cmds[lane].tag == kCmdFill;
    const uint64_t fm = __ballot(isf);
    if (isf) fill_ix[RankBelow(fm)] = static_cast<uint8_t>(lane);  // (every wave keeps its own copy)
    const uint32_t nfill = static_cast<uint32_t>(__popcll(fm));
    WaveSync();
    uint32_t fo = 0, prepared = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const Cmd cmd = cmds[i];
        switch (cmd.tag) {
            case kCmdCircle: {
                const float bx0 = static_cast<float>(cmd.body[1] & 0xffffu), by0 = static_cast<float>(cmd.body[1] >> 16);
                const float bx1 = static_cast<float>(cmd.body[2] & 0xffffu), by1 = static_cast<float>(cmd.body[2] >> 16);
                const float cx = bx0 + (bx1 - bx0) * 0.5f, cy = by0 + (by1 - by0) * 0.5f;
                const float dx = px - cx, dy = py - cy;
                const float r = sqrtf(dx * dx + dy * dy);
                const _Float16 alpha = ToHalf(Sat(fminf(cx - bx0, cy - by0) - r));
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
                if (fo >= prepared) prepared = PrepareFills<true>(S, cmds, fill_ix, nfill, fo, x0, y0);  // uniform over the workgroup
                ++fo;
                const uint32_t hdr = cmds[i].body[0];
                if ((hdr >> row) & 1u) {
                    const uint32_t f = (hdr >> 16) + static_cast<uint32_t>(__popc(hdr & ((1u << row) - 1u)));
                    st.sa = st.sa + reinterpret_cast<const _Float16 *>(&S.contrib[f][0])[xi];
                }
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

}  // namespace

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
            uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + prow) * P.fb_stride + static_cast<size_t>(pxi) * 4;
            // stage the list through LDS in chunks (24-byte commands, 8-byte aligned: copied as
            // 64-bit words, coalesced) and interpret; the interpreter state stays in registers
            auto stage = [&](uint32_t c0, uint32_t m) {
                WaveSync();
                const uint2 *g = reinterpret_cast<const uint2 *>(src + 6u * c0);
                uint2 *l = reinterpret_cast<uint2 *>(cmds);
                for (uint32_t w = lane; w < 3u * m; w += 64u) l[w] = g[w];
                WaveSync();
            };
            if (quarter) {  // (the two pixel layouts keep their state in separate live ranges)
                PixelState1 s1;
                s1.r = s1.g = s1.b = static_cast<_Float16>(1.0f);
                s1.df = 1e9f;
                s1.sa = static_cast<_Float16>(0.0f);
                for (uint32_t c0 = 0; c0 < n_cmd; c0 += kFineChunk) {
                    const uint32_t m = min(kFineChunk, n_cmd - c0);
                    stage(c0, m);
                    Interpret1(cmds, m, px0, py, s1);
                }
                if (lane_on && pyi < P.height && pxi < P.width) *reinterpret_cast<uint32_t *>(dst) = enc(s1.r, s1.g, s1.b);
            } else {
                PixelState st;
                st.r01 = st.r23 = st.g01 = st.g23 = st.b01 = st.b23 = Splat(static_cast<_Float16>(1.0f));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    st.df[k] = 1e9f;
                    st.sa[k] = static_cast<_Float16>(0.0f);
                }
                for (uint32_t c0 = 0; c0 < n_cmd; c0 += kFineChunk) {
                    const uint32_t m = min(kFineChunk, n_cmd - c0);
                    stage(c0, m);
                    Interpret(cmds, m, px0, py, st);
                }
                if (pyi < P.height && pxi < P.width) {
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

// K3 (default): the same interpreter with row-sparse Fill evaluation (see PrepareFills above).
// Slot scheme as in pm_fine_kernel: the four waves of a workgroup take four consecutive slots
// of the same pass; a tile with a long list owns four aligned slots, i.e. exactly one workgroup.
__global__ __launch_bounds__(kThreads, PM_FINE_WPS) void pm_fine_sparse_kernel(FrameParams P) {
    __shared__ SparseLds S;
    if (blockIdx.x >= P.fine_grid) {
        ClearStripRow(P, blockIdx.x - P.fine_grid);
        return;
    }
    const uint32_t lane = LaneId(), wave = threadIdx.x >> 6;
    const uint32_t n_a = P.ctr_cur->vheavy_count, n_b = P.ctr_cur->heavy_count, n_c = P.ctr_cur->light_count;
    const uint32_t wave_global = blockIdx.x * kWaves + wave;
    const uint32_t n_waves = P.fine_grid * kWaves;
    // more long lists than workgroups: splitting a tile only costs work, every tile gets one wave
    const bool dense = n_a + n_b >= n_waves || P.split_mode == 0;
    const uint32_t sh = dense ? 0u : 2u;
    const uint32_t n_heavy = n_a + n_b;
    const uint32_t s_h = n_heavy << sh;
    const uint32_t n_slots = s_h + n_c;
    const uint8_t *lut = P.lut_lin2srgb;
    auto enc = [&](_Float16 r, _Float16 g, _Float16 b) -> uint32_t {
        return static_cast<uint32_t>(lut[__builtin_bit_cast(uint16_t, r)]) |
               (static_cast<uint32_t>(lut[__builtin_bit_cast(uint16_t, g)]) << 8) |
               (static_cast<uint32_t>(lut[__builtin_bit_cast(uint16_t, b)]) << 16) | 0xff000000u;
    };
    // slot -> queue entry; queues A and B are one class here (4 waves per tile)
    auto slot_entry = [&](uint32_t slot) -> uint32_t {
        if (slot < s_h) {
            const uint32_t t = slot >> sh;
            return t < n_a ? t : P.queue_cap + (t - n_a);
        }
        return 2u * P.queue_cap + (slot - s_h);
    };
    auto pass_slot = [&](uint32_t pass) -> uint32_t {
        return pass * n_waves + ((pass & 1u) ? (n_waves - 1u - wave_global) : wave_global);
    };
    uint32_t slot = pass_slot(0);
    uint4 qe = make_uint4(0u, 0u, 0u, 0u);
    if (slot < n_slots) qe = P.queue[slot_entry(slot)];
    for (uint32_t pass = 0; pass * n_waves < n_slots; ++pass) {
        const uint32_t cur_slot = slot;
        const uint4 cur = qe;
        slot = pass_slot(pass + 1u);
        if ((pass + 1u) * n_waves < n_slots && slot < n_slots) qe = P.queue[slot_entry(slot)];
        if (cur_slot >= n_slots) continue;
        const bool wg_mode = cur_slot < s_h && sh != 0;  // uniform over the workgroup (slots are aligned)
        const uint32_t tile = cur.x;
        unsigned long long t_begin = 0;
        if (P.dbg_time) t_begin = wall_clock64();
        const uint32_t n_cmd = cur.w;
        if (n_cmd != 0) {  // 0: the coarse kernel found one opaque colour and wrote it
            const uint32_t *src = reinterpret_cast<const uint32_t *>(P.ptcl + cur.y);
            const uint32_t tx = tile % P.tiles_x;
            const uint32_t ty_rel = tile / P.tiles_x;
            const uint32_t x0 = tx * kTileW;
            const uint32_t y0 = (P.row0 + ty_rel) * kTileH;
            if (wg_mode) {
                // lane -> 1 pixel: x = x0 + (lane & 15), row = 4 * (slot & 3) + lane / 16
                const uint32_t xi = lane & 15u;
                const uint32_t prow = 4u * (cur_slot & 3u) + (lane >> 4);
                const uint32_t pxi = x0 + xi, pyi = y0 + prow;
                PixelState1 s1;
                s1.r = s1.g = s1.b = static_cast<_Float16>(1.0f);
                s1.df = 1e9f;
                s1.sa = static_cast<_Float16>(0.0f);
                for (uint32_t c0 = 0; c0 < n_cmd; c0 += kSpChunk) {
                    const uint32_t m = min(kSpChunk, n_cmd - c0);
                    __syncthreads();  // the previous chunk (or tile) is done with the shared arrays
                    {
                        const uint2 *g = reinterpret_cast<const uint2 *>(src + 6u * c0);
                        uint2 *l = reinterpret_cast<uint2 *>(S.cmds[0]);
                        for (uint32_t w = threadIdx.x; w < 3u * m; w += kThreads) l[w] = g[w];
                    }
                    __syncthreads();
                    InterpretSparseWG(S, S.cmds[0], S.fill_ix[wave], m, x0, y0, prow, xi, s1);
                }
                __syncthreads();  // the other waves may still read this wave's fragments
                if (pyi < P.height && pxi < P.width) {
                    uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + prow) * P.fb_stride + static_cast<size_t>(pxi) * 4;
                    *reinterpret_cast<uint32_t *>(dst) = enc(s1.r, s1.g, s1.b);
                }
            } else {
                // lane -> 4 pixels: x = x0 + 4 * (lane & 3) + k, row = lane / 4
                const uint32_t pxi = x0 + (lane & 3u) * 4u;
                const uint32_t prow = lane >> 2;
                const uint32_t pyi = y0 + prow;
                Cmd *const cmds = S.cmds[wave];
                PixelStateS st;
                st.r01 = st.r23 = st.g01 = st.g23 = st.b01 = st.b23 = Splat(static_cast<_Float16>(1.0f));
                st.sa01 = st.sa23 = Splat(static_cast<_Float16>(0.0f));
#pragma unroll
                for (int k = 0; k < 4; ++k) st.df[k] = 1e9f;
                for (uint32_t c0 = 0; c0 < n_cmd; c0 += kSpChunk) {
                    const uint32_t m = min(kSpChunk, n_cmd - c0);
                    WaveSync();
                    {
                        const uint2 *g = reinterpret_cast<const uint2 *>(src + 6u * c0);
                        uint2 *l = reinterpret_cast<uint2 *>(cmds);
                        for (uint32_t w = lane; w < 3u * m; w += 64u) l[w] = g[w];
                    }
                    WaveSync();
                    InterpretSparse(S, cmds, S.fill_ix[wave], m, x0, y0, st);
                }
                if (pyi < P.height && pxi < P.width) {
                    uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + prow) * P.fb_stride + static_cast<size_t>(pxi) * 4;
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
        }
        if (P.dbg_time && lane == 0) {
            unsigned long long *d = P.dbg_time + 4ull * cur_slot;
            d[0] = t_begin;
            d[1] = wall_clock64();
            d[2] = tile | (wg_mode ? 0x80000000u : 0u);
            d[3] = (static_cast<unsigned long long>(wave_global) << 32) | n_cmd;
        }
    }
}

// ---- launch wrappers (called from pm_context.hip) -----------------------------------------

void LaunchClear(const FrameParams &p, uint32_t n_striprows, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    PM_LAUNCH(pm_clear_kernel, dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, p);
}

void LaunchFine(const FrameParams &p, uint32_t clear_blocks, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    if (p.fine_sparse)
        PM_LAUNCH(pm_fine_sparse_kernel, dim3(p.fine_grid + clear_blocks), dim3(kThreads), stream, t0, t1, p);
    else
        PM_LAUNCH(pm_fine_kernel, dim3(p.fine_grid + clear_blocks), dim3(kThreads), stream, t0, t1, p);
}

}  // namespace pm
