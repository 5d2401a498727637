// Hand-written HIP kernels for gfx950 (CDNA4, wave64): the compute path of
// piet-metal re-designed for MI355X.
//
// Reference semantics being reproduced (bit-exact against oracle/):
//   tileKernel   TestApp/PietRender.metal:160-454  (+ TileEncoder :69-157)
//   renderKernel TestApp/PietRender.metal:457-566  (+ stroke/renderDf :49-60)
//   composite    TestApp/PietRender.metal:16-44
//
// Decomposition (NOT the reference's thread-per-tile / 256 MiB tile buffer):
//
//   pm_bin_kernel   one 256-thread workgroup per strip row (16 tiles x 1 tile).
//       - item bboxes vs strip row: wave64 ballots + prefix ranks compact the
//         candidate items in paint order;
//       - ALL segments of ALL candidates form one flat stream; every lane
//         evaluates the reference's "phase 1" segment vote (PietRender.metal
//         :258-295 fills, :374-399 polylines) for one stream element, votes
//         are compacted in order and the surviving segments (16 B each) are
//         appended to a bump-allocated arena record in HBM;
//       - tiles no item touches are cleared to the background right here with
//         16-byte coalesced stores; the others are pushed on a tile queue.
//   pm_tile_kernel  persistent 256-thread workgroups pull tiles off the queue.
//       - candidates are filtered by a per-tile hit bit, their surviving
//         segments again form one flat stream; each lane runs the reference's
//         "phase 2" test for (tile, segment) (:302-357, :406-440) and emits
//         0..3 commands; block-wide scans give every command its slot in an
//         LDS-resident command list (never written to HBM);
//       - the same 256 threads then become the tile's 256 pixels and interpret
//         the list (renderKernel), in the command order the reference defines,
//         with binary16 accumulators (native v_*_f16, no contraction);
//       - opaque-solid detection (TileEncoder::encodeSolid/end) is tracked
//         uniformly so Bail tiles are written as one constant.
//
// Compile with -ffp-contract=off: every source-level f32/f16 operation is one
// IEEE rounding, as in the oracle.
#include <hip/hip_runtime.h>

#include "pm_device.h"

namespace pm {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr uint32_t kMaxPending = 768;  // LDS command slots: one stream round emits <= 3*256

// ---------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t LaneId() { return __lane_id(); }

__device__ __forceinline__ uint32_t RankBelow(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

__device__ __forceinline__ uint32_t WaveInclusiveScan(uint32_t v) {
    const uint32_t lane = LaneId();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= static_cast<uint32_t>(d)) v += t;
    }
    return v;
}

__device__ __forceinline__ float Sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
__device__ __forceinline__ float Sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// "not all four corners strictly on one side" test used throughout tileKernel
// (PietRender.metal:241, :289, :340, :349, :394, :431).
__device__ __forceinline__ bool Straddles(float s00, float s01, float s10, float s11) {
    return s00 * s01 + s00 * s10 + s00 * s11 < 3.0f;
}

__device__ __forceinline__ uint32_t LoadU32(const uint8_t *p) { return *reinterpret_cast<const uint32_t *>(p); }
__device__ __forceinline__ float LoadF32(const uint8_t *p) { return *reinterpret_cast<const float *>(p); }
__device__ __forceinline__ float2 LoadF2(const uint8_t *p) { return *reinterpret_cast<const float2 *>(p); }

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

// Block-wide ordered rank of a predicate (256 threads, 4 waves).  s_part must
// hold kWaves words.  Contains two barriers.
__device__ __forceinline__ uint32_t BlockRank(bool pred, uint32_t *s_part, uint32_t *total) {
    const uint64_t m = __ballot(pred);
    const uint32_t wave = threadIdx.x >> 6;
    if (LaneId() == 0) s_part[wave] = static_cast<uint32_t>(__popcll(m));
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t v = s_part[w];
        if (w < static_cast<int>(wave)) base += v;
        tot += v;
    }
    __syncthreads();
    *total = tot;
    return base + RankBelow(m);
}

// Block-wide exclusive scan of arbitrary u32 values.  Two barriers.
__device__ __forceinline__ uint32_t BlockExclusiveScan(uint32_t v, uint32_t *s_part, uint32_t *total) {
    const uint32_t incl = WaveInclusiveScan(v);
    const uint32_t wave = threadIdx.x >> 6;
    if (LaneId() == 63) s_part[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t x = s_part[w];
        if (w < static_cast<int>(wave)) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// Largest c in [0, n) with off[c] <= e (off ascending, off[0] == 0).
__device__ __forceinline__ uint32_t FindOwner(const uint32_t *off, uint32_t n, uint32_t e) {
    uint32_t lo = 0, hi = n;  // invariant: off[lo] <= e, (hi == n or off[hi] > e)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

}  // namespace

// =====================================================================================
// K1: binning, one workgroup per strip row
// =====================================================================================

__global__ __launch_bounds__(kThreads) void pm_bin_kernel(FrameParams P) {
    __shared__ uint32_t s_part[kWaves];
    __shared__ uint32_t s_cidx[kThreads];   // candidate item index
    __shared__ uint32_t s_cmask[kThreads];  // candidate per-tile hit mask (16 bits)
    __shared__ uint32_t s_ctag[kThreads];
    __shared__ uint32_t s_cpts[kThreads];   // points_ix (or item byte offset for lines)
    __shared__ uint32_t s_cnpt[kThreads];   // n_points as stored
    __shared__ float s_chw[kThreads];       // 0.5*width + 0.5 for polylines
    __shared__ uint32_t s_coff[kThreads + 1];
    __shared__ uint32_t s_hitmask;
    __shared__ uint32_t s_rec;
    __shared__ uint32_t s_qbase;

    const uint32_t tid = threadIdx.x;
    const uint32_t strip = blockIdx.x % P.strips_x;
    const uint32_t row_rel = blockIdx.x / P.strips_x;
    const uint32_t ty = P.row0 + row_rel;
    const int sx0 = static_cast<int>(strip * kGroupW);
    const int y0 = static_cast<int>(ty * kTileH);
    const int sy0 = y0 & ~static_cast<int>(kGroupH - 1);

    if (blockIdx.x == 0 && tid == 0) {
        // The counters of the NEXT frame (the other parity) are idle now: reset them
        // here so that no separate memset launch is needed.
        P.ctr_next->arena_top = kArenaBase;
        P.ctr_next->queue_count = 0;
        P.ctr_next->overflow = 0;
    }
    if (tid == 0) {
        s_hitmask = 0;
        P.striprow_head[blockIdx.x] = 0;
    }
    __syncthreads();

    const uint8_t *scene = P.scene;
    const uint32_t n_items = LoadU32(scene);
    const uint32_t items_ix = LoadU32(scene + 4);
    uint32_t *link = &P.striprow_head[blockIdx.x];  // where the next record offset goes

    for (uint32_t ib = 0; ib < n_items; ib += kThreads) {
        // ---- candidate items of this batch, in paint order ----------------------
        const uint32_t i = ib + tid;
        bool cand = false;
        uint32_t mask = 0;
        if (i < n_items) {
            const uint2 bb = *reinterpret_cast<const uint2 *>(scene + 8 + static_cast<size_t>(i) * 8);
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
        uint32_t ncand;
        const uint32_t cpos = BlockRank(cand, s_part, &ncand);
        if (ncand == 0) continue;  // uniform
        if (cand) {
            s_cidx[cpos] = i;
            s_cmask[cpos] = mask;
            atomicOr(&s_hitmask, mask);
        }
        __syncthreads();

        // ---- candidate headers + stream offsets ------------------------------------
        uint32_t nseg = 0;
        uint32_t tag = 0, rgba = 0, aux0 = 0, aux1 = 0;
        if (tid < ncand) {
            const uint32_t idx = s_cidx[tid];
            const uint8_t *item = scene + items_ix + static_cast<size_t>(idx) * kItemSize;
            tag = LoadU32(item) & 0xffffu;
            nseg = 1;  // every candidate owns >= 1 stream element (keeps seg_off dense)
            uint32_t pts = 0, npt = 0;
            float hw = 0.0f;
            if (tag == kItemCircle) {
                const uint2 bb = *reinterpret_cast<const uint2 *>(scene + 8 + static_cast<size_t>(idx) * 8);
                aux0 = bb.x;
                aux1 = bb.y;
            } else if (tag == kItemLine) {
                rgba = LoadU32(item + 8);
                aux0 = LoadU32(item + 12);  // width bits
                pts = items_ix + idx * static_cast<uint32_t>(kItemSize) + 16;  // start,end live in the item
                npt = 2;
            } else if (tag == kItemFill) {
                rgba = LoadU32(item + 8);
                npt = LoadU32(item + 12);
                pts = LoadU32(item + 16);
                if (npt > 1) nseg = npt;
            } else if (tag == kItemPoly) {
                rgba = LoadU32(item + 4);
                aux0 = LoadU32(item + 8);  // width bits
                npt = LoadU32(item + 12);
                pts = LoadU32(item + 16);
                hw = 0.5f * __uint_as_float(aux0) + 0.5f;
                if (npt > 2) nseg = npt - 1;
            } else {
                tag = 0;
            }
            s_ctag[tid] = tag;
            s_cpts[tid] = pts;
            s_cnpt[tid] = npt;
            s_chw[tid] = hw;
        }
        uint32_t total_seg;
        const uint32_t off = BlockExclusiveScan(nseg, s_part, &total_seg);
        if (tid < ncand) s_coff[tid] = off;
        if (tid == 0) {
            s_coff[ncand] = total_seg;
            const uint32_t size = kRecHdrDwords + kCandDwords * ncand + 4u * total_seg;
            const uint32_t rec = atomicAdd(&P.ctr_cur->arena_top, size);
            if (rec + size > P.arena_cap || rec + size < rec) {
                P.ctr_cur->overflow = 1;
                s_rec = 0;
            } else {
                s_rec = rec;
                *link = rec;
                P.arena[rec + 0] = 0;       // next
                P.arena[rec + 1] = ncand;
                P.arena[rec + 2] = total_seg;
            }
        }
        __syncthreads();
        const uint32_t rec = s_rec;
        if (rec == 0) break;  // arena exhausted (host re-renders with a larger arena)
        link = &P.arena[rec];
        uint32_t *cand_rec = P.arena + rec + kRecHdrDwords;
        float4 *segs = reinterpret_cast<float4 *>(P.arena + rec + kRecHdrDwords + kCandDwords * ncand);
        if (tid < ncand) {
            uint32_t *cr = cand_rec + kCandDwords * tid;
            cr[0] = tag | (s_cmask[tid] << 16);
            cr[1] = rgba;
            cr[2] = aux0;
            cr[3] = aux1;
            cr[5] = s_cidx[tid];
        }

        // ---- flat segment stream: phase-1 votes, ordered compaction -------------------
        uint32_t vbase = 0;
        for (uint32_t e0 = 0; e0 < total_seg; e0 += kThreads) {
            const uint32_t e = e0 + tid;
            bool vote = false;
            bool first = false;
            uint32_t c = 0;
            float4 seg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < total_seg) {
                c = FindOwner(s_coff, ncand, e);
                const uint32_t k = e - s_coff[c];
                first = (k == 0);
                const uint32_t ctag = s_ctag[c];
                const uint32_t npt = s_cnpt[c];
                const uint8_t *pts = scene + s_cpts[c];
                if (ctag == kItemFill) {
                    if (k < npt) {  // npt == 0 leaves the placeholder element voteless
                        const uint32_t k1 = (k + 1 == npt) ? 0u : k + 1;
                        const float2 a = LoadF2(pts + static_cast<size_t>(k) * 8);
                        const float2 b = LoadF2(pts + static_cast<size_t>(k1) * 8);
                        seg = make_float4(a.x, a.y, b.x, b.y);
                        vote = VoteFill(seg, y0, sx0);
                    }
                } else if (ctag == kItemPoly) {
                    if (k + 1 < npt) {
                        const float2 a = LoadF2(pts + static_cast<size_t>(k) * 8);
                        const float2 b = LoadF2(pts + static_cast<size_t>(k + 1) * 8);
                        seg = make_float4(a.x, a.y, b.x, b.y);
                        const int y_test = sy0 + static_cast<int>(((k & 31u) >> 4) * kTileH);
                        vote = VotePoly(seg, s_chw[c], y_test, sx0, sy0);
                    }
                } else if (ctag == kItemLine) {
                    const float2 a = LoadF2(pts);
                    const float2 b = LoadF2(pts + 8);
                    seg = make_float4(a.x, a.y, b.x, b.y);
                    vote = true;  // lines have no strip-level cull (PietRender.metal:223-247)
                }
            }
            uint32_t nvote;
            const uint32_t pos = vbase + BlockRank(vote, s_part, &nvote);
            if (first) cand_rec[kCandDwords * c + 4] = pos;  // seg_off of candidate c
            if (vote) segs[pos] = seg;
            vbase += nvote;
        }
        if (tid == 0) P.arena[rec + 3] = vbase;  // number of segments that survived
        __syncthreads();  // s_c* arrays are rewritten by the next batch
    }

    // ---- queue the touched tiles, clear the untouched ones ------------------------------
    __syncthreads();
    const uint32_t hitmask = s_hitmask;
    const uint32_t tiles_here = min(kStripTiles, P.tiles_x - strip * kStripTiles);
    const uint32_t valid = (tiles_here >= 32u) ? 0xffffffffu : ((1u << tiles_here) - 1u);
    const uint32_t qmask = hitmask & valid;
    if (tid == 0 && qmask) s_qbase = atomicAdd(&P.ctr_cur->queue_count, static_cast<uint32_t>(__popc(qmask)));
    __syncthreads();
    if (tid < kStripTiles && ((qmask >> tid) & 1u)) {
        const uint32_t rank = __popc(qmask & ((1u << tid) - 1u));
        P.queue[s_qbase + rank] = row_rel * P.tiles_x + strip * kStripTiles + tid;
    }
    const uint32_t clear = ~hitmask & valid;
    if (clear) {
        // 16 pixel rows x 1024 B: thread -> (row = it*4 + tid/64, 16 B = 4 px at lane*4)
        const uint32_t lane16 = tid & 63u;
        const uint32_t t = lane16 >> 2;  // tile of these 4 pixels
        if ((clear >> t) & 1u) {
            const uint32_t px = static_cast<uint32_t>(sx0) + lane16 * 4u;
#pragma unroll
            for (uint32_t it = 0; it < 4; ++it) {
                const uint32_t r = it * 4u + (tid >> 6);
                const uint32_t py = static_cast<uint32_t>(y0) + r;
                if (py < P.height && px < P.width) {
                    uint8_t *dst = P.fb + static_cast<size_t>(row_rel * kTileH + r) * P.fb_stride + static_cast<size_t>(px) * 4;
                    if (px + 4 <= P.width && P.fb_vec16) {
                        *reinterpret_cast<uint4 *>(dst) = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
                    } else {
                        for (uint32_t k = 0; k < 4 && px + k < P.width; ++k)
                            reinterpret_cast<uint32_t *>(dst)[k] = 0xffffffffu;
                    }
                }
            }
        }
    }
}

// =====================================================================================
// K2: per-tile command build (LDS) + per-pixel interpreter
// =====================================================================================

namespace {

struct PixelState {
    _Float16 r, g, b;  // half3 rgb (PietRender.metal:470)
    float df;          // :471
    _Float16 sa;       // half signedArea (:472)
};

__device__ __forceinline__ _Float16 HMix(_Float16 x, _Float16 y, _Float16 a) { return x + (y - x) * a; }

__device__ __forceinline__ _Float16 HalfFromBits(uint32_t b) {
    const uint16_t u = static_cast<uint16_t>(b);
    return __builtin_bit_cast(_Float16, u);
}

__device__ __forceinline__ void Blend(PixelState &st, uint32_t rg, uint32_t ba, _Float16 alpha) {
    const _Float16 fa = HalfFromBits(ba >> 16) * alpha;  // fg.a * alpha
    st.r = HMix(st.r, HalfFromBits(rg), fa);
    st.g = HMix(st.g, HalfFromBits(rg >> 16), fa);
    st.b = HMix(st.b, HalfFromBits(ba), fa);
}

// renderKernel's command loop (PietRender.metal:474-560) over an LDS-resident list.
__device__ __forceinline__ void Interpret(const Cmd *cmds, uint32_t n, float px, float py, PixelState &st) {
    for (uint32_t i = 0; i < n; ++i) {
        const Cmd cmd = cmds[i];
        switch (cmd.tag) {
            case kCmdCircle: {  // :481-494
                const float x0 = static_cast<float>(cmd.body[1] & 0xffffu), y0 = static_cast<float>(cmd.body[1] >> 16);
                const float x1 = static_cast<float>(cmd.body[2] & 0xffffu), y1 = static_cast<float>(cmd.body[2] >> 16);
                const float cx = x0 + (x1 - x0) * 0.5f, cy = y0 + (y1 - y0) * 0.5f;
                const float dx = px - cx, dy = py - cy;
                const float r = sqrtf(dx * dx + dy * dy);
                const float circle_r = fminf(cx - x0, cy - y0);
                const _Float16 alpha = static_cast<_Float16>(Sat(circle_r - r));
                const _Float16 zero = static_cast<_Float16>(0.0f);
                st.r = HMix(st.r, zero, alpha);
                st.g = HMix(st.g, zero, alpha);
                st.b = HMix(st.b, zero, alpha);
                break;
            }
            case kCmdLine: {  // stroke(), :49-55
                const float sx = __uint_as_float(cmd.body[1]), sy = __uint_as_float(cmd.body[2]);
                const float ex = __uint_as_float(cmd.body[3]), ey = __uint_as_float(cmd.body[4]);
                const float lx = ex - sx, ly = ey - sy;
                const float dx = px - sx, dy = py - sy;
                const float t = Sat((lx * dx + ly * dy) / (lx * lx + ly * ly));
                const float fx = lx * t - dx, fy = ly * t - dy;
                st.df = fminf(st.df, sqrtf(fx * fx + fy * fy));
                break;
            }
            case kCmdStroke: {  // :500-507, renderDf :58-60
                const float half_width = __uint_as_float(cmd.body[0]);
                const _Float16 alpha = static_cast<_Float16>(Sat(half_width + 0.5f - st.df));
                Blend(st, cmd.body[2], cmd.body[3], alpha);
                st.df = 1e9f;
                break;
            }
            case kCmdFill: {  // :508-529
                const float sx = __uint_as_float(cmd.body[1]) - px, sy = __uint_as_float(cmd.body[2]) - py;
                const float ex = __uint_as_float(cmd.body[3]) - px, ey = __uint_as_float(cmd.body[4]) - py;
                const float wx = Sat(sy), wy = Sat(ey);
                if (wx != wy) {
                    const float tx = (wx - sy) / (ey - sy);
                    const float ty = (wy - sy) / (ey - sy);
                    const float xsx = sx + (ex - sx) * tx;
                    const float xsy = sx + (ex - sx) * ty;
                    const float xmin = fminf(fminf(xsx, xsy), 1.0f) - 1e-6f;
                    const float xmax = fmaxf(xsx, xsy);
                    const float b = fminf(xmax, 1.0f);
                    const float c = fmaxf(b, 0.0f);
                    const float d = fmaxf(xmin, 0.0f);
                    const float area = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
                    st.sa = st.sa + static_cast<_Float16>(area * (wx - wy));
                }
                break;
            }
            case kCmdFillEdge: {  // :530-534 (half + float => f32 add, one rounding)
                const float sgn = static_cast<float>(static_cast<int>(cmd.body[0]));
                const float v = sgn * Sat(py - __uint_as_float(cmd.body[1]) + 1.0f);
                st.sa = static_cast<_Float16>(static_cast<float>(st.sa) + v);
                break;
            }
            case kCmdDrawFill: {  // :535-545
                _Float16 alpha = st.sa + static_cast<_Float16>(static_cast<float>(static_cast<int>(cmd.body[0])));
                alpha = static_cast<_Float16>(fminf(fabsf(static_cast<float>(alpha)), 1.0f));
                Blend(st, cmd.body[2], cmd.body[3], alpha);
                st.sa = static_cast<_Float16>(0.0f);
                break;
            }
            case kCmdSolid: {  // :546-551
                Blend(st, cmd.body[1], cmd.body[2], static_cast<_Float16>(1.0f));
                break;
            }
            default:
                break;
        }
    }
}

struct Emit {
    uint32_t n;      // commands of this stream element (0..2)
    Cmd c0, c1;
};

}  // namespace

template <bool kCapture>
__global__ __launch_bounds__(kThreads) void pm_tile_kernel(FrameParams P) {
    __shared__ Cmd s_cmds[kMaxPending];
    __shared__ uint32_t s_part[kWaves * 4];
    __shared__ uint32_t s_htag[kThreads];   // hit candidates of this tile
    __shared__ uint32_t s_hrgba[kThreads];
    __shared__ uint32_t s_haux0[kThreads];
    __shared__ uint32_t s_haux1[kThreads];
    __shared__ uint32_t s_hseg[kThreads];   // first surviving segment of the candidate
    __shared__ uint32_t s_hcnt[kThreads];   // stream elements of the candidate
    __shared__ uint32_t s_hoff[kThreads + 1];
    __shared__ int s_backdrop[kThreads];
    __shared__ uint32_t s_any[kThreads];
    __shared__ uint32_t s_solid_rgba;

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6;
    const uint32_t qn = P.ctr_cur->queue_count;

    for (uint32_t q = blockIdx.x; q < qn; q += gridDim.x) {
        const uint32_t tile = P.queue[q];
        const uint32_t tx = tile % P.tiles_x;
        const uint32_t ty_rel = tile / P.tiles_x;
        const uint32_t ty = P.row0 + ty_rel;
        const int x0 = static_cast<int>(tx * kTileW);
        const int y0 = static_cast<int>(ty * kTileH);
        const float fx0 = static_cast<float>(x0), fy0 = static_cast<float>(y0);
        const float fx1 = static_cast<float>(x0 + static_cast<int>(kTileW));
        const float fy1 = static_cast<float>(y0 + static_cast<int>(kTileH));
        const uint32_t tbit = tx & (kStripTiles - 1);
        const uint32_t sr = ty_rel * P.strips_x + tx / kStripTiles;

        const uint32_t pxi = static_cast<uint32_t>(x0) + (tid & 15u);
        const uint32_t pyi = static_cast<uint32_t>(y0) + (tid >> 4);
        const float px = static_cast<float>(pxi), py = static_cast<float>(pyi);
        PixelState st;
        st.r = st.g = st.b = static_cast<_Float16>(1.0f);
        st.df = 1e9f;
        st.sa = static_cast<_Float16>(0.0f);

        uint32_t solid_color = 0xffffffffu;  // TileEncoder::solidColor (:74)
        uint32_t n_pending = 0;              // commands waiting in s_cmds
        uint32_t list_len = 0;               // logical list length since tileBegin (capture)

        uint32_t rec = P.striprow_head[sr];
        while (rec != 0) {
            const uint32_t next = P.arena[rec + 0];
            const uint32_t ncand = P.arena[rec + 1];
            const uint32_t nsurv = P.arena[rec + 3];
            const uint32_t *cand_rec = P.arena + rec + kRecHdrDwords;
            const float4 *segs = reinterpret_cast<const float4 *>(P.arena + rec + kRecHdrDwords + kCandDwords * ncand);

            // ---- candidates that hit this tile, in paint order -------------------------
            bool hit = false;
            uint32_t w0 = 0, rgba = 0, aux0 = 0, aux1 = 0, seg_off = 0, seg_end = 0;
            if (tid < ncand) {
                const uint4 a = *reinterpret_cast<const uint4 *>(cand_rec + kCandDwords * tid);
                w0 = a.x; rgba = a.y; aux0 = a.z; aux1 = a.w;
                seg_off = cand_rec[kCandDwords * tid + 4];
                seg_end = (tid + 1 < ncand) ? cand_rec[kCandDwords * (tid + 1) + 4] : nsurv;
                hit = ((w0 >> (16 + tbit)) & 1u) != 0;
                const uint32_t tg = w0 & 0xffffu;
                // circles own one pseudo element; the others as many as survived phase 1
                if (hit && tg != kItemCircle && seg_end == seg_off) hit = false;
                if (tg == 0) hit = false;
            }
            uint32_t nh;
            const uint32_t hpos = BlockRank(hit, s_part, &nh);
            uint32_t cnt = 0;
            if (hit) {
                cnt = ((w0 & 0xffffu) == kItemCircle) ? 1u : (seg_end - seg_off);
                s_htag[hpos] = w0 & 0xffffu;
                s_hrgba[hpos] = rgba;
                s_haux0[hpos] = aux0;
                s_haux1[hpos] = aux1;
                s_hseg[hpos] = seg_off;
                s_hcnt[hpos] = cnt;
                s_backdrop[hpos] = 0;
                s_any[hpos] = 0;
            }
            __syncthreads();
            // stream offsets over the hit candidates
            uint32_t stream_len;
            {
                const uint32_t v = (tid < nh) ? s_hcnt[tid] : 0u;
                const uint32_t o = BlockExclusiveScan(v, s_part, &stream_len);
                if (tid < nh) s_hoff[tid] = o;
                if (tid == 0) s_hoff[nh] = stream_len;
            }
            __syncthreads();

            // ---- stream rounds: phase-2 tests -> ordered commands ------------------------
            for (uint32_t e0 = 0; e0 < stream_len; e0 += kThreads) {
                const uint32_t e = e0 + tid;
                Emit em;
                em.n = 0;
                bool is_last = false;
                bool draws = false;  // any of this lane's commands clears solidColor
                uint32_t c = 0, ctag = 0;
                if (e < stream_len) {
                    c = FindOwner(s_hoff, nh, e);
                    const uint32_t k = e - s_hoff[c];
                    is_last = (k + 1 == s_hcnt[c]);
                    ctag = s_htag[c];
                    if (ctag == kItemCircle) {  // :218-222
                        em.n = 1;
                        em.c0.tag = kCmdCircle;
                        em.c0.body[0] = 0;
                        em.c0.body[1] = s_haux0[c];
                        em.c0.body[2] = s_haux1[c];
                        em.c0.body[3] = 0;
                        em.c0.body[4] = 0;
                        draws = true;
                    } else {
                        const float4 s = segs[s_hseg[c] + k];
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
                            const float s_top_left = Sgn(left + fy0 * b + cc);
                            const float s00 = Sgn(top + left + cc);
                            const float s01 = Sgn(top + right + cc);
                            const float s10 = Sgn(bot + left + cc);
                            const float s11 = Sgn(bot + right + cc);
                            if (s_top_left == Sgn(a) && ymin <= fy0) {
                                const int d = -static_cast<int>(s00);  // backdrop -= s00
                                if (d != 0) atomicAdd(&s_backdrop[c], d);
                            }
                            const bool straddle = Straddles(s00, s01, s10, s11);
                            if (xmin < fx0 && xmax > fx0) {
                                const float tt = (s.x - fx0) / b;
                                const float y_edge = s.y + (s.w - s.y) * tt;  // mix(start.y, end.y, tt)
                                if (y_edge >= fy0 && y_edge < fy1) {
                                    em.n = 2;
                                    em.c0.tag = kCmdFillEdge;
                                    em.c0.body[0] = static_cast<uint32_t>(static_cast<int>(s00));
                                    em.c0.body[1] = __float_as_uint(y_edge);
                                    em.c0.body[2] = 0; em.c0.body[3] = 0; em.c0.body[4] = 0;
                                    em.c1.tag = kCmdFill;
                                    em.c1.body[0] = 0;
                                    if (b > 0.0f) {
                                        em.c1.body[1] = __float_as_uint(s.x); em.c1.body[2] = __float_as_uint(s.y);
                                        em.c1.body[3] = __float_as_uint(fx0); em.c1.body[4] = __float_as_uint(y_edge);
                                    } else {
                                        em.c1.body[1] = __float_as_uint(fx0); em.c1.body[2] = __float_as_uint(y_edge);
                                        em.c1.body[3] = __float_as_uint(s.z); em.c1.body[4] = __float_as_uint(s.w);
                                    }
                                } else if (straddle) {
                                    em.n = 1;
                                }
                            } else if (straddle && xmin < fx1 && xmax > fx0) {
                                em.n = 1;
                            }
                            if (em.n == 1) {
                                em.c0.tag = kCmdFill;
                                em.c0.body[0] = 0;
                                em.c0.body[1] = __float_as_uint(s.x); em.c0.body[2] = __float_as_uint(s.y);
                                em.c0.body[3] = __float_as_uint(s.z); em.c0.body[4] = __float_as_uint(s.w);
                            }
                            if (em.n) atomicOr(&s_any[c], 1u);
                        } else {
                            // Line (:223-247) and Poly phase 2 (:406-440) share the inflated-box test
                            const float width = __uint_as_float(s_haux0[c]);
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
                                em.n = 1;
                                em.c0.tag = kCmdLine;
                                em.c0.body[0] = 0;
                                em.c0.body[1] = __float_as_uint(s.x); em.c0.body[2] = __float_as_uint(s.y);
                                em.c0.body[3] = __float_as_uint(s.z); em.c0.body[4] = __float_as_uint(s.w);
                                draws = true;
                                atomicOr(&s_any[c], 1u);
                            }
                        }
                    }
                }
                __syncthreads();  // per-candidate accumulators complete for elements <= this round

                // ---- per-item closing command (DrawFill / Solid / Stroke) -----------------------
                bool has_fin = false;
                bool opaque_solid = false;
                Cmd fin;
                fin.tag = 0;
                if (is_last) {
                    const uint32_t rgba = s_hrgba[c];
                    const uint32_t rg = P.lut_srgb2lin[rgba & 0xffu] | (P.lut_srgb2lin[(rgba >> 8) & 0xffu] << 16);
                    const uint32_t ba = P.lut_srgb2lin[(rgba >> 16) & 0xffu] | (P.lut_unorm2h[rgba >> 24] << 16);
                    if (ctag == kItemFill) {  // :359-363
                        const int backdrop = s_backdrop[c];
                        if (s_any[c]) {
                            has_fin = true;
                            fin.tag = kCmdDrawFill;
                            fin.body[0] = static_cast<uint32_t>(backdrop);
                            fin.body[1] = rgba; fin.body[2] = rg; fin.body[3] = ba; fin.body[4] = 0;
                            draws = true;
                        } else if (backdrop != 0) {
                            has_fin = true;
                            fin.tag = kCmdSolid;
                            fin.body[0] = rgba; fin.body[1] = rg; fin.body[2] = ba; fin.body[3] = 0; fin.body[4] = 0;
                            opaque_solid = (rgba & 0xff000000u) == 0xff000000u;  // :132
                        }
                    } else if (ctag == kItemPoly || ctag == kItemLine) {  // :441-443, :243
                        if (s_any[c]) {
                            has_fin = true;
                            fin.tag = kCmdStroke;
                            fin.body[0] = __float_as_uint(0.5f * __uint_as_float(s_haux0[c]));
                            fin.body[1] = rgba; fin.body[2] = rg; fin.body[3] = ba; fin.body[4] = 0;
                            draws = true;
                        }
                    }
                }
                const uint32_t lane_total = em.n + (has_fin ? 1u : 0u);

                // ---- block-wide slots ------------------------------------------------------------
                const uint32_t incl = WaveInclusiveScan(lane_total);
                const uint32_t local = incl - lane_total;
                // position (within the wave) of the last opaque Solid / last solid-clearing command
                int my_solid = opaque_solid ? static_cast<int>(local + em.n) : -1;
                int my_draw = draws ? static_cast<int>(local + lane_total - 1) : -1;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    my_solid = max(my_solid, __shfl_xor(my_solid, d, 64));
                    my_draw = max(my_draw, __shfl_xor(my_draw, d, 64));
                }
                if (LaneId() == 63) {
                    s_part[wave * 4 + 0] = incl;
                    s_part[wave * 4 + 1] = static_cast<uint32_t>(my_solid);
                    s_part[wave * 4 + 2] = static_cast<uint32_t>(my_draw);
                }
                __syncthreads();
                uint32_t wbase = 0, round_total = 0;
                int last_solid = -1, last_draw = -1;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const uint32_t tot = s_part[w * 4 + 0];
                    const int so = static_cast<int>(s_part[w * 4 + 1]);
                    const int dr = static_cast<int>(s_part[w * 4 + 2]);
                    if (so >= 0) last_solid = static_cast<int>(round_total) + so;
                    if (dr >= 0) last_draw = static_cast<int>(round_total) + dr;
                    if (w < static_cast<int>(wave)) wbase += tot;
                    round_total += tot;
                }
                const uint32_t pos = wbase + local;  // slot of this lane's first command in the round

                uint32_t base;       // s_cmds slot of round position 0 (may be "negative")
                uint32_t first_kept; // round positions below this are dropped
                if (last_solid >= 0) {
                    // TileEncoder::encodeSolid with an opaque colour (:132-135): the list restarts
                    // at tileBegin, so everything before it -- including pixels already blended by
                    // an earlier flush -- is forgotten.
                    first_kept = static_cast<uint32_t>(last_solid);
                    n_pending = 0;
                    list_len = 0;
                    st.r = st.g = st.b = static_cast<_Float16>(1.0f);
                    base = 0u - first_kept;
                } else {
                    first_kept = 0;
                    if (n_pending + round_total > kMaxPending) {
                        Interpret(s_cmds, n_pending, px, py, st);
                        n_pending = 0;
                        __syncthreads();  // every pixel done with s_cmds before it is overwritten
                    }
                    base = n_pending;
                }
                {
                    uint32_t p = pos;
                    if (em.n >= 1) {
                        if (p >= first_kept) {
                            s_cmds[base + p] = em.c0;
                            if (kCapture) {
                                const uint32_t li = list_len + p - first_kept;
                                if (li < P.dbg_max) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + li] = em.c0;
                            }
                        }
                        ++p;
                    }
                    if (em.n == 2) {
                        if (p >= first_kept) {
                            s_cmds[base + p] = em.c1;
                            if (kCapture) {
                                const uint32_t li = list_len + p - first_kept;
                                if (li < P.dbg_max) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + li] = em.c1;
                            }
                        }
                        ++p;
                    }
                    if (has_fin) {
                        if (p >= first_kept) {
                            s_cmds[base + p] = fin;
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
                        if (opaque_solid && static_cast<int>(p) == last_solid) s_solid_rgba = fin.body[0];
                    }
                }
                n_pending = base + round_total;
                list_len += round_total - first_kept;
                __syncthreads();
                if (last_solid >= 0) solid_color = s_solid_rgba;
                if (last_draw > last_solid) solid_color = 0;  // encodeCircle/Line/Stroke/DrawFill (:81,:90,:99,:124)
            }
            rec = next;
            __syncthreads();  // s_h* arrays are rebuilt for the next record
        }

        // ---- TileEncoder::end() (:144-151) + composite (:34-44) ------------------------------
        uint32_t out;
        if (solid_color != 0) {
            out = solid_color;  // Bail: the tile is one opaque colour, bytes as stored
        } else {
            Interpret(s_cmds, n_pending, px, py, st);
            // linear -> sRGB + unorm8 (:563-565) through the pinned table
            const uint32_t r8 = P.lut_lin2srgb[__builtin_bit_cast(uint16_t, st.r)];
            const uint32_t g8 = P.lut_lin2srgb[__builtin_bit_cast(uint16_t, st.g)];
            const uint32_t b8 = P.lut_lin2srgb[__builtin_bit_cast(uint16_t, st.b)];
            out = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
        }
        if (pxi < P.width && pyi < P.height) {
            uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + (tid >> 4)) * P.fb_stride + static_cast<size_t>(pxi) * 4;
            *reinterpret_cast<uint32_t *>(dst) = out;
        }
        if (kCapture && tid == 0) {
            // list as the reference leaves it: {Bail} or cmds + End
            P.dbg_solid[tile] = solid_color;
            if (solid_color != 0) {
                P.dbg_counts[tile] = 1;
                if (P.dbg_max > 0) {
                    Cmd bail;
                    bail.tag = kCmdBail;
                    bail.body[0] = bail.body[1] = bail.body[2] = bail.body[3] = bail.body[4] = 0;
                    P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max] = bail;
                }
            } else {
                P.dbg_counts[tile] = list_len + 1;
                if (list_len < P.dbg_max) {
                    Cmd end;
                    end.tag = kCmdEnd;
                    end.body[0] = end.body[1] = end.body[2] = end.body[3] = end.body[4] = 0;
                    P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + list_len] = end;
                }
            }
        }
        __syncthreads();  // s_cmds / s_part reuse by the next tile
    }
}

// ---- launch wrappers (called from pm_context.hip) -----------------------------------------

void LaunchBin(const FrameParams &p, uint32_t n_striprows, hipStream_t stream) {
    hipLaunchKernelGGL(pm_bin_kernel, dim3(n_striprows), dim3(kThreads), 0, stream, p);
}

void LaunchTiles(const FrameParams &p, uint32_t grid, bool capture, hipStream_t stream) {
    if (capture)
        hipLaunchKernelGGL(pm_tile_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, p);
    else
        hipLaunchKernelGGL(pm_tile_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, p);
}

}  // namespace pm
