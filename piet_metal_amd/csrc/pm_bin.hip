// pm_index_kernel, pm_rowcull_kernel, pm_bin_kernel: scene index and the strip-level half of tileKernel
// (see pm_kernels_common.h for the decomposition and the rules shared by the three files)
#include "pm_kernels_common.h"
#include <pm_params.h>  // gfx950/pm_params.h: kernel arguments held in a VGPR, read with v_readlane

namespace pm {

// =====================================================================================
// K0: scene index, once per scene
// =====================================================================================

__global__ void pm_index_kernel(const uint8_t *scene, uint32_t n_items, uint32_t items_ix, const uint32_t *chunk_base,
                                uint32_t n_chunks, float4 *chunk_bbox) {
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_chunks) return;
    const uint32_t item = FindOwner(chunk_base, n_items, ch);
    const uint8_t *it = scene + items_ix + static_cast<size_t>(item) * kItemSize;
    const uint32_t tag = LoadU32(it) & 0xffffu;
    const uint32_t npt = LoadU32(it + 12);
    const uint8_t *pts = scene + LoadU32(it + 16);
    const uint32_t nseg = (tag == kItemFill) ? FillSegs(npt) : PolySegs(npt);
    const uint32_t k0 = (ch - chunk_base[item]) * kChunkSegs;
    const uint32_t k1 = min(k0 + kChunkSegs, nseg);
    float xmin = 0.f, ymin = 0.f, xmax = 0.f, ymax = 0.f;
    if (tag == kItemFill && (LoadU32(it + 4) & kFillCompound)) {
        // compound fill: the box of the segments that exist (a chunk of separators only keeps an
        // empty box no strip row can pass)
        xmin = ymin = 3.0e38f;
        xmax = ymax = -3.0e38f;
        for (uint32_t k = k0; k < k1; ++k) {
            float2 a, b;
            if (!FillSegmentEnds(pts, npt, true, k, a, b)) continue;
            xmin = fminf(xmin, fminf(a.x, b.x)); ymin = fminf(ymin, fminf(a.y, b.y));
            xmax = fmaxf(xmax, fmaxf(a.x, b.x)); ymax = fmaxf(ymax, fmaxf(a.y, b.y));
        }
        chunk_bbox[ch] = make_float4(xmin, ymin, xmax, ymax);
        return;
    }
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

    if (blockIdx.x == 0 && tid < kTicketParts) PM_PP(ctr_next)->ticket[tid].count = 0;
    if (blockIdx.x == 0 && tid == 0) {
        // The counters of the NEXT frame (the other parity) are idle now: reset them
        // here so that no separate memset launch is needed.
        PM_PP(ctr_next)->arena_top = 0;
        PM_PP(ctr_next)->ptcl_top = 0;
#pragma unroll
        for (uint32_t k = 0; k < kClasses; ++k) PM_PP(ctr_next)->cls[k].count = 0;
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
            float hw = 0.0f;
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
                            float2 a, b;
                            if (FillSegmentEnds(pts, s_cnpt[vc], (s_caux0[vc] & kFillCompound) != 0, k, a, b)) {
                                seg = make_float4(a.x, a.y, b.x, b.y);
                                vote = VoteFill(seg, y0, sx0);
                            }
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
                const bool even_odd = tag == kItemFill && (s_caux0[tid] & kFillEvenOdd) != 0;
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
                        // a tile wholly inside a fill is covered if its winding is non-zero / odd
                        const bool pseudo = tag == kItemCircle || (tag == kItemFill && (even_odd ? (run & 1) != 0 : run != 0));
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
    // cost class of the tile's list (0 = longest): the number of thresholds the estimate does not exceed
    static_assert(kClasses == 8, "seven thresholds spelled out below");
#define PM_THR(k) ((est <= ParamU32<offsetof(FrameParams, class_thr) + 4 * (k)>(PR)) ? 1u : 0u)
    const uint32_t cls = PM_THR(0) + PM_THR(1) + PM_THR(2) + PM_THR(3) + PM_THR(4) + PM_THR(5) + PM_THR(6);
#undef PM_THR
    uint32_t my_mask = 0;    // queued tiles of this lane's class
    uint32_t lane_cnt = 0;   // lane c < kClasses: tiles of class c in this strip row
    uint32_t queued = 0;
#pragma unroll
    for (uint32_t k = 0; k < kClasses; ++k) {
        const uint32_t mk = static_cast<uint32_t>(__ballot(is_queued && cls == k));
        if (cls == k) my_mask = mk;
        if (lane == k) lane_cnt = static_cast<uint32_t>(__popc(mk));
        queued |= mk;
    }
    // command-list slots of a queued tile: an element emits at most 2 commands + its item's
    // closing command, plus End
    const uint32_t slots = is_queued ? 3u * est + 1u : 0u;
    const uint32_t slots_incl = WaveInclusiveScan(slots);
    const uint32_t qtotal = WaveLast(slots_incl);
    // list space and the class queue positions: ONE atomic instruction, a lane per counter
    uint32_t qres = 0;
    if (queued) {  // uniform
        if (lane < kClasses && lane_cnt) qres = atomicAdd(&PM_PP(ctr_cur)->cls[lane].count, lane_cnt);
        if (lane == kClasses) qres = atomicAdd(&PM_PP(ctr_cur)->ptcl_top, qtotal);
    }
    // tiles with nothing to draw are background: no item touches them, or every touching
    // item lost all its segments in phase 1 (the reference writes Bail/white for them).  Their
    // pixels are written by pm_clear_kernel from tile_state: 25 MB of stores per 4K frame that
    // would otherwise stall these latency-bound workgroups in bursts.
    const uint32_t tile = row_rel * PM_PU(tiles_x) + strip * kStripTiles + lane;
    if (tile_lane)  // what this kernel decided per tile: 0 = queued, else the tile's colour
        PM_PP(tile_state)[tile] = is_queued ? 0u : (is_solid ? s_solid_rgba[lane] : 0xffffffffu);
    if (queued) {
        const uint32_t base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(qres), kClasses));
        const uint32_t q_base = static_cast<uint32_t>(__shfl(static_cast<int>(qres), static_cast<int>(cls)));  // my class's queue position
        const bool fits = base + qtotal <= PM_PU(ptcl_cap) && base + qtotal >= base;
        // (on overflow the tiles are still queued but marked "no list": the tile kernels skip
        //  them, the frame has holes, and pm_sync re-renders it with a larger arena)
        if (!fits && lane == 0) PM_PP(ctr_cur)->overflow = 1;
        if (is_queued) {
            const uint32_t list_slot = fits ? base + (slots_incl - slots) : 0xffffffffu;
            PM_PP(tile_ptcl)[tile] = list_slot;
            // A queue entry is everything the tile kernels need to start: {tile, first command
            // slot, first binning record of the strip row, commands written (pm_coarse_kernel)}
            const uint4 entry = make_uint4(tile, list_slot, head, 0u);
            const uint32_t below = (1u << lane) - 1u;
            PM_PP(queue)[cls * PM_PU(queue_cap) + q_base + __popc(my_mask & below)] = entry;
        }
    }
    stamp(5);  // queues + list slots done
    (void)prof_chunks;
    stamp(7);
}

// =====================================================================================
// ---- launch wrappers (called from pm_context.hip) -----------------------------------------

void LaunchIndex(const uint8_t *scene, uint32_t n_items, uint32_t items_ix, const uint32_t *chunk_base, uint32_t n_chunks,
                 float4 *chunk_bbox, hipStream_t stream) {
    if (n_chunks == 0) return;
    hipLaunchKernelGGL(pm_index_kernel, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, scene, n_items, items_ix, chunk_base,
                       n_chunks, chunk_bbox);
}

void LaunchBin(const FrameParams &p, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    const uint32_t n_striprows = p.n_sr_active;
    if (p.use_row_lists) hipLaunchKernelGGL(pm_rowcull_kernel, dim3(p.row1 - p.row0), dim3(kBinThreads), 0, stream, p);
    if (p.dbg_bin)
        PM_LAUNCH(pm_bin_kernel<true>, dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, p);
    else
        PM_LAUNCH(pm_bin_kernel<false>, dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, p);
}

}  // namespace pm
