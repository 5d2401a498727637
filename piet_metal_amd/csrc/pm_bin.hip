// pm_index_kernel, pm_rowcull_kernel, pm_bin_kernel: scene index and the strip-level half of tileKernel
// (see pm_kernels_common.h for the decomposition and the rules shared by the three files)
#include "pm_bin_rows.h"

namespace pm {

// =====================================================================================
// K0: scene index, once per scene
// =====================================================================================

__global__ __launch_bounds__(256) void pm_index_kernel(const uint8_t *scene, uint32_t n_items, uint32_t items_ix, const uint32_t *chunk_base,
                                                       uint32_t n_chunks, float4 *chunk_bbox, float4 *sup_bbox) {
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    // an empty box no strip row can pass (also what a chunk of nothing but separators keeps)
    float xmin = 3.0e38f, ymin = 3.0e38f, xmax = -3.0e38f, ymax = -3.0e38f;
    if (ch < n_chunks) {
        const uint32_t item = FindOwner(chunk_base, n_items, ch);
        const uint8_t *it = scene + items_ix + static_cast<size_t>(item) * kItemSize;
        const uint32_t tag = LoadU32(it) & 0xffffu;
        // (only Fill and StrokePolyLine items have a point array: a Line's words 3 and 4 are its width and start.x, and a chunk
        //  FindOwner lands on such an item -- it owns none -- keeps the empty box)
        const bool has_points = tag == kItemFill || tag == kItemPoly;
        const uint32_t npt = has_points ? LoadU32(it + 12) : 0u;
        const uint8_t *pts = scene + (has_points ? LoadU32(it + 16) : 0u);
        const uint32_t nseg = (tag == kItemFill) ? FillSegs(npt) : PolySegs(npt);
        const uint32_t k0 = (ch - chunk_base[item]) * kChunkSegs;
        const uint32_t k1 = min(k0 + kChunkSegs, nseg);
        if (tag == kItemFill && (LoadU32(it + 4) & kFillCompound)) {
            // compound fill: the box of the segments that exist
            for (uint32_t k = k0; k < k1; ++k) {
                float2 a, b;
                if (!FillSegmentEnds(pts, npt, true, k, a, b)) continue;
                xmin = fminf(xmin, fminf(a.x, b.x)); ymin = fminf(ymin, fminf(a.y, b.y));
                xmax = fmaxf(xmax, fmaxf(a.x, b.x)); ymax = fmaxf(ymax, fmaxf(a.y, b.y));
            }
        } else if (has_points) {
            // points k0 .. k1 (the fill's closing segment wraps to point 0)
            for (uint32_t k = k0; k <= k1; ++k) {
                const uint32_t pi = (tag == kItemFill && k == npt) ? 0u : k;
                const float2 p = LoadF2(pts + static_cast<size_t>(pi) * 8);
                xmin = fminf(xmin, p.x); ymin = fminf(ymin, p.y);
                xmax = fmaxf(xmax, p.x); ymax = fmaxf(ymax, p.y);
            }
        }
        chunk_bbox[ch] = make_float4(xmin, ymin, xmax, ymax);
    }
    // the super-chunk's box: union over the kSuperChunks consecutive lanes that hold its chunks (of whatever items)
    static_assert(kSuperChunks == 8, "three butterfly steps");
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, d)); ymin = fminf(ymin, __shfl_xor(ymin, d));
        xmax = fmaxf(xmax, __shfl_xor(xmax, d)); ymax = fmaxf(ymax, __shfl_xor(ymax, d));
    }
    if ((ch & (kSuperChunks - 1u)) == 0u && ch < n_chunks) sup_bbox[ch / kSuperChunks] = make_float4(xmin, ymin, xmax, ymax);
}

// =====================================================================================
// K1a: per-tile-row item lists (large scenes only), row_parts workgroups per tile row
// =====================================================================================
//
// With thousands of items every strip-row workgroup of pm_bin_kernel would scan every bbox of
// the band (PietRender.metal:191-208 does exactly that per threadgroup).  The row part of that
// test does not depend on the strip, so for large scenes it is done once per tile row here and
// the strip rows of the row scan the (much shorter) row list instead.  Lists keep paint order;
// their sizes are known to the host from the same predicate -- per tile row AND per part of the item range (round 5: a row's
// scan is cut into row_parts workgroups, each with its own, host-computed place in the row's list; one workgroup walking
// 20 000 items was ten dependent steps, 24 us in front of a 100 us frame) -- so part p of row r writes exactly
// row_base[r * parts + p + 1] - row_base[r * parts + p] entries.
__global__ __launch_bounds__(kBinThreads) void pm_rowcull_kernel(FrameParams P) {
    __shared__ uint32_t s_part[kBinWaves];
    const uint32_t tid = threadIdx.x;
    const uint32_t row_rel = blockIdx.x / P.row_parts, part = blockIdx.x % P.row_parts;
    const int y0 = static_cast<int>((P.row0 + row_rel) * kTileH);
    uint32_t out = P.row_base[blockIdx.x];
    // A lane tests kPer consecutive items per step (paint order = lane order = item order), one block scan per 2 048
    // items: with one item per lane the 10 000 items of config 4 were 40 dependent steps, 30 us in front of every frame.
    constexpr uint32_t kPer = kRowCullPer;
    static_assert(kBinThreads * kPer == kRowCullStep, "what the host sizes a part by");
    const uint32_t j_end = min(P.n_band_items, (part + 1u) * P.row_part_items);
    for (uint32_t jb = part * P.row_part_items; jb < j_end; jb += kBinThreads * kPer) {
        const uint32_t j0 = jb + tid * kPer;
        uint2 bb[kPer];
        uint32_t it[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            bb[u] = make_uint2(0u, 0u);
            it[u] = 0;
            if (j0 + u < j_end) {
                bb[u] = P.band_bbox[j0 + u];
                it[u] = P.band_item[j0 + u];
            }
        }
        uint32_t hits = 0;
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const int by = static_cast<int>(bb[u].x >> 16), bw = static_cast<int>(bb[u].y >> 16);
            if (j0 + u < j_end && bw >= y0 && by < y0 + static_cast<int>(kTileH)) hits |= 1u << u;  // row part of :198 / :214
        }
        uint32_t total;
        uint32_t pos = out + BlockExclusiveScan<kBinWaves>(static_cast<uint32_t>(__popc(hits)), s_part, &total);
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u)
            if ((hits >> u) & 1u) {
                P.row_bbox[pos] = bb[u];
                P.row_item[pos] = it[u];
                ++pos;
            }
        out += total;
    }
}

// =====================================================================================
// K1: binning, one workgroup (kW = 4) or one wave (kW = 1) per strip row: BinStripRows, pm_bin_rows.h
// =====================================================================================
// (the leading arguments: BinEntry, pm_bin_rows.h -- copies of P's fields of these names that reach the kernel in SGPRs)
template <bool kProfile, int kW>
__global__ __launch_bounds__(64 * kW, 5) void pm_bin_kernel(const uint4 *e_sr_desc, const uint2 *e_band_bbox, const uint32_t *e_band_item,
                                                             const uint32_t *e_lut_srgb2lin, const uint32_t *e_lut_unorm2h, uint32_t e_n_band_items,
                                                             uint32_t e_use_row_lists, uint32_t e_bin_grid, FrameParams P) {
    // (blocks beyond the strip rows' grid: the pixels of the strip rows no item reaches -- FrameParams::clear_in_bin)
    if (blockIdx.x >= e_bin_grid) {
        ClearStripRow<kW>(P, P.idle_sr[blockIdx.x - e_bin_grid]);
        return;
    }
    __shared__ BinLds<kW> L;
    const BinEntry E{e_sr_desc, e_band_bbox, e_band_item, e_lut_srgb2lin, e_lut_unorm2h, e_n_band_items, e_use_row_lists};
    BinStripRows<kProfile, kW, false>(P, L, nullptr, &E);
}

// =====================================================================================
// ---- launch wrappers (called from pm_context.hip) -----------------------------------------

void LaunchIndex(const uint8_t *scene, uint32_t n_items, uint32_t items_ix, const uint32_t *chunk_base, uint32_t n_chunks,
                 float4 *chunk_bbox, float4 *sup_bbox, hipStream_t stream) {
    if (n_chunks == 0) return;
    hipLaunchKernelGGL(pm_index_kernel, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, scene, n_items, items_ix, chunk_base,
                       n_chunks, chunk_bbox, sup_bbox);
}

void LaunchBin(const FrameParams &p, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    const uint32_t n_striprows = p.bin_grid + (p.clear_in_bin ? p.n_idle_sr : 0u);
    if (p.use_row_lists) hipLaunchKernelGGL(pm_rowcull_kernel, dim3((p.row1 - p.row0) * p.row_parts), dim3(kBinThreads), 0, stream, p);
#define PM_BIN_ARGS p.sr_desc, p.band_bbox, p.band_item, p.lut_srgb2lin, p.lut_unorm2h, p.n_band_items, p.use_row_lists, p.bin_grid, p
    if (p.bin_waves == 1) {
        if (p.dbg_bin)
            PM_LAUNCH((pm_bin_kernel<true, 1>), dim3(n_striprows), dim3(64), stream, t0, t1, PM_BIN_ARGS);
        else
            PM_LAUNCH((pm_bin_kernel<false, 1>), dim3(n_striprows), dim3(64), stream, t0, t1, PM_BIN_ARGS);
    } else if (p.dbg_bin)
        PM_LAUNCH((pm_bin_kernel<true, 4>), dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, PM_BIN_ARGS);
    else
        PM_LAUNCH((pm_bin_kernel<false, 4>), dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, PM_BIN_ARGS);
#undef PM_BIN_ARGS
}

}  // namespace pm
