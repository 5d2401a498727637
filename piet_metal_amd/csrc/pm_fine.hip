// pm_fine_kernel (+ pm_clear_kernel, pm_coverage_kernel): the tile stage as launches of its own
// (see pm_kernels_common.h for the decomposition and the rules shared by the kernel files)
#include "pm_fine_tile.h"
#include <type_traits>

namespace pm {

__global__ __launch_bounds__(kBinThreads) void pm_clear_kernel(FrameParams P) { ClearStripRow(P, blockIdx.x); }

// K3: per-pixel interpreter (renderKernel :457-566) over the per-tile command lists, with row-sparse
// Fill evaluation (PrepareFills) and item-parallel rendering of the tiles with long lists
// (RenderChunkWG).  Persistent grid; the four waves of a workgroup take four consecutive slots of
// the same pass, and a tile with a long list owns four aligned slots, i.e. exactly one workgroup.
//
// kFused: the wave (or, for a long list, wave 0 of the workgroup) first builds the tile's list with
// CoarseTile -- pm_coarse_kernel's body -- and interprets it straight away: no launch boundary
// between the two stages; a single wave reads its list back while it is still in L2, a workgroup
// finds the first three chunks in LDS (CoarseTile drops them into the waiting waves' staging areas).
// Blocks beyond P.fine_grid write the pixels of the tiles binning resolved (ClearStripRow).
// kProf: the developer timeline build (pm_debug_time_tiles); P.dbg_time is only read there.
// kCapture (pm_debug_capture_ptcl): the fused kernel also records every list it builds in the
// reference's layout -- the lists of the frame path itself, LDS drop into the waiting waves included.
// kDense: every tile one wave's, whatever its list -- the instantiation the host launches for a frame the previous frame of the same
// scene found dense (below): without the workgroup paths the kernel fits 80 VGPRs and 19 KB of LDS, SIX workgroups per CU instead of five.
template <bool kFused, bool kProf, bool kCapture = false, bool kDense = false>
__global__ __launch_bounds__(kThreads, kDense ? 6 : 5) void pm_fine_kernel(FrameParams P) {
    __shared__ std::conditional_t<kDense, DenseLds, SparseLds> S;
    if (blockIdx.x >= P.fine_grid) {
        ClearStripRow(P, blockIdx.x - P.fine_grid);
        return;
    }
    const uint32_t lane = LaneId(), wave = WaveId();
    uint32_t cls_end[kClasses];  // running totals of the class queues (longest lists first)
    {
        uint32_t run = 0;
#pragma unroll
        for (uint32_t k = 0; k < kClasses; ++k) {
            run += P.ctr_cur->cls[k].count;
            cls_end[k] = run;
        }
    }
    // Which workgroup of the hand-out this block is: blocks are dispatched round robin over the 8
    // XCDs (block b runs on XCD b % 8, tools/probes/atomic_probe.hip), and neighbouring slots are
    // tiles of one strip row that read the same binning record -- so runs of four workgroups (16
    // slots) are dealt to the XCDs instead of single ones, and a record is fetched into one L2, not four.
    uint32_t wg = blockIdx.x;
    if ((P.fine_grid & 31u) == 0u) {
        const uint32_t xcd = wg & 7u, i = wg >> 3;
        wg = (((i >> 2) << 3) + xcd) * 4u + (i & 3u);
    }
    const uint32_t wave_global = wg * kWaves + wave;
    const uint32_t n_waves = P.fine_grid * kWaves;
    const uint32_t n_tiles = cls_end[kClasses - 1];
    const uint32_t n_heavy = P.n_heavy_classes ? cls_end[min(P.n_heavy_classes, kClasses) - 1u] : 0u;
    // A workgroup per long list is a latency measure: it pays while waves would otherwise idle.  Once the long lists alone, at a
    // workgroup each, would occupy every wave of the grid, splitting a tile only costs work: every tile gets one wave.  (The
    // rule used to be "more long lists than WAVES": held-out workload 2 -- 2 k blobs at 2048^2, 16 k tiles, half of them long --
    // ran its long lists on workgroups with three frames' worth of tiles waiting: sustained 164 -> 133 us per frame without.)
    const bool dense = kDense || static_cast<uint64_t>(n_heavy) * P.dense_factor >= n_waves || P.split_mode == 0;
    // (what the host picks the next frame's instantiation by: pm_context.hip, Enqueue.  A frame WITHOUT long lists is the one-wave
    //  kernel's too -- nothing in it would get a workgroup, and six workgroups per CU take its many short lists faster than five:
    //  held-out 3, 20 k glyph-like paths over every tile of a 4K frame, 114 -> 106 us alone, -6 % per frame in flight)
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.host_dense != nullptr)
        *P.host_dense = (static_cast<uint64_t>(n_heavy) * P.dense_factor >= P.verdict_waves || P.split_mode == 0 || n_heavy == 0u) ? 2u : 1u;
    const uint32_t sh = dense ? 0u : 2u;
    const uint32_t s_h = n_heavy << sh;  // slots of the tiles with long lists: a workgroup (4 slots) each
    const uint32_t n_slots = s_h + (n_tiles - n_heavy);
    // slot -> queue entry
    auto slot_entry = [&](uint32_t slot) -> uint32_t {
        const uint32_t t = slot < s_h ? (slot >> sh) : n_heavy + (slot - s_h);  // position in [longest ... shortest]
        uint32_t qix = t;
#pragma unroll
        for (uint32_t k = 1; k < kClasses; ++k)
            if (t >= cls_end[k - 1]) qix = k * P.queue_cap + (t - cls_end[k - 1]);
        return qix;
    };
    auto pass_slot = [&](uint32_t pass) -> uint32_t {
        return pass * n_waves + ((pass & 1u) ? (n_waves - 1u - wave_global) : wave_global);
    };
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    // Hand-out.  The passes that hold workgroup tiles (at least the first) are static, snake order
    // over the sorted slots as before.  The slots after them -- the shortest lists -- are dealt in
    // n_decks = min(kTicketParts, waves) interleaved decks (slot n_static + p + n_decks * j is card j of deck p); a
    // wave that has finished its static share draws from deck wave_global % kTicketParts until the
    // deck is empty.  Per-tile times vary by 2x around what the estimates predict, so a wave's
    // second tile lands on whoever is free instead of on whoever the snake says (one counter for
    // the whole grid would cap at ~90 draws/us; 32 counters in 32 cache lines cost nothing
    // measurable, tools/probes/atomic_probe.hip).
    // P.handout_static: the host saw other frames in flight -- with neighbours filling the idle
    // SIMDs a drawn hand-out only adds contention (sustained throughput -3.5 %), alone it ends the
    // launch 4.6 us earlier.
    // (a dense frame -- every tile a single wave's, s_h = its long lists -- deals the first pass only: thirteen static
    //  passes of config 4's 65 k tiles left the slowest wave 40 % behind the mean)
    const uint32_t static_passes = P.handout_static ? 0x7fffffffu / n_waves : (dense ? 1u : max(1u, (s_h + n_waves - 1u) / n_waves));
    const uint32_t n_static = static_passes * n_waves;
    // (a grid with fewer waves than decks -- a small device or partition -- uses as many decks as it
    //  has waves: a deck nobody draws from would leave its tiles unrendered)
    const uint32_t n_decks = min(kTicketParts, n_waves);
    const uint32_t part = wave_global % n_decks;
    uint32_t *const deck_ctr = &P.ctr_cur->ticket[part].count;
    uint32_t slot = pass_slot(0);
    uint4 qe = make_uint4(0u, 0u, 0u, 0u);
    uint32_t qix = 0;
    bool have = slot < n_slots;
    if (have) {
        qix = slot_entry(slot);
        qe = P.queue[qix];
    }
    for (uint32_t pass = 0; have; ++pass) {
        const uint32_t cur_slot = slot;
        const uint4 cur = Scalar4(qe);
        const bool draws = pass + 1u >= static_passes;  // the next tile is a drawn one
        if (!draws) {
            slot = pass_slot(pass + 1u);
            have = slot < n_slots;
            if (have) {
                qix = slot_entry(slot);
                qe = P.queue[qix];
            }
        }
        // (called once per tile, when only the encoding of its pixels is left: a card in hand is a
        //  tile nobody else can take)
        // (issuing the draw earlier, when the last chunk is staged, and looking at it here was
        //  measured: no difference -- what a drawn hand-out costs is that every wave stays busy)
        auto next_card = [&]() {
            if (!draws) return;
            uint32_t j = 0;
            if (lane == 0) j = __hip_atomic_fetch_add(deck_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            j = __builtin_amdgcn_readfirstlane(j);
            slot = n_static + part + n_decks * j;
            have = slot < n_slots;
            if (have) {
                qix = slot_entry(slot);
                qe = P.queue[qix];
            }
        };
        const bool wg_mode = cur_slot < s_h && sh != 0;  // uniform over the workgroup (slots are aligned)
        const uint32_t tile = (cur.x >> 16) * P.tiles_x + (cur.x & 0xffffu);  // (queue entries name a tile by column | row << 16)
        unsigned long long t_begin = 0;
        PhaseTicks prof;
        CoarseTicks ct;
        if (kProf) t_begin = wall_clock64();
        // (a wave for every queued tile: the workgroup tiles make their chunks' fragments together, RenderChunkWG)
        const uint32_t n_cmd = RenderQueuedTile<kFused, kProf, kCapture, false>(P, S, cur, wg_mode, cur_slot, pass, lane, wave, lanes_below, next_card, prof, ct,
                                                                                n_slots <= n_waves);
        if (kProf && lane == 0) {
            unsigned long long *d = P.dbg_time + 12ull * cur_slot;
            d[0] = t_begin;
            d[1] = wall_clock64();
            d[2] = tile | (wg_mode ? 0x80000000u : 0u);
            d[3] = (static_cast<unsigned long long>(wave_global) << 32) | n_cmd;
            d[4] = prof.a;  // workgroup mode: ticks in phase A (incl. its barriers) / phase B
            d[5] = prof.b;
            d[6] = prof.c;  // fused kernel: when the tile's list was complete (0: separate coarse pass)
            d[7] = prof.busy;  // workgroup mode: ticks this wave spent on its own items in phase A
            // list building by stage (fused kernel; workgroup mode: wave 0's row)
            d[8] = ct.hdr | (ct.cand << 32);
            d[9] = ct.owner | (ct.scan << 32);
            d[10] = ct.seg | (ct.emit << 32);
            d[11] = ct.rounds | (ct.records << 32);
        }
    }
}

// Validation kernel behind pm_fill_coverage: the winding coverage (alpha before colour) of the
// frame's Fill commands with an f32 accumulator -- signedArea as `float` instead of `half`
// (SURVEY.md D7's second mode, north star: "coverage within 1 ULP of the f32 reference").  One
// workgroup per tile, one pixel per thread, the command list read straight from HBM; every
// operation is the one of renderKernel :508-545 in binary32, in list order.  Not on the frame path.
__global__ __launch_bounds__(kThreads) void pm_coverage_kernel(FrameParams P, const uint32_t *tile_solid, float *out, uint32_t out_stride) {
    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % P.tiles_x, ty_rel = tile / P.tiles_x;
    const uint32_t xi = threadIdx.x & 15u, yi = threadIdx.x >> 4;
    const uint32_t pxi = tx * kTileW + xi, pyi = (P.row0 + ty_rel) * kTileH + yi;
    if (pxi >= P.width || pyi >= P.height) return;
    const float px = static_cast<float>(pxi), py = static_cast<float>(pyi);
    float cov = 0.0f;
    const uint32_t state = P.tile_state[tile];
    if (state != 0) {  // resolved by binning: background, or one opaque colour
        cov = state != 0xffffffffu ? 1.0f : 0.0f;
    } else {
        const uint32_t solid = tile_solid[tile];  // TileEncoder::end() as pm_coarse_kernel<true> recorded it
        if (solid != 0) {
            cov = solid != 0xffffffffu ? 1.0f : 0.0f;
        } else {
            const Cmd *cmds = reinterpret_cast<const Cmd *>(P.tarena + P.tile_ptcl[tile]);
            const uint32_t n = P.tile_ncmd[tile];
            float sa = 0.0f;
            for (uint32_t i = 0; i < n; ++i) {
                const Cmd cmd = cmds[i];
                if (cmd.tag == kCmdFill) {
                    const float sx = __uint_as_float(cmd.body[1]) - px, sy = __uint_as_float(cmd.body[2]) - py;
                    const float ex = __uint_as_float(cmd.body[3]) - px, ey = __uint_as_float(cmd.body[4]) - py;
                    const float wx = Sat(sy), wy = Sat(ey);
                    if (wx != wy) {
                        const float tx_ = (wx - sy) / (ey - sy);
                        const float ty_ = (wy - sy) / (ey - sy);
                        const float xsx = sx + (ex - sx) * tx_;
                        const float xsy = sx + (ex - sx) * ty_;
                        const float xmin = fminf(fminf(xsx, xsy), 1.0f) - 1e-6f;
                        const float xmax = fmaxf(xsx, xsy);
                        const float b = fminf(xmax, 1.0f);
                        const float c = fmaxf(b, 0.0f);
                        const float d = fmaxf(xmin, 0.0f);
                        const float area = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
                        sa += area * (wx - wy);
                    }
                } else if (cmd.tag == kCmdFillEdge) {
                    sa += static_cast<float>(static_cast<int>(cmd.body[0])) * Sat(py - __uint_as_float(cmd.body[1]) + 1.0f);
                } else if (cmd.tag == kCmdDrawFill) {
                    cov = sa + static_cast<float>(static_cast<int>(cmd.body[0]));
                    if (cmd.body[4] & kFillEvenOdd) cov = fabsf(cov - 2.0f * rintf(0.5f * cov));
                    else cov = fminf(fabsf(cov), 1.0f);
                    sa = 0.0f;
                } else if (cmd.tag == kCmdSolid) {
                    cov = 1.0f;
                }
            }
        }
    }
    out[static_cast<size_t>(ty_rel * kTileH + yi) * out_stride + pxi] = cov;
}

// ---- launch wrappers (called from pm_context.hip) -----------------------------------------

void LaunchClear(const FrameParams &p, uint32_t n_striprows, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    PM_LAUNCH(pm_clear_kernel, dim3(n_striprows), dim3(kBinThreads), stream, t0, t1, p);
}

void LaunchCoverage(const FrameParams &p, uint32_t n_tiles, const uint32_t *tile_solid, float *out, uint32_t out_stride, hipStream_t stream) {
    hipLaunchKernelGGL(pm_coverage_kernel, dim3(n_tiles), dim3(kThreads), 0, stream, p, tile_solid, out, out_stride);
}

void LaunchFine(const FrameParams &p, uint32_t clear_blocks, bool fused, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    const dim3 grid(p.fine_grid + clear_blocks), block(kThreads);
    if (fused && p.dbg_counts) {  // list capture from the frame path's own kernel
        PM_LAUNCH((pm_fine_kernel<true, false, true>), grid, block, stream, t0, t1, p);
    } else if (p.dbg_time) {
        if (fused)
            PM_LAUNCH((pm_fine_kernel<true, true>), grid, block, stream, t0, t1, p);
        else
            PM_LAUNCH((pm_fine_kernel<false, true>), grid, block, stream, t0, t1, p);
    } else if (fused && p.fine_dense) {
        PM_LAUNCH((pm_fine_kernel<true, false, false, true>), grid, block, stream, t0, t1, p);
    } else if (fused) {
        PM_LAUNCH((pm_fine_kernel<true, false>), grid, block, stream, t0, t1, p);
    } else {
        PM_LAUNCH((pm_fine_kernel<false, false>), grid, block, stream, t0, t1, p);
    }
}

}  // namespace pm
