// pm_frame_kernel: a whole frame in ONE launch -- the two dispatches of PietRenderer.m:69-88 (tileKernel, renderKernel) as roles
// of one resident grid.  (See pm_kernels_common.h for the decomposition; pm_bin_rows.h and pm_fine_tile.h hold the two stages.)
//
// Two launches per frame end twice: binning when its heaviest strip row is through (the mean one at 60 % of that), the tile
// kernel when its longest list is (the mean wave at 65 %), a launch boundary in between -- and nothing of the second runs
// under the first's tail.  Here a workgroup
//   1. bins ONE strip row (BinStripRows<.., kOne>: pieces stored write-through),
//   2. renders that row's first tiles itself, straight from its own LDS hand-over (FrameRowLds: no queue, no wait) -- the
//      longest list with all four waves if it is a workgroup's, else up to four tiles, a wave each --
//   3. and then takes tiles from the frame's FIFOs, where every row left the tiles it did not keep: fifo[0] for tiles a whole
//      workgroup renders (wave 0 takes them and calls the other three), fifo[1 + XCD] for single-wave tiles (every wave for
//      itself), until every strip row has been handed over and the FIFOs are empty;
//   4. in its first idle moment it writes the pixels of its row's resolved tiles and of its share of the strip rows no item reaches.
// A frame is over when its slowest CHAIN is -- a strip row, then its longest list -- not when the slowest row and then the longest
// list are.
//
// Hand-over protocol (MI355X: eight XCDs, L2s not coherent with each other).  Producer: pieces and FIFO entries are `sc1`
// stores; every wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, THEN the row's entries are written: an entry that
// can be seen names pieces that are in memory.  FIFO places are reserved with one returning atomic per row (RowTailIssue), an
// entry is in place when its words are non-zero; whoever takes it zeroes it again (the arrays are all zero between frames).
// Consumer: looks at {tail, head} with one agent-scope load, takes a ticket (atomic on head) only when head < tail, then reads
// entry[ticket] until it is there; pieces are read with agent-scope loads (CoarseTile<.., kCoh>).  A ticket beyond the final tail
// (two waves saw the same last entry) is given up when `done` is up.  `done`: every binning workgroup counts itself in after its
// entries (eight counters by blockIdx % 8, the last of each part counts the part in, the last part raises the flag).
// No wait is unbounded: after P.spin_ticks a wave gives up, raises *P.host_fail, and pm_sync renders the frame again with two
// launches.  Nothing here depends on which workgroup runs where or when: the grid only has to be resident (pm_context.hip takes
// the occupancy API's word for that, and never puts two one-launch frames on the device at once).
#include "pm_bin_rows.h"
#include "pm_fine_tile.h"

namespace pm {

namespace {

// the working sets of the two roles share their bytes; the hand-over between them lives beside
struct FrameLds {
    union {
        BinLds<4, false> bin;
        SparseLds tile;
    };
    FrameRowLds row;
};
// Four workgroups per CU (128 VGPRs).  At five (96) each role alone fits its registers -- pm_bin_kernel and pm_fine_kernel do --
// but the two in one kernel spill 45 of them to scratch, and every tile then takes 1.6 times as long (measured: config 2's
// FIFO tiles 15.5 -> 9.5 us at four per CU).  The grid is 4 x CUs workgroups; a frame with more strip rows than that gives its
// lightest workgroups a second row (pm_context.hip, the one-launch chains).
#ifndef PM_FRAME_WPS
#define PM_FRAME_WPS 4
#endif
static_assert(sizeof(FrameLds) <= 40960, "four workgroups per CU");

// pixels of one strip row's resolved tiles, this wave's quarter of the rows (ClearStripRow with the tile states in hand)
__device__ __forceinline__ void ClearStripRowWave(const FrameParams &P, uint32_t striprow, uint32_t state, uint32_t lane, uint32_t wave) {
    const uint32_t strip = striprow % P.strips_x;
    const uint32_t row_rel = striprow / P.strips_x;
    const uint32_t tx = strip * kStripTiles + (lane >> 2);
    if (tx >= P.tiles_x || state == 0u) return;
    const uint32_t col = StoreOrder(state, P.fb_bgra);
    const uint32_t px = strip * kGroupW + lane * 4u;
    const uint32_t y0 = (P.row0 + row_rel) * kTileH;
#pragma unroll
    for (uint32_t it = 0; it < kTileH / kWaves; ++it) {
        const uint32_t r = it * kWaves + wave;
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

// One wave takes the next entry of FIFO f, if there is one.  Uniform over the wave.
__device__ __forceinline__ bool PopEntry(const FrameParams &P, uint32_t f, uint4 *out, uint32_t lane) {
    Fifo *const ff = &P.ctr_cur->fifo[f];
    const uint2 th = LoadCoherent8(ff);  // {tail, head}
    if (static_cast<int>(__builtin_amdgcn_readfirstlane(static_cast<int>(th.x - th.y))) <= 0) return false;
    uint32_t t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(&ff->head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t)));
    if (t >= P.fifo_cap) return false;  // (cannot happen: a tile is pushed once)
    uint4 *const e = P.fifo + static_cast<size_t>(f) * P.fifo_cap + t;
    const unsigned long long t0 = PollClock();
    for (;;) {
        const uint4 v = Scalar4(LoadCoherent16(e));
        if (v.y != 0u && v.w != 0u) {  // in place (both halves)
            if (lane == 0) StoreWT16(e, make_uint4(0u, 0u, 0u, 0u));  // the arrays are all zero between frames
            *out = v;
            return true;
        }
        // Not yet -- or never: a ticket taken for an entry another wave had seen as well.  Once every row is handed over the
        // tails are final.
        if (__builtin_amdgcn_readfirstlane(static_cast<int>(LoadCoherent4(&P.ctr_cur->done_top.done))) != 0 &&
            t >= static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(LoadCoherent4(&ff->tail)))))
            return false;
        if (__builtin_amdgcn_readfirstlane(static_cast<int>(PollClock() - t0 > P.spin_ticks ? 1 : 0))) {
            if (lane == 0) *P.host_fail = 1u;
            return false;
        }
        SleepPoll();
    }
}

__device__ __forceinline__ bool FifoEmpty(const FrameParams &P, uint32_t f) {
    const uint2 th = LoadCoherent8(&P.ctr_cur->fifo[f]);
    return static_cast<int>(__builtin_amdgcn_readfirstlane(static_cast<int>(th.x - th.y))) <= 0;
}

}  // namespace

// kCapture (pm_debug_capture_ptcl): every list the frame builds is also recorded in the reference's layout -- the lists of
// this very launch, FIFO hand-overs and write-through pieces included.
// kProf (pm_debug_time_frame): per workgroup 32 clocks / counts to P.dbg_time, written in one burst when a wave leaves --
//   [0] entry, [1] strip row binned and handed over, [2] tiles for the whole workgroup rendered, [3] 10 ns ticks inside them;
//   wave w, [8 + 6 w ..]: end of the tile it kept, first moment without work, exit, tiles taken from the FIFOs, ticks inside
//   them, polls that found nothing.
template <bool kCapture, bool kProf = false>
__global__ __launch_bounds__(kThreads, PM_FRAME_WPS) void pm_frame_kernel(FrameParams P) {
    unsigned long long pf_entry = 0, pf_bin = 0, pf_own = 0, pf_idle = 0, pf_twg = 0, pf_tlight = 0;
    uint32_t pf_nwg = 0, pf_nlight = 0, pf_polls = 0;
    if (kProf) pf_entry = wall_clock64();
    __shared__ FrameLds S;
    FrameRowLds &R = S.row;
    const bool bins = (P.one_launch & 1u) != 0, steals = (P.one_launch & 2u) != 0;
    const bool has_row = bins && blockIdx.x < P.one_grid_rows;
    if (threadIdx.x < kStripTiles) R.state[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        R.n_keep = 0;
        R.keep_heavy = 0;
        R.wg_call = 0;
        R.tiles_done = 0;
        R.busy = 0;
        R.striprow = 0xffffffffu;
    }
    __syncthreads();
    if (bins) {
        // ---- role 1: the strip row (block 0 also resets the other parity's counters) ----
        __builtin_amdgcn_s_setprio(3);  // a strip row is the head of a chain (its tiles follow): it wins the issue arbitration over tiles
        if (has_row || blockIdx.x == 0) BinStripRows<false, 4, true>(P, S.bin, &R);
        __builtin_amdgcn_s_setprio(0);
        if (has_row && WaveId() == kWaves - 1 && LaneId() == 0) {
            // this row is handed over (RowTailFinish, the same wave: its places in the FIFOs were reserved before)
            Counters *const c = P.ctr_cur;
            const uint32_t part = blockIdx.x & (kFifoShards - 1u);
            const uint32_t expect = (P.one_grid_rows - part + kFifoShards - 1u) / kFifoShards;  // binning workgroups b with b % 8 == part
            if (__hip_atomic_fetch_add(&c->done_part[part].count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == expect) {
                const uint32_t n_parts = min(P.one_grid_rows, kFifoShards);
                if (__hip_atomic_fetch_add(&c->done_top.parts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == n_parts) StoreWT4(&c->done_top.done, 1u);
            }
        }
        __syncthreads();  // R is complete; the binning role's LDS is done with
    }
    if (kProf) pf_bin = wall_clock64();
    // ---- roles 2 and 3: the row's own first tiles, then tiles from the FIFOs until the frame is done; clearing in the first
    //      idle moment.  ONE place renders a tile, whatever kind and wherever from (the tile stage is 3 000 instructions long) ----
    // (nothing in a vector register lives across the binning role: its register allocation is that of pm_bin_kernel, to the last one)
    const uint32_t lane = LaneId(), wave = WaveId();
    PhaseTicks prof;
    CoarseTicks ct;
    auto no_card = [] {};
    // the pixels this workgroup owes: its row's resolved tiles, its share of the strip rows no item reaches (this wave's quarter)
    auto clear_owed = [&] {
        const uint32_t sr = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(R.striprow)));
        if (sr != 0xffffffffu) ClearStripRowWave(P, sr, R.state[lane >> 2], lane, wave);
        const uint32_t g = gridDim.x;
        for (uint32_t j = (blockIdx.x + g - P.one_grid_rows % g) % g; j < P.n_idle_sr; j += g) ClearStripRowWave(P, P.idle_sr[j], 0xffffffffu, lane, wave);
    };
    // what the row's workgroup kept: its longest list for all four waves (through the call word), or a tile per wave
    uint32_t own_left = 0;
    {
        const uint32_t n_keep = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(R.n_keep)));
        if (n_keep != 0u) {
            if (R.keep_heavy) {
                if (threadIdx.x == 0) {
                    R.h_entry = R.entry[0];
                    R.wg_call = 1u;
                }
                __syncthreads();
            } else if (wave < n_keep) {
                own_left = 1u;
            }
        }
    }
    const uint32_t own = blockIdx.x & (kFifoShards - 1u);
    uint32_t other = 1u;        // the other shard this wave looks at next
    bool cleared = !bins;       // (a steal-only launch: the binning launch has cleared)
    unsigned long long idle_since = PollClock();
    bool kept = false;
    for (;;) {
        const uint32_t call = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(*reinterpret_cast<volatile uint32_t *>(&R.wg_call))));
        if (call == 2u && own_left == 0u) break;  // (wave 0 may see the frame's end before this wave has rendered the tile it kept)
        uint4 e = make_uint4(0u, 0u, 0u, 0u);
        uint32_t kind = call == 1u ? 2u : 0u;  // 0 nothing, 1 a tile of this wave's own, 2 a tile of the whole workgroup
        // (wave 0 calls the workgroup together only when the other waves are free: a tile for a whole workgroup is the frame's
        //  longest kind, and three waves in the middle of tiles of their own would keep it waiting)
        if (kind == 0u && wave == 0u && steals && own_left == 0u && *reinterpret_cast<volatile uint32_t *>(&R.busy) == 0u) {
            if (PopEntry(P, 0u, &e, lane)) {
                if (lane == 0) {
                    R.h_entry = e;
                    R.wg_call = 1u;
                }
                kind = 2u;
            }
        }
        if (kind == 0u) {
            kept = false;
            if (own_left != 0u) {
                own_left = 0u;
                e = Scalar4(R.entry[wave]);
                kind = 1u;
                kept = true;
            } else if (steals) {
                bool got = PopEntry(P, 1u + own, &e, lane);
                if (!got) {
                    got = PopEntry(P, 1u + ((own + other) & (kFifoShards - 1u)), &e, lane);
                    other = other % (kFifoShards - 1u) + 1u;
                }
                if (got) kind = 1u;
            }
        }
        if (kind != 0u) {
            uint32_t parity = 0;
            if (kind == 2u) {
                LdsBarrier();  // every wave is here: the entry is in R
                e = Scalar4(R.h_entry);
                parity = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(R.tiles_done)));
            } else if (lane == 0 && wave != 0u) {
                atomicAdd(&R.busy, 1u);
            }
            // (the lane number made here, per tile, from a value the compiler cannot see through: everything derived from it --
            //  LDS addresses, pixel coordinates -- is otherwise hoisted in front of the loop, kept live across it and spilled)
            const uint32_t ln = Opaque(lane);
            unsigned long long pf_t0 = 0;
            if (kProf) pf_t0 = wall_clock64();
            RenderQueuedTile<true, false, kCapture, true>(P, S.tile, e, kind == 2u, wave, parity, ln, wave, (1ull << ln) - 1ull, no_card, prof, ct);
            if (kProf) {
                const unsigned long long t1 = wall_clock64();
                if (kind == 2u) {
                    pf_nwg += 1;
                    pf_twg += t1 - pf_t0;
                } else if (pf_own == 0 && kept) {
                    pf_own = t1;
                } else {
                    pf_nlight += 1;
                    pf_tlight += t1 - pf_t0;
                }
            }
            if (kind == 2u) {
                LdsBarrier();  // ... and has read it
                if (threadIdx.x == 0) {
                    R.wg_call = 0u;
                    R.tiles_done = parity + 1u;
                }
                LdsBarrier();  // nobody loops on the old call
            } else if (lane == 0 && wave != 0u) {
                atomicAdd(&R.busy, 0xffffffffu);
            }
            idle_since = PollClock();
            continue;
        }
        // nothing to render right now: the pixels this workgroup owes
        if (kProf && pf_idle == 0) pf_idle = wall_clock64();
        if (!cleared) {
            cleared = true;
            clear_owed();
            continue;
        }
        if (wave == 0u) {
            // the end: every row handed over, every FIFO empty (the tails are final once `done` is up, the heads only grow)
            bool over = !steals;
            if (steals && __builtin_amdgcn_readfirstlane(static_cast<int>(LoadCoherent4(&P.ctr_cur->done_top.done))) != 0) {
                over = true;
                for (uint32_t f = 0; f < kFifos && over; ++f) over = FifoEmpty(P, f);
            }
            if (!over && __builtin_amdgcn_readfirstlane(static_cast<int>(PollClock() - idle_since > P.spin_ticks ? 1 : 0))) {
                if (lane == 0) *P.host_fail = 1u;  // (pm_sync renders the frame again, with two launches)
                over = true;
            }
            if (over) {
                if (lane == 0) R.wg_call = 2u;
                break;
            }
        }
        if (kProf) pf_polls += 1;
        SleepPoll();
    }
    if (!cleared) clear_owed();  // (a wave that was busy until the end)
    if (kProf && lane == 0) {
        unsigned long long *d = P.dbg_time + 32ull * blockIdx.x;
        if (wave == 0u) {
            d[0] = pf_entry;
            d[1] = pf_bin;
            d[2] = pf_nwg;
            d[3] = pf_twg;
        }
        unsigned long long *w = d + 8u + 6u * wave;
        w[0] = pf_own;
        w[1] = pf_idle;
        w[2] = wall_clock64();
        w[3] = pf_nlight;
        w[4] = pf_tlight;
        w[5] = pf_polls;
    }
}

void LaunchFrame(const FrameParams &p, uint32_t grid, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    if (p.dbg_time)
        PM_LAUNCH((pm_frame_kernel<false, true>), dim3(grid), dim3(kThreads), stream, t0, t1, p);
    else if (p.dbg_counts)
        PM_LAUNCH(pm_frame_kernel<true>, dim3(grid), dim3(kThreads), stream, t0, t1, p);
    else
        PM_LAUNCH(pm_frame_kernel<false>, dim3(grid), dim3(kThreads), stream, t0, t1, p);
}

int FrameKernelResidency() {
#ifdef PM_EMU
    return 5;  // (the CPU emulation runs one workgroup at a time: residency is not its subject)
#else
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pm_frame_kernel<false>, kThreads, 0) != hipSuccess) return 0;
    return n;
#endif
}

}  // namespace pm
