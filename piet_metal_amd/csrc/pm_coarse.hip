// pm_coarse_kernel: the tile-level half of tileKernel + TileEncoder, one wave per queued tile
// (see pm_kernels_common.h for the decomposition and the rules shared by the three files)
#include "pm_kernels_common.h"
#include "pm_coarse_tile.h"

namespace pm {

// K2: per-tile command lists (tile-level half of tileKernel), one wave per tile
// =====================================================================================

template <bool kCapture>
__global__ __launch_bounds__(kThreads, PM_COARSE_WPS) void pm_coarse_kernel(FrameParams P) {
    __shared__ CoarseLds s_lds[kWaves];
    CoarseLds &L = s_lds[WaveId()];

    const uint32_t lane = LaneId();
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    uint32_t cls_end[kClasses];  // running totals of the class queues (longest lists first)
    {
        uint32_t run = 0;
#pragma unroll
        for (uint32_t k = 0; k < kClasses; ++k) {
            run += P.ctr_cur->cls[k].count;
            cls_end[k] = run;
        }
    }
    const uint32_t n_total = cls_end[kClasses - 1];

    // Static snake hand-out over [longest lists..., shortest...]: pass k gives wave g the slot
    // k*G + g (k even) or k*G + (G-1-g) (k odd).  No atomics: one device-scope counter tops out
    // near 90 dequeues/us on this chip, far below the tile rate.
    const uint32_t wave_global = blockIdx.x * kWaves + (WaveId());
    const uint32_t n_waves = gridDim.x * kWaves;

    for (uint32_t pass = 0; pass * n_waves < n_total; ++pass) {
        const uint32_t slot = pass * n_waves + ((pass & 1u) ? (n_waves - 1u - wave_global) : wave_global);
        if (slot >= n_total) continue;
        uint32_t qix = slot;  // class 0
#pragma unroll
        for (uint32_t k = 1; k < kClasses; ++k)
            if (slot >= cls_end[k - 1]) qix = k * P.queue_cap + (slot - cls_end[k - 1]);
        const uint4 qe = Scalar4(P.queue[qix]);
        CoarseTile<kCapture>(P, L, qe, lane, lanes_below);
        WaveSync();  // L reuse by the next tile
    }
}

// ---- launch wrappers (called from pm_context.hip) -----------------------------------------

void LaunchCoarse(const FrameParams &p, uint32_t grid, bool capture, hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    if (capture)
        PM_LAUNCH(pm_coarse_kernel<true>, dim3(grid), dim3(kThreads), stream, t0, t1, p);
    else
        PM_LAUNCH(pm_coarse_kernel<false>, dim3(grid), dim3(kThreads), stream, t0, t1, p);
}

}  // namespace pm
