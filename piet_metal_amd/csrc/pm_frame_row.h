// What a strip row's workgroup hands from its binning role to its tile role inside pm_frame_kernel (one launch per frame).
#pragma once
#include "pm_device.h"

namespace pm {

// A one-launch frame: how many of its strip row's queued tiles a workgroup renders itself, straight after binning them, without
// any hand-over -- the row's longest list if that is one a whole workgroup renders (n_heavy of them in the row), else up to
// four single-wave tiles, a wave each.  The others go to the frame's FIFOs.
__host__ __device__ inline uint32_t OneLaunchKeep(uint32_t n_heavy, uint32_t n_queued) { return n_heavy ? 1u : (n_queued < 4u ? n_queued : 4u); }

// Outside the union of the two roles' working sets: written at the end of the binning role, read by the tile role.
struct FrameRowLds {
    uint4 entry[4];                // queue entries of the tiles the workgroup keeps (longest list first)
    uint32_t state[kStripTiles];   // tile_state of the row's tiles: 0 queued, else the resolved colour (the workgroup writes those pixels)
    uint32_t n_keep, keep_heavy;
    uint32_t striprow;             // the row (strip + row of the band x strips), 0xffffffff: none
    // the FIFO stage
    uint32_t wg_call;              // 0 nothing, 1 wave 0 holds a tile for the whole workgroup (h_entry), 2 the frame is done
    uint32_t tiles_done;           // workgroup tiles rendered so far (parity of the hand-over words)
    uint32_t busy;                 // waves 1-3 in the middle of a tile of their own
    uint4 h_entry;
};

}  // namespace pm
