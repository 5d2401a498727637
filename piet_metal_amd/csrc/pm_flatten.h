#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/piet_metal_amd.h"

namespace pm {

// Device buffers of the flatten stage, kept between calls (grow only): the parsed paths and the
// scratch arrays.  With `resident` set the paths of the last upload are flattened again -- the
// per-frame re-encode of an animation (PietRenderer.m:90-101 does it on the CPU) then costs four
// kernels and two small read-backs, no allocation and no path upload.
struct FlattenCache {
    pm_path *d_paths = nullptr;
    pm_path_el *d_els = nullptr;
    uint32_t *d_u32 = nullptr;
    double *d_bbox = nullptr;
    size_t cap_paths = 0, cap_els = 0, cap_u32 = 0, cap_bbox = 0;
    size_t n_paths = 0, n_els = 0;  // what is resident in d_paths / d_els
    size_t max_items = 0;           // ... and an upper bound of the items they encode to
    // pinned: {totals[4], error flag, pad} then, at +32, the head of the scene the kernels just wrote
    // (SimpleGroup, boxes, items) -- fetched in the same wait as the totals, for validation and arena sizing
    uint8_t *h_meta = nullptr;
    size_t cap_meta = 0, meta_bytes = 0;  // meta_bytes: valid bytes at h_meta + 32 (0: not fetched)
    bool resident = false;
    void Free();
    hipError_t Reserve(size_t n_paths, size_t n_els);  // room for this many paths / elements (pm_create)
};

// Flatten + encode on the device (see pm_flatten.hip).  Synchronises `stream` (once, at the end).
// use_resident: ignore h_paths / h_els and flatten the paths resident in `cache` again.
// On PM_ERR_CAPACITY *scene_bytes holds the size that would have been needed.
int FlattenEncodeOnDevice(hipStream_t stream, FlattenCache *cache, bool use_resident, const pm_path *h_paths, size_t n_paths, const pm_path_el *h_els,
                          size_t n_els, const double affine[6], float width_scale, uint8_t *d_scene, size_t scene_cap,
                          size_t *scene_bytes, uint32_t *n_items_out, hipError_t *hip_error);

}  // namespace pm
