#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/piet_metal_amd.h"

namespace pm {

// Flatten + encode on the device (see pm_flatten.hip).  Synchronises `stream`.
// On PM_ERR_CAPACITY *scene_bytes holds the size that would have been needed.
int FlattenEncodeOnDevice(hipStream_t stream, const pm_path *h_paths, size_t n_paths, const pm_path_el *h_els,
                          size_t n_els, const double affine[6], float width_scale, uint8_t *d_scene, size_t scene_cap,
                          size_t *scene_bytes, uint32_t *n_items_out, hipError_t *hip_error);

}  // namespace pm
