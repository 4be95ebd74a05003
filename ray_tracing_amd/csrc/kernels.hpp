// Launchers for the HIP kernels in kernels.hip.
#pragma once

#include <hip/hip_runtime_api.h>

#include "device_types.hpp"

namespace rt {

// Whole-path kernel.  `stack_needed` = entries of per-lane traversal stack the
// scene's deepest BVH can use; `count` keeps the reference's debug counters;
// `variant` 0 = plain nested loops, 1 = per-lane state machine (same results).
hipError_t launch_path_kernel(const DevScene& sc, const DevCam& cam, const DevRows& rows, const DevFrame& fr,
                              uint32_t n_frames, uint32_t stack_needed, bool count, int variant,
                              hipStream_t stream);

}  // namespace rt
