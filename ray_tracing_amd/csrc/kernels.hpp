// Launchers for the HIP kernels in kernels.hip.
#pragma once

#include <hip/hip_runtime_api.h>

#include "device_types.hpp"

namespace rt {

// Plain whole-path kernel (one lane per pixel, nested loops).  `stack_needed` =
// entries of per-lane traversal stack the scene's deepest BVH can use; `count`
// keeps the reference's debug counters.
hipError_t launch_path_kernel(const DevScene& sc, const DevCam& cam, const DevRows& rows, const DevFrame& fr,
                              uint32_t n_frames, uint32_t stack_needed, bool count, hipStream_t stream);

// Persistent whole-path kernel (per-lane state machine, vote-scheduled, work
// queues).  `q.head` must point at 8 * 32 words of device memory; `n_blocks`
// is the number of resident workgroups to launch.  With `q.cost` and `order_out`
// (one word per 8x8 tile each) the launch also leaves the tile order for the
// next launch of the same canvas in `order_out`; `q.order`, when set, is such an
// order from an earlier launch.
hipError_t launch_persistent_kernel(const DevScene& sc, const DevCam& cam, const DevRows& rows, const DevFrame& fr,
                                    DevQueue q, uint32_t* order_out, uint32_t n_frames, uint32_t stack_needed,
                                    bool count, bool big, uint32_t n_blocks, void* pool_records, hipStream_t stream);
// `pool_records` != NULL runs the pooled form of the kernel (lanes only traverse; shading happens in
// steps of its own over a pool of ray slots per wave): device memory of pool_record_bytes(n_blocks).
size_t pool_record_bytes(uint32_t n_blocks);
uint32_t persistent_tile_count(const DevCam& cam, const DevRows& rows);

// Scratch of the no-allocation render entry point: a device array of the code object, one copy
// per GPU.  noalloc_scratch_init looks its address up once per device (call it with that device
// current); noalloc_scratch_view hands out the pieces and the host-side tile-order state that
// goes with them (shared by every context on that GPU).  Canvases of more than
// kNoAllocMaxTiles 8x8-pixel tiles do not fit.
constexpr uint32_t kNoAllocMaxTiles = 1u << 18;  // 16.7 M pixels, e.g. 5120 x 2880; 3 MiB of scratch per GPU
struct NoAllocView {
    uint32_t* head;
    uint32_t* cost;
    uint32_t* order[2];
    int* order_current;
    uint32_t* order_sig;  // 5 words
};
hipError_t noalloc_scratch_init(int device);
bool noalloc_scratch_view(int device, NoAllocView* out);

// Binding 8 of this context's rows from binding 7 as it stands after `n_frames` frames that began at
// `cam` (shader/src/lib.rs:85-108: the accumulator over the sample count, gamma 1/2.2).
hipError_t launch_present(const DevCam& cam, const DevRows& rows, const DevFrame& fr, uint32_t n_frames, hipStream_t stream);

// Accumulators of one canvas added in list order into `out` (which may be the first of them).
constexpr uint32_t kMaxGroup = 16;
struct AccumulatorList {
    const float* src[kMaxGroup];
    uint32_t n;
};
hipError_t launch_sum_accumulators(float* out, const AccumulatorList& list, uint32_t n_pixels, hipStream_t stream);

// DevTri::wn (the world-space shading normal) of every triangle whose object exactly one instance shows:
// `owner` names each triangle's object, `object_instance` that object's one instance (0xFFFFFFFF: none or
// several).  To run after the scene or the instances changed, before the next render on `stream`.
hipError_t launch_world_normals(DevTri* tris, uint32_t n_tris, const uint32_t* owner, const uint32_t* object_instance,
                                const DevInstance* instances, hipStream_t stream);

// Binding 8 (RGBA32F) -> 8-bit RGBA, one word per pixel, R in the low byte.
hipError_t launch_present_bytes(const float* present, uint32_t n_pixels, uint32_t* rgba8, hipStream_t stream);

// Workgroups of the persistent kernel one CU can hold (occupancy query).
int persistent_blocks_per_cu(uint32_t stack_needed, bool count, bool big, bool pool);

}  // namespace rt
