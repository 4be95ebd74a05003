// HIP kernels for gfx950 (MI355X).  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "trace_device.hpp"

namespace rt {

using namespace dev;

namespace {

constexpr int kBlock = 256;  // 4 waves; each wave owns an 8x8 pixel tile
constexpr int kTileW = 32;   // block tile: 32 x 8 pixels
constexpr int kTileH = 8;

// Sum over the 64 lanes of a wave; the total is valid in lane 0.
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Consecutive workgroup ids are dealt round-robin over the 8 XCDs, each with
// its own L2.  Give every XCD a contiguous run of tiles (so neighbouring
// tiles, which walk the same BVH nodes, share an L2) with a mapping that stays
// a bijection when the tile count is not a multiple of 8.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t n) {
    uint32_t xcd = bid & 7u, k = bid >> 3;
    uint32_t q = n >> 3, r = n & 7u;
    uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + k;
}

// Row `ly` of this context's share of the canvas -> canvas row.
__device__ __forceinline__ uint32_t global_row(const DevRows& rows, uint32_t ly) {
    if (rows.world <= 1u) return ly;
    uint32_t s = ly / rows.strip_h, o = ly - s * rows.strip_h;
    return (s * rows.world + rows.rank) * rows.strip_h + o;
}

}  // namespace

// ---------------------------------------------------------------------------
// Whole-path kernel: the shape of the reference's shader::main
// (shader/src/lib.rs:23-109).  One lane = one pixel = `n_frames` consecutive
// one-sample frames of that pixel; the accumulator value stays in registers
// between them (a pixel's frame k+1 depends only on its own frame k through
// the seed, lib.rs:57-60) and is written once.
// ---------------------------------------------------------------------------
template <int STACK_N, bool COUNT>
__global__ __launch_bounds__(kBlock) void path_kernel(DevScene sc, DevCam cam, DevRows rows, DevFrame fr,
                                                      uint32_t n_frames, uint32_t tiles_x, uint32_t n_tiles) {
    __shared__ uint32_t s_stack[STACK_N * kBlock];

    const uint32_t tile = xcd_remap(blockIdx.x, n_tiles);
    const uint32_t tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = tile_x * kTileW + wave * 8u + (lane & 7u);
    const uint32_t ly = tile_y * kTileH + (lane >> 3);
    const bool inside = x < cam.W && ly < rows.local_rows;
    const uint32_t y = global_row(rows, ly);

    unsigned long long n_rays = 0, n_box = 0, n_internal = 0, n_tri = 0, n_paths = 0;

    if (inside && y < cam.H) {
        uint32_t* stack = &s_stack[threadIdx.x];
        const uint32_t idx = x + y * cam.W;
        float4 acc = reinterpret_cast<const float4*>(fr.acc)[idx];
        float4 present = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        HitRecord first;
        record_new(first);

        for (uint32_t f = 0; f < n_frames; ++f) {
            // shader/src/lib.rs:57-69
            uint32_t seed = get_seed(cam.frame, x, y, acc.x, acc.y, acc.z);
            // Ray::get_color, trace.rs:175-206
            V3 color = v3(1.0f, 1.0f, 1.0f);
            float jx = rand_float(seed, 0.0f, 1.0f);
            float jy = rand_float(seed, 0.0f, 1.0f);
            Ray ray = camera_ray(cam, (float)x + jx, (float)y + jy);
            ray.dir = normalize(ray.dir);
            bool finished = false;
            for (uint32_t seg = 0; seg < cam.depth; ++seg) {
                HitRecord rec;
                closest_hit<COUNT>(sc, ray, rec, stack, kBlock);
                n_rays += 1;
                if (COUNT) {
                    n_box += rec.box_tests;
                    n_internal += rec.internal_visits;
                    n_tri += rec.triangle_tests;
                }
                if (seg == 0) first = rec;
                int state = shade(sc, cam, rec, ray, seed, color);
                if (state != STATE_RAY) {
                    finished = true;
                    break;
                }
            }
            if (!finished) color = v3(0.0f, 0.0f, 0.0f);  // trace.rs:204-205
            n_paths += 1;
            float4 next_acc;
            accumulate(cam, acc, color, next_acc, present);
            acc = next_acc;
            // the event loop's bookkeeping between frames, main.rs:147-148
            cam.frame += 1u;
            cam.frames_without_move += 1.0f;
        }

        reinterpret_cast<float4*>(fr.acc)[idx] = acc;
        if (fr.present) reinterpret_cast<float4*>(fr.present)[idx] = present;
        if (fr.hit_triangle) {
            fr.hit_triangle[idx] = first.triangle_id;
            fr.hit_instance[idx] = first.instance_id;
            fr.hit_t_bits[idx] = __float_as_uint(first.t);
        }
        if (COUNT && fr.box_tests) {
            fr.box_tests[idx] = first.box_tests;
            fr.triangle_tests[idx] = first.triangle_tests;
        }
    }

    n_rays = wave_sum(n_rays);
    n_paths = wave_sum(n_paths);
    if (COUNT) {
        n_box = wave_sum(n_box);
        n_internal = wave_sum(n_internal);
        n_tri = wave_sum(n_tri);
    }
    if (lane == 0 && fr.stats) {
        atomicAdd(&fr.stats->rays, n_rays);
        atomicAdd(&fr.stats->paths, n_paths);
        if (COUNT) {
            atomicAdd(&fr.stats->box_tests, n_box);
            atomicAdd(&fr.stats->internal_visits, n_internal);
            atomicAdd(&fr.stats->triangle_tests, n_tri);
        }
    }
}

hipError_t launch_path_kernel(const DevScene& sc, const DevCam& cam, const DevRows& rows, const DevFrame& fr,
                              uint32_t n_frames, uint32_t stack_needed, bool count, hipStream_t stream) {
    const uint32_t tiles_x = (cam.W + kTileW - 1) / kTileW;
    const uint32_t tiles_y = (rows.local_rows + kTileH - 1) / kTileH;
    const uint32_t n_tiles = tiles_x * tiles_y;
    if (n_tiles == 0) return hipSuccess;
    dim3 grid(n_tiles), block(kBlock);
#define RT_LAUNCH(N, C) \
    hipLaunchKernelGGL((path_kernel<N, C>), grid, block, 0, stream, sc, cam, rows, fr, n_frames, tiles_x, n_tiles)
    if (stack_needed <= 32) {
        if (count) RT_LAUNCH(32, true); else RT_LAUNCH(32, false);
    } else if (stack_needed <= 64) {
        if (count) RT_LAUNCH(64, true); else RT_LAUNCH(64, false);
    } else {
        return hipErrorInvalidValue;
    }
#undef RT_LAUNCH
    return hipGetLastError();
}

}  // namespace rt
