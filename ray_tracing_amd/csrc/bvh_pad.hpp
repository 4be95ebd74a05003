// Flat boxes in the trees of the GPU builders.
//
// The reference's slab test accepts a box only when t_near < t_far (trace.rs:208-229): a box with no
// thickness along an axis -- the bounds of an axis-aligned wall, of any set of coplanar axis-aligned
// triangles -- gives t_near == t_far on that axis and is never hit, and what it holds is invisible.
// The reference's own builder runs the same risk and gets away with it when its leaves happen to mix
// such triangles with others (the walls of the Cornell box do); a builder that splits by surface area
// isolates walls in leaves of their own.  A BVH's boxes may be larger than what they hold, so the GPU
// builders widen every box that is thinner than `pad` along an axis by `pad` on both sides of that
// axis, where pad = 2^-16 of the mesh's largest extent: a couple of hundred units in the last place of
// a coordinate of the size of the mesh, so the two quotients stay apart for any ray origin within a
// few mesh sizes, and far too little to let a ray into a box it should have missed by a visible margin.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define RT_PAD_HD __host__ __device__ inline
#else
#define RT_PAD_HD inline
#endif

namespace rt {

// `lo`, `hi`: the mesh's bounds (the root's box)
RT_PAD_HD float flat_box_pad(const float lo[3], const float hi[3]) {
    float ext = 0.0f, mag = 0.0f;
    for (int k = 0; k < 3; ++k) {
        const float e = hi[k] - lo[k], a = __builtin_fabsf(lo[k]), b = __builtin_fabsf(hi[k]);
        ext = e > ext ? e : ext;
        mag = a > mag ? a : mag;
        mag = b > mag ? b : mag;
    }
    const float scale = ext > 0.0f ? ext : (mag > 0.0f ? mag : 1.0f);
    return scale * 1.52587890625e-05f;  // 2^-16
}

RT_PAD_HD void widen_flat_box(float lo[3], float hi[3], float pad) {
    for (int k = 0; k < 3; ++k) {
        if (hi[k] - lo[k] < pad) {
            lo[k] -= pad;
            hi[k] += pad;
        }
    }
}

}  // namespace rt
