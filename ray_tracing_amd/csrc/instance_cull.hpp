// A cull above the instance list.
//
// trace_ray tries every instance in index order (shader/src/modules/trace.rs:76-95): the ray is moved
// into the instance's object space (inv . pos, inv . dir) and Mesh::hit's first act is the slab test of
// the object's root box (hit.rs:76-86): the instance is left at once when that test misses or its
// entry distance lies beyond the clamp.  For most (ray, instance) pairs that is all that happens, and it
// costs a matrix-vector product, three reciprocals and a slab test.  This file replaces it, for the
// pairs where the outcome is plain, by one slab test in WORLD space against a box stored with the
// instance: when the world-space ray clearly misses that box, the reference's object-space test is
// known to fail, and the instance is passed by.  Everything else goes the reference's way.  The
// reference's counters count the root box test either way (hit.rs:80).
//
// Why "clearly misses the stored box" implies "the reference's test fails".
//
// Write A, T' for the stored inverse transform (the twelve floats the kernels multiply with), B for
// the object's root box and U = { x : A x + T' in B } for its exact pre-image in world space.
//
// (1) What the reference computes is an exact slab test of a slightly different ray against a slightly
//     different box.  np = fl(A p + T') and nd = fl(A d) are sums of three or four products rounded after
//     every operation: |np - (A p + T')| <= 2^-21 (|A| |p| + |T'|) and |nd - A d| <= 2^-21 |A| |d|,
//     componentwise with absolute values.  Each quotient fl(fl(b - np_i) / nd_i) is the exact quotient
//     times (1 + theta), |theta| <= 2^-23, that is the exact quotient for a plane moved by at most
//     2^-23 (|b| + |np_i|); the comparisons that follow (t_near < t_far, t_far > 0, dist > t_hi) are exact.
//     Mapped back to world space by A^-1, origin, direction and planes are off by at most
//         e(t) = 2^-20 k (|p| + |t| |d| + S)        at the ray parameter t,
//     with k = |A|_inf |A^-1|_inf the condition number of the transform and S a bound on the world
//     coordinates of U (it also bounds |A^-1 T'|).  t is the same parameter in both spaces: the
//     reference does not renormalise nd (trace.rs:88-91).
// (2) Where the perturbed ray could touch the perturbed box, |t| |d| <= |p| + S + margin: the box is no
//     farther than that from the origin.  So everywhere it matters e(t) <= 2^-19 k (|p| + S).
// (3) Hence: if the EXACT world ray misses U inflated by m = 2^-19 k (|p| + S) on every side -- or
//     meets it only at negative t, or only beyond the clamp -- then the perturbed ray misses the
//     perturbed box, or meets it only at negative t, or only beyond the clamp: a point of the
//     perturbed ray inside the perturbed box at parameter t would put the exact ray's point at that t
//     inside the inflated U.  In each of the three cases the reference leaves the instance (hit.rs:83).
// (4) The stored box [wlo, whi] holds U (its eight corners, mapped by A^-1 in double precision,
//     rounded outwards) inflated by 2^-12 S, and is only made when k <= 16; the test below inflates it
//     by another 2^-12 max|p_i|.  Together at least 2^-12 (|p| + S) >= 8 m (128 m for a rigid motion
//     with uniform scale, k = 1): what is left over covers the rounding of the world-space test itself
//     (approximate reciprocals, one rounding per operation: planes off by less than 2^-20 (|p| + S),
//     1/256 of the inflation) many times over.
// (5) The test only ever says "clearly missed" through comparisons that are true; a NaN anywhere (a
//     non-finite ray, 0 * inf when an axis-parallel ray starts on a face plane of the inflated box and
//     the minimum / maximum pick the NaN) makes them false or leaves an infinity that belongs to a
//     real miss, so what cannot be decided goes the reference's way.
//
// rt_selftest_instance_cull draws transforms, boxes and rays (most of them aimed at the rim of the
// box) and counts pairs that were passed by although the reference's own arithmetic accepts them.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define RT_CULL_HD __host__ __device__ __forceinline__
#else
#define RT_CULL_HD inline
#endif

namespace rt {

constexpr float kCullEta = 0.000244140625f;  // 2^-12
constexpr double kCullMaxCondition = 16.0;

// The world-space slab test.  (wrx, wry, wrz): reciprocals of the world direction (approximate ones
// will do); m_p = kCullEta * max |p_i|.  True only when the ray misses the box inflated by m_p, or meets
// it only behind the origin, or only beyond t_hi.
RT_CULL_HD bool instance_clearly_missed(const float wlo[3], const float whi[3], float px, float py, float pz, float wrx,
                                        float wry, float wrz, float m_p, float t_hi) {
    const float ax = ((wlo[0] - m_p) - px) * wrx, bx = ((whi[0] + m_p) - px) * wrx;
    const float ay = ((wlo[1] - m_p) - py) * wry, by = ((whi[1] + m_p) - py) * wry;
    const float az = ((wlo[2] - m_p) - pz) * wrz, bz = ((whi[2] + m_p) - pz) * wrz;
    const float t_near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float t_far = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    return (int(t_near > t_far) | int(t_far < 0.0f) | int(t_near > t_hi)) != 0;
}

// Host: the stored box of an instance from its packed inverse transform (X, Y, Z, T' columns, as the
// kernels read it) and the object's root box.  False when no box can vouch for the instance.
inline bool instance_world_bound(const float inv[12], const float lo[3], const float hi[3], float wlo[3], float whi[3]) {
    for (int i = 0; i < 12; ++i)
        if (!std::isfinite(inv[i])) return false;
    for (int i = 0; i < 3; ++i)
        if (!std::isfinite(lo[i]) || !std::isfinite(hi[i]) || lo[i] > hi[i]) return false;
    // A = [X Y Z] (columns), object = A world + T'
    const double a[3][3] = {{inv[0], inv[3], inv[6]}, {inv[1], inv[4], inv[7]}, {inv[2], inv[5], inv[8]}};
    const double t[3] = {inv[9], inv[10], inv[11]};
    const double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
                 c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    const double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    double ai[3][3];  // A^-1
    ai[0][0] = c00 / det;
    ai[1][0] = c01 / det;
    ai[2][0] = c02 / det;
    ai[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
    ai[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det;
    ai[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
    ai[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
    ai[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
    ai[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
    double na = 0.0, nai = 0.0;  // row-sum norms
    for (int r = 0; r < 3; ++r) {
        na = std::fmax(na, std::fabs(a[r][0]) + std::fabs(a[r][1]) + std::fabs(a[r][2]));
        nai = std::fmax(nai, std::fabs(ai[r][0]) + std::fabs(ai[r][1]) + std::fabs(ai[r][2]));
    }
    if (!std::isfinite(na * nai) || na * nai > kCullMaxCondition) return false;
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int corner = 0; corner < 8; ++corner) {
        const double o[3] = {(corner & 1 ? hi[0] : lo[0]) - t[0], (corner & 2 ? hi[1] : lo[1]) - t[1],
                             (corner & 4 ? hi[2] : lo[2]) - t[2]};
        for (int r = 0; r < 3; ++r) {
            const double w = ai[r][0] * o[0] + ai[r][1] * o[1] + ai[r][2] * o[2];
            mn[r] = std::fmin(mn[r], w);
            mx[r] = std::fmax(mx[r], w);
        }
    }
    double s = 0.0;
    for (int r = 0; r < 3; ++r) s = std::fmax(s, std::fmax(std::fabs(mn[r]), std::fabs(mx[r])));
    // the translation A^-1 T' is a world point too (the image of the object-space origin)
    for (int r = 0; r < 3; ++r) s = std::fmax(s, std::fabs(ai[r][0] * t[0] + ai[r][1] * t[1] + ai[r][2] * t[2]));
    const double grow = (double)kCullEta * s;
    for (int r = 0; r < 3; ++r) {
        const double l = mn[r] - grow, h = mx[r] + grow;
        if (!std::isfinite(l) || !std::isfinite(h) || std::fabs(l) > 1e37 || std::fabs(h) > 1e37) return false;
        wlo[r] = std::nextafterf((float)l, -INFINITY);  // outwards whichever way the conversion rounded
        whi[r] = std::nextafterf((float)h, INFINITY);
    }
    return true;
}

}  // namespace rt
