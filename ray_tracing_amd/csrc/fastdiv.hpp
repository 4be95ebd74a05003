// Correctly rounded binary32 division by a divisor whose correctly rounded
// reciprocal is already known.
//
// The slab test divides six numerators by the three components of the ray
// direction (shader/src/modules/trace.rs:209-210), and the direction does not
// change during a traversal.  With r = RN(1/d) computed once per traversal by
// an ordinary IEEE division,
//
//     q0 = RN(a * r)
//     q1 = RN(q0 + (a - q0*d) * r)      (the residual a - q0*d is exact in an FMA)
//     q2 = RN(q1 + (a - q1*d) * r)
//
// q1 is the rounding of a value within 2^-47 relative of a/d, hence within one
// ulp of it, and then q2 = RN(a/d) exactly (Markstein's division theorem: a
// faithful quotient corrected once with the correctly rounded reciprocal).
// The argument needs every intermediate to stay in the normal range.  That is
// guaranteed by range conditions on the OPERANDS OF THE SUBTRACTION that forms
// each numerator a = b - p (b a box coordinate, p a ray origin coordinate):
//     b == 0 or 2^-24 <= |b| <= 2^24      (checked once per scene, on the host)
//     p == 0 or 2^-24 <= |p| <= 2^24      (checked once per traversal)
//             2^-24 <= |d| <= 2^24        (checked once per traversal)
// Then a is 0 or 2^-47 <= |a| <= 2^25 (a non-zero difference of two such floats
// is a multiple of 2^-47), the quotient lies in [2^-71, 2^49], the residuals
// are at least 2^-95 in magnitude when not zero, and every product stays in
// the normal range.  A zero numerator
// gives a zero quotient (its sign may differ from IEEE's; only comparisons
// consume the quotients, so the sign of a zero cannot matter).  Outside these
// ranges callers use the plain division.  Both paths return the IEEE
// quotient, so which one ran cannot be observed in the results.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

constexpr float kDivLo = 5.9604644775390625e-08f;  // 2^-24
constexpr float kDivHi = 16777216.0f;              // 2^24

// true when v may be a divisor
RT_HD bool div_den_ok(float d) {
    float ad = __builtin_fabsf(d);
    return (int(ad >= kDivLo) & int(ad <= kDivHi)) != 0;
}

// true when v may be an operand of the subtraction that forms a numerator
RT_HD bool div_term_ok(float v) {
    float av = __builtin_fabsf(v);
    return (int(v == 0.0f) | (int(av >= kDivLo) & int(av <= kDivHi))) != 0;
}

// a / d given r = RN(1/d), under the range conditions above.
RT_HD float div_by_rcp(float a, float d, float r) {
    float q = a * r;
    float e = __builtin_fmaf(-q, d, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-q, d, a);
    return __builtin_fmaf(e, r, q);
}

// ---------------------------------------------------------------------------
// The slab test decided from approximate quotients.
//
// The slab test (trace.rs:208-229) only COMPARES its six quotients: with each
// other (t_near < t_far), with zero (t_far > 0) and, in the caller, with the
// clamp (t_near > t_hi).  Under the range conditions above, q = RN(a * r) has
// the sign of the IEEE quotient t = RN(a / d), is zero exactly when t is, and
// |q - t| < 2^-22 |q|  (r is within 2^-24 relative of 1/d, the product rounds
// once more, t is within 2^-24 of a/d).  Maxima and minima of such values keep
// that bound (the larger of two approximations is within 2^-22 of the larger
// of the two exact values), and the sign of min-of-max is preserved term by
// term.  So with T_near, T_far the exact values and tn, tf the approximate
// ones:
//     |tf - tn| > 2^-21 (|tn| + |tf|)   =>  (T_near < T_far) == (tn < tf)
//     (T_far > 0) == (tf > 0)           always
//     |tn - t_hi| > 2^-21 |tn|          =>  (T_near > t_hi) == (tn > t_hi)
// When either premise fails (ties included: a flat box, whose two planes give
// one and the same quotient, always lands here when the ray crosses it) the
// test is "ambiguous" and the caller evaluates the exact quotients instead.
// The margins carry a factor 2 of slack over the bound.
struct SlabApprox {
    float t_near, t_far;
};

constexpr float kSlabMargin = 4.76837158203125e-07f;  // 2^-21
// The same shortcut with reciprocals from the hardware's approximate instruction (v_rcp_f32, at
// most 1 ulp off: r within 2^-23 relative of 1/d) instead of correctly rounded ones: the bound
// becomes |q - t| < (2^-23 + 2^-24 + 2^-24 + second order) |q| < 2^-21.9 |q|, so comparisons are
// clear-cut beyond twice that; 2^-20 leaves the same factor of slack as above.
constexpr float kSlabMarginApproxRcp = 9.5367431640625e-07f;  // 2^-20

RT_HD SlabApprox slab_approx(float lox, float loy, float loz, float hix, float hiy, float hiz, float px, float py,
                             float pz, float rx, float ry, float rz) {
    float nx = (lox - px) * rx, ny = (loy - py) * ry, nz = (loz - pz) * rz;
    float fx = (hix - px) * rx, fy = (hiy - py) * ry, fz = (hiz - pz) * rz;
    SlabApprox s;
    s.t_near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(nx, fx), __builtin_fminf(ny, fy)), __builtin_fminf(nz, fz));
    s.t_far = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(nx, fx), __builtin_fmaxf(ny, fy)), __builtin_fmaxf(nz, fz));
    return s;
}

// Written so that a NaN anywhere (a NaN clamp) reads as ambiguous.
RT_HD bool slab_ambiguous(SlabApprox s, float t_hi, float margin = kSlabMargin) {
    float an = __builtin_fabsf(s.t_near), af = __builtin_fabsf(s.t_far);
    bool clear_order = __builtin_fabsf(s.t_far - s.t_near) > margin * (an + af);
    bool clear_clamp = __builtin_fabsf(s.t_near - t_hi) > margin * an;
    return !(clear_order && clear_clamp);
}

// The value hits_bounding would return, good for comparisons only: the entry
// distance when the box is hit, +inf when it is not.
RT_HD float slab_dist(SlabApprox s) {
    bool ordered = s.t_near < s.t_far, ahead = s.t_far > 0.0f;
    return (ordered && ahead) ? s.t_near : __builtin_huge_valf();
}

}  // namespace rt
