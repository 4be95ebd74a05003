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
    return ad >= kDivLo && ad <= kDivHi;
}

// true when v may be an operand of the subtraction that forms a numerator
RT_HD bool div_term_ok(float v) {
    float av = __builtin_fabsf(v);
    return v == 0.0f || (av >= kDivLo && av <= kDivHi);
}

// a / d given r = RN(1/d), under the range conditions above.
RT_HD float div_by_rcp(float a, float d, float r) {
    float q = a * r;
    float e = __builtin_fmaf(-q, d, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-q, d, a);
    return __builtin_fmaf(e, r, q);
}

}  // namespace rt
