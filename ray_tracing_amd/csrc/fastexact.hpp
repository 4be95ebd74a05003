// Correctly rounded square root, reciprocal and "n / 65535" in fewer instructions than the
// compiler's general expansions, for operands in a range where that can be CHECKED: each routine
// below is compared, on the device, against the compiler's IEEE result for every single operand
// of its range (rt_selftest_fastmath_device).  Outside the range the callers use the plain
// operators.  The compiler's own sequences spend most of their instructions on scaling
// denormals and on selecting special cases; inside the range neither is needed.
#pragma once

#include <cstdint>

namespace rt {

constexpr float kFxLo = 7.888609052210118e-31f;  // 2^-100
constexpr float kFxHi = 1.2676506002282294e30f;  // 2^100

// x is a positive, normal float in [2^-100, 2^100] (false for NaN)
__device__ __forceinline__ bool fx_ok(float x) { return (int(x >= kFxLo) & int(x <= kFxHi)) != 0; }

// sqrt(x) for fx_ok(x): the hardware estimate (1 ulp), then its neighbours' residuals decide
__device__ __forceinline__ float fx_sqrt(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_below = __builtin_fmaf(-below, s, x);
    const float r_above = __builtin_fmaf(-above, s, x);
    const float t = (r_below <= 0.0f) ? below : s;
    return (r_above > 0.0f) ? above : t;
}

// 1 / x for fx_ok(|x|): the hardware estimate (1 ulp) and one Newton step with exact residual
__device__ __forceinline__ float fx_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// n / 65535 for n = 0, 1, ..., 65535 (as a float): the divisor of rand_float, mod.rs:46-48
__device__ __forceinline__ float fx_div_65535(float n) {
    const float r = 1.0f / 65535.0f;  // folded at compile time, correctly rounded
    const float q = n * r;
    const float e = __builtin_fmaf(-q, 65535.0f, n);
    return __builtin_fmaf(e, r, q);
}

}  // namespace rt
