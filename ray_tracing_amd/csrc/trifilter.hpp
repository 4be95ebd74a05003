// The inside test of the ray/triangle intersection decided from a cheaper,
// approximate form, with the exact form as fallback.
//
// After the plane hit point h the reference evaluates, per edge k (from p_a to
// p_b), s_k = dot(n, cross(p_b - p_a, h - p_a)) in binary32 and rejects the hit
// when s_k < 0 (shader/src/modules/hit.rs:160-188).  Only the SIGNS of the three
// values are used.  In real arithmetic, with g = h - p0, m_k = n x (p_b - p_a):
//     s_0 = m_0 . g
//     s_1 = m_1 . g + c_1,   c_1 = m_1 . (p0 - p1)
//     s_2 = c_1 - s_0 - s_1  (the edges sum to zero and p0 lies on edge 2)
// The record below holds m_0, m_1, c_1 (computed in binary64 from the binary32
// inputs, rounded once), so the three signs cost 11 operations instead of 60.
//
// Error bound.  Let E >= the 1-norm of every edge and G = |g|_1.  The
// reference's binary32 evaluation of s_k differs from the real value by at
// most 7.1 u E (G + E) (u = 2^-24: two rounded inputs per product, the
// difference, and three more roundings in the dot product, with |n_i| <= 1);
// the approximate forms above differ from it by at most 17 u E G + 5 u E^2.
// Together that is below 2^-19.4 E (G + E).  With the margin
//     M = 2^-18 E (G + E)
// a value with |s_k| > M has the sign the reference computes.  Anything else
// (NaN and infinities included, the comparisons are written so) is "unclear"
// and the caller runs the exact test.  Triangles whose edges fall outside
// [2^-40, 2^40] get an infinite margin: they are always unclear.
#pragma once

#include <cmath>
#include <cstdint>

#include "device_types.hpp"
#include "fastdiv.hpp"  // RT_HD

namespace rt {

struct TriSigns {
    float s_min;      // smallest of the three approximate edge values
    float s_min_abs;  // smallest magnitude among them
    float margin;     // M
};

// g = h - p0, evaluated as the reference evaluates it for its first edge.
// (s0, s1) = the two dot products m0 . g and m1 . g, however they were formed.
RT_HD TriSigns tri_filter_signs(float s0, float s1_dot, float c1, float margin_scale, float e_len, float gx, float gy,
                                float gz) {
    float s1 = s1_dot + c1;
    float s2 = (c1 - s0) - s1;
    float G = (__builtin_fabsf(gx) + __builtin_fabsf(gy)) + __builtin_fabsf(gz);
    TriSigns r;
    r.s_min = __builtin_fminf(__builtin_fminf(s0, s1), s2);
    r.s_min_abs = __builtin_fminf(__builtin_fminf(__builtin_fabsf(s0), __builtin_fabsf(s1)), __builtin_fabsf(s2));
    r.margin = margin_scale * (G + e_len);
    return r;
}

RT_HD bool tri_filter_unclear(const TriSigns& r) { return !(r.s_min_abs > r.margin); }

#if !defined(__HIP_DEVICE_COMPILE__)
// Host: fill the record of one triangle from its binary32 vertices and the
// binary32 plane (normal, d) the exact test uses.
inline void tri_filter_build(const float p0[3], const float p1[3], const float p2[3], const float n[3], float d,
                             DevTriTest& o) {
    double e0[3], e1[3], e2[3], nn[3];
    for (int k = 0; k < 3; ++k) {
        e0[k] = double(p1[k]) - double(p0[k]);
        e1[k] = double(p2[k]) - double(p1[k]);
        e2[k] = double(p0[k]) - double(p2[k]);
        nn[k] = n[k];
    }
    auto cross = [](const double a[3], const double b[3], double out[3]) {
        out[0] = a[1] * b[2] - a[2] * b[1];
        out[1] = a[2] * b[0] - a[0] * b[2];
        out[2] = a[0] * b[1] - a[1] * b[0];
    };
    double m0[3], m1[3];
    cross(nn, e0, m0);
    cross(nn, e1, m1);
    double c1 = -(m1[0] * e0[0] + m1[1] * e0[1] + m1[2] * e0[2]);  // m1 . (p0 - p1)
    auto norm1 = [](const double a[3]) { return std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2]); };
    double E = std::fmax(norm1(e0), std::fmax(norm1(e1), norm1(e2)));
    bool ok = std::isfinite(E) && E >= 0x1p-40 && E <= 0x1p40 && std::isfinite(nn[0]) && std::isfinite(nn[1]) &&
              std::isfinite(nn[2]) && std::fabs(nn[0]) <= 1.001 && std::fabs(nn[1]) <= 1.001 && std::fabs(nn[2]) <= 1.001;
    for (int k = 0; k < 3; ++k) {
        o.n[k] = n[k];
        o.p0[k] = p0[k];
    }
    o.m01x[0] = float(m0[0]);
    o.m01x[1] = float(m1[0]);
    o.m01y[0] = float(m0[1]);
    o.m01y[1] = float(m1[1]);
    o.m01z[0] = float(m0[2]);
    o.m01z[1] = float(m1[2]);
    o.d = d;
    o.c1 = float(c1);
    float Ef = ok ? std::nextafter(float(E), INFINITY) : INFINITY;
    o.e_len = Ef;
    o.margin_scale = ok ? Ef * 0x1p-18f : INFINITY;
}
#endif

}  // namespace rt
