// Host-side vector algebra in glam 0.30.5 operation order (the reference's
// math crate, Cargo.lock:710-716).  Everything here is plain IEEE binary32
// with no contraction, so a value computed on the host (an instance inverse, a
// triangle plane) has exactly the bits the reference's shader would compute
// for it per ray (shader/src/modules/trace.rs:87, hit.rs:145-148).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/rt_abi.h"

namespace rt {

struct V3 {
    float x, y, z;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 ld(const float* p) { return V3{p[0], p[1], p[2]}; }
inline void st(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) {
    float r = 1.0f / length(a);
    return a * r;
}
inline V3 vmin(V3 a, V3 b) { return V3{std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return V3{std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

// glam Affine3A::transform_vector3 / transform_point3
inline V3 transform_vector3(const rt_affine3a& m, V3 v) {
    return (ld(m.x_axis) * v.x + ld(m.y_axis) * v.y) + ld(m.z_axis) * v.z;
}
inline V3 transform_point3(const rt_affine3a& m, V3 v) { return transform_vector3(m, v) + ld(m.translation); }

// glam Mat3A::inverse followed by Affine3A::inverse: the reference evaluates
// this per ray per instance (trace.rs:87); it only depends on the instance.
inline rt_affine3a affine_inverse(const rt_affine3a& m) {
    V3 X = ld(m.x_axis), Y = ld(m.y_axis), Z = ld(m.z_axis), T = ld(m.translation);
    V3 t0 = cross(Y, Z), t1 = cross(Z, X), t2 = cross(X, Y);
    float det = dot(Z, t2);
    float inv_det = 1.0f / det;
    V3 c0 = t0 * inv_det, c1 = t1 * inv_det, c2 = t2 * inv_det;
    V3 ix{c0.x, c1.x, c2.x}, iy{c0.y, c1.y, c2.y}, iz{c0.z, c1.z, c2.z};
    V3 mt = (ix * T.x + iy * T.y) + iz * T.z;
    rt_affine3a out;
    std::memset(&out, 0, sizeof out);
    st(out.x_axis, ix);
    st(out.y_axis, iy);
    st(out.z_axis, iz);
    st(out.translation, -mt);
    return out;
}

// glam Mat3A::from_quat + Affine3A::from_scale_rotation_translation
// (program/src/main.rs:23-57 builds every transform of the default scene so).
inline rt_affine3a affine_from_srt(V3 scale, const float q[4], V3 translation) {
    float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    float xx = qx * x2, xy = qx * y2, xz = qx * z2;
    float yy = qy * y2, yz = qy * z2, zz = qz * z2;
    float wx = qw * x2, wy = qw * y2, wz = qw * z2;
    V3 rx{1.0f - (yy + zz), xy + wz, xz - wy};
    V3 ry{xy - wz, 1.0f - (xx + zz), yz + wx};
    V3 rz{xz + wy, yz - wx, 1.0f - (xx + yy)};
    rt_affine3a out;
    std::memset(&out, 0, sizeof out);
    st(out.x_axis, rx * scale.x);
    st(out.y_axis, ry * scale.y);
    st(out.z_axis, rz * scale.z);
    st(out.translation, translation);
    return out;
}

inline uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

}  // namespace rt
